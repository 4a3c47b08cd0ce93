// fic_capi.cpp -- C ABI (include/fic.h) over the gfx950 kernels.  Host-side orchestration only:
// validation, device buffers, launch order, result copies.  No compute happens on the CPU and
// there is no CPU fallback: without a HIP device every compute entry returns FIC_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fic.h"
#include "fic_device.h"
#include "fic_launch.h"

namespace {

thread_local std::string g_err;
thread_local int g_err_code = 0;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    g_err_code = code;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(FIC_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

int ilog2(int v)
{
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}

// Geometry as the reference derives it (FC:111-116, FC:1019-1022) + what it needs to not throw.
int make_geometry(int w, int h, int B, int wK, int n_iso, int planes, FicGeom* out)
{
    if (B != 4 && B != 8 && B != 16)
        return fail(FIC_E_GEOMETRY, "blockgroesse B=%d unsupported (GUI values 4, 8, 16; B=2 divides by zero at FC:1022)", B);
    if (w <= 0 || h <= 0 || (w % 2) || (h % 2))
        return fail(FIC_E_GEOMETRY, "image %dx%d: width and height must be positive and even (scaleImage FC:970-1007 overruns otherwise)", w, h);
    if ((w % B) || (h % B))
        return fail(FIC_E_GEOMETRY, "image %dx%d is not a multiple of B=%d (ArrayIndexOutOfBounds in the reference)", w, h, B);
    FicGeom g;
    memset(&g, 0, sizeof(g));
    g.W = w; g.H = h; g.B = B; g.n = B * B; g.lgn = ilog2(B * B);
    g.Ws = w / 2; g.Hs = h / 2; g.abstand = B / 4;
    g.Rw = w / B; g.Rh = h / B; g.Nr = g.Rw * g.Rh;
    g.Dw = g.Rw * 2 - 3; g.Dh = g.Rh * 2 - 3;
    if (g.Dw < 1 || g.Dh < 1)
        return fail(FIC_E_GEOMETRY, "image %dx%d with B=%d has no domain blocks (Dw=%d Dh=%d)", w, h, B, g.Dw, g.Dh);
    g.Nd = g.Dw * g.Dh;
    if ((long long)g.Nd * 8 >= 0x7FFFFFFFll || (long long)w * h >= 0x7FFFFFFFll)
        return fail(FIC_E_GEOMETRY, "image %dx%d too large for 32-bit candidate indices", w, h);
    if (out == nullptr) return FIC_OK;
    if (wK < 1 || wK > g.Dw || wK > g.Dh)
        return fail(FIC_E_WINDOW, "widthKernel wK=%d outside 1..min(Dw=%d,Dh=%d) (negative index at FC:145)", wK, g.Dw, g.Dh);
    if (n_iso != 1 && n_iso != 8) return fail(FIC_E_ARGUMENT, "n_iso=%d: only 1 (reference) or 8 (extension)", n_iso);
    if (planes < 1) return fail(FIC_E_ARGUMENT, "planes=%d", planes);
    g.wK = wK; g.n_iso = n_iso; g.planes = planes;
    g.DW = g.n / 4;
    int NR = 1, NC = 1;
    fic_fast_variant(B, n_iso, &NR, &NC);
    g.NR = NR;
    int tsz = 64 * NR;
    g.tiles = (g.Nr + tsz - 1) / tsz;
    g.Nr_pad = g.tiles * tsz;
    g.Nd_pad = g.Nd + FIC_POOL_PAD;
    g.full = (wK == g.Dw && wK == g.Dh) ? 1 : 0;
    *out = g;
    return FIC_OK;
}

}  // namespace

struct fic_ctx {
    int device = 0;
    FicGeom g;
    FicBuffers b;
    FicOutputs o;
    uint8_t* gray_own = nullptr;     // context-owned input copy
    int32_t* argb_stage = nullptr;   // staging for ARGB uploads
    int32_t* collage = nullptr;
    uint8_t* decoded = nullptr;      // decoder output image(s)
    FicDecodeState* dec_state = nullptr;   // decoder loop state [planes] and per-pixel squared changes [planes][W*H]
    uint32_t* dec_sq = nullptr;
    void* mfma_poolB = nullptr;      // opt-in matrix-core sweep: B fragments, A fragments, range constants
    void* mfma_rngA = nullptr;
    void* mfma_sw = nullptr;
    int* mfma_rconst = nullptr;
    int mfma_bf16 = 0;               // operand type the fragment stores were built for
    void* q_pool = nullptr;          // k_sweep_q ("sweep" = 6): A fragments, flat-tile flags, B fragments, error bounds, published theta
    void* q_flat = nullptr;
    void* q_rng = nullptr;
    void* q_E = nullptr;
    void* q_thg = nullptr;
    unsigned long long* q_stats = nullptr;   // "sweep_stats" = 1: device counters of k_sweep_q (fic_ctx_sweep_stats)
    uint32_t* d4_rng = nullptr;      // k_sweep_d4: range / domain slots of the group-Fourier form (n_iso = 8, B = 8 / 16)
    uint32_t* d4_pool = nullptr;
    bool have_input = false;
    bool encoded_any = false;
    hipStream_t last_stream = nullptr;
    hipStream_t own_stream = nullptr; // non-blocking stream of the multi-device entry (created on demand)
    int opt_sweep = 0, opt_chunks = 0, opt_time = 0;
    int last_chunks = 0, last_kind = 0;
    std::vector<hipEvent_t> ev;      // pairs start/stop
    double acc_ms = 0.0;
    int acc_n = 0;
    std::mutex mu;
};

namespace {

template <typename T>
int dev_alloc(T** p, size_t count)
{
    HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
    return FIC_OK;
}

int ctx_free_all(fic_ctx* c)
{
    (void)hipSetDevice(c->device);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    c->ev.clear();
    if (c->own_stream) { (void)hipStreamDestroy(c->own_stream); c->own_stream = nullptr; }
    void* ptrs[] = {c->gray_own, c->argb_stage, c->collage, c->decoded, c->dec_state, c->dec_sq, c->mfma_poolB, c->mfma_rngA, c->mfma_sw, c->mfma_rconst, c->q_pool, c->q_flat, c->q_rng, c->q_E, c->q_thg, c->q_stats, c->d4_rng, c->d4_pool, c->b.scaled, c->b.pool_pix, c->b.pool_st, c->b.pool_var,
                    c->b.pool_s64, c->b.rng_pix, c->b.rng_st, c->b.key, c->o.idx_local, c->o.idx_global, c->o.iso,
                    c->o.a, c->o.b, c->o.err, c->o.qrows, c->o.records};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    return FIC_OK;
}

int flush_events(fic_ctx* c)
{
    for (size_t i = 0; i + 1 < c->ev.size(); i += 2) {
        HIP_TRY(hipEventSynchronize(c->ev[i + 1]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        c->acc_ms += ms;
        c->acc_n += 1;
        (void)hipEventDestroy(c->ev[i]);
        (void)hipEventDestroy(c->ev[i + 1]);
    }
    c->ev.clear();
    return FIC_OK;
}

// Idle single-plane contexts of the one-shot entry points, most recently used last.
std::mutex g_cache_mu;
std::vector<fic_ctx*> g_cache;
constexpr size_t kCacheSlots = 16;   // the multi-device entry parks one context per device

fic_ctx* cache_take(int device, int w, int h, int B, int wK, int n_iso)
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (size_t i = g_cache.size(); i-- > 0;) {
        fic_ctx* c = g_cache[i];
        const FicGeom& g = c->g;
        if (c->device == device && g.W == w && g.H == h && g.B == B && g.wK == wK && g.n_iso == n_iso && g.planes == 1) {
            g_cache.erase(g_cache.begin() + (long)i);
            return c;
        }
    }
    return nullptr;
}

}  // namespace

extern "C" void fic_ctx_destroy(fic_ctx* c);

namespace {

void cache_give(fic_ctx* c)
{
    fic_ctx* evict = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        g_cache.push_back(c);
        if (g_cache.size() > kCacheSlots) {
            evict = g_cache.front();
            g_cache.erase(g_cache.begin());
        }
    }
    if (evict) fic_ctx_destroy(evict);
}


// ---- sweep orchestration helpers (used by fic_ctx_encode) ------------------------------------------------------

// "time_sweep": bracket the sweep launch with events on its stream; durations are read later by flush_events.
int time_begin(fic_ctx* c, hipStream_t s)
{
    if (!c->opt_time) return FIC_OK;
    if (c->ev.size() >= 256) {                     // bound the pending list: fold finished launches into the accumulators
        int rc = flush_events(c);
        if (rc) return rc;
    }
    hipEvent_t e0 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    c->ev.push_back(e0);
    HIP_TRY(hipEventRecord(e0, s));
    return FIC_OK;
}
int time_end(fic_ctx* c, hipStream_t s)
{
    if (!c->opt_time) return FIC_OK;
    hipEvent_t e1 = nullptr;
    HIP_TRY(hipEventCreate(&e1));
    c->ev.push_back(e1);
    HIP_TRY(hipEventRecord(e1, s));
    return FIC_OK;
}

// Pool chunks per range tile for the VALU sweeps (k_sweep_fast, k_sweep_d4).  Measured (profiles/r01l_chunk_sweep.txt):
// every chunk pays a start-up (its first block is evaluated exactly and tau restarts), so chunks stay >= 4096 blocks
// while aiming at ~48 K wave tasks for load balance; only a launch that could not otherwise fill the chip (one small
// image) splits finer.  `base_waves` = wave tasks per chunk.
int valu_chunk_count(const fic_ctx* c, long long base_waves)
{
    const FicGeom& g = c->g;
    int nchunks = c->opt_chunks;
    if (nchunks <= 0) {
        long long want = (49152 + base_waves - 1) / base_waves;
        long long cap = g.Nd / 4096;
        if (cap < 1) cap = 1;
        long long nc = want < cap ? want : cap;
        if (base_waves * nc < 4096) {
            long long cap2 = g.Nd / 512;
            if (cap2 < 1) cap2 = 1;
            long long fill = (4096 + base_waves - 1) / base_waves;
            nc = fill < cap2 ? fill : cap2;
        }
        nchunks = (int)(nc < 1 ? 1 : nc);
    }
    return nchunks > g.Nd ? g.Nd : nchunks;
}

// VALU sweep k_sweep_fast ("sweep" = 2) over tiles [tile0, tile0 + ntiles).
int valu_sweep(fic_ctx* c, int tile0, int ntiles, hipStream_t s, int* nchunks_out)
{
    const FicGeom& g = c->g;
    int NR, NC;
    fic_fast_variant(g.B, g.n_iso, &NR, &NC);
    int nchunks = valu_chunk_count(c, (long long)ntiles * (g.n_iso / NC) * g.planes);
    int chunk_len = (g.Nd + nchunks - 1) / nchunks;
    chunk_len = (chunk_len + 1) & ~1;              // even: the sweep consumes blocks in pairs
    nchunks = (g.Nd + chunk_len - 1) / chunk_len;
    if (fic_launch_sweep_fast(c->b, g, tile0, ntiles, chunk_len, nchunks, s)) return fail(FIC_E_HIP, "k_sweep_fast launch failed");
    *nchunks_out = nchunks;
    return FIC_OK;
}

// VALU sweep for n_iso = 8 with the isometries taken algebraically (k_sweep_d4, fic_d4.hip): slot prep + sweep.
int d4_available(const FicGeom& g) { return g.full && g.n_iso == 8 && g.NR == 1 && fic_d4_words(g.B) > 0; }
int d4_prep(fic_ctx* c, hipStream_t s)
{
    const FicGeom& g = c->g;
    const size_t NW = (size_t)fic_d4_words(g.B), P = (size_t)g.planes;
    if (!c->d4_rng) HIP_TRY(hipMalloc((void**)&c->d4_rng, P * g.tiles * NW * 64 * 4));
    if (!c->d4_pool) HIP_TRY(hipMalloc((void**)&c->d4_pool, P * g.Nd_pad * NW * 4));
    if (fic_launch_d4_prep(c->b, c->d4_rng, c->d4_pool, g, s)) return fail(FIC_E_HIP, "k_range_d4 / k_pool_d4 launch failed");
    return FIC_OK;
}
int d4_sweep(fic_ctx* c, int tile0, int ntiles, hipStream_t s, int* nchunks_out)
{
    const FicGeom& g = c->g;
    int nchunks = valu_chunk_count(c, (long long)ntiles * g.planes);
    const int chunk_len = (g.Nd + nchunks - 1) / nchunks;
    nchunks = (g.Nd + chunk_len - 1) / chunk_len;
    if (fic_launch_sweep_d4(c->b, c->d4_rng, c->d4_pool, g, tile0, ntiles, chunk_len, nchunks, s))
        return fail(FIC_E_HIP, "k_sweep_d4 launch failed");
    *nchunks_out = nchunks;
    return FIC_OK;
}

// Opt-in matrix-core sweeps: geometry of the fragment stores.
struct MatrixCoreShape {
    bool iso8;
    bool bf16;                       // bf16 operands (B = 4 / 8, B = 16 with 8 isometries; never with "sweep" = 4), else i8
    int steps;                       // MFMA steps per block: K = 16 (bf16) or K = 32 (i8)
    int ndtiles, ndtiles_alloc;      // domain tiles (x32 blocks), + 1 spare for the prefetch
    int nctiles_alloc;               // n_iso = 1: column tiles (x32 ranges), padded for the last workgroup
    int G8, ngroups8;                // n_iso = 8: range blocks per workgroup, number of groups
    int ct1;                         // n_iso = 1: column tiles per workgroup
};
MatrixCoreShape matrix_core_shape(const FicGeom& g, int kind)
{
    MatrixCoreShape m;
    m.iso8 = g.n_iso == 8;
    // bf16 operands wherever they are faster: B = 4/8, and B = 16 with 8 isometries (51 vs 79 ms at 4096x4096; with
    // 1 isometry the i8 kernel wins there, 8.6 vs 9.2 ms)
    m.bf16 = kind == 3 && (g.B <= 8 || g.n_iso == 8);
    m.steps = m.bf16 ? fic_bf16_steps(g.B) : (g.n <= 32 ? 1 : g.n / 32);
    m.ndtiles = (g.Nd + 31) / 32;
    m.ndtiles_alloc = m.ndtiles + 1;
    m.nctiles_alloc = g.Nr_pad / 32 + 32;
    m.G8 = m.bf16 ? fic_bf16_group8(g.B) : fic_mfma8_group(g.B);
    m.ngroups8 = (g.Nr_pad + m.G8 - 1) / m.G8;
    m.ct1 = m.bf16 ? fic_bf16_ct1(g.B) : fic_mfma1_ct(g.B);
    return m;
}
int matrix_core_prep(fic_ctx* c, int kind, hipStream_t s)
{
    const FicGeom& g = c->g;
    const MatrixCoreShape m = matrix_core_shape(g, kind);
    const size_t P = (size_t)g.planes;
    if ((c->mfma_poolB || c->mfma_rngA || c->mfma_sw || c->mfma_rconst) && c->mfma_bf16 != (int)m.bf16) {     // operand type changed ("sweep" 3 <-> 4): new fragment stores
        HIP_TRY(hipStreamSynchronize(s));
        void* old[] = {c->mfma_poolB, c->mfma_rngA, c->mfma_sw, c->mfma_rconst};
        for (void* p : old) if (p) (void)hipFree(p);
        c->mfma_poolB = c->mfma_rngA = c->mfma_sw = nullptr;
        c->mfma_rconst = nullptr;
    }
    {
        // every store is guarded on its own pointer: a failed hipMalloc leaves the others usable and is retried next time
        c->mfma_bf16 = (int)m.bf16;
        const size_t rtiles = m.iso8 ? (size_t)m.ngroups8 * (m.G8 / 4) : (size_t)m.nctiles_alloc;   // 32-row/column range tiles
        if (!c->mfma_poolB) HIP_TRY(hipMalloc(&c->mfma_poolB, P * m.ndtiles_alloc * m.steps * 64 * 16));
        if (!c->mfma_rngA) HIP_TRY(hipMalloc(&c->mfma_rngA, P * rtiles * m.steps * 64 * 16));
        if (!c->mfma_sw) HIP_TRY(hipMalloc(&c->mfma_sw, P * m.ndtiles_alloc * 32 * 8));
        if (!m.bf16 && !c->mfma_rconst)
            HIP_TRY(hipMalloc((void**)&c->mfma_rconst, m.iso8 ? P * rtiles * 16 * sizeof(int) : P * rtiles * 32 * 16));
    }
    if (m.bf16) {
        const int rtiles = m.iso8 ? m.ngroups8 * (m.G8 / 4) : m.nctiles_alloc;
        if (fic_launch_bf16_prep(c->b, c->mfma_poolB, c->mfma_sw, c->mfma_rngA, g, m.ndtiles_alloc, rtiles, s))
            return fail(FIC_E_HIP, "bf16 fragment prep launch failed");
    } else if (m.iso8) {
        if (fic_launch_mfma_prep_pool(c->b.pool_pix, c->mfma_poolB, g, m.ndtiles_alloc, s) ||
            fic_launch_mfma_prep_range(c->b.rng_pix, c->b.rng_st, c->mfma_rngA, c->mfma_rconst, g, m.ngroups8, s))
            return fail(FIC_E_HIP, "mfma prep launch failed");
    } else if (fic_launch_mfma1_prep(c->b, c->mfma_poolB, c->mfma_sw, c->mfma_rngA, c->mfma_rconst, g, m.ndtiles_alloc,
                                     m.nctiles_alloc, s)) {
        return fail(FIC_E_HIP, "mfma1 prep launch failed");
    }
    return FIC_OK;
}
int matrix_core_sweep(fic_ctx* c, int kind, int tile0, int tile1, hipStream_t s, int* nchunks_out)
{
    const FicGeom& g = c->g;
    const MatrixCoreShape m = matrix_core_shape(g, kind);
    const int tsz = 64 * g.NR;
    int nchunks = c->opt_chunks;
    if (nchunks <= 0) {
        const long long ranges = (long long)(tile1 - tile0) * tsz;
        const long long per_wg = m.iso8 ? m.G8 : 32LL * m.ct1;
        const long long base_wg = (ranges + per_wg - 1) / per_wg * g.planes;   // workgroups per chunk
        long long want = (4096 + base_wg - 1) / base_wg;                       // ~4 resident per CU x 256 CUs x 4
        long long cap = m.ndtiles / 256;                                       // >= 256 domain tiles per chunk: start-up < 10 %
        if (cap < 1) cap = 1;
        long long nc = want < cap ? want : cap;
        if (base_wg * nc < 1024) {                                             // one small image: fill the chip first,
            long long cap2 = m.ndtiles / 32;                                   // even at a higher per-chunk start-up share
            if (cap2 < 1) cap2 = 1;
            long long fill = (1024 + base_wg - 1) / base_wg;
            nc = fill < cap2 ? fill : cap2;
        }
        nchunks = (int)(nc < 1 ? 1 : nc);
    }
    if (nchunks > m.ndtiles) nchunks = m.ndtiles;
    const int tiles_per_chunk = (m.ndtiles + nchunks - 1) / nchunks;
    nchunks = (m.ndtiles + tiles_per_chunk - 1) / tiles_per_chunk;
    if (m.iso8) {
        const int g0 = (tile0 * tsz) / m.G8, g1 = (tile1 * tsz + m.G8 - 1) / m.G8;   // groups covering the tile span
        const int rc = m.bf16 ? fic_launch_sweep_bf16(c->b, c->mfma_poolB, c->mfma_sw, c->mfma_rngA, g, m.ngroups8 * (m.G8 / 4), g0,
                                                      g1 - g0, m.ndtiles, m.ndtiles_alloc, tiles_per_chunk, nchunks, s)
                              : fic_launch_sweep_mfma(c->b, c->mfma_poolB, c->mfma_rngA, c->mfma_rconst, g, m.ngroups8, g0,
                                                      g1 - g0, m.ndtiles, m.ndtiles_alloc, tiles_per_chunk, nchunks, s);
        if (rc) return fail(FIC_E_HIP, "8-isometry matrix-core sweep launch failed");
    } else {
        const int ct_begin = tile0 * (tsz / 32), ct_end = tile1 * (tsz / 32);
        const int rc = m.bf16 ? fic_launch_sweep_bf16_1(c->b, c->mfma_poolB, c->mfma_sw, c->mfma_rngA, g, ct_begin, ct_end,
                                                        m.ndtiles, m.ndtiles_alloc, m.nctiles_alloc, tiles_per_chunk, nchunks, s)
                              : fic_launch_sweep_mfma1(c->b, c->mfma_poolB, c->mfma_sw, c->mfma_rngA, c->mfma_rconst, g, ct_begin,
                                                       ct_end, m.ndtiles, m.ndtiles_alloc, m.nctiles_alloc, tiles_per_chunk,
                                                       nchunks, s);
        if (rc) return fail(FIC_E_HIP, "1-isometry matrix-core sweep launch failed");
    }
    *nchunks_out = nchunks;
    return FIC_OK;
}

// Default full-search sweep (fic_q.hip): shapes of its stores, fused prep (pool build + range prep + fragments), launch.
struct QShape {
    int ndtiles, ndtiles_alloc;      // domain tiles (x32 blocks), + zero tiles for the unrolled loop's overrun and prefetch
    int unroll;                      // unroll factor of the sweep loop
    int CT;                          // column tiles (x32 range copies) per workgroup
    int nct_alloc;                   // column tiles allocated (padded for the last workgroup)
};
QShape q_shape(const FicGeom& g)
{
    QShape q;
    q.ndtiles = (g.Nd + 31) / 32;
    q.unroll = fic_q_unroll(g.B, g.n_iso);
    q.ndtiles_alloc = q.ndtiles + 2 * q.unroll;
    q.CT = fic_q_ct(g.B);
    const int nct = g.Nr_pad * fic_q_cols_per_range(g.B, g.n_iso) / 32;
    q.nct_alloc = (nct + q.CT - 1) / q.CT * q.CT + q.CT;
    return q;
}
int q_prep(fic_ctx* c, int tile0, int tile1, hipStream_t s)
{
    const FicGeom& g = c->g;
    const QShape q = q_shape(g);
    const size_t P = (size_t)g.planes, NK = (size_t)g.n / 16;
    if (!c->q_pool) HIP_TRY(hipMalloc(&c->q_pool, P * q.ndtiles_alloc * NK * 64 * 16));
    if (!c->q_flat) HIP_TRY(hipMalloc(&c->q_flat, P * q.ndtiles_alloc * sizeof(uint32_t)));
    if (!c->q_rng) {                                   // zeroed once: column tiles past the last range group stay zero fragments
        HIP_TRY(hipMalloc(&c->q_rng, P * q.nct_alloc * NK * 64 * 16));
        HIP_TRY(hipMemsetAsync(c->q_rng, 0, P * q.nct_alloc * NK * 64 * 16, s));
    }
    if (!c->q_E) HIP_TRY(hipMalloc(&c->q_E, P * g.Nr_pad * sizeof(float)));
    if (!c->q_thg) HIP_TRY(hipMalloc(&c->q_thg, P * g.Nr_pad * sizeof(uint32_t)));
    const int tsz = 64 * g.NR;
    const int grp0 = tile0 * tsz / 64, grp1 = tile1 * tsz / 64;      // 64-range groups covering the span
    if (fic_launch_q_prep(c->b, c->q_pool, c->q_flat, c->q_rng, c->q_E, c->q_thg, g, q.ndtiles_alloc, q.nct_alloc, grp0, grp1 - grp0, s))
        return fail(FIC_E_HIP, "k_pool_q / k_range_q launch failed");
    return FIC_OK;
}
int q_sweep(fic_ctx* c, int tile0, int tile1, hipStream_t s, int* nchunks_out)
{
    const FicGeom& g = c->g;
    const QShape q = q_shape(g);
    const int tsz = 64 * g.NR;
    const int cpr = fic_q_cols_per_range(g.B, g.n_iso);
    const int ct_begin = tile0 * tsz * cpr / 32, ct_end = tile1 * tsz * cpr / 32;
    int nchunks = c->opt_chunks;
    if (nchunks <= 0) {
        // a chunk's start-up is about one domain tile of extra work per range (fic_q.hip), so chunks only need to be
        // long enough to hide that (>= 16 tiles) and numerous enough to fill the chip: the kernels run two workgroups per
        // CU (VGPR-bound), i.e. 512 at a time -- two rounds of them balance the tail
        const long long base_wg = (long long)((ct_end - ct_begin + q.CT - 1) / q.CT) * g.planes;
        long long want = (1024 + base_wg - 1) / base_wg;
        long long cap = q.ndtiles / 16;
        if (cap < 1) cap = 1;
        nchunks = (int)(want < cap ? want : cap);
    }
    if (nchunks > q.ndtiles) nchunks = q.ndtiles;
    int tiles_per_chunk = (q.ndtiles + nchunks - 1) / nchunks;
    tiles_per_chunk = (tiles_per_chunk + q.unroll - 1) / q.unroll * q.unroll;     // whole iterations of the unrolled sweep loop
    nchunks = (q.ndtiles + tiles_per_chunk - 1) / tiles_per_chunk;
    if (fic_launch_sweep_q(c->b, c->q_pool, c->q_flat, c->q_rng, c->q_E, c->q_thg, g, ct_begin, ct_end, q.ndtiles, q.ndtiles_alloc,
                           q.nct_alloc, tiles_per_chunk, nchunks, s, c->q_stats))
        return fail(FIC_E_HIP, "k_sweep_q launch failed");
    *nchunks_out = nchunks;
    return FIC_OK;
}

}  // namespace

extern "C" {

const char* fic_version(void) { return "fic-hip 0.1 (gfx950)"; }
const char* fic_last_error(void) { return g_err.c_str(); }
int fic_last_error_code(void) { return g_err_code; }

int fic_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fic_geometry(int w, int h, int B, int* Rw, int* Rh, int* Dw, int* Dh)
{
    FicGeom g;
    int rc = make_geometry(w, h, B, 1, 1, 1, &g);
    if (rc) return rc;
    if (Rw) *Rw = g.Rw;
    if (Rh) *Rh = g.Rh;
    if (Dw) *Dw = g.Dw;
    if (Dh) *Dh = g.Dh;
    return FIC_OK;
}

int fic_is_greyscale_argb(const int32_t* argb, int w, int h)
{
    if (!argb || w <= 0 || h <= 0) return fail(FIC_E_ARGUMENT, "fic_is_greyscale_argb: bad argument");
    size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; i++) {
        int r = (argb[i] >> 16) & 0xff, g = (argb[i] >> 8) & 0xff, b = argb[i] & 0xff;
        if (r != g || g != b) return 0;
    }
    return 1;
}

int64_t fic_write_run_gray(const int32_t* qrows, int n_ranges, int w, int h, int B, int wK, uint8_t* out,
                           int64_t capacity)
{
    if (!qrows || !out || n_ranges < 0) return fail(FIC_E_ARGUMENT, "fic_write_run_gray: bad argument");
    int64_t need = 20 + 12 * (int64_t)n_ranges;
    if (capacity < need) return fail(FIC_E_CAPACITY, "fic_write_run_gray: need %lld bytes, have %lld", (long long)need, (long long)capacity);
    auto put = [](uint8_t* p, int32_t v) {
        uint32_t u = (uint32_t)v;
        p[0] = (uint8_t)(u >> 24); p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u;
    };
    const int32_t hdr[5] = {0, w, h, B, wK};          // FC:234-238
    for (int i = 0; i < 5; i++) put(out + 4 * i, hdr[i]);
    uint8_t* p = out + 20;
    for (int64_t i = 0; i < 3 * (int64_t)n_ranges; i++, p += 4) put(p, qrows[i]);   // FC:241-245
    return need;
}

fic_ctx* fic_ctx_create(int device, int w, int h, int B, int wK, int n_iso, int planes)
{
    FicGeom g;
    if (make_geometry(w, h, B, wK, n_iso, planes, &g)) return nullptr;
    int ndev = fic_device_count();
    if (ndev <= 0) { fail(FIC_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)"); return nullptr; }
    if (device < 0 || device >= ndev) { fail(FIC_E_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail(FIC_E_HIP, "hipSetDevice(%d) failed", device); return nullptr; }
    fic_ctx* c = new fic_ctx();
    c->device = device;
    c->g = g;
    memset(&c->b, 0, sizeof(c->b));
    memset(&c->o, 0, sizeof(c->o));
    const size_t P = (size_t)planes;
    int rc = FIC_OK;
    auto A = [&](int r) { if (rc == FIC_OK) rc = r; };
    A(dev_alloc(&c->b.scaled, P * g.Ws * g.Hs));
    A(dev_alloc(&c->b.pool_pix, P * g.Nd_pad * g.n));
    A(dev_alloc(&c->b.pool_st, P * g.Nd_pad));
    A(dev_alloc(&c->b.pool_var, P * g.Nd_pad));
    A(dev_alloc(&c->b.pool_s64, P * g.Nd_pad));
    A(dev_alloc(&c->b.rng_st, P * g.Nr_pad));
    A(dev_alloc(&c->b.key, P * g.Nr_pad));
    A(dev_alloc(&c->o.idx_local, P * g.Nr));
    A(dev_alloc(&c->o.idx_global, P * g.Nr));
    A(dev_alloc(&c->o.iso, P * g.Nr));
    A(dev_alloc(&c->o.a, P * g.Nr));
    A(dev_alloc(&c->o.b, P * g.Nr));
    A(dev_alloc(&c->o.err, P * g.Nr));
    A(dev_alloc(&c->o.qrows, P * g.Nr * 3));
    A(dev_alloc(&c->o.records, P * g.Nr * 6));
    if (rc == FIC_OK) {
        // zero the pool once: the FIC_POOL_PAD tail blocks stay zero forever (prefetch over-read)
        hipError_t e = hipMemset(c->b.pool_pix, 0, P * g.Nd_pad * g.n);
        if (e == hipSuccess) e = hipMemset(c->b.pool_st, 0, P * g.Nd_pad * sizeof(FicDomStat));
        if (e == hipSuccess) e = hipMemset(c->b.pool_var, 0, P * g.Nd_pad * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(c->b.pool_s64, 0, P * g.Nd_pad * sizeof(double));
        if (e != hipSuccess) rc = fail(FIC_E_HIP, "hipMemset: %s", hipGetErrorString(e));
    }
    if (rc != FIC_OK) {
        ctx_free_all(c);
        delete c;
        return nullptr;
    }
    return c;
}

void fic_ctx_destroy(fic_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    ctx_free_all(c);
    delete c;
}

int fic_ctx_set_gray_host(fic_ctx* c, const uint8_t* gray)
{
    if (!c || !gray) return fail(FIC_E_ARGUMENT, "fic_ctx_set_gray_host: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    size_t bytes = (size_t)c->g.planes * c->g.W * c->g.H;
    if (!c->gray_own) { int rc = dev_alloc(&c->gray_own, bytes); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->last_stream));   // a previous encode on a non-blocking stream may still read gray_own
    HIP_TRY(hipMemcpy(c->gray_own, gray, bytes, hipMemcpyHostToDevice));
    c->b.gray = c->gray_own;
    c->have_input = true;
    return FIC_OK;
}

int fic_ctx_set_argb_host(fic_ctx* c, const int32_t* argb)
{
    if (!c || !argb) return fail(FIC_E_ARGUMENT, "fic_ctx_set_argb_host: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    size_t npix = (size_t)c->g.planes * c->g.W * c->g.H;
    if (!c->gray_own) { int rc = dev_alloc(&c->gray_own, npix); if (rc) return rc; }
    if (!c->argb_stage) { int rc = dev_alloc(&c->argb_stage, npix); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    HIP_TRY(hipMemcpy(c->argb_stage, argb, npix * sizeof(int32_t), hipMemcpyHostToDevice));
    if (fic_launch_argb_to_gray(c->argb_stage, c->gray_own, npix, nullptr)) return fail(FIC_E_HIP, "k_argb_to_gray launch failed");
    HIP_TRY(hipStreamSynchronize(nullptr));
    c->b.gray = c->gray_own;
    c->have_input = true;
    return FIC_OK;
}

int fic_ctx_set_gray_device(fic_ctx* c, const void* dev_gray)
{
    if (!c || !dev_gray) return fail(FIC_E_ARGUMENT, "fic_ctx_set_gray_device: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    c->b.gray = (uint8_t*)dev_gray;
    c->have_input = true;
    return FIC_OK;
}

int fic_ctx_encode(fic_ctx* c, int range_begin, int range_count, void* hip_stream)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_ctx_encode: null context");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_input) return fail(FIC_E_STATE, "fic_ctx_encode: no input image set");
    const FicGeom& g = c->g;
    if (range_count < 0) range_count = g.Nr - range_begin;
    if (range_begin < 0 || range_count < 0 || range_begin + range_count > g.Nr)
        return fail(FIC_E_ARGUMENT, "fic_ctx_encode: range span [%d,%d) outside 0..%d", range_begin, range_begin + range_count, g.Nr);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)hip_stream;
    c->last_stream = s;
    if (range_count == 0) return FIC_OK;

    // which sweep
    int kind = c->opt_sweep;
    if (kind == 0) {
        // Windowed search: the generic kernel.  Full search: the matrix-core sweep k_sweep_q (fic_q.hip) -- same codebook
        // bits as the VALU sweeps, several times faster (north_star's "no MFMA" premise is refuted by
        // the evidence it asked for: profiles/r01z_cfg2_default_pmc_summary.txt, 0.7 % of HBM peak, VALU busy 93.5 %).
        // Very small launches stay on the VALU sweep (no fragment prep).
        // FIC_SWEEP=<n> overrides the automatic choice process-wide wherever kernel n applies ("sweep" option wins);
        // FIC_SWEEP=5 means "the VALU-only default": k_sweep_d4 where it is built, else k_sweep_fast.
        const long long pairs = (long long)g.planes * range_count * g.Nd;
        const int valu = d4_available(g) ? 5 : 2;
        kind = !g.full ? 1 : (pairs >= 2000000LL ? 6 : valu);
        const char* env = getenv("FIC_SWEEP");
        if (env && env[0] >= '2' && env[0] <= '6' && env[1] == '\0' && g.full) {
            const int want = env[0] - '0';
            kind = want == 5 ? valu : want;
        }
    }
    if (kind >= 2 && !g.full) return fail(FIC_E_ARGUMENT, "fast sweep needs full search (wK == Dw == Dh)");
    if (kind == 5 && !d4_available(g)) return fail(FIC_E_ARGUMENT, "sweep 5 (k_sweep_d4) needs full search, n_iso = 8 and B = 8");
    const int tsz = 64 * g.NR;
    const int tile0 = range_begin / tsz;
    const int tile1 = (range_begin + range_count + tsz - 1) / tsz;
    const int ntiles = tile1 - tile0;
    int nchunks = 1;
    int rc = FIC_OK;
    // pool build (createCodebuch FC:119) + range prep
    if (fic_launch_scale(c->b.gray, c->b.scaled, g, s)) return fail(FIC_E_HIP, "k_scale launch failed");
    if (kind == 6) {
        // fused: k_pool_q = pool + statistics + A fragments, k_range_q = range statistics + key reset + copies + B fragments
        rc = q_prep(c, tile0, tile1, s);
    } else {
        if (fic_launch_pool(c->b.scaled, c->b.pool_pix, c->b.pool_st, c->b.pool_var, c->b.pool_s64, g, s))
            return fail(FIC_E_HIP, "k_pool launch failed");
        // the lane-transposed store of range blocks + isometry copies: only these sweeps read it (134 MB at 4096x4096 with
        // 8 isometries), so it is allocated on their first use, not with the context
        if (kind != 5 && !c->b.rng_pix) {
            int rca = dev_alloc(&c->b.rng_pix, (size_t)g.planes * g.Nr_pad * g.n_iso * g.DW);
            if (rca) return rca;
        }
        // k_sweep_d4 reads its own slot store and the finaliser reads the image: no isometry copies to build then
        if (fic_launch_range(c->b.gray, c->b.rng_pix, c->b.rng_st, g, s, kind == 5 ? 0 : 1)) return fail(FIC_E_HIP, "k_range launch failed");
        // one 2-D fill: rows = planes (pitch Nr_pad keys), width = the tile span of this shard
        HIP_TRY(hipMemset2DAsync(c->b.key + (size_t)tile0 * tsz, (size_t)g.Nr_pad * sizeof(unsigned long long), 0xFF,
                                 (size_t)ntiles * tsz * sizeof(unsigned long long), (size_t)g.planes, s));
        if (kind == 5) rc = d4_prep(c, s);
        else if (kind >= 3) rc = matrix_core_prep(c, kind, s);          // fragment prep belongs to pool build / range prep: not timed
    }
    if (rc == FIC_OK) rc = time_begin(c, s);
    if (rc == FIC_OK) {
        if (kind == 1) {
            if (fic_launch_sweep_generic(c->b, g, range_begin, range_count, s)) rc = fail(FIC_E_HIP, "k_sweep_generic launch failed");
        } else if (kind == 6) {
            rc = q_sweep(c, tile0, tile1, s, &nchunks);
        } else if (kind == 5) {
            rc = d4_sweep(c, tile0, tile1 - tile0, s, &nchunks);
        } else if (kind >= 3) {
            rc = matrix_core_sweep(c, kind, tile0, tile1, s, &nchunks);
        } else {
            rc = valu_sweep(c, tile0, tile1 - tile0, s, &nchunks);
        }
    }
    if (rc == FIC_OK) rc = time_end(c, s);
    if (rc != FIC_OK) return rc;
    c->last_chunks = nchunks;
    c->last_kind = kind;
    // kinds 5 and 6 build no lane-transposed isometry copies: the finaliser recomputes the winner's covariance from the image
    if (fic_launch_finalize(c->b, c->o, g, range_begin, range_count, s, (kind == 5 || kind == 6) ? 1 : 0)) return fail(FIC_E_HIP, "k_finalize launch failed");
    c->encoded_any = true;
    return FIC_OK;
}

int fic_ctx_sync(fic_ctx* c)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_ctx_sync: null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    return FIC_OK;
}

int fic_ctx_get_results_host(fic_ctx* c, int32_t* idx_local, float* a, float* b, int32_t* iso, int32_t* qrows,
                             int32_t* idx_global, float* err)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_ctx_get_results_host: null context");
    if (!c->encoded_any) return fail(FIC_E_STATE, "fic_ctx_get_results_host: nothing encoded yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    size_t n = (size_t)c->g.planes * c->g.Nr;
    if (idx_local) HIP_TRY(hipMemcpy(idx_local, c->o.idx_local, n * 4, hipMemcpyDeviceToHost));
    if (a) HIP_TRY(hipMemcpy(a, c->o.a, n * 4, hipMemcpyDeviceToHost));
    if (b) HIP_TRY(hipMemcpy(b, c->o.b, n * 4, hipMemcpyDeviceToHost));
    if (iso) HIP_TRY(hipMemcpy(iso, c->o.iso, n * 4, hipMemcpyDeviceToHost));
    if (qrows) HIP_TRY(hipMemcpy(qrows, c->o.qrows, n * 12, hipMemcpyDeviceToHost));
    if (idx_global) HIP_TRY(hipMemcpy(idx_global, c->o.idx_global, n * 4, hipMemcpyDeviceToHost));
    if (err) HIP_TRY(hipMemcpy(err, c->o.err, n * 4, hipMemcpyDeviceToHost));
    return FIC_OK;
}

int fic_ctx_result_device_ptrs(fic_ctx* c, void** idx_local, void** a, void** b, void** iso, void** qrows,
                               void** idx_global, void** err)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_ctx_result_device_ptrs: null context");
    if (idx_local) *idx_local = c->o.idx_local;
    if (a) *a = c->o.a;
    if (b) *b = c->o.b;
    if (iso) *iso = c->o.iso;
    if (qrows) *qrows = c->o.qrows;
    if (idx_global) *idx_global = c->o.idx_global;
    if (err) *err = c->o.err;
    return FIC_OK;
}

int fic_ctx_records_device_ptr(fic_ctx* c, void** records)
{
    if (!c || !records) return fail(FIC_E_ARGUMENT, "fic_ctx_records_device_ptr: null argument");
    *records = c->o.records;
    return FIC_OK;
}

int fic_ctx_collage_host(fic_ctx* c, int32_t* argb_out)
{
    if (!c || !argb_out) return fail(FIC_E_ARGUMENT, "fic_ctx_collage_host: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->encoded_any) return fail(FIC_E_STATE, "fic_ctx_collage_host: nothing encoded yet");
    HIP_TRY(hipSetDevice(c->device));
    size_t npix = (size_t)c->g.planes * c->g.W * c->g.H;
    if (!c->collage) { int rc = dev_alloc(&c->collage, npix); if (rc) return rc; }
    if (fic_launch_collage(c->b, c->o, c->collage, c->g, c->last_stream)) return fail(FIC_E_HIP, "k_collage launch failed");
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    HIP_TRY(hipMemcpy(argb_out, c->collage, npix * sizeof(int32_t), hipMemcpyDeviceToHost));
    return FIC_OK;
}

int fic_ctx_set_option(fic_ctx* c, const char* name, int value)
{
    if (!c || !name) return fail(FIC_E_ARGUMENT, "fic_ctx_set_option: null argument");
    if (!strcmp(name, "sweep")) {
        if (value < 0 || value > 6)
            return fail(FIC_E_ARGUMENT, "sweep must be 0 (auto), 1 (generic), 2 (fast, VALU), 3 (matrix-core, exact covariances), 4 (matrix-core, i8 operands), 5 (VALU, group-Fourier isometries) or 6 (matrix-core, normalised f16 prune GEMM)");
        c->opt_sweep = value;
    } else if (!strcmp(name, "chunks")) {
        if (value < 0) return fail(FIC_E_ARGUMENT, "chunks must be >= 0");
        c->opt_chunks = value;
    } else if (!strcmp(name, "time_sweep")) {
        c->opt_time = value ? 1 : 0;
    } else if (!strcmp(name, "sweep_stats")) {
        HIP_TRY(hipSetDevice(c->device));
        if (value && !c->q_stats) {
            HIP_TRY(hipMalloc((void**)&c->q_stats, 4 * sizeof(unsigned long long)));
            HIP_TRY(hipMemset(c->q_stats, 0, 4 * sizeof(unsigned long long)));
        } else if (!value && c->q_stats) {
            HIP_TRY(hipStreamSynchronize(c->last_stream));
            (void)hipFree(c->q_stats);
            c->q_stats = nullptr;
        }
    } else {
        return fail(FIC_E_ARGUMENT, "unknown option '%s'", name);
    }
    return FIC_OK;
}

int fic_ctx_sweep_time(fic_ctx* c, double* total_ms, int* launches, int reset)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_ctx_sweep_time: null context");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int rc = flush_events(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->acc_ms;
    if (launches) *launches = c->acc_n;
    if (reset) { c->acc_ms = 0.0; c->acc_n = 0; }
    return FIC_OK;
}

int fic_ctx_sweep_stats(fic_ctx* c, uint64_t* out4, int reset)
{
    if (!c || !out4) return fail(FIC_E_ARGUMENT, "fic_ctx_sweep_stats: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->q_stats) return fail(FIC_E_STATE, "fic_ctx_sweep_stats: set the option \"sweep_stats\" first");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    HIP_TRY(hipMemcpy(out4, c->q_stats, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(c->q_stats, 0, 4 * sizeof(uint64_t)));
    return FIC_OK;
}

int fic_ctx_info(fic_ctx* c, int* out10)
{
    if (!c || !out10) return fail(FIC_E_ARGUMENT, "fic_ctx_info: null argument");
    const FicGeom& g = c->g;
    int v[10] = {g.Rw, g.Rh, g.Nr, g.Dw, g.Dh, g.Nd, g.NR, g.tiles, c->last_chunks, c->last_kind};
    memcpy(out10, v, sizeof(v));
    return FIC_OK;
}

int fic_debug_sqrt_f64(int device, uint32_t first, uint32_t count, double* out)
{
    if (!out || count == 0) return fail(FIC_E_ARGUMENT, "fic_debug_sqrt_f64: bad argument");
    int ndev = fic_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return fail(FIC_E_NO_DEVICE, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    double* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (size_t)count * sizeof(double)));
    int rc = FIC_OK;
    if (fic_launch_sqrt_probe(d, first, count, nullptr)) rc = fail(FIC_E_HIP, "k_sqrt_probe launch failed");
    if (rc == FIC_OK) {
        hipError_t e = hipMemcpy(out, d, (size_t)count * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(FIC_E_HIP, "hipMemcpy: %s", hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}

int fic_ctx_debug_pool_host(fic_ctx* c, uint8_t* pix, uint32_t* sum, uint32_t* var, uint8_t* scaled)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_ctx_debug_pool_host: null context");
    if (!c->encoded_any) return fail(FIC_E_STATE, "fic_ctx_debug_pool_host: nothing encoded yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    const FicGeom& g = c->g;
    for (int p = 0; p < g.planes; p++) {
        if (pix)
            HIP_TRY(hipMemcpy(pix + (size_t)p * g.Nd * g.n, c->b.pool_pix + (size_t)p * g.Nd_pad * g.n, (size_t)g.Nd * g.n,
                              hipMemcpyDeviceToHost));
        if (var)
            HIP_TRY(hipMemcpy(var + (size_t)p * g.Nd, c->b.pool_var + (size_t)p * g.Nd_pad, (size_t)g.Nd * 4,
                              hipMemcpyDeviceToHost));
        if (sum) {
            std::vector<FicDomStat> st(g.Nd);
            HIP_TRY(hipMemcpy(st.data(), c->b.pool_st + (size_t)p * g.Nd_pad, (size_t)g.Nd * sizeof(FicDomStat),
                              hipMemcpyDeviceToHost));
            for (int i = 0; i < g.Nd; i++) sum[(size_t)p * g.Nd + i] = st[i].sum;
        }
    }
    if (scaled) HIP_TRY(hipMemcpy(scaled, c->b.scaled, (size_t)g.planes * g.Ws * g.Hs, hipMemcpyDeviceToHost));
    return FIC_OK;
}

// ---- decoder (decodeGreyScale FC:356-421) ------------------------------------------------------
namespace {
// Device arenas of the stream decoders, kept between calls (the GUI decodes after every encode, CTL:178-179): one
// allocation per (device, size class) instead of four hipMalloc/hipFree per call.  fic_release_cache() frees them.
struct Arena {
    int device = -1;
    size_t bytes = 0;
    char* base = nullptr;
};
std::mutex g_arena_mu;
std::vector<Arena> g_arenas;
constexpr size_t kArenaSlots = 4;

int arena_take(int device, size_t bytes, Arena* out)
{
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        for (size_t i = g_arenas.size(); i-- > 0;)
            if (g_arenas[i].device == device && g_arenas[i].bytes >= bytes && g_arenas[i].bytes <= 2 * bytes + (1u << 20)) {
                *out = g_arenas[i];
                g_arenas.erase(g_arenas.begin() + (long)i);
                return FIC_OK;
            }
    }
    out->device = device;
    out->bytes = bytes;
    HIP_TRY(hipMalloc((void**)&out->base, bytes));
    return FIC_OK;
}
void arena_give(const Arena& a)
{
    Arena evict;
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        g_arenas.push_back(a);
        if (g_arenas.size() <= kArenaSlots) return;
        evict = g_arenas.front();
        g_arenas.erase(g_arenas.begin());
    }
    (void)hipSetDevice(evict.device);
    (void)hipFree(evict.base);
}
void arena_release_all()
{
    std::vector<Arena> drop;
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        drop.swap(g_arenas);
    }
    for (const Arena& a : drop) { (void)hipSetDevice(a.device); (void)hipFree(a.base); }
}
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

// Runs the reconstruction loop on the device.  Iterations are enqueued in groups of 8 and the
// per-plane loop state is read back after each group (a converging decode takes 6-7 iterations),
// so there is one host sync per group, none per iteration.
//   d_state [planes], d_sqbuf u32 [planes][W*H]: scratch of the caller
static int run_decode_loop(const FicGeom& g, uint8_t* d_scaled, uint8_t* d_image, const int32_t* d_qrows,
                           const int32_t* d_iso, FicDecodeState* d_state, uint32_t* d_sqbuf, const float* avg_in,
                           float* avg_out, int* iters_out, int* seq_out, hipStream_t s)
{
    const size_t P = (size_t)g.planes;
    std::vector<FicDecodeState> st(P);
    memset(st.data(), 0, P * sizeof(FicDecodeState));
    for (size_t p = 0; p < P; p++) st[p].avg = avg_in ? avg_in[p] : 0.0f;   // static avgError is never reset (FC:20)
    int rc = FIC_OK;
    hipError_t e = hipMemcpyAsync(d_state, st.data(), P * sizeof(FicDecodeState), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_image, 128, P * g.W * g.H, s);        // generateGrayImage FC:1142-1148
    if (e != hipSuccess) rc = fail(FIC_E_HIP, "decode init: %s", hipGetErrorString(e));
    for (int counter = 0; rc == FIC_OK && counter < 50; counter++) {
        if (fic_launch_decode_iteration(d_scaled, d_image, d_qrows, d_iso, d_state, d_sqbuf, counter, g, s)) {
            rc = fail(FIC_E_HIP, "decode iteration launch failed");
            break;
        }
        if ((counter & 7) == 7 || counter == 49) {
            e = hipMemcpyAsync(st.data(), d_state, P * sizeof(FicDecodeState), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { rc = fail(FIC_E_HIP, "decode readback: %s", hipGetErrorString(e)); break; }
            bool all = true;
            for (size_t p = 0; p < P; p++) all = all && st[p].done;
            if (all) break;
        }
    }
    if (rc != FIC_OK) return rc;
    for (size_t p = 0; p < P; p++) {
        if (st[p].bad_index)
            return fail(FIC_E_ARGUMENT, "decode: a codebook row of plane %zu points outside the domain pool "
                                        "(ArrayIndexOutOfBounds at FC:394 in the reference)", p);
        if (avg_out) avg_out[p] = st[p].avg_out;
        if (iters_out) iters_out[p] = st[p].iters;
        if (seq_out) seq_out[p] = st[p].seq_sums;
    }
    return FIC_OK;
}

int fic_ctx_decode_host(fic_ctx* c, uint8_t* gray_out, float* avg_error_out, int* iterations_out)
{
    if (!c || !gray_out) return fail(FIC_E_ARGUMENT, "fic_ctx_decode_host: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->encoded_any) return fail(FIC_E_STATE, "fic_ctx_decode_host: nothing encoded yet");
    HIP_TRY(hipSetDevice(c->device));
    const FicGeom& g = c->g;
    size_t npix = (size_t)g.planes * g.W * g.H;
    if (!c->decoded) { int rc = dev_alloc(&c->decoded, npix); if (rc) return rc; }
    if (!c->dec_state) { int rc = dev_alloc(&c->dec_state, (size_t)g.planes); if (rc) return rc; }
    if (!c->dec_sq) { int rc = dev_alloc(&c->dec_sq, npix); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    int rc = run_decode_loop(g, c->b.scaled, c->decoded, c->o.qrows, g.n_iso > 1 ? c->o.iso : nullptr, c->dec_state, c->dec_sq,
                             nullptr, avg_error_out, iterations_out, nullptr, c->last_stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(gray_out, c->decoded, npix, hipMemcpyDeviceToHost));
    return FIC_OK;
}

static int32_t run_be32(const uint8_t* run, int64_t off)
{
    return (int32_t)(((uint32_t)run[off] << 24) | ((uint32_t)run[off + 1] << 16) | ((uint32_t)run[off + 2] << 8) |
                     (uint32_t)run[off + 3]);
}

static int decode_gray_run_impl(const uint8_t* run, int64_t len, int device, uint8_t* gray_out, int64_t capacity, int* w_out,
                                int* h_out, float* avg_error_io, int* iterations, int* seq_sums)
{
    if (!run || len < 20) return fail(FIC_E_ARGUMENT, "fic_decode_gray_run: stream shorter than the 20-byte header");
    if (run_be32(run, 0) != 0)
        return fail(FIC_E_NOT_GREY, "fic_decode_gray_run: isRGB = %d (FC:548-552 dispatches to decodeRGB)", run_be32(run, 0));
    const int w = run_be32(run, 4), h = run_be32(run, 8), B = run_be32(run, 12), wK = run_be32(run, 16);
    FicGeom g;
    int rc = make_geometry(w, h, B, wK, 1, 1, &g);
    if (rc) return rc;
    if (w_out) *w_out = w;
    if (h_out) *h_out = h;
    if (len < 20 + 12 * (int64_t)g.Nr)
        return fail(FIC_E_ARGUMENT, "fic_decode_gray_run: %lld bytes, need %lld (EOFException in the reference)",
                    (long long)len, (long long)(20 + 12 * (int64_t)g.Nr));
    if (!gray_out || capacity < (int64_t)w * h) return fail(FIC_E_CAPACITY, "fic_decode_gray_run: output needs %d bytes", w * h);
    int ndev = fic_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return fail(FIC_E_NO_DEVICE, "no HIP device %d (this library has no CPU path)", device);
    HIP_TRY(hipSetDevice(device));
    std::vector<int32_t> q((size_t)g.Nr * 3);
    for (size_t i = 0; i < q.size(); i++) q[i] = run_be32(run, 20 + 4 * (int64_t)i);          // FC:372-374
    const size_t npix = (size_t)w * h;
    const size_t o_scaled = 0, o_image = o_scaled + align256((size_t)g.Ws * g.Hs), o_q = o_image + align256(npix),
                 o_state = o_q + align256(q.size() * 4), o_sq = o_state + align256(sizeof(FicDecodeState)),
                 total = o_sq + align256(npix * 4);
    Arena ar;
    rc = arena_take(device, total, &ar);
    if (rc) return rc;
    hipError_t e = hipMemcpy(ar.base + o_q, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(FIC_E_HIP, "fic_decode_gray_run: %s", hipGetErrorString(e));
    float avg = avg_error_io ? *avg_error_io : 0.0f;
    if (rc == FIC_OK)
        rc = run_decode_loop(g, (uint8_t*)(ar.base + o_scaled), (uint8_t*)(ar.base + o_image), (const int32_t*)(ar.base + o_q), nullptr,
                             (FicDecodeState*)(ar.base + o_state), (uint32_t*)(ar.base + o_sq), &avg, &avg, iterations, seq_sums, nullptr);
    if (rc == FIC_OK) {
        e = hipMemcpy(gray_out, ar.base + o_image, npix, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(FIC_E_HIP, "fic_decode_gray_run: %s", hipGetErrorString(e));
    }
    if (rc == FIC_OK && avg_error_io) *avg_error_io = avg;
    arena_give(ar);
    return rc;
}

int fic_decode_gray_run(const uint8_t* run, int64_t len, int device, uint8_t* gray_out, int64_t capacity, int* w_out,
                        int* h_out, float* avg_error_io, int* iterations)
{
    return decode_gray_run_impl(run, len, device, gray_out, capacity, w_out, h_out, avg_error_io, iterations, nullptr);
}

// Test hook: the decoder's reproduction of Java's `avgError += (float) v[i]` loop (FC:407) on arbitrary values.
int fic_debug_float_sum(int device, float carry, const uint32_t* vals, int count, float* out)
{
    if (!vals || !out || count < 0) return fail(FIC_E_ARGUMENT, "fic_debug_float_sum: bad argument");
    int ndev = fic_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return fail(FIC_E_NO_DEVICE, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    uint32_t* d = nullptr;
    float* r = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (size_t)(count + 4) * 4));
    hipError_t e = hipMalloc((void**)&r, 4);
    if (e == hipSuccess) e = hipMemcpy(d, vals, (size_t)count * 4, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? FIC_OK : fail(FIC_E_HIP, "fic_debug_float_sum: %s", hipGetErrorString(e));
    if (rc == FIC_OK && fic_launch_float_sum_probe(carry, d, count, r, nullptr)) rc = fail(FIC_E_HIP, "k_float_sum_probe launch failed");
    if (rc == FIC_OK) {
        e = hipMemcpy(out, r, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(FIC_E_HIP, "fic_debug_float_sum: %s", hipGetErrorString(e));
    }
    (void)hipFree(d);
    if (r) (void)hipFree(r);
    return rc;
}

// Test hook: fic_decode_gray_run that also reports how many iterations needed the sequential (Java-order) float sum.
int fic_debug_decode_gray_run(const uint8_t* run, int64_t len, int device, uint8_t* gray_out, int64_t capacity,
                              float* avg_error_io, int* iterations, int* seq_sums)
{
    return decode_gray_run_impl(run, len, device, gray_out, capacity, nullptr, nullptr, avg_error_io, iterations, seq_sums);
}

// ---- decodeRGB (FC:430-508) -----------------------------------------------------------------------
int fic_decode_rgb_run(const uint8_t* run, int64_t len, int device, int32_t* argb_out, int64_t capacity_pixels,
                       int* w_out, int* h_out, float* avg_error_io, int* iterations)
{
    if (!run || len < 20) return fail(FIC_E_ARGUMENT, "fic_decode_rgb_run: stream shorter than the 20-byte header");
    if (run_be32(run, 0) == 0) return fail(FIC_E_ARGUMENT, "fic_decode_rgb_run: isRGB = 0 (FC:548-550 dispatches to decodeGreyScale)");
    const int w = run_be32(run, 4), h = run_be32(run, 8), B = run_be32(run, 12), wK = run_be32(run, 16);
    FicGeom g;
    int rc = make_geometry(w, h, B, wK, 1, 1, &g);
    if (rc) return rc;
    if (w_out) *w_out = w;
    if (h_out) *h_out = h;
    if (len < 20 + 20 * (int64_t)g.Nr)
        return fail(FIC_E_ARGUMENT, "fic_decode_rgb_run: %lld bytes, need %lld (EOFException in the reference)",
                    (long long)len, (long long)(20 + 20 * (int64_t)g.Nr));
    if (!argb_out || capacity_pixels < (int64_t)w * h) return fail(FIC_E_CAPACITY, "fic_decode_rgb_run: output needs %d ints", w * h);
    int ndev = fic_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return fail(FIC_E_NO_DEVICE, "no HIP device %d (this library has no CPU path)", device);
    HIP_TRY(hipSetDevice(device));
    std::vector<int32_t> q((size_t)g.Nr * 5);
    for (size_t i = 0; i < q.size(); i++) q[i] = run_be32(run, 20 + 4 * (int64_t)i);          // FC:446-450
    const size_t npix = (size_t)w * h;
    std::vector<int32_t> init(npix, (int32_t)0xff808080u);                          // generateGrayImage FC:1142-1148
    const size_t o_scaled = 0, o_image = o_scaled + align256((size_t)g.Ws * g.Hs * 4), o_q = o_image + align256(npix * 4),
                 o_state = o_q + align256(q.size() * 4), o_sq = o_state + align256(sizeof(FicDecodeState)),
                 total = o_sq + align256(npix * 4);
    Arena ar;
    rc = arena_take(device, total, &ar);
    if (rc) return rc;
    int32_t* d_scaled = (int32_t*)(ar.base + o_scaled);
    int32_t* d_image = (int32_t*)(ar.base + o_image);
    int32_t* d_q = (int32_t*)(ar.base + o_q);
    FicDecodeState* d_state = (FicDecodeState*)(ar.base + o_state);
    uint32_t* d_sq = (uint32_t*)(ar.base + o_sq);
    FicDecodeState st;
    memset(&st, 0, sizeof(st));
    st.avg = avg_error_io ? *avg_error_io : 0.0f;
    hipError_t e = hipMemcpy(d_q, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_image, init.data(), npix * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_state, &st, sizeof(st), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(FIC_E_HIP, "fic_decode_rgb_run: %s", hipGetErrorString(e));
    for (int counter = 0; rc == FIC_OK && counter < 50; counter++) {
        if (fic_launch_decode_iteration_rgb(d_scaled, d_image, d_q, d_state, d_sq, counter, g, nullptr)) {
            rc = fail(FIC_E_HIP, "decodeRGB iteration launch failed");
            break;
        }
        if ((counter & 7) == 7 || counter == 49) {
            e = hipMemcpy(&st, d_state, sizeof(st), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { rc = fail(FIC_E_HIP, "decodeRGB readback: %s", hipGetErrorString(e)); break; }
            if (st.done) break;
        }
    }
    if (rc == FIC_OK && st.bad_index)
        rc = fail(FIC_E_ARGUMENT, "decodeRGB: a codebook row points outside the domain pool (ArrayIndexOutOfBounds at FC:477)");
    if (rc == FIC_OK) {
        e = hipMemcpy(argb_out, d_image, npix * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(FIC_E_HIP, "fic_decode_rgb_run: %s", hipGetErrorString(e));
    }
    if (rc == FIC_OK) {
        if (avg_error_io) *avg_error_io = st.avg_out;
        if (iterations) *iterations = st.iters;
    }
    arena_give(ar);
    return rc;
}

// ---- joint-RGB encode (encodeRGB FC:171-219) -----------------------------------------------------
// A context owns the device working set of `planes` colour images of one geometry (config-5 style batches; the one-shot
// entry keeps a few single-image contexts).  The kernels are per image: a batch is their launch sequence per plane on the
// caller's stream.  The covariance sums stay sequential f32 in the reference's order (FC:781-792: they exceed 2^24).
struct fic_rgb_ctx {
    int device = 0;
    FicGeom g;
    int32_t* argb_own = nullptr;     // context-owned input copy
    const int32_t* argb = nullptr;   // input in use (own copy or the caller's device pointer)
    int32_t* scaled = nullptr;       // per plane: [H/2][W/2]
    uint16_t* pool_sum = nullptr;    // [N_d][n]
    float* pool_cf = nullptr;        // [N_d][n] (full search at B = 4 / 8)
    FicRgbDomStat* pool_st = nullptr;
    int16_t* rng_t = nullptr;
    FicRgbRngStat* rng_st = nullptr;
    unsigned long long* key = nullptr;
    int32_t *idx_local = nullptr, *idx_global = nullptr, *qrows = nullptr, *collage = nullptr;
    float *a = nullptr, *bR = nullptr, *bG = nullptr, *bB = nullptr;
    int32_t* dec_image = nullptr;    // decoder: image, scaled image, state, per-pixel squared changes (one plane at a time)
    int32_t* dec_scaled = nullptr;
    FicDecodeState* dec_state = nullptr;
    uint32_t* dec_sq = nullptr;
    bool have_input = false, encoded_any = false, have_collage = false;
    hipStream_t last_stream = nullptr;
    std::mutex mu;
};

namespace {
void rgb_free_all(fic_rgb_ctx* c)
{
    (void)hipSetDevice(c->device);
    void* ptrs[] = {c->argb_own, c->scaled, c->pool_sum, c->pool_cf, c->pool_st, c->rng_t, c->rng_st, c->key, c->idx_local,
                    c->idx_global, c->qrows, c->collage, c->a, c->bR, c->bG, c->bB, c->dec_image, c->dec_scaled, c->dec_state, c->dec_sq};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
}
// buffers of plane p as the per-image kernels expect them
void rgb_plane(const fic_rgb_ctx* c, int p, FicRgbBuffers* b, FicRgbOutputs* o)
{
    const FicGeom& g = c->g;
    const size_t P = (size_t)p, nd = (size_t)g.Nd, nr = (size_t)g.Nr, n = (size_t)g.n;
    b->argb = const_cast<int32_t*>(c->argb) + P * g.W * g.H;
    b->scaled = c->scaled + P * g.Ws * g.Hs;
    b->pool_sum = c->pool_sum + P * nd * n;
    b->pool_cf = c->pool_cf ? c->pool_cf + P * nd * n : nullptr;
    b->pool_st = c->pool_st + P * nd;
    b->rng_t = c->rng_t + P * nr * n;
    b->rng_st = c->rng_st + P * nr;
    b->key = c->key + P * nr;
    o->idx_local = c->idx_local + P * nr;
    o->idx_global = c->idx_global + P * nr;
    o->a = c->a + P * nr;
    o->bR = c->bR + P * nr;
    o->bG = c->bG + P * nr;
    o->bB = c->bB + P * nr;
    o->qrows = c->qrows + P * nr * 5;
}
// idle single-image contexts of the one-shot RGB entry, most recently used last
std::mutex g_rgb_mu;
std::vector<fic_rgb_ctx*> g_rgb_cache;
}  // namespace

fic_rgb_ctx* fic_rgb_ctx_create(int device, int w, int h, int B, int wK, int planes)
{
    FicGeom g;
    if (make_geometry(w, h, B, wK, 1, planes, &g)) return nullptr;
    int ndev = fic_device_count();
    if (ndev <= 0) { fail(FIC_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)"); return nullptr; }
    if (device < 0 || device >= ndev) { fail(FIC_E_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail(FIC_E_HIP, "hipSetDevice(%d) failed", device); return nullptr; }
    fic_rgb_ctx* c = new fic_rgb_ctx();
    c->device = device;
    c->g = g;
    const size_t P = (size_t)planes, npix = (size_t)w * h, nr = (size_t)g.Nr, nd = (size_t)g.Nd, n = (size_t)g.n;
    int rc = FIC_OK;
    auto A = [&](int r) { if (rc == FIC_OK) rc = r; };
    A(dev_alloc(&c->scaled, P * g.Ws * g.Hs));
    A(dev_alloc(&c->pool_sum, P * nd * n));
    if (g.full && g.B <= 8) A(dev_alloc(&c->pool_cf, P * nd * n));      // fast full-search sweep (k_sweep_rgb_fast)
    A(dev_alloc(&c->pool_st, P * nd));
    A(dev_alloc(&c->rng_t, P * nr * n));
    A(dev_alloc(&c->rng_st, P * nr));
    A(dev_alloc(&c->key, P * nr));
    A(dev_alloc(&c->idx_local, P * nr));
    A(dev_alloc(&c->idx_global, P * nr));
    A(dev_alloc(&c->a, P * nr));
    A(dev_alloc(&c->bR, P * nr));
    A(dev_alloc(&c->bG, P * nr));
    A(dev_alloc(&c->bB, P * nr));
    A(dev_alloc(&c->qrows, P * nr * 5));
    A(dev_alloc(&c->collage, P * npix));
    if (rc != FIC_OK) {
        rgb_free_all(c);
        delete c;
        return nullptr;
    }
    return c;
}

void fic_rgb_ctx_destroy(fic_rgb_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    rgb_free_all(c);
    delete c;
}

int fic_rgb_ctx_set_argb_host(fic_rgb_ctx* c, const int32_t* argb)
{
    if (!c || !argb) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_set_argb_host: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const size_t npix = (size_t)c->g.planes * c->g.W * c->g.H;
    if (!c->argb_own) { int rc = dev_alloc(&c->argb_own, npix); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->last_stream));     // a previous encode may still read the copy
    HIP_TRY(hipMemcpy(c->argb_own, argb, npix * sizeof(int32_t), hipMemcpyHostToDevice));
    c->argb = c->argb_own;
    c->have_input = true;
    return FIC_OK;
}

int fic_rgb_ctx_set_argb_device(fic_rgb_ctx* c, const void* dev_argb)
{
    if (!c || !dev_argb) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_set_argb_device: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    c->argb = (const int32_t*)dev_argb;
    c->have_input = true;
    return FIC_OK;
}

int fic_rgb_ctx_encode(fic_rgb_ctx* c, int with_collage, void* hip_stream)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_encode: null context");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_input) return fail(FIC_E_STATE, "fic_rgb_ctx_encode: no input image set");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)hip_stream;
    c->last_stream = s;
    const FicGeom& g = c->g;
    FicGeom g1 = g;
    g1.planes = 1;
    for (int p = 0; p < g.planes; p++) {
        FicRgbBuffers b;
        FicRgbOutputs o;
        rgb_plane(c, p, &b, &o);
        if (fic_launch_rgb_encode(b, o, with_collage ? c->collage + (size_t)p * g.W * g.H : nullptr, g1, s))
            return fail(FIC_E_HIP, "RGB kernel launch failed (plane %d)", p);
    }
    c->encoded_any = true;
    c->have_collage = with_collage != 0;
    return FIC_OK;
}

int fic_rgb_ctx_sync(fic_rgb_ctx* c)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_sync: null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    return FIC_OK;
}

int fic_rgb_ctx_get_results_host(fic_rgb_ctx* c, int32_t* idx_local, float* a, float* bR, float* bG, float* bB, int32_t* qrows5,
                                 int32_t* collage_argb)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_get_results_host: null context");
    if (!c->encoded_any) return fail(FIC_E_STATE, "fic_rgb_ctx_get_results_host: nothing encoded yet");
    if (collage_argb && !c->have_collage) return fail(FIC_E_STATE, "fic_rgb_ctx_get_results_host: the last encode built no collage");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    const size_t n = (size_t)c->g.planes * c->g.Nr;
    if (idx_local) HIP_TRY(hipMemcpy(idx_local, c->idx_local, n * 4, hipMemcpyDeviceToHost));
    if (a) HIP_TRY(hipMemcpy(a, c->a, n * 4, hipMemcpyDeviceToHost));
    if (bR) HIP_TRY(hipMemcpy(bR, c->bR, n * 4, hipMemcpyDeviceToHost));
    if (bG) HIP_TRY(hipMemcpy(bG, c->bG, n * 4, hipMemcpyDeviceToHost));
    if (bB) HIP_TRY(hipMemcpy(bB, c->bB, n * 4, hipMemcpyDeviceToHost));
    if (qrows5) HIP_TRY(hipMemcpy(qrows5, c->qrows, n * 20, hipMemcpyDeviceToHost));
    if (collage_argb) HIP_TRY(hipMemcpy(collage_argb, c->collage, (size_t)c->g.planes * c->g.W * c->g.H * 4, hipMemcpyDeviceToHost));
    return FIC_OK;
}

// decodeRGB (FC:430-508) from the context's quantised rows, plane by plane, everything device resident
int fic_rgb_ctx_decode_host(fic_rgb_ctx* c, int32_t* argb_out, float* avg_error_out, int* iterations_out)
{
    if (!c || !argb_out) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_decode_host: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->encoded_any) return fail(FIC_E_STATE, "fic_rgb_ctx_decode_host: nothing encoded yet");
    HIP_TRY(hipSetDevice(c->device));
    const FicGeom& g = c->g;
    FicGeom g1 = g;
    g1.planes = 1;
    const size_t npix = (size_t)g.W * g.H;
    if (!c->dec_image) { int rc = dev_alloc(&c->dec_image, npix); if (rc) return rc; }
    if (!c->dec_scaled) { int rc = dev_alloc(&c->dec_scaled, (size_t)g.Ws * g.Hs); if (rc) return rc; }
    if (!c->dec_state) { int rc = dev_alloc(&c->dec_state, 1); if (rc) return rc; }
    if (!c->dec_sq) { int rc = dev_alloc(&c->dec_sq, npix); if (rc) return rc; }
    hipStream_t s = c->last_stream;
    std::vector<int32_t> init(npix, (int32_t)0xff808080u);                          // generateGrayImage FC:1142-1148
    for (int p = 0; p < g.planes; p++) {
        FicDecodeState st;
        memset(&st, 0, sizeof(st));
        HIP_TRY(hipMemcpyAsync(c->dec_image, init.data(), npix * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->dec_state, &st, sizeof(st), hipMemcpyHostToDevice, s));
        for (int counter = 0; counter < 50; counter++) {
            if (fic_launch_decode_iteration_rgb(c->dec_scaled, c->dec_image, c->qrows + (size_t)p * g.Nr * 5, c->dec_state, c->dec_sq,
                                                counter, g1, s))
                return fail(FIC_E_HIP, "decodeRGB iteration launch failed");
            if ((counter & 7) == 7 || counter == 49) {
                HIP_TRY(hipMemcpyAsync(&st, c->dec_state, sizeof(st), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                if (st.done) break;
            }
        }
        if (st.bad_index) return fail(FIC_E_ARGUMENT, "decodeRGB: a codebook row of plane %d points outside the domain pool", p);
        HIP_TRY(hipMemcpy(argb_out + (size_t)p * npix, c->dec_image, npix * 4, hipMemcpyDeviceToHost));
        if (avg_error_out) avg_error_out[p] = st.avg_out;
        if (iterations_out) iterations_out[p] = st.iters;
    }
    return FIC_OK;
}

namespace {
fic_rgb_ctx* rgb_cache_take(int device, int w, int h, int B, int wK)
{
    std::lock_guard<std::mutex> lk(g_rgb_mu);
    for (size_t i = g_rgb_cache.size(); i-- > 0;) {
        fic_rgb_ctx* c = g_rgb_cache[i];
        const FicGeom& g = c->g;
        if (c->device == device && g.W == w && g.H == h && g.B == B && g.wK == wK && g.planes == 1) {
            g_rgb_cache.erase(g_rgb_cache.begin() + (long)i);
            return c;
        }
    }
    return nullptr;
}
void rgb_cache_give(fic_rgb_ctx* c)
{
    fic_rgb_ctx* evict = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_rgb_mu);
        g_rgb_cache.push_back(c);
        if (g_rgb_cache.size() > 4) {
            evict = g_rgb_cache.front();
            g_rgb_cache.erase(g_rgb_cache.begin());
        }
    }
    if (evict) fic_rgb_ctx_destroy(evict);
}
}  // namespace

int fic_encode_rgb_argb(const int32_t* argb, int w, int h, int B, int wK, int device, int32_t* idx_local, float* a,
                        float* bR, float* bG, float* bB, int32_t* qrows5, int32_t* collage_argb)
{
    if (!argb || !idx_local || !a || !bR || !bG || !bB) return fail(FIC_E_ARGUMENT, "fic_encode_rgb_argb: null argument");
    // the GUI re-encodes the same image on every slider move (CTL:125-145): keep the last few working sets
    fic_rgb_ctx* c = rgb_cache_take(device, w, h, B, wK);
    if (!c) c = fic_rgb_ctx_create(device, w, h, B, wK, 1);
    if (!c) return g_err_code ? g_err_code : FIC_E_HIP;   // fic_rgb_ctx_create recorded why
    int rc = fic_rgb_ctx_set_argb_host(c, argb);
    if (rc == FIC_OK) rc = fic_rgb_ctx_encode(c, collage_argb ? 1 : 0, nullptr);
    if (rc == FIC_OK) rc = fic_rgb_ctx_get_results_host(c, idx_local, a, bR, bG, bB, qrows5, collage_argb);
    const std::string keep = g_err;
    const int keep_code = g_err_code;
    if (rc == FIC_OK) rgb_cache_give(c);
    else fic_rgb_ctx_destroy(c);
    g_err = keep;
    g_err_code = keep_code;
    return rc;
}

int64_t fic_write_run_rgb(const int32_t* qrows5, int n_ranges, int w, int h, int B, int wK, uint8_t* out, int64_t capacity)
{
    if (!qrows5 || !out || n_ranges < 0) return fail(FIC_E_ARGUMENT, "fic_write_run_rgb: bad argument");
    int64_t need = 20 + 20 * (int64_t)n_ranges;
    if (capacity < need) return fail(FIC_E_CAPACITY, "fic_write_run_rgb: need %lld bytes, have %lld", (long long)need, (long long)capacity);
    auto put = [](uint8_t* p, int32_t v) {
        uint32_t u = (uint32_t)v;
        p[0] = (uint8_t)(u >> 24); p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u;
    };
    const int32_t hdr[5] = {1, w, h, B, wK};          // FC:234-238, isRGB = 1
    for (int i = 0; i < 5; i++) put(out + 4 * i, hdr[i]);
    uint8_t* p = out + 20;
    for (int64_t i = 0; i < 5 * (int64_t)n_ranges; i++, p += 4) put(p, qrows5[i]);   // FC:249-256
    return need;
}

static void fic_release_comms_();
static int encode_oneshot(const uint8_t* gray, const int32_t* argb, int w, int h, int B, int wK, int n_iso, int device,
                          int32_t* idx_local, float* a, float* b, int32_t* iso, int32_t* qrows)
{
    if ((!gray && !argb) || !idx_local || !a || !b) return fail(FIC_E_ARGUMENT, "fic_encode_gray: null argument");
    // The GUI calls encode once per slider move with the same geometry (CTL:125-145): keep the last few
    // working sets instead of paying ~18 hipMalloc/hipFree per call.
    fic_ctx* c = cache_take(device, w, h, B, wK, n_iso);
    if (!c) c = fic_ctx_create(device, w, h, B, wK, n_iso, 1);
    if (!c) return g_err_code ? g_err_code : FIC_E_HIP;   // fic_ctx_create recorded why
    int rc = gray ? fic_ctx_set_gray_host(c, gray) : fic_ctx_set_argb_host(c, argb);
    if (rc == FIC_OK) rc = fic_ctx_encode(c, 0, -1, nullptr);
    if (rc == FIC_OK) rc = fic_ctx_get_results_host(c, idx_local, a, b, iso, qrows, nullptr, nullptr);
    std::string keep = g_err;
    int keep_code = g_err_code;
    if (rc == FIC_OK) cache_give(c);
    else fic_ctx_destroy(c);
    g_err = keep;
    g_err_code = keep_code;
    return rc;
}

void fic_release_cache(void)
{
    std::vector<fic_ctx*> drop;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        drop.swap(g_cache);
    }
    for (fic_ctx* c : drop) fic_ctx_destroy(c);
    std::vector<fic_rgb_ctx*> drop_rgb;
    {
        std::lock_guard<std::mutex> lk(g_rgb_mu);
        drop_rgb.swap(g_rgb_cache);
    }
    for (fic_rgb_ctx* c : drop_rgb) fic_rgb_ctx_destroy(c);
    fic_release_comms_();
    arena_release_all();
}

int fic_encode_gray_argb(const int32_t* argb, int w, int h, int B, int wK, int n_iso, int device, int32_t* idx_local,
                         float* a, float* b, int32_t* iso, int32_t* qrows)
{
    return encode_oneshot(nullptr, argb, w, h, B, wK, n_iso, device, idx_local, a, b, iso, qrows);
}

int fic_encode_gray_u8(const uint8_t* gray, int w, int h, int B, int wK, int n_iso, int device, int32_t* idx_local,
                       float* a, float* b, int32_t* iso, int32_t* qrows)
{
    return encode_oneshot(gray, nullptr, w, h, B, wK, n_iso, device, idx_local, a, b, iso, qrows);
}

// ---- in-library multi-device encode (SURVEY.md 8b "n_gpus", 8e) -------------------------------------------------
// FractalCompression.encode (FC:54-59) is ONE synchronous call on one host thread; the range loop it replaces
// (FC:125-159) carries no state between iterations, so the call shards its range blocks over n_gpus devices from that
// one thread: one context and one non-blocking stream per device, tile-aligned spans (the rule of sharding.shard_spans),
// every device builds its own replica of the pool from the replicated image, and the 24-byte codebook records are
// gathered to device 0 with ONE grouped RCCL send/recv (latency-bound on xGMI; no ring, no all-reduce).
// RCCL is loaded on first use (librccl.so.1): a single-GPU host needs no RCCL.
namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::vector<int> devs;               // devices of the live communicators
    std::vector<ncclComm_t> comms;
};
std::mutex g_multi_mu;
Rccl g_rccl;

int rccl_load()
{
    if (g_rccl.lib) return FIC_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(FIC_E_HIP, "multi-device encode needs RCCL: %s", dlerror());
#define FIC_RCCL_SYM(field, name)                                                                   \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                                          \
    if (!g_rccl.field) { dlclose(h); return fail(FIC_E_HIP, "librccl lacks %s", name); }
    FIC_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    FIC_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    FIC_RCCL_SYM(GroupStart, "ncclGroupStart")
    FIC_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    FIC_RCCL_SYM(Send, "ncclSend")
    FIC_RCCL_SYM(Recv, "ncclRecv")
    FIC_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef FIC_RCCL_SYM
    g_rccl.lib = h;
    return FIC_OK;
}
void rccl_drop_comms()
{
    for (ncclComm_t c : g_rccl.comms)
        if (c) (void)g_rccl.CommDestroy(c);
    g_rccl.comms.clear();
    g_rccl.devs.clear();
}
#define RCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) return fail(FIC_E_HIP, "%s: %s", #expr, g_rccl.GetErrorString(r_));  \
    } while (0)
// one communicator per device, created once per device list and kept (ncclCommInitAll costs ~a second)
int rccl_comms(const std::vector<int>& devs)
{
    int rc = rccl_load();
    if (rc) return rc;
    if (g_rccl.devs == devs && !g_rccl.comms.empty()) return FIC_OK;
    rccl_drop_comms();
    g_rccl.comms.assign(devs.size(), nullptr);
    ncclResult_t r = g_rccl.CommInitAll(g_rccl.comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) {
        g_rccl.comms.clear();
        return fail(FIC_E_HIP, "ncclCommInitAll(%d devices): %s", (int)devs.size(), g_rccl.GetErrorString(r));
    }
    g_rccl.devs = devs;
    return FIC_OK;
}

// gather of the span records [begin_i, begin_i + count_i) of every context's `records` array into context 0's
int gather_rccl(const std::vector<fic_ctx*>& ctx, const std::vector<int>& begin, const std::vector<int>& count)
{
    std::vector<int> devs;
    for (fic_ctx* c : ctx) devs.push_back(c->device);
    int rc = rccl_comms(devs);
    if (rc) return rc;
    RCCL_TRY(g_rccl.GroupStart());
    for (size_t i = 1; i < ctx.size(); i++) {
        if (count[i] == 0) continue;
        const size_t off = (size_t)begin[i] * 6, cnt = (size_t)count[i] * 6;
        HIP_TRY(hipSetDevice(ctx[i]->device));
        RCCL_TRY(g_rccl.Send(ctx[i]->o.records + off, cnt, ncclInt32, 0, g_rccl.comms[i], ctx[i]->own_stream));
        HIP_TRY(hipSetDevice(ctx[0]->device));
        RCCL_TRY(g_rccl.Recv(ctx[0]->o.records + off, cnt, ncclInt32, (int)i, g_rccl.comms[0], ctx[0]->own_stream));
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return FIC_OK;
}
// the same gather as plain device copies: logical shards that share a physical device (FIC_FAKE_DEVICES, a test knob --
// RCCL refuses two ranks on one device) and FIC_GATHER=copy
int gather_copy(const std::vector<fic_ctx*>& ctx, const std::vector<int>& begin, const std::vector<int>& count)
{
    for (size_t i = 1; i < ctx.size(); i++) {
        if (count[i] == 0) continue;
        const size_t off = (size_t)begin[i] * 6, bytes = (size_t)count[i] * 6 * sizeof(int32_t);
        HIP_TRY(hipSetDevice(ctx[i]->device));
        if (ctx[i]->device == ctx[0]->device)
            HIP_TRY(hipMemcpyAsync(ctx[0]->o.records + off, ctx[i]->o.records + off, bytes, hipMemcpyDeviceToDevice, ctx[i]->own_stream));
        else
            HIP_TRY(hipMemcpyPeerAsync(ctx[0]->o.records + off, ctx[0]->device, ctx[i]->o.records + off, ctx[i]->device, bytes,
                                       ctx[i]->own_stream));
    }
    for (size_t i = 1; i < ctx.size(); i++) {
        HIP_TRY(hipSetDevice(ctx[i]->device));
        HIP_TRY(hipStreamSynchronize(ctx[i]->own_stream));
    }
    return FIC_OK;
}

int encode_multi(const uint8_t* gray, const int32_t* argb, int w, int h, int B, int wK, int n_iso, int n_gpus,
                 int32_t* idx_local, float* a, float* b, int32_t* iso, int32_t* qrows)
{
    if ((!gray && !argb) || !idx_local || !a || !b) return fail(FIC_E_ARGUMENT, "fic_encode_gray_multi: null argument");
    if (n_gpus < 1) return fail(FIC_E_ARGUMENT, "n_gpus=%d", n_gpus);
    FicGeom g;
    int rc = make_geometry(w, h, B, wK, n_iso, 1, &g);
    if (rc) return rc;
    const int ndev = fic_device_count();
    if (ndev <= 0) return fail(FIC_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    const char* fake = getenv("FIC_FAKE_DEVICES");        // test knob: logical devices beyond the real ones share them round-robin
    if (n_gpus > ndev && !(fake && atoi(fake) >= n_gpus))
        return fail(FIC_E_NO_DEVICE, "n_gpus=%d but %d HIP device(s) visible", n_gpus, ndev);
    const bool distinct = n_gpus <= ndev;
    const char* gmode = getenv("FIC_GATHER");
    const bool use_rccl = distinct && !(gmode && !strcmp(gmode, "copy"));
    std::lock_guard<std::mutex> lk(g_multi_mu);
    // tile-aligned spans: no sweep tile is computed twice (same rule as sharding.shard_spans)
    const int tsz = 64 * g.NR;
    std::vector<int> begin(n_gpus), count(n_gpus);
    for (int r = 0; r < n_gpus; r++) {
        const long long t0 = (long long)g.tiles * r / n_gpus, t1 = (long long)g.tiles * (r + 1) / n_gpus;
        const int bb = (int)(t0 * tsz < g.Nr ? t0 * tsz : g.Nr), ee = (int)(t1 * tsz < g.Nr ? t1 * tsz : g.Nr);
        begin[r] = bb;
        count[r] = ee - bb;
    }
    std::vector<fic_ctx*> ctx(n_gpus, nullptr);
    auto drop = [&](bool keep) {
        const std::string keep_err = g_err;
        const int keep_code = g_err_code;
        for (fic_ctx* c : ctx)
            if (c) { if (keep) cache_give(c); else fic_ctx_destroy(c); }
        g_err = keep_err;
        g_err_code = keep_code;
    };
    for (int i = 0; i < n_gpus && rc == FIC_OK; i++) {
        const int dev = i % ndev;
        ctx[i] = cache_take(dev, w, h, B, wK, n_iso);
        if (!ctx[i]) ctx[i] = fic_ctx_create(dev, w, h, B, wK, n_iso, 1);
        if (!ctx[i]) { rc = g_err_code ? g_err_code : FIC_E_HIP; break; }
        if (!ctx[i]->own_stream) {
            hipError_t e = hipSetDevice(dev);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx[i]->own_stream, hipStreamNonBlocking);
            if (e != hipSuccess) { rc = fail(FIC_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); break; }
        }
        // the whole image on every device: each builds its own pool replica (cheaper than shipping the 16x-expanded pool)
        rc = gray ? fic_ctx_set_gray_host(ctx[i], gray) : fic_ctx_set_argb_host(ctx[i], argb);
    }
    for (int i = 0; i < n_gpus && rc == FIC_OK; i++)
        rc = fic_ctx_encode(ctx[i], begin[i], count[i], ctx[i]->own_stream);      // a rank without tiles: count 0, nothing launched
    if (rc == FIC_OK) rc = use_rccl ? gather_rccl(ctx, begin, count) : gather_copy(ctx, begin, count);
    std::vector<int32_t> rec;
    if (rc == FIC_OK) {
        rec.resize((size_t)g.Nr * 6);
        hipError_t e = hipSetDevice(ctx[0]->device);
        if (e == hipSuccess) e = hipMemcpyAsync(rec.data(), ctx[0]->o.records, rec.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx[0]->own_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx[0]->own_stream);
        if (e != hipSuccess) rc = fail(FIC_E_HIP, "codebook copy: %s", hipGetErrorString(e));
    }
    if (rc != FIC_OK) {
        // leave no stream with work in flight behind a failed call
        for (fic_ctx* c : ctx)
            if (c && c->own_stream) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->own_stream); }
        drop(false);
        return rc;
    }
    for (int j = 0; j < g.Nr; j++) {
        const int32_t* r6 = &rec[(size_t)j * 6];
        idx_local[j] = r6[0];
        memcpy(&a[j], &r6[1], 4);
        memcpy(&b[j], &r6[2], 4);
        if (iso) iso[j] = r6[3];
        if (qrows) { qrows[3 * j] = r6[0]; qrows[3 * j + 1] = r6[4]; qrows[3 * j + 2] = r6[5]; }
    }
    drop(true);
    return FIC_OK;
}

}  // namespace

// Test hook: loads RCCL, creates (and keeps) one communicator per device 0..n-1 and, for n >= 2, runs the gather's
// grouped send/recv pattern on 6-int records.  Returns FIC_OK or a negative code.
int fic_debug_rccl_selftest(int n)
{
    const int ndev = fic_device_count();
    if (n < 1 || n > ndev) return fail(FIC_E_NO_DEVICE, "fic_debug_rccl_selftest: %d of %d devices", n, ndev);
    std::lock_guard<std::mutex> lk(g_multi_mu);
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) devs[i] = i;
    int rc = rccl_comms(devs);
    if (rc) return rc;
    std::vector<int32_t*> buf(n, nullptr);
    std::vector<hipStream_t> st(n, nullptr);
    const size_t cnt = 6 * 1000;
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && e == hipSuccess; i++) {
        e = hipSetDevice(i);
        if (e == hipSuccess) e = hipMalloc((void**)&buf[i], cnt * n * 4);
        if (e == hipSuccess) e = hipMemset(buf[i], i + 1, cnt * n * 4);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    }
    if (e != hipSuccess) rc = fail(FIC_E_HIP, "rccl selftest setup: %s", hipGetErrorString(e));
    if (rc == FIC_OK && n >= 2) {
        ncclResult_t r = g_rccl.GroupStart();
        for (int i = 1; i < n && r == ncclSuccess; i++) {
            (void)hipSetDevice(i);
            r = g_rccl.Send(buf[i] + cnt * i, cnt, ncclInt32, 0, g_rccl.comms[i], st[i]);
            (void)hipSetDevice(0);
            if (r == ncclSuccess) r = g_rccl.Recv(buf[0] + cnt * i, cnt, ncclInt32, i, g_rccl.comms[0], st[0]);
        }
        if (r == ncclSuccess) r = g_rccl.GroupEnd();
        if (r != ncclSuccess) rc = fail(FIC_E_HIP, "rccl selftest: %s", g_rccl.GetErrorString(r));
        std::vector<int32_t> host(cnt * n);
        if (rc == FIC_OK) {
            (void)hipSetDevice(0);
            e = hipStreamSynchronize(st[0]);
            if (e == hipSuccess) e = hipMemcpy(host.data(), buf[0], host.size() * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = fail(FIC_E_HIP, "rccl selftest readback: %s", hipGetErrorString(e));
            for (int i = 0; i < n && rc == FIC_OK; i++) {
                const int32_t want = 0x01010101 * (i + 1);
                for (size_t k = 0; k < cnt; k++)
                    if (host[cnt * i + k] != want) { rc = fail(FIC_E_HIP, "rccl selftest: wrong data from device %d", i); break; }
            }
        }
    }
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(i);
        if (st[i]) { (void)hipStreamSynchronize(st[i]); (void)hipStreamDestroy(st[i]); }
        if (buf[i]) (void)hipFree(buf[i]);
    }
    return rc;
}

static void fic_release_comms_()
{
    std::lock_guard<std::mutex> lk(g_multi_mu);
    if (g_rccl.lib) rccl_drop_comms();
}

int fic_encode_gray_argb_multi(const int32_t* argb, int w, int h, int B, int wK, int n_iso, int n_gpus, int32_t* idx_local,
                               float* a, float* b, int32_t* iso, int32_t* qrows)
{
    if (n_gpus == 1) return encode_oneshot(nullptr, argb, w, h, B, wK, n_iso, 0, idx_local, a, b, iso, qrows);
    return encode_multi(nullptr, argb, w, h, B, wK, n_iso, n_gpus, idx_local, a, b, iso, qrows);
}

int fic_encode_gray_u8_multi(const uint8_t* gray, int w, int h, int B, int wK, int n_iso, int n_gpus, int32_t* idx_local,
                             float* a, float* b, int32_t* iso, int32_t* qrows)
{
    if (n_gpus == 1) return encode_oneshot(gray, nullptr, w, h, B, wK, n_iso, 0, idx_local, a, b, iso, qrows);
    return encode_multi(gray, nullptr, w, h, B, wK, n_iso, n_gpus, idx_local, a, b, iso, qrows);
}

}  // extern "C"
