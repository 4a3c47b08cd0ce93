// fic_capi.cpp -- C ABI (include/fic.h) over the gfx950 kernels.  Host-side orchestration only:
// validation, device buffers, launch order, result copies.  No compute happens on the CPU and
// there is no CPU fallback: without a HIP device every compute entry returns FIC_E_NO_DEVICE.
// (decoder entries: fic_capi_decode.cpp; joint-RGB contexts: fic_capi_rgb.cpp; multi-device entry: fic_capi_multi.cpp)
#include "fic_internal.h"

namespace ficd {

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

static int ilog2(int v)
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

}  // namespace ficd

using namespace ficd;


namespace {

int ctx_free_all(fic_ctx* c)
{
    (void)hipSetDevice(c->device);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    c->ev.clear();
    if (c->own_stream) { (void)hipStreamDestroy(c->own_stream); c->own_stream = nullptr; }
    void* ptrs[] = {c->gray_own, c->argb_stage, c->collage, c->decoded, c->dec_state, c->dec_sq, c->mfma_poolB, c->mfma_rngA, c->mfma_sw, c->mfma_rconst, c->q_pool, c->q_flat, c->q_rng, c->q_rngC, c->q_E, c->q_thg, c->q_fin, c->q_stats, c->d4_rng, c->d4_pool, c->b.scaled, c->b.pool_pix, c->b.pool_st, c->b.pool_var,
                    c->b.pool_s64, c->b.rng_pix, c->b.rng_st, c->b.key, c->o.idx_local, c->o.idx_global, c->o.iso,
                    c->o.a, c->o.b, c->o.err, c->o.qrows, c->o.records};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c->host_rec) { (void)hipHostFree(c->host_rec); c->host_rec = nullptr; }
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

}  // namespace

fic_ctx* ficd::cache_take(int device, int w, int h, int B, int wK, int n_iso)
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


void ficd::cache_give(fic_ctx* c)
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

namespace {

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

// The exact-covariance matrix-core sweeps of round 1 ("sweep" = 3 / 4, fic_bf16.hip / fic_mfma.hip) lose to k_sweep_q everywhere; they are
// kept as independent cross-checks for the test-suite (a third and fourth implementation of the same search) and are compiled
// only into builds with FIC_BUILD_XCHECK (build.py: on by default, FIC_BUILD_XCHECK=0 for a deployment build).
#ifdef FIC_BUILD_XCHECK
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

#else
int matrix_core_prep(fic_ctx*, int, hipStream_t) { return fail(FIC_E_ARGUMENT, "sweep 3 / 4 (cross-check kernels) are not in this build (FIC_BUILD_XCHECK=0)"); }
int matrix_core_sweep(fic_ctx*, int, int, int, hipStream_t, int*) { return fail(FIC_E_ARGUMENT, "sweep 3 / 4 (cross-check kernels) are not in this build (FIC_BUILD_XCHECK=0)"); }
#endif
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
    if (!c->q_rngC) HIP_TRY(hipMalloc(&c->q_rngC, P * g.Nr_pad * fic_q_cols_per_range(g.B, g.n_iso) * g.n));
    if (!c->q_E) HIP_TRY(hipMalloc(&c->q_E, P * g.Nr_pad * sizeof(float)));
    // (+ the padded column tiles behind the last plane: the sweep's fast path reads theta_g of every column it owns)
    if (!c->q_thg) HIP_TRY(hipMalloc(&c->q_thg, (P * g.Nr_pad + (size_t)2 * q.CT * 32 + 64) * sizeof(uint32_t)));
    const int tsz = 64 * g.NR;
    const int grp0 = tile0 * tsz / 64, grp1 = tile1 * tsz / 64;      // 64-range groups covering the span
    if (fic_launch_q_prep(c->b, c->q_pool, c->q_flat, c->q_rng, c->q_rngC, c->q_E, c->q_thg, g, q.ndtiles_alloc, q.nct_alloc, grp0, grp1 - grp0, s))
        return fail(FIC_E_HIP, "k_pool_q / k_range_q launch failed");
    return FIC_OK;
}
int q_sweep(fic_ctx* c, int tile0, int tile1, hipStream_t s, int* nchunks_out, bool fused_fin = false, int r_begin = 0, int r_count = 0)
{
    const FicGeom& g = c->g;
    const QShape q = q_shape(g);
    const int tsz = 64 * g.NR;
    const int cpr = fic_q_cols_per_range(g.B, g.n_iso);
    const int ct_begin = tile0 * tsz * cpr / 32, ct_end = tile1 * tsz * cpr / 32;
    int nchunks = c->opt_chunks;
    if (nchunks <= 0) {
        // Two reasons to split the pool into chunks, each with its own price (a chunk starts with theta unset: until its
        // first good candidate has been evaluated every pair of every tile is queued -- tens of tiles' worth of work):
        //  fill:    fewer workgroups than the chip holds at once (`resident`): split until two rounds of them exist, but
        //           keep >= 30 (8 isometries) / 14 (1 isometry) tiles per chunk;
        //  balance: the kernel ends with its slowest wave, and one round of equally long waves leaves ~25 % of the
        //           wave-cycles idle (profiles/r02w_chunk_count_sweep.txt): aim for 8 rounds, but only with chunks long
        //           enough (>= 1024 tiles) that the start-up is noise.
        const long long base_wg = (long long)((ct_end - ct_begin + q.CT - 1) / q.CT) * g.planes;
        const long long resident = 256LL * fic_q_resident(g.B);
        auto ceil_div = [](long long a, long long b) { return (a + b - 1) / b; };
        // (one image, profiles/r03n_single_image_chunk_count_sweep.json: a chunk restarts theta, and what that costs against
        //  the parallelism it buys depends on the columns per range block -- with 8 isometries chunks below ~30 domain tiles
        //  lose (512x512: 0.086 ms per encode at 16 chunks, 0.092-0.094 at 28), with 1 isometry 14-tile chunks still gain
        //  (0.060 at 35 chunks, 0.061 at 28, 0.076 at 16))
        const long long min_tiles = g.n_iso == 8 ? 30 : 14;
        long long fill = base_wg < resident ? ceil_div(2 * resident, base_wg) : 1;
        if (fill > q.ndtiles / min_tiles) fill = q.ndtiles / min_tiles;
        long long bal = ceil_div(8 * resident, base_wg);
        if (bal > q.ndtiles / 1024) bal = q.ndtiles / 1024;
        const long long nc = fill > bal ? fill : bal;
        nchunks = (int)(nc < 1 ? 1 : nc);
    }
    if (nchunks > q.ndtiles) nchunks = q.ndtiles;
    int tiles_per_chunk = (q.ndtiles + nchunks - 1) / nchunks;
    tiles_per_chunk = (tiles_per_chunk + q.unroll - 1) / q.unroll * q.unroll;     // whole iterations of the unrolled sweep loop
    nchunks = (q.ndtiles + tiles_per_chunk - 1) / tiles_per_chunk;
    if (fused_fin && !c->q_fin) {                      // one counter per (plane, column group); the sweep leaves them at zero
        const size_t nfin = (size_t)g.planes * (q.nct_alloc / q.CT + 1);
        HIP_TRY(hipMalloc((void**)&c->q_fin, nfin * sizeof(unsigned int)));
        HIP_TRY(hipMemsetAsync(c->q_fin, 0, nfin * sizeof(unsigned int), s));
    }
    if (fic_launch_sweep_q(c->b, c->q_pool, c->q_flat, c->q_rng, c->q_rngC, c->q_E, c->q_thg, g, ct_begin, ct_end, q.ndtiles, q.ndtiles_alloc,
                           q.nct_alloc, tiles_per_chunk, nchunks, s, c->q_stats, c->opt_noflag, fused_fin ? &c->o : nullptr, c->q_fin,
                           r_begin, r_count))
        return fail(FIC_E_HIP, "k_sweep_q launch failed");
    *nchunks_out = nchunks;
    c->last_tiles_per_chunk = tiles_per_chunk;
    return FIC_OK;
}

}  // namespace

extern "C" {

#ifdef FIC_BUILD_XCHECK
const char* fic_version(void) { return "fic-hip 0.3 (gfx950, +xcheck sweeps)"; }
#else
const char* fic_version(void) { return "fic-hip 0.3 (gfx950)"; }
#endif
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
        // hipMemset on device memory only ENQUEUES on the null stream; encodes run on the caller's stream, which may be
        // non-blocking (torch's are): without this wait a first encode issued right away can be overtaken by the fill
        // (seen with two processes sharing one GPU: bench.py's post-run check context, round 3)
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
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
        // (k_sweep_q's queue entries carry the domain block in 24 bits: pools of 2^24 blocks or more -- images beyond
        //  16000 x 16000 at B = 8 -- stay on the VALU sweep)
        kind = !g.full ? 1 : ((pairs >= 2000000LL && g.Nd < (1 << 24)) ? 6 : valu);
        const char* env = getenv("FIC_SWEEP");
        if (env && env[0] >= '2' && env[0] <= '6' && env[1] == '\0' && g.full) {
            const int want = env[0] - '0';
            kind = want == 5 ? valu : want;
        }
    }
    if (kind >= 2 && !g.full) return fail(FIC_E_ARGUMENT, "fast sweep needs full search (wK == Dw == Dh)");
#ifndef FIC_BUILD_XCHECK
    if (kind == 3 || kind == 4) return fail(FIC_E_ARGUMENT, "sweep %d (cross-check kernel) is not in this build (FIC_BUILD_XCHECK=0)", kind);
#endif
    if (kind == 5 && !d4_available(g)) return fail(FIC_E_ARGUMENT, "sweep 5 (k_sweep_d4) needs full search, n_iso = 8 and B = 8");
    if (kind == 6 && g.Nd >= (1 << 24)) return fail(FIC_E_ARGUMENT, "sweep 6 (k_sweep_q) needs a pool of fewer than 2^24 blocks");
    const int tsz = 64 * g.NR;
    const int tile0 = range_begin / tsz;
    const int tile1 = (range_begin + range_count + tsz - 1) / tsz;
    const int ntiles = tile1 - tile0;
    int nchunks = 1;
    int rc = FIC_OK;
    // pool build (createCodebuch FC:119) + range prep
    // (a small launch of the default sweep builds the scaled image inside its one fused prep kernel: fic_q.hip, k_prep_q8)
    bool fused_prep = false;
    if (kind == 6) {
        const QShape q = q_shape(g);
        fused_prep = fic_q_prep_fused(g, q.ndtiles_alloc, tile1 * (64 * g.NR) / 64 - tile0 * (64 * g.NR) / 64) != 0;
    }
    if (!fused_prep && fic_launch_scale(c->b.gray, c->b.scaled, g, s)) return fail(FIC_E_HIP, "k_scale launch failed");
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
            rc = q_sweep(c, tile0, tile1, s, &nchunks, fused_prep, range_begin, range_count);   // small launches: finalise in the sweep's tail
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
    c->last_fused = fused_prep ? 1 : 0;
    // kinds 5 and 6 build no lane-transposed isometry copies: the finaliser recomputes the winner's covariance from the image
    if (!fused_prep &&
        fic_launch_finalize(c->b, c->o, g, range_begin, range_count, s, (kind == 5 || kind == 6) ? 1 : 0)) return fail(FIC_E_HIP, "k_finalize launch failed");
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
    // idx_local / a / b / iso / qrows are the six words of the packed 24-byte record (finalize_store): ONE copy into a pinned
    // buffer of the context and an unpack on the host instead of up to five blocking copies of ~15 us each -- the one-shot entry
    // the JNI host calls (fic_encode_gray_*) is three of them shorter per call.
    // (up to 65 536 range blocks: beyond that the copies' latency no longer matters and the host loop would)
    if (n > (size_t)65536) {
        if (idx_local) HIP_TRY(hipMemcpy(idx_local, c->o.idx_local, n * 4, hipMemcpyDeviceToHost));
        if (a) HIP_TRY(hipMemcpy(a, c->o.a, n * 4, hipMemcpyDeviceToHost));
        if (b) HIP_TRY(hipMemcpy(b, c->o.b, n * 4, hipMemcpyDeviceToHost));
        if (iso) HIP_TRY(hipMemcpy(iso, c->o.iso, n * 4, hipMemcpyDeviceToHost));
        if (qrows) HIP_TRY(hipMemcpy(qrows, c->o.qrows, n * 12, hipMemcpyDeviceToHost));
    } else if (idx_local || a || b || iso || qrows) {
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->host_rec) HIP_TRY(hipHostMalloc((void**)&c->host_rec, n * 6 * sizeof(int32_t), hipHostMallocDefault));
        HIP_TRY(hipMemcpy(c->host_rec, c->o.records, n * 6 * sizeof(int32_t), hipMemcpyDeviceToHost));
        const int32_t* r = c->host_rec;
        for (size_t i = 0; i < n; i++, r += 6) {
            if (idx_local) idx_local[i] = r[0];
            if (a) memcpy(&a[i], &r[1], 4);
            if (b) memcpy(&b[i], &r[2], 4);
            if (iso) iso[i] = r[3];
            if (qrows) { qrows[3 * i] = r[0]; qrows[3 * i + 1] = r[4]; qrows[3 * i + 2] = r[5]; }
        }
    }
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
    } else if (!strcmp(name, "q_shape")) {             // MFMA shape of the 1-isometry k_sweep_q at B = 8 / 16 (tests, A/B runs)
        if (value < 0 || value > 2) return fail(FIC_E_ARGUMENT, "q_shape must be 0 (by pool size), 1 (16x16x32) or 2 (32x32x16)");
        c->g.q_shape = value;
    } else if (!strcmp(name, "q_noflag")) {            // diagnostic: k_sweep_q without any flagged tile (wrong codebooks): its floor
        c->opt_noflag = value ? 1 : 0;
    } else if (!strcmp(name, "time_sweep")) {
        c->opt_time = value ? 1 : 0;
    } else if (!strcmp(name, "sweep_stats")) {
        HIP_TRY(hipSetDevice(c->device));
        if (value && !c->q_stats) {
            HIP_TRY(hipMalloc((void**)&c->q_stats, 8 * sizeof(unsigned long long)));
            HIP_TRY(hipMemset(c->q_stats, 0, 8 * sizeof(unsigned long long)));
            HIP_TRY(hipStreamSynchronize(nullptr));            // the fill is only enqueued (null stream); sweeps run on other streams
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

int fic_ctx_sweep_stats(fic_ctx* c, uint64_t* out8, int reset)
{
    if (!c || !out8) return fail(FIC_E_ARGUMENT, "fic_ctx_sweep_stats: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->q_stats) return fail(FIC_E_STATE, "fic_ctx_sweep_stats: set the option \"sweep_stats\" first");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    HIP_TRY(hipMemcpy(out8, c->q_stats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) {
        HIP_TRY(hipMemset(c->q_stats, 0, 8 * sizeof(uint64_t)));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
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

int fic_ctx_last_kernel(fic_ctx* c, char* out, int capacity)
{
    if (!c || !out || capacity < 1) return fail(FIC_E_ARGUMENT, "fic_ctx_last_kernel: bad argument");
    const FicGeom& g = c->g;
    const int NK = g.n / 16, kind = c->last_kind;
    const char* multi = c->last_chunks > 1 ? "true" : "false";
    const bool shorts = kind == 6 && fic_q_multi_kind(c->last_chunks, c->last_tiles_per_chunk) == 2;    // short pool chunks: k_sweep_qs / k_sweep_q16s
    char buf[96];
    if (kind == 6 && fic_q_shape16(g) && shorts) snprintf(buf, sizeof(buf), "k_sweep_q16s<%d>", NK);
    else if (kind == 6 && fic_q_shape16(g)) snprintf(buf, sizeof(buf), "k_sweep_q16<%d, %s>", NK, multi);
    else if (kind == 6 && shorts) snprintf(buf, sizeof(buf), "k_sweep_qs<%d, %d>", NK, g.n_iso == 1 ? 0 : (g.B == 4 ? 1 : 2));
    else if (kind == 6) snprintf(buf, sizeof(buf), "k_sweep_q<%d, %d, %s>", NK, g.n_iso == 1 ? 0 : (g.B == 4 ? 1 : 2), multi);
    else if (kind == 5) snprintf(buf, sizeof(buf), "k_sweep_d4");
    else if (kind == 2) snprintf(buf, sizeof(buf), "k_sweep_fast");
    else if (kind == 1) snprintf(buf, sizeof(buf), "k_sweep_generic");
    else if (kind == 3 || kind == 4) {
        const bool bf16 = kind == 3 && (g.B <= 8 || g.n_iso == 8);
        snprintf(buf, sizeof(buf), "%s%s", bf16 ? "k_sweep_bf16" : "k_sweep_mfma", g.n_iso == 8 ? "" : (bf16 ? "_1" : "1"));
    } else snprintf(buf, sizeof(buf), "(none)");
    // a small launch of the default sweep: one fused prep kernel before it, k_finalize's work in its tail (2 kernels, not 5)
    snprintf(out, (size_t)capacity, "%s%s", buf, kind == 6 && c->last_fused ? " after k_prep_q8, finalising" : "");
    return FIC_OK;
}

int fic_sweep_ranges_per_pool_read(int kind, int B, int n_iso)
{
    if ((B != 4 && B != 8 && B != 16) || (n_iso != 1 && n_iso != 8)) return 0;
    if (kind == 6) return fic_q_ctw_host(B) * 32 / fic_q_cols_per_range(B, n_iso);
    if (kind == 2 || kind == 5) {
        int NR = 1, NC = 1;
        fic_fast_variant(B, n_iso, &NR, &NC);
        return 64 * (kind == 5 ? 1 : NR);
    }
    return 0;
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

}  // extern "C"

int ficd::encode_oneshot(const uint8_t* gray, const int32_t* argb, int w, int h, int B, int wK, int n_iso, int device,
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
    ErrKeep keep;
    if (rc == FIC_OK) cache_give(c);
    else fic_ctx_destroy(c);
    return rc;
}

extern "C" {

void fic_release_cache(void)
{
    std::vector<fic_ctx*> drop;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        drop.swap(g_cache);
    }
    for (fic_ctx* c : drop) fic_ctx_destroy(c);
    release_rgb_cache();
    release_comms();
    release_decoder_arenas();
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

}  // extern "C"
