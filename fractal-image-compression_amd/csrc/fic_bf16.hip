// fic_bf16.hip -- OPT-IN matrix-core sweeps with bf16 operands ("sweep" = 3 at B = 4, B = 8, and B = 16 with 8 isometries): k_sweep_bf16<NK>
// (8 isometries) and k_sweep_bf16_1<NK> (reference algorithm, 1 isometry).  gfx950 (MI355X / CDNA4) only, wave64.
// Compile with -ffp-contract=off (every float expression rounds once per operation, like the Java reference:
// FractalCompression.java = FC, Domainblock.java = DB).
//
// Why a second operand type.  The i8 kernels of fic_mfma.hip spend ~4x more issue slots on the VALU epilogue than
// on their two MFMAs per 32x32 tile (B = 8): u8 pixels only fit i8 after a shift by 128, so every accumulator
// element needs the rank-2 correction  Aq*sum(d) - rem*dM + Kq  before it can be compared.  bf16 has an 8-bit
// significand, so the CENTRED pixels  r - rM  and  d - dM  (integers in [-255, 255]; rM = getMittelwert FC:67-73,
// dM = Domainblock.mittelWert DB:92-98) are exact bf16 values, every product (<= 255^2) and every partial sum
// (|.| <= n * 255^2 <= 64 * 65025 < 2^24) is an exactly representable f32 integer, and
//     v_mfma_f32_32x32x16_bf16  accumulates  kovarianz = sum_i (r_i - rM) * (d_i - dM)     (FC:665-672)
// itself -- exactly, in any summation order -- straight into the accumulator.  The epilogue shrinks from 56 to 28
// VALU instructions per 32x32 tile (16 multiplies by fl(1/s32), a max3 tree, one compare per range block).  At B = 16
// (n = 256: 16 bf16 steps against 8 i8 steps; 256 * 255^2 = 16 646 400 is still < 2^24) the 8-isometry kernel is
// faster than its i8 counterpart (51 vs 79 ms at 4096x4096), the 1-isometry kernel is not (9.2 vs 8.6 ms) and the i8
// kernel k_sweep_mfma1 stays in use there.
//
// What is the same as in k_sweep_fast / k_sweep_mfma, so that the result is the same bits: a
// conservative f32 prune test with a 2^-18 margin, the exact f64 Java epilogue for every pair that fails it, the
// (error, candidate) lexicographic 64-bit atomicMin, tau only ever raised by pairs of lower index that ARE evaluated,
// the exactly evaluated first tile of every pool chunk.  What differs is when the exact epilogue runs (deferred, from
// an LDS queue) and the level tau is raised with (the tested product times (1 - 2^-17), below).
// tests/test_gpu_mfma.py and tests/test_gpu_fullsize.py compare all sweeps against the oracle and each other.
//
// Fragment format (both operands; A rows and B columns use the same lane map, cdna_hip_programming.md section 3):
//   frag[tile][m][lane] = 8 bf16 = pixels [16m + 8h, 16m + 8h + 8) of row/column (lane & 31), h = lane >> 5,
//   m = 0 .. NK-1, NK = n / 16 MFMA steps (B = 4: 1, B = 8: 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>
#include "fic_device.h"
#include "fic_launch.h"
#include "fic_devfn.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// Level of a pair as the sweeps raise tau with it: the tested product |cov| * fl(1/s32) times (1 - 2^-17).  It is below
// the pair's true |cov| / sqrt(var) by at least 2^-18 relative (the product carries < 2^-22 of rounding), so a later
// pair whose tested product does not exceed it has a strictly smaller |r| -- the invariant of DESIGN.md section 4.2.
#define FIC_BF16_LEVEL 0.99999237060546875f
#define FIC_TAU_ALL 8192.0f                // > |cov| / sqrt(var) for any pair (|cov| <= sqrt(n)*255*sqrt(var) <= 4080*sqrt(var))

// two integers in [-255, 255] -> two bf16 (exact: 8 significant bits), element 0 in the low half
__device__ __forceinline__ int bf16_pair(int lo, int hi)
{
    return (int)((__float_as_uint((float)lo) >> 16) | (__float_as_uint((float)hi) & 0xFFFF0000u));
}
// 8 consecutive u8 pixels (two little-endian dwords) minus mean -> one fragment
__device__ __forceinline__ v4i bf16_fragment(uint32_t w0, uint32_t w1, int mean)
{
    v4i v;
    v[0] = bf16_pair((int)(w0 & 255u) - mean, (int)((w0 >> 8) & 255u) - mean);
    v[1] = bf16_pair((int)((w0 >> 16) & 255u) - mean, (int)(w0 >> 24) - mean);
    v[2] = bf16_pair((int)(w1 & 255u) - mean, (int)((w1 >> 8) & 255u) - mean);
    v[3] = bf16_pair((int)((w1 >> 16) & 255u) - mean, (int)(w1 >> 24) - mean);
    return v;
}

// pool -> fragments of d - dM (blocks beyond N_d: zeros, covariance exactly 0), and per domain tile the 32 values
// w = fl(1 / (float)sqrt(var)) (0 for flat blocks and padding): in block order for k_sweep_bf16 (the lane's own
// column), in accumulator-row order [half][e] for k_sweep_bf16_1 (the 16 rows a lane holds).
__global__ __launch_bounds__(256) void k_pool_bf16(const uint8_t* __restrict__ pool_pix, const FicDomStat* __restrict__ pool_st,
                                                   v4i* __restrict__ poolF, float* __restrict__ pool_w, FicGeom g,
                                                   int ndtiles_alloc, int NK, int iso8)
{
    int i = blockIdx.x * 256 + threadIdx.x;                  // (dtile, m, lane)
    int plane = blockIdx.y;
    if (i >= ndtiles_alloc * NK * 64) return;
    int lane = i & 63, m = (i >> 6) % NK, dtile = (i >> 6) / NK;
    int d = dtile * 32 + (lane & 31), h = lane >> 5;
    int off = 16 * m + 8 * h;
    v4i v = {0, 0, 0, 0};
    if (d < g.Nd) {
        const int dM = (int)(pool_st[(size_t)plane * g.Nd_pad + d].sum >> g.lgn);     // sum / n, n = 2^lgn
        const uint2 px = *(const uint2*)(pool_pix + ((size_t)plane * g.Nd_pad + d) * g.n + off);
        v = bf16_fragment(px.x, px.y, dM);
    }
    poolF[(size_t)plane * ndtiles_alloc * NK * 64 + i] = v;
    if (m == 0 && lane < 32) {
        int hh = lane >> 4, e = lane & 15;
        int dd = iso8 ? d : dtile * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        float w = 0.0f;
        if (dd < g.Nd) {
            const float s32 = pool_st[(size_t)plane * g.Nd_pad + dd].s32;
            w = s32 == 0.0f ? 0.0f : __fdiv_rn(1.0f, s32);
        }
        pool_w[((size_t)plane * ndtiles_alloc + dtile) * 32 + lane] = w;
    }
}

// ranges -> fragments of r - rM.  iso8: 32-row tile = 4 range blocks x 8 isometry copies, row = 8q + k
// (range 4*tile + q); otherwise 32 consecutive range blocks per tile, copy 0.  Ranges beyond Nr_pad: zeros.
__global__ __launch_bounds__(256) void k_range_bf16(const uint32_t* __restrict__ rng_pix, const FicRngStat* __restrict__ rng_st,
                                                    v4i* __restrict__ rngF, FicGeom g, int ntiles_alloc, int NK, int iso8)
{
    int i = blockIdx.x * 256 + threadIdx.x;                  // (tile, m, lane)
    int plane = blockIdx.y;
    if (i >= ntiles_alloc * NK * 64) return;
    int lane = i & 63, m = (i >> 6) % NK, tile = (i >> 6) / NK;
    int r = lane & 31, h = lane >> 5;
    int j = iso8 ? tile * 4 + (r >> 3) : tile * 32 + r;
    int k = iso8 ? (r & 7) : 0;
    int dw = (16 * m + 8 * h) / 4;
    v4i v = {0, 0, 0, 0};
    if (j < g.Nr_pad) {
        const uint32_t* rp = rng_pix + (size_t)plane * g.Nr_pad * g.n_iso * g.DW;
        const int rM = rng_st[(size_t)plane * g.Nr_pad + j].rM;
        v = bf16_fragment(rp[rng_word_index(g, j, k, dw)], rp[rng_word_index(g, j, k, dw + 1)], rM);
    }
    rngF[(size_t)plane * ntiles_alloc * NK * 64 + i] = v;
}

__device__ __forceinline__ v16f mfma_bf16(v4i a, v4i b, v16f c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// wave-wide reductions of a per-lane u32 (every lane of the wave must be active; the result is wave-uniform):
// quad swaps, half-row mirror, row mirror (DPP), then the four row results through SGPRs.
template <bool IS_MAX>
__device__ __forceinline__ uint32_t wave_reduce_u32(uint32_t v)
{
    auto op = [](uint32_t x, uint32_t y) { return IS_MAX ? (x > y ? x : y) : (x < y ? x : y); };
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return op(op(r0, r1), op(r2, r3));
}
// lexicographic minimum of the (error, candidate) keys of a wave
__device__ __forceinline__ unsigned long long wave_min_key(unsigned long long k)
{
    const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
    const uint32_t hmin = wave_reduce_u32<false>(hi);
    const uint32_t lmin = wave_reduce_u32<false>(hi == hmin ? lo : 0xFFFFFFFFu);
    return ((unsigned long long)hmin << 32) | lmin;
}

// ---------------------------------------------------------------------------------------------
// k_sweep_bf16<NK> : 8 isometries, B = 4 (NK = 1) / B = 8 (NK = 4).
//   rows (A) = range copies: a 32-row tile = 4 range blocks x 8 copies, row = 8q + k.  The result layout
//              (col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)) gives accumulator element e of EVERY lane
//              range q = e>>2 -- so tau is wave-uniform per element group -- and isometry k = (e&3) + 4*(lane>>5).
//              A wave keeps the fragments of its T = 4 row tiles in VGPRs for the whole sweep (no LDS traffic
//              in the loop); a workgroup = 4 waves = 64 range blocks.
//   cols (B) = 32 consecutive domain blocks, streamed from HBM/L2 (two VGPR buffers, the loop is unrolled by two
//              so they swap roles without moves); the four waves of a workgroup sweep the same pool chunk, so
//              three of their four loads of a tile hit in L1.
//   The MFMAs of the next tile are issued BEFORE the epilogue of the current one (two accumulator sets), so a
//   single wave keeps the matrix pipe busy while its VALU works through the prune test.
//   Prune test per pair (f32):  |cov| * fl(1/s32) > tau   (k_sweep_mfma1's form).
//   tau of a range is owned by one wave and is wave-uniform.
//   Flagged pairs (rare).  The kernel is VALU-issue bound, so the exact path is kept off the wave's instruction
//   stream: a flagged pair raises tau at once (its level is the tested product times (1 - 2^-17), reduced over
//   the wave with DPP) and is PUSHED -- covariance and candidate id, 8 bytes -- onto the wave's LDS
//   queue; the Java epilogue (f64 divide, FC:677-683) and the 64-bit atomicMin run later, 64 queued pairs at a
//   time, one per lane.  Every queued pair is evaluated before the wave exits, so raising tau on its behalf is the
//   same invariant as before (tau never exceeds the level of an evaluated lower-index pair).
//   The chunk's first domain tile is evaluated exactly in place (it initialises tau): keys and levels are reduced
//   inside the wave and ONE lane issues the atomicMin per range -- 64 lanes hammering one address cost 2 ms per
//   chunk here.
// ---------------------------------------------------------------------------------------------
struct Bf16Args {
    const v4i* poolF;                // [plane][ndtiles_alloc][NK][64]
    const float* pool_w;             // [plane][ndtiles_alloc][32]   fl(1/s32) per domain block (0: flat or padding)
    const double* pool_s64;
    const v4i* rngF;                 // [plane][nrtiles_alloc][NK][64]
    const FicRngStat* rng_st;
    unsigned long long* key;
    int Nd, Nd_pad, Nr, Nr_pad;
    int ndtiles, ndtiles_alloc, nrtiles_alloc;
    int group0, ngroups_launch;      // first workgroup-sized range group of this shard, groups in this launch
    int tiles_per_chunk, nchunks, planes;
};
// Row (column) tiles per wave: 4 at B = 4/8 (a workgroup = 64 range blocks x 8 copies, or 512 range blocks); 2 at
// B = 16, where one tile's fragments already take 64 VGPRs (the kernel then runs one wave per SIMD).
__host__ __device__ constexpr int fic_bf16_T(int NK) { return NK <= 4 ? 4 : 2; }
#define FIC_BF16_QCAP 1280           // queue entries per wave: one tile can flag at most 1024 pairs
#define FIC_BF16_QFLUSH 256          // evaluate the queue before a tile's pushes once it holds this many

template <int NK>
__global__ __launch_bounds__(256) void k_sweep_bf16(Bf16Args A)
{
    constexpr int T = fic_bf16_T(NK), RT = 4 * T;
    __shared__ uint2 sQ[4][FIC_BF16_QCAP];             // per wave: {covariance (f32 bits), candidate | range-in-wave << 27}
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int combo_, gx_;                                   // XCD-aware 1-D grid: combo = (pool chunk, plane), x = range group
    xcd_decode(blockIdx.x, A.nchunks * A.planes, A.ngroups_launch, combo_, gx_);
    const int group = A.group0 + gx_;
    const int plane = combo_ / A.nchunks;
    const int chunk = combo_ % A.nchunks;
    const int jw = (group * RT + wave * T) * 4;        // first range block of this wave (16 range blocks)
    const FicRngStat* rst = A.rng_st + (size_t)plane * A.Nr_pad;
    const int dt0 = chunk * A.tiles_per_chunk;
    int dt1 = dt0 + A.tiles_per_chunk;
    if (dt1 > A.ndtiles) dt1 = A.ndtiles;
    if (dt0 >= dt1) return;

    v4i a[T][NK];
    {
        const v4i* ra = A.rngF + ((size_t)plane * A.nrtiles_alloc + (size_t)group * RT + wave * T) * NK * 64 + lane;
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int m = 0; m < NK; m++) a[t][m] = ra[(t * NK + m) * 64];
    }
    // ranges with rem == 0: error 0 for every block (FC:677) -> nothing after the first tile can win
    float tau[T][4];
    uint32_t remnz = 0;                                // bit 4t+q: rem != 0 (wave-uniform)
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = jw + 4 * t + q;
            const int rem = __builtin_amdgcn_readfirstlane(j < A.Nr ? rst[j].rem : 0);   // padding ranges: never flagged
            tau[t][q] = rem == 0 ? FIC_TAU_ALL : 0.0f;
            remnz |= rem != 0 ? 1u << (4 * t + q) : 0u;
        }

    const int jcol = lane & 31, half = lane >> 5;
    const v4i* pb = A.poolF + (size_t)plane * A.ndtiles_alloc * NK * 64 + lane;
    const float* pw = A.pool_w + (size_t)plane * A.ndtiles_alloc * 32 + jcol;
    const double* p64 = A.pool_s64 + (size_t)plane * A.Nd_pad;
    unsigned long long* keyp = A.key + (size_t)plane * A.Nr_pad;
    const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint2* const myq = sQ[wave];
    int qn = 0;                                        // queued pairs (wave-uniform)

    // Java epilogue + atomicMin for the queued pairs, one per lane
    auto flush = [&]() {
        for (int base = 0; base < qn; base += 64) {
            const int i = base + lane;
            if (i < qn) {
                const uint2 ent = myq[i];
                const int cov = (int)__uint_as_float(ent.x);
                const uint32_t cand = ent.y & 0x07FFFFFFu;
                const int j = jw + (int)(ent.y >> 27);
                const float err = exact_error(cov, rst[j].rem, p64[cand >> 3]);
                atomicMin(&keyp[j], ((unsigned long long)f32_orderable(err) << 32) | cand);
            }
        }
        qn = 0;
    };
    // a tile with flagged pairs (p[e] = cov * w as tested): raise tau, queue the pairs
    auto flagged_tile = [&](const v16f& acc, const float (&p)[16], const bool (&flag)[4], int t, int dt) {
        if (qn > FIC_BF16_QFLUSH) flush();
        const uint32_t d8 = (uint32_t)(dt * 32 + jcol) * 8u + (uint32_t)(4 * half);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = jw + 4 * t + q;                              // wave-uniform
            if (__builtin_amdgcn_ballot_w64(flag[q]) == 0 || j >= A.Nr) continue;
            float mp = 0.0f;                                           // largest flagged |cov * w| on this lane
#pragma unroll
            for (int ee = 0; ee < 4; ee++) {
                const int e = 4 * q + ee;
                const bool pe = fabsf(p[e]) > tau[t][q];               // implies a valid, non-flat block (w > 0)
                const unsigned long long be = __builtin_amdgcn_ballot_w64(pe);
                if (be == 0) continue;
                const int idx = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(be >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)be, 0u));
                if (pe) {
                    myq[idx] = make_uint2(__float_as_uint(acc[e]), (d8 + (uint32_t)ee) | ((uint32_t)(4 * t + q) << 27));
                    mp = fmaxf(mp, fabsf(p[e]));
                }
                qn += __builtin_popcountll(be);
            }
            const float lmax = __uint_as_float(wave_reduce_u32<true>(__float_as_uint(__fmul_rn(mp, FIC_BF16_LEVEL))));   // >= 0: uint order == float order
            if (remnz & (1u << (4 * t + q))) tau[t][q] = fmaxf(tau[t][q], lmax);
        }
    };
    // the chunk's first domain tile: every pair evaluated exactly, in place
    auto forced_tile = [&](const v16f& acc, int t, int dt, float w) {
        const int d = dt * 32 + jcol;
        const bool valid = d < A.Nd;
        const double s64 = valid ? p64[d] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = jw + 4 * t + q;                              // wave-uniform
            if (j >= A.Nr) continue;
            const int rem = rst[j].rem;
            unsigned long long best = FIC_KEY_NONE;
            float mp = 0.0f;
            if (valid) {
#pragma unroll
                for (int ee = 0; ee < 4; ee++) {
                    const int cov = (int)acc[4 * q + ee];
                    const float err = exact_error(cov, rem, s64);
                    const unsigned long long k = ((unsigned long long)f32_orderable(err) << 32) |
                                                 ((uint32_t)d * 8u + (uint32_t)(ee + 4 * half));
                    best = k < best ? k : best;
                    mp = fmaxf(mp, fabsf(__fmul_rn(acc[4 * q + ee], w)));
                }
            }
            const unsigned long long kmin = wave_min_key(best);        // one atomic per (wave, range)
            if (lane == 0 && kmin != FIC_KEY_NONE) atomicMin(&keyp[j], kmin);
            const float lmax = __uint_as_float(wave_reduce_u32<true>(__float_as_uint(__fmul_rn(mp, FIC_BF16_LEVEL))));
            if (rem != 0) tau[t][q] = fmaxf(tau[t][q], lmax);
        }
    };
    auto tile_mfma = [&](const v4i (&at)[NK], const v4i (&bt)[NK]) {
        v16f acc = zero;
#pragma unroll
        for (int m = 0; m < NK; m++) acc = mfma_bf16(at[m], bt[m], acc);
        return acc;
    };

    v4i b0[NK], b1[NK];
#pragma unroll
    for (int m = 0; m < NK; m++) b0[m] = pb[((size_t)dt0 * NK + m) * 64];
    float w = pw[(size_t)dt0 * 32];
    v16f acc = tile_mfma(a[0], b0);

    // one domain tile: bc = its fragments, bn = buffer for the next tile's (the fragment store has one spare tile)
    auto step = [&](auto forced, int dt, const v4i (&bc)[NK], v4i (&bn)[NK]) {
        constexpr bool FORCE = decltype(forced)::value;
#pragma unroll
        for (int m = 0; m < NK; m++) bn[m] = pb[((size_t)(dt + 1) * NK + m) * 64];
        const float nw = pw[(size_t)(dt + 1) * 32];
#pragma unroll
        for (int t = 0; t < T; t++) {
            // issue the next tile's MFMAs first: (t+1, dt), or (0, dt+1) on the prefetched fragments
            v16f nacc;
            if (t + 1 < T) nacc = tile_mfma(a[t + 1], bc);
            else nacc = tile_mfma(a[0], bn);
            if constexpr (FORCE) {
                forced_tile(acc, t, dt, w);
            } else {
                // acc[e] = kovarianz of (range 4t+q copy k, domain d), exact
                float p[16];
#pragma unroll
                for (int e = 0; e < 16; e++) p[e] = __fmul_rn(acc[e], w);   // |fl(cov * w)| == fl(|cov| * w)
                bool flag[4];
                bool any = false;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float mx = fmaxf(fmaxf(fabsf(p[4 * q]), fabsf(p[4 * q + 1])), fmaxf(fabsf(p[4 * q + 2]), fabsf(p[4 * q + 3])));
                    flag[q] = mx > tau[t][q];
                    any |= flag[q];
                }
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) != 0, 0)) flagged_tile(acc, p, flag, t, dt);
            }
            acc = nacc;
        }
        w = nw;
    };
    int dt = dt0;
    step(std::true_type{}, dt, b0, b1);
    dt++;
    for (; dt + 1 < dt1; dt += 2) {
        step(std::false_type{}, dt, b1, b0);
        step(std::false_type{}, dt + 1, b0, b1);
    }
    if (dt < dt1) step(std::false_type{}, dt, b1, b0);
    flush();
}

// ---------------------------------------------------------------------------------------------
// k_sweep_bf16_1<NK> : the reference algorithm (n_iso = 1), B = 4 / 8.  Transposed mapping, as in k_sweep_mfma1:
//   rows (A) = 32 consecutive domain blocks, streamed; cols (B) = 32 range blocks; a wave keeps the fragments of
//   its CTW = 4 column tiles in VGPRs; a workgroup = 16 column tiles = 512 range blocks.
//   col = lane&31 is the lane's range block: its tau sits in one VGPR per column tile (LDS copy so that the two
//   lane halves of a column see each other's updates: ds_max_u32 when a pair is flagged, re-read afterwards).
//   row = (e&3) + 8(e>>2) + 4(lane>>5) is element e's domain block: w = fl(1/s32) of the lane's 16 rows comes as
//   four 16-byte loads per domain tile (stored per tile in exactly that order by k_pool_bf16).
// Prune test per pair (f32):  |cov| * fl(1/s32) > tau  -- k_sweep_mfma1's test.
// As in k_sweep_bf16: the next tile's MFMAs are issued before the current tile's epilogue; flagged pairs raise tau
// at once and are queued in LDS, the Java epilogue and the atomicMin run later, one queued pair per lane (here
// one tile in six flags something -- 128 range blocks per wave, a dozen records each -- so the deferred epilogue
// matters more than in the 8-isometry kernel); the chunk's first domain tile is evaluated exactly in place, one
// atomicMin per lane.
// ---------------------------------------------------------------------------------------------
struct Bf16Args1 {
    const v4i* poolF;
    const float* pool_w;             // [plane][ndtiles_alloc][2][16]
    const double* pool_s64;
    const v4i* rngF;                 // [plane][nctiles_alloc][NK][64]
    const FicRngStat* rng_st;
    unsigned long long* key;
    int Nd, Nd_pad, Nr, Nr_pad;
    int ndtiles, ndtiles_alloc, nctiles_alloc;
    int ct_begin, ct_end;            // column tiles (x32 ranges) of this shard
    int nctg;                        // column-tile groups (workgroups) in this launch
    int tiles_per_chunk, nchunks, planes;
};

template <int NK>
__global__ __launch_bounds__(256) void k_sweep_bf16_1(Bf16Args1 A)
{
    constexpr int CTW = fic_bf16_T(NK), CT = 4 * CTW;
    __shared__ uint32_t sTau[CT * 32];
    __shared__ uint2 sQ[4][FIC_BF16_QCAP];             // per wave: {covariance (f32 bits), domain block | range-in-wave << 24}
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int combo_, gx_;
    xcd_decode(blockIdx.x, A.nchunks * A.planes, A.nctg, combo_, gx_);
    const int plane = combo_ / A.nchunks;
    const int chunk = combo_ % A.nchunks;
    const int ct0 = A.ct_begin + gx_ * CT;                   // first column tile of the workgroup
    const int ctw0 = ct0 + wave * CTW;                       // ... of this wave
    const FicRngStat* rst = A.rng_st + (size_t)plane * A.Nr_pad;
    for (int i = threadIdx.x; i < CT * 32; i += 256) {
        const int j = ct0 * 32 + i;
        const int rem = j < A.Nr ? rst[j].rem : 0;                 // padding ranges: never flagged
        sTau[i] = rem == 0 ? __float_as_uint(FIC_TAU_ALL) : 0u;   // rem == 0: nothing after the first block can win
    }
    v4i rb[CTW][NK];
    {
        const v4i* rp = A.rngF + ((size_t)plane * A.nctiles_alloc + ctw0) * NK * 64 + lane;
#pragma unroll
        for (int ci = 0; ci < CTW; ci++)
#pragma unroll
            for (int m = 0; m < NK; m++) rb[ci][m] = rp[(ci * NK + m) * 64];
    }
    __syncthreads();
    const int dt0 = chunk * A.tiles_per_chunk;
    int dt1 = dt0 + A.tiles_per_chunk;
    if (dt1 > A.ndtiles) dt1 = A.ndtiles;
    if (dt0 >= dt1 || ctw0 >= A.ct_end) return;
    int nci = A.ct_end - ctw0;                               // column tiles this wave really owns (wave-uniform)
    if (nci > CTW) nci = CTW;

    const int jcol = lane & 31, half = lane >> 5;
    float tau[CTW];
    uint32_t raise = 0;                                      // bit ci: this lane's range block has rem != 0 and exists
#pragma unroll
    for (int ci = 0; ci < CTW; ci++) {
        const uint32_t t0 = sTau[(wave * CTW + ci) * 32 + jcol];
        tau[ci] = __uint_as_float(t0);
        raise |= (t0 == 0u && (ctw0 + ci) * 32 + jcol < A.Nr) ? 1u << ci : 0u;
    }

    const v4i* pa = A.poolF + (size_t)plane * A.ndtiles_alloc * NK * 64 + lane;
    const float4* pw = (const float4*)(A.pool_w + (size_t)plane * A.ndtiles_alloc * 32) + half * 4;
    const double* p64 = A.pool_s64 + (size_t)plane * A.Nd_pad;
    unsigned long long* keyp = A.key + (size_t)plane * A.Nr_pad;
    const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint2* const myq = sQ[wave];
    int qn = 0;                                              // queued pairs (wave-uniform)

    // Java epilogue + atomicMin for the queued pairs, one per lane
    auto flush = [&]() {
        for (int base = 0; base < qn; base += 64) {
            const int i = base + lane;
            if (i < qn) {
                const uint2 ent = myq[i];
                const int cov = (int)__uint_as_float(ent.x);
                const uint32_t d = ent.y & 0x00FFFFFFu;
                const int j = ctw0 * 32 + (int)(ent.y >> 24);
                const float err = exact_error(cov, rst[j].rem, p64[d]);
                atomicMin(&keyp[j], ((unsigned long long)f32_orderable(err) << 32) | d);
            }
        }
        qn = 0;
    };
    // publish a lane's new level for its range block and pick up the other lane half's
    auto raise_tau = [&](int ci, float mp) {
        const uint32_t lb = __float_as_uint(__fmul_rn(mp, FIC_BF16_LEVEL));     // >= 0: uint order == float order
        if (((raise >> ci) & 1u) && lb != 0u) atomicMax(&sTau[(wave * CTW + ci) * 32 + jcol], lb);
        tau[ci] = __uint_as_float(sTau[(wave * CTW + ci) * 32 + jcol]);
    };
    // a tile with flagged pairs (p[e] = cov * w as tested): queue them, raise tau
    auto flagged_tile = [&](const v16f& acc, const float (&p)[16], int ci, int dt) {
        if (qn > FIC_BF16_QFLUSH) flush();
        const bool lane_ok = (ctw0 + ci) * 32 + jcol < A.Nr;
        const uint32_t tag = (uint32_t)(dt * 32 + 4 * half) | ((uint32_t)(ci * 32 + jcol) << 24);
        float mp = 0.0f;                                     // largest flagged |cov * w| on this lane
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const bool pe = lane_ok && fabsf(p[e]) > tau[ci];                  // implies a real, non-flat block (w > 0)
            const unsigned long long be = __builtin_amdgcn_ballot_w64(pe);
            if (be == 0) continue;
            const int idx = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(be >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)be, 0u));
            if (pe) {
                myq[idx] = make_uint2(__float_as_uint(acc[e]), tag + (uint32_t)((e & 3) + 8 * (e >> 2)));
                mp = fmaxf(mp, fabsf(p[e]));
            }
            qn += __builtin_popcountll(be);
        }
        raise_tau(ci, mp);
    };
    // the chunk's first domain tile: every pair evaluated exactly, in place; one atomicMin per lane
    auto forced_tile = [&](const v16f& acc, const float (&p)[16], int ci, int dt) {
        const int j = (ctw0 + ci) * 32 + jcol;
        float mp = 0.0f;
        if (j < A.Nr) {
            const int rem = rst[j].rem;
            unsigned long long best = FIC_KEY_NONE;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int d = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (d < A.Nd) {
                    const float err = exact_error((int)acc[e], rem, p64[d]);
                    const unsigned long long k = ((unsigned long long)f32_orderable(err) << 32) | (uint32_t)d;
                    best = k < best ? k : best;
                    mp = fmaxf(mp, fabsf(p[e]));
                }
            }
            if (best != FIC_KEY_NONE) atomicMin(&keyp[j], best);
        }
        raise_tau(ci, mp);
    };
    auto tile_mfma = [&](const v4i (&at)[NK], const v4i (&bt)[NK]) {
        v16f acc = zero;
#pragma unroll
        for (int m = 0; m < NK; m++) acc = mfma_bf16(at[m], bt[m], acc);
        return acc;
    };

    v4i a0[NK], a1[NK];
#pragma unroll
    for (int m = 0; m < NK; m++) a0[m] = pa[((size_t)dt0 * NK + m) * 64];
    v16f acc = tile_mfma(a0, rb[0]);

    // one domain tile: ac = its fragments, an = buffer for the next tile's (the fragment store has one spare tile)
    auto step = [&](auto forced, int dt, const v4i (&ac)[NK], v4i (&an)[NK]) {
        constexpr bool FORCE = decltype(forced)::value;
#pragma unroll
        for (int m = 0; m < NK; m++) an[m] = pa[((size_t)(dt + 1) * NK + m) * 64];
        float w[16];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const float4 w4 = pw[(size_t)dt * 8 + v];
            w[4 * v] = w4.x; w[4 * v + 1] = w4.y; w[4 * v + 2] = w4.z; w[4 * v + 3] = w4.w;
        }
#pragma unroll
        for (int ci = 0; ci < CTW; ci++) {
            // issue the next tile's MFMAs first: column tile ci+1 of this domain tile, or column tile 0 of the next
            v16f nacc;
            if (ci + 1 < CTW) nacc = tile_mfma(ac, rb[ci + 1]);
            else nacc = tile_mfma(an, rb[0]);
            if (ci < nci) {                                  // wave-uniform
                float p[16];
#pragma unroll
                for (int e = 0; e < 16; e++) p[e] = __fmul_rn(acc[e], w[e]);   // |fl(cov * w)| == fl(|cov| * w)
                if constexpr (FORCE) {
                    forced_tile(acc, p, ci, dt);
                } else {
                    float mx = fabsf(p[0]);
#pragma unroll
                    for (int e = 1; e < 16; e++) mx = fmaxf(mx, fabsf(p[e]));
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx > tau[ci]) != 0, 0)) flagged_tile(acc, p, ci, dt);
                }
            }
            acc = nacc;
        }
    };
    int dt = dt0;
    step(std::true_type{}, dt, a0, a1);
    dt++;
    for (; dt + 1 < dt1; dt += 2) {
        step(std::false_type{}, dt, a1, a0);
        step(std::false_type{}, dt + 1, a0, a1);
    }
    if (dt < dt1) step(std::false_type{}, dt, a1, a0);
    flush();
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
int fic_bf16_steps(int B) { return B * B / 16; }
int fic_bf16_group8(int B) { return 16 * fic_bf16_T(B * B / 16); }   // range blocks per workgroup, 8-isometry kernel
int fic_bf16_ct1(int B) { return 4 * fic_bf16_T(B * B / 16); }       // column tiles (x32 ranges) per workgroup, 1-isometry kernel

int fic_launch_bf16_prep(const FicBuffers& b, void* poolF, void* pool_w, void* rngF, const FicGeom& g, int ndtiles_alloc,
                         int nrtiles_alloc, hipStream_t s)
{
    const int NK = fic_bf16_steps(g.B);
    hipLaunchKernelGGL(k_pool_bf16, dim3((ndtiles_alloc * NK * 64 + 255) / 256, g.planes), dim3(256), 0, s,
                       (const uint8_t*)b.pool_pix, (const FicDomStat*)b.pool_st, (v4i*)poolF, (float*)pool_w, g,
                       ndtiles_alloc, NK, g.n_iso == 8 ? 1 : 0);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_range_bf16, dim3((nrtiles_alloc * NK * 64 + 255) / 256, g.planes), dim3(256), 0, s,
                       (const uint32_t*)b.rng_pix, (const FicRngStat*)b.rng_st, (v4i*)rngF, g, nrtiles_alloc, NK,
                       g.n_iso == 8 ? 1 : 0);
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_sweep_bf16(const FicBuffers& b, const void* poolF, const void* pool_w, const void* rngF, const FicGeom& g,
                          int nrtiles_alloc,
                          int group0, int ngroups_launch, int ndtiles, int ndtiles_alloc, int tiles_per_chunk, int nchunks,
                          hipStream_t s)
{
    Bf16Args A;
    A.poolF = (const v4i*)poolF; A.pool_w = (const float*)pool_w; A.pool_s64 = b.pool_s64;
    A.rngF = (const v4i*)rngF; A.rng_st = b.rng_st; A.key = b.key;
    A.Nd = g.Nd; A.Nd_pad = g.Nd_pad; A.Nr = g.Nr; A.Nr_pad = g.Nr_pad;
    A.ndtiles = ndtiles; A.ndtiles_alloc = ndtiles_alloc; A.nrtiles_alloc = nrtiles_alloc;
    A.group0 = group0; A.ngroups_launch = ngroups_launch; A.tiles_per_chunk = tiles_per_chunk; A.nchunks = nchunks;
    A.planes = g.planes;
    if ((group0 + ngroups_launch) * (fic_bf16_group8(g.B) / 4) > nrtiles_alloc || g.Nd >= (1 << 24)) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)(nchunks * g.planes) * (unsigned)ngroups_launch), block(256);
    if (g.B == 4) hipLaunchKernelGGL((k_sweep_bf16<1>), grid, block, 0, s, A);
    else if (g.B == 8) hipLaunchKernelGGL((k_sweep_bf16<4>), grid, block, 0, s, A);
    else if (g.B == 16) hipLaunchKernelGGL((k_sweep_bf16<16>), grid, block, 0, s, A);
    else return (int)hipErrorInvalidValue;
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_sweep_bf16_1(const FicBuffers& b, const void* poolF, const void* pool_w, const void* rngF, const FicGeom& g,
                            int ct_begin, int ct_end, int ndtiles, int ndtiles_alloc, int nctiles_alloc, int tiles_per_chunk,
                            int nchunks, hipStream_t s)
{
    Bf16Args1 A;
    A.poolF = (const v4i*)poolF; A.pool_w = (const float*)pool_w; A.pool_s64 = b.pool_s64;
    A.rngF = (const v4i*)rngF; A.rng_st = b.rng_st; A.key = b.key;
    A.Nd = g.Nd; A.Nd_pad = g.Nd_pad; A.Nr = g.Nr; A.Nr_pad = g.Nr_pad;
    A.ndtiles = ndtiles; A.ndtiles_alloc = ndtiles_alloc; A.nctiles_alloc = nctiles_alloc;
    A.ct_begin = ct_begin; A.ct_end = ct_end; A.tiles_per_chunk = tiles_per_chunk; A.nchunks = nchunks;
    const int CT = fic_bf16_ct1(g.B);
    A.nctg = (ct_end - ct_begin + CT - 1) / CT;
    A.planes = g.planes;
    if (ct_begin + A.nctg * CT > nctiles_alloc || g.Nd >= (1 << 24)) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)(nchunks * g.planes) * (unsigned)A.nctg), block(256);
    if (g.B == 4) hipLaunchKernelGGL((k_sweep_bf16_1<1>), grid, block, 0, s, A);
    else if (g.B == 8) hipLaunchKernelGGL((k_sweep_bf16_1<4>), grid, block, 0, s, A);
    else if (g.B == 16) hipLaunchKernelGGL((k_sweep_bf16_1<16>), grid, block, 0, s, A);
    else return (int)hipErrorInvalidValue;
    FIC_LAUNCH_CHECK();
    return 0;
}
