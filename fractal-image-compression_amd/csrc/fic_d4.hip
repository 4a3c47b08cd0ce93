// fic_d4.hip -- k_sweep_d4: the VALU full-pool sweep for n_iso = 8 with the isometries taken algebraically.
// gfx950 (MI355X / CDNA4) only, wave64.  Compile with -ffp-contract=off (FractalCompression.java = FC).
//
// k_sweep_fast<.,1,8> keeps the 8 isometry copies of a range block in VGPRs and spends 8 * n/4 v_dot4 per (range,
// domain) pair.  The 8 isometries are the dihedral group D4 acting on the B x B pixel positions, and on every orbit of
// positions the 8 inner products  dot_k = sum_pos copy_k(r)[pos] * d[pos]  are a correlation ON THE GROUP, which D4's
// irreducible representations (four 1-dimensional, one 2-dimensional) diagonalise:
//     16 * dot_k = base[form[k].b] + form[k].s * e[form[k].e]
//     b0 = T0+T1+T2+T3, b1 = T0+T1-T2-T3, b2 = T0-T1+T2-T3, b3 = T0-T1-T2+T3, e0 = T4+T5, e1 = T4-T5, e2 = T6+T7, e3 = T6-T7
//     T[acc] = sum over the accumulator's slots of U[slot] * V[slot]
// where U (range side) and V (domain side) are signed sums of <= 8 pixels (|U| <= 2040, |V| <= 4080: i16) --
// n + n/2 products instead of 8n, as v_dot2c_i32_i16 pairs: 48 per pair at B = 8 instead of 128 dot4.  All of it is
// integer and exact, so kovarianz (FC:665-672) of every copy is the same integer as before and everything downstream
// (prune test, exact epilogue, strict-'<' order) is k_sweep_fast's code path.  The tables come from
// tools/gen_d4_tables.py, which derives them from iso_source / iso_inverse and checks the closed form against brute force.
// Because the largest and the smallest of {b + e, b - e} are b + |e| and b - |e|, the prune test needs only four sums
// and four differences.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "fic_device.h"
#include "fic_launch.h"
#include "fic_devfn.h"
#include "fic_d4_tables.h"

typedef short s16x2 __attribute__((ext_vector_type(2)));
#define D4_BIAS 0x10000000u               // 2^28 > every |T[acc]| (<= 16 slots * 2040 * 4080 < 2^27)
#define D4_BIAS_B 0x40000000u             // 2^30 > |T0 + T1|, |T2 + T3| (< 2^28 + 2^26 with the mean correction in T0)

template <int B> struct D4;
template <> struct D4<8> {
    static constexpr int SLOTS = FIC_D4_B8_SLOTS, TERMS = FIC_D4_B8_TERMS;
    static constexpr const int16_t (*U)[FIC_D4_B8_TERMS][2] = fic_d4_b8_U;
    static constexpr const int16_t (*V)[FIC_D4_B8_TERMS][2] = fic_d4_b8_V;
    static constexpr const int* acc = fic_d4_b8_acc;
    static constexpr const int (*form)[3] = fic_d4_b8_form;
};
// One block (n pixels in registers) -> its NW packed slot pairs.  The table is walked at compile time: every slot is a
// signed sum of <= 8 pixels, so the whole transform is ~n*12 integer adds with constant operands.
template <int B, bool RANGE_SIDE>
__device__ __forceinline__ void d4_slots(const int (&px)[B * B], uint32_t (&out)[D4<B>::SLOTS / 2])
{
#pragma unroll
    for (int w = 0; w < D4<B>::SLOTS / 2; w++) {
        int v[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            int acc = 0;
#pragma unroll
            for (int t = 0; t < D4<B>::TERMS; t++) {
                const int pos = RANGE_SIDE ? D4<B>::U[2 * w + h][t][0] : D4<B>::V[2 * w + h][t][0];
                const int wt = RANGE_SIDE ? D4<B>::U[2 * w + h][t][1] : D4<B>::V[2 * w + h][t][1];
                if (wt != 0) acc += wt * px[pos];
            }
            v[h] = acc;
        }
        out[w] = ((uint32_t)v[0] & 0xFFFFu) | ((uint32_t)v[1] << 16);
    }
}

// ranges -> U slots, two i16 per dword, lane-transposed like rng_pix: rng_d4[plane][tile][word][64].  One thread per
// range block; the stores of a wave are 256-byte runs.
template <int B>
__global__ __launch_bounds__(64) void k_range_d4(const uint8_t* __restrict__ gray, uint32_t* __restrict__ rng_d4, FicGeom g)
{
    constexpr int NW = D4<B>::SLOTS / 2;
    const int plane = blockIdx.y, tile = blockIdx.x, lane = threadIdx.x;
    const int j = tile * 64 + lane;
    uint32_t out[NW];
    if (j < g.Nr) {
        const uint8_t* blk = gray + (size_t)plane * g.W * g.H + (size_t)((j / g.Rw) * B) * g.W + (j % g.Rw) * B;
        int px[B * B];
#pragma unroll
        for (int y = 0; y < B; y++)
#pragma unroll
            for (int x4 = 0; x4 < B / 4; x4++) {
                const uint32_t q = *(const uint32_t*)(blk + (size_t)y * g.W + 4 * x4);   // W, B multiples of 4: aligned
#pragma unroll
                for (int t = 0; t < 4; t++) px[y * B + 4 * x4 + t] = (int)((q >> (8 * t)) & 255u);
            }
        d4_slots<B, true>(px, out);
    } else {
#pragma unroll
        for (int w = 0; w < NW; w++) out[w] = 0;
    }
    uint32_t* dst = rng_d4 + ((size_t)plane * g.tiles + tile) * NW * 64 + lane;
#pragma unroll
    for (int w = 0; w < NW; w++) dst[(size_t)w * 64] = out[w];
}

// pool -> V slots, two i16 per dword, one record of NW dwords per domain block (the wave-uniform operand of the sweep)
template <int B>
__global__ __launch_bounds__(64) void k_pool_d4(const uint8_t* __restrict__ pool_pix, uint32_t* __restrict__ pool_d4, FicGeom g)
{
    constexpr int NW = D4<B>::SLOTS / 2;
    const int plane = blockIdx.y;
    const int d = blockIdx.x * 64 + threadIdx.x;
    if (d >= g.Nd_pad) return;
    uint32_t out[NW];
    if (d < g.Nd) {
        const uint32_t* blk = (const uint32_t*)(pool_pix + ((size_t)plane * g.Nd_pad + d) * g.n);
        int px[B * B];
#pragma unroll
        for (int q4 = 0; q4 < B * B / 4; q4++) {
            const uint32_t q = blk[q4];
#pragma unroll
            for (int t = 0; t < 4; t++) px[4 * q4 + t] = (int)((q >> (8 * t)) & 255u);
        }
        d4_slots<B, false>(px, out);
    } else {
#pragma unroll
        for (int w = 0; w < NW; w++) out[w] = 0;
    }
    uint4* dst = (uint4*)(pool_d4 + ((size_t)plane * g.Nd_pad + d) * NW);
#pragma unroll
    for (int w = 0; w < NW; w += 4) dst[w / 4] = make_uint4(out[w], out[w + 1], out[w + 2], out[w + 3]);
}

struct D4Args {
    const uint32_t* pool_d4;
    const FicDomStat* pool_st;
    const double* pool_s64;
    const uint32_t* rng_d4;
    const FicRngStat* rng_st;
    unsigned long long* key;
    int Nd, Nd_pad, Nr, Nr_pad, lgn, tiles;
    int tile0, ntiles;         // tiles [tile0, tile0+ntiles) of 64 ranges
    int chunk_len, nchunks;
    int planes;
};

template <int B>
__global__ __launch_bounds__(256) void k_sweep_d4(D4Args A)
{
    constexpr int NW = D4<B>::SLOTS / 2;                // dot2 pairs per (range, domain)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int combo_, bx_;                                    // XCD-aware 1-D grid: combo = (pool chunk, plane), x = range-tile group
    xcd_decode(blockIdx.x, A.nchunks * A.planes, (A.ntiles + 3) / 4, combo_, bx_);
    const int chunk = combo_ % A.nchunks, plane = combo_ / A.nchunks;
    const int tl = bx_ * 4 + wave;
    if (tl >= A.ntiles) return;                         // wave-uniform
    const int tile = A.tile0 + tl;
    const int d0 = __builtin_amdgcn_readfirstlane(chunk * A.chunk_len);   // wave-uniform: scalar address arithmetic
    int d1 = d0 + A.chunk_len;
    if (d1 > A.Nd) d1 = A.Nd;
    d1 = __builtin_amdgcn_readfirstlane(d1);
    if (d0 >= d1) return;

    // ---- this lane's range block: its U slots stay in VGPRs for the whole sweep ----
    uint32_t ru[NW];
    {
        const uint32_t* rp = A.rng_d4 + ((size_t)plane * A.tiles + tile) * NW * 64 + lane;
#pragma unroll
        for (int w = 0; w < NW; w++) ru[w] = rp[(size_t)w * 64];
    }
    const int j = tile * 64 + lane;
    const FicRngStat rst = A.rng_st[(size_t)plane * A.Nr_pad + j];
    const int neg_rM = -rst.rM, neg_rem = -rst.rem;
    float best_err = INFINITY, tau = 0.0f;              // tau = 0 and a forced first block, as k_sweep_fast's 8-copy variants
    uint32_t best_cand = 0xFFFFFFFFu;

    const AS4 uint32_t* px = (const AS4 uint32_t*)(A.pool_d4 + (size_t)plane * A.Nd_pad * NW);
    const AS4 u32x2* st = (const AS4 u32x2*)(A.pool_st + (size_t)plane * A.Nd_pad);
    const AS4 double* s64p = (const AS4 double*)(A.pool_s64 + (size_t)plane * A.Nd_pad);

    // The domain record (NW dwords) lives in SGPRs; the next record is requested as soon as the last v_dot2c of the
    // current one has been issued, so its latency overlaps the combine / test part of the loop body.
    uint32_t pv[NW];
    u32x2 sc;
    const AS4 uint32_t* pcur = px + (size_t)d0 * NW;    // wave-uniform running pointer
#pragma unroll
    for (int w = 0; w < NW; w++) pv[w] = pcur[w];
    sc = st[d0];
    for (int d = d0; d < d1; d++) {
        __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): record d landed
        const int Sd = (int)sc.x, dM = (int)(sc.x >> A.lgn);
        const float s32 = __uint_as_float(sc.y);
        // 16 * -(rM*sum(d) + dM*rem) rides in T0, which enters all four bases with +; the other accumulators start from
        // their first product (no zeroing moves).  The table orders the dot2 pairs round-robin over the accumulators:
        // consecutive v_dot2c are independent, and each half of the record can be re-requested (record d + 1; the pool
        // tail is padded: Nd_pad > Nd) as soon as its last v_dot2c has been issued.
        int T[8];
        T[0] = 16 * (__mul24(neg_rM, Sd) + __mul24(neg_rem, dM)) + (int)D4_BIAS_B;
        T[2] = (int)D4_BIAS_B;
        T[4] = T[5] = T[6] = T[7] = (int)D4_BIAS;
        auto dot = [&](int w) {
            const int a = D4<B>::acc[w];
            bool first = a == 1 || a == 3;              // T1, T3 start from their first product; the others are preloaded
#pragma unroll
            for (int v = 0; v < w; v++) first = first && (D4<B>::acc[v] != a);
            const int c = first ? 0 : T[a];
            T[a] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, ru[w]), __builtin_bit_cast(s16x2, pv[w]), c, false);
        };
        pcur += NW;                                     // record d + 1
#pragma unroll
        for (int w = 0; w < NW / 2; w++) dot(w);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = 0; w < NW / 2; w++) pv[w] = pcur[w];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = NW / 2; w < NW; w++) dot(w);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = NW / 2; w < NW; w++) pv[w] = pcur[w];
        sc = st[d + 1];
        __builtin_amdgcn_sched_barrier(0);
        // Prune test: the largest |16*cov_k| over the 8 copies is max_i (|b_i| + |e_j(i)|)  (|b + e| and |b - e| peak at
        // |b| + |e|), with b = u +/- x and e = T4 +/- T5 or T6 +/- T7.  T0 and T2 carry a bias of 2^30, T4..T7 one of 2^28
        // (every |sum| stays below its bias), so each |p +/- q| is an UNSIGNED absolute difference and v_sad_u32 adds the
        // other magnitude in the same instruction: two v_sad_u32 per i.
        const uint32_t u0 = (uint32_t)(T[0] + T[1]), u1 = (uint32_t)(T[0] - T[1]);            // u + 2^30
        const uint32_t x0 = (uint32_t)(T[2] + T[3]), x1 = (uint32_t)(T[2] - T[3]);            // x + 2^30
        const uint32_t nx0 = 2u * D4_BIAS_B - x0, nx1 = 2u * D4_BIAS_B - x1;                  // -x + 2^30
        const uint32_t t4 = (uint32_t)T[4], t5 = (uint32_t)T[5], t6 = (uint32_t)T[6], t7 = (uint32_t)T[7];
        const uint32_t n5 = 2u * D4_BIAS - t5, n7 = 2u * D4_BIAS - t7;                        // -T + 2^28
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (D4<B>::form[k][2] > 0) {
                const int bi = D4<B>::form[k][0], ej = D4<B>::form[k][1];
                const uint32_t el = ej < 2 ? t4 : t6, er = ej == 0 ? n5 : ej == 1 ? t5 : ej == 2 ? n7 : t7;
                const uint32_t bl = bi < 2 ? u0 : u1, br = bi == 0 ? nx0 : bi == 1 ? x0 : bi == 2 ? nx1 : x1;
                uint32_t ae, mi;                            // operands are plain VALU results: no MFMA hazard for the asm
                asm("v_sad_u32 %0, %1, %2, 0" : "=v"(ae) : "v"(el), "v"(er));
                asm("v_sad_u32 %0, %1, %2, %3" : "=v"(mi) : "v"(bl), "v"(br), "v"(ae));
                m = max(m, mi);
            }
        const int li = (int)__fmul_rn(tau, s32);        // tau <= 8192, s32 <= 4080: exact floor
        const bool force = (d == d0);                   // wave-uniform: the chunk's first block initialises tau
        const bool any = force | (m > 16u * (uint32_t)li);
        if (__builtin_expect(__any(any), 0)) {
            const double s64 = s64p[d];
            const int e4 = T[4] - (int)D4_BIAS, e5 = T[5] - (int)D4_BIAS, e6 = T[6] - (int)D4_BIAS, e7 = T[7] - (int)D4_BIAS;
            const int e[4] = {e4 + e5, e4 - e5, e6 + e7, e6 - e7};
            const int uu0 = (int)(u0 - D4_BIAS_B), uu1 = (int)(u1 - D4_BIAS_B), xx0 = (int)(x0 - D4_BIAS_B), xx1 = (int)(x1 - D4_BIAS_B);
            const int b[4] = {uu0 + xx0, uu0 - xx0, uu1 + xx1, uu1 - xx1};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int cov = (b[D4<B>::form[k][0]] + D4<B>::form[k][2] * e[D4<B>::form[k][1]]) >> 4;   // exact: 16 | 16*cov
                if (force || cov > li || cov < -li) {
                    const float er = exact_error(cov, -neg_rem, s64);
                    if (er < best_err) {                 // strict '<' in ascending candidate order (FC:627)
                        best_err = er;
                        best_cand = (uint32_t)d * 8u + (uint32_t)k;
                    }
                    const float lvl = (s32 == 0.0f) ? 0.0f : __fmul_rn(__fdiv_rn(fabsf((float)cov), s32), 0.99999618530273437500f);
                    tau = fmaxf(tau, lvl);
                }
            }
            if (neg_rem == 0 && best_cand != 0xFFFFFFFFu) tau = 8192.0f;   // rem == 0: error 0 for every block (FC:677)
        }
    }
    if (j < A.Nr && best_cand != 0xFFFFFFFFu)
        atomicMin(&A.key[(size_t)plane * A.Nr_pad + j], ((unsigned long long)f32_orderable(best_err) << 32) | best_cand);
}

// host-side launchers
int fic_d4_words(int B) { return B == 8 ? FIC_D4_B8_SLOTS / 2 : 0; }   // B = 16: 192 dwords per record do not fit the SGPR file

int fic_launch_d4_prep(const FicBuffers& b, uint32_t* rng_d4, uint32_t* pool_d4, const FicGeom& g, hipStream_t s)
{
    if (g.B != 8) return (int)hipErrorInvalidValue;    // the sweep is built for B = 8 (at B = 16 the record does not fit the SGPR file)
    dim3 gr((unsigned)g.tiles, g.planes), gd((unsigned)((g.Nd_pad + 63) / 64), g.planes);
    hipLaunchKernelGGL((k_range_d4<8>), gr, dim3(64), 0, s, (const uint8_t*)b.gray, rng_d4, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_pool_d4<8>), gd, dim3(64), 0, s, (const uint8_t*)b.pool_pix, pool_d4, g);
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_sweep_d4(const FicBuffers& b, const uint32_t* rng_d4, const uint32_t* pool_d4, const FicGeom& g, int tile0,
                        int ntiles, int chunk_len, int nchunks, hipStream_t s)
{
    if (g.n_iso != 8 || g.NR != 1 || !fic_d4_words(g.B)) return (int)hipErrorInvalidValue;
    D4Args A;
    A.pool_d4 = pool_d4; A.pool_st = b.pool_st; A.pool_s64 = b.pool_s64; A.rng_d4 = rng_d4; A.rng_st = b.rng_st; A.key = b.key;
    A.Nd = g.Nd; A.Nd_pad = g.Nd_pad; A.Nr = g.Nr; A.Nr_pad = g.Nr_pad; A.lgn = g.lgn; A.tiles = g.tiles;
    A.tile0 = tile0; A.ntiles = ntiles; A.chunk_len = chunk_len; A.nchunks = nchunks; A.planes = g.planes;
    dim3 grid((unsigned)(nchunks * g.planes) * (unsigned)((ntiles + 3) / 4)), block(256);
    hipLaunchKernelGGL((k_sweep_d4<8>), grid, block, 0, s, A);
    FIC_LAUNCH_CHECK();
    return 0;
}
