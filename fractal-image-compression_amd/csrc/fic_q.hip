// fic_q.hip -- the DEFAULT full-search sweep ("sweep" = 6): k_sweep_q<NK, MODE>, a matrix-core sweep whose MFMA output IS the
// prune test, plus its fused prep kernels k_pool_q (createCodebuch FC:1015-1050 + Domainblock DB:23-29,92-115 + operand
// fragments) and k_range_q (getRangeblock FC:588-602 + getMittelwert FC:67-73 + isometry copies + operand fragments).
// gfx950 (MI355X / CDNA4) only, wave64.  Compile with -ffp-contract=off (FractalCompression.java = FC, Domainblock.java = DB).
//
// Why another sweep.  k_sweep_bf16 (fic_bf16.hip) lets the matrix cores produce the exact covariance and then spends 28 VALU
// instructions per 32x32 tile turning it into the prune test |cov| * fl(1/sqrt(var)) > tau -- as much issue time as the four
// MFMAs of the tile, and the two do not overlap.  Here the DOMAIN operand is normalised before it meets the matrix core:
//     A[d][i] = f16( (d_i - dM) * fl(1/sqrt(var_d)) )      |A| <= 1, 11 significant bits
//     B[i][c] = f16( r_i - rM )  of range copy c             exact: integers in [-255, 255]
// so   acc = sum_i A*B  approximates  q = kovarianz / sqrt(var)  (FC:665-680 without the range constant `rem`) with
//     |acc - q| <= E_r = 2^-10.5 * ||r - rM||_2 (+2^-16)       (derivation below; E_r is a per-range constant)
// and the whole epilogue is  max|acc| > theta  (v_max3 tree + one compare: 9 VALU per tile, hidden beside the MFMAs).
// Pairs that pass are queued (4 bytes each) and evaluated EXACTLY later, 64 at a time, one per lane: integer covariance by
// v_dot4_u32_u8 from the u8 pixels, then the Java epilogue exact_error() (f64 divide, FC:677-683) and the lexicographic
// (error, candidate) atomicMin.  The codebook bits therefore come from the same arithmetic as in every other sweep; the
// matrix core only decides which pairs can be skipped, and it may only err on the side of evaluating too many.
//
// Invariant (restated for approximate values).  L(Y) = |kovarianz_Y| / sqrt(var_Y) in real arithmetic.
// theta of a range only ever takes values  fl(fl((m - E) * (1 - 2^-17)) - E)  where m = |acc_X| of a pair X that IS evaluated
// exactly; since L(X) >= m - E, an unflagged later pair Y (|acc_Y| <= theta) has  L(Y) <= |acc_Y| + E <= (1 - 2^-18) L(X):
// its |r| is strictly smaller than X's, its error not smaller, and -- coming later -- it cannot win a tie (FC:627 strict <).
// In general: a pair Y may be skipped whenever SOME pair X of the same range has (error_X, index_X) < (error_Y, index_Y)
// lexicographically -- then Y is not the minimum, whether or not X itself is ever evaluated (if X is skipped, it is not the
// minimum either); the true minimum is never skipped, and it is evaluated exactly.  X with a lower index and L(Y) <=
// (1 - 2^-18) L(X) is such a pair; so is X in any index order once its error is STRICTLY smaller (next paragraph).
// Chunk start.  The first domain tile of a pool chunk seeds theta from the tile's LARGEST |acc| (pair X*), which may have a
// higher index than pairs it prunes; that is sound when L(X*) >= 0.26 rem >= 0.25 (1 + 2^-20) rem (rem = the range block's
// `varianzRange`, FC:671: r = kovarianz / (rem sqrt(var)) = L / rem), because then |r_X*| >= 1/4 and 1 - r^2 of a pair with r^2
// smaller by 2^-17 relative is a strictly larger float, so the pruned pair's error is STRICTLY larger and the index order is
// irrelevant.  The bound is per range block (round 3; before: 0.26 n for all of them, n - 1 being the largest possible rem --
// on natural images the many smooth blocks, whose L is small in absolute terms, then never seeded and evaluated the whole first
// tile of every pool chunk).  Ranges whose first tile has no such pair evaluate that whole tile, as the older sweeps do for
// every range.  Candidate 0 is always evaluated (the global fallback winner, FC:613-632: every
// pair with r^2 rounding 1 - r^2 to 1, flat blocks included, ties with it at error = rem^2 and loses on index).
//
// The 8 isometries at half the matrix work ("folded" mode, B = 8 / 16).  The isometries come in four pairs {k, k'} that differ
// by the point reflection pos -> n-1-pos (0/2 = identity / rot180, 1/3 = rot90 / rot270, 4/5 = the mirrors, 6/7 = transpose /
// anti-transpose): copy_k'[pos] = copy_k[n-1-pos].  Pairing every position with its reflection, c = copy_k centred, x = the
// normalised domain block:
//     q_k  = sum_pos c x = Be + Bo,   q_k' = Be - Bo,   Be = sum_{pos<n/2} (c+c')(x+x')/2,   Bo = sum_{pos<n/2} (c-c')(x-x')/2
// so max(|q_k|, |q_k'|) = |Be| + |Bo|: per range block 4 "even" rows and 4 "odd" rows of K = n/2 instead of 8 rows of K = n --
// half the MFMAs for the same prune test (c+c', c-c' are integers in [-510, 510]: still exact f16; the error bound is the
// same: sum|A||D| <= 1/2 (||c+c'|| ||x+x'|| + ||c-c'|| ||x-x'||) <= ||r - rM||).  A flagged (range, pair of isometries,
// block) evaluates both isometries exactly.
//
// Error bound.  w = fl(1/fl32(sqrt(var))) and x_i = fl((d_i - dM) * w) carry < 2^-21 relative error; f16 rounding 2^-11
// relative (x_i is 0 or >= 1/2040 > 2^-14: never subnormal); products of an 8-bit and an 11-bit significand are exact in
// f32; the NK*16-term accumulation inside the MFMAs adds at most 2^-24 * 17 NK relative to sum |A||B| <= ||r-rM|| (NK <= 16:
// < 2^-15.9).  Total |acc - q| <= (2^-11 * 1.07) ||r - rM||_2; E_r uses 2^-10.5 (7.0e-4) plus 1.6e-5 absolute.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <type_traits>
#include "fic_device.h"
#include "fic_launch.h"
#include "fic_devfn.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define FIC_Q_LEVEL 0.99999237060546875f   // 1 - 2^-17
#define FIC_Q_TAU_ALL 3.0e38f              // "never flagged": above the test value of any pair (grey: L <= 16 * 255; joint RGB, where the
                                           //   operand is greyD / vD with vD as small as 1: ~1e8)
#define FIC_Q_TAU_NONE (-1.0f)             // nothing evaluated yet: every real pair of the tile is flagged
#define FIC_Q_LMIN 0.26f                   // a pair with L >= 0.26 rem has |r| >= 1/4: it prunes in any index order (header, "Chunk start")
#define FIC_Q_ECOEF 7.0e-4f                // >= 2^-10.5
#define FIC_Q_EABS 1.6e-5f
#define FIC_Q_QCAP 1280                    // queue entries per wave: one tile can flag at most 1024 pairs
#define FIC_Q_QFLUSH 256                   // evaluate the queue before a tile's pushes once it holds this many

#ifndef FIC_Q_CTW_B8
#define FIC_Q_CTW_B8 4
#endif
#ifndef FIC_Q_CTW_B4
#define FIC_Q_CTW_B4 8
#endif
#ifndef FIC_Q_WPG
#define FIC_Q_WPG 4                        // waves per workgroup of k_sweep_q (the waves are independent: no barrier, own queue)
#endif
#ifndef FIC_Q_WAVES_B8
#define FIC_Q_WAVES_B8 2                   // waves per SIMD the register budget is cut for (launch bounds)
#endif
#ifndef FIC_Q_WAVES_B4
#define FIC_Q_WAVES_B4 3
#endif
__host__ __device__ constexpr int fic_q_ctw(int NK) { return NK == 4 ? FIC_Q_CTW_B8 : (NK < 4 ? FIC_Q_CTW_B4 : 2); }   // column tiles (x32 range copies) per wave
#define FIC_Q_THG_NONE 0xBF800000u         // theta_g "nothing published": -1.0f; levels are positive floats, so signed-integer order = float order
#ifndef FIC_Q_GFAST_TILES
#define FIC_Q_GFAST_TILES 48               // pool chunks of at most this many domain tiles run k_sweep_qs / k_sweep_q16s (theta_g refreshed in the fast path; 0: never, A/B builds)
#endif
#ifndef FIC_Q_SEED
#define FIC_Q_SEED 1                       // pool chunks after the first take their first theta from domain tile 0 (0: A/B builds)
#endif
#define FIC_Q_UNROLL 2                     // domain tiles per iteration of the sweep loop (two fragment buffers swap roles)

#ifndef FIC_Q_SHAPE16
#define FIC_Q_SHAPE16 1                    // 1 isometry at B = 8 / 16: v_mfma_f32_16x16x32_f16 (k_sweep_q16) instead of 32x32x16
#endif
// Which 8 consecutive operand elements slot (s, lane) of a 32-row (or 32-column) fragment tile of NK slots holds.
//   shape 0, v_mfma_f32_32x32x16:  row = lane & 31,                       elements [16 s + 8 (lane >> 5), +8)
//   shape 1, v_mfma_f32_16x16x32:  [sub-tile s / KS][K step s % KS], KS = NK / 2:
//                                  row = 16 (s / KS) + (lane & 15),       elements [32 (s % KS) + 8 (lane >> 4), +8)
__device__ __forceinline__ void frag_slot(int shape16, int NK, int s, int lane, int& row, int& p0)
{
    if (!shape16) {
        row = lane & 31;
        p0 = 16 * s + 8 * (lane >> 5);
    } else {
        const int KS = NK >> 1;
        row = 16 * (s / KS) + (lane & 15);
        p0 = 32 * (s % KS) + 8 * (lane >> 4);
    }
}

// two floats -> packed f16 pair (round to nearest even), element 0 in the low half
__device__ __forceinline__ int f16_pair(float lo, float hi)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(int, v);
}

// ---------------------------------------------------------------------------------------------
// k_pool_q : one workgroup per domain tile (32 consecutive pool blocks).  createCodebuch FC:1015-1050 (block k = (c,r) at
// scaled (c*abstand, r*abstand), pixel order rx + ry*B), Domainblock statistics (DB:92-115: int mean, exact integer
// variance, Math.sqrt((double) var)), and the tile's A fragments for k_sweep_q.
//   poolQ[plane][dtile][m][lane] = 8 f16 = normalised pixels [16m + 8h, +8) of block 32*dtile + (lane&31), h = lane>>5.
//   dflat[plane][dtile] = 1 (u32) when all 32 blocks are flat or beyond N_d (their fragments are zero).
// ---------------------------------------------------------------------------------------------
// FUSE: the scaled image is not read but made on the way (scaleImage FC:970-1007, the body of k_scale: 2:1 box average of the
// grey image, 4th tap 128 when x + 1 >= HEIGHT, FC:993) -- small launches only, where k_scale is nothing but a launch; `scaled`
// is still written (every block row its own B pixels: the overlap makes that 16-fold redundant, which only a small launch can afford).
template <int B, bool FUSE>
__device__ __forceinline__ void pool_q_body(int dtile, int plane, const uint8_t* __restrict__ scaled_in, const uint8_t* __restrict__ gray,
                                            uint8_t* __restrict__ scaled_out, uint8_t* __restrict__ pool_pix,
                                            FicDomStat* __restrict__ pool_st, uint32_t* __restrict__ pool_var,
                                            double* __restrict__ pool_s64, v4i* __restrict__ poolQ,
                                            uint32_t* __restrict__ dflat, const FicGeom& g, int ndtiles_alloc, int folded, int shape16)
{
    __shared__ __attribute__((aligned(16))) uint8_t pix[32][256 + 16];
    __shared__ int s_sum[32][16], s_sq[32][16];
    __shared__ float s_w[32];
    __shared__ int s_mean[32];
    __shared__ int s_nonflat;
    constexpr int n = B * B, NK = n / 16;
    if (threadIdx.x == 0) s_nonflat = 0;
    // (a) rows of the tile's blocks: thread = (block i, row ry); a row is ONE load of B bytes (the address is only abstand-aligned:
    //     gfx950 global loads take that), byte sums by v_sad_u8, squares by v_dot4
    for (int t = threadIdx.x; t < 32 * B; t += 256) {
        const int i = t / B, ry = t % B;
        const int d = dtile * 32 + i;
        uint32_t w[B / 4];
#pragma unroll
        for (int q = 0; q < B / 4; q++) w[q] = 0u;
        if (d < g.Nd) {
            const int c = d % g.Dw, r = d / g.Dw;
            const size_t so = (size_t)plane * g.Ws * g.Hs + (size_t)(r * g.abstand + ry) * g.Ws + c * g.abstand;
            if constexpr (!FUSE) {
                __builtin_memcpy(w, scaled_in + so, B);
            } else {
                const int ys = r * g.abstand + ry, xs0 = c * g.abstand;
                const uint8_t* im = gray + (size_t)plane * g.W * g.H + (size_t)(2 * ys) * g.W + 2 * xs0;
                uint32_t top[B / 2], bot[B / 2];                         // 2B grey pixels of the two source rows
                __builtin_memcpy(top, im, 2 * B);
                __builtin_memcpy(bot, im + g.W, 2 * B);
#pragma unroll
                for (int x = 0; x < B; x++) {
                    const uint32_t t2 = top[x >> 1] >> (16 * (x & 1)), b2 = bot[x >> 1] >> (16 * (x & 1));
                    const int tap4 = (2 * (xs0 + x) + 1 >= g.H) ? 128 : (int)((b2 >> 8) & 0xffu);      // FC:993 compares with the HEIGHT
                    const int m = (int)(t2 & 0xffu) + (int)((t2 >> 8) & 0xffu) + (int)(b2 & 0xffu) + tap4;
                    w[x >> 2] |= (uint32_t)(m / 4) << (8 * (x & 3));
                }
                __builtin_memcpy(scaled_out + so, w, B);
            }
        }
        uint32_t sum = 0, sq = 0;
#pragma unroll
        for (int q = 0; q < B / 4; q++) {
            sum = __builtin_amdgcn_sad_u8(w[q], 0u, sum);
            sq = __builtin_amdgcn_udot4(w[q], w[q], sq, false);
            *(uint32_t*)&pix[i][ry * B + 4 * q] = w[q];
        }
        s_sum[i][ry] = (int)sum;
        s_sq[i][ry] = (int)sq;
    }
    __syncthreads();
    // the tile's 32 blocks are 32 n contiguous bytes of pool_pix: 16 bytes per lane, straight from LDS
    for (int t = threadIdx.x; t < 32 * n / 16; t += 256) {
        const int i = t / (n / 16), q = t % (n / 16);
        const int d = dtile * 32 + i;
        if (d < g.Nd) *(uint4*)(pool_pix + ((size_t)plane * g.Nd_pad + d) * n + 16 * q) = *(const uint4*)&pix[i][16 * q];
    }
    // (b) per-block statistics
    if (threadIdx.x < 32) {
        const int i = threadIdx.x, d = dtile * 32 + i;
        int S = 0, Q = 0;
        for (int t = 0; t < B; t++) { S += s_sum[i][t]; Q += s_sq[i][t]; }
        const int m = S >> g.lgn;                                  // mittelWert = S / n (DB:97)
        const int var = Q - 2 * m * S + n * m * m;                 // sum (d - m)^2, exact integer < 2^24 (DB:110-111)
        const double s64 = __dsqrt_rn((double)var);                // Math.sqrt((double) variance), FC:677,680
        const float s32 = (float)s64;
        float w = 0.0f;
        if (d < g.Nd) {
            const size_t o = (size_t)plane * g.Nd_pad + d;
            FicDomStat st;
            st.sum = (uint32_t)S;
            st.s32 = s32;
            pool_st[o] = st;
            pool_var[o] = (uint32_t)var;
            pool_s64[o] = s64;
            if (var != 0) { w = __fdiv_rn(1.0f, s32); atomicOr(&s_nonflat, 1); }
        }
        s_w[i] = w;
        s_mean[i] = m;
    }
    __syncthreads();
    // (c) fragments
    for (int t = threadIdx.x; t < NK * 64; t += 256) {
        const int lane = t & 63, m = t >> 6;
        int i, pos0;
        frag_slot(shape16, NK, m, lane, i, pos0);      // (the folded form below is always shape 0)
        const int h = lane >> 5;
        const float w = s_w[i];
        const int dM = s_mean[i];
        v4i v;
        // (8 consecutive pixels = one 8-byte LDS read; rows of pix are 272 bytes apart, positions are multiples of 8)
        if (!folded) {
            const uint2 pw = *(const uint2*)&pix[i][pos0];
            int px[8];
#pragma unroll
            for (int u = 0; u < 8; u++) px[u] = (int)(((u < 4 ? pw.x : pw.y) >> (8 * (u & 3))) & 0xffu);
#pragma unroll
            for (int u = 0; u < 4; u++)
                v[u] = f16_pair(__fmul_rn((float)(px[2 * u] - dM), w), __fmul_rn((float)(px[2 * u + 1] - dM), w));
        } else {
            // steps m < NK/2: even part (x + x')/2 of positions [16m + 8h, +8), x' at n-1-pos; steps m >= NK/2: odd part (x - x')/2
            const bool odd = m >= NK / 2;
            const int p0 = 16 * (odd ? m - NK / 2 : m) + 8 * h;
            const uint2 pw = *(const uint2*)&pix[i][p0], qw = *(const uint2*)&pix[i][n - 8 - p0];   // qw: positions n-8-p0 .. n-1-p0
            float f[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int a = (int)(((u < 4 ? pw.x : pw.y) >> (8 * (u & 3))) & 0xffu);
                const int b = (int)(((u < 4 ? qw.y : qw.x) >> (8 * (3 - (u & 3)))) & 0xffu);           // position n-1-(p0+u)
                f[u] = __fmul_rn(__fmul_rn((float)(odd ? a - b : a + b - 2 * dM), w), 0.5f);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = f16_pair(f[2 * u], f[2 * u + 1]);
        }
        poolQ[((size_t)plane * ndtiles_alloc + dtile) * NK * 64 + t] = v;
    }
    if (threadIdx.x == 0) dflat[(size_t)plane * ndtiles_alloc + dtile] = s_nonflat ? 0u : 1u;
}
template <int B>
__global__ __launch_bounds__(256) void k_pool_q(const uint8_t* __restrict__ scaled, uint8_t* __restrict__ pool_pix,
                                                FicDomStat* __restrict__ pool_st, uint32_t* __restrict__ pool_var,
                                                double* __restrict__ pool_s64, v4i* __restrict__ poolQ,
                                                uint32_t* __restrict__ dflat, FicGeom g, int ndtiles_alloc, int folded, int shape16)
{
    pool_q_body<B, false>(blockIdx.x, blockIdx.y, scaled, nullptr, nullptr, pool_pix, pool_st, pool_var, pool_s64, poolQ, dflat, g,
                          ndtiles_alloc, folded, shape16);
}

// ---------------------------------------------------------------------------------------------
// k_range_q : one workgroup per 64 range blocks.  getRangeblock FC:588-602 (R-channel bytes, row-major), getMittelwert
// FC:67-73 (rM = sum / n, int), rem = sum - n*rM (= varianzRange after the loop at FC:665-672), the error bound E_r, the
// reset of the search key and of the published theta, and the B fragments for k_sweep_q.
//   copy_k[pos] = r[iso_source(iso_inverse(k), pos)]  so that  dot(copy_k, d) == dot(r, iso_k(d))   
//   (also stored as plain bytes, rngC, for the exact evaluation of the pairs the sweep flags)
//   column rr of the sweep: mode 0 (1 isometry): range rr; mode 1 (8 isometries, direct): range rr >> 3, copy rr & 7;
//   mode 2 (8 isometries, folded): range rr >> 2, isometry pair rr & 3 = {0,2}, {1,3}, {4,5}, {6,7}.
//   rngQ[plane][ctile][m][lane] = 8 f16 of column 32*ctile + (lane&31), h = lane>>5: centred pixels [16m + 8h, +8) of the
//   copy; folded: steps m < NK/2 the even part c + c' of positions [16m + 8h, +8), steps m >= NK/2 the odd part c - c'.
// Statistics: 4 threads per block (byte sums by v_sad_u8, squares by v_dot4), combined inside the quad.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_range_q(const uint8_t* __restrict__ gray, FicRngStat* __restrict__ rng_st,
                                                 float* __restrict__ rngE, unsigned long long* __restrict__ key,
                                                 uint32_t* __restrict__ theta_g, v4i* __restrict__ rngQ,
                                                 uint32_t* __restrict__ rngC, FicGeom g, int nct_alloc, int grp0, int mode,
                                                 int staged, int shape16)
{
    __shared__ __attribute__((aligned(16))) uint8_t blk[64 * (256 + 4)];
    extern __shared__ __attribute__((aligned(16))) uint8_t cpy[];   // staged: [columns][n + 8] the copies as bytes
    __shared__ int s_rM[64];
    const int plane = blockIdx.y;
    const int grp = grp0 + blockIdx.x;
    const int B = g.B, n = g.n, DW = g.DW, NK = n / 16;
    const int lgB = B == 4 ? 2 : (B == 8 ? 3 : 4);
    const int stride = n + 4;
    const int j0 = grp * 64;
    const uint8_t* img = gray + (size_t)plane * g.W * g.H;
    for (int i = threadIdx.x; i < 64 * DW; i += 256) {         // DW words per block
        const int l = i / DW, wd = i % DW;
        const int j = j0 + l;
        uint32_t v = 0;
        if (j < g.Nr) {
            const int pos = wd * 4;
            const uint8_t* p = img + (size_t)((j / g.Rw) * B + (pos >> lgB)) * g.W + (j % g.Rw) * B + (pos & (B - 1));
            v = *(const uint32_t*)p;                           // B, W multiples of 4: aligned
        }
        *(uint32_t*)&blk[l * stride + wd * 4] = v;
    }
    __syncthreads();
    {
        const int l = threadIdx.x >> 2, part = threadIdx.x & 3;   // 4 threads per block, DW/4 words each
        const uint32_t* w = (const uint32_t*)(blk + l * stride) + part * (DW / 4);
        uint32_t S = 0, Q = 0;
        for (int i = 0; i < DW / 4; i++) {
            S = __builtin_amdgcn_sad_u8(w[i], 0u, S);
            Q = __builtin_amdgcn_udot4(w[i], w[i], Q, false);
        }
        S += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S, 0xB1, 0xF, 0xF, true);   // quad: lanes 1,0,3,2
        Q += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Q, 0xB1, 0xF, 0xF, true);
        S += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S, 0x4E, 0xF, 0xF, true);   // quad: lanes 2,3,0,1
        Q += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Q, 0x4E, 0xF, 0xF, true);
        const int j = j0 + l;
        const int rM = (int)(S >> g.lgn);
        if (part == 0) {
            s_rM[l] = rM;
            if (j < g.Nr_pad) {
                const int ss = (int)Q - 2 * rM * (int)S + n * rM * rM;                    // sum (r - rM)^2, exact integer
                const size_t o = (size_t)plane * g.Nr_pad + j;
                FicRngStat st;
                st.rM = j < g.Nr ? rM : 0;
                st.rem = j < g.Nr ? (int)S - (rM << g.lgn) : 0;
                rng_st[o] = st;
                rngE[o] = __fadd_rn(__fmul_rn(__fsqrt_rn((float)ss), FIC_Q_ECOEF), FIC_Q_EABS);
                key[o] = FIC_KEY_NONE;
                theta_g[o] = FIC_Q_THG_NONE;
            }
        }
    }
    __syncthreads();
    const int cpr = mode == 0 ? 1 : (mode == 1 ? 8 : 4);       // columns per range block
    const int lgc = mode == 0 ? 0 : (mode == 1 ? 3 : 2);
    const int cols = 64 * cpr;                                 // columns of this workgroup
    const int ct0 = (j0 * cpr) / 32;
    // The columns' isometry copies as bytes, gathered ONCE through the affine form of iso_source: to global memory (rngC, the
    // operand of the exact evaluation of flagged pairs, q_flush) and -- when they fit (`staged`) -- to LDS, from where the
    // fragments below read 8 consecutive bytes instead of gathering every pixel a second time.
    const int cstride = n + 8;
    for (int t = threadIdx.x; t < cols * DW; t += 256) {
        const int col = t / DW, wd = t % DW;
        const int l = col >> lgc, c = col & (cpr - 1);
        if (j0 + l >= g.Nr_pad) break;
        const int k = mode == 2 ? (c == 0 ? 0 : (c == 1 ? 1 : (c == 2 ? 4 : 6))) : c;   // folded: the pair's first isometry
        int ax, bx, cx, ay, by, cy;
        iso_affine(iso_inverse(k), B - 1, ax, bx, cx, ay, by, cy);
        const int s0 = cx + cy * B, sx = ax + ay * B, sy = bx + by * B;                 // source index = s0 + sx*x + sy*y
        const uint8_t* b = blk + l * stride;
        uint32_t w = 0;
        if (k == 0) {
            w = *(const uint32_t*)(b + 4 * wd);                // the identity copy: the block's own dword (stride is a multiple of 4)
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int p = 4 * wd + u;
                w |= (uint32_t)b[s0 + sx * (p & (B - 1)) + sy * (p >> lgB)] << (8 * u);
            }
        }
        rngC[((size_t)plane * g.Nr_pad * cpr + (size_t)j0 * cpr + col) * DW + wd] = w;
        if (staged) *(uint32_t*)(cpy + col * cstride + 4 * wd) = w;
    }
    if (staged) __syncthreads();
    // fragments: thread = (column tile, m, lane)
    for (int t = threadIdx.x; t < (cols / 32) * NK * 64; t += 256) {
        const int lane = t & 63, m = (t >> 6) % NK, ctl = (t >> 6) / NK;
        int c32, pos0;
        frag_slot(shape16, NK, m, lane, c32, pos0);    // (shape 1 only with mode 0)
        const int col = ctl * 32 + c32, h = lane >> 5;
        const int l = col >> lgc, c = col & (cpr - 1);
        const int rM = s_rM[l];
        const bool real = j0 + l < g.Nr;
        int a[8];
        if (staged) {
            const uint8_t* cp = cpy + col * cstride;
            if (mode != 2) {
                const uint2 w = *(const uint2*)(cp + pos0);
#pragma unroll
                for (int u = 0; u < 8; u++) a[u] = (int)(((u < 4 ? w.x : w.y) >> (8 * (u & 3))) & 0xffu) - rM;
            } else {
                const bool odd = m >= NK / 2;
                const int q0 = 16 * (odd ? m - NK / 2 : m) + 8 * h;
                const uint2 w = *(const uint2*)(cp + q0), v = *(const uint2*)(cp + n - 8 - q0);   // v: positions n-8-q0 .. n-1-q0
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int c0 = (int)(((u < 4 ? w.x : w.y) >> (8 * (u & 3))) & 0xffu);
                    const int c1 = (int)(((u < 4 ? v.y : v.x) >> (8 * (3 - (u & 3)))) & 0xffu);    // position n-1-(q0+u)
                    a[u] = odd ? c0 - c1 : c0 + c1 - 2 * rM;
                }
            }
        } else {
            const int k = mode == 2 ? (c == 0 ? 0 : (c == 1 ? 1 : (c == 2 ? 4 : 6))) : c;
            int ax, bx, cx, ay, by, cy;
            iso_affine(iso_inverse(k), B - 1, ax, bx, cx, ay, by, cy);
            const int s0 = cx + cy * B, sx = ax + ay * B, sy = bx + by * B;
            const uint8_t* b = blk + l * stride;
            if (mode != 2) {
                const int p0 = pos0;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int p = p0 + u;
                    a[u] = (int)b[s0 + sx * (p & (B - 1)) + sy * (p >> lgB)] - rM;
                }
            } else {
                const bool odd = m >= NK / 2;
                const int q0 = 16 * (odd ? m - NK / 2 : m) + 8 * h;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int p = q0 + u, pr = n - 1 - p;
                    const int c0 = b[s0 + sx * (p & (B - 1)) + sy * (p >> lgB)], c1 = b[s0 + sx * (pr & (B - 1)) + sy * (pr >> lgB)];
                    a[u] = odd ? c0 - c1 : c0 + c1 - 2 * rM;
                }
            }
        }
        v4i v;
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = real ? f16_pair((float)a[2 * u], (float)a[2 * u + 1]) : 0;
        if (ct0 + ctl < nct_alloc) rngQ[((size_t)plane * nct_alloc + ct0 + ctl) * NK * 64 + (size_t)m * 64 + lane] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// k_range_q8<MODE> : k_range_q for B = 8 (MODE 0: 1 isometry, MODE 2: 8 isometries folded) -- the two shapes the benchmark
// spends its range prep on.  Same outputs, bit for bit; what differs is how the bytes move:
//   * range rows enter LDS as 8-byte loads; statistics as in k_range_q;
//   * the four copies of a block (the first isometries of the folded pairs: identity, rot 90, mirror, transpose -- gathered
//     through the INVERSE map) are made in registers, one wave per copy: a block is 16 dwords, a mirror is
//     a byte swap per dword, an 8x8 byte transpose is 32 v_perm_b32 (four 4x4 tiles, two stages each) -- instead of four
//     ds_read_u8 + shifts per output dword;
//   * rngC leaves as 16 bytes per lane, contiguous over the workgroup (it is 64 B per column, columns consecutive).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void t4x4_bytes(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
    // rows a, b, c, d of a 4x4 byte tile -> its columns (v_perm_b32: selector bytes 0-3 pick from the second source, 4-7 from the first)
    const uint32_t t0 = __builtin_amdgcn_perm(b, a, 0x06020400u), t1 = __builtin_amdgcn_perm(b, a, 0x07030501u);
    const uint32_t t2 = __builtin_amdgcn_perm(d, c, 0x06020400u), t3 = __builtin_amdgcn_perm(d, c, 0x07030501u);
    o0 = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    o1 = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    o2 = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    o3 = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}
// out[y][x] = in[x][y] of an 8x8 byte block held as 16 dwords (row y = dwords 2y, 2y + 1)
__device__ __forceinline__ void b8_transpose(const uint32_t (&r)[16], uint32_t (&o)[16])
{
#pragma unroll
    for (int ty = 0; ty < 2; ty++)
#pragma unroll
        for (int tx = 0; tx < 2; tx++)
            t4x4_bytes(r[2 * (4 * ty + 0) + tx], r[2 * (4 * ty + 1) + tx], r[2 * (4 * ty + 2) + tx], r[2 * (4 * ty + 3) + tx],
                       o[2 * (4 * tx + 0) + ty], o[2 * (4 * tx + 1) + ty], o[2 * (4 * tx + 2) + ty], o[2 * (4 * tx + 3) + ty]);
}
// out[y][x] = in[y][7 - x]
__device__ __forceinline__ void b8_mirror(const uint32_t (&r)[16], uint32_t (&o)[16])
{
#pragma unroll
    for (int y = 0; y < 8; y++) {
        o[2 * y] = __builtin_bswap32(r[2 * y + 1]);
        o[2 * y + 1] = __builtin_bswap32(r[2 * y]);
    }
}
template <int MODE>
__device__ __forceinline__ void range_q8_body(int grp, int plane, const uint8_t* __restrict__ gray, FicRngStat* __restrict__ rng_st,
                                              float* __restrict__ rngE, unsigned long long* __restrict__ key,
                                              uint32_t* __restrict__ theta_g, v4i* __restrict__ rngQ,
                                              uint32_t* __restrict__ rngC, const FicGeom& g, int nct_alloc, int shape16)
{
    constexpr int B = 8, n = 64, DW = 16, NK = 4, CPR = MODE == 0 ? 1 : 4, STR = 80;     // STR: LDS bytes per block / copy (16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) uint8_t blk[64 * STR];
    __shared__ __attribute__((aligned(16))) uint8_t cpy[MODE == 0 ? 16 : 256 * STR];
    __shared__ int s_rM[64];
    const int j0 = grp * 64;
    const uint8_t* img = gray + (size_t)plane * g.W * g.H;
    for (int i = threadIdx.x; i < 64 * B; i += 256) {          // one 8-byte row per thread and pass
        const int l = i >> 3, ry = i & 7;
        const int j = j0 + l;
        uint2 v = make_uint2(0u, 0u);
        if (j < g.Nr) v = *(const uint2*)(img + (size_t)((j / g.Rw) * B + ry) * g.W + (j % g.Rw) * B);   // W, B multiples of 8: aligned
        *(uint2*)&blk[l * STR + ry * B] = v;
    }
    __syncthreads();
    {
        const int l = threadIdx.x >> 2, part = threadIdx.x & 3;   // 4 threads per block, 4 words each
        const uint4 w4 = *(const uint4*)(blk + l * STR + part * 16);
        const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
        uint32_t S = 0, Q = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            S = __builtin_amdgcn_sad_u8(w[i], 0u, S);
            Q = __builtin_amdgcn_udot4(w[i], w[i], Q, false);
        }
        S += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S, 0xB1, 0xF, 0xF, true);   // quad: lanes 1,0,3,2
        Q += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Q, 0xB1, 0xF, 0xF, true);
        S += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S, 0x4E, 0xF, 0xF, true);   // quad: lanes 2,3,0,1
        Q += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Q, 0x4E, 0xF, 0xF, true);
        const int j = j0 + l;
        const int rM = (int)(S >> 6);
        if (part == 0) {
            s_rM[l] = rM;
            if (j < g.Nr_pad) {
                const int ss = (int)Q - 2 * rM * (int)S + n * rM * rM;                    // sum (r - rM)^2, exact integer
                const size_t o = (size_t)plane * g.Nr_pad + j;
                FicRngStat st;
                st.rM = j < g.Nr ? rM : 0;
                st.rem = j < g.Nr ? (int)S - (rM << 6) : 0;
                rng_st[o] = st;
                rngE[o] = __fadd_rn(__fmul_rn(__fsqrt_rn((float)ss), FIC_Q_ECOEF), FIC_Q_EABS);
                key[o] = FIC_KEY_NONE;
                theta_g[o] = FIC_Q_THG_NONE;
            }
        }
    }
    if constexpr (MODE == 2) {
        // wave c makes copy c of all 64 blocks: c = 0 identity, 1 = isometry 1 (gathered through its inverse, rot 270:
        // transpose of the mirror), 2 = isometry 4 (mirror left-right), 3 = isometry 6 (transpose)
        const int c = threadIdx.x >> 6, l = threadIdx.x & 63;
        uint32_t r[16], o[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = *(const uint4*)(blk + l * STR + 16 * q);
            r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
        }
        if (c == 0) {
#pragma unroll
            for (int q = 0; q < 16; q++) o[q] = r[q];
        } else if (c == 1) {
            uint32_t m[16];
            b8_mirror(r, m);
            b8_transpose(m, o);
        } else if (c == 2) {
            b8_mirror(r, o);
        } else {
            b8_transpose(r, o);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) *(uint4*)(cpy + (l * 4 + c) * STR + 16 * q) = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
    __syncthreads();
    const uint8_t* const cps = MODE == 0 ? blk : cpy;           // the columns' copies as bytes, STR apart
    constexpr int cols = 64 * CPR;
    const int ct0 = (j0 * CPR) / 32;
    // rngC: cols x 64 bytes, contiguous: 16 bytes per lane
    for (int t = threadIdx.x; t < cols * 4; t += 256) {
        const int col = t >> 2, q = t & 3;
        if (j0 + col / CPR < g.Nr_pad)
            *(uint4*)(rngC + ((size_t)plane * g.Nr_pad * CPR + (size_t)j0 * CPR + col) * DW + 4 * q) = *(const uint4*)(cps + col * STR + 16 * q);
    }
    // fragments: thread = (column tile, m, lane), exactly as the staged branch of k_range_q
    for (int t = threadIdx.x; t < (cols / 32) * NK * 64; t += 256) {
        const int lane = t & 63, m = (t >> 6) % NK, ctl = (t >> 6) / NK;
        int c32, pos0;
        frag_slot(MODE == 0 ? shape16 : 0, NK, m, lane, c32, pos0);
        const int col = ctl * 32 + c32, h = lane >> 5;
        const int l = col / CPR;
        const int rM = s_rM[l];
        const bool real = j0 + l < g.Nr;
        const uint8_t* cp = cps + col * STR;
        int a[8];
        if constexpr (MODE != 2) {
            const uint2 w = *(const uint2*)(cp + pos0);
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] = (int)(((u < 4 ? w.x : w.y) >> (8 * (u & 3))) & 0xffu) - rM;
        } else {
            const bool odd = m >= NK / 2;
            const int q0 = 16 * (odd ? m - NK / 2 : m) + 8 * h;
            const uint2 w = *(const uint2*)(cp + q0), v = *(const uint2*)(cp + n - 8 - q0);   // v: positions n-8-q0 .. n-1-q0
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int c0 = (int)(((u < 4 ? w.x : w.y) >> (8 * (u & 3))) & 0xffu);
                const int c1 = (int)(((u < 4 ? v.y : v.x) >> (8 * (3 - (u & 3)))) & 0xffu);    // position n-1-(q0+u)
                a[u] = odd ? c0 - c1 : c0 + c1 - 2 * rM;
            }
        }
        v4i v;
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = real ? f16_pair((float)a[2 * u], (float)a[2 * u + 1]) : 0;
        if (ct0 + ctl < nct_alloc) rngQ[((size_t)plane * nct_alloc + ct0 + ctl) * NK * 64 + (size_t)m * 64 + lane] = v;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_range_q8(const uint8_t* __restrict__ gray, FicRngStat* __restrict__ rng_st,
                                                  float* __restrict__ rngE, unsigned long long* __restrict__ key,
                                                  uint32_t* __restrict__ theta_g, v4i* __restrict__ rngQ,
                                                  uint32_t* __restrict__ rngC, FicGeom g, int nct_alloc, int grp0, int shape16)
{
    range_q8_body<MODE>(grp0 + blockIdx.x, blockIdx.y, gray, rng_st, rngE, key, theta_g, rngQ, rngC, g, nct_alloc, shape16);
}
// k_prep_q8<MODE> : scale + pool build + range prep of a SMALL launch (one 512x512 image: 493 + 64 workgroups) as ONE launch -- the
// three kernels are 5-8 us each there and all of it is launch ramp (profiles/r03H_single8_kernel_stats.csv).  Workgroups
// [0, npool) are domain tiles (pool_q_body with the 2:1 average made on the way), the rest are groups of 64 range blocks.
template <int MODE>
__global__ __launch_bounds__(256) void k_prep_q8(const uint8_t* __restrict__ gray, uint8_t* __restrict__ scaled, uint8_t* __restrict__ pool_pix,
                                                 FicDomStat* __restrict__ pool_st, uint32_t* __restrict__ pool_var,
                                                 double* __restrict__ pool_s64, v4i* __restrict__ poolQ, uint32_t* __restrict__ dflat,
                                                 FicRngStat* __restrict__ rng_st, float* __restrict__ rngE,
                                                 unsigned long long* __restrict__ key, uint32_t* __restrict__ theta_g,
                                                 v4i* __restrict__ rngQ, uint32_t* __restrict__ rngC, FicGeom g, int ndtiles_alloc,
                                                 int nct_alloc, int grp0, int shape16)
{
    if ((int)blockIdx.x < ndtiles_alloc)
        pool_q_body<8, true>(blockIdx.x, blockIdx.y, nullptr, gray, scaled, pool_pix, pool_st, pool_var, pool_s64, poolQ, dflat, g, ndtiles_alloc,
                             MODE == 2 ? 1 : 0, shape16);
    else
        range_q8_body<MODE>(grp0 + (int)blockIdx.x - ndtiles_alloc, blockIdx.y, gray, rng_st, rngE, key, theta_g, rngQ, rngC, g, nct_alloc, shape16);
}

// ---------------------------------------------------------------------------------------------
// k_sweep_q<NK, MODE>      MODE 0: 1 isometry (the reference algorithm); 1: 8 isometries, one column per copy (B = 4);
//                          2: 8 isometries folded by the point reflection (B = 8 / 16; see the header); 3: joint RGB (end of file)
//   rows (A) = 32 consecutive domain blocks, streamed from L2/HBM (two VGPR buffers; a buffer is reloaded as soon as its
//              last MFMAs have issued);
//   cols (B) = 32 range columns (a copy, or in MODE 2 an isometry pair); a wave keeps the fragments of its CTW column tiles
//              in VGPRs for the whole sweep, a workgroup = 4 waves = 4*CTW column tiles.
//   acc[e] of lane (col = lane&31, half = lane>>5) belongs to (column col, domain block 32*dt + (e&3) + 8(e>>2) + 4*half);
//   MODE 2 has two accumulators per tile (even / odd part, NK/2 MFMAs each) and tests |even| + |odd|.
//   theta of the lane's range sits in one VGPR per column tile; the lanes that share a range (two halves x the range's
//   columns) exchange raised values by DPP quad permutes + v_permlane32_swap when something is flagged (no LDS).
//   The next tile's MFMAs are issued interleaved with the current tile's epilogue (sched_group_barrier).
// ---------------------------------------------------------------------------------------------
struct QArgs {
    const v4i* poolQ;                // [plane][ndtiles_alloc][NK][64]
    const uint32_t* dflat;           // [plane][ndtiles_alloc]
    const uint8_t* pool_pix;         // [plane][Nd_pad][n]
    const FicDomStat* pool_st;
    const double* pool_s64;
    const v4i* rngQ;                 // [plane][nct_alloc][NK][64]
    const uint32_t* rngC;            // [plane][Nr_pad * columns per range][n] the columns' isometry copies as bytes (exact path)
    const FicRngStat* rng_st;
    const float* rngE;
    unsigned long long* key;
    uint32_t* theta_g;               // [plane][Nr_pad] best published level per range (order-preserving integer image; 0 = none)
    const int16_t* rgb_rng;          // MODE 3 (joint RGB): greyR_i [N_r][n], R+G+B per pool pixel [N_d][n], pool statistics
    const uint16_t* rgb_pool;
    const FicRgbDomStat* rgb_dst;
    unsigned long long* stats;       // optional [8]: tile epilogues, tiles with flagged pairs, queued entries, waves; of every
                                     //   64th wave: shader-clock cycles and 100 MHz ticks it was alive (summed), their number
    int Nd, Nd_pad, Nr, Nr_pad, n, lgn, W, H, Rw;
    int ndtiles, ndtiles_loop, ndtiles_alloc, nct_alloc;   // ndtiles_loop: ndtiles rounded up to the loop's unroll factor
    int ct_begin, ct_end;            // column tiles (x32 columns) of this shard
    int nctg;                        // column-tile groups (workgroups) in this launch
    int tiles_per_chunk, nchunks, planes;
    // fused finalise (small launches; fin_count == nullptr: k_finalize runs as its own launch)
    unsigned int* fin_count;         // [planes][nctg] workgroups of a (plane, column group) that have finished; the last one finalises
    FicOutputs out;
    FicGeom geom;
    const uint8_t* gray;
    const uint32_t* pool_var;
    int r_begin, r_end;              // the range blocks this launch encodes
    int dbg_noflag;                  // diagnostic (option "q_noflag"): theta = "never" for every range -- the sweep's floor without any flagged tile (WRONG codebooks)
};

__device__ __forceinline__ v16f mfma_f16(v4i a, v4i b, v16f c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// Every node is a three-input maximum (v_max3_f32 with |.| source modifiers): a two-input fmaxf of MFMA results costs two
// extra canonicalising v_max_f32 x, x under IEEE mode, hence the constant 0 as third operand at the root.
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float max16_abs(const v16f& a)
{
    const float m0 = max3f(fabsf(a[0]), fabsf(a[1]), fabsf(a[2]));
    const float m1 = max3f(fabsf(a[3]), fabsf(a[4]), fabsf(a[5]));
    const float m2 = max3f(fabsf(a[6]), fabsf(a[7]), fabsf(a[8]));
    const float m3 = max3f(fabsf(a[9]), fabsf(a[10]), fabsf(a[11]));
    const float m4 = max3f(fabsf(a[12]), fabsf(a[13]), fabsf(a[14]));
    return max3f(max3f(m0, m1, m2), max3f(m3, m4, fabsf(a[15])), 0.0f);
}
// MODE 2: |even| + |odd| per element, then the maximum
__device__ __forceinline__ float max16_sum(const v16f& a, const v16f& b)
{
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = fabsf(a[e]) + fabsf(b[e]);
    const float m0 = fmaxf(fmaxf(v[0], v[1]), v[2]), m1 = fmaxf(fmaxf(v[3], v[4]), v[5]), m2 = fmaxf(fmaxf(v[6], v[7]), v[8]);
    const float m3 = fmaxf(fmaxf(v[9], v[10]), v[11]), m4 = fmaxf(fmaxf(v[12], v[13]), v[14]);
    return fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), v[15]));
}

template <int MODE> struct QMode {
    static constexpr int CSHIFT = (MODE == 0 || MODE == 3) ? 0 : (MODE == 1 ? 3 : 2);   // log2(columns per range block)
    static constexpr int NISO = (MODE == 0 || MODE == 3) ? 1 : 8;
};
// first isometry of column sub-index c (MODE 2: of the pair {0,2}, {1,3}, {4,5}, {6,7})
template <int MODE> __device__ __forceinline__ int q_col_iso(int c) { return MODE == 2 ? (c == 0 ? 0 : (c == 1 ? 1 : (c == 2 ? 4 : 6))) : c; }

// Lexicographic (error, candidate) minimum into a range's search key.  The 8-byte atomic executes at the memory side, where
// updates of ONE address serialise (~12 ns each, chip-wide): on natural images a smooth range block has thousands of near-equal
// candidates, every one evaluated exactly, nearly none an improvement -- so the key is read first (a device-scope load: past
// the L1, from L2; a stale value is an OLDER one, i.e. larger: the atomic is then issued needlessly, never skipped wrongly)
// and the atomic issued only for a candidate that beats it (q_flush's grey path does the same with the read hoisted above its gathers).
__device__ __forceinline__ void q_key_min(unsigned long long* keyp, unsigned long long best)
{
    if (best < __hip_atomic_load(keyp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(keyp, best);
}

// Exact evaluation of the queued entries, one per lane: kovarianz = sum r*d - rM*sum(d) - dM*rem (exact integers), then
// getErrorVarianceCovariance FC:674-683 and the strict-'<' scan of FC:619-632 as a lexicographic atomicMin.  A MODE 2 entry
// is an isometry pair: both copies are evaluated against the one domain block.
// (Inlined: as a real call it costs the kernel 50 more VGPRs for the calling convention, i.e. a wave per SIMD.)
template <int NK, int MODE>
__device__ __forceinline__ void q_flush(const QArgs& A, const uint32_t* myq, int qn, int plane, int ctw0, int lane)
{
    constexpr int NISO = QMode<MODE>::NISO, CSHIFT = QMode<MODE>::CSHIFT;
    constexpr int B = NK == 1 ? 4 : (NK == 4 ? 8 : 16);
    const FicRngStat* rst = A.rng_st + (size_t)plane * A.Nr_pad;
    const double* p64 = A.pool_s64 + (size_t)plane * A.Nd_pad;
    const FicDomStat* pst = A.pool_st + (size_t)plane * A.Nd_pad;
    unsigned long long* keyp = A.key + (size_t)plane * A.Nr_pad;
    const uint32_t* cop = A.rngC + (size_t)plane * A.Nr_pad * (1 << CSHIFT) * (A.n / 4);
    for (int base = 0; base < qn; base += 64) {
        const int i = base + lane;
        if (i < qn) {
            const uint32_t ent = myq[i];
            const uint32_t d = ent & 0x00FFFFFFu;
            if (d >= (uint32_t)A.Nd) continue;                 // a zero row behind the pool, flagged while theta was "none"
            const int col = ctw0 * 32 + (int)(ent >> 24);
            if constexpr (MODE == 3) {
                // getErrorVarianceCovarianceRGB FC:781-803: kovarianz accumulated in f32 in the order i = 0..n-1 (the sums exceed
                // 2^24: the order is part of the result), r in f32, error = varianzRange^2 * (1 - r^2); rst[].rem = varianzRange
                const FicRngStat rs = rst[col];
                const FicRgbDomStat ds = A.rgb_dst[d];
                const uint4* rt = (const uint4*)(A.rgb_rng + (size_t)col * (B * B));
                const uint4* ps = (const uint4*)(A.rgb_pool + (size_t)d * (B * B));
                float kov = 0.0f;
#pragma unroll 4
                for (int q = 0; q < B * B / 8; q++) {
                    const uint4 r4 = rt[q], p4 = ps[q];
                    const uint32_t rw[4] = {r4.x, r4.y, r4.z, r4.w}, pw[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        kov = __fadd_rn(kov, (float)((int)(int16_t)(rw[u] & 0xffffu) * ((int)(pw[u] & 0xffffu) - ds.msum)));
                        kov = __fadd_rn(kov, (float)((int)(int16_t)(rw[u] >> 16) * ((int)(pw[u] >> 16) - ds.msum)));
                    }
                }
                const float vRf = (float)rs.rem;
                float r = (rs.rem == 0 || ds.vD == 0) ? 0.0f : __fdiv_rn(kov, __fmul_rn(vRf, (float)ds.vD));
                r = __fmul_rn(r, r);
                const float e = __fmul_rn(__fmul_rn(vRf, vRf), __fsub_rn(1.0f, r));
                q_key_min(&keyp[col], ((unsigned long long)f32_orderable(e) << 32) | d);
                continue;
            }
            const int j = col >> CSHIFT, k = q_col_iso<MODE>(col & ((1 << CSHIFT) - 1));
            const int k2 = k == 0 ? 2 : (k == 1 ? 3 : k + 1);                       // MODE 2: the pair's second isometry
            const unsigned long long cur = __hip_atomic_load(&keyp[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // q_key_min's read, issued with the gathers
            const FicRngStat rs = rst[j];
            const int Sd = (int)pst[d].sum;
            const double s64 = p64[d];
            // sum copy_k[pos] d[pos]; the partner copy is the point reflection: sum copy_k[n-1-pos] d[pos] = sum_q copy_k[q]
            // d[n-1-q], i.e. the mirrored dword of the domain block with its bytes reversed
            const uint4* cw = (const uint4*)(cop + (size_t)col * (B * B / 4));
            const uint4* dw = (const uint4*)(A.pool_pix + ((size_t)plane * A.Nd_pad + d) * A.n);
            uint32_t s = 0, s2 = 0;
#pragma unroll
            for (int q = 0; q < B * B / 16; q++) {
                const uint4 c4 = cw[q], d4 = dw[q];
                s = __builtin_amdgcn_udot4(c4.x, d4.x, s, false);
                s = __builtin_amdgcn_udot4(c4.y, d4.y, s, false);
                s = __builtin_amdgcn_udot4(c4.z, d4.z, s, false);
                s = __builtin_amdgcn_udot4(c4.w, d4.w, s, false);
                if constexpr (MODE == 2) {
                    const uint4 m4 = dw[B * B / 16 - 1 - q];
                    s2 = __builtin_amdgcn_udot4(c4.x, __builtin_bswap32(m4.w), s2, false);
                    s2 = __builtin_amdgcn_udot4(c4.y, __builtin_bswap32(m4.z), s2, false);
                    s2 = __builtin_amdgcn_udot4(c4.z, __builtin_bswap32(m4.y), s2, false);
                    s2 = __builtin_amdgcn_udot4(c4.w, __builtin_bswap32(m4.x), s2, false);
                }
            }
            const int base_c = rs.rM * Sd + (Sd >> A.lgn) * rs.rem;
            const float err = exact_error((int)s - base_c, rs.rem, s64);
            unsigned long long best = ((unsigned long long)f32_orderable(err) << 32) | (d * (uint32_t)NISO + (uint32_t)k);
            if constexpr (MODE == 2) {
                const float err2 = exact_error((int)s2 - base_c, rs.rem, s64);
                const unsigned long long b2 = ((unsigned long long)f32_orderable(err2) << 32) | (d * (uint32_t)NISO + (uint32_t)k2);
                best = b2 < best ? b2 : best;
            }
            if (best < cur) atomicMin(&keyp[j], best);
        }
    }
}

// maximum over the lanes that share a range block: the two lane halves (rows 0-31 / 32-63 of the same column) and the
// range's 2^CSHIFT adjacent columns.  Every lane of the wave must be active.
// SEED: the prefix seed's CTW back-to-back shares exchange the lane halves through ds_bpermute instead of v_permlane32_swap --
// k_sweep_q16's seed lost winners with the sibling instruction v_permlane16_swap in exactly that position (DESIGN.md section 7,
// item 6; cause not found), and the seed runs once per pool chunk: the LDS crossbar costs nothing there.  Flagged tiles keep
// the swap (one share per tile, on the sweep's critical path, never seen to fail).
template <int MODE, bool SEED = false>
__device__ __forceinline__ float q_share_max(float v)
{
    if constexpr (QMode<MODE>::CSHIFT >= 2) {
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
    }
    if constexpr (QMode<MODE>::CSHIFT == 3)
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xF, 0xF, true)));  // row_half_mirror
    if constexpr (SEED) {
        const int partner = (int)((threadIdx.x & 63u) ^ 32u) << 2;
        return fmaxf(v, __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v))));
    } else {
        const uint32_t u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // {lanes 0-31 twice, lanes 32-63 twice}
        return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
}

// Fused finalise of a SMALL launch (one 512x512 image: k_finalize is 7 us of launch ramp behind a 58 us sweep).  Every wave
// calls this when it is done (also waves without work) and waits there for the other waves of its workgroup (a barrier: the
// waves hold nothing anybody needs, the launch is one round); the last workgroup of a (plane, column group) -- all pool
// chunks of that group have then merged their candidates into `key` -- does, with all its lanes, what k_finalize does for the
// group's range blocks (128 with 8 isometries, 512 with 1): the winner's covariance gathered from the image, fit, clamp,
// quantise, packed record (finalize_store).
// Visibility without any cache maintenance (an agent-scope fence = buffer_wbl2 + buffer_inv per wave: the sweep ran 2x longer
// with one): `key` is touched by nothing but 8-byte device-scope atomics in this kernel -- they execute at the memory side
// (MI355X_MICROARCH.md, "8-B agent atomics both sides") -- so a wave only has to WAIT for its own atomicMins (s_waitcnt
// vmcnt(0)) before the barrier; thread 0 then adds to the group's counter, and the workgroup whose add returns nchunks - 1
// reads every key by a returning atomic (umin with ~0).  Everything else it reads was written by the prep kernel.  The
// counter is left at 0 for the next launch.  s_wdone: the workgroup's "we are last" flag.
template <int NK, int MODE>
__device__ __forceinline__ void q_finalize_tail(const QArgs& A, int plane, int gx, int* s_wdone, int lane)
{
    (void)lane;
    if constexpr (MODE == 3) {
        return;
    } else {
        constexpr int CSHIFT = QMode<MODE>::CSHIFT, NISO = QMode<MODE>::NISO;
        constexpr int B = NK == 1 ? 4 : (NK == 4 ? 8 : 16);
        constexpr int CT = FIC_Q_WPG * fic_q_ctw(NK);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             // this wave's atomicMins have been performed
        __syncthreads();                                                         // ... and so have those of the workgroup's other waves
        if (threadIdx.x == 0) {
            unsigned int* cnt = A.fin_count + (size_t)plane * A.nctg + gx;
            const int last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)A.nchunks - 1u;
            if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_wdone = last;                                                     // the group's last workgroup?
        }
        __syncthreads();
        if (!*(volatile int*)s_wdone) return;
        const FicGeom& g = A.geom;
        const int j0 = ((A.ct_begin + gx * CT) * 32) >> CSHIFT, nj = (CT * 32) >> CSHIFT;
        for (int jj = (int)threadIdx.x; jj < nj; jj += 64 * FIC_Q_WPG) {
            const int j = j0 + jj;
            if (j < A.r_begin || j >= A.r_end) continue;
            // (a real read-modify-write, executed where the sweep's atomicMins were: the compiler would turn an idempotent
            //  __hip_atomic_fetch_min(p, ~0) into an L2-served load)
            unsigned long long kk;
            const unsigned long long none = FIC_KEY_NONE;
            asm volatile("global_atomic_umin_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(kk) : "v"(A.key + (size_t)plane * A.Nr_pad + j), "v"(none) : "memory");
            const uint32_t c = (uint32_t)kk;
            const int wloc = (int)(c / (uint32_t)NISO), k = (int)(c % (uint32_t)NISO);    // full search: window-local == pool index
            const int gi = wloc;
            const uint32_t* pp = (const uint32_t*)(A.pool_pix + ((size_t)plane * A.Nd_pad + gi) * A.n);
            const uint8_t* blk = A.gray + (size_t)plane * g.W * g.H + (size_t)((j / g.Rw) * B) * g.W + (j % g.Rw) * B;
            uint32_t acc = 0, s2;
            iso_dot<B, false>(blk, g.W, k, pp, acc, s2);
            finalize_store(A.out, g, plane, j, kk, wloc, gi, k, acc, A.pool_st[(size_t)plane * A.Nd_pad + gi],
                           A.rng_st[(size_t)plane * A.Nr_pad + j], A.pool_var[(size_t)plane * A.Nd_pad + gi]);
        }
    }
}

// Launch bounds: the register budget is cut for FIC_Q_WAVES_* waves per SIMD (<= 256 VGPRs at B = 4 / 8), which also makes the
// compiler emit the MFMAs with VGPR destinations whatever the build flags say (the library is built with -mllvm
// -amdgpu-mfma-vgpr-form; with accumulators in AGPRs every element the epilogue tests costs a v_accvgpr_read first).
// MULTI: the launch has more than one pool chunk (theta_g is shared between the waves of a range); a one-chunk launch is a
// separate instantiation without any of that code, see (3) in slow_tile.
// GFAST (k_sweep_qs, short pool chunks): theta_g is refreshed in the fast path (refresh_thg) instead of being loaded in every
// flagged tile.
template <int NK, int MODE, bool MULTI, bool GFAST>
__device__ __forceinline__ void sweep_q_body(const QArgs& A)
{
    constexpr int CTW = fic_q_ctw(NK), CT = FIC_Q_WPG * CTW;
    constexpr int CSHIFT = QMode<MODE>::CSHIFT;
    constexpr bool FOLD = MODE == 2;
    constexpr int NKA = FOLD ? NK / 2 : NK;                  // MFMA steps per accumulator
    __shared__ uint32_t sQ[FIC_Q_WPG][FIC_Q_QCAP + 4];         // per wave: domain block | column-in-wave << 24; [QCAP] = fill count
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int combo_, gx_;
    xcd_decode(blockIdx.x, A.nchunks * A.planes, A.nctg, combo_, gx_);
    const int plane = combo_ / A.nchunks;
    const int chunk = combo_ % A.nchunks;
    const int ctw0 = A.ct_begin + gx_ * CT + wave * CTW;     // first column tile of this wave
    const int dt0 = chunk * A.tiles_per_chunk;               // tiles_per_chunk is a multiple of the loop's unroll factor;
    int dt1 = dt0 + A.tiles_per_chunk;                       // the last chunk runs into the store's zero tiles (flagged flat)
    if (dt1 > A.ndtiles_loop) dt1 = A.ndtiles_loop;
    __shared__ int s_wdone;                                  // fused finalise: "this workgroup is its column group's last" (q_finalize_tail)
    if (dt0 >= A.ndtiles || ctw0 >= A.ct_end) {              // (waves are independent: no barrier in the sweep itself)
        if (A.fin_count) q_finalize_tail<NK, MODE>(A, plane, gx_, &s_wdone, lane);
        return;
    }
    const unsigned long long clk0 = A.stats ? clock64() : 0ull, tick0 = A.stats ? wall_clock64() : 0ull;
    int nci = A.ct_end - ctw0;                               // column tiles this wave really owns (wave-uniform)
    if (nci > CTW) nci = CTW;

    const int jcol = lane & 31, half = lane >> 5;
    const FicRngStat* rst = A.rng_st + (size_t)plane * A.Nr_pad;
    uint32_t* const thg = A.theta_g + (size_t)plane * A.Nr_pad;
    uint32_t* const myq = sQ[wave];
    uint32_t* const myqn = myq + FIC_Q_QCAP;                 // the queue's fill count (the slow path allocates with ds_add_rtn)
    const uint32_t myqn_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)myqn;   // its LDS byte address
    if (lane == 0) *myqn = 0u;

    v4i rb[CTW][NK];
    {
        const v4i* rp = A.rngQ + ((size_t)plane * A.nct_alloc + ctw0) * NK * 64 + lane;
#pragma unroll
        for (int ci = 0; ci < CTW; ci++)
#pragma unroll
            for (int m = 0; m < NK; m++) rb[ci][m] = rp[(ci * NK + m) * 64];
    }
    // per lane and column tile: the range behind the column, its theta, its error bound
    float tau[CTW], E[CTW], lmn[CTW];                        // lmn: L from which a pair of this range prunes in ANY index order (header)
    float gl[GFAST ? CTW : 1];                               // MULTI: theta_g of the lane's column per column tile, as last loaded (refresh_thg)
#pragma unroll
    for (int ci = 0; ci < (GFAST ? CTW : 1); ci++) gl[ci] = -1.0f;
    uint32_t okbits = 0, raise = 0;                          // bit ci: the column's range exists / may raise theta (rem != 0)
#pragma unroll
    for (int ci = 0; ci < CTW; ci++) {
        const int col = (ctw0 + ci) * 32 + jcol;
        const int j = col >> CSHIFT;
        const bool ok = j < A.Nr && ci < nci;
        const int rem = ok ? rst[j].rem : 0;
        E[ci] = ok ? A.rngE[(size_t)plane * A.Nr_pad + j] : 0.0f;
        // rem == 0: error 0 for every block (FC:677) -> only candidate 0 can win; padding: never flagged
        tau[ci] = (ok && rem != 0 && !A.dbg_noflag) ? FIC_Q_TAU_NONE : FIC_Q_TAU_ALL;
        lmn[ci] = FIC_Q_LMIN * (float)rem;
        okbits |= ok ? 1u << ci : 0u;
        raise |= (ok && rem != 0) ? 1u << ci : 0u;
    }

    const v4i* pa = A.poolQ + (size_t)plane * A.ndtiles_alloc * NK * 64 + lane;
    // flat-tile flags through the scalar cache (wave-uniform address; a vector load here would make the slow path wait for
    // the prefetched fragment loads as well: s_waitcnt vmcnt(0))
    const uint32_t AS4* pflat = (const uint32_t AS4*)(uintptr_t)(A.dflat + (size_t)plane * A.ndtiles_alloc);
    const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int qn = 0;                                              // queued entries (wave-uniform)
    unsigned st_slow = 0, st_pairs = 0;                      // instrumentation (wave-uniform; reported when A.stats is set)

    auto flush = [&]() __attribute__((always_inline)) {
        q_flush<NK, MODE>(A, myq, qn, plane, ctw0, lane);
        qn = 0;
        if (lane == 0) *myqn = 0u;
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): atomics left in flight here make the compiler tighten the
    };                                                       // fast path's waits for the prefetched fragments
    // test value of element e: |q| (MODE 0/1) or |even| + |odd| = the larger |q| of the isometry pair (MODE 2)
    auto val = [&](const v16f& acc, const v16f& acc2, int e) __attribute__((always_inline)) {
        return FOLD ? fabsf(acc[e]) + fabsf(acc2[e]) : fabsf(acc[e]);
    };
    // A tile with flagged entries (or a chunk's first tile): queue the entries, raise theta.
    //   theta moves three ways: (1) FIRST: seeded from the tile's largest test value per range -- out of index order, hence
    //   only when that pair's L >= lmn = 0.26 rem (header); (2) from the flagged entries of this tile (they have lower indices than
    //   everything still to come in this chunk); (3) from theta_g, the best level any wave has published for the range --
    //   other chunks, any index order, hence published only for pairs with L >= lmn.
    auto slow_tile = [&](const v16f& acc, const v16f& acc2, float mx, int ci, int dt, bool first) __attribute__((always_inline)) {
        // only zeros flagged (theta still "none"): an all-flat tile seen before anything was evaluated needs nothing
        if (!first && __builtin_amdgcn_ballot_w64(mx > 0.0f) == 0 && pflat[dt] != 0u) return;
        st_slow++;
        const bool ok = (okbits >> ci) & 1u;
        const bool mayraise = (raise >> ci) & 1u;
        const int colw = ci * 32 + jcol;                     // column in wave
        const int jg = ((ctw0 + ci) * 32 + jcol) >> CSHIFT;  // the column's range block (valid when ok)
        // (3): a load, waited for at the end of the tile -- a memory round trip, 1.1 us per flagged tile against 0.4 us without
        // (tools/phases.py; merging it a flagged tile later is no cheaper: the vector-memory counter is in-order, the next wait
        // for domain fragments waits for it too).  Fine where flagged tiles are rare (long chunks); with short chunks -- one
        // 512x512 image: 19 of the 46 us a wave spent in its loop -- the GFAST instantiation refreshes theta_g in the fast path
        // instead (refresh_thg in step(): +3..11 % on long chunks, hence not there).  The one-chunk instantiation -- no other wave
        // sees this range -- has no vector memory operation in a flagged tile at all.
        const uint32_t g = (MULTI && !GFAST && mayraise) ? thg[jg] : FIC_Q_THG_NONE;
        const uint32_t ent0 = (uint32_t)(dt * 32 + 4 * half) | ((uint32_t)colw << 24);   // entry of element 0; element e adds its row
        if (first) {                                         // (1)
            const float lo = __fsub_rn(q_share_max<MODE>(ok ? mx : 0.0f), E[ci]);
            if (mayraise && lo >= lmn[ci]) tau[ci] = fmaxf(tau[ci], __fsub_rn(__fmul_rn(lo, FIC_Q_LEVEL), E[ci]));
        }
        // candidate 0 = (block 0, copy 0): always evaluated (MODE 2: with its partner, copy 2)
        const bool cand0 = first && dt == 0 && half == 0 && ok && (jcol & ((1 << CSHIFT) - 1)) == 0;
        // Per-lane mask of the elements above theta, built without scalar branches: bit e = sign(theta - value_e), shifted in by
        // v_alignbit.  (Strict '>': value == theta gives +0.  theta = "none" (-1) flags everything, zeros included.)
        uint32_t hm = 0;
#pragma unroll
        for (int e = 15; e >= 0; e--)
            hm = __builtin_amdgcn_alignbit(hm, __float_as_uint(__fsub_rn(tau[ci], val(acc, acc2, e))), 31);
        if (!ok) hm = 0;
        if (cand0) hm |= 1u;
        const int cnt = __builtin_popcount(hm);
        // Queue slots: every lane that has something takes cnt consecutive entries from the wave's fill count (an LDS atomic; no
        // scalar walk over the lanes).  The first entry is written without a loop -- nearly always the only one.
        if (cnt != 0) {
            // (inline asm: as an atomicAdd the compiler's atomic optimiser turns it back into a scalar loop over the lanes)
            uint32_t w;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(myqn_lds), "v"((uint32_t)cnt) : "memory");
            int e = __builtin_ctz(hm);
            myq[w] = ent0 + (uint32_t)((e & 3) + 8 * (e >> 2));
            for (uint32_t h = hm & (hm - 1); h; h &= h - 1) {
                e = __builtin_ctz(h);
                myq[++w] = ent0 + (uint32_t)((e & 3) + 8 * (e >> 2));
            }
        }
        // a lane with flagged elements has its largest element among them: that is mx
        const float mp = cnt != 0 ? mx : -1.0f;
        // (2): level of the best flagged entry, shared by the lanes of the range
        const float lo2 = q_share_max<MODE>((mayraise && mp >= 0.0f) ? __fsub_rn(mp, E[ci]) : -1.0f);
        if (mayraise && lo2 >= 0.0f) {
            const float lb = __fsub_rn(__fmul_rn(lo2, FIC_Q_LEVEL), E[ci]);
            tau[ci] = fmaxf(tau[ci], lb);
            if (MULTI && lo2 >= lmn[ci] && lb > 0.0f && half == 0 && (jcol & ((1 << CSHIFT) - 1)) == 0) atomicMax((int*)&thg[jg], __float_as_int(lb));
        }
        if (MULTI && !GFAST && mayraise) tau[ci] = fmaxf(tau[ci], __uint_as_float(g));
        // the fill count after this tile (LDS operations of a wave complete in order); one tile adds at most 1024 entries
        const int tot = __builtin_amdgcn_readfirstlane((int)*(volatile __attribute__((address_space(3))) uint32_t*)(__attribute__((address_space(3))) uint32_t*)myqn);   // ds_read, not a flat load
        st_pairs += (unsigned)(tot - qn);
        qn = tot;
        if (qn > FIC_Q_QFLUSH) flush();
    };
    // accumulator(s) of one 32x32 tile: FOLD: even part from steps [0, NK/2), odd part from steps [NK/2, NK)
    auto tile_mfma = [&](const v4i (&at)[NK], const v4i (&bt)[NK], v16f& acc, v16f& acc2) __attribute__((always_inline)) {
        acc = zero;
#pragma unroll
        for (int m = 0; m < NKA; m++) acc = mfma_f16(at[m], bt[m], acc);
        if constexpr (FOLD) {
            acc2 = zero;
#pragma unroll
            for (int m = NKA; m < NK; m++) acc2 = mfma_f16(at[m], bt[m], acc2);
        }
    };

    // Two fragment buffers.  A step computes on `ac` (domain tile dt) and finishes with the first tile of `an` (tile dt + 1);
    // as soon as the last MFMAs that read `ac` have been issued (column tile CTW-1), `ac` is reloaded with tile dt + 2, which
    // is first needed at the end of the NEXT step: every load has CTW + 1 tile epilogues (a step and a quarter) to land.
    v4i a0[NK], a1[NK];
    v16f acc, acc2 = zero;
    // Prefix seed (pool chunks after the first).  theta may be the level of ANY pair X with a LOWER index than the pairs it
    // prunes, evaluated or not: if X was pruned itself, then by a pair that beats or ties-and-precedes it, and that pair beats a
    // later Y with L(Y) <= (1 - 2^-18) L(X) all the same (header, "Invariant").  Domain tile 0 holds the lowest indices of the
    // pool, so a chunk that starts further in takes its first theta from tile 0's largest test value per range -- one tile of
    // extra matrix work per chunk -- instead of starting at "none", where the chunk's whole first tile is evaluated exactly for
    // every range that cannot seed out of order (L < 0.26 rem: the smooth blocks of natural images).
    if constexpr (MULTI && FIC_Q_SEED) {
        if (dt0 != 0) {                                      // (wave-uniform)
#pragma unroll
            for (int m = 0; m < NK; m++) a0[m] = pa[(size_t)m * 64];
#pragma unroll
            for (int ci = 0; ci < CTW; ci++) {
                tile_mfma(a0, rb[ci], acc, acc2);
                const float mx = FOLD ? max16_sum(acc, acc2) : max16_abs(acc);
                const bool ok = (okbits >> ci) & 1u, mayraise = (raise >> ci) & 1u;
                const float lo = __fsub_rn(q_share_max<MODE, true>(ok ? mx : 0.0f), E[ci]);
                if (mayraise && lo > 0.0f) tau[ci] = fmaxf(tau[ci], __fsub_rn(__fmul_rn(lo, FIC_Q_LEVEL), E[ci]));
            }
        }
    }
#pragma unroll
    for (int m = 0; m < NK; m++) a0[m] = pa[((size_t)dt0 * NK + m) * 64];
#pragma unroll
    for (int m = 0; m < NK; m++) a1[m] = pa[((size_t)(dt0 + 1) * NK + m) * 64];
    tile_mfma(a0, rb[0], acc, acc2);

    // one domain tile (the fragment store has spare zero tiles behind the pool for the loop's overrun and the prefetch)
    // (3) theta_g, the best level any wave has published for the lane's ranges (other pool chunks): every step merges the levels of
    // half of the column tiles -- loaded two steps ago, next to the fragment loads, so nobody waits for them -- and requests
    // them again: a load and a v_max per column tile and two domain tiles.  PH = which half (the loop alternates two steps).
    auto refresh_thg = [&](auto ph) __attribute__((always_inline)) {
        if constexpr (MULTI && GFAST) {
            constexpr int PH = decltype(ph)::value, H = CTW / 2;
#pragma unroll
            for (int c = PH * H; c < PH * H + H; c++) {
                tau[c] = fmaxf(tau[c], gl[c]);               // (columns that never raise hold TAU_ALL; "nothing published" is -1)
                gl[c] = __uint_as_float(thg[((ctw0 + c) * 32 + jcol) >> CSHIFT]);
            }
        }
    };
    auto step = [&](int dt, v4i (&ac)[NK], const v4i (&an)[NK], const bool first, auto ph) __attribute__((always_inline)) {
#pragma unroll
        for (int ci = 0; ci < CTW; ci++) {
            // the next tile's MFMAs: column tile ci+1 of this domain tile, or column tile 0 of the next domain tile
            v16f nacc, nacc2 = zero;
            if (ci + 1 < CTW) tile_mfma(ac, rb[ci + 1], nacc, nacc2);
            else tile_mfma(an, rb[0], nacc, nacc2);
            if (ci == CTW - 2) {
#pragma unroll
                for (int m = 0; m < NK; m++) ac[m] = pa[((size_t)(dt + 2) * NK + m) * 64];
                refresh_thg(ph);
            }
            const float mx = FOLD ? max16_sum(acc, acc2) : max16_abs(acc);
            const bool hit = mx > tau[ci];
            if constexpr (NK == 4 && !FOLD) {
                // M M v v v M v v v M v v v (8 x v_max3 + compare): the epilogue reads a tile whose last MFMA was issued >= 64 cycles ago
                __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
            } else if constexpr (NK == 4 && FOLD) {
#ifndef FIC_Q_SCHED
#define FIC_Q_SCHED 3
#endif
#if FIC_Q_SCHED == 0
                // 4 MFMAs (two 2-step accumulators), 16 adds + 8 max + compare: M M v*8 M v*8 M v*9
                __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 9, 0);
#elif FIC_Q_SCHED == 2
                __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 25, 0);
#elif FIC_Q_SCHED == 3
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 7, 0);
#endif
            }
            // a chunk's first tile always takes the slow path: theta is seeded there and candidate 0 is queued
            if (__builtin_expect((__builtin_amdgcn_ballot_w64(hit) != 0) | first, 0)) slow_tile(acc, acc2, mx, ci, dt, first);
            acc = nacc;
            if constexpr (FOLD) acc2 = nacc2;
        }
    };
#ifdef FIC_Q_PHASES
    const unsigned long long ph1 = clock64();
#endif
    for (int dt = dt0; dt < dt1; dt += 2) {
        step(dt, a0, a1, dt == dt0, std::integral_constant<int, 0>());     // first: a chunk's first tile (theta is seeded there)
        step(dt + 1, a1, a0, false, std::integral_constant<int, 1>());
    }
#ifdef FIC_Q_PHASES
    const unsigned long long ph2 = clock64();
#endif
    flush();
#ifdef FIC_Q_PHASES
    // diagnostic build: the sampled waves report where their cycles went -- prologue | loop | final flush; [3] = loop cycles of the first step pair
    if (A.stats && lane == 0 && (blockIdx.x & 15) == 0 && wave == 0) {
        atomicAdd(&A.stats[0], ph1 - clk0);
        atomicAdd(&A.stats[1], ph2 - ph1);
        atomicAdd(&A.stats[2], (unsigned long long)clock64() - ph2);
        atomicAdd(&A.stats[4], (unsigned long long)(clock64() - clk0));
        atomicAdd(&A.stats[5], (unsigned long long)(wall_clock64() - tick0));
        atomicAdd(&A.stats[6], 1ull);
        atomicAdd(&A.stats[3], (unsigned long long)st_slow);
        atomicAdd(&A.stats[7], (unsigned long long)st_pairs);
    }
    if (false)
#endif
    if (A.stats && lane == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)(dt1 - dt0) * (unsigned)nci);
        atomicAdd(&A.stats[1], (unsigned long long)st_slow);
        atomicAdd(&A.stats[2], (unsigned long long)st_pairs);
        atomicAdd(&A.stats[3], 1ull);
        if ((blockIdx.x & 15) == 0 && wave == 0) {          // a sample: same-address atomics serialise
            atomicAdd(&A.stats[4], (unsigned long long)(clock64() - clk0));
            atomicAdd(&A.stats[5], (unsigned long long)(wall_clock64() - tick0));
            atomicAdd(&A.stats[6], 1ull);
        }
    }
    if (A.fin_count) q_finalize_tail<NK, MODE>(A, plane, gx_, &s_wdone, lane);
}

// ---------------------------------------------------------------------------------------------
// k_sweep_q16<NK, MULTI> : the 1-isometry sweep (the reference algorithm, MODE 0) at B = 8 / 16 on v_mfma_f32_16x16x32_f16.
// Same search, same queue, same exact path (q_flush) as k_sweep_q<NK, 0>; what changes is the matrix instruction and with it
// the lane layout.  Why: the dense sweeps are power-limited (the chip holds 1.6-1.75 GHz under 32x32x16), and it holds a
// higher clock under the 16x16x32 shape at the same work per output -- tools/mfma_shape.hip, this kernel's fast path on random
// data: 1.93-1.96 GHz against 1.60-1.76, 7-15 % less time (profiles/r03g_mfma_shape_16x16x32_vs_32x32x16_microbench.txt).
//   unit = 32 domain blocks x 16 range columns: 2 row tiles x KS = NK/2 chained MFMAs of K = 32 -> two 4-element accumulators;
//   lane (c = lane & 15, g = lane >> 4) holds column c against rows 16 rt + 4 g + i (rt = accumulator, i = element):
//   ONE column per lane and unit, so theta stays one VGPR per unit (2 CTW units per wave); the four lanes of a column
//   (g = 0..3) share raised values by v_permlane16_swap + v_permlane32_swap.  Epilogue per unit: 4 v_max3 + compare.
//   Fragments come in the [sub-tile][K step][lane] order of frag_slot(shape 1) from k_pool_q / k_range_q*.
// ---------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mfma16_f16(v4i a, v4i b, v4f c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float max8_abs(const v4f& x0, const v4f& x1)
{
    const float m0 = max3f(fabsf(x0[0]), fabsf(x0[1]), fabsf(x0[2])), m1 = max3f(fabsf(x0[3]), fabsf(x1[0]), fabsf(x1[1]));
    return max3f(m0, m1, max3f(fabsf(x1[2]), fabsf(x1[3]), 0.0f));
}
// maximum over the four lanes (row groups g = 0..3) that hold the same column.  Every lane of the wave must be active.
// (The exchange between rows 0/1 and 2/3 goes through ds_bpermute, not v_permlane16_swap: a build with the swap here lost winners
//  in the units whose prefix seed it computed -- deterministically, on 512x512 pools, with an ISA that reads correctly; every
//  variant without the swap was right.  Cause not found; this is slow-path code, the LDS crossbar
//  costs nothing measurable.)
template <bool SEED = false>
__device__ __forceinline__ float q16_share_max(float v)
{
    const int partner = (int)((threadIdx.x & 63u) ^ 16u) << 2;
    const float w = fmaxf(v, __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v))));
    if constexpr (SEED) return fmaxf(w, __int_as_float(__builtin_amdgcn_ds_bpermute((int)((threadIdx.x & 63u) ^ 32u) << 2, __float_as_int(w))));
    const uint32_t uw = __float_as_uint(w);
    const auto q = __builtin_amdgcn_permlane32_swap(uw, uw, false, false);    // the two halves
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
template <int NK, bool MULTI, bool GFAST>
__device__ __forceinline__ void sweep_q16_body(const QArgs& A)
{
    constexpr int CTW = fic_q_ctw(NK), CT = FIC_Q_WPG * CTW, CU = 2 * CTW, KS = NK / 2;
    __shared__ uint32_t sQ[FIC_Q_WPG][FIC_Q_QCAP + 4];         // per wave: domain block | column-in-wave << 24; [QCAP] = fill count
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int combo_, gx_;
    xcd_decode(blockIdx.x, A.nchunks * A.planes, A.nctg, combo_, gx_);
    const int plane = combo_ / A.nchunks;
    const int chunk = combo_ % A.nchunks;
    const int ctw0 = A.ct_begin + gx_ * CT + wave * CTW;     // first column tile (x32 columns) of this wave
    const int dt0 = chunk * A.tiles_per_chunk;
    int dt1 = dt0 + A.tiles_per_chunk;
    if (dt1 > A.ndtiles_loop) dt1 = A.ndtiles_loop;
    __shared__ int s_wdone;                                  // fused finalise: "this workgroup is its column group's last" (q_finalize_tail)
    if (dt0 >= A.ndtiles || ctw0 >= A.ct_end) {              // (waves are independent: no barrier in the sweep itself)
        if (A.fin_count) q_finalize_tail<NK, 0>(A, plane, gx_, &s_wdone, lane);
        return;
    }
    const unsigned long long clk0 = A.stats ? clock64() : 0ull, tick0 = A.stats ? wall_clock64() : 0ull;
    int nci = A.ct_end - ctw0;                               // column tiles this wave really owns (wave-uniform)
    if (nci > CTW) nci = CTW;

    const int c16 = lane & 15, g4 = lane >> 4;
    const FicRngStat* rst = A.rng_st + (size_t)plane * A.Nr_pad;
    uint32_t* const thg = A.theta_g + (size_t)plane * A.Nr_pad;
    uint32_t* const myq = sQ[wave];
    uint32_t* const myqn = myq + FIC_Q_QCAP;
    const uint32_t myqn_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)myqn;
    if (lane == 0) *myqn = 0u;

    v4i rb[CTW][NK];                                         // rb[ci][sub * KS + ks]: columns 16 sub + (lane & 15) of column tile ci
    {
        const v4i* rp = A.rngQ + ((size_t)plane * A.nct_alloc + ctw0) * NK * 64 + lane;
#pragma unroll
        for (int ci = 0; ci < CTW; ci++)
#pragma unroll
            for (int m = 0; m < NK; m++) rb[ci][m] = rp[(ci * NK + m) * 64];
    }
    float tau[CU], E[CU], lmn[CU];
    float gl[GFAST ? CU : 1];
#pragma unroll
    for (int cj = 0; cj < (GFAST ? CU : 1); cj++) gl[cj] = -1.0f;
    uint32_t okbits = 0, raise = 0;
#pragma unroll
    for (int cj = 0; cj < CU; cj++) {
        const int j = ctw0 * 32 + cj * 16 + c16;             // the lane's column of unit cj = its range block
        const bool ok = j < A.Nr && (cj >> 1) < nci;
        const int rem = ok ? rst[j].rem : 0;
        E[cj] = ok ? A.rngE[(size_t)plane * A.Nr_pad + j] : 0.0f;
        tau[cj] = (ok && rem != 0 && !A.dbg_noflag) ? FIC_Q_TAU_NONE : FIC_Q_TAU_ALL;
        lmn[cj] = FIC_Q_LMIN * (float)rem;
        okbits |= ok ? 1u << cj : 0u;
        raise |= (ok && rem != 0) ? 1u << cj : 0u;
    }
    const v4i* pa = A.poolQ + (size_t)plane * A.ndtiles_alloc * NK * 64 + lane;
    const uint32_t AS4* pflat = (const uint32_t AS4*)(uintptr_t)(A.dflat + (size_t)plane * A.ndtiles_alloc);
    const v4f zero = {0, 0, 0, 0};
    int qn = 0;
    unsigned st_slow = 0, st_pairs = 0;

    auto flush = [&]() __attribute__((always_inline)) {
        q_flush<NK, 0>(A, myq, qn, plane, ctw0, lane);
        qn = 0;
        if (lane == 0) *myqn = 0u;
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0), as in k_sweep_q
    };
    // element e of the lane's unit: accumulator e >> 2, element e & 3 = row 16 (e >> 2) + 4 g + (e & 3)
    auto val = [&](const v4f& x0, const v4f& x1, int e) __attribute__((always_inline)) { return fabsf(e < 4 ? x0[e] : x1[e - 4]); };
    // A unit with flagged entries (or a chunk's first tile): exactly slow_tile of k_sweep_q with this kernel's lane layout
    auto slow_unit = [&](const v4f& x0, const v4f& x1, float mx, int cj, int dt, bool first) __attribute__((always_inline)) {
        if (!first && __builtin_amdgcn_ballot_w64(mx > 0.0f) == 0 && pflat[dt] != 0u) return;
        st_slow++;
        const bool ok = (okbits >> cj) & 1u;
        const bool mayraise = (raise >> cj) & 1u;
        const int colw = cj * 16 + c16;                      // column in wave
        const int jg = ctw0 * 32 + colw;                     // the column's range block (valid when ok)
        const uint32_t g = (MULTI && !GFAST && mayraise) ? thg[jg] : FIC_Q_THG_NONE;    // (3); GFAST: in the fast path (refresh_thg)
        const uint32_t ent0 = (uint32_t)(dt * 32 + 4 * g4) | ((uint32_t)colw << 24);     // entry of element 0; element e adds 16 (e >> 2) + (e & 3)
        if (first) {                                         // (1) out-of-order seed, only from a pair with L >= lmn
            const float lo = __fsub_rn(q16_share_max<>(ok ? mx : 0.0f), E[cj]);
            if (mayraise && lo >= lmn[cj]) tau[cj] = fmaxf(tau[cj], __fsub_rn(__fmul_rn(lo, FIC_Q_LEVEL), E[cj]));
        }
        const bool cand0 = first && dt == 0 && g4 == 0 && ok;                // candidate 0 = block 0: always evaluated
        uint32_t hm = 0;
#pragma unroll
        for (int e = 7; e >= 0; e--)
            hm = __builtin_amdgcn_alignbit(hm, __float_as_uint(__fsub_rn(tau[cj], val(x0, x1, e))), 31);
                                                                             // (8 elements shifted in: bits 0..7)
        if (!ok) hm = 0;
        if (cand0) hm |= 1u;
        const int cnt = __builtin_popcount(hm);
        if (cnt != 0) {
            uint32_t w;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(myqn_lds), "v"((uint32_t)cnt) : "memory");
            int e = __builtin_ctz(hm);
            myq[w] = ent0 + (uint32_t)((e & 3) + 16 * (e >> 2));
            for (uint32_t h = hm & (hm - 1); h; h &= h - 1) {
                e = __builtin_ctz(h);
                myq[++w] = ent0 + (uint32_t)((e & 3) + 16 * (e >> 2));
            }
        }
        const float mp = cnt != 0 ? mx : -1.0f;
        const float lo2 = q16_share_max<>((mayraise && mp >= 0.0f) ? __fsub_rn(mp, E[cj]) : -1.0f);   // (2)
        if (mayraise && lo2 >= 0.0f) {
            const float lb = __fsub_rn(__fmul_rn(lo2, FIC_Q_LEVEL), E[cj]);
            tau[cj] = fmaxf(tau[cj], lb);
            if (MULTI && lo2 >= lmn[cj] && lb > 0.0f && g4 == 0) atomicMax((int*)&thg[jg], __float_as_int(lb));
        }
        if (MULTI && !GFAST && mayraise) tau[cj] = fmaxf(tau[cj], __uint_as_float(g));       // (3)
        const int tot = __builtin_amdgcn_readfirstlane((int)*(volatile __attribute__((address_space(3))) uint32_t*)(__attribute__((address_space(3))) uint32_t*)myqn);
        st_pairs += (unsigned)(tot - qn);
        qn = tot;
        if (qn > FIC_Q_QFLUSH) flush();
    };
    // the two accumulators of unit cj against domain fragments at[rt * KS + ks]
    auto unit_mfma = [&](const v4i (&at)[NK], int cj, v4f& x0, v4f& x1) __attribute__((always_inline)) {
        x0 = zero;
        x1 = zero;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) x0 = mfma16_f16(at[ks], rb[cj >> 1][(cj & 1) * KS + ks], x0);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) x1 = mfma16_f16(at[KS + ks], rb[cj >> 1][(cj & 1) * KS + ks], x1);
    };

    v4i a0[NK], a1[NK];
    v4f x0, x1;
    if constexpr (MULTI && FIC_Q_SEED) {                     // prefix seed from domain tile 0, as in k_sweep_q
        if (dt0 != 0) {
#pragma unroll
            for (int m = 0; m < NK; m++) a0[m] = pa[(size_t)m * 64];
#pragma unroll
            for (int cj = 0; cj < CU; cj++) {
                unit_mfma(a0, cj, x0, x1);
                const float mx = max8_abs(x0, x1);
                const bool ok = (okbits >> cj) & 1u, mayraise = (raise >> cj) & 1u;
                const float lo = __fsub_rn(q16_share_max<true>(ok ? mx : 0.0f), E[cj]);
                if (mayraise && lo > 0.0f) tau[cj] = fmaxf(tau[cj], __fsub_rn(__fmul_rn(lo, FIC_Q_LEVEL), E[cj]));
            }
        }
    }
#pragma unroll
    for (int m = 0; m < NK; m++) a0[m] = pa[((size_t)dt0 * NK + m) * 64];
#pragma unroll
    for (int m = 0; m < NK; m++) a1[m] = pa[((size_t)(dt0 + 1) * NK + m) * 64];
    unit_mfma(a0, 0, x0, x1);
    auto refresh_thg = [&](auto ph) __attribute__((always_inline)) {      // as in k_sweep_q: half of the units per step
        if constexpr (MULTI && GFAST) {
            constexpr int PH = decltype(ph)::value, H = CU / 2;
#pragma unroll
            for (int c = PH * H; c < PH * H + H; c++) {
                tau[c] = fmaxf(tau[c], gl[c]);
                gl[c] = __uint_as_float(thg[ctw0 * 32 + c * 16 + c16]);
            }
        }
    };
    auto step = [&](int dt, v4i (&ac)[NK], const v4i (&an)[NK], const bool first, auto ph) __attribute__((always_inline)) {
#pragma unroll
        for (int cj = 0; cj < CU; cj++) {
            v4f n0, n1;
            if (cj + 1 < CU) unit_mfma(ac, cj + 1, n0, n1);
            else unit_mfma(an, 0, n0, n1);
            if (cj == CU - 2) {
#pragma unroll
                for (int m = 0; m < NK; m++) ac[m] = pa[((size_t)(dt + 2) * NK + m) * 64];
                refresh_thg(ph);
            }
            const float mx = max8_abs(x0, x1);
            const bool hit = mx > tau[cj];
            if (__builtin_expect((__builtin_amdgcn_ballot_w64(hit) != 0) | first, 0)) slow_unit(x0, x1, mx, cj, dt, first);
            x0 = n0;
            x1 = n1;
        }
    };
    for (int dt = dt0; dt < dt1; dt += 2) {
        step(dt, a0, a1, dt == dt0, std::integral_constant<int, 0>());
        step(dt + 1, a1, a0, false, std::integral_constant<int, 1>());
    }
    flush();
    if (A.stats && lane == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)(dt1 - dt0) * (unsigned)nci);
        atomicAdd(&A.stats[1], (unsigned long long)st_slow);
        atomicAdd(&A.stats[2], (unsigned long long)st_pairs);
        atomicAdd(&A.stats[3], 1ull);
        if ((blockIdx.x & 15) == 0 && wave == 0) {
            atomicAdd(&A.stats[4], (unsigned long long)(clock64() - clk0));
            atomicAdd(&A.stats[5], (unsigned long long)(wall_clock64() - tick0));
            atomicAdd(&A.stats[6], 1ull);
        }
    }
    if (A.fin_count) q_finalize_tail<NK, 0>(A, plane, gx_, &s_wdone, lane);
}

// The kernels: one pool chunk / several (theta_g shared between the waves of a range) / several SHORT ones (k_sweep_qs).
template <int NK, int MODE, bool MULTI>
__global__ __launch_bounds__(64 * FIC_Q_WPG, NK == 4 ? FIC_Q_WAVES_B8 : (NK < 4 ? FIC_Q_WAVES_B4 : 1)) void k_sweep_q(QArgs A)
{
    sweep_q_body<NK, MODE, MULTI, false>(A);
}
template <int NK, int MODE>
__global__ __launch_bounds__(64 * FIC_Q_WPG, NK == 4 ? FIC_Q_WAVES_B8 : (NK < 4 ? FIC_Q_WAVES_B4 : 1)) void k_sweep_qs(QArgs A)
{
    sweep_q_body<NK, MODE, true, true>(A);
}
template <int NK, bool MULTI>
__global__ __launch_bounds__(64 * FIC_Q_WPG, NK == 4 ? FIC_Q_WAVES_B8 : 1) void k_sweep_q16(QArgs A)
{
    sweep_q16_body<NK, MULTI, false>(A);
}
template <int NK>
__global__ __launch_bounds__(64 * FIC_Q_WPG, NK == 4 ? FIC_Q_WAVES_B8 : 1) void k_sweep_q16s(QArgs A)
{
    sweep_q16_body<NK, true, true>(A);
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
// multi: 0 = one pool chunk, 1 = several, 2 = several short ones (fic_q_multi_kind)
template <int NK, int MODE>
static void q_launch_nm(int multi, dim3 grid, dim3 block, hipStream_t s, const QArgs& A)
{
    if (multi == 2) hipLaunchKernelGGL((k_sweep_qs<NK, MODE>), grid, block, 0, s, A);
    else if (multi) hipLaunchKernelGGL((k_sweep_q<NK, MODE, true>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((k_sweep_q<NK, MODE, false>), grid, block, 0, s, A);
}
int fic_q_multi_kind(int nchunks, int tiles_per_chunk) { return nchunks <= 1 ? 0 : (tiles_per_chunk <= FIC_Q_GFAST_TILES ? 2 : 1); }
// NK = n / 16; mode 0 / 1 / 2 as fic_q_mode, 3 = joint RGB.  false: no such kernel
static bool q_launch(int NK, int mode, int multi, dim3 grid, dim3 block, hipStream_t s, const QArgs& A, int shape16 = 0)
{
    if (shape16 && mode == 0 && NK == 4) {
        if (multi == 2) hipLaunchKernelGGL((k_sweep_q16s<4>), grid, block, 0, s, A);
        else if (multi) hipLaunchKernelGGL((k_sweep_q16<4, true>), grid, block, 0, s, A);
        else hipLaunchKernelGGL((k_sweep_q16<4, false>), grid, block, 0, s, A);
    } else if (shape16 && mode == 0 && NK == 16) {
        if (multi == 2) hipLaunchKernelGGL((k_sweep_q16s<16>), grid, block, 0, s, A);
        else if (multi) hipLaunchKernelGGL((k_sweep_q16<16, true>), grid, block, 0, s, A);
        else hipLaunchKernelGGL((k_sweep_q16<16, false>), grid, block, 0, s, A);
    } else if (NK == 1 && mode == 0) q_launch_nm<1, 0>(multi, grid, block, s, A);
    else if (NK == 1 && mode == 1) q_launch_nm<1, 1>(multi, grid, block, s, A);
    else if (NK == 1 && mode == 3) q_launch_nm<1, 3>(multi, grid, block, s, A);
    else if (NK == 4 && mode == 0) q_launch_nm<4, 0>(multi, grid, block, s, A);
    else if (NK == 4 && mode == 2) q_launch_nm<4, 2>(multi, grid, block, s, A);
    else if (NK == 4 && mode == 3) q_launch_nm<4, 3>(multi, grid, block, s, A);
    else if (NK == 16 && mode == 0) q_launch_nm<16, 0>(multi, grid, block, s, A);
    else if (NK == 16 && mode == 2) q_launch_nm<16, 2>(multi, grid, block, s, A);
    else if (NK == 16 && mode == 3) q_launch_nm<16, 3>(multi, grid, block, s, A);
    else return false;
    return true;
}
int fic_q_ct(int B) { return FIC_Q_WPG * fic_q_ctw(B * B / 16); }       // column tiles (x32 columns) per workgroup
int fic_q_ctw_host(int B) { return fic_q_ctw(B * B / 16); }
// 0: 1 isometry; 1: 8 isometries, one column per copy; 2: 8 isometries folded into 4 columns per range block
int fic_q_mode(int B, int n_iso) { return n_iso == 1 ? 0 : (B == 4 ? 1 : 2); }
int fic_q_cols_per_range(int B, int n_iso) { const int m = fic_q_mode(B, n_iso); return m == 0 ? 1 : (m == 1 ? 8 : 4); }
int fic_q_unroll(int B, int n_iso) { (void)B; (void)n_iso; return FIC_Q_UNROLL; }
// 1: this 1-isometry sweep runs on v_mfma_f32_16x16x32_f16 (k_sweep_q16; fragments in that lane order).  Measured
// (profiles/r03h_mfma_16x16x32_vs_32x32x16_1iso_sweep_ab.txt): the chip holds 1.78 GHz under it against 1.55 under 32x32x16 and
// large pools gain 7.5 % (4096x4096: B = 8 26.1 -> 24.3 ms, B = 16 7.21 -> 6.71 ms); its units are 16 columns wide, so a
// small pool -- where a fifth of the tiles is flagged -- pays for twice as many flagged units and loses 3-15 % (64 x 512x512,
// 4 x 2048x2048, one image).  Hence only from 10^5 MFMA K-steps per range column on: B = 8 from ~25 000 domain tiles, B = 16 from ~6 000.
int fic_q_shape16(const FicGeom& g)
{
    const long long ndtiles = (g.Nd + 31) / 32;
    int want = g.q_shape;                                    // option "q_shape"; FIC_Q_SHAPE=1|2 does the same process-wide (profiling runs)
    if (want == 0) {
        const char* env = getenv("FIC_Q_SHAPE");
        if (env && (env[0] == '1' || env[0] == '2') && env[1] == '\0') want = env[0] - '0';
    }
    if (!FIC_Q_SHAPE16 || g.n_iso != 1 || g.B < 8 || want == 2) return 0;
    return (want == 1 || ndtiles * (g.n / 16) >= 100000) ? 1 : 0;
}
// workgroups of k_sweep_q a CU holds at once (a workgroup puts one wave on each SIMD; VGPR-bound)
int fic_q_resident(int B) { return (B == 4 ? FIC_Q_WAVES_B4 : (B == 8 ? FIC_Q_WAVES_B8 : 1)) * 4 / FIC_Q_WPG; }

// 1: the prep of this launch is one fused kernel that makes the scaled image itself (fic_ctx_encode then skips k_scale)
int fic_q_prep_fused(const FicGeom& g, int ndtiles_alloc, int ngrp)
{
    return (g.B == 8 && (long long)g.planes * (ndtiles_alloc + ngrp) <= 1024) ? 1 : 0;
}
int fic_launch_q_prep(const FicBuffers& b, void* poolQ, void* dflat, void* rngQ, void* rngC, void* rngE, void* theta_g,
                      const FicGeom& g, int ndtiles_alloc, int nct_alloc, int grp0, int ngrp, hipStream_t s)
{
    const int mode = fic_q_mode(g.B, g.n_iso);
    const int shape16 = fic_q_shape16(g);
    if (fic_q_prep_fused(g, ndtiles_alloc, ngrp)) {
        auto pk = mode == 0 ? k_prep_q8<0> : k_prep_q8<2>;
        hipLaunchKernelGGL(pk, dim3(ndtiles_alloc + ngrp, g.planes), dim3(256), 0, s, (const uint8_t*)b.gray, b.scaled, b.pool_pix, b.pool_st,
                           b.pool_var, b.pool_s64, (v4i*)poolQ, (uint32_t*)dflat, b.rng_st, (float*)rngE, b.key, (uint32_t*)theta_g,
                           (v4i*)rngQ, (uint32_t*)rngC, g, ndtiles_alloc, nct_alloc, grp0, shape16);
        FIC_LAUNCH_CHECK();
        return 0;
    }
    auto pool = g.B == 4 ? k_pool_q<4> : (g.B == 8 ? k_pool_q<8> : k_pool_q<16>);
    hipLaunchKernelGGL(pool, dim3(ndtiles_alloc, g.planes), dim3(256), 0, s, (const uint8_t*)b.scaled, b.pool_pix,
                       b.pool_st, b.pool_var, b.pool_s64, (v4i*)poolQ, (uint32_t*)dflat, g, ndtiles_alloc, mode == 2 ? 1 : 0, shape16);
    FIC_LAUNCH_CHECK();
    // LDS staging of the byte copies when they fit beside the 16.6 KB of raw blocks (not at B = 16 with 8 isometries: 66 KB)
    const size_t stage_bytes = (size_t)64 * fic_q_cols_per_range(g.B, g.n_iso) * (g.n + 8);
    const int staged = stage_bytes <= 40 * 1024 ? 1 : 0;
    if (g.B == 8 && mode != 1) {                               // the benchmark's shapes: copies by register permutes, 16-byte stores
        auto rk = mode == 0 ? k_range_q8<0> : k_range_q8<2>;
        hipLaunchKernelGGL(rk, dim3(ngrp, g.planes), dim3(256), 0, s, (const uint8_t*)b.gray, b.rng_st, (float*)rngE, b.key,
                           (uint32_t*)theta_g, (v4i*)rngQ, (uint32_t*)rngC, g, nct_alloc, grp0, shape16);
    } else {
        hipLaunchKernelGGL(k_range_q, dim3(ngrp, g.planes), dim3(256), staged ? stage_bytes : 0, s, (const uint8_t*)b.gray, b.rng_st,
                           (float*)rngE, b.key, (uint32_t*)theta_g, (v4i*)rngQ, (uint32_t*)rngC, g, nct_alloc, grp0, mode, staged, shape16);
    }
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_sweep_q(const FicBuffers& b, const void* poolQ, const void* dflat, const void* rngQ, const void* rngC, const void* rngE,
                       void* theta_g, const FicGeom& g, int ct_begin, int ct_end, int ndtiles, int ndtiles_alloc,
                       int nct_alloc, int tiles_per_chunk, int nchunks, hipStream_t s, unsigned long long* stats, int dbg_noflag,
                       const FicOutputs* fin_out, unsigned int* fin_count, int r_begin, int r_count)
{
    QArgs A;
    A.stats = stats;
    A.dbg_noflag = dbg_noflag;
    // fused finalise: the sweep's last workgroup per (plane, column group) does k_finalize's work for that group
    A.fin_count = fin_out ? fin_count : nullptr;
    if (fin_out) A.out = *fin_out; else memset(&A.out, 0, sizeof(A.out));
    A.geom = g; A.gray = b.gray; A.pool_var = b.pool_var; A.r_begin = r_begin; A.r_end = r_begin + r_count;
    A.poolQ = (const v4i*)poolQ; A.dflat = (const uint32_t*)dflat; A.pool_pix = b.pool_pix; A.pool_st = b.pool_st;
    A.pool_s64 = b.pool_s64; A.rngQ = (const v4i*)rngQ; A.rngC = (const uint32_t*)rngC; A.rng_st = b.rng_st;
    A.rngE = (const float*)rngE; A.key = b.key; A.theta_g = (uint32_t*)theta_g;
    A.Nd = g.Nd; A.Nd_pad = g.Nd_pad; A.Nr = g.Nr; A.Nr_pad = g.Nr_pad; A.n = g.n; A.lgn = g.lgn; A.W = g.W; A.H = g.H; A.Rw = g.Rw;
    const int unroll = fic_q_unroll(g.B, g.n_iso);
    A.ndtiles = ndtiles; A.ndtiles_loop = (ndtiles + unroll - 1) / unroll * unroll; A.ndtiles_alloc = ndtiles_alloc; A.nct_alloc = nct_alloc;
    A.ct_begin = ct_begin; A.ct_end = ct_end; A.tiles_per_chunk = tiles_per_chunk; A.nchunks = nchunks;
    const int CT = fic_q_ct(g.B);
    A.nctg = (ct_end - ct_begin + CT - 1) / CT;
    A.planes = g.planes;
    A.rgb_rng = nullptr; A.rgb_pool = nullptr; A.rgb_dst = nullptr;
    // the loop reads up to ndtiles_loop + prefetch distance tiles; chunks must be whole unrolled iterations
    if (ct_begin + A.nctg * CT > nct_alloc || g.Nd >= (1 << 24) || A.ndtiles_loop + unroll > ndtiles_alloc || tiles_per_chunk % unroll)
        return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)(nchunks * g.planes) * (unsigned)A.nctg), block(64 * FIC_Q_WPG);
    const int mode = fic_q_mode(g.B, g.n_iso);
    if (!q_launch(g.B * g.B / 16, mode, fic_q_multi_kind(nchunks, tiles_per_chunk), grid, block, s, A, fic_q_shape16(g))) return (int)hipErrorInvalidValue;
    FIC_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// Joint RGB (encodeRGB FC:171-219) through the same sweep: k_sweep_q<NK, 3>.
// getErrorVarianceCovarianceRGB (FC:760-808) has the shape of the grey error with other constants: with
//   greyR_i = sum over the channels of (range pixel - channel mean), greyD_i likewise for the domain block,
//   kovarianz = sum greyR_i greyD_i,  varianzRange = sum greyR_i (= vR, an integer in [0, 3(n-1)]),  varianzDomain = sum greyD_i (= vD)
//   r = kovarianz / (vR * vD)  (0 when vR == 0 or vD == 0),   error = vR^2 (1 - r^2)
// i.e. the grey formulas with rem := vR and q := kovarianz / vD.  So the domain operand is A[d][i] = f16(greyD_i / vD) (zero for
// vD == 0), the range operand B[i][c] = f16(greyR_i) (integers in [-765, 765]: exact), acc ~ q, and theta / the queue / the
// any-order rules are those of the grey sweep (L >= 0.26 vR of the range block; qst[].rem = vR).  Differences:
//  * ||A_d|| is not 1: |acc - q| <= 2^-11 * 1.07 * ||greyR|| * ||A_d||, bounded per range with Amax = max_d ||A_d|| (the blocks
//    that can win have small vD and large norm, so the bound is tight where it matters);
//  * Java accumulates kovarianz in f32 in the order i = 0..n-1 and the sums exceed 2^24: the value the reference compares is
//    kov_J with |kov_J - kovarianz| <= n 2^-24 sum|greyR_i greyD_i|, i.e. another 4e-6 * ||greyR|| * Amax on q -- inside the
//    slack of the 7.0e-4 coefficient (2^-11 * 1.07 = 5.2e-4); the exact evaluation of a flagged pair reproduces that order;
//  * r is a correctly rounded f32 quotient (FC:800) instead of an f64 quotient cast to f32: still a monotone function of
//    |kov_J / vD| for a fixed range, which is all the pruning argument of the header needs.
// Inputs are what k_pool_rgb / k_range_rgb (fic_rgb.hip) already produce: R+G+B per pool pixel, greyR_i, the statistics.
// =============================================================================================
__global__ __launch_bounds__(256) void k_pool_qrgb(const uint16_t* __restrict__ pool_sum, const FicRgbDomStat* __restrict__ dst,
                                                   v4i* __restrict__ poolQ, uint32_t* __restrict__ dflat,
                                                   uint32_t* __restrict__ amax, FicGeom g)
{
    __shared__ uint32_t s_norm;
    const int n = g.n, NK = n / 16;
    const int dtile = blockIdx.x;
    if (threadIdx.x == 0) s_norm = 0u;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int d = dtile * 32 + threadIdx.x;
        float norm = 0.0f;
        if (d < g.Nd) {
            const FicRgbDomStat ds = dst[d];
            if (ds.vD != 0) {
                const uint16_t* ps = pool_sum + (size_t)d * n;
                long long s2 = 0;
                for (int i = 0; i < n; i++) {
                    const int gd = (int)ps[i] - ds.msum;
                    s2 += gd * gd;
                }
                // ||A_d|| rounded up: the factor covers sqrt, the division and the f16 rounding of the elements
                norm = __fmul_rn(__fdiv_rn(__fsqrt_rn((float)s2), (float)ds.vD), 1.001f);
            }
        }
        atomicMax(&s_norm, __float_as_uint(norm));          // non-negative floats order like their bit patterns
    }
    for (int t = threadIdx.x; t < NK * 64; t += 256) {
        const int lane = t & 63, m = t >> 6;
        const int d = dtile * 32 + (lane & 31), h = lane >> 5;
        v4i v = {0, 0, 0, 0};
        if (d < g.Nd) {
            const FicRgbDomStat ds = dst[d];
            if (ds.vD != 0) {
                const uint16_t* p = pool_sum + (size_t)d * n + 16 * m + 8 * h;
                const float vd = (float)ds.vD;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    v[u] = f16_pair(__fdiv_rn((float)((int)p[2 * u] - ds.msum), vd), __fdiv_rn((float)((int)p[2 * u + 1] - ds.msum), vd));
            }
        }
        poolQ[(size_t)dtile * NK * 64 + t] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        dflat[dtile] = s_norm ? 0u : 1u;
        if (s_norm) atomicMax(amax, s_norm);
    }
}

// one workgroup per 64 range blocks (= 2 column tiles): B fragments, rem := varianzRange, E_r, key / theta reset
__global__ __launch_bounds__(256) void k_range_qrgb(const int16_t* __restrict__ rng_t, const FicRgbRngStat* __restrict__ rst,
                                                    FicRngStat* __restrict__ qst, float* __restrict__ rngE,
                                                    unsigned long long* __restrict__ key, uint32_t* __restrict__ theta_g,
                                                    v4i* __restrict__ rngQ, const uint32_t* __restrict__ amax, FicGeom g,
                                                    int nct_alloc)
{
    const int n = g.n, NK = n / 16;
    const int j0 = blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const int j = j0 + threadIdx.x;
        if (j < g.Nr) {
            const int16_t* rt = rng_t + (size_t)j * n;
            long long s2 = 0;
            for (int i = 0; i < n; i++) s2 += (int)rt[i] * (int)rt[i];
            const float am = __uint_as_float(*amax);
            FicRngStat st;
            st.rM = 0;
            st.rem = rst[j].vR;
            qst[j] = st;
            rngE[j] = __fadd_rn(__fmul_rn(__fmul_rn(__fmul_rn(__fsqrt_rn((float)s2), 1.0001f), am), FIC_Q_ECOEF),
                                __fmul_rn(FIC_Q_EABS, __fadd_rn(1.0f, am)));
            key[j] = FIC_KEY_NONE;
            theta_g[j] = FIC_Q_THG_NONE;
        }
    }
    for (int t = threadIdx.x; t < 2 * NK * 64; t += 256) {
        const int lane = t & 63, m = (t >> 6) % NK, ctl = (t >> 6) / NK;
        const int j = j0 + ctl * 32 + (lane & 31), h = lane >> 5;
        v4i v = {0, 0, 0, 0};
        if (j < g.Nr) {
            const int16_t* p = rng_t + (size_t)j * n + 16 * m + 8 * h;
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = f16_pair((float)p[2 * u], (float)p[2 * u + 1]);
        }
        const int ct = j0 / 32 + ctl;
        if (ct < nct_alloc) rngQ[((size_t)ct * NK + m) * 64 + lane] = v;
    }
}

// prep + sweep of one RGB image (planes = 1 geometry); q buffers as in QArgs, sized by the caller (fic_capi_rgb.cpp)
int fic_launch_rgbq(const uint16_t* pool_sum, const FicRgbDomStat* pool_st, const int16_t* rng_t, const FicRgbRngStat* rng_st,
                    unsigned long long* key, void* poolQ, void* dflat, void* rngQ, void* qst, void* rngE, void* theta_g, void* amax,
                    const FicGeom& g, int ndtiles, int ndtiles_alloc, int nct_alloc, int tiles_per_chunk, int nchunks, hipStream_t s)
{
    if (hipMemsetAsync(amax, 0, sizeof(uint32_t), s) != hipSuccess) return (int)hipErrorUnknown;
    hipLaunchKernelGGL(k_pool_qrgb, dim3(ndtiles_alloc), dim3(256), 0, s, pool_sum, pool_st, (v4i*)poolQ, (uint32_t*)dflat,
                       (uint32_t*)amax, g);
    FIC_LAUNCH_CHECK();
    const int CT = fic_q_ct(g.B);
    const int nct = (g.Nr + 31) / 32;
    hipLaunchKernelGGL(k_range_qrgb, dim3(nct_alloc / 2), dim3(256), 0, s, rng_t, rng_st, (FicRngStat*)qst, (float*)rngE, key,
                       (uint32_t*)theta_g, (v4i*)rngQ, (const uint32_t*)amax, g, nct_alloc);
    FIC_LAUNCH_CHECK();
    QArgs A;
    A.stats = nullptr;
    A.dbg_noflag = 0;
    A.fin_count = nullptr; memset(&A.out, 0, sizeof(A.out)); A.geom = g; A.gray = nullptr; A.pool_var = nullptr; A.r_begin = 0; A.r_end = 0;
    A.poolQ = (const v4i*)poolQ; A.dflat = (const uint32_t*)dflat; A.pool_pix = nullptr; A.pool_st = nullptr; A.pool_s64 = nullptr;
    A.rngQ = (const v4i*)rngQ; A.rngC = nullptr; A.rng_st = (const FicRngStat*)qst; A.rngE = (const float*)rngE; A.key = key;
    A.theta_g = (uint32_t*)theta_g;
    A.rgb_rng = rng_t; A.rgb_pool = pool_sum; A.rgb_dst = pool_st;
    A.Nd = g.Nd; A.Nd_pad = g.Nd; A.Nr = g.Nr; A.Nr_pad = g.Nr; A.n = g.n; A.lgn = g.lgn; A.W = g.W; A.H = g.H; A.Rw = g.Rw;
    const int unroll = FIC_Q_UNROLL;
    A.ndtiles = ndtiles; A.ndtiles_loop = (ndtiles + unroll - 1) / unroll * unroll; A.ndtiles_alloc = ndtiles_alloc; A.nct_alloc = nct_alloc;
    A.ct_begin = 0; A.ct_end = nct; A.tiles_per_chunk = tiles_per_chunk; A.nchunks = nchunks;
    A.nctg = (nct + CT - 1) / CT;
    A.planes = 1;
    if (A.nctg * CT > nct_alloc || (nct_alloc & 1) || g.Nd >= (1 << 24) || A.ndtiles_loop + unroll > ndtiles_alloc || tiles_per_chunk % unroll)
        return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)nchunks * (unsigned)A.nctg), block(64 * FIC_Q_WPG);
    if (!q_launch(g.B * g.B / 16, 3, fic_q_multi_kind(nchunks, tiles_per_chunk), grid, block, s, A)) return (int)hipErrorInvalidValue;
    FIC_LAUNCH_CHECK();
    return 0;
}
