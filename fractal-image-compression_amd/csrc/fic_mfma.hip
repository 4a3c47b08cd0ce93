// fic_mfma.hip -- OPT-IN matrix-core sweeps with i8 operands: k_sweep_mfma<NM> (8 iso) and k_sweep_mfma1<NM> (1 iso), B = 4/8/16.
// These were the round's first matrix-core kernels.  "sweep" = 3 now runs the bf16-operand kernels of fic_bf16.hip
// wherever those are faster (B = 4/8, and B = 16 with 8 isometries) and k_sweep_mfma1<8> at B = 16 with 1 isometry;
// "sweep" = 4 runs the kernels of this file at every block size (kept for comparison and tested to the same bar).
// gfx950 (MI355X / CDNA4) only, wave64.  Compile with -ffp-contract=off: every float expression must round once
// per operation exactly like the Java reference (FractalCompression.java = FC, Domainblock.java = DB).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "fic_device.h"
#include "fic_launch.h"
#include "fic_devfn.h"

// ---------------------------------------------------------------------------------------------
// k_sweep_mfma<NM> : OPT-IN matrix-core variant of the full-pool sweep, i8 operands ("sweep" = 4; n_iso = 8; B = 4/8/16).
//
// north_star asks for a VALU-only sweep and k_sweep_fast is that kernel (and the default).  It
// is pinned at the v_dot4 issue ceiling; the inner products of all (range copy,
// domain) pairs are, however, an exact integer GEMM, and this kernel hands them to
// v_mfma_i32_32x32x32_i8.  Everything downstream -- the conservative prune test, the exact
// f64 epilogue, the (error, candidate) lexicographic minimum -- is the same arithmetic as
// k_sweep_fast, so the result is the same bits (tests compare all three sweeps).
//
//   u8 -> i8:   x' = x - 128 = x ^ 0x80 (both operands, done once in the prep kernels), and
//               sum r*d = sum r'*d' + 128*Sr + 128*Sd - 16384*n   exactly.
//   rows (A)  = range copies: a 32-row tile = 4 range blocks x 8 isometry copies, row = 8q + k.
//               The MFMA result layout (col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)) then
//               gives register e of EVERY lane the same range q = e>>2 -- so per-range constants
//               are wave-uniform -- and isometry k = (e&3) + 4*(lane>>5).
//   cols (B)  = 32 consecutive domain blocks, streamed; lane&31 = the lane's domain, whose
//               {sum, sqrt(var)} sit in two VGPRs.
//   K = n pixels = NM MFMA steps of 32 (B = 4: one step, upper half zero); lane (r, h) supplies bytes
//   [32m+16h, 32m+16h+16) of its row/col for step m -- the same map on both operands, so any k
//   order is consistent.
//   A workgroup (4 waves) keeps 32 KiB of A fragments in LDS (32/16/4 row tiles for B = 4/8/16); its
//   waves split the row tiles and all sweep the same pool chunk (their B-tile loads hit in L1).
//   tau (the prune threshold) of a range belongs to the wave that owns its row tile; it lives in
//   LDS and is raised with ds_max_u32 by the rare exact path; winners go straight to the global
//   key with the same 64-bit atomicMin as k_sweep_fast.  The first domain tile of a chunk is
//   always evaluated exactly (it initialises tau).
// ---------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define FIC_TAU_ALL 8192.0f                // > |cov| / sqrt(var) for any pair (|cov| <= sqrt(n)*255*sqrt(var) <= 4080*sqrt(var))
// NM = MFMA steps of K = 32 per block (B = 4: 1 with the upper half of K zero, B = 8: 2, B = 16: 8).
// Row tiles (4 range blocks x 8 isometry copies) per workgroup: always 32 KiB of A fragments.
__host__ __device__ constexpr int mfma8_rt(int NM) { return NM == 1 ? 32 : NM == 2 ? 16 : 4; }

// pool -> B fragments: poolB[plane][dtile][m][lane] = bytes [32m+16h, +16) of domain block dtile*32 + (lane&31),
// i8-shifted (x ^ 0x80); blocks beyond N_d read as pixel 0 (so their covariance is exactly 0), bytes beyond n as i8 0.
__global__ __launch_bounds__(256) void k_pool_mfma(const uint8_t* __restrict__ pool_pix, v4i* __restrict__ poolB,
                                                   FicGeom g, int ndtiles_alloc, int NM)
{
    int i = blockIdx.x * 256 + threadIdx.x;            // (dtile, m, lane)
    int plane = blockIdx.y;
    if (i >= ndtiles_alloc * NM * 64) return;
    int lane = i & 63, m = (i >> 6) % NM, dtile = (i >> 6) / NM;
    int d = dtile * 32 + (lane & 31), h = lane >> 5;
    int off = 32 * m + 16 * h;
    v4i v = {0, 0, 0, 0};
    if (off < g.n) {
        v = (v4i){(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
        if (d < g.Nd) v ^= *(const v4i*)(pool_pix + ((size_t)plane * g.Nd_pad + d) * g.n + off);
    }
    poolB[(size_t)plane * ndtiles_alloc * NM * 64 + i] = v;
}

// ranges -> A fragments (row = 8q + k of a row tile: range block 4t+q, isometry copy k) + per-range constants
// {K = 128*Sr - 16384*n, A = 128 - rM, negR = -rem}, 16 ints per row tile
__global__ __launch_bounds__(256) void k_range_mfma(const uint32_t* __restrict__ rng_pix,
                                                    const FicRngStat* __restrict__ rng_st, v4i* __restrict__ rngA,
                                                    int* __restrict__ rconst, FicGeom g, int NM, int RT, int ngroups)
{
    int i = blockIdx.x * 256 + threadIdx.x;            // (group, t, m, lane)
    int plane = blockIdx.y;
    int total = ngroups * RT * NM * 64;
    if (i >= total) return;
    int lane = i & 63, m = (i >> 6) % NM, t = ((i >> 6) / NM) % RT, group = ((i >> 6) / NM) / RT;
    int r = lane & 31, h = lane >> 5, q = r >> 3, k = r & 7;
    int j = (group * RT + t) * 4 + q;
    int off = 32 * m + 16 * h;
    const uint32_t* rp = rng_pix + (size_t)plane * g.Nr_pad * g.n_iso * g.DW;
    v4i v = {0, 0, 0, 0};
    if (off < g.n) {
#pragma unroll
        for (int w = 0; w < 4; w++)
            v[w] = (int)((j < g.Nr_pad ? rp[rng_word_index(g, j, k, off / 4 + w)] : 0u) ^ 0x80808080u);
    }
    rngA[(size_t)plane * total + i] = v;
    if (m == 0 && lane < 16) {                          // 16 ints per (group, t)
        int qq = lane & 3, what = lane >> 2;
        int jj = (group * RT + t) * 4 + qq;
        FicRngStat st = {0, 0};
        if (jj < g.Nr_pad) st = rng_st[(size_t)plane * g.Nr_pad + jj];
        int Sr = st.rM * g.n + st.rem;
        int val = what == 0 ? 128 * Sr - 16384 * g.n : what == 1 ? 128 - st.rM : what == 2 ? -st.rem : 0;
        rconst[(((size_t)plane * ngroups + group) * RT + t) * 16 + lane] = val;
    }
}

struct MfmaArgs {
    const v4i* poolB;
    const FicDomStat* pool_st;
    const double* pool_s64;
    const v4i* rngA;
    const int* rconst;
    unsigned long long* key;
    int Nd, Nd_pad, Nr, Nr_pad, lgn;
    int ndtiles, ndtiles_alloc, ngroups;
    int group0, ngroups_launch;      // first range group of this shard, groups in this launch
    int tiles_per_chunk, nchunks, planes;
};

template <int NM>
__global__ __launch_bounds__(256) void k_sweep_mfma(MfmaArgs A)
{
    constexpr int RT = mfma8_rt(NM);
    __shared__ v4i sA[RT * NM * 64];                   // 32 KiB: A fragments of RT*4 range blocks x 8 copies
    __shared__ __attribute__((aligned(16))) uint32_t sTau[RT * 4];   // per range: prune threshold (f32 bits, >= 0);
                                                                     // range 4t+q belongs to the wave that owns row tile t
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int combo_, gx_;                                   // XCD-aware 1-D grid: combo = (pool chunk, plane), x = range group
    xcd_decode(blockIdx.x, A.nchunks * A.planes, A.ngroups_launch, combo_, gx_);
    const int group = A.group0 + gx_;
    const int plane = combo_ / A.nchunks;
    const int chunk = combo_ % A.nchunks;              // all 4 waves sweep the same pool chunk (B tiles hit in L1) ...
    const int t_first = wave * (RT / 4);               // ... and split the row tiles
    const size_t gbase = (size_t)plane * A.ngroups + group;
    const AS4 int* rc_base = (const AS4 int*)A.rconst + gbase * RT * 16;

    for (int i = threadIdx.x; i < RT * NM * 64; i += 256) sA[i] = A.rngA[gbase * RT * NM * 64 + i];
    // ranges with rem == 0: error 0 for every block (FC:677) -> nothing after the first tile can win
    if (threadIdx.x < RT * 4)
        sTau[threadIdx.x] = (A.rconst[(gbase * RT + (threadIdx.x >> 2)) * 16 + 8 + (threadIdx.x & 3)] == 0)
                                ? __float_as_uint(FIC_TAU_ALL) : 0u;
    __syncthreads();
    const int dt0 = chunk * A.tiles_per_chunk;
    int dt1 = dt0 + A.tiles_per_chunk;
    if (dt1 > A.ndtiles) dt1 = A.ndtiles;
    if (dt0 >= dt1) return;

    const int jcol = lane & 31, half = lane >> 5;
    const v4i* pb = A.poolB + (size_t)plane * A.ndtiles_alloc * NM * 64;
    const FicDomStat* pst = A.pool_st + (size_t)plane * A.Nd_pad;
    const double* p64 = A.pool_s64 + (size_t)plane * A.Nd_pad;
    const int jbase = group * RT * 4;                   // first range block of the workgroup
    unsigned long long* keyp = A.key + (size_t)plane * A.Nr_pad;
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    v4i b[NM], nb[NM];
#pragma unroll
    for (int m = 0; m < NM; m++) b[m] = pb[((size_t)dt0 * NM + m) * 64 + lane];
    int dcur = dt0 * 32 + jcol;
    FicDomStat st = pst[dcur < A.Nd_pad ? dcur : 0];
    for (int dt = dt0; dt < dt1; dt++) {
        // prefetch the next domain tile (the fragment store has one spare tile)
#pragma unroll
        for (int m = 0; m < NM; m++) nb[m] = pb[((size_t)(dt + 1) * NM + m) * 64 + lane];
        int dnext = (dt + 1) * 32 + jcol;
        FicDomStat nst = pst[dnext < A.Nd_pad ? dnext : 0];
        const int d = dt * 32 + jcol;
        const bool valid = d < A.Nd;
        const int Sd = valid ? (int)st.sum : 0;
        const int dM = Sd >> A.lgn;
        const float s32 = valid ? st.s32 : 0.0f;
        const bool force = (dt == dt0);                 // wave-uniform

        for (int t = t_first; t < t_first + RT / 4; t++) {
            const uint4 tau4 = *(const uint4*)&sTau[t * 4];
            const AS4 int* rc = rc_base + t * 16;       // wave-uniform -> scalar loads
            v16i acc = zero;
#pragma unroll
            for (int m = 0; m < NM; m++)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(sA[(t * NM + m) * 64 + lane], b[m], acc, 0, 0, 0);

            // Per range q: cov_e = acc[e] + Kq + Aq*sum(d) - rem*dM (exact, 24-bit operands); the pair needs the
            // exact epilogue iff |cov_e| > li = floor(tau*s32)  <=>  (unsigned)(cov_e + li) > 2*li.
            uint32_t offq[4], spanq[4];
            bool anyq[4];
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int Kq = rc[q], Aq = rc[4 + q], nRq = rc[8 + q];
                const float tau = __uint_as_float(q == 0 ? tau4.x : q == 1 ? tau4.y : q == 2 ? tau4.z : tau4.w);
                int lin;
                asm("v_mul_i32_i24 %0, %1, %2" : "=v"(lin) : "s"(nRq), "v"(dM));
                asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(lin) : "s"(Aq), "v"(Sd));
                const int li = (int)__fmul_rn(tau, s32);            // < 2^30 by construction (tau <= 8192, s32 <= 4080)
                offq[q] = (uint32_t)(lin + Kq + li);
                spanq[q] = 2u * (uint32_t)li;
                const uint32_t u0 = (uint32_t)acc[4 * q + 0] + offq[q], u1 = (uint32_t)acc[4 * q + 1] + offq[q];
                const uint32_t u2 = (uint32_t)acc[4 * q + 2] + offq[q], u3 = (uint32_t)acc[4 * q + 3] + offq[q];
                const uint32_t m01 = u0 > u1 ? u0 : u1, m23 = u2 > u3 ? u2 : u3;
                anyq[q] = (m01 > m23 ? m01 : m23) > spanq[q];
                any |= anyq[q];
            }
            if (__builtin_expect(__any(any || force), 0)) {
                const double s64 = valid ? p64[d] : 0.0;
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int q = e >> 2;
                    const int j = jbase + 4 * t + q;
                    if (!((anyq[q] || force) && valid && j < A.Nr)) continue;   // whole range untouched on this lane
                    const uint32_t u = (uint32_t)acc[e] + offq[q];
                    if (u > spanq[q] || force) {
                        const int cov = (int)(u - (spanq[q] >> 1));
                        const int rem = -rc[8 + q];
                        const float err = exact_error(cov, rem, s64);
                        const uint32_t cand = (uint32_t)d * 8u + (uint32_t)((e & 3) + 4 * half);
                        atomicMin(&keyp[j], ((unsigned long long)f32_orderable(err) << 32) | cand);
                        if (rem != 0) {
                            float lvl = (s32 == 0.0f) ? 0.0f
                                                      : __fmul_rn(__fdiv_rn(fabsf((float)cov), s32), 0.99999618530273437500f);
                            atomicMax(&sTau[4 * t + q], __float_as_uint(lvl));   // lvl >= 0: uint order == float order
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < NM; m++) b[m] = nb[m];
        st = nst;
    }
}

// ---------------------------------------------------------------------------------------------
// k_sweep_mfma1 : OPT-IN matrix-core sweep for the reference algorithm (n_iso = 1), i8 operands, B = 4 / 8 / 16
// ("sweep" = 4; "sweep" = 3 at B = 16).  Same exactness scheme as k_sweep_mfma; the mapping is transposed:
//   rows (A) = 32 consecutive domain blocks, streamed; cols (B) = 32 range blocks, fixed in LDS.
//   col = lane&31 is the lane's range block: its constants {K, A, negR} and its tau come from
//   LDS with one ds_read_b128 per column tile (tau is raised there by the exact path with
//   ds_max_u32, so the two lane halves of a column see each other's updates).
//   row = (e&3) + 8(e>>2) + 4(lane>>5) is register e's domain block: {sum, 1/sqrt(var)} of the
//   lane's 16 rows are loaded once per domain tile (stored per tile in exactly that order).
//   NM = n/32 MFMA steps of K = 32 (B = 4: one step, upper half of K zero).
// Prune test per pair (f32):  |cov| * fl(1/s32) > tau  -- same 2^-18 margin as everywhere
// (the extra 1/s32 rounding adds 2^-24); flat blocks carry w = 0 and are never flagged.
// ---------------------------------------------------------------------------------------------
struct Mfma1Args {
    const v4i* poolA;                // [plane][ndtiles_alloc][NM][64]  domain fragments (i8-shifted)
    const uint2* pool_sw;            // [plane][ndtiles_alloc][2][16]   {sum, 1/s32 bits} in accumulator-row order
    const FicDomStat* pool_st;
    const double* pool_s64;
    const v4i* rngB;                 // [plane][nctiles_alloc][NM][64]  range fragments
    const int4* rconst;              // [plane][nctiles_alloc*32]       {K, A, negR, tau0 bits}
    unsigned long long* key;
    int Nd, Nd_pad, Nr, Nr_pad, lgn;
    int ndtiles, ndtiles_alloc, nctiles_alloc;
    int ct_begin, ct_end;            // column tiles (x32 ranges) of this shard
    int nctg;                        // column-tile groups (workgroups) in this launch
    int tiles_per_chunk, nchunks, planes;
};

template <int NM>
__global__ __launch_bounds__(256) void k_sweep_mfma1(Mfma1Args A)
{
    constexpr int CT = NM == 1 ? 32 : NM == 2 ? 16 : 4;      // column tiles per workgroup (32 KiB of fragments)
    constexpr int CTW = CT / 4;                              // per wave
    __shared__ v4i sB[CT * NM * 64];
    __shared__ int4 sC[CT * 32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int combo_, gx_;                                         // XCD-aware 1-D grid: combo = (pool chunk, plane)
    xcd_decode(blockIdx.x, A.nchunks * A.planes, A.nctg, combo_, gx_);
    const int plane = combo_ / A.nchunks;
    const int ct0 = A.ct_begin + gx_ * CT;            // first column tile of the workgroup
    const size_t cbase = (size_t)plane * A.nctiles_alloc + ct0;
    for (int i = threadIdx.x; i < CT * NM * 64; i += 256) sB[i] = A.rngB[cbase * NM * 64 + i];
    for (int i = threadIdx.x; i < CT * 32; i += 256) sC[i] = A.rconst[cbase * 32 + i];
    __syncthreads();
    const int chunk = combo_ % A.nchunks;
    const int dt0 = chunk * A.tiles_per_chunk;
    int dt1 = dt0 + A.tiles_per_chunk;
    if (dt1 > A.ndtiles) dt1 = A.ndtiles;
    if (dt0 >= dt1 || ct0 + wave * CTW >= A.ct_end) return;

    const int jcol = lane & 31, half = lane >> 5;
    const v4i* pa = A.poolA + (size_t)plane * A.ndtiles_alloc * NM * 64;
    const uint4* psw = (const uint4*)(A.pool_sw + (size_t)plane * A.ndtiles_alloc * 32);
    const FicDomStat* pst = A.pool_st + (size_t)plane * A.Nd_pad;
    const double* p64 = A.pool_s64 + (size_t)plane * A.Nd_pad;
    unsigned long long* keyp = A.key + (size_t)plane * A.Nr_pad;
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    for (int dt = dt0; dt < dt1; dt++) {
        v4i a[NM];
#pragma unroll
        for (int m = 0; m < NM; m++) a[m] = pa[((size_t)dt * NM + m) * 64 + lane];
        int Sd[16], dM[16];
        float wd[16];
#pragma unroll
        for (int v = 0; v < 8; v++) {                        // 16 x {sum, w} of this half's accumulator rows
            const uint4 s4 = psw[((size_t)dt * 2 + half) * 8 + v];
            Sd[2 * v] = (int)s4.x;     wd[2 * v] = __uint_as_float(s4.y);
            Sd[2 * v + 1] = (int)s4.z; wd[2 * v + 1] = __uint_as_float(s4.w);
        }
#pragma unroll
        for (int e = 0; e < 16; e++) dM[e] = Sd[e] >> A.lgn;
        const bool force = (dt == dt0);                      // wave-uniform

        for (int ci = 0; ci < CTW; ci++) {
            const int ctl = wave * CTW + ci;                 // column tile inside the workgroup
            if (ct0 + ctl >= A.ct_end) break;                // wave-uniform
            v16i acc = zero;
#pragma unroll
            for (int m = 0; m < NM; m++)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], sB[(ctl * NM + m) * 64 + lane], acc, 0, 0, 0);
            const int4 c = sC[ctl * 32 + jcol];              // {K, A, negR, tau}
            const float tau = __int_as_float(c.w);
            int cov[16];
            bool any = false;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                int lin;
                asm("v_mul_i32_i24 %0, %1, %2" : "=v"(lin) : "v"(c.z), "v"(dM[e]));
                asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(lin) : "v"(c.y), "v"(Sd[e]));
                cov[e] = acc[e] + lin + c.x;                 // kovarianz, exact
                any |= __fmul_rn(fabsf((float)cov[e]), wd[e]) > tau;
            }
            if (__builtin_expect(__any(any || force), 0)) {
                const int j = (ct0 + ctl) * 32 + jcol;       // the lane's range block
                const int rem = -c.z;
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int d = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if ((force || __fmul_rn(fabsf((float)cov[e]), wd[e]) > tau) && d < A.Nd && j < A.Nr) {
                        const float err = exact_error(cov[e], rem, p64[d]);
                        atomicMin(&keyp[j], ((unsigned long long)f32_orderable(err) << 32) | (uint32_t)d);
                        if (rem != 0) {
                            const float s32 = pst[d].s32;
                            const float lvl = (s32 == 0.0f) ? 0.0f
                                                            : __fmul_rn(__fdiv_rn(fabsf((float)cov[e]), s32), 0.99999618530273437500f);
                            atomicMax((unsigned int*)&sC[ctl * 32 + jcol].w, __float_as_uint(lvl));
                        }
                    }
                }
            }
        }
    }
}

// fragment prep for k_sweep_mfma1: domain fragments + per-row {sum, 1/s32} in accumulator-row order
__global__ __launch_bounds__(256) void k_pool_mfma1(const uint8_t* __restrict__ pool_pix, const FicDomStat* __restrict__ pool_st,
                                                    v4i* __restrict__ poolA, uint2* __restrict__ pool_sw, FicGeom g,
                                                    int ndtiles_alloc, int NM)
{
    int i = blockIdx.x * 256 + threadIdx.x;                  // (dtile, m, lane)
    int plane = blockIdx.y;
    if (i >= ndtiles_alloc * NM * 64) return;
    int lane = i & 63, m = (i >> 6) % NM, dtile = (i >> 6) / NM;
    int d = dtile * 32 + (lane & 31), h = lane >> 5;
    int off = 32 * m + 16 * h;
    v4i v = {0, 0, 0, 0};                                    // i8 zeros: padding contributes nothing
    if (d < g.Nd && off < g.n) {
        v = *(const v4i*)(pool_pix + ((size_t)plane * g.Nd_pad + d) * g.n + off);
        v ^= (v4i){(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    }
    poolA[(size_t)plane * ndtiles_alloc * NM * 64 + i] = v;
    if (m == 0 && lane < 32) {                               // 32 stats per tile: [half][e]
        int hh = lane >> 4, e = lane & 15;
        int dd = dtile * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        uint2 o = {0u, 0u};
        if (dd < g.Nd) {
            FicDomStat st = pool_st[(size_t)plane * g.Nd_pad + dd];
            o.x = st.sum;
            o.y = __float_as_uint(st.s32 == 0.0f ? 0.0f : __fdiv_rn(1.0f, st.s32));
        }
        pool_sw[((size_t)plane * ndtiles_alloc + dtile) * 32 + lane] = o;
    }
}

// range fragments + constants {K = 128*Sr - 16384*n, A = 128 - rM, negR = -rem, tau0}
__global__ __launch_bounds__(256) void k_range_mfma1(const uint32_t* __restrict__ rng_pix, const FicRngStat* __restrict__ rng_st,
                                                     v4i* __restrict__ rngB, int4* __restrict__ rconst, FicGeom g,
                                                     int nctiles_alloc, int NM)
{
    int i = blockIdx.x * 256 + threadIdx.x;                  // (ctile, m, lane)
    int plane = blockIdx.y;
    if (i >= nctiles_alloc * NM * 64) return;
    int lane = i & 63, m = (i >> 6) % NM, ctile = (i >> 6) / NM;
    int j = ctile * 32 + (lane & 31), h = lane >> 5;
    int off = 32 * m + 16 * h;
    v4i v = {0, 0, 0, 0};
    const bool in = j < g.Nr_pad;
    if (in && off < g.n) {
        const uint32_t* rp = rng_pix + (size_t)plane * g.Nr_pad * g.n_iso * g.DW;
#pragma unroll
        for (int w = 0; w < 4; w++) v[w] = (int)(rp[rng_word_index(g, j, 0, off / 4 + w)] ^ 0x80808080u);
    }
    rngB[(size_t)plane * nctiles_alloc * NM * 64 + i] = v;
    if (m == 0 && lane < 32) {
        int4 c = {0, 0, 0, 0};
        if (in) {
            FicRngStat st = rng_st[(size_t)plane * g.Nr_pad + j];
            int Sr = st.rM * g.n + st.rem;
            c.x = 128 * Sr - 16384 * g.n;
            c.y = 128 - st.rM;
            c.z = -st.rem;
            c.w = (st.rem == 0) ? __float_as_int(8192.0f) : 0;   // rem == 0: nothing after the first block can win
        }
        rconst[(size_t)plane * nctiles_alloc * 32 + ctile * 32 + lane] = c;
    }
}

// host-side launchers
// opt-in matrix-core sweep, n_iso = 8 (B = 4 / 8 / 16): fragment prep + sweep
static int mfma_nm(const FicGeom& g) { return g.n <= 32 ? 1 : g.n / 32; }
int fic_mfma8_group(int B) { return mfma8_rt(B == 4 ? 1 : B == 8 ? 2 : 8) * 4; }     // range blocks per workgroup
int fic_launch_mfma_prep_pool(const uint8_t* pool_pix, void* poolB, const FicGeom& g, int ndtiles_alloc, hipStream_t s)
{
    const int NM = mfma_nm(g);
    hipLaunchKernelGGL(k_pool_mfma, dim3((ndtiles_alloc * NM * 64 + 255) / 256, g.planes), dim3(256), 0, s, pool_pix,
                       (v4i*)poolB, g, ndtiles_alloc, NM);
    FIC_LAUNCH_CHECK();
    return 0;
}
int fic_launch_mfma_prep_range(const uint32_t* rng_pix, const FicRngStat* rng_st, void* rngA, int* rconst,
                               const FicGeom& g, int ngroups, hipStream_t s)
{
    const int NM = mfma_nm(g), RT = mfma8_rt(NM);
    int total = ngroups * RT * NM * 64;
    hipLaunchKernelGGL(k_range_mfma, dim3((total + 255) / 256, g.planes), dim3(256), 0, s, rng_pix, rng_st, (v4i*)rngA,
                       rconst, g, NM, RT, ngroups);
    FIC_LAUNCH_CHECK();
    return 0;
}
int fic_launch_sweep_mfma(const FicBuffers& b, const void* poolB, const void* rngA, const int* rconst, const FicGeom& g,
                          int ngroups, int group0, int ngroups_launch, int ndtiles, int ndtiles_alloc, int tiles_per_chunk,
                          int nchunks, hipStream_t s)
{
    MfmaArgs A;
    A.poolB = (const v4i*)poolB; A.pool_st = b.pool_st; A.pool_s64 = b.pool_s64;
    A.rngA = (const v4i*)rngA; A.rconst = rconst; A.key = b.key;
    A.Nd = g.Nd; A.Nd_pad = g.Nd_pad; A.Nr = g.Nr; A.Nr_pad = g.Nr_pad; A.lgn = g.lgn;
    A.ndtiles = ndtiles; A.ndtiles_alloc = ndtiles_alloc; A.ngroups = ngroups;
    A.group0 = group0; A.ngroups_launch = ngroups_launch; A.tiles_per_chunk = tiles_per_chunk; A.nchunks = nchunks;
    A.planes = g.planes;
    dim3 grid((unsigned)(nchunks * g.planes) * (unsigned)ngroups_launch), block(256);
    if (g.B == 4) hipLaunchKernelGGL((k_sweep_mfma<1>), grid, block, 0, s, A);
    else if (g.B == 8) hipLaunchKernelGGL((k_sweep_mfma<2>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((k_sweep_mfma<8>), grid, block, 0, s, A);
    FIC_LAUNCH_CHECK();
    return 0;
}

// opt-in matrix-core sweep, n_iso = 1 (B = 4 / 8 / 16): fragment prep + sweep
int fic_mfma1_ct(int B) { return B == 4 ? 32 : B == 8 ? 16 : 4; }    // column tiles per workgroup (k_sweep_mfma1's CT)

int fic_launch_mfma1_prep(const FicBuffers& b, void* poolA, void* pool_sw, void* rngB, void* rconst, const FicGeom& g,
                          int ndtiles_alloc, int nctiles_alloc, hipStream_t s)
{
    const int NM = g.n <= 32 ? 1 : g.n / 32;
    hipLaunchKernelGGL(k_pool_mfma1, dim3((ndtiles_alloc * NM * 64 + 255) / 256, g.planes), dim3(256), 0, s,
                       (const uint8_t*)b.pool_pix, (const FicDomStat*)b.pool_st, (v4i*)poolA, (uint2*)pool_sw, g,
                       ndtiles_alloc, NM);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_range_mfma1, dim3((nctiles_alloc * NM * 64 + 255) / 256, g.planes), dim3(256), 0, s,
                       (const uint32_t*)b.rng_pix, (const FicRngStat*)b.rng_st, (v4i*)rngB, (int4*)rconst, g,
                       nctiles_alloc, NM);
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_sweep_mfma1(const FicBuffers& b, const void* poolA, const void* pool_sw, const void* rngB,
                           const void* rconst, const FicGeom& g, int ct_begin, int ct_end, int ndtiles, int ndtiles_alloc,
                           int nctiles_alloc, int tiles_per_chunk, int nchunks, hipStream_t s)
{
    Mfma1Args A;
    A.poolA = (const v4i*)poolA; A.pool_sw = (const uint2*)pool_sw; A.pool_st = b.pool_st; A.pool_s64 = b.pool_s64;
    A.rngB = (const v4i*)rngB; A.rconst = (const int4*)rconst; A.key = b.key;
    A.Nd = g.Nd; A.Nd_pad = g.Nd_pad; A.Nr = g.Nr; A.Nr_pad = g.Nr_pad; A.lgn = g.lgn;
    A.ndtiles = ndtiles; A.ndtiles_alloc = ndtiles_alloc; A.nctiles_alloc = nctiles_alloc;
    A.ct_begin = ct_begin; A.ct_end = ct_end; A.tiles_per_chunk = tiles_per_chunk; A.nchunks = nchunks;
    const int CT = fic_mfma1_ct(g.B);
    A.nctg = (ct_end - ct_begin + CT - 1) / CT;
    A.planes = g.planes;
    dim3 grid((unsigned)(nchunks * g.planes) * (unsigned)A.nctg), block(256);
    if (g.B == 4) hipLaunchKernelGGL((k_sweep_mfma1<1>), grid, block, 0, s, A);
    else if (g.B == 8) hipLaunchKernelGGL((k_sweep_mfma1<2>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((k_sweep_mfma1<8>), grid, block, 0, s, A);
    FIC_LAUNCH_CHECK();
    return 0;
}
