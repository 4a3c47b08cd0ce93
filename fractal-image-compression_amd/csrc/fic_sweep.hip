// fic_sweep.hip -- the pool sweeps: k_sweep_generic (any window, wave = range block) and k_sweep_fast (full pool, DEFAULT).
// gfx950 (MI355X / CDNA4) only, wave64.  Compile with -ffp-contract=off: every float expression must round once
// per operation exactly like the Java reference (FractalCompression.java = FC, Domainblock.java = DB).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "fic_device.h"
#include "fic_launch.h"
#include "fic_devfn.h"

// ---------------------------------------------------------------------------------------------
// k_sweep_generic : exact evaluation of every candidate of a window, one wave per range block.
// Lanes stride over the wK*wK*n_iso candidates (window-major, isometry-minor), keep a running
// (error, candidate) minimum and the wave takes the lexicographic minimum with xor-shuffles:
// identical to the strict '<' scan in ascending order of FC:619-632.
// Serves every wK (the GUI's 2/4/8/16 windows) and is the in-GPU cross-check of k_sweep_fast.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sweep_generic(const uint32_t* __restrict__ pool_pix,
                                                       const FicDomStat* __restrict__ pool_st,
                                                       const double* __restrict__ pool_s64,
                                                       const uint32_t* __restrict__ rng_pix,
                                                       const FicRngStat* __restrict__ rng_st,
                                                       unsigned long long* __restrict__ key, FicGeom g, int r_begin,
                                                       int r_count)
{
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int jr = blockIdx.x * 4 + wave;
    int plane = blockIdx.y;
    if (jr >= r_count) return;
    int j = r_begin + jr;
    const uint32_t* pp = pool_pix + (size_t)plane * g.Nd_pad * g.DW;
    const FicDomStat* ps = pool_st + (size_t)plane * g.Nd_pad;
    const double* p64 = pool_s64 + (size_t)plane * g.Nd_pad;
    const uint32_t* rp = rng_pix + (size_t)plane * g.Nr_pad * g.n_iso * g.DW;
    FicRngStat rs = rng_st[(size_t)plane * g.Nr_pad + j];
    int ncand = g.wK * g.wK * g.n_iso;
    unsigned long long best = FIC_KEY_NONE;
    for (int c = lane; c < ncand; c += 64) {
        int wloc = c / g.n_iso, k = c % g.n_iso;
        int gi = window_to_global(g, j, wloc);
        FicDomStat ds = ps[gi];
        int dM = (int)(ds.sum >> g.lgn);
        uint32_t acc = 0;
        for (int dw = 0; dw < g.DW; dw++)
            acc = __builtin_amdgcn_udot4(rp[rng_word_index(g, j, k, dw)], pp[(size_t)gi * g.DW + dw], acc, false);
        // kovarianz = sum (r-rM)(d-dM) = sum r*d - rM*sum(d) - dM*rem   (exact integers)
        int cov = (int)acc - rs.rM * (int)ds.sum - dM * rs.rem;
        float e = exact_error(cov, rs.rem, p64[gi]);
        unsigned long long kk = ((unsigned long long)f32_orderable(e) << 32) | (uint32_t)c;
        best = kk < best ? kk : best;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(best, off, 64);
        best = o < best ? o : best;
    }
    if (lane == 0) key[(size_t)plane * g.Nr_pad + j] = best;
}

// ---------------------------------------------------------------------------------------------
// k_sweep_fast : full-pool search (wK == Dw == Dh).
//
// Mapping (MI355X-first, not the CPU loop nest):
//   * lane  = range block.  Its n pixels (and, for n_iso = 8, the 8 isometry copies) stay in
//     VGPRs for the whole sweep: NR ranges x NC copies x DW dwords per lane.
//   * the domain pool is the wave-uniform operand: each wave streams it through the SCALAR
//     path (s_load_dwordx16 -> SGPRs) and feeds it straight into v_dot4_u32_u8 as the SGPR
//     source, double-buffered so the next block's load is in flight during the current dot4s.
//     Pool bytes are read once per wave (64*NR ranges): the LDS/register reuse the roofline
//     analysis in DESIGN.md calls T.
//   * every lane sees candidates in ascending index order, so Java's strict '<' tie rule
//     (FC:627) is a per-lane running minimum: no cross-lane reduction inside a chunk; chunks
//     and isometry groups combine with one 64-bit atomicMin on (error, candidate).
//
// Exactness: cov is an exact integer.  The Java error (FC:677-683) is a non-increasing
// function of |r| = |cov| / (rem * sqrt(var)), so a candidate can only win if its |cov|/sqrt(var)
// is not below that of the best evaluated so far.  Per pair we test  |cov| <= tau * s32  in f32
// (tau carries a 2^-18 safety margin against <= 2^-22 of accumulated rounding);
// only candidates failing the test run the exact f64 epilogue.  tau starts as NaN so the first
// candidate of a chunk is always evaluated; lanes with rem == 0 (error 0 for every domain,
// FC:677) pin tau to FLT_MAX after it.
// ---------------------------------------------------------------------------------------------
struct SweepArgs {
    const uint32_t* pool_pix;
    const FicDomStat* pool_st;
    const double* pool_s64;
    const uint32_t* rng_pix;
    const FicRngStat* rng_st;
    unsigned long long* key;
    int Nd, Nd_pad, Nr, Nr_pad, n_iso, lgn;
    int tile0, ntiles;         // tiles [tile0, tile0+ntiles) of 64*NR ranges
    int chunk_len, nchunks;    // domain chunk length (multiple of 2), number of chunks
    int planes;
};

template <int DW, int NR, int NC>
__global__ __launch_bounds__(256) void k_sweep_fast(SweepArgs A)
{
    constexpr int SEG = DW < 16 ? DW : 16;     // dwords per scalar load
    constexpr int NSEG = DW / SEG;             // scalar loads per domain block (1, or 4 for B = 16)
    typedef uint32_t segv __attribute__((ext_vector_type(SEG)));

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware 1-D grid (xcd_decode, fic_devfn.h): combo = (pool chunk x isometry group, plane), x = range-tile group
    const int gy_ = A.nchunks * (A.n_iso / NC);
    int combo_, bx_;
    xcd_decode(blockIdx.x, gy_ * A.planes, (A.ntiles + 3) / 4, combo_, bx_);
    const int by_ = combo_ % gy_, bz_ = combo_ / gy_;
    const int tl = bx_ * 4 + wave;
    if (tl >= A.ntiles) return;                        // wave-uniform
    const int tile = A.tile0 + tl;
    const int ngroups = A.n_iso / NC;
    const int chunk = by_ / ngroups;
    const int kbase = (by_ % ngroups) * NC;
    const int plane = bz_;

    const int d0 = chunk * A.chunk_len;
    int d1 = d0 + A.chunk_len;
    if (d1 > A.Nd) d1 = A.Nd;
    if (d0 >= d1) return;

    // ---- this lane's range blocks -> VGPRs ------------------------------------------------
    uint32_t r[NR][NC][DW];
    int neg_rM[NR], neg_rem[NR];
    float best_err[NR], tau[NR];
    uint32_t best_cand[NR];
    {
        const uint32_t* rp = A.rng_pix + (size_t)plane * A.Nr_pad * A.n_iso * DW;
#pragma unroll
        for (int rs = 0; rs < NR; rs++) {
#pragma unroll
            for (int k = 0; k < NC; k++)
#pragma unroll
                for (int dw = 0; dw < DW; dw++)
                    r[rs][k][dw] = rp[((((size_t)tile * NR + rs) * A.n_iso + kbase + k) * DW + dw) * 64 + lane];
            FicRngStat st = A.rng_st[(size_t)plane * A.Nr_pad + ((size_t)tile * NR + rs) * 64 + lane];
            neg_rM[rs] = -st.rM;
            neg_rem[rs] = -st.rem;
            best_err[rs] = INFINITY;
            best_cand[rs] = 0xFFFFFFFFu;
            // "nothing evaluated yet": NaN makes the f32 test fail; the integer test of the 8-copy variants
            // starts from 0 and forces the chunk's first block instead
            tau[rs] = NC >= 8 ? 0.0f : __uint_as_float(0x7FC00000u);
        }
    }

    // ---- scalar streams ---------------------------------------------------------------------
    const AS4 segv* px = (const AS4 segv*)(A.pool_pix + (size_t)plane * A.Nd_pad * DW);
    const AS4 u32x2* st = (const AS4 u32x2*)(A.pool_st + (size_t)plane * A.Nd_pad);
    const AS4 double* s64p = (const AS4 double*)(A.pool_s64 + (size_t)plane * A.Nd_pad);

    uint32_t acc[NR][NC];

    // dot4 over one SEG-dword segment of a domain block against all resident range copies
    auto dots = [&](const segv& p, int seg) {
#pragma unroll
        for (int w = 0; w < SEG; w++)
#pragma unroll
            for (int rs = 0; rs < NR; rs++)
#pragma unroll
                for (int k = 0; k < NC; k++)
                    acc[rs][k] = __builtin_amdgcn_udot4(r[rs][k][seg * SEG + w], p[w], acc[rs][k], false);
    };
    auto begin_domain = [&](const u32x2& s) {
        int Sd = (int)s.x;
        int dM = (int)(s.x >> A.lgn);
#pragma unroll
        for (int rs = 0; rs < NR; rs++) {
            // -(rM*sum(d) + dM*rem): kovarianz = sum r*d + this   (24-bit operands, exact)
            uint32_t base = (uint32_t)(__mul24(neg_rM[rs], Sd) + __mul24(neg_rem[rs], dM));
#pragma unroll
            for (int k = 0; k < NC; k++) acc[rs][k] = base;
        }
    };
    // exact epilogue of one (range, copy, domain): strict '<' in ascending candidate order (FC:627), then raise tau:
    // any later candidate with |cov'|/sqrt(var') <= (1-2^-18) * |cov|/sqrt(var) has |r'| <= |r|, hence
    // error' >= error >= best, and can be skipped.
    auto evaluate = [&](int rs, int k, int d, float s32, double s64) {
        const int cov = (int)acc[rs][k];
        const float e = exact_error(cov, -neg_rem[rs], s64);
        if (e < best_err[rs]) {
            best_err[rs] = e;
            best_cand[rs] = (uint32_t)d * (uint32_t)A.n_iso + (uint32_t)(kbase + k);
        }
        const float lvl = (s32 == 0.0f) ? 0.0f : __fmul_rn(__fdiv_rn(fabsf((float)cov), s32), 0.99999618530273437500f);
        const float t = tau[rs];
        tau[rs] = (t != t) ? lvl : fmaxf(t, lvl);
    };
    auto end_domain = [&](int d, const u32x2& s) {
        const float s32 = __uint_as_float(s.y);
        if constexpr (NC >= 8) {
            // 8 isometry copies share rem/tau: reduce their covariances with integer max/min and test the two
            // extremes once --  |cov| > tau*s32  <=>  cov > li or cov < -li  with  li = floor(tau*s32)  (cov integer).
            const bool force = (d == d0);              // wave-uniform: the chunk's first block initialises tau
            bool any = force;
            int li[NR];
#pragma unroll
            for (int rs = 0; rs < NR; rs++) {
                li[rs] = (int)__fmul_rn(tau[rs], s32);  // tau <= 8192, s32 <= 4080: exact floor, far below 2^31
                int mx = (int)acc[rs][0], mn = mx;
#pragma unroll
                for (int k = 1; k < NC; k++) {
                    mx = max(mx, (int)acc[rs][k]);
                    mn = min(mn, (int)acc[rs][k]);
                }
                any |= (mx > li[rs]) | (mn < -li[rs]);
            }
            if (__builtin_expect(__any(any), 0)) {
                const double s64 = s64p[d];
#pragma unroll
                for (int rs = 0; rs < NR; rs++) {
#pragma unroll
                    for (int k = 0; k < NC; k++) {
                        const int cov = (int)acc[rs][k];
                        if (force || cov > li[rs] || cov < -li[rs]) evaluate(rs, k, d, s32, s64);
                    }
                    // rem == 0: error 0 for every block (FC:677); 8192 >= |cov|/sqrt(var) for any pair
                    if (neg_rem[rs] == 0 && best_cand[rs] != 0xFFFFFFFFu) tau[rs] = 8192.0f;
                }
            }
        } else {
            bool any = false;
            bool flag[NR][NC];
#pragma unroll
            for (int rs = 0; rs < NR; rs++) {
                const float lim = __fmul_rn(tau[rs], s32);
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    const float cf = fabsf((float)(int)acc[rs][k]);
                    flag[rs][k] = !(cf <= lim);        // NaN lim -> evaluate
                    any |= flag[rs][k];
                }
            }
            if (__builtin_expect(__any(any), 0)) {
                const double s64 = s64p[d];
#pragma unroll
                for (int rs = 0; rs < NR; rs++) {
#pragma unroll
                    for (int k = 0; k < NC; k++)
                        if (flag[rs][k]) evaluate(rs, k, d, s32, s64);
                    if (neg_rem[rs] == 0 && best_cand[rs] != 0xFFFFFFFFu) tau[rs] = 3.402823466e+38f;
                }
            }
        }
    };

    if constexpr (NSEG == 1) {
        // two domain blocks per trip, buffers A/B; chunk_len is even, the pool tail is zero-padded
        segv pa = px[d0];
        u32x2 sa = st[d0];
        for (int d = d0; d < d1; d += 2) {
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): buffer A landed
            segv pb = px[d + 1];
            u32x2 sb = st[d + 1];
            __builtin_amdgcn_sched_barrier(0);         // keep the prefetch issue above the dot4s
            begin_domain(sa);
            dots(pa, 0);
            end_domain(d, sa);
            __builtin_amdgcn_s_waitcnt(0xC07F);        // buffer B landed
            pa = px[d + 2];
            sa = st[d + 2];
            __builtin_amdgcn_sched_barrier(0);
            if (d + 1 < d1) {
                begin_domain(sb);
                dots(pb, 0);
                end_domain(d + 1, sb);
            }
        }
    } else {
        // B = 16: four 16-dword segments per domain block, buffers alternate A B A B
        segv pa = px[(size_t)d0 * NSEG];
        u32x2 sc = st[d0];
        for (int d = d0; d < d1; d++) {
            const size_t sbase = (size_t)d * NSEG;
            u32x2 sn;
            segv pb;
#pragma unroll
            for (int sgi = 0; sgi < NSEG; sgi += 2) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                pb = px[sbase + sgi + 1];
                __builtin_amdgcn_sched_barrier(0);
                if (sgi == 0) begin_domain(sc);
                dots(pa, sgi);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                pa = px[sbase + sgi + 2];              // last trip: first segment of block d+1
                if (sgi + 2 == NSEG) sn = st[d + 1];
                __builtin_amdgcn_sched_barrier(0);
                dots(pb, sgi + 1);
            }
            end_domain(d, sc);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            sc = sn;
        }
    }

    // ---- publish: lexicographic (error, candidate) minimum across chunks / isometry groups ----
#pragma unroll
    for (int rs = 0; rs < NR; rs++) {
        int j = (tile * NR + rs) * 64 + lane;
        if (j < A.Nr && best_cand[rs] != 0xFFFFFFFFu) {
            unsigned long long kk = ((unsigned long long)f32_orderable(best_err[rs]) << 32) | best_cand[rs];
            atomicMin(&A.key[(size_t)plane * A.Nr_pad + j], kk);
        }
    }
}

// host-side launchers
int fic_launch_sweep_generic(const FicBuffers& b, const FicGeom& g, int r_begin, int r_count, hipStream_t s)
{
    hipLaunchKernelGGL(k_sweep_generic, dim3((r_count + 3) / 4, g.planes), dim3(256), 0, s,
                       (const uint32_t*)b.pool_pix, b.pool_st, b.pool_s64, b.rng_pix, b.rng_st, b.key, g, r_begin,
                       r_count);
    FIC_LAUNCH_CHECK();
    return 0;
}

// Kernel-variant table of the fast sweep: (B, n_iso) -> <DW, NR, NC>.
//   B=4 : n_iso 1 -> <4,4,1>    n_iso 8 -> <4,1,8>
//   B=8 : n_iso 1 -> <16,2,1>   n_iso 8 -> <16,1,8>
//   B=16: n_iso 1 -> <64,2,1>   n_iso 8 -> <64,1,2> (4 isometry groups on grid.y)
int fic_fast_variant(int B, int n_iso, int* NR, int* NC)
{
    int nr = 0, nc = 0;
    if (B == 4) { nr = n_iso == 1 ? 4 : 1; nc = n_iso == 1 ? 1 : 8; }
    else if (B == 8) { nr = n_iso == 1 ? 2 : 1; nc = n_iso == 1 ? 1 : 8; }
    else if (B == 16) { nr = n_iso == 1 ? 2 : 1; nc = n_iso == 1 ? 1 : 2; }
    else return -1;
    if (n_iso != 1 && n_iso != 8) return -1;
    if (NR) *NR = nr;
    if (NC) *NC = nc;
    return 0;
}

int fic_launch_sweep_fast(const FicBuffers& b, const FicGeom& g, int tile0, int ntiles, int chunk_len, int nchunks,
                          hipStream_t s)
{
    SweepArgs A;
    A.pool_pix = (const uint32_t*)b.pool_pix;
    A.pool_st = b.pool_st;
    A.pool_s64 = b.pool_s64;
    A.rng_pix = b.rng_pix;
    A.rng_st = b.rng_st;
    A.key = b.key;
    A.Nd = g.Nd; A.Nd_pad = g.Nd_pad; A.Nr = g.Nr; A.Nr_pad = g.Nr_pad; A.n_iso = g.n_iso; A.lgn = g.lgn;
    A.tile0 = tile0; A.ntiles = ntiles; A.chunk_len = chunk_len; A.nchunks = nchunks; A.planes = g.planes;
    int NR, NC;
    if (fic_fast_variant(g.B, g.n_iso, &NR, &NC) || NR != g.NR) return -1;
    dim3 grid((unsigned)(nchunks * (g.n_iso / NC) * g.planes) * (unsigned)((ntiles + 3) / 4), 1, 1);
    dim3 block(256);
    if (g.B == 4 && g.n_iso == 1) hipLaunchKernelGGL((k_sweep_fast<4, 4, 1>), grid, block, 0, s, A);
    else if (g.B == 4) hipLaunchKernelGGL((k_sweep_fast<4, 1, 8>), grid, block, 0, s, A);
    else if (g.B == 8 && g.n_iso == 1) hipLaunchKernelGGL((k_sweep_fast<16, 2, 1>), grid, block, 0, s, A);
    else if (g.B == 8) hipLaunchKernelGGL((k_sweep_fast<16, 1, 8>), grid, block, 0, s, A);
    else if (g.B == 16 && g.n_iso == 1) hipLaunchKernelGGL((k_sweep_fast<64, 2, 1>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((k_sweep_fast<64, 1, 2>), grid, block, 0, s, A);
    FIC_LAUNCH_CHECK();
    return 0;
}
