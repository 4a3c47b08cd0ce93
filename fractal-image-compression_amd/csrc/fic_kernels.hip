// fic_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for the bvk_ss19 grey encode hot path.
//
// Written for wave64 / SIMD-32 CDNA4 only.  Compile with -ffp-contract=off: every
// float expression below must round once per operation exactly like the Java
// reference (FractalCompression.java = FC, Domainblock.java = DB).
//
// Pipeline (one launch each, all planes of a batch in one grid):
//   k_argb_to_gray   RasterImage.argb -> R channel bytes          (FC:596, FC:977)
//   k_scale          2:1 box average                              (scaleImage     FC:970-1007)
//   k_pool           expanded domain pool + per-block sum/var     (createCodebuch FC:1015-1050, DB:23-29,92-115)
//   k_range_stat / k_range_copies   range blocks: rM/rem, isometry copies   (getRangeblock FC:588-602, getMittelwert FC:67-73)
//   k_sweep_fast     full-pool search, lane = range block (DEFAULT) (getBestDomainblock FC:613-632 + getErrorVarianceCovariance FC:655-687)
//   k_sweep_generic  any window, wave = range block               (same, + getDomainBlockIndex FC:516-545, generateKernel FC:84-100)
//   k_sweep_mfma     opt-in matrix-core full-pool search (+ k_pool_mfma, k_range_mfma fragment prep)
//   k_finalize       (a,b) fit, clamp, quantise                   (FC:634-642, writeData FC:242-244)
//   k_collage        one-step collage image                       (getBestGeneratedCollage FC:269-300)
// Decoder:  k_decode_paint / k_decode_paint_rgb / k_decode_step   (decodeGreyScale FC:356-421, decodeRGB FC:430-508)
// Joint RGB: k_scale_rgb, k_pool_rgb, k_range_rgb, k_sweep_rgb, k_finalize_rgb, k_collage_rgb (encodeRGB FC:171-219 ...)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "fic_device.h"
#include "fic_launch.h"

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
#define AS4 __attribute__((address_space(4)))

// Java (int) cast of a float, JLS 5.1.3: NaN -> 0, saturating.
__device__ __forceinline__ int java_f2i(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

// Monotone map f32 -> u32 (unsigned order == float order); -0 folded onto +0.
__device__ __forceinline__ uint32_t f32_orderable(float f)
{
    uint32_t u = __float_as_uint(f + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_orderable(uint32_t o)
{
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}

// getErrorVarianceCovariance FC:674-683 given the exact integer sums.
//   cov  = kovarianz   (exact integer, |cov| < 2^24 so the Java float accumulation is exact)
//   rem  = varianzRange (exact integer 0..n-1)
//   s64  = Math.sqrt((double) varianzSquare), correctly rounded; s64 == 0  <=>  variance == 0
__device__ __forceinline__ float exact_error(int cov, int rem, double s64)
{
    float remf = (float)rem;
    float r;
    if (rem == 0 || s64 == 0.0)
        r = 0.0f;
    else
        r = (float)((double)cov / ((double)remf * s64));
    r = __fmul_rn(r, r);
    return __fmul_rn(__fmul_rn(remf, remf), __fsub_rn(1.0f, r));
}

// Source pixel index of isometry k at output position (x,y): out[y][x] = d[sy][sx].
// k = 0 identity, 1 rot90cw, 2 rot180, 3 rot270cw, 4 mirror L-R, 5 mirror T-B, 6 transpose, 7 anti-transpose.
// (Extension -- the reference has no isometries; definition shared with oracle/fic_oracle.c fo_iso_source.)
__device__ __forceinline__ int iso_source(int k, int B, int x, int y)
{
    int m = B - 1, sx, sy;
    switch (k) {
    default:
    case 0: sx = x;     sy = y;     break;
    case 1: sx = y;     sy = m - x; break;
    case 2: sx = m - x; sy = m - y; break;
    case 3: sx = m - y; sy = x;     break;
    case 4: sx = m - x; sy = y;     break;
    case 5: sx = x;     sy = m - y; break;
    case 6: sx = y;     sy = x;     break;
    case 7: sx = m - y; sy = m - x; break;
    }
    return sx + sy * B;
}
__device__ __forceinline__ int iso_inverse(int k) { return k == 1 ? 3 : (k == 3 ? 1 : k); }

// getDomainBlockIndex FC:516-545
__device__ __forceinline__ int domain_block_index(int xr, int yr, int Rw, int Rh, int Dw)
{
    int i = 0;
    if (yr == 0) yr = 1;
    if (xr == 0) xr = 1;
    if (yr == Rh - 1) yr = yr - 1;
    if (xr == Rw - 1) xr = xr - 1;
    if (xr > 1) {
        if (yr == 0) i = xr;
        else i = (xr * 2) - 2 + (yr + yr - 1) * Dw;
    } else if (xr == 1) {
        if (yr == 0) i = xr;
        else i = xr + (yr + yr - 1) * Dw;
    }
    return i;
}
// generateKernel FC:84-100
__device__ __forceinline__ void window_origin(int i, int Dw, int Dh, int wK, int& dy, int& dx)
{
    dy = i / Dw - wK / 2;
    dx = i % Dw - wK / 2;
    if (dx < 0) dx = 0;
    if (dy < 0) dy = 0;
    if (dx + wK >= Dw) dx = Dw - wK;
    if (dy + wK >= Dh) dy = Dh - wK;
}
// window-local candidate -> global pool index for range j (FC:128-150)
__device__ __forceinline__ int window_to_global(const FicGeom& g, int j, int wloc)
{
    if (g.full) return wloc;
    int xr = j % g.Rw, yr = j / g.Rw;
    int i = domain_block_index(xr, yr, g.Rw, g.Rh, g.Dw);
    int dy, dx;
    window_origin(i, g.Dw, g.Dh, g.wK, dy, dx);
    int ky = wloc / g.wK, kx = wloc % g.wK;
    return dx + kx + (dy + ky) * g.Dw;
}

// Address of dword dw of isometry copy k of range j in the lane-transposed range store:
//   rng_pix[plane][tile][rs][k][dw][lane],  j = tile*64*NR + rs*64 + lane.
__device__ __forceinline__ size_t rng_word_index(const FicGeom& g, int j, int k, int dw)
{
    int tsz = 64 * g.NR;
    int tile = j / tsz, s = j % tsz;
    int rs = s >> 6, lane = s & 63;
    return ((((size_t)tile * g.NR + rs) * g.n_iso + k) * g.DW + dw) * 64 + lane;
}

// ---------------------------------------------------------------------------------------------
// k_argb_to_gray : RasterImage.argb (int ARGB) -> R channel bytes
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_argb_to_gray(const int32_t* __restrict__ argb, uint8_t* __restrict__ gray,
                                                      size_t npix)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < npix) gray[i] = (uint8_t)((argb[i] >> 16) & 0xff);
}

// ---------------------------------------------------------------------------------------------
// k_scale : scaleImage FC:970-1007 for even W,H.  One thread per scaled pixel.
// Quirk kept: the 4th tap is 128 when x+1 >= image.HEIGHT (FC:993).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scale(const uint8_t* __restrict__ gray, uint8_t* __restrict__ scaled, FicGeom g)
{
    int xs = blockIdx.x * 256 + threadIdx.x;
    int ys = blockIdx.y;
    int plane = blockIdx.z;
    if (xs >= g.Ws) return;
    const uint8_t* im = gray + (size_t)plane * g.W * g.H;
    int x = 2 * xs, y = 2 * ys;
    int m = im[x + (size_t)y * g.W];
    m += im[x + 1 + (size_t)y * g.W];
    m += im[x + (size_t)(y + 1) * g.W];
    m += (x + 1 >= g.H) ? 128 : (int)im[x + 1 + (size_t)(y + 1) * g.W];
    scaled[(size_t)plane * g.Ws * g.Hs + (size_t)ys * g.Ws + xs] = (uint8_t)(m / 4);
}

// ---------------------------------------------------------------------------------------------
// k_pool : createCodebuch FC:1015-1050 + Domainblock DB:23-29.
// One thread per (domain block, pixel row); B threads cooperate on a block so the pool
// rows are written as whole B-byte runs.  Block k = (c,r) sits at scaled (c*abstand, r*abstand);
// pixel order inside a block is rx + ry*B (FC:1036).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pool(const uint8_t* __restrict__ scaled, uint8_t* __restrict__ pool_pix,
                                              FicDomStat* __restrict__ pool_st, uint32_t* __restrict__ pool_var,
                                              double* __restrict__ pool_s64, FicGeom g)
{
    // thread t handles row (t % B) of block (t / B)
    const int B = g.B;
    const int per_blk = 256 / B;                   // domain blocks per workgroup
    int local = threadIdx.x / B, ry = threadIdx.x % B;
    int d = blockIdx.x * per_blk + local;
    int plane = blockIdx.y;
    __shared__ int s_sum[256];
    __shared__ int s_sq[256];
    int sum = 0, sq = 0;
    bool ok = d < g.Nd;
    uint8_t row[16];
    if (ok) {
        int c = d % g.Dw, r = d / g.Dw;
        const uint8_t* src = scaled + (size_t)plane * g.Ws * g.Hs + (size_t)(r * g.abstand + ry) * g.Ws + c * g.abstand;
        uint8_t* dst = pool_pix + ((size_t)plane * g.Nd_pad + d) * g.n + ry * B;
#pragma unroll
        for (int rx = 0; rx < 16; rx++)
            if (rx < B) {
                int v = src[rx];
                row[rx] = (uint8_t)v;
                sum += v;
                sq += v * v;
            }
        if (B == 4) {
            *(uint32_t*)dst = *(uint32_t*)row;
        } else if (B == 8) {
            *(uint2*)dst = *(uint2*)row;
        } else {
            *(uint4*)dst = *(uint4*)row;
        }
    }
    s_sum[threadIdx.x] = sum;
    s_sq[threadIdx.x] = sq;
    __syncthreads();
    if (ok && ry == 0) {
        int S = 0, Q = 0;
        for (int t = 0; t < B; t++) {
            S += s_sum[threadIdx.x + t];
            Q += s_sq[threadIdx.x + t];
        }
        // mittelWert = S / n (DB:97);  variance = sum (d - m)^2 = Q - 2 m S + n m^2  (exact integer < 2^24,
        // equal to the float accumulation of DB:110-111 because every partial sum is an exact float).
        int m = S >> g.lgn;
        int var = Q - 2 * m * S + g.n * m * m;
        double s64 = __dsqrt_rn((double)var);      // Math.sqrt((double) variance), FC:677,680
        size_t o = (size_t)plane * g.Nd_pad + d;
        FicDomStat st;
        st.sum = (uint32_t)S;
        st.s32 = (float)s64;
        pool_st[o] = st;
        pool_var[o] = (uint32_t)var;
        pool_s64[o] = s64;
    }
}

// ---------------------------------------------------------------------------------------------
// k_range : getRangeblock FC:588-602 + getMittelwert FC:67-73, and the isometry copies.
// One thread per range block.  copy_k[pos] = r[iso_source(inverse(k), pos)] so that
//   dot(copy_k, d) == dot(r, iso_k(d)).
// ---------------------------------------------------------------------------------------------
// (a) statistics: one thread per range block, rows read as 4-byte words, byte sums with v_sad_u8
__global__ __launch_bounds__(256) void k_range_stat(const uint8_t* __restrict__ gray, FicRngStat* __restrict__ rng_st, FicGeom g)
{
    int j = blockIdx.x * 256 + threadIdx.x;
    int plane = blockIdx.y;
    if (j >= g.Nr_pad) return;
    FicRngStat st;
    st.rM = 0;
    st.rem = 0;
    if (j < g.Nr) {
        const int B = g.B;
        const uint8_t* im = gray + (size_t)plane * g.W * g.H + (size_t)((j / g.Rw) * B) * g.W + (j % g.Rw) * B;
        uint32_t S = 0;
        for (int y = 0; y < B; y++) {
            const uint32_t* row = (const uint32_t*)(im + (size_t)y * g.W);      // B is a multiple of 4, W of B
            for (int x = 0; x < B / 4; x++) S = __builtin_amdgcn_sad_u8(row[x], 0u, S);
        }
        st.rM = (int)(S >> g.lgn);
        st.rem = (int)S - (st.rM << g.lgn);
    }
    rng_st[(size_t)plane * g.Nr_pad + j] = st;
}

// (b) pixels: one thread per output dword (range j, isometry copy k, dword dw); consecutive threads are
// consecutive lanes of the lane-transposed store, so every wave writes one coalesced 256-byte run.
// One workgroup per group of 64 range blocks (one "rs" slice of a tile): the blocks are read word-wise
// into LDS (row stride n+4 bytes: bank-conflict-free for the byte gathers), then every output dword of
// every isometry copy is assembled from 4 LDS bytes and written as part of a coalesced 256-byte run.
__global__ __launch_bounds__(256) void k_range_copies(const uint8_t* __restrict__ gray, uint32_t* __restrict__ rng_pix, FicGeom g)
{
    __shared__ __attribute__((aligned(16))) uint8_t blk[64 * (256 + 4)];
    const int plane = blockIdx.y;
    const int grp = blockIdx.x;                                // (tile, rs)
    const int B = g.B, n = g.n, DW = g.DW;
    const int stride = n + 4;
    const int j0 = grp * 64;
    const uint8_t* img = gray + (size_t)plane * g.W * g.H;
    for (int i = threadIdx.x; i < 64 * DW; i += 256) {         // DW words per block
        const int lane = i / DW, wd = i % DW;
        const int j = j0 + lane;
        uint32_t v = 0;
        if (j < g.Nr) {                                        // tile padding stays zero (never written back)
            const int pos = wd * 4;
            const uint8_t* p = img + (size_t)((j / g.Rw) * B + pos / B) * g.W + (j % g.Rw) * B + pos % B;
            v = *(const uint32_t*)p;                           // B, W multiples of 4: aligned
        }
        *(uint32_t*)&blk[lane * stride + wd * 4] = v;
    }
    __syncthreads();
    uint32_t* out = rng_pix + (size_t)plane * g.Nr_pad * g.n_iso * DW + (size_t)grp * g.n_iso * DW * 64;
    for (int o = threadIdx.x; o < g.n_iso * DW * 64; o += 256) {   // [k][dw][lane]; k, dw are wave-uniform
        const int lane = o & 63, dw = (o >> 6) % DW, k = (o >> 6) / DW;
        const int ki = iso_inverse(k);
        const uint8_t* b = blk + lane * stride;
        uint32_t w = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int pos = dw * 4 + t;
            w |= (uint32_t)b[iso_source(ki, B, pos % B, pos / B)] << (8 * t);
        }
        out[o] = w;
    }
}

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
};

template <int DW, int NR, int NC>
__global__ __launch_bounds__(256) void k_sweep_fast(SweepArgs A)
{
    constexpr int SEG = DW < 16 ? DW : 16;     // dwords per scalar load
    constexpr int NSEG = DW / SEG;             // scalar loads per domain block (1, or 4 for B = 16)
    typedef uint32_t segv __attribute__((ext_vector_type(SEG)));

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tl = blockIdx.x * 4 + wave;
    if (tl >= A.ntiles) return;                        // wave-uniform
    const int tile = A.tile0 + tl;
    const int ngroups = A.n_iso / NC;
    const int chunk = blockIdx.y / ngroups;
    const int kbase = (blockIdx.y % ngroups) * NC;
    const int plane = blockIdx.z;

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

// ---------------------------------------------------------------------------------------------
// k_sweep_mfma : OPT-IN matrix-core variant of the full-pool sweep ("sweep" = 3; B = 8, n_iso = 8).
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
//   K = 64 pixels = 2 MFMAs; lane (r, h) supplies bytes [32m+16h, 32m+16h+16) of its row/col for
//   MFMA m -- the same map on both operands, so any k order is consistent.
//   A workgroup (4 waves) shares 64 range blocks (16 row tiles) through LDS and each wave
//   sweeps its own pool chunk; tau (the prune threshold) is private to a wave, kept in LDS and
//   raised with ds_max_u32 by the rare exact path; winners go straight to the global key with
//   the same 64-bit atomicMin as k_sweep_fast.  The first domain tile of a chunk is always
//   evaluated exactly (it initialises tau).
// ---------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define FIC_MFMA_RT 16                     // row tiles per workgroup: 16 x 4 = 64 range blocks
#define FIC_TAU_ALL 4096.0f                // >= |cov| / sqrt(var) for any pair (|cov| <= 2040 * sqrt(var))

// pool -> B fragments: poolB[plane][dtile][m][lane] = 16 bytes of domain (dtile*32 + lane&31), i8-shifted
__global__ __launch_bounds__(256) void k_pool_mfma(const uint8_t* __restrict__ pool_pix, v4i* __restrict__ poolB,
                                                   FicGeom g, int ndtiles_alloc)
{
    int i = blockIdx.x * 256 + threadIdx.x;            // (dtile, m, lane)
    int plane = blockIdx.y;
    if (i >= ndtiles_alloc * 128) return;
    int lane = i & 63, m = (i >> 6) & 1, dtile = i >> 7;
    int d = dtile * 32 + (lane & 31), h = lane >> 5;
    v4i v = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};   // pixel 0 for the padding blocks
    if (d < g.Nd) {
        v = *(const v4i*)(pool_pix + ((size_t)plane * g.Nd_pad + d) * 64 + 32 * m + 16 * h);
        v ^= (v4i){(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    }
    poolB[(size_t)plane * ndtiles_alloc * 128 + i] = v;
}

// ranges -> A fragments + per-range constants {K = 128*Sr - 16384*n, A = 128 - rM, negR = -rem}
__global__ __launch_bounds__(256) void k_range_mfma(const uint32_t* __restrict__ rng_pix,
                                                    const FicRngStat* __restrict__ rng_st, v4i* __restrict__ rngA,
                                                    int* __restrict__ rconst, FicGeom g)
{
    int i = blockIdx.x * 256 + threadIdx.x;            // (group, t, m, lane)
    int plane = blockIdx.y;
    int total = g.tiles * FIC_MFMA_RT * 128;
    if (i >= total) return;
    int lane = i & 63, m = (i >> 6) & 1, t = (i >> 7) % FIC_MFMA_RT, group = (i >> 7) / FIC_MFMA_RT;
    int r = lane & 31, h = lane >> 5, q = r >> 3, k = r & 7;
    int j = group * 64 + 4 * t + q;
    const uint32_t* rp = rng_pix + (size_t)plane * g.Nr_pad * g.n_iso * g.DW;
    v4i v;
#pragma unroll
    for (int w = 0; w < 4; w++) v[w] = (int)(rp[rng_word_index(g, j, k, 8 * m + 4 * h + w)] ^ 0x80808080u);
    rngA[(size_t)plane * total + i] = v;
    if (m == 0 && lane < 16) {                          // 16 ints per (group, t)
        int qq = lane & 3, what = lane >> 2;
        FicRngStat st = rng_st[(size_t)plane * g.Nr_pad + group * 64 + 4 * t + qq];
        int Sr = st.rM * g.n + st.rem;
        int val = what == 0 ? 128 * Sr - 16384 * g.n : what == 1 ? 128 - st.rM : what == 2 ? -st.rem : 0;
        rconst[((size_t)plane * g.tiles * FIC_MFMA_RT + (size_t)group * FIC_MFMA_RT + t) * 16 + lane] = val;
    }
}

struct MfmaArgs {
    const v4i* poolB;
    const FicDomStat* pool_st;
    const double* pool_s64;
    const v4i* rngA;
    const int* rconst;
    unsigned long long* key;
    int Nd, Nd_pad, Nr_pad, lgn;
    int ndtiles, ndtiles_alloc, ngroups;
    int group0;                      // first range group of this shard
    int tiles_per_chunk, nchunks;
};

__global__ __launch_bounds__(256) void k_sweep_mfma(MfmaArgs A)
{
    constexpr int RT = FIC_MFMA_RT;
    __shared__ v4i sA[RT * 2 * 64];                    // 32 KiB: A fragments of 64 range blocks x 8 copies
    __shared__ __attribute__((aligned(16))) uint32_t sTau[RT * 4];   // per range: prune threshold (f32 bits, >= 0);
                                                                     // range 4t+q belongs to the wave that owns row tile t
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int group = A.group0 + blockIdx.x;
    const int plane = blockIdx.z;
    const int chunk = blockIdx.y;                      // all 4 waves sweep the same pool chunk (B tiles hit in L1) ...
    const int t_first = wave * (RT / 4);               // ... and split the 16 row tiles
    const size_t gbase = (size_t)plane * A.ngroups + group;
    const AS4 int* rc_base = (const AS4 int*)A.rconst + gbase * RT * 16;

    for (int i = threadIdx.x; i < RT * 2 * 64; i += 256) sA[i] = A.rngA[gbase * RT * 128 + i];
    // ranges with rem == 0: error 0 for every block (FC:677) -> nothing after the first tile can win
    if (wave == 0) sTau[lane] = (A.rconst[(gbase * RT + (lane >> 2)) * 16 + 8 + (lane & 3)] == 0) ? __float_as_uint(FIC_TAU_ALL) : 0u;
    __syncthreads();
    if (chunk >= A.nchunks) return;
    const int dt0 = chunk * A.tiles_per_chunk;
    int dt1 = dt0 + A.tiles_per_chunk;
    if (dt1 > A.ndtiles) dt1 = A.ndtiles;
    if (dt0 >= dt1) return;

    const int jcol = lane & 31, half = lane >> 5;
    const v4i* pb = A.poolB + (size_t)plane * A.ndtiles_alloc * 128;
    const FicDomStat* pst = A.pool_st + (size_t)plane * A.Nd_pad;
    const double* p64 = A.pool_s64 + (size_t)plane * A.Nd_pad;
    unsigned long long* keyp = A.key + (size_t)plane * A.Nr_pad + (size_t)group * 64;
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    v4i b0 = pb[(size_t)dt0 * 128 + lane], b1 = pb[(size_t)dt0 * 128 + 64 + lane];
    int dcur = dt0 * 32 + jcol;
    FicDomStat st = pst[dcur < A.Nd_pad ? dcur : 0];
    for (int dt = dt0; dt < dt1; dt++) {
        // prefetch the next domain tile (the fragment store has one spare tile)
        v4i nb0 = pb[(size_t)(dt + 1) * 128 + lane], nb1 = pb[(size_t)(dt + 1) * 128 + 64 + lane];
        int dnext = (dt + 1) * 32 + jcol;
        FicDomStat nst = pst[dnext < A.Nd_pad ? dnext : 0];
        const int d = dt * 32 + jcol;
        const bool valid = d < A.Nd;
        const int Sd = valid ? (int)st.sum : 0;
        const int dM = Sd >> A.lgn;
        const float s32 = valid ? st.s32 : 0.0f;
        const bool force = (dt == dt0);                 // wave-uniform

        for (int t = t_first; t < t_first + RT / 4; t++) {
            v4i a0 = sA[(t * 2 + 0) * 64 + lane], a1 = sA[(t * 2 + 1) * 64 + lane];
            const uint4 tau4 = *(const uint4*)&sTau[t * 4];
            const AS4 int* rc = rc_base + t * 16;       // wave-uniform -> scalar loads
            v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, zero, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc, 0, 0, 0);

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
                const int li = (int)__fmul_rn(tau, s32);            // < 2^30 by construction (tau <= 4096, s32 <= 4080)
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
                    if (!((anyq[q] || force) && valid)) continue;       // whole range untouched on this lane: skip its 4 copies
                    const uint32_t u = (uint32_t)acc[e] + offq[q];
                    if (u > spanq[q] || force) {
                        const int cov = (int)(u - (spanq[q] >> 1));
                        const int rem = -rc[8 + q];
                        const float err = exact_error(cov, rem, s64);
                        const uint32_t cand = (uint32_t)d * 8u + (uint32_t)((e & 3) + 4 * half);
                        atomicMin(&keyp[4 * t + q], ((unsigned long long)f32_orderable(err) << 32) | cand);
                        if (rem != 0) {
                            float lvl = (s32 == 0.0f) ? 0.0f
                                                      : __fmul_rn(__fdiv_rn(fabsf((float)cov), s32), 0.99999618530273437500f);
                            atomicMax(&sTau[4 * t + q], __float_as_uint(lvl));   // lvl >= 0: uint order == float order
                        }
                    }
                }
            }
        }
        b0 = nb0; b1 = nb1; st = nst;
    }
}

// ---------------------------------------------------------------------------------------------
// k_sweep_mfma1 : OPT-IN matrix-core sweep for the reference algorithm (n_iso = 1), B = 4 / 8 / 16
// ("sweep" = 3).  Same exactness scheme as k_sweep_mfma; the mapping is transposed:
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
    int tiles_per_chunk, nchunks;
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
    const int plane = blockIdx.z;
    const int ct0 = A.ct_begin + blockIdx.x * CT;            // first column tile of the workgroup
    const size_t cbase = (size_t)plane * A.nctiles_alloc + ct0;
    for (int i = threadIdx.x; i < CT * NM * 64; i += 256) sB[i] = A.rngB[cbase * NM * 64 + i];
    for (int i = threadIdx.x; i < CT * 32; i += 256) sC[i] = A.rconst[cbase * 32 + i];
    __syncthreads();
    const int chunk = blockIdx.y;
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

// ---------------------------------------------------------------------------------------------
// k_finalize : getBestDomainblock tail FC:634-642 + writeData quantiser FC:242-244.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(const uint32_t* __restrict__ pool_pix,
                                                  const FicDomStat* __restrict__ pool_st,
                                                  const uint32_t* __restrict__ pool_var,
                                                  const uint32_t* __restrict__ rng_pix,
                                                  const FicRngStat* __restrict__ rng_st,
                                                  const unsigned long long* __restrict__ key, FicOutputs out, FicGeom g,
                                                  int r_begin, int r_count)
{
    int jr = blockIdx.x * 256 + threadIdx.x;
    int plane = blockIdx.y;
    if (jr >= r_count) return;
    int j = r_begin + jr;
    unsigned long long kk = key[(size_t)plane * g.Nr_pad + j];
    uint32_t c = (uint32_t)kk;
    int wloc = (int)(c / (uint32_t)g.n_iso), k = (int)(c % (uint32_t)g.n_iso);
    int gi = window_to_global(g, j, wloc);
    const uint32_t* pp = pool_pix + ((size_t)plane * g.Nd_pad + gi) * g.DW;
    const uint32_t* rp = rng_pix + (size_t)plane * g.Nr_pad * g.n_iso * g.DW;
    FicDomStat ds = pool_st[(size_t)plane * g.Nd_pad + gi];
    FicRngStat rs = rng_st[(size_t)plane * g.Nr_pad + j];
    uint32_t acc = 0;
    for (int dw = 0; dw < g.DW; dw++) acc = __builtin_amdgcn_udot4(rp[rng_word_index(g, j, k, dw)], pp[dw], acc, false);
    int dM = (int)(ds.sum >> g.lgn);
    int cov = (int)acc - rs.rM * (int)ds.sum - dM * rs.rem;
    float var = (float)pool_var[(size_t)plane * g.Nd_pad + gi];
    float a = __fdiv_rn((float)cov, var);              // FC:634  (0/0 -> NaN when a flat block wins)
    if (a < -1.0f) a = -1.0f;                          // FC:636-639 (NaN passes through)
    else if (a > 1.0f) a = 1.0f;
    float b = __fsub_rn((float)rs.rM, __fmul_rn(a, (float)dM));   // FC:641, never fused
    size_t o = (size_t)plane * g.Nr + j;
    out.qrows[3 * o + 0] = wloc;                       // (int) imageInfo[row][0]
    out.qrows[3 * o + 1] = java_f2i(__fmul_rn(a, 100.0f));
    out.qrows[3 * o + 2] = java_f2i(b);
    if (a != a) a = __uint_as_float(0x7FC00000u);      // canonical NaN (Java has one NaN value)
    if (b != b) b = __uint_as_float(0x7FC00000u);
    out.idx_local[o] = wloc;
    out.idx_global[o] = gi;
    out.iso[o] = k;
    out.a[o] = a;
    out.b[o] = b;
    out.err[o] = f32_from_orderable((uint32_t)(kk >> 32));
}

// ---------------------------------------------------------------------------------------------
// k_collage : getBestGeneratedCollage FC:269-300 -- value = clamp((int)(a*domain + b)) with the
// UNQUANTISED float a,b (FC:287), grey ARGB out.  One thread per pixel.  For n_iso = 8 the domain
// pixel is read through the winning isometry (extension).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_collage(const uint8_t* __restrict__ pool_pix, FicOutputs out,
                                                 int32_t* __restrict__ collage, FicGeom g)
{
    int x = blockIdx.x * 256 + threadIdx.x;
    int y = blockIdx.y;
    int plane = blockIdx.z;
    if (x >= g.W) return;
    int j = (y / g.B) * g.Rw + (x / g.B);
    int rx = x % g.B, ry = y % g.B;
    size_t o = (size_t)plane * g.Nr + j;
    int gi = out.idx_global[o];
    int src = iso_source(out.iso[o], g.B, rx, ry);
    int domain = pool_pix[((size_t)plane * g.Nd_pad + gi) * g.n + src];
    float v = __fadd_rn(__fmul_rn(out.a[o], (float)domain), out.b[o]);
    int value = java_f2i(v);
    value = value < 0 ? 0 : (value > 255 ? 255 : value);
    collage[(size_t)plane * g.W * g.H + (size_t)y * g.W + x] =
        (int32_t)(0xff000000u | ((uint32_t)value << 16) | ((uint32_t)value << 8) | (uint32_t)value);
}

// ---------------------------------------------------------------------------------------------
// Decoder, decodeGreyScale FC:356-421.  One iteration = k_scale (the pool of FC:382 is only ever
// read at the winning block, so domain pixels come straight from the scaled image: block g=(c,r)
// pixel (rx,ry) is scaled[(r*abstand+ry)*Ws + c*abstand+rx], FC:1027-1037) + k_decode_paint +
// k_decode_step.  The image is updated in place like FC:404: every thread reads its own old pixel
// (the 'range' of FC:391) and the scaled copy made before the paint.
//   qrows  : the ints of the .run stream per range {i_local, (int)(a*100), (int)b}  (FC:372-374)
//   a      = (float) q / 100f  (FC:373), b = (float) q
//   value  = clamp((int)(a*domain + b))                              (FC:396-402)
//   SSD    = sum (range - value)^2 as an exact integer.  Java accumulates it in float (FC:407):
//            identical whenever SSD < 2^24 (all partial sums are then exact floats) -- always the
//            case for the iteration that ends the loop with avgError < 1 when w*h <= 2^24.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_decode_paint(const uint8_t* __restrict__ scaled, uint8_t* __restrict__ image,
                                                      const int32_t* __restrict__ qrows, const int32_t* __restrict__ iso,
                                                      FicDecodeState* __restrict__ state, int counter, FicGeom g)
{
    int x = blockIdx.x * 256 + threadIdx.x;
    int y = blockIdx.y;
    int plane = blockIdx.z;
    FicDecodeState* st = state + plane;
    if (st->done) return;                                  // uniform per plane
    unsigned long long sq = 0;
    if (x < g.W) {
        int j = (y / g.B) * g.Rw + (x / g.B);
        int rx = x % g.B, ry = y % g.B;
        size_t o = (size_t)plane * g.Nr + j;
        int wloc = qrows[3 * o + 0];
        float a = __fdiv_rn((float)qrows[3 * o + 1], 100.0f);
        float b = (float)qrows[3 * o + 2];
        bool ok = wloc >= 0 && wloc < g.wK * g.wK;
        int gi = ok ? window_to_global(g, j, wloc) : 0;
        if (!ok || gi < 0 || gi >= g.Nd) {
            st->bad_index = 1;                             // Java: ArrayIndexOutOfBoundsException at FC:394
        } else {
            int src = iso ? iso_source(iso[o], g.B, rx, ry) : rx + ry * g.B;
            int c = gi % g.Dw, r = gi / g.Dw;
            int domain = scaled[(size_t)plane * g.Ws * g.Hs + (size_t)(r * g.abstand + src / g.B) * g.Ws + c * g.abstand + src % g.B];
            int value = java_f2i(__fadd_rn(__fmul_rn(a, (float)domain), b));
            value = value < 0 ? 0 : (value > 255 ? 255 : value);
            size_t p = (size_t)plane * g.W * g.H + (size_t)y * g.W + x;
            int d = (int)image[p] - value;
            image[p] = (uint8_t)value;
            sq = (unsigned long long)(d * d);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    __shared__ unsigned long long s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (t) atomicAdd(&st->ssd[counter], t);
    }
}

// Loop control of FC:413-417, one thread per plane.
__global__ void k_decode_step(FicDecodeState* __restrict__ state, int counter, int wh, int planes)
{
    int plane = blockIdx.x * blockDim.x + threadIdx.x;
    if (plane >= planes) return;
    FicDecodeState* st = state + plane;
    if (st->done) return;
    float avg = __fadd_rn(st->avg, (float)st->ssd[counter]);   // avgError += ... (carry-in is 0 after iteration 0)
    avg = __fdiv_rn(avg, (float)wh);                           // FC:413
    st->iters = counter + 1;
    st->avg_out = avg;
    if (avg < 1.0f) st->done = 1;                              // FC:414
    else st->avg = (counter != 49) ? 0.0f : avg;               // FC:416
    if (counter == 49) st->done = 1;
}

// =============================================================================================
// Joint-RGB encode (encodeRGB FC:171-219) -- what FC.encode dispatches to for colour input.
// One 'a' for the three channels, three 'b'.  Quirks kept on purpose (they are in unknown.run):
//   * scaleImageRGB's 4th tap re-reads (x, y+1)                                   FC:945-947
//   * varianzSquare = varianceR + varianceG + mittelWertB (sic)                   FC:776
//   * Domainblock.variance is never set for RGB, so varianzDomain starts at sqrt(0) and just
//     accumulates sum(greyD)                                                      FC:778,791, DB:30-41
//   * r is pure f32 (no double)                                                   FC:800
//   * kovarianz sums can exceed 2^24: the f32 accumulation order i = 0..n-1 matters, so every
//     candidate is accumulated sequentially by one lane                           FC:781-792
// greyR_i and greyD_i are exact small integers (sums of three channel deviations), and so are
// their products (< 2^24); only the running sum rounds.
// =============================================================================================
__device__ __forceinline__ int ch_r(int32_t v) { return (v >> 16) & 0xff; }
__device__ __forceinline__ int ch_g(int32_t v) { return (v >> 8) & 0xff; }
__device__ __forceinline__ int ch_b(int32_t v) { return v & 0xff; }

// scaleImageRGB FC:901-962 (even W,H): sum = p00 + p10 + p01 + (x+1 >= HEIGHT ? 128 : p01), per channel
__global__ __launch_bounds__(256) void k_scale_rgb(const int32_t* __restrict__ argb, int32_t* __restrict__ scaled, FicGeom g)
{
    int xs = blockIdx.x * 256 + threadIdx.x, ys = blockIdx.y;
    if (xs >= g.Ws) return;
    int x = 2 * xs, y = 2 * ys;
    int32_t p00 = argb[x + (size_t)y * g.W], p10 = argb[x + 1 + (size_t)y * g.W], p01 = argb[x + (size_t)(y + 1) * g.W];
    bool edge = (x + 1 >= g.H);
    int mR = (ch_r(p00) + ch_r(p10) + ch_r(p01) + (edge ? 128 : ch_r(p01))) / 4;
    int mG = (ch_g(p00) + ch_g(p10) + ch_g(p01) + (edge ? 128 : ch_g(p01))) / 4;
    int mB = (ch_b(p00) + ch_b(p10) + ch_b(p01) + (edge ? 128 : ch_b(p01))) / 4;
    scaled[(size_t)ys * g.Ws + xs] = (int32_t)(0xff000000u | ((uint32_t)mR << 16) | ((uint32_t)mG << 8) | (uint32_t)mB);
}

// createCodebuchRGB FC:1058-1093 + Domainblock RGB ctor DB:30-41.  One thread per domain block.
__global__ __launch_bounds__(256) void k_pool_rgb(const int32_t* __restrict__ scaled, uint16_t* __restrict__ pool_sum,
                                                  FicRgbDomStat* __restrict__ st, FicGeom g)
{
    int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= g.Nd) return;
    int c = d % g.Dw, r = d / g.Dw;
    const int32_t* src = scaled + (size_t)(r * g.abstand) * g.Ws + c * g.abstand;
    int sR = 0, sG = 0, sB = 0;
    for (int ry = 0; ry < g.B; ry++)
        for (int rx = 0; rx < g.B; rx++) {
            int32_t v = src[(size_t)ry * g.Ws + rx];
            sR += ch_r(v); sG += ch_g(v); sB += ch_b(v);
            pool_sum[(size_t)d * g.n + rx + ry * g.B] = (uint16_t)(ch_r(v) + ch_g(v) + ch_b(v));
        }
    int mR = sR >> g.lgn, mG = sG >> g.lgn, mB = sB >> g.lgn;
    int vR = 0, vG = 0;
    for (int ry = 0; ry < g.B; ry++)
        for (int rx = 0; rx < g.B; rx++) {
            int32_t v = src[(size_t)ry * g.Ws + rx];
            vR += (ch_r(v) - mR) * (ch_r(v) - mR);
            vG += (ch_g(v) - mG) * (ch_g(v) - mG);
        }
    FicRgbDomStat o;
    o.msum = mR + mG + mB;
    o.vD = (sR + sG + sB) - g.n * o.msum;                                   // varianzDomain after the loop (exact)
    o.varsq = __fadd_rn(__fadd_rn((float)vR, (float)vG), (float)mB);        // FC:776
    o.mR = mR; o.mG = mG; o.mB = mB;
    o.pad0 = 0; o.pad1 = 0;
    st[d] = o;
}

// getRangeblockRGB FC:564-577 + the range half of FC:767-786.  One thread per range block.
__global__ __launch_bounds__(256) void k_range_rgb(const int32_t* __restrict__ argb, int16_t* __restrict__ rng_t,
                                                   FicRgbRngStat* __restrict__ st, FicGeom g)
{
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= g.Nr) return;
    const int32_t* im = argb + (size_t)((j / g.Rw) * g.B) * g.W + (j % g.Rw) * g.B;
    int sR = 0, sG = 0, sB = 0;
    for (int y = 0; y < g.B; y++)
        for (int x = 0; x < g.B; x++) {
            int32_t v = im[(size_t)y * g.W + x];
            sR += ch_r(v); sG += ch_g(v); sB += ch_b(v);
        }
    FicRgbRngStat o;
    o.mR = sR >> g.lgn; o.mG = sG >> g.lgn; o.mB = sB >> g.lgn;
    int msum = o.mR + o.mG + o.mB;
    o.vR = (sR + sG + sB) - g.n * msum;                                      // varianzRange (exact)
    st[j] = o;
    for (int y = 0; y < g.B; y++)
        for (int x = 0; x < g.B; x++) {
            int32_t v = im[(size_t)y * g.W + x];
            rng_t[(size_t)j * g.n + x + y * g.B] = (int16_t)(ch_r(v) + ch_g(v) + ch_b(v) - msum);   // greyR_i
        }
}

// kovarianz of FC:781-792: sequential f32 accumulation of exact integer products
__device__ __forceinline__ float rgb_kovarianz(const int16_t* __restrict__ rt, const uint16_t* __restrict__ ps, int msum, int n)
{
    float kov = 0.0f;
    for (int i = 0; i < n; i++) kov = __fadd_rn(kov, (float)((int)rt[i] * ((int)ps[i] - msum)));
    return kov;
}

// getBestDomainblockRGB scan FC:702-715 + getErrorVarianceCovarianceRGB FC:794-803: one wave per range block
__global__ __launch_bounds__(256) void k_sweep_rgb(const uint16_t* __restrict__ pool_sum,
                                                   const FicRgbDomStat* __restrict__ dst,
                                                   const int16_t* __restrict__ rng_t,
                                                   const FicRgbRngStat* __restrict__ rst,
                                                   unsigned long long* __restrict__ key, FicGeom g)
{
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int j = blockIdx.x * 4 + wave;
    if (j >= g.Nr) return;
    FicRgbRngStat rs = rst[j];
    const int16_t* rt = rng_t + (size_t)j * g.n;
    float vRf = (float)rs.vR;
    int ncand = g.wK * g.wK;
    unsigned long long best = FIC_KEY_NONE;
    for (int c = lane; c < ncand; c += 64) {
        int gi = window_to_global(g, j, c);
        FicRgbDomStat ds = dst[gi];
        float kov = rgb_kovarianz(rt, pool_sum + (size_t)gi * g.n, ds.msum, g.n);
        float r = (rs.vR == 0 || ds.vD == 0) ? 0.0f : __fdiv_rn(kov, __fmul_rn(vRf, (float)ds.vD));
        r = __fmul_rn(r, r);
        float e = __fmul_rn(__fmul_rn(vRf, vRf), __fsub_rn(1.0f, r));
        unsigned long long kk = ((unsigned long long)f32_orderable(e) << 32) | (uint32_t)c;
        best = kk < best ? kk : best;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(best, off, 64);
        best = o < best ? o : best;
    }
    if (lane == 0) key[j] = best;
}

// getBestDomainblockRGB tail FC:717-734 + writeData RGB quantiser FC:250-254
__global__ __launch_bounds__(256) void k_finalize_rgb(const uint16_t* __restrict__ pool_sum,
                                                      const FicRgbDomStat* __restrict__ dst,
                                                      const int16_t* __restrict__ rng_t,
                                                      const FicRgbRngStat* __restrict__ rst,
                                                      const unsigned long long* __restrict__ key, FicRgbOutputs out, FicGeom g)
{
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= g.Nr) return;
    int c = (int)(uint32_t)key[j];
    int gi = window_to_global(g, j, c);
    FicRgbDomStat ds = dst[gi];
    FicRgbRngStat rs = rst[j];
    float kov = rgb_kovarianz(rng_t + (size_t)j * g.n, pool_sum + (size_t)gi * g.n, ds.msum, g.n);
    float a = __fdiv_rn(kov, ds.varsq);                 // FC:718
    if (a > 1.0f) a = 1.0f;                             // FC:721-724 (NaN passes)
    if (a < -1.0f) a = -1.0f;
    float bR = __fsub_rn((float)rs.mR, __fmul_rn(a, (float)ds.mR));
    float bG = __fsub_rn((float)rs.mG, __fmul_rn(a, (float)ds.mG));
    float bB = __fsub_rn((float)rs.mB, __fmul_rn(a, (float)ds.mB));
    out.qrows[5 * j + 0] = c;
    out.qrows[5 * j + 1] = java_f2i(__fmul_rn(a, 1000000.0f));
    out.qrows[5 * j + 2] = java_f2i(__fmul_rn(bR, 100000.0f));
    out.qrows[5 * j + 3] = java_f2i(__fmul_rn(bG, 100000.0f));
    out.qrows[5 * j + 4] = java_f2i(bB);
    const float qnan = __uint_as_float(0x7FC00000u);
    out.idx_local[j] = c;
    out.idx_global[j] = gi;
    out.a[j] = (a != a) ? qnan : a;
    out.bR[j] = (bR != bR) ? qnan : bR;
    out.bG[j] = (bG != bG) ? qnan : bG;
    out.bB[j] = (bB != bB) ? qnan : bB;
}

// getBestGeneratedCollageRGB FC:308-347.  One thread per pixel; domain pixel from the scaled image.
__global__ __launch_bounds__(256) void k_collage_rgb(const int32_t* __restrict__ scaled, FicRgbOutputs out,
                                                     int32_t* __restrict__ collage, FicGeom g)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.W) return;
    int j = (y / g.B) * g.Rw + (x / g.B);
    int rx = x % g.B, ry = y % g.B;
    int gi = out.idx_global[j];
    int c = gi % g.Dw, r = gi / g.Dw;
    int32_t d = scaled[(size_t)(r * g.abstand + ry) * g.Ws + c * g.abstand + rx];
    float a = out.a[j];
    int vR = java_f2i(__fadd_rn(__fmul_rn(a, (float)ch_r(d)), out.bR[j]));
    int vG = java_f2i(__fadd_rn(__fmul_rn(a, (float)ch_g(d)), out.bG[j]));
    int vB = java_f2i(__fadd_rn(__fmul_rn(a, (float)ch_b(d)), out.bB[j]));
    vR = vR < 0 ? 0 : (vR > 255 ? 255 : vR);
    vG = vG < 0 ? 0 : (vG > 255 ? 255 : vG);
    vB = vB < 0 ? 0 : (vB > 255 ? 255 : vB);
    collage[(size_t)y * g.W + x] = (int32_t)(0xff000000u | ((uint32_t)vR << 16) | ((uint32_t)vG << 8) | (uint32_t)vB);
}

// decodeRGB paint FC:463-499: a = q1/1000000f, bR = q2/100000f, bG = q3/100000f, bB = (float)q4 (FC:446-450);
// the squared change of the three channels is summed as an int per pixel before the float add (FC:493).
__global__ __launch_bounds__(256) void k_decode_paint_rgb(const int32_t* __restrict__ scaled, int32_t* __restrict__ image,
                                                          const int32_t* __restrict__ qrows5,
                                                          FicDecodeState* __restrict__ st, int counter, FicGeom g)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (st->done) return;
    unsigned long long sq = 0;
    if (x < g.W) {
        int j = (y / g.B) * g.Rw + (x / g.B);
        int rx = x % g.B, ry = y % g.B;
        int wloc = qrows5[5 * j + 0];
        float a = __fdiv_rn((float)qrows5[5 * j + 1], 1000000.0f);
        float bR = __fdiv_rn((float)qrows5[5 * j + 2], 100000.0f);
        float bG = __fdiv_rn((float)qrows5[5 * j + 3], 100000.0f);
        float bB = (float)qrows5[5 * j + 4];
        bool ok = wloc >= 0 && wloc < g.wK * g.wK;
        int gi = ok ? window_to_global(g, j, wloc) : 0;
        if (!ok || gi < 0 || gi >= g.Nd) {
            st->bad_index = 1;
        } else {
            int c = gi % g.Dw, r = gi / g.Dw;
            int32_t d = scaled[(size_t)(r * g.abstand + ry) * g.Ws + c * g.abstand + rx];
            int vR = java_f2i(__fadd_rn(__fmul_rn(a, (float)ch_r(d)), bR));
            int vG = java_f2i(__fadd_rn(__fmul_rn(a, (float)ch_g(d)), bG));
            int vB = java_f2i(__fadd_rn(__fmul_rn(a, (float)ch_b(d)), bB));
            vR = vR < 0 ? 0 : (vR > 255 ? 255 : vR);
            vG = vG < 0 ? 0 : (vG > 255 ? 255 : vG);
            vB = vB < 0 ? 0 : (vB > 255 ? 255 : vB);
            size_t p = (size_t)y * g.W + x;
            int32_t cur = image[p];
            int dR = ch_r(cur) - vR, dG = ch_g(cur) - vG, dB = ch_b(cur) - vB;
            image[p] = (int32_t)(0xff000000u | ((uint32_t)vR << 16) | ((uint32_t)vG << 8) | (uint32_t)vB);
            sq = (unsigned long long)(dR * dR + dG * dG + dB * dB);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    __shared__ unsigned long long s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (t) atomicAdd(&st->ssd[counter], t);
    }
}

// ---------------------------------------------------------------------------------------------
// k_sqrt_probe : test hook -- out[i] = sqrt((double)(first + i)) as the pool kernel computes it,
// so the test-suite can compare every possible variance value (0 .. 2^24) with the host's sqrt.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sqrt_probe(double* __restrict__ out, uint32_t first, uint32_t count)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = __dsqrt_rn((double)(first + i));
}

// =============================================================================================
// host-side launchers
// =============================================================================================
#define FIC_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e_ = hipGetLastError();         \
        if (e_ != hipSuccess) return (int)e_;      \
    } while (0)

int fic_launch_argb_to_gray(const int32_t* argb, uint8_t* gray, size_t npix, hipStream_t s)
{
    unsigned blocks = (unsigned)((npix + 255) / 256);
    hipLaunchKernelGGL(k_argb_to_gray, dim3(blocks), dim3(256), 0, s, argb, gray, npix);
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_scale(const uint8_t* gray, uint8_t* scaled, const FicGeom& g, hipStream_t s)
{
    hipLaunchKernelGGL(k_scale, dim3((g.Ws + 255) / 256, g.Hs, g.planes), dim3(256), 0, s, gray, scaled, g);
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_pool(const uint8_t* scaled, uint8_t* pool_pix, FicDomStat* st, uint32_t* var, double* s64,
                    const FicGeom& g, hipStream_t s)
{
    int per_blk = 256 / g.B;
    hipLaunchKernelGGL(k_pool, dim3((g.Nd + per_blk - 1) / per_blk, g.planes), dim3(256), 0, s, scaled, pool_pix, st,
                       var, s64, g);
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_range(const uint8_t* gray, uint32_t* rng_pix, FicRngStat* rst, const FicGeom& g, hipStream_t s)
{
    hipLaunchKernelGGL(k_range_stat, dim3((g.Nr_pad + 255) / 256, g.planes), dim3(256), 0, s, gray, rst, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_range_copies, dim3(g.tiles * g.NR, g.planes), dim3(256), 0, s, gray, rng_pix, g);
    FIC_LAUNCH_CHECK();
    return 0;
}

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
    A.tile0 = tile0; A.ntiles = ntiles; A.chunk_len = chunk_len; A.nchunks = nchunks;
    int NR, NC;
    if (fic_fast_variant(g.B, g.n_iso, &NR, &NC) || NR != g.NR) return -1;
    dim3 grid((ntiles + 3) / 4, nchunks * (g.n_iso / NC), g.planes);
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

int fic_launch_finalize(const FicBuffers& b, const FicOutputs& out, const FicGeom& g, int r_begin, int r_count,
                        hipStream_t s)
{
    hipLaunchKernelGGL(k_finalize, dim3((r_count + 255) / 256, g.planes), dim3(256), 0, s, (const uint32_t*)b.pool_pix,
                       b.pool_st, b.pool_var, b.rng_pix, b.rng_st, b.key, out, g, r_begin, r_count);
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_collage(const FicBuffers& b, const FicOutputs& out, int32_t* collage, const FicGeom& g, hipStream_t s)
{
    hipLaunchKernelGGL(k_collage, dim3((g.W + 255) / 256, g.H, g.planes), dim3(256), 0, s, b.pool_pix, out, collage, g);
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_sqrt_probe(double* out, uint32_t first, uint32_t count, hipStream_t s)
{
    hipLaunchKernelGGL(k_sqrt_probe, dim3((count + 255) / 256), dim3(256), 0, s, out, first, count);
    FIC_LAUNCH_CHECK();
    return 0;
}

// opt-in matrix-core sweep (B = 8, n_iso = 8): fragment prep + sweep
int fic_launch_mfma_prep_pool(const uint8_t* pool_pix, void* poolB, const FicGeom& g, int ndtiles_alloc, hipStream_t s)
{
    hipLaunchKernelGGL(k_pool_mfma, dim3((ndtiles_alloc * 128 + 255) / 256, g.planes), dim3(256), 0, s, pool_pix,
                       (v4i*)poolB, g, ndtiles_alloc);
    FIC_LAUNCH_CHECK();
    return 0;
}
int fic_launch_mfma_prep_range(const uint32_t* rng_pix, const FicRngStat* rng_st, void* rngA, int* rconst,
                               const FicGeom& g, hipStream_t s)
{
    int total = g.tiles * FIC_MFMA_RT * 128;
    hipLaunchKernelGGL(k_range_mfma, dim3((total + 255) / 256, g.planes), dim3(256), 0, s, rng_pix, rng_st, (v4i*)rngA,
                       rconst, g);
    FIC_LAUNCH_CHECK();
    return 0;
}
int fic_launch_sweep_mfma(const FicBuffers& b, const void* poolB, const void* rngA, const int* rconst, const FicGeom& g,
                          int group0, int ngroups_launch, int ndtiles, int ndtiles_alloc, int tiles_per_chunk, int nchunks,
                          hipStream_t s)
{
    MfmaArgs A;
    A.poolB = (const v4i*)poolB; A.pool_st = b.pool_st; A.pool_s64 = b.pool_s64;
    A.rngA = (const v4i*)rngA; A.rconst = rconst; A.key = b.key;
    A.Nd = g.Nd; A.Nd_pad = g.Nd_pad; A.Nr_pad = g.Nr_pad; A.lgn = g.lgn;
    A.ndtiles = ndtiles; A.ndtiles_alloc = ndtiles_alloc; A.ngroups = g.tiles;
    A.group0 = group0; A.tiles_per_chunk = tiles_per_chunk; A.nchunks = nchunks;
    hipLaunchKernelGGL(k_sweep_mfma, dim3(ngroups_launch, nchunks, g.planes), dim3(256), 0, s, A);
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
    dim3 grid((ct_end - ct_begin + CT - 1) / CT, nchunks, g.planes), block(256);
    if (g.B == 4) hipLaunchKernelGGL((k_sweep_mfma1<1>), grid, block, 0, s, A);
    else if (g.B == 8) hipLaunchKernelGGL((k_sweep_mfma1<2>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((k_sweep_mfma1<8>), grid, block, 0, s, A);
    FIC_LAUNCH_CHECK();
    return 0;
}

// joint-RGB encode, single image: scale, pool, ranges, sweep, finalise (+ collage when asked)
int fic_launch_rgb_encode(const FicRgbBuffers& b, const FicRgbOutputs& out, int32_t* collage, const FicGeom& g,
                          hipStream_t s)
{
    hipLaunchKernelGGL(k_scale_rgb, dim3((g.Ws + 255) / 256, g.Hs), dim3(256), 0, s, (const int32_t*)b.argb, b.scaled, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pool_rgb, dim3((g.Nd + 255) / 256), dim3(256), 0, s, (const int32_t*)b.scaled, b.pool_sum,
                       b.pool_st, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_range_rgb, dim3((g.Nr + 255) / 256), dim3(256), 0, s, (const int32_t*)b.argb, b.rng_t, b.rng_st, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sweep_rgb, dim3((g.Nr + 3) / 4), dim3(256), 0, s, (const uint16_t*)b.pool_sum,
                       (const FicRgbDomStat*)b.pool_st, (const int16_t*)b.rng_t, (const FicRgbRngStat*)b.rng_st, b.key, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize_rgb, dim3((g.Nr + 255) / 256), dim3(256), 0, s, (const uint16_t*)b.pool_sum,
                       (const FicRgbDomStat*)b.pool_st, (const int16_t*)b.rng_t, (const FicRgbRngStat*)b.rng_st,
                       (const unsigned long long*)b.key, out, g);
    FIC_LAUNCH_CHECK();
    if (collage) {
        hipLaunchKernelGGL(k_collage_rgb, dim3((g.W + 255) / 256, g.H), dim3(256), 0, s, (const int32_t*)b.scaled, out,
                           collage, g);
        FIC_LAUNCH_CHECK();
    }
    return 0;
}

// one decodeRGB iteration (FC:458-505): scaleImageRGB of the current image, paint, loop control
int fic_launch_decode_iteration_rgb(int32_t* scaled, int32_t* image, const int32_t* qrows5, FicDecodeState* state,
                                    int counter, const FicGeom& g, hipStream_t s)
{
    hipLaunchKernelGGL(k_scale_rgb, dim3((g.Ws + 255) / 256, g.Hs), dim3(256), 0, s, (const int32_t*)image, scaled, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_decode_paint_rgb, dim3((g.W + 255) / 256, g.H), dim3(256), 0, s, (const int32_t*)scaled, image,
                       qrows5, state, counter, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_decode_step, dim3(1), dim3(64), 0, s, state, counter, g.W * g.H, 1);
    FIC_LAUNCH_CHECK();
    return 0;
}

// one decoder iteration: scale (FC:382 -> createCodebuch -> scaleImage), paint, loop control
int fic_launch_decode_iteration(uint8_t* scaled, uint8_t* image, const int32_t* qrows, const int32_t* iso,
                                FicDecodeState* state, int counter, const FicGeom& g, hipStream_t s)
{
    hipLaunchKernelGGL(k_scale, dim3((g.Ws + 255) / 256, g.Hs, g.planes), dim3(256), 0, s, (const uint8_t*)image,
                       scaled, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_decode_paint, dim3((g.W + 255) / 256, g.H, g.planes), dim3(256), 0, s, scaled, image, qrows, iso,
                       state, counter, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_decode_step, dim3((g.planes + 63) / 64), dim3(64), 0, s, state, counter, g.W * g.H, g.planes);
    FIC_LAUNCH_CHECK();
    return 0;
}
