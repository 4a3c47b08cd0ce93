// fic_prep.hip -- pool build, range prep, finalise, collage.  The other translation units: fic_sweep.hip (sweeps),
// fic_mfma.hip (opt-in matrix-core sweeps), fic_decode.hip, fic_rgb.hip.  Overview of ALL kernels follows.
// gfx950 (MI355X / CDNA4) kernels for the bvk_ss19 grey encode hot path.
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
#include "fic_devfn.h"

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
// k_finalize : getBestDomainblock tail FC:634-642 + writeData quantiser FC:242-244.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(const uint32_t* __restrict__ pool_pix,
                                                  const FicDomStat* __restrict__ pool_st,
                                                  const uint32_t* __restrict__ pool_var,
                                                  const uint32_t* __restrict__ rng_pix,
                                                  const uint8_t* __restrict__ gray,     // non-null: no isometry copies were built
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
    if (gray) {
        // the winner's copy straight from the image: copy_k[pos] = r[iso_source(iso_inverse(k), pos)]  (affine gather, fic_devfn.h)
        const uint8_t* blk = gray + (size_t)plane * g.W * g.H + (size_t)((j / g.Rw) * g.B) * g.W + (j % g.Rw) * g.B;
        uint32_t s2;
        if (g.B == 4) iso_dot<4, false>(blk, g.W, k, pp, acc, s2);
        else if (g.B == 8) iso_dot<8, false>(blk, g.W, k, pp, acc, s2);
        else iso_dot<16, false>(blk, g.W, k, pp, acc, s2);
    } else {
        for (int dw = 0; dw < g.DW; dw++) acc = __builtin_amdgcn_udot4(rp[rng_word_index(g, j, k, dw)], pp[dw], acc, false);
    }
    finalize_store(out, g, plane, j, kk, wloc, gi, k, acc, ds, rs, pool_var[(size_t)plane * g.Nd_pad + gi]);
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
// k_sqrt_probe : test hook -- out[i] = sqrt((double)(first + i)) as the pool kernel computes it,
// so the test-suite can compare every possible variance value (0 .. 2^24) with the host's sqrt.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sqrt_probe(double* __restrict__ out, uint32_t first, uint32_t count)
{
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = __dsqrt_rn((double)(first + i));
}

// host-side launchers
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

int fic_launch_range(const uint8_t* gray, uint32_t* rng_pix, FicRngStat* rst, const FicGeom& g, hipStream_t s, int with_copies)
{
    hipLaunchKernelGGL(k_range_stat, dim3((g.Nr_pad + 255) / 256, g.planes), dim3(256), 0, s, gray, rst, g);
    FIC_LAUNCH_CHECK();
    if (!with_copies) return 0;                        // k_sweep_d4 works on its own slot store (k_range_d4)
    hipLaunchKernelGGL(k_range_copies, dim3(g.tiles * g.NR, g.planes), dim3(256), 0, s, gray, rng_pix, g);
    FIC_LAUNCH_CHECK();
    return 0;
}

int fic_launch_finalize(const FicBuffers& b, const FicOutputs& out, const FicGeom& g, int r_begin, int r_count,
                        hipStream_t s, int from_gray)
{
    hipLaunchKernelGGL(k_finalize, dim3((r_count + 255) / 256, g.planes), dim3(256), 0, s, (const uint32_t*)b.pool_pix,
                       b.pool_st, b.pool_var, b.rng_pix, from_gray ? (const uint8_t*)b.gray : (const uint8_t*)nullptr, b.rng_st,
                       b.key, out, g, r_begin, r_count);
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
