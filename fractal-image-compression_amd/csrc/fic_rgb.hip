// fic_rgb.hip -- the joint-RGB path: encodeRGB (FC:171-219) and decodeRGB (FC:430-508).
// gfx950 (MI355X / CDNA4) only, wave64.  Compile with -ffp-contract=off: every float expression must round once
// per operation exactly like the Java reference (FractalCompression.java = FC, Domainblock.java = DB).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "fic_device.h"
#include "fic_launch.h"
#include "fic_devfn.h"

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
    argb += (size_t)blockIdx.z * g.W * g.H;                               // blockIdx.z = image of the batch
    scaled += (size_t)blockIdx.z * g.Ws * g.Hs;
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
    scaled += (size_t)blockIdx.y * g.Ws * g.Hs;                           // blockIdx.y = image of the batch
    pool_sum += (size_t)blockIdx.y * g.Nd * g.n;
    st += (size_t)blockIdx.y * g.Nd;
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
    argb += (size_t)blockIdx.y * g.W * g.H;                               // blockIdx.y = image of the batch
    rng_t += (size_t)blockIdx.y * g.Nr * g.n;
    st += (size_t)blockIdx.y * g.Nr;
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
    pool_sum += (size_t)blockIdx.y * g.Nd * g.n;                          // blockIdx.y = image of the batch
    dst += (size_t)blockIdx.y * g.Nd;
    rng_t += (size_t)blockIdx.y * g.Nr * g.n;
    rst += (size_t)blockIdx.y * g.Nr;
    key += (size_t)blockIdx.y * g.Nr;
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

// ---------------------------------------------------------------------------------------------
// Full search (wK == Dw == Dh), B = 4 / 8: the k_sweep_fast mapping for the joint-RGB error.
// k_sweep_rgb gives every lane its own candidate, so a wave-level load touches 64 different pool blocks (2-byte
// reads, 128-byte stride): fine for the GUI's small windows, 8.8e9 pairs/s on a full pool.  Here lane = range block
// (its n greyR_i sit in VGPRs as exact f32 integers) and the domain block is wave-uniform: greyD_i = (R+G+B)_i - msum
// comes from a pre-centred f32 copy of the pool through the scalar cache and feeds v_mul_f32 as an SGPR operand.
// The arithmetic per pair is the reference's, in the reference's order: kov += greyR_i * greyD_i for i = 0..n-1
// (exact product < 2^24, one f32 rounding per add, FC:781-792), then FC:794-803.  Pool chunks run in different
// workgroups and meet in the (error, candidate) key through a 64-bit atomicMin, as in the grey sweeps.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pool_rgb_centred(const uint16_t* __restrict__ pool_sum, const FicRgbDomStat* __restrict__ st,
                                                          float* __restrict__ pool_cf, FicGeom g)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)g.Nd * g.n) return;
    pool_sum += (size_t)blockIdx.y * g.Nd * g.n;                          // blockIdx.y = image of the batch
    pool_cf += (size_t)blockIdx.y * g.Nd * g.n;
    st += (size_t)blockIdx.y * g.Nd;
    pool_cf[i] = (float)((int)pool_sum[i] - st[i >> g.lgn].msum);           // greyD_i, |.| <= 765: exact
}

template <int N>
__global__ __launch_bounds__(256) void k_sweep_rgb_fast(const float* __restrict__ pool_cf, const FicRgbDomStat* __restrict__ dst,
                                                        const int16_t* __restrict__ rng_t, const FicRgbRngStat* __restrict__ rst,
                                                        unsigned long long* __restrict__ key, FicGeom g, int chunk_len)
{
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);                  // 64 range blocks per wave
    if (tile * 64 >= g.Nr) return;
    pool_cf += (size_t)blockIdx.z * g.Nd * N;                             // blockIdx.z = image of the batch
    dst += (size_t)blockIdx.z * g.Nd;
    rng_t += (size_t)blockIdx.z * g.Nr * N;
    rst += (size_t)blockIdx.z * g.Nr;
    key += (size_t)blockIdx.z * g.Nr;
    const int j = tile * 64 + lane;
    const int jj = j < g.Nr ? j : g.Nr - 1;                                // tail lanes shadow the last block
    const int d0 = blockIdx.y * chunk_len;
    int d1 = d0 + chunk_len;
    if (d1 > g.Nd) d1 = g.Nd;
    float rt[N];
#pragma unroll
    for (int i = 0; i < N; i++) rt[i] = (float)rng_t[(size_t)jj * N + i];  // greyR_i, |.| <= 765: exact
    const FicRgbRngStat rs = rst[jj];
    const float vRf = (float)rs.vR;
    const float vR2 = __fmul_rn(vRf, vRf);
    const AS4 float* pc = (const AS4 float*)pool_cf;
    const AS4 FicRgbDomStat* ds4 = (const AS4 FicRgbDomStat*)dst;
    unsigned long long best = FIC_KEY_NONE;
    for (int d = d0; d < d1; d++) {
        const AS4 float* gd = pc + (size_t)d * N;
        float kov = 0.0f;
#pragma unroll
        for (int i = 0; i < N; i++) kov = __fadd_rn(kov, __fmul_rn(rt[i], gd[i]));
        const int vD = ds4[d].vD;
        float r = (rs.vR == 0 || vD == 0) ? 0.0f : __fdiv_rn(kov, __fmul_rn(vRf, (float)vD));   // FC:795-800
        r = __fmul_rn(r, r);
        const float e = __fmul_rn(vR2, __fsub_rn(1.0f, r));                                      // FC:801-802
        const unsigned long long kk = ((unsigned long long)f32_orderable(e) << 32) | (uint32_t)d;
        best = kk < best ? kk : best;
    }
    if (j < g.Nr && best != FIC_KEY_NONE) atomicMin(&key[j], best);
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
    {
        const size_t p = blockIdx.y;                                      // image of the batch
        pool_sum += p * g.Nd * g.n; dst += p * g.Nd; rng_t += p * g.Nr * g.n; rst += p * g.Nr; key += p * g.Nr;
        out.idx_local += p * g.Nr; out.idx_global += p * g.Nr; out.a += p * g.Nr; out.bR += p * g.Nr; out.bG += p * g.Nr;
        out.bB += p * g.Nr; out.qrows += p * g.Nr * 5;
    }
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
    {
        const size_t p = blockIdx.z;                                      // image of the batch
        scaled += p * g.Ws * g.Hs; collage += p * g.W * g.H;
        out.idx_global += p * g.Nr; out.a += p * g.Nr; out.bR += p * g.Nr; out.bG += p * g.Nr; out.bB += p * g.Nr;
    }
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
                                                          FicDecodeState* __restrict__ st, uint32_t* __restrict__ sqbuf,
                                                          int counter, FicGeom g)
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
        sqbuf[((size_t)j * g.B + ry) * g.B + rx] = (uint32_t)sq;     // Java's visiting order (FC:461-466)
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

// host-side launchers
// joint-RGB encode of g.planes images (buffers: image p at p x the per-image size): scale, pool, ranges, sweep, finalise
// (+ collage when asked).  Every kernel takes the image from the grid, so a batch costs one launch per stage, not per image;
// only the matrix-core full search runs image by image (its fragment buffers hold one image; it is the choice for big ones).
int fic_launch_rgb_encode(const FicRgbBuffers& b, const FicRgbOutputs& out, int32_t* collage, const FicGeom& g,
                          hipStream_t s, const FicRgbQ* q)
{
    const unsigned P = (unsigned)g.planes;
    hipLaunchKernelGGL(k_scale_rgb, dim3((g.Ws + 255) / 256, g.Hs, P), dim3(256), 0, s, (const int32_t*)b.argb, b.scaled, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pool_rgb, dim3((g.Nd + 255) / 256, P), dim3(256), 0, s, (const int32_t*)b.scaled, b.pool_sum,
                       b.pool_st, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_range_rgb, dim3((g.Nr + 255) / 256, P), dim3(256), 0, s, (const int32_t*)b.argb, b.rng_t, b.rng_st, g);
    FIC_LAUNCH_CHECK();
    if (g.full && q && q->poolQ) {
        // full search on the matrix cores (fic_q.hip, k_sweep_q<NK, 3>): the MFMA output prunes, flagged pairs are evaluated
        // with the reference's sequential f32 sums
        FicGeom g1 = g;
        g1.planes = 1;
        for (size_t p = 0; p < P; p++)
            if (fic_launch_rgbq(b.pool_sum + p * g.Nd * g.n, b.pool_st + p * g.Nd, b.rng_t + p * g.Nr * g.n, b.rng_st + p * g.Nr,
                                b.key + p * g.Nr, q->poolQ, q->dflat, q->rngQ, q->qst, q->rngE, q->theta_g, q->amax, g1, q->ndtiles,
                                q->ndtiles_alloc, q->nct_alloc, q->tiles_per_chunk, q->nchunks, s))
                return (int)hipErrorUnknown;
    } else if (g.full && g.B <= 8 && b.pool_cf) {
        // full search: lane = range block, wave-uniform domain blocks, pool chunks across workgroups
        const size_t total = (size_t)g.Nd * g.n;
        hipLaunchKernelGGL(k_pool_rgb_centred, dim3((unsigned)((total + 255) / 256), P), dim3(256), 0, s,
                           (const uint16_t*)b.pool_sum, (const FicRgbDomStat*)b.pool_st, b.pool_cf, g);
        FIC_LAUNCH_CHECK();
        if (hipMemsetAsync(b.key, 0xFF, (size_t)P * g.Nr * sizeof(unsigned long long), s) != hipSuccess) return (int)hipErrorUnknown;
        const int tiles = (g.Nr + 63) / 64;
        int nchunks = (8192 + tiles * (int)P - 1) / (tiles * (int)P);      // ~8 waves per SIMD
        if (nchunks > g.Nd / 64) nchunks = g.Nd / 64;
        if (nchunks < 1) nchunks = 1;
        const int chunk_len = (g.Nd + nchunks - 1) / nchunks;
        nchunks = (g.Nd + chunk_len - 1) / chunk_len;
        dim3 grid((tiles + 3) / 4, nchunks, P);
        if (g.B == 4)
            hipLaunchKernelGGL((k_sweep_rgb_fast<16>), grid, dim3(256), 0, s, (const float*)b.pool_cf, (const FicRgbDomStat*)b.pool_st,
                               (const int16_t*)b.rng_t, (const FicRgbRngStat*)b.rng_st, b.key, g, chunk_len);
        else
            hipLaunchKernelGGL((k_sweep_rgb_fast<64>), grid, dim3(256), 0, s, (const float*)b.pool_cf, (const FicRgbDomStat*)b.pool_st,
                               (const int16_t*)b.rng_t, (const FicRgbRngStat*)b.rng_st, b.key, g, chunk_len);
    } else {
        hipLaunchKernelGGL(k_sweep_rgb, dim3((g.Nr + 3) / 4, P), dim3(256), 0, s, (const uint16_t*)b.pool_sum,
                           (const FicRgbDomStat*)b.pool_st, (const int16_t*)b.rng_t, (const FicRgbRngStat*)b.rng_st, b.key, g);
    }
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finalize_rgb, dim3((g.Nr + 255) / 256, P), dim3(256), 0, s, (const uint16_t*)b.pool_sum,
                       (const FicRgbDomStat*)b.pool_st, (const int16_t*)b.rng_t, (const FicRgbRngStat*)b.rng_st,
                       (const unsigned long long*)b.key, out, g);
    FIC_LAUNCH_CHECK();
    if (collage) {
        hipLaunchKernelGGL(k_collage_rgb, dim3((g.W + 255) / 256, g.H, P), dim3(256), 0, s, (const int32_t*)b.scaled, out,
                           collage, g);
        FIC_LAUNCH_CHECK();
    }
    return 0;
}

// one decodeRGB iteration (FC:458-505): scaleImageRGB of the current image, paint, loop control
int fic_launch_decode_iteration_rgb(int32_t* scaled, int32_t* image, const int32_t* qrows5, FicDecodeState* state,
                                    uint32_t* sqbuf, int counter, const FicGeom& g, hipStream_t s)
{
    hipLaunchKernelGGL(k_scale_rgb, dim3((g.Ws + 255) / 256, g.Hs), dim3(256), 0, s, (const int32_t*)image, scaled, g);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_decode_paint_rgb, dim3((g.W + 255) / 256, g.H), dim3(256), 0, s, (const int32_t*)scaled, image,
                       qrows5, state, sqbuf, counter, g);
    FIC_LAUNCH_CHECK();
    return fic_launch_decode_step(state, sqbuf, counter, g.W * g.H, 1, s);
}
