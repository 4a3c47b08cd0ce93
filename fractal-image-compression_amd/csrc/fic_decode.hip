// fic_decode.hip -- decodeGreyScale (FC:356-421) on the device: k_decode_paint + k_decode_step.
// gfx950 (MI355X / CDNA4) only, wave64.  Compile with -ffp-contract=off: every float expression must round once
// per operation exactly like the Java reference (FractalCompression.java = FC, Domainblock.java = DB).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "fic_device.h"
#include "fic_launch.h"
#include "fic_devfn.h"

// ---------------------------------------------------------------------------------------------
// Decoder, decodeGreyScale FC:356-421.  One iteration = k_scale (the pool of FC:382 is only ever
// read at the winning block, so domain pixels come straight from the scaled image: block g=(c,r)
// pixel (rx,ry) is scaled[(r*abstand+ry)*Ws + c*abstand+rx], FC:1027-1037) + k_decode_paint +
// k_decode_step.  The image is updated in place like FC:404: every thread reads its own old pixel
// (the 'range' of FC:391) and the scaled copy made before the paint.
//   qrows  : the ints of the .run stream per range {i_local, (int)(a*100), (int)b}  (FC:372-374)
//   a      = (float) q / 100f  (FC:373), b = (float) q
//   value  = clamp((int)(a*domain + b))                              (FC:396-402)
//   SSD    = sum (range - value)^2 as an exact integer.  Java accumulates it in float, pixel by pixel in
//            range-block order (FC:385-407): the same number whenever every partial sum is an exact float
//            (integer carry-in, total < 2^24).  Otherwise -- an iteration that changes a lot, a decode that
//            does not converge, images above 4096x4096 -- k_decode_step re-accumulates the per-pixel squares
//            (kept in `sqbuf` in Java's visiting order) exactly as FC:407 does, one float add per pixel in that order
//            (java_float_sum: a parallel scan over parity-dependent rounding maps, same bits as the sequential loop).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_decode_paint(const uint8_t* __restrict__ scaled, uint8_t* __restrict__ image,
                                                      const int32_t* __restrict__ qrows, const int32_t* __restrict__ iso,
                                                      FicDecodeState* __restrict__ state, uint32_t* __restrict__ sqbuf,
                                                      int counter, FicGeom g)
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
        // Java's visiting order: range blocks row-major, pixels row-major inside a block (FC:385-389)
        sqbuf[(size_t)plane * g.W * g.H + ((size_t)j * g.B + ry) * g.B + rx] = (uint32_t)sq;
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

// Java's avgError accumulation (FC:407), `avg += (float) d_i` for i = 0 .. count-1 in order, reproduced exactly by a whole
// workgroup.  While the sum stays inside one binade it is m * u (u = its ulp, m an integer) and adding the integer d = a*u + r
// (0 <= r < u) gives m + a + c with c = [r > u/2] or, on a tie r == u/2, the parity of m + a (round to nearest even): a map of
// m that depends only on m's parity.  Such maps compose associatively -- (A0, A1) = increment for an even / odd input -- so
// every thread folds its 64 consecutive values for both parities, an ordered tree over the workgroup composes the 1024 maps,
// and thread 0 applies the result to m.  In a block that would leave the binade (m' >= 2^24, at most ~17 times: the sum only
// grows) thread 0 applies the per-thread maps up to the run where that happens, adds that run in plain float and the scan
// restarts behind it; the head of the array is added in plain float while the carry-in still has a fractional part.
#define FIC_SUM_THREADS 1024
#define FIC_SUM_RUN 64
__device__ float java_float_sum(float carry, const uint32_t* __restrict__ d, int count)
{
    __shared__ uint32_t f0[FIC_SUM_THREADS], f1[FIC_SUM_THREADS];      // maps, composed in place by the tree
    __shared__ uint32_t g0[FIC_SUM_THREADS], g1[FIC_SUM_THREADS];      // the per-thread maps, kept for a block that leaves the binade
    __shared__ float s_sum;
    __shared__ int s_pos, s_k;
    __shared__ uint32_t s_m;
    const int t = threadIdx.x;
    if (t == 0) {
        float sum = carry;
        int i = 0;
        // fractional carry-in: plain adds until the sum is integer-valued (and the position 16-byte aligned for the vector loads)
        while (i < count && (sum != truncf(sum) || sum < 0.0f || (i & 3))) sum = __fadd_rn(sum, (float)d[i++]);
        s_sum = sum;
        s_pos = i;
    }
    __syncthreads();
    for (;;) {
        const int base = s_pos;
        if (base >= count) break;
        if (t == 0) {
            // binade of the (integer-valued) sum: u = 2^k, k = max(0, exponent - 23); m = sum / u exactly
            const float sum = s_sum;
            int k = 0;
            if (sum >= 16777216.0f) k = (int)((__float_as_uint(sum) >> 23) & 0xFF) - 127 - 23;
            s_k = k;
            s_m = (uint32_t)(sum * __uint_as_float((uint32_t)(127 - k) << 23));      // exact power-of-two scaling
        }
        __syncthreads();
        const int k = s_k;
        const uint32_t u_half = k > 0 ? 1u << (k - 1) : 0u, mask = k > 0 ? (1u << k) - 1u : 0u;
        uint32_t a0 = 0, a1 = 1;                      // running m (mod 2^32) for an even / odd start, kept as increments + parity
        {
            uint32_t m0 = 0, m1 = 1;                  // representatives with the right parity
            const int i0 = base + t * FIC_SUM_RUN;    // base is a multiple of 4 and d is 16-byte aligned: whole uint4 loads
            uint4 v4[FIC_SUM_RUN / 4];
#pragma unroll
            for (int j = 0; j < FIC_SUM_RUN / 4; j++)   // all loads in flight before the first use
                v4[j] = (i0 + 4 * j < count) ? *(const uint4*)(d + i0 + 4 * j) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < FIC_SUM_RUN; j++) {
                uint32_t v = (j & 3) == 0 ? v4[j >> 2].x : ((j & 3) == 1 ? v4[j >> 2].y : ((j & 3) == 2 ? v4[j >> 2].z : v4[j >> 2].w));
                if (i0 + j >= count) v = 0;
                const uint32_t a = v >> k, r = v & mask;      // (zeros past the end change nothing: a = r = 0)
                const uint32_t up = (k > 0 && r > u_half) ? 1u : 0u, tie = (k > 0 && r == u_half) ? 1u : 0u;
                m0 = m0 + a;
                m0 += up + (tie & m0 & 1u);
                m1 = m1 + a;
                m1 += up + (tie & m1 & 1u);
            }
            a0 = m0;                                  // increment for an even input
            a1 = m1 - 1u;                             // increment for an odd input
        }
        f0[t] = a0;
        f1[t] = a1;
        g0[t] = a0;
        g1[t] = a1;
        __syncthreads();
        for (int stride = 1; stride < FIC_SUM_THREADS; stride <<= 1) {
            if ((t & (2 * stride - 1)) == 0) {
                const uint32_t l0 = f0[t], l1 = f1[t], r0 = f0[t + stride], r1 = f1[t + stride];
                f0[t] = l0 + ((l0 & 1u) ? r1 : r0);                 // even input: parity after the left map = parity of l0
                f1[t] = l1 + (((1u + l1) & 1u) ? r1 : r0);          // odd input
            }
            __syncthreads();
        }
        if (t == 0) {
            const uint32_t m = s_m;
            int end = base + FIC_SUM_THREADS * FIC_SUM_RUN;
            if (end > count) end = count;
            const unsigned long long mn = (unsigned long long)m + ((m & 1u) ? f1[0] : f0[0]);
            if (mn < 16777216ull && (unsigned long long)(uint32_t)((m & 1u) ? f1[0] : f0[0]) == (mn - m)) {
                s_sum = (float)(uint32_t)mn * __uint_as_float((uint32_t)(127 + k) << 23);     // exact: m' < 2^24
            } else {
                // The sum leaves the binade inside this block.  The per-thread maps are valid up to the thread in whose
                // run that happens: apply them one by one, add that one run in plain float, and restart behind it (the
                // maps of the threads after it were folded with the old ulp).
                uint32_t mm = m;
                int tt = 0;
                for (; tt < FIC_SUM_THREADS; tt++) {
                    const unsigned long long nx = (unsigned long long)mm + ((mm & 1u) ? g1[tt] : g0[tt]);
                    if (nx >= 16777216ull) break;
                    mm = (uint32_t)nx;
                }
                float sum = (float)mm * __uint_as_float((uint32_t)(127 + k) << 23);
                int i = base + tt * FIC_SUM_RUN;
                end = i + FIC_SUM_RUN < end ? i + FIC_SUM_RUN : end;
                for (; i < end; i++) sum = __fadd_rn(sum, (float)d[i]);
                s_sum = sum;
            }
            s_pos = end;
        }
        __syncthreads();
    }
    return s_sum;
}

// Loop control of FC:413-417: one workgroup per plane.  avgError += ... (FC:407) is a float accumulation: fl(carry + SSD)
// when that is provably what Java gets, else java_float_sum over sqbuf.
__global__ __launch_bounds__(FIC_SUM_THREADS) void k_decode_step(FicDecodeState* __restrict__ state,
                                                                const uint32_t* __restrict__ sqbuf, int counter, int wh, int planes)
{
    const int plane = blockIdx.x;
    FicDecodeState* st = state + plane;
    if (st->done) return;                                      // uniform per workgroup
    const float carry = st->avg;                               // 0 after iteration 0; the static's old value before it (FC:20)
    const unsigned long long ssd = st->ssd[counter];
    const bool exact = carry == truncf(carry) && carry >= 0.0f && (double)carry + (double)ssd < 16777216.0;
    float avg;
    if (exact) avg = __fadd_rn(carry, (float)ssd);             // every partial sum is an exact float: order is irrelevant
    else avg = java_float_sum(carry, sqbuf + (size_t)plane * wh, wh);
    if (threadIdx.x != 0) return;
    if (!exact) st->seq_sums += 1;
    avg = __fdiv_rn(avg, (float)wh);                           // FC:413
    st->iters = counter + 1;
    st->avg_out = avg;
    if (avg < 1.0f) st->done = 1;                              // FC:414
    else st->avg = (counter != 49) ? 0.0f : avg;               // FC:416
    if (counter == 49) st->done = 1;
}

// test hook: out[0] = java_float_sum(carry, vals, count)
__global__ __launch_bounds__(FIC_SUM_THREADS) void k_float_sum_probe(float carry, const uint32_t* __restrict__ vals, int count,
                                                                    float* __restrict__ out)
{
    const float r = java_float_sum(carry, vals, count);
    if (threadIdx.x == 0) out[0] = r;
}
int fic_launch_float_sum_probe(float carry, const uint32_t* vals, int count, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_float_sum_probe, dim3(1), dim3(FIC_SUM_THREADS), 0, s, carry, vals, count, out);
    FIC_LAUNCH_CHECK();
    return 0;
}

// host-side launchers
// loop control of one decoder iteration (shared by the grey and the RGB decoder)
int fic_launch_decode_step(FicDecodeState* state, const uint32_t* sqbuf, int counter, int wh, int planes, hipStream_t s)
{
    hipLaunchKernelGGL(k_decode_step, dim3(planes), dim3(FIC_SUM_THREADS), 0, s, state, sqbuf, counter, wh, planes);
    FIC_LAUNCH_CHECK();
    return 0;
}

// one decoder iteration: scale (FC:382 -> createCodebuch -> scaleImage), paint, loop control
int fic_launch_decode_iteration(uint8_t* scaled, uint8_t* image, const int32_t* qrows, const int32_t* iso,
                                FicDecodeState* state, uint32_t* sqbuf, int counter, const FicGeom& g, hipStream_t s)
{
    if (fic_launch_scale(image, scaled, g, s)) return -1;      // FC:382 -> createCodebuch -> scaleImage
    hipLaunchKernelGGL(k_decode_paint, dim3((g.W + 255) / 256, g.H, g.planes), dim3(256), 0, s, scaled, image, qrows, iso,
                       state, sqbuf, counter, g);
    FIC_LAUNCH_CHECK();
    return fic_launch_decode_step(state, sqbuf, counter, g.W * g.H, g.planes, s);
}
