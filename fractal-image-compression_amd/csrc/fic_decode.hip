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
//            (kept in `sqbuf` in Java's visiting order) sequentially in f32, one add per pixel like FC:407.
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

// Loop control of FC:413-417: one wave per plane, lane 0 works.  avgError += ... (FC:407) is a float accumulation:
// fl(carry + SSD) when that is provably what Java gets, else the sequential sum over sqbuf.
__global__ __launch_bounds__(64) void k_decode_step(FicDecodeState* __restrict__ state, const uint32_t* __restrict__ sqbuf,
                                                   int counter, int wh, int planes)
{
    const int plane = blockIdx.x;
    if (plane >= planes || threadIdx.x != 0) return;
    FicDecodeState* st = state + plane;
    if (st->done) return;
    const float carry = st->avg;                               // 0 after iteration 0; the static's old value before it (FC:20)
    const unsigned long long ssd = st->ssd[counter];
    float avg;
    if (carry == truncf(carry) && carry >= 0.0f && (double)carry + (double)ssd < 16777216.0) {
        avg = __fadd_rn(carry, (float)ssd);                    // every partial sum is an exact float: order is irrelevant
    } else {
        const uint4* q = (const uint4*)(sqbuf + (size_t)plane * wh);      // wh is a multiple of 16
        avg = carry;
        for (int i = 0; i < wh / 4; i++) {
            const uint4 v = q[i];
            avg = __fadd_rn(avg, (float)v.x);
            avg = __fadd_rn(avg, (float)v.y);
            avg = __fadd_rn(avg, (float)v.z);
            avg = __fadd_rn(avg, (float)v.w);
        }
        st->seq_sums += 1;
    }
    avg = __fdiv_rn(avg, (float)wh);                           // FC:413
    st->iters = counter + 1;
    st->avg_out = avg;
    if (avg < 1.0f) st->done = 1;                              // FC:414
    else st->avg = (counter != 49) ? 0.0f : avg;               // FC:416
    if (counter == 49) st->done = 1;
}

// host-side launchers
// loop control of one decoder iteration (shared by the grey and the RGB decoder)
int fic_launch_decode_step(FicDecodeState* state, const uint32_t* sqbuf, int counter, int wh, int planes, hipStream_t s)
{
    hipLaunchKernelGGL(k_decode_step, dim3(planes), dim3(64), 0, s, state, sqbuf, counter, wh, planes);
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
