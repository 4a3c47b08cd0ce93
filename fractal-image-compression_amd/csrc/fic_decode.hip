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
// One thread = one pixel ROW of one range block (B pixels): the codebook row, its fit and its domain block are looked up once
// per B pixels, the image row is one B-byte load + store, the squares one contiguous run of `sqbuf`.  Threads are ordered
// (block row, pixel row, block column): neighbours in a wave touch neighbouring bytes of the same image row.
template <int B>
__global__ __launch_bounds__(256) void k_decode_paint(const uint8_t* __restrict__ scaled, uint8_t* __restrict__ image,
                                                      const int32_t* __restrict__ qrows, const int32_t* __restrict__ iso,
                                                      FicDecodeState* __restrict__ state, uint32_t* __restrict__ sqbuf,
                                                      int counter, FicGeom g)
{
    const int plane = blockIdx.y;
    FicDecodeState* st = state + plane;
    if (st->done) return;                                  // uniform per plane
    const int t = blockIdx.x * 256 + threadIdx.x;          // (by * B + ry) * Rw + bx
    unsigned long long sq = 0;
    if (t < g.Nr * B) {
        const int bx = t % g.Rw, row = t / g.Rw;           // row = by * B + ry = the image row
        const int ry = row % B, by = row / B;
        const int j = by * g.Rw + bx;
        const size_t o = (size_t)plane * g.Nr + j;
        const int wloc = qrows[3 * o + 0];
        const float a = __fdiv_rn((float)qrows[3 * o + 1], 100.0f);
        const float b = (float)qrows[3 * o + 2];
        const bool ok = wloc >= 0 && wloc < g.wK * g.wK;
        const int gi = ok ? window_to_global(g, j, wloc) : 0;
        uint8_t* prow = image + (size_t)plane * g.W * g.H + (size_t)row * g.W + bx * B;
        uint32_t* srow = sqbuf + (size_t)plane * g.W * g.H + ((size_t)j * B + ry) * B;   // Java's visiting order (FC:385-389)
        uint32_t old[B / 4], neu[B / 4], sqv[B];
        __builtin_memcpy(old, prow, B);
        if (!ok || gi < 0 || gi >= g.Nd) {
            st->bad_index = 1;                             // Java: ArrayIndexOutOfBoundsException at FC:394
#pragma unroll
            for (int x = 0; x < B; x++) sqv[x] = 0u;
        } else {
            const int c = gi % g.Dw, r = gi / g.Dw;
            const uint8_t* dom = scaled + (size_t)plane * g.Ws * g.Hs + (size_t)(r * g.abstand) * g.Ws + c * g.abstand;
            uint8_t dpx[B];
            const int k = iso ? iso[o] : 0;
            if (k == 0) {
                __builtin_memcpy(dpx, dom + (size_t)ry * g.Ws, B);       // the block's row ry: B consecutive scaled pixels
            } else {
                int ax, bx_, cx, ay, by_, cy;
                iso_affine(k, B - 1, ax, bx_, cx, ay, by_, cy);          // source (sx, sy) = (ax x + bx y + cx, ay x + by y + cy)
                const int s0 = (cy + by_ * ry) * g.Ws + cx + bx_ * ry, sx = ay * g.Ws + ax;
#pragma unroll
                for (int x = 0; x < B; x++) dpx[x] = dom[s0 + sx * x];
            }
#pragma unroll
            for (int q = 0; q < B / 4; q++) neu[q] = 0u;
#pragma unroll
            for (int x = 0; x < B; x++) {
                int value = java_f2i(__fadd_rn(__fmul_rn(a, (float)dpx[x]), b));
                value = value < 0 ? 0 : (value > 255 ? 255 : value);
                const int dd = (int)((old[x >> 2] >> (8 * (x & 3))) & 0xffu) - value;
                neu[x >> 2] |= (uint32_t)value << (8 * (x & 3));
                sqv[x] = (uint32_t)(dd * dd);
                sq += sqv[x];
            }
            __builtin_memcpy(prow, neu, B);
        }
#pragma unroll
        for (int q = 0; q < B / 4; q++) *(uint4*)(srow + 4 * q) = make_uint4(sqv[4 * q], sqv[4 * q + 1], sqv[4 * q + 2], sqv[4 * q + 3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    __shared__ unsigned long long s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (tot) atomicAdd(&st->ssd[counter], tot);
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
#define FIC_SUM_SEG (FIC_SUM_THREADS * FIC_SUM_RUN)     // values one workgroup folds per round = one segment of the multi-workgroup scan
#define FIC_SUM_SAT 16777216u                            // increments are kept up to 2^24: anything larger leaves the binade anyway
#define FIC_SUM_KMAPS 20                                 // binades k = 0 .. 19 (ulp 2^k): from k = 20 on, half an ulp (2^19) exceeds every addend
                                                         //   (a squared change is at most 3 * 255^2 = 195075), so the sum cannot move
// (A0, A1) then (r0, r1): the map "first l, then r" -- increments for an even / odd input, saturating at FIC_SUM_SAT (a saturated
// map stays saturated: its parity is meaningless, and so is everything composed behind it)
__device__ __forceinline__ void map_compose(uint32_t& l0, uint32_t& l1, uint32_t r0, uint32_t r1)
{
    const uint32_t n0 = l0 + ((l0 & 1u) ? r1 : r0);                 // even input: parity after the left map = parity of l0
    const uint32_t n1 = l1 + (((1u + l1) & 1u) ? r1 : r0);          // odd input
    l0 = (l0 >= FIC_SUM_SAT || n0 >= FIC_SUM_SAT) ? FIC_SUM_SAT : n0;   // (all operands <= 2^24: no 32-bit wrap)
    l1 = (l1 >= FIC_SUM_SAT || n1 >= FIC_SUM_SAT) ? FIC_SUM_SAT : n1;
}
// The map of FIC_SUM_RUN consecutive values (in registers) for binade k, as increments for an even / odd start.
__device__ __forceinline__ void run_maps(const uint4 (&v4)[FIC_SUM_RUN / 4], int k, uint32_t& a0, uint32_t& a1)
{
    const uint32_t u_half = k > 0 ? 1u << (k - 1) : 0u, mask = k > 0 ? (1u << k) - 1u : 0u;
    uint32_t m0 = 0, m1 = 1;                      // representatives with the right parity
#pragma unroll
    for (int j = 0; j < FIC_SUM_RUN; j++) {
        const uint32_t v = (j & 3) == 0 ? v4[j >> 2].x : ((j & 3) == 1 ? v4[j >> 2].y : ((j & 3) == 2 ? v4[j >> 2].z : v4[j >> 2].w));
        const uint32_t a = v >> k, r = v & mask;      // (zeros past the end change nothing: a = r = 0)
        const uint32_t up = (k > 0 && r > u_half) ? 1u : 0u, tie = (k > 0 && r == u_half) ? 1u : 0u;
        m0 = m0 + a;
        m0 += up + (tie & m0 & 1u);
        m1 = m1 + a;
        m1 += up + (tie & m1 & 1u);
    }
    a0 = m0 < FIC_SUM_SAT ? m0 : FIC_SUM_SAT;     // (64 values below 2^18 each: no wrap before the clamp)
    a1 = m1 - 1u < FIC_SUM_SAT ? m1 - 1u : FIC_SUM_SAT;
}
__device__ float java_float_sum(float carry, const uint32_t* __restrict__ d, int count)
{
    __shared__ uint32_t pm[FIC_SUM_THREADS];                           // m after the runs 0 .. t (saturated): the inclusive scan applied to m
    __shared__ uint32_t wv0[FIC_SUM_THREADS / 64], wv1[FIC_SUM_THREADS / 64];
    __shared__ uint32_t stage[FIC_SUM_THREADS];                        // values staged for thread 0's plain float adds
    __shared__ float s_sum;
    __shared__ int s_pos, s_k, s_tt, s_more;
    __shared__ uint32_t s_m;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // Head: while the carry-in still has a fractional part (or the position is not 16-byte aligned) the adds are plain float adds
    // in order -- by thread 0, out of LDS, a staged chunk at a time (a dependent global load per add costs a microsecond each,
    // and a fractional carry lives until the sum passes 2^23: thousands of adds).
    if (t == 0) { s_sum = carry; s_pos = 0; s_more = 1; }
    __syncthreads();
    while (s_more) {                                                    // uniform
        const int p0 = s_pos;
        stage[t] = p0 + t < count ? d[p0 + t] : 0u;
        __syncthreads();
        if (t == 0) {
            float sum = s_sum;
            int i = p0;
            const int lim = p0 + FIC_SUM_THREADS < count ? p0 + FIC_SUM_THREADS : count;
            while (i < lim && (sum != truncf(sum) || sum < 0.0f || (i & 3))) sum = __fadd_rn(sum, (float)stage[i++ - p0]);
            s_sum = sum;
            s_pos = i;
            s_more = (i == lim && i < count && (sum != truncf(sum) || sum < 0.0f)) ? 1 : 0;   // chunk used up, still fractional
        }
        __syncthreads();
    }
    for (;;) {
        const int base = s_pos;
        if (base >= count) break;
        if (t == 0) {
            // binade of the (integer-valued) sum: u = 2^k, k = max(0, exponent - 23); m = sum / u exactly
            const float sum = s_sum;
            int k = 0;
            if (sum >= 16777216.0f) k = (int)((__float_as_uint(sum) >> 23) & 0xFF) - 127 - 23;
            s_k = k;
            s_m = k < FIC_SUM_KMAPS ? (uint32_t)(sum * __uint_as_float((uint32_t)(127 - k) << 23)) : 0u;   // exact power-of-two scaling
            s_tt = FIC_SUM_THREADS;
        }
        __syncthreads();
        const int k = s_k;
        if (k >= FIC_SUM_KMAPS) break;                // (uniform) half an ulp exceeds every addend: the sum no longer moves (FC:407)
        uint32_t a0 = 0, a1 = 0;                      // increments for an even / odd start
        {
            const int i0 = base + t * FIC_SUM_RUN;    // base is a multiple of 4 and d is 16-byte aligned: whole uint4 loads
            uint4 v4[FIC_SUM_RUN / 4];
#pragma unroll
            for (int j = 0; j < FIC_SUM_RUN / 4; j++) {  // all loads in flight before the first use; values past the end read as 0
                const int i = i0 + 4 * j;
                uint4 v = (i < count) ? *(const uint4*)(d + i) : make_uint4(0, 0, 0, 0);
                if (i + 1 >= count) v.y = 0;
                if (i + 2 >= count) v.z = 0;
                if (i + 3 >= count) v.w = 0;
                v4[j] = v;
            }
            run_maps(v4, k, a0, a1);
        }
        // ordered INCLUSIVE scan of the 1024 run maps: thread t ends with the map of the runs 0 .. t (lanes of a wave by
        // shuffles, then the 16 wave totals)
#pragma unroll
        for (int stride = 1; stride < 64; stride <<= 1) {
            uint32_t l0 = __shfl_up(a0, stride, 64), l1 = __shfl_up(a1, stride, 64);
            if (lane >= stride) { map_compose(l0, l1, a0, a1); a0 = l0; a1 = l1; }
        }
        if (lane == 63) { wv0[wave] = a0; wv1[wave] = a1; }
        __syncthreads();
        if (wave > 0) {
            uint32_t l0 = wv0[0], l1 = wv1[0];
            for (int w = 1; w < wave; w++) map_compose(l0, l1, wv0[w], wv1[w]);
            map_compose(l0, l1, a0, a1);
            a0 = l0;
            a1 = l1;
        }
        const uint32_t m = s_m;
        const uint32_t inc = (m & 1u) ? a1 : a0;                        // saturated at 2^24: m + inc < 2^24 only for true increments
        const bool over = inc >= FIC_SUM_SAT || m + inc >= FIC_SUM_SAT;
        pm[t] = over ? FIC_SUM_SAT : m + inc;
        if (over) atomicMin(&s_tt, t);                                  // the first run in which the sum leaves the binade
        __syncthreads();
        const int tt = s_tt;
        int end = base + FIC_SUM_THREADS * FIC_SUM_RUN;
        if (end > count) end = count;
        if (tt < FIC_SUM_THREADS) {
            // The sum leaves the binade inside run tt: the maps are valid up to the run before it; that one run is added in plain
            // float (staged through LDS), and the scan restarts behind it with the new ulp.
            const int r0 = base + tt * FIC_SUM_RUN;
            if (t < FIC_SUM_RUN) stage[t] = r0 + t < count ? d[r0 + t] : 0u;
            __syncthreads();
            if (t == 0) {
                float sum = (float)(tt == 0 ? m : pm[tt - 1]) * __uint_as_float((uint32_t)(127 + k) << 23);
                const int rend = r0 + FIC_SUM_RUN < end ? r0 + FIC_SUM_RUN : end;
                for (int i = r0; i < rend; i++) sum = __fadd_rn(sum, (float)stage[i - r0]);
                s_sum = sum;
                s_pos = rend;
            }
        } else if (t == 0) {
            s_sum = (float)pm[FIC_SUM_THREADS - 1] * __uint_as_float((uint32_t)(127 + k) << 23);     // exact: m' < 2^24
            s_pos = end;
        }
        __syncthreads();
    }
    return s_sum;
}

// ---------------------------------------------------------------------------------------------
// The same sum across the whole chip (round 3; round 2 walked the 16.8 M squares of a 4096x4096 iteration through ONE
// workgroup, 256 rounds of 65 536: 4.4 ms for 67 MB).  The maps of java_float_sum compose associatively, so:
//   k_float_sum_maps   every workgroup folds ONE segment of 65 536 values into its map (A0, A1) -- for EVERY binade k the sum can
//                      be in while it crosses the segment (k = 0 .. kmax, kmax from the exact integer total): which binade that
//                      will be is not known in advance, because the float sum may lag far behind the exact prefix (addends below
//                      half an ulp leave it where it is);
//   float_sum_walk     one workgroup walks the segments in order: binade and m of the running sum, the segment's map for that
//                      binade, done -- unless the sum would leave the binade inside the segment (m + increment >= 2^24; the sum
//                      only grows, so that happens at most ~20 times) or still has a fractional part: those segments, and only
//                      those, go through java_float_sum.
// maps: u32 [segments][FIC_SUM_KMAPS][2].
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int float_sum_segments(int count) { return (count + FIC_SUM_SEG - 1) / FIC_SUM_SEG; }
// Scratch of one sum behind the values: segment maps u32 [segments][KMAPS][2], the segments' exact totals u64 [segments], and the
// maps of the 16 sub-segments (one wave's 4096 values each) of every segment u32 [segments][KMAPS][16][2]
#define FIC_SUM_SUBS (FIC_SUM_THREADS / 64)
#define FIC_SUM_SUB (64 * FIC_SUM_RUN)
__host__ __device__ inline size_t float_sum_scratch_words(int count)
{
    return (size_t)float_sum_segments(count) * (FIC_SUM_KMAPS * 2 + 2 + FIC_SUM_KMAPS * FIC_SUM_SUBS * 2);
}
__device__ __forceinline__ unsigned long long* float_sum_totals(uint32_t* scratch, int count)
{
    return (unsigned long long*)(scratch + (size_t)float_sum_segments(count) * FIC_SUM_KMAPS * 2);
}
__device__ __forceinline__ uint32_t* float_sum_submaps(uint32_t* scratch, int count)
{
    return scratch + (size_t)float_sum_segments(count) * (FIC_SUM_KMAPS * 2 + 2);
}
// binade (ulp exponent) of a non-negative value: 0 below 2^24
__device__ __forceinline__ int float_sum_binade(double v)
{
    int k = 0;
    while (k < FIC_SUM_KMAPS - 1 && v >= 16777216.0 * (double)(1ull << k)) k++;
    return k;
}
// exact integer total of one segment (u64), one workgroup of 256 threads per segment
__device__ void float_sum_segment_total(const uint32_t* __restrict__ d, int count, int seg, unsigned long long* __restrict__ out)
{
    __shared__ unsigned long long part[4];
    const int t = threadIdx.x;
    unsigned long long acc = 0;
    const int i0 = seg * FIC_SUM_SEG;
    for (int i = i0 + 4 * t; i < i0 + FIC_SUM_SEG && i < count; i += 4 * 256) {
        const uint4 v = *(const uint4*)(d + i);
        acc += (unsigned long long)v.x + (i + 1 < count ? v.y : 0u);
        acc += (unsigned long long)(i + 2 < count ? v.z : 0u) + (i + 3 < count ? v.w : 0u);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((t & 63) == 0) part[t >> 6] = acc;
    __syncthreads();
    if (t == 0) out[seg] = part[0] + part[1] + part[2] + part[3];
}
// The maps of segment `seg` for the binades the running sum can be in while it crosses the segment; every other binade's
// entry is "saturated" (the walk then takes the sequential-order path, which is always right).  Which binades: the float
// sum F after i addends of exact total P_i obeys (carry + P_i) / f <= F <= f (carry + P_i) with f = (1 + 2^-24)^count --
// every add rounds to nearest, relative error <= 2^-24 either way -- so with the exact totals of the segments before this
// one (segtot) the sum enters the segment no lower than binade(lo / f) and leaves it no higher than binade(f hi).
__device__ void float_sum_segment_maps(const uint32_t* __restrict__ d, int count, int seg, float carry, float f,
                                       const unsigned long long* __restrict__ segtot, uint32_t* __restrict__ maps,
                                       uint32_t* __restrict__ submaps)
{
    uint32_t* const sub = submaps + (size_t)seg * FIC_SUM_KMAPS * FIC_SUM_SUBS * 2;        // [k][wave][2]
    __shared__ uint32_t w0[FIC_SUM_THREADS / 64], w1[FIC_SUM_THREADS / 64];
    __shared__ unsigned long long pre[FIC_SUM_THREADS / 64];
    __shared__ int s_klo, s_khi;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    {
        unsigned long long acc = 0;                             // exact total of the segments before this one
        for (int i = t; i < seg; i += FIC_SUM_THREADS) acc += segtot[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) pre[wave] = acc;
        __syncthreads();
        if (t == 0) {
            unsigned long long before = 0;
            for (int w = 0; w < FIC_SUM_THREADS / 64; w++) before += pre[w];
            const double c = carry > 0.0f ? (double)carry : 0.0, ff = (double)f;
            int klo = float_sum_binade(((double)before + c) / ff - 1.0);
            int khi = float_sum_binade(((double)(before + segtot[seg]) + c) * ff + 1.0);
            if (!(carry == carry) || carry < 0.0f) { klo = 1; khi = 0; }      // NaN / negative carry-in: no map, sequential-order path
            s_klo = klo;
            s_khi = khi;
        }
        __syncthreads();
    }
    const int klo = s_klo, khi = s_khi;
    for (int i = t; i < FIC_SUM_KMAPS * FIC_SUM_SUBS; i += FIC_SUM_THREADS) {
        const int k = i / FIC_SUM_SUBS;
        if (k < klo || k > khi) {
            sub[2 * i] = FIC_SUM_SAT;
            sub[2 * i + 1] = FIC_SUM_SAT;
            if ((i % FIC_SUM_SUBS) == 0) {
                maps[((size_t)seg * FIC_SUM_KMAPS + k) * 2 + 0] = FIC_SUM_SAT;
                maps[((size_t)seg * FIC_SUM_KMAPS + k) * 2 + 1] = FIC_SUM_SAT;
            }
        }
    }
    if (klo > khi) return;
    const int i0 = seg * FIC_SUM_SEG + t * FIC_SUM_RUN;
    uint4 v4[FIC_SUM_RUN / 4];
#pragma unroll
    for (int j = 0; j < FIC_SUM_RUN / 4; j++) {
        const int i = i0 + 4 * j;
        uint4 v = (i < count) ? *(const uint4*)(d + i) : make_uint4(0, 0, 0, 0);
        if (i + 1 >= count) v.y = 0;
        if (i + 2 >= count) v.z = 0;
        if (i + 3 >= count) v.w = 0;
        v4[j] = v;
    }
    for (int k = klo; k <= khi; k++) {
        uint32_t a0, a1;
        run_maps(v4, k, a0, a1);
        // ordered composition: lanes of a wave by shuffles (lane t absorbs lane t + stride), then the 16 waves
#pragma unroll
        for (int stride = 1; stride < 64; stride <<= 1) {
            const uint32_t r0 = __shfl_down(a0, stride, 64), r1 = __shfl_down(a1, stride, 64);
            if ((lane & (2 * stride - 1)) == 0) map_compose(a0, a1, r0, r1);
        }
        if (lane == 0) {
            w0[wave] = a0;
            w1[wave] = a1;
            sub[(k * FIC_SUM_SUBS + wave) * 2 + 0] = a0;       // the map of this wave's 4096 values: the walk's second level
            sub[(k * FIC_SUM_SUBS + wave) * 2 + 1] = a1;
        }
        __syncthreads();
        if (t == 0) {
            uint32_t l0 = w0[0], l1 = w1[0];
            for (int w = 1; w < FIC_SUM_THREADS / 64; w++) map_compose(l0, l1, w0[w], w1[w]);
            maps[((size_t)seg * FIC_SUM_KMAPS + k) * 2 + 0] = l0;
            maps[((size_t)seg * FIC_SUM_KMAPS + k) * 2 + 1] = l1;
        }
        __syncthreads();
    }
}
// The walk over the segments; every thread of the workgroup returns the sum.  The maps come through LDS, 256 segments at a
// time (read one by one from global memory by the walking thread they cost a dependent load -- a microsecond -- per segment).
// A segment in which the sum leaves its binade (or still has a fractional part) is walked again at its second level, the 16
// sub-segments of 4096 values, and only the sub-segment where it happens takes java_float_sum.
#define FIC_SUM_WALK_SEGS 256
__device__ float float_sum_walk(float carry, const uint32_t* __restrict__ d, int count, const uint32_t* __restrict__ maps,
                                const uint32_t* __restrict__ submaps, int* fallbacks)
{
    __shared__ uint32_t w_maps[FIC_SUM_WALK_SEGS * FIC_SUM_KMAPS * 2];
    __shared__ uint32_t w_sub[FIC_SUM_KMAPS * FIC_SUM_SUBS * 2];
    __shared__ float w_sum;
    __shared__ int w_seg, w_part;
    const int t = threadIdx.x, nseg = float_sum_segments(count);
    // one step: the map `mp` (increments for an even / odd m) applied to the integer-valued sum; false: not applicable here
    auto apply = [](float& sum, const uint32_t* mp, int stride_k) {
        if (!(sum == sum) || sum != truncf(sum) || sum < 0.0f) return false;
        int k = 0;
        if (sum >= 16777216.0f) k = (int)((__float_as_uint(sum) >> 23) & 0xFF) - 127 - 23;
        if (k >= FIC_SUM_KMAPS) return true;                                       // too large to move: unchanged
        const uint32_t m = (uint32_t)(sum * __uint_as_float((uint32_t)(127 - k) << 23));
        const uint32_t inc = mp[(size_t)k * stride_k + (m & 1u)];
        if (inc >= FIC_SUM_SAT || m + inc >= FIC_SUM_SAT) return false;            // leaves the binade inside
        sum = (float)(m + inc) * __uint_as_float((uint32_t)(127 + k) << 23);       // exact: m' < 2^24
        return true;
    };
    if (t == 0) { w_sum = carry; w_seg = 0; }
    __syncthreads();
    for (int c0 = 0; c0 < nseg; c0 += FIC_SUM_WALK_SEGS) {              // uniform
        const int c1 = c0 + FIC_SUM_WALK_SEGS < nseg ? c0 + FIC_SUM_WALK_SEGS : nseg;
        for (int i = t; i < (c1 - c0) * FIC_SUM_KMAPS * 2; i += FIC_SUM_THREADS) w_maps[i] = maps[(size_t)c0 * FIC_SUM_KMAPS * 2 + i];
        __syncthreads();
        for (;;) {
            if (t == 0) {
                float sum = w_sum;
                int seg = w_seg;
                while (seg < c1) {
                    if (!(sum == sum)) { seg = nseg; break; }                     // NaN stays NaN
                    if (!apply(sum, w_maps + (size_t)(seg - c0) * FIC_SUM_KMAPS * 2, 2)) break;
                    seg++;
                }
                w_sum = sum;
                w_seg = seg;
                w_part = 0;
            }
            __syncthreads();
            const int seg = w_seg;
            if (seg >= c1) break;                                       // this chunk of segments is done (or everything is)
            // second level: the 16 sub-segments of segment `seg`
            for (int i = t; i < FIC_SUM_KMAPS * FIC_SUM_SUBS * 2; i += FIC_SUM_THREADS)
                w_sub[i] = submaps[(size_t)seg * FIC_SUM_KMAPS * FIC_SUM_SUBS * 2 + i];
            __syncthreads();
            for (;;) {
                if (t == 0) {
                    float sum = w_sum;
                    int part = w_part;
                    while (part < FIC_SUM_SUBS && apply(sum, w_sub + part * 2, FIC_SUM_SUBS * 2)) part++;
                    w_sum = sum;
                    w_part = part;
                }
                __syncthreads();
                const int part = w_part;
                if (part >= FIC_SUM_SUBS) break;
                const long long off = (long long)seg * FIC_SUM_SEG + (long long)part * FIC_SUM_SUB;
                const int len = count - off < FIC_SUM_SUB ? (int)(count - off) : FIC_SUM_SUB;
                const float s2 = len > 0 ? java_float_sum(w_sum, d + off, len) : w_sum;
                __syncthreads();
                if (t == 0) {
                    w_sum = s2;
                    w_part = part + 1;
                    if (fallbacks && len > 0) *fallbacks += 1;
                }
                __syncthreads();
            }
            if (t == 0) w_seg = seg + 1;
            __syncthreads();
        }
        if (w_seg >= nseg) break;
        __syncthreads();                                                // everyone has read w_seg before the maps are replaced
    }
    return w_sum;
}

// Loop control of FC:413-417.  avgError += ... (FC:407) is a float accumulation: fl(carry + SSD) when that is provably what
// Java gets (every partial sum an exact float), else the scan above over sqbuf: k_decode_maps on every CU, then k_decode_step.
__device__ __forceinline__ bool decode_sum_is_exact(float carry, unsigned long long ssd)
{
    return carry == truncf(carry) && carry >= 0.0f && (double)carry + (double)ssd < 16777216.0;
}
__global__ __launch_bounds__(256) void k_decode_segtot(const FicDecodeState* __restrict__ state, const uint32_t* __restrict__ sqbuf,
                                                       uint32_t* __restrict__ scratch, int counter, int wh)
{
    const int plane = blockIdx.y;
    const FicDecodeState* st = state + plane;
    if (st->done || decode_sum_is_exact(st->avg, st->ssd[counter])) return;      // uniform per workgroup
    float_sum_segment_total(sqbuf + (size_t)plane * wh, wh, blockIdx.x, float_sum_totals(scratch + plane * float_sum_scratch_words(wh), wh));
}
__global__ __launch_bounds__(FIC_SUM_THREADS) void k_decode_maps(const FicDecodeState* __restrict__ state, const uint32_t* __restrict__ sqbuf,
                                                                uint32_t* __restrict__ scratch, int counter, int wh, float f)
{
    const int plane = blockIdx.y, seg = blockIdx.x;
    const FicDecodeState* st = state + plane;
    if (st->done) return;                                      // uniform per workgroup
    const float carry = st->avg;
    if (decode_sum_is_exact(carry, st->ssd[counter])) return;
    uint32_t* sc = scratch + plane * float_sum_scratch_words(wh);
    float_sum_segment_maps(sqbuf + (size_t)plane * wh, wh, seg, carry, f, float_sum_totals(sc, wh), sc, float_sum_submaps(sc, wh));
}
__global__ __launch_bounds__(FIC_SUM_THREADS) void k_decode_step(FicDecodeState* __restrict__ state,
                                                                const uint32_t* __restrict__ sqbuf, const uint32_t* __restrict__ maps,
                                                                int counter, int wh, int planes)
{
    const int plane = blockIdx.x;
    FicDecodeState* st = state + plane;
    if (st->done) return;                                      // uniform per workgroup
    const float carry = st->avg;                               // 0 after iteration 0; the static's old value before it (FC:20)
    const unsigned long long ssd = st->ssd[counter];
    const bool exact = decode_sum_is_exact(carry, ssd);
    float avg;
    if (exact) avg = __fadd_rn(carry, (float)ssd);             // every partial sum is an exact float: order is irrelevant
    else {
        const uint32_t* sc = maps + plane * float_sum_scratch_words(wh);
        avg = float_sum_walk(carry, sqbuf + (size_t)plane * wh, wh, sc, float_sum_submaps(const_cast<uint32_t*>(sc), wh), nullptr);
    }
    if (threadIdx.x != 0) return;
    if (!exact) st->seq_sums += 1;
    avg = __fdiv_rn(avg, (float)wh);                           // FC:413
    st->iters = counter + 1;
    st->avg_out = avg;
    if (avg < 1.0f) st->done = 1;                              // FC:414
    else st->avg = (counter != 49) ? 0.0f : avg;               // FC:416
    if (counter == 49) st->done = 1;
}

// test hook: out[0] = Java's sum through the same code as the decoder (segment totals, segment maps on every CU, then the
// walk); out[1] = segments that went through java_float_sum
__global__ __launch_bounds__(256) void k_float_sum_probe_tot(const uint32_t* __restrict__ vals, int count, uint32_t* __restrict__ scratch)
{
    float_sum_segment_total(vals, count, blockIdx.x, float_sum_totals(scratch, count));
}
__global__ __launch_bounds__(FIC_SUM_THREADS) void k_float_sum_probe_maps(float carry, const uint32_t* __restrict__ vals, int count,
                                                                         uint32_t* __restrict__ scratch, float f)
{
    float_sum_segment_maps(vals, count, blockIdx.x, carry, f, float_sum_totals(scratch, count), scratch, float_sum_submaps(scratch, count));
}
__global__ __launch_bounds__(FIC_SUM_THREADS) void k_float_sum_probe(float carry, const uint32_t* __restrict__ vals, int count,
                                                                    const uint32_t* __restrict__ scratch, float* __restrict__ out)
{
    __shared__ int fb;
    if (threadIdx.x == 0) fb = 0;
    __syncthreads();
    const float r = float_sum_walk(carry, vals, count, scratch, float_sum_submaps(const_cast<uint32_t*>(scratch), count), &fb);
    if (threadIdx.x == 0) { out[0] = r; out[1] = (float)fb; }
}
// (1 + 2^-24)^count, rounded up: how far the float sum of `count` addends can drift from the exact total, as a factor
static float float_sum_drift(size_t count) { return (float)(exp((double)count * 5.9604645e-8 * 1.0001) * 1.0001); }
size_t fic_float_sum_map_words(size_t count) { return float_sum_scratch_words((int)count); }
// u32 words of the decoder's scratch for `planes` images of wh pixels: the squares in Java's visiting order + the scan's scratch
size_t fic_decode_sq_words(size_t planes, size_t wh) { return planes * (wh + float_sum_scratch_words((int)wh)); }
int fic_launch_float_sum_probe(float carry, const uint32_t* vals, int count, uint32_t* scratch, float* out2, hipStream_t s)
{
    if (count > 0) {
        const int nseg = float_sum_segments(count);
        hipLaunchKernelGGL(k_float_sum_probe_tot, dim3(nseg), dim3(256), 0, s, vals, count, scratch);
        FIC_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_float_sum_probe_maps, dim3(nseg), dim3(FIC_SUM_THREADS), 0, s, carry, vals, count, scratch, float_sum_drift((size_t)count));
        FIC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_float_sum_probe, dim3(1), dim3(FIC_SUM_THREADS), 0, s, carry, vals, count, (const uint32_t*)scratch, out2);
    FIC_LAUNCH_CHECK();
    return 0;
}

// host-side launchers
// loop control of one decoder iteration (shared by the grey and the RGB decoder); sqbuf: fic_decode_sq_words(planes, wh) words
int fic_launch_decode_step(FicDecodeState* state, uint32_t* sqbuf, int counter, int wh, int planes, hipStream_t s)
{
    uint32_t* scratch = sqbuf + (size_t)planes * wh;
    const int nseg = float_sum_segments(wh);
    hipLaunchKernelGGL(k_decode_segtot, dim3(nseg, planes), dim3(256), 0, s, (const FicDecodeState*)state, (const uint32_t*)sqbuf, scratch,
                       counter, wh);
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_decode_maps, dim3(nseg, planes), dim3(FIC_SUM_THREADS), 0, s, (const FicDecodeState*)state,
                       (const uint32_t*)sqbuf, scratch, counter, wh, float_sum_drift((size_t)wh));
    FIC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_decode_step, dim3(planes), dim3(FIC_SUM_THREADS), 0, s, state, (const uint32_t*)sqbuf, (const uint32_t*)scratch,
                       counter, wh, planes);
    FIC_LAUNCH_CHECK();
    return 0;
}

// one decoder iteration: scale (FC:382 -> createCodebuch -> scaleImage), paint, loop control
int fic_launch_decode_iteration(uint8_t* scaled, uint8_t* image, const int32_t* qrows, const int32_t* iso,
                                FicDecodeState* state, uint32_t* sqbuf, int counter, const FicGeom& g, hipStream_t s)
{
    if (fic_launch_scale(image, scaled, g, s)) return -1;      // FC:382 -> createCodebuch -> scaleImage
    auto paint = g.B == 4 ? k_decode_paint<4> : (g.B == 8 ? k_decode_paint<8> : k_decode_paint<16>);
    hipLaunchKernelGGL(paint, dim3((g.Nr * g.B + 255) / 256, g.planes), dim3(256), 0, s, scaled, image, qrows, iso,
                       state, sqbuf, counter, g);
    FIC_LAUNCH_CHECK();
    return fic_launch_decode_step(state, sqbuf, counter, g.W * g.H, g.planes, s);
}
