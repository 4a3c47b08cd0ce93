// fic_devfn.h -- device helpers shared by the kernel translation units (Java cast semantics, orderable keys,
// the exact Java error epilogue, isometries, the reference's window/index logic, range-store addressing).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fic_device.h"

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

// The same map in affine form: iso_source(k, B, x, y) = sx + sy * B with
//   sx = ax*x + bx*y + cx,  sy = ay*x + by*y + cy   (m = B - 1),
// so a copy can be gathered with one multiply-add per pixel instead of a switch (and straight from an image with row
// pitch W: offset = (cy*W + cx) + x*(ay*W + ax) + y*(by*W + bx)).
__device__ __forceinline__ void iso_affine(int k, int m, int& ax, int& bx, int& cx, int& ay, int& by, int& cy)
{
    ax = bx = cx = ay = by = cy = 0;
    switch (k) {
    default:
    case 0: ax = 1;          by = 1;          break;
    case 1: bx = 1;          ay = -1; cy = m; break;
    case 2: ax = -1; cx = m; by = -1; cy = m; break;
    case 3: bx = -1; cx = m; ay = 1;          break;
    case 4: ax = -1; cx = m; by = 1;          break;
    case 5: ax = 1;          by = -1; cy = m; break;
    case 6: bx = 1;          ay = 1;          break;
    case 7: bx = -1; cx = m; ay = -1; cy = m; break;
    }
}

// Exact sums of products of an isometry copy of a range block with a domain block, the range pixels gathered straight
// from the image (no stored copies):
//   s  = sum_pos copy_k[pos] * d[pos],     copy_k[pos] = r[iso_source(iso_inverse(k), pos)]      
//   s2 = sum_pos copy_k[n-1-pos] * d[pos]  -- the copy of k's point-reflected partner isometry (0/2, 1/3, 4/5, 6/7) -- if PAIR
// blk = top-left pixel of the range block in an image of row pitch W; dom = the domain block's n bytes as dwords.
template <int B, bool PAIR>
__device__ __forceinline__ void iso_dot(const uint8_t* __restrict__ blk, int W, int k, const uint32_t* __restrict__ dom,
                                        uint32_t& s, uint32_t& s2)
{
    constexpr int n = B * B, CH = n < 64 ? n : 64, ROWS = CH / B;
    int ax, bx, cx, ay, by, cy;
    iso_affine(iso_inverse(k), B - 1, ax, bx, cx, ay, by, cy);
    const int ox = ay * W + ax, oy = by * W + bx, o0 = cy * W + cx;
    s = 0;
    s2 = 0;
    for (int c = 0; c < n / CH; c++) {                     // 64 positions at a time: one pass at B = 4 / 8, four at B = 16
        const int obase = o0 + c * ROWS * oy;
        uint32_t rw[CH / 4];
#pragma unroll
        for (int q = 0; q < CH / 4; q++) {
            uint32_t w = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int pl = 4 * q + t;
                w |= (uint32_t)blk[obase + (pl % B) * ox + (pl / B) * oy] << (8 * t);
            }
            rw[q] = w;
        }
#pragma unroll
        for (int q = 0; q < CH / 4; q++) {
            const int qg = c * (CH / 4) + q;
            s = __builtin_amdgcn_udot4(rw[q], dom[qg], s, false);
            // partner: sum_pos copy[n-1-pos] d[pos] = sum_q copy[q] d[n-1-q] -> the mirrored dword, bytes reversed
            if constexpr (PAIR) s2 = __builtin_amdgcn_udot4(rw[q], __builtin_bswap32(dom[n / 4 - 1 - qg]), s2, false);
        }
    }
}

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

// XCD-aware decode of a 1-D grid of ncombos x ntg workgroups.  The dispatcher deals workgroups round-robin over
// the 8 XCDs by linear id (MI355X_MICROARCH.md "Workgroup dispatch"), so inside every window of 8*ntg consecutive
// ids, id % 8 selects one of 8 "combos" (a pool chunk of a plane: the data the workgroups share through L2) and
// id / 8 the range-tile group x: all ntg workgroups of a combo land on ONE XCD and are dispatched together, so
// they stream the chunk in lock-step out of that XCD's L2.  Placement only affects speed, never results.
__device__ __forceinline__ void xcd_decode(unsigned p, int ncombos, int ntg, int& combo, int& x)
{
    const unsigned full = (unsigned)(ncombos / 8) * 8u * (unsigned)ntg;
    if (p < full) {
        const unsigned r = p >> 3;
        combo = (int)((r / (unsigned)ntg) * 8u + (p & 7u));
        x = (int)(r % (unsigned)ntg);
    } else {                                   // the last ncombos % 8 combos: plain order
        const unsigned q = p - full;
        combo = (ncombos / 8) * 8 + (int)(q / (unsigned)ntg);
        x = (int)(q % (unsigned)ntg);
    }
}

// host-side: surface a failed kernel launch to the C ABI
#define FIC_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e_ = hipGetLastError();         \
        if (e_ != hipSuccess) return (int)e_;      \
    } while (0)
