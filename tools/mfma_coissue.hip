// mfma_coissue.hip -- does VALU work hide beside v_mfma_f32_32x32x16_bf16 on gfx950, and when does it not?
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_coissue.hip -o gpurun_out/mfma_coissue
// Each variant runs ITER iterations of "one tile": 4 MFMAs (K = 64) and/or an epilogue of NV VALU ops, on a grid that
// puts WPS waves on every SIMD.  Prints cycles per tile per SIMD (2.4 GHz assumed from the event time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define ITER 4096

__device__ __forceinline__ v16f mf(v4i a, v4i b, v16f c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// MODE 0: MFMA only (4 dependent per tile).  1: epilogue only, on the accumulator registers (no MFMA in flight).
// 2: MFMAs of tile i+1 issued before the epilogue of tile i (two accumulator sets) -- the kernel's structure.
// 3: like 2, but the epilogue works on registers no MFMA ever touches.   4: like 2 with 4 INDEPENDENT accumulators
// per tile (16 MFMA outputs each K = 16) -- not the same math, only the issue pattern.
template <int MODE>
__global__ __launch_bounds__(256) void k(const v4i* in, float* out, float w, float tau)
{
    v4i a[4], b[4];
    for (int m = 0; m < 4; m++) { a[m] = in[threadIdx.x * 4 + m]; b[m] = in[1024 + threadIdx.x * 4 + m]; }
    const v16f zero = {0};
    v16f acc = zero, nacc = zero;
    float priv[16];
    for (int e = 0; e < 16; e++) priv[e] = (float)(threadIdx.x + e);
    int hits = 0;
    if (MODE >= 2) { for (int m = 0; m < 4; m++) acc = mf(a[m], b[m], acc); }
    for (int it = 0; it < ITER; it++) {
        if (MODE == 0) {                                   // one long dependent chain: nothing for the compiler to drop
            for (int m = 0; m < 4; m++) acc = mf(a[m], b[m], acc);
            continue;
        }
        if (MODE == 2 || MODE == 3) { nacc = zero; for (int m = 0; m < 4; m++) nacc = mf(a[m], b[m], nacc); }
        if (MODE == 4) { nacc = mf(a[0], b[0], zero); v16f x1 = mf(a[1], b[1], zero), x2 = mf(a[2], b[2], zero), x3 = mf(a[3], b[3], zero);
                         nacc[1] += x1[0] + x2[0] + x3[0]; }
        float p[16];
        for (int e = 0; e < 16; e++) p[e] = (MODE == 3 ? priv[e] : acc[e]) * w;
        bool any = false;
        for (int q = 0; q < 4; q++) {
            float mx = fmaxf(fmaxf(fabsf(p[4 * q]), fabsf(p[4 * q + 1])), fmaxf(fabsf(p[4 * q + 2]), fabsf(p[4 * q + 3])));
            any |= mx > tau;
        }
        if (__builtin_amdgcn_ballot_w64(any) != 0) hits++;
        if (MODE == 3) for (int e = 0; e < 16; e += 5) priv[e] += 1.0f;
        if (MODE == 1) acc[it & 15] += 1.0f; else acc = nacc;
        b[0][0] ^= it;
    }
    float s = 0;
    for (int e = 0; e < 16; e++) s += acc[e] + priv[e];
    out[blockIdx.x * 256 + threadIdx.x] = s + hits;
}

__device__ __forceinline__ bool epilogue(const v16f& acc, float w, float tau)
{
    float p[16];
    for (int e = 0; e < 16; e++) p[e] = acc[e] * w;
    bool any = false;
    for (int q = 0; q < 4; q++) {
        float mx = fmaxf(fmaxf(fabsf(p[4 * q]), fabsf(p[4 * q + 1])), fmaxf(fabsf(p[4 * q + 2]), fabsf(p[4 * q + 3])));
        any |= mx > tau;
    }
    return any;
}
// MODE 5: no software pipelining: chain -> epilogue on the same accumulator (occupancy does the overlap).
// MODE 6: phases of 4 tiles: 16 MFMAs into 4 accumulator sets, then the 4 epilogues.
// MODE 7: like 2, but the accumulator is first copied to plain VGPRs with 16 v_mov (the epilogue never reads MFMA results).
// MODE 8: three accumulator sets: the epilogue reads the set whose chain finished TWO chains ago.
// MODE 9: like 2, but the 16 multiplies of the epilogue are issued right after the chain that produced them (before
//         the next chain is issued), the max tree and compares beside the next chain.
template <int MODE>
__global__ __launch_bounds__(256) void k2(const v4i* in, float* out, float w, float tau)
{
    v4i a[4], b[4];
    for (int m = 0; m < 4; m++) { a[m] = in[threadIdx.x * 4 + m]; b[m] = in[1024 + threadIdx.x * 4 + m]; }
    const v16f zero = {0};
    int hits = 0;
    float s = 0;
    if (MODE == 5) {
        for (int it = 0; it < ITER; it++) {
            v16f acc = zero;
            for (int m = 0; m < 4; m++) acc = mf(a[m], b[m], acc);
            if (__builtin_amdgcn_ballot_w64(epilogue(acc, w, tau)) != 0) { hits++; s += acc[3]; }
            b[0][0] ^= it;
        }
    } else if (MODE == 6) {
        for (int it = 0; it < ITER; it += 4) {
            v16f acc[4];
            for (int t = 0; t < 4; t++) { acc[t] = zero; for (int m = 0; m < 4; m++) acc[t] = mf(a[m], b[(m + t) & 3], acc[t]); }
            for (int t = 0; t < 4; t++) if (__builtin_amdgcn_ballot_w64(epilogue(acc[t], w, tau)) != 0) { hits++; s += acc[t][3]; }
            b[0][0] ^= it;
        }
    } else if (MODE == 8) {
        v16f a0 = zero, a1 = zero;
        for (int m = 0; m < 4; m++) a0 = mf(a[m], b[m], a0);
        for (int m = 0; m < 4; m++) a1 = mf(a[m], b[(m + 1) & 3], a1);
        for (int it = 0; it < ITER; it++) {
            v16f a2 = zero;
            for (int m = 0; m < 4; m++) a2 = mf(a[m], b[m], a2);
            if (__builtin_amdgcn_ballot_w64(epilogue(a0, w, tau)) != 0) { hits++; s += a0[3]; }
            a0 = a1; a1 = a2;
            b[0][0] ^= it;
        }
        s += a0[0] + a1[0];
    } else if (MODE == 9) {
        v16f acc = zero;
        for (int m = 0; m < 4; m++) acc = mf(a[m], b[m], acc);
        float p[16];
        for (int e = 0; e < 16; e++) p[e] = acc[e] * w;
        for (int it = 0; it < ITER; it++) {
            v16f nacc = zero;
            for (int m = 0; m < 4; m++) nacc = mf(a[m], b[m], nacc);
            bool any = false;
            for (int q = 0; q < 4; q++) {
                float mx = fmaxf(fmaxf(fabsf(p[4 * q]), fabsf(p[4 * q + 1])), fmaxf(fabsf(p[4 * q + 2]), fabsf(p[4 * q + 3])));
                any |= mx > tau;
            }
            if (__builtin_amdgcn_ballot_w64(any) != 0) { hits++; s += p[3]; }
            for (int e = 0; e < 16; e++) p[e] = nacc[e] * w;     // waits for the chain, then 16 multiplies with no MFMA in flight
            b[0][0] ^= it;
        }
        s += p[0];
    } else {
        v16f acc = zero;
        for (int m = 0; m < 4; m++) acc = mf(a[m], b[m], acc);
        for (int it = 0; it < ITER; it++) {
            v16f nacc = zero;
            for (int m = 0; m < 4; m++) nacc = mf(a[m], b[m], nacc);
            v16f c;
            for (int e = 0; e < 16; e++) asm volatile("v_mov_b32 %0, %1" : "=v"(c[e]) : "v"(acc[e]));
            if (__builtin_amdgcn_ballot_w64(epilogue(c, w, tau)) != 0) { hits++; s += c[3]; }
            acc = nacc;
            b[0][0] ^= it;
        }
        s += acc[0];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s + hits;
}

template <int MODE>
static void run2(const char* name, int wps, const v4i* in, float* out)
{
    const int grid = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k2<MODE>, dim3(grid), dim3(256), 0, 0, in, out, 0.5f, 1e30f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k2<MODE>, dim3(grid), dim3(256), 0, 0, in, out, 0.5f, 1e30f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s waves/SIMD %d : %8.3f ms  -> %6.1f cycles per tile per SIMD\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)ITER * wps));
}

template <int MODE>
static void run(const char* name, int wps, const v4i* in, float* out)
{
    const int grid = 256 * wps;                               // one 4-wave workgroup per CU per wave-per-SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, 0.5f, 1e30f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, 0.5f, 1e30f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s waves/SIMD %d : %8.3f ms  -> %6.1f cycles per tile per SIMD\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)ITER * wps));
}

int main()
{
    v4i* in; float* out;
    hipMalloc(&in, 2048 * 16); hipMemset(in, 0, 2048 * 16); hipMalloc(&out, 256 * 8 * 256 * 4 * 2);
    for (int wps = 1; wps <= 4; wps++) {
        run2<5>("5 chain -> epilogue on the same accumulator, no pipelining", wps, in, out);
        run2<6>("6 phases of 4 tiles: 16 MFMAs, then 4 epilogues", wps, in, out);
        run2<7>("7 like 2, accumulator copied to plain VGPRs first", wps, in, out);
        run2<8>("8 three accumulator sets, epilogue on the oldest", wps, in, out);
        run2<9>("9 multiplies before the next chain, max tree beside it", wps, in, out);
    }
    for (int wps = 1; wps <= 2; wps++) {
        run<0>("0 MFMA only (4 dependent, K=64)", wps, in, out);
        run<1>("1 epilogue only (16 mul, max tree, 4 cmp)", wps, in, out);
        run<2>("2 next tile's MFMAs + epilogue on the other acc set", wps, in, out);
        run<3>("3 MFMAs + epilogue on registers MFMA never touches", wps, in, out);
        run<4>("4 four INDEPENDENT MFMAs + epilogue", wps, in, out);
    }
    return 0;
}
