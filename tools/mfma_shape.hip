// mfma_shape.hip -- VERDICT r2 #2(b): does v_mfma_f32_16x16x32_f16 beat v_mfma_f32_32x32x16_f16 for the 1-isometry prune GEMM
// of k_sweep_q at B = 8 (K = 64), at the same output tile per wave, on random data, with the kernel's own epilogue?
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_shape.hip -o gpurun_out/mfma_shape
// Both variants: a wave keeps 128 range columns (64 VGPRs of f16 fragments) and streams domain tiles of 32 rows x K = 64
// (4 KB per wave and tile, from an L2-resident 2 MB pool of random f16 in [-1, 1]); per domain tile
//   shape 0: 4 column tiles x (4 chained 32x32x16 MFMAs -> 16 outputs per lane; 8 v_max3 + compare)
//   shape 1: 8 column tiles x (2 row tiles x 2 chained 16x16x32 MFMAs -> 8 outputs per lane; 4 v_max3 + compare)
// i.e. the same 16 MFMA-pipe cycles per output, 9 VALU against 10 per 128 matrix cycles.  The next unit's MFMAs are issued
// before the current unit's epilogue, as in the kernel.  theta is far above every output: no slow path, the fast path only.
// Prints ms, cycles per 32x32 tile-equivalent per SIMD at the measured in-kernel clock, and the clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// CTW: 32-column tiles per wave (registers: 16 CTW VGPRs of range fragments); WAVES: waves per SIMD the register budget is cut for
template <int SHAPE, int CTW, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k(const v4i* __restrict__ pool, const v4i* __restrict__ rng, int ndt, float tau,
                                             unsigned long long* __restrict__ clk, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i rb[4 * CTW];
#pragma unroll
    for (int i = 0; i < 4 * CTW; i++) rb[i] = rng[(((blockIdx.x * 4 + wave) % 2048) * 16 + i) * 64 + lane];
    const v4i* pa = pool + lane;
    const unsigned long long c0 = clock64(), t0 = wall_clock64();
    int hits = 0;
    v4i a0[4], a1[4];
#pragma unroll
    for (int m = 0; m < 4; m++) a0[m] = pa[(0 * 4 + m) * 64];
#pragma unroll
    for (int m = 0; m < 4; m++) a1[m] = pa[(1 * 4 + m) * 64];
    if constexpr (SHAPE == 0) {
        const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto tile = [&](const v4i (&at)[4], int ci) __attribute__((always_inline)) {
            v16f acc = zero;
#pragma unroll
            for (int m = 0; m < 4; m++)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, at[m]), __builtin_bit_cast(f16x8, rb[ci * 4 + m]), acc, 0, 0, 0);
            return acc;
        };
        auto epi = [&](const v16f& a) __attribute__((always_inline)) {
            const float m0 = max3f(fabsf(a[0]), fabsf(a[1]), fabsf(a[2])), m1 = max3f(fabsf(a[3]), fabsf(a[4]), fabsf(a[5]));
            const float m2 = max3f(fabsf(a[6]), fabsf(a[7]), fabsf(a[8])), m3 = max3f(fabsf(a[9]), fabsf(a[10]), fabsf(a[11]));
            const float m4 = max3f(fabsf(a[12]), fabsf(a[13]), fabsf(a[14]));
            return max3f(max3f(m0, m1, m2), max3f(m3, m4, fabsf(a[15])), 0.0f);
        };
        v16f acc = tile(a0, 0);
        auto step = [&](int dt, v4i (&ac)[4], const v4i (&an)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int ci = 0; ci < CTW; ci++) {
                v16f nacc = ci + 1 < CTW ? tile(ac, ci + 1) : tile(an, 0);
                if (ci == CTW - 2) {
#pragma unroll
                    for (int m = 0; m < 4; m++) ac[m] = pa[(((dt + 2) % ndt) * 4 + m) * 64];
                }
                if (__builtin_amdgcn_ballot_w64(epi(acc) > tau) != 0) hits++;
                acc = nacc;
            }
        };
        for (int dt = 0; dt < ndt; dt += 2) {
            step(dt, a0, a1);
            step(dt + 1, a1, a0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = acc[0] + (float)hits;
    } else {
        // fragments reinterpreted for 16x16x32: at[2 * rt + ks], rb[2 * cj + ks]; values are random either way
        const v4f zero = {0, 0, 0, 0};
        auto unit = [&](const v4i (&at)[4], int cj, v4f& x0, v4f& x1) __attribute__((always_inline)) {
            x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, at[0]), __builtin_bit_cast(f16x8, rb[2 * cj]), zero, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, at[1]), __builtin_bit_cast(f16x8, rb[2 * cj + 1]), x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, at[2]), __builtin_bit_cast(f16x8, rb[2 * cj]), zero, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, at[3]), __builtin_bit_cast(f16x8, rb[2 * cj + 1]), x1, 0, 0, 0);
        };
        auto epi = [&](const v4f& x0, const v4f& x1) __attribute__((always_inline)) {
            const float m0 = max3f(fabsf(x0[0]), fabsf(x0[1]), fabsf(x0[2])), m1 = max3f(fabsf(x0[3]), fabsf(x1[0]), fabsf(x1[1]));
            return max3f(m0, m1, max3f(fabsf(x1[2]), fabsf(x1[3]), 0.0f));
        };
        v4f x0, x1;
        unit(a0, 0, x0, x1);
        auto step = [&](int dt, v4i (&ac)[4], const v4i (&an)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int cj = 0; cj < 2 * CTW; cj++) {
                v4f n0, n1;
                if (cj + 1 < 2 * CTW) unit(ac, cj + 1, n0, n1);
                else unit(an, 0, n0, n1);
                if (cj == 2 * CTW - 2) {
#pragma unroll
                    for (int m = 0; m < 4; m++) ac[m] = pa[(((dt + 2) % ndt) * 4 + m) * 64];
                }
                if (__builtin_amdgcn_ballot_w64(epi(x0, x1) > tau) != 0) hits++;
                x0 = n0;
                x1 = n1;
            }
        };
        for (int dt = 0; dt < ndt; dt += 2) {
            step(dt, a0, a1);
            step(dt + 1, a1, a0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = x0[0] + (float)hits;
    }
    if (lane == 0 && wave == 0 && (blockIdx.x & 15) == 0) {
        atomicAdd(&clk[0], (unsigned long long)(clock64() - c0));
        atomicAdd(&clk[1], (unsigned long long)(wall_clock64() - t0));
    }
}

static uint64_t sm(uint64_t& z)
{
    z += 0x9E3779B97F4A7C15ull;
    uint64_t x = z;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

int main(int argc, char** argv)
{
    const int ndt = 512, nwg = 2048, reps = argc > 1 ? atoi(argv[1]) : 40;     // 512 workgroups = 2 per CU: 2 waves per SIMD
    const size_t np = (size_t)ndt * 4 * 64, nr = (size_t)nwg * 4 * 16 * 64;
    _Float16* hp = (_Float16*)malloc(np * 16);
    _Float16* hr = (_Float16*)malloc(nr * 16);
    uint64_t z = 1234;
    for (size_t i = 0; i < np * 8; i++) hp[i] = (_Float16)(((double)(sm(z) >> 40) / 8388608.0 - 1.0) * 0.25);           // normalised domain pixels
    for (size_t i = 0; i < nr * 8; i++) hr[i] = (_Float16)(float)((int)(sm(z) >> 55) - 255);                            // centred range pixels
    v4i *dp, *dr;
    unsigned long long* clk;
    float* out;
    hipMalloc(&dp, np * 16);
    hipMalloc(&dr, nr * 16);
    hipMalloc(&clk, 16);
    hipMalloc(&out, (size_t)nwg * 256 * 4);
    hipMemcpy(dp, hp, np * 16, hipMemcpyHostToDevice);
    hipMemcpy(dr, hr, nr * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int ctw, int waves) {
        const int grid = 256 * waves;                        // `waves` workgroups per CU = `waves` waves per SIMD
        hipMemset(clk, 0, 16);
        hipDeviceSynchronize();
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dp, dr, ndt, 1e30f, clk, out);
        hipDeviceSynchronize();
        hipMemset(clk, 0, 16);
        hipEventRecord(e0, 0);
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dp, dr, ndt, 1e30f, clk, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2];
        hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
        const double ghz = c[1] ? (double)c[0] / (double)c[1] / 10.0 : 0.0;
        const double tiles = (double)reps * grid * 4 * ndt * ctw;                 // 32x32 tile-equivalents
        const double flop = tiles * 32.0 * 32.0 * 64.0 * 2.0;
        printf("%-10s CTW %d, %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  frac %.3f of 2500  clock %.3f GHz  %.1f cycles per 32x32xK64 tile per SIMD (floor 128)\n",
               name, ctw, waves, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e15, ghz, ms * 1e-3 * ghz * 1e9 / (tiles / 1024.0));
    };
    for (int pass = 0; pass < 2; pass++) {
        run("32x32x16", k<0, 4, 2>, 4, 2);
        run("16x16x32", k<1, 4, 2>, 4, 2);
        run("16x16x32", k<1, 3, 3>, 3, 3);
        run("16x16x32", k<1, 2, 3>, 2, 3);
        run("16x16x32", k<1, 2, 4>, 2, 4);
        run("32x32x16", k<0, 3, 3>, 3, 3);
        run("32x32x16", k<0, 2, 4>, 2, 4);
    }
    return 0;
}
