// mfma_valu_mix.hip -- what does ONE v_mfma_f32_32x32x16_f16 per tile cost beside NV VALU instructions of epilogue?
// (B = 4: K = n = 16, i.e. one MFMA per 32x32 tile of (domain, range) pairs; k_sweep_q<1, 0> has NV = 8 v_max3 + 1 v_cmp.)
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_valu_mix.hip -o gpurun_out/mfma_valu_mix
// A wave keeps 8 column tiles of range fragments (32 VGPRs) and streams domain tiles (one 16-byte load per lane and tile, L2-resident);
// per (domain tile, column tile): the MFMA of the NEXT unit is issued, then NV v_max3 over the current unit's 16 outputs (NV = 8 is
// the full reduction; fewer = a partial one, to see the slope) and, with CMP, the compare + ballot + branch of the kernel.
// Prints cycles per tile per SIMD at the in-kernel clock for NV in {0, 2, 4, 6, 8} x waves per SIMD in {1, 2, 3, 4}: the matrix
// pipe alone needs 32; if VALU and MFMA share one issue port at 4 cycles per instruction the line is max(32, 4 (NV + CMP + 1)).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <int NV, int CMP, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k(const v4i* __restrict__ pool, const v4i* __restrict__ rng, int ndt, float tau,
                                                 unsigned long long* __restrict__ clk, float* __restrict__ out)
{
    constexpr int CTW = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i rb[CTW];
#pragma unroll
    for (int i = 0; i < CTW; i++) rb[i] = rng[(((blockIdx.x * 4 + wave) % 2048) * CTW + i) * 64 + lane];
    const v4i* pa = pool + lane;
    const unsigned long long c0 = clock64(), t0 = wall_clock64();
    const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int hits = 0;
    float keep = 0.0f;
    v4i a0 = pa[0], a1 = pa[64];
    auto tile = [&](const v4i& at, int ci) __attribute__((always_inline)) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, at), __builtin_bit_cast(f16x8, rb[ci]), zero, 0, 0, 0);
    };
    auto epi = [&](const v16f& a) __attribute__((always_inline)) {
        float m = 0.0f;
        if constexpr (NV >= 8) {
            const float m0 = max3f(fabsf(a[0]), fabsf(a[1]), fabsf(a[2])), m1 = max3f(fabsf(a[3]), fabsf(a[4]), fabsf(a[5]));
            const float m2 = max3f(fabsf(a[6]), fabsf(a[7]), fabsf(a[8])), m3 = max3f(fabsf(a[9]), fabsf(a[10]), fabsf(a[11]));
            const float m4 = max3f(fabsf(a[12]), fabsf(a[13]), fabsf(a[14]));
            m = max3f(max3f(m0, m1, m2), max3f(m3, m4, fabsf(a[15])), 0.0f);
        } else if constexpr (NV == 6) {
            const float m0 = max3f(fabsf(a[0]), fabsf(a[1]), fabsf(a[2])), m1 = max3f(fabsf(a[3]), fabsf(a[4]), fabsf(a[5]));
            const float m2 = max3f(fabsf(a[6]), fabsf(a[7]), fabsf(a[8])), m3 = max3f(fabsf(a[9]), fabsf(a[10]), fabsf(a[11]));
            m = max3f(max3f(m0, m1, m2), m3, 0.0f);
        } else if constexpr (NV == 4) {
            const float m0 = max3f(fabsf(a[0]), fabsf(a[1]), fabsf(a[2])), m1 = max3f(fabsf(a[3]), fabsf(a[4]), fabsf(a[5]));
            const float m2 = max3f(fabsf(a[6]), fabsf(a[7]), fabsf(a[8]));
            m = max3f(m0, m1, m2);
        } else if constexpr (NV == 2) {
            m = max3f(max3f(fabsf(a[0]), fabsf(a[1]), fabsf(a[2])), fabsf(a[3]), fabsf(a[4]));
        } else {
            m = a[0];
        }
        return m;
    };
    v16f acc = tile(a0, 0);
    auto step = [&](int dt, v4i& ac, const v4i& an) __attribute__((always_inline)) {
#pragma unroll
        for (int ci = 0; ci < CTW; ci++) {
            v16f nacc = ci + 1 < CTW ? tile(ac, ci + 1) : tile(an, 0);
            if (ci == CTW - 2) ac = pa[((dt + 2) % ndt) * 64];
            const float m = epi(acc);
            if constexpr (CMP) {
                if (__builtin_amdgcn_ballot_w64(m > tau) != 0) hits++;
            } else {
                keep = m;                                 // NV = 0 without compare: the MFMA stream alone
            }
            acc = nacc;
        }
    };
    for (int dt = 0; dt < ndt; dt += 2) {
        step(dt, a0, a1);
        step(dt + 1, a1, a0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + keep + (float)hits;
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
    const int ndt = 2048, reps = argc > 1 ? atoi(argv[1]) : 20;
    const size_t np = (size_t)ndt * 64, nr = (size_t)2048 * 8 * 64;
    _Float16* hp = (_Float16*)malloc(np * 16);
    _Float16* hr = (_Float16*)malloc(nr * 16);
    uint64_t z = 99;
    for (size_t i = 0; i < np * 8; i++) hp[i] = (_Float16)(((double)(sm(z) >> 40) / 8388608.0 - 1.0) * 0.25);
    for (size_t i = 0; i < nr * 8; i++) hr[i] = (_Float16)(float)((int)(sm(z) >> 55) - 255);
    v4i *dp, *dr;
    unsigned long long* clk;
    float* out;
    hipMalloc(&dp, np * 16);
    hipMalloc(&dr, nr * 16);
    hipMalloc(&clk, 16);
    hipMalloc(&out, (size_t)1024 * 256 * 4);
    hipMemcpy(dp, hp, np * 16, hipMemcpyHostToDevice);
    hipMemcpy(dr, hr, nr * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](int nv, int cmp, auto kern, int waves) {
        const int grid = 256 * waves;
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dp, dr, ndt, 1e30f, clk, out);
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
        const double tiles = (double)reps * grid * 4 * ndt * 8;
        printf("NV %d%s, %d waves/SIMD: %8.3f ms  clock %.3f GHz  %6.1f cycles per tile per SIMD (pipe 32; one port at 4 cycles: %d)  frac of 2.5 PF %.3f\n",
               nv, cmp ? "+cmp" : "     ", waves, ms, ghz, ms * 1e-3 * ghz * 1e9 / (tiles / 1024.0), 4 * (nv + cmp + 1),
               tiles * 32.0 * 32.0 * 16.0 * 2.0 / (ms * 1e-3) / 2.5e15);
    };
#define ROW(NV, CMP)                           \
    run(NV, CMP, k<NV, CMP, 1>, 1);            \
    run(NV, CMP, k<NV, CMP, 2>, 2);            \
    run(NV, CMP, k<NV, CMP, 3>, 3);            \
    run(NV, CMP, k<NV, CMP, 4>, 4);
    for (int pass = 0; pass < 2; pass++) {
        ROW(0, 0)
        ROW(2, 1)
        ROW(4, 1)
        ROW(6, 1)
        ROW(8, 1)
    }
    return 0;
}
