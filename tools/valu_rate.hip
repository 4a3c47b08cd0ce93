// valu_rate.hip -- measures wave64 VALU issue rates on gfx950 (cycles per instruction per SIMD)
// for the opcodes the sweep kernel is made of.  Evidence for the VALU roofline in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void k(uint32_t* out, unsigned long long* clk, int iters, uint32_t sval)
{
    uint32_t a[8], b = threadIdx.x * 2654435761u + 12345u;
    float fa[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa[8], pb = {1.0001f, 0.9999f};
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x + i; fa[i] = 1.0f + i; pa[i] = f2{1.0f + i, 2.0f + i}; }
    uint32_t s = __builtin_amdgcn_readfirstlane(sval);
    unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (OP == 0) {
#define X(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(b));
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "s"(s));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(fa[i]) : "v"(fa[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pa[i]) : "v"(pb));
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(i) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "s"(s));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(fa[i]) : "v"(a[i]));
                REP8(X)
#undef X
            } else if (OP == 7) {
#define X(i) asm volatile("v_cmp_nle_f32 vcc, |%0|, %1" :: "v"(fa[i]), "v"(fa[(i + 1) & 7]) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 8) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(fa[i]) : "s"(s));
                REP8(X)
#undef X
            } else if (OP == 9) {
#define X(i) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(b));
                REP8(X)
#undef X
            } else if (OP == 10) {
#define X(i) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(b));
                REP8(X)
#undef X
            } else if (OP == 11) {
#define X(i) asm volatile("v_max3_i32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(a[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if (OP == 12) {
#define X(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            }
        }
    }
    unsigned long long t1 = clock64(), w1 = wall_clock64();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i] + (uint32_t)fa[i] + (uint32_t)pa[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int OP>
void run(const char* name, int waves_per_simd)
{
    const int CUS = 256, iters = 20000;
    int blocks = CUS * waves_per_simd;        // 256-thread blocks: 4 waves = 1 per SIMD
    uint32_t* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, (size_t)blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk, 100, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < blocks; i++) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
    cyc /= blocks; wall /= blocks;
    double ghz = cyc / (wall / 100e6) / 1e9;                    // wall_clock64 ticks at 100 MHz
    double instr_per_wave = (double)iters * 32;
    double cyc_per_instr_simd = cyc / (instr_per_wave * waves_per_simd);   // SIMD cycles per wave-instruction
    double chip_lane_ops = (double)blocks * 4 * instr_per_wave * 64 / (ms * 1e-3);
    printf("%-28s waves/SIMD=%d  %.3f ms  clk=%.2f GHz  cycles/instr/SIMD=%.2f  chip lane-ops/s=%.3e\n", name,
           waves_per_simd, ms, ghz, cyc_per_instr_simd, chip_lane_ops);
    hipFree(out); hipFree(clk);
}

int main()
{
    for (int w : {2, 4, 8}) {
        run<0>("v_dot4_u32_u8 v,v", w);
        run<1>("v_dot4_u32_u8 v,s", w);
        run<2>("v_fma_f32", w);
        run<3>("v_pk_fma_f32", w);
        run<4>("v_mad_i32_i24 v,s", w);
        run<5>("v_add_u32", w);
        run<6>("v_cvt_f32_i32", w);
        run<7>("v_cmp_nle_f32 |v|,v", w);
        run<8>("v_mul_f32 s,v", w);
        run<9>("v_dot2c_i32_i16 s,v", w);
        run<10>("v_dot2c_i32_i16 v,v", w);
        run<11>("v_max3_i32", w);
        run<12>("v_sub_u32", w);
    }
    return 0;
}
