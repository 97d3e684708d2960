// microbench_valu.hip -- per-SIMD issue rate of the fp32 VALU forms the neighbour loops use.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_valu.hip -o tools/microbench_valu
// Reports cycles per wave64 instruction per SIMD at saturation, using the in-kernel shader clock
// (s_memtime) so DVFS does not distort the number.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

#define ITERS 16384
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float s, unsigned long long *clk,
                                                           unsigned long long *rclk)
{
    float a[8];
    float2v p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2v{a[i], a[i] + 1.f}; }
    float2v s2 = {s, s * 0.5f};
    int li = threadIdx.x & 7;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
            if (MODE == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(s2));
            if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
            if (MODE == 4) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
            if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
            if (MODE == 6) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(s) : "vcc");
            if (MODE == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 9) asm volatile("v_fma_f32 %0, %0, -1.0, %1" : "+v"(a[i]) : "s"(s));
            if (MODE == 10) asm volatile("v_fma_f32 %0, %0, %1, 1.0 clamp" : "+v"(a[i]) : "v"(s));
            if (MODE == 11) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
            if (MODE == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
            if (MODE == 13) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 14) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 15) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(a[i]) : "s20");
            if (MODE == 16) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(li));
            if (MODE == 17) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (MODE == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 19) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 20) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(a[i]));
            if (MODE == 21) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(s));
            if (MODE == 22) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" :: "v"(a[i]), "v"(s) : "s20", "s21");
            if (MODE == 23) asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(a[i]) : "v"(s));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[blockIdx.x] = t1 - t0; rclk[blockIdx.x] = r1 - r0; }
}

template <int MODE> void run(const char *name, int wavesPerSimd, float *d, unsigned long long *clk)
{
    int blocks = 256 * wavesPerSimd; // 256 threads = 4 waves = one per SIMD
    unsigned long long *rclk = clk + 256 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 1.0001f, clk, rclk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    k<MODE><<<blocks, 256>>>(d, 1.0001f, clk, rclk);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks), hr(blocks);
    (void)hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hr.data(), rclk, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0, ravg = 0;
    for (auto v : h) avg += (double)v;
    for (auto v : hr) ravg += (double)v;
    avg /= blocks; ravg /= blocks;
    // (a) each wave ran ITERS*8 instructions in avg shader cycles while sharing its SIMD with
    //     wavesPerSimd waves (if the blocks are spread evenly, w per CU);
    // (b) the same from the launch's wall time (HIP events): all 1024 SIMDs of the chip issued
    //     blocks*4*ITERS*8 wave-instructions in ms at the in-kernel clock s_memtime/s_memrealtime
    const double ghz = avg / ravg * 0.1;
    const double wallCyc = ms * 1e-3 * ghz * 1e9 * 1024.0 / ((double)blocks * 4 * ITERS * 8.0);
    printf("%-22s w/SIMD=%d  in-wave %.2f  wall %.2f cycles per wave-instruction per SIMD  (clock %.2f GHz, %.3f ms)\n",
           name, wavesPerSimd, avg / (ITERS * 8.0) / wavesPerSimd, wallCyc, ghz, ms);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    unsigned long long *clk; (void)hipMalloc(&clk, 2 * 256 * 8 * 8);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, d, clk);
        run<1>("v_mul_f32", w, d, clk);
        run<8>("v_add_f32", w, d, clk);
        run<4>("v_sub_f32 sgpr", w, d, clk);
        run<9>("v_fma(x,-1,sgpr)", w, d, clk);
        run<10>("v_fma clamp", w, d, clk);
        run<11>("v_mul_f32 sgpr", w, d, clk);
        run<21>("v_fmac_f32", w, d, clk);
        run<2>("v_pk_fma_f32", w, d, clk);
        run<3>("v_pk_mul_f32", w, d, clk);
        run<5>("v_pk_add_f32", w, d, clk);
        run<6>("v_max_f32", w, d, clk);
        run<23>("v_med3_f32", w, d, clk);
        run<7>("v_cmp_lt_f32 vcc", w, d, clk);
        run<22>("v_cmp_lt_f32 sgpr", w, d, clk);
        run<12>("v_cndmask_b32", w, d, clk);
        run<18>("v_and_b32", w, d, clk);
        run<19>("v_add_u32", w, d, clk);
        run<20>("v_mbcnt_lo", w, d, clk);
        run<13>("v_sqrt_f32", w, d, clk);
        run<14>("v_rcp_f32", w, d, clk);
        run<15>("v_readlane_b32", w, d, clk);
        run<16>("ds_bpermute+wait", w, d, clk);
        run<17>("v_mov_dpp row_shr", w, d, clk);
    }
    return 0;
}
