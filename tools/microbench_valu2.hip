// microbench_valu2.hip -- issue classes of the gfx950 vector ALU with DISTINCT register operands,
// dependent-chain latency, and SGPR / literal operand costs.  Follow-up of microbench_valu.hip.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_valu2.hip -o tools/microbench_valu2
// Every figure is chip-wide wall time (HIP events) x in-kernel clock (s_memtime / s_memrealtime)
// x 1024 SIMDs / wave-instructions issued: cycles per wave64 instruction per SIMD at saturation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 8192
#define NCH 8

// OP is an asm template over %0 = dest/accumulator a[i], %1 = b[i], %2 = c[i], %3 = sgpr s
#define DEFK(NAME, OP, CHAINS)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, float s, unsigned long long *clk)    \
    {                                                                                            \
        float a[NCH], b[NCH], c[NCH];                                                            \
        for (int i = 0; i < NCH; i++) {                                                          \
            a[i] = threadIdx.x * 0.001f + i;                                                     \
            b[i] = a[i] * 0.5f + 1.f;                                                            \
            c[i] = a[i] * 0.25f + 2.f;                                                           \
        }                                                                                        \
        asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "vcc");                            \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
        for (int it = 0; it < ITERS; it++) {                                                     \
            _Pragma("unroll") for (int r = 0; r < NCH / CHAINS; r++)                             \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++)                                   \
                asm volatile(OP : "+v"(a[i]) : "v"(b[i]), "v"(c[i]), "s"(s));                    \
        }                                                                                        \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                \
        float r = 0;                                                                             \
        for (int i = 0; i < NCH; i++) r += a[i] + b[i] + c[i];                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                          \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; } \
    }

DEFK(k_fma, "v_fma_f32 %0, %1, %2, %0", 8)
DEFK(k_fma_d3, "v_fma_f32 %0, %1, %2, %1", 8)
DEFK(k_fmac, "v_fmac_f32 %0, %1, %2", 8)
DEFK(k_mul, "v_mul_f32 %0, %1, %2", 8)
DEFK(k_add, "v_add_f32 %0, %1, %2", 8)
DEFK(k_sub, "v_sub_f32 %0, %1, %2", 8)
DEFK(k_subrev, "v_subrev_f32 %0, %1, %2", 8)
DEFK(k_max, "v_max_f32 %0, %1, %2", 8)
DEFK(k_min, "v_min_f32 %0, %1, %2", 8)
DEFK(k_med3, "v_med3_f32 %0, %0, %1, %2", 8)
DEFK(k_fma_clamp, "v_fma_f32 %0, %1, %2, %0 clamp", 8)
DEFK(k_mul_clamp, "v_mul_f32_e64 %0, %1, %2 clamp", 8)
DEFK(k_fma_neg, "v_fma_f32 %0, -%1, %2, %0", 8)
DEFK(k_fma_abs, "v_fma_f32 %0, |%1|, %2, %0", 8)
DEFK(k_fma_lit, "v_fma_f32 %0, %1, %2, 1.0", 8)
DEFK(k_fma_sgpr, "v_fma_f32 %0, %1, %3, %0", 8)
DEFK(k_mul_sgpr, "v_mul_f32 %0, %3, %1", 8)
DEFK(k_mul_lit, "v_mul_f32 %0, 2.0, %1", 8)
DEFK(k_mul_lit32, "v_mul_f32 %0, 0x3fc00001, %1", 8)
DEFK(k_rsq, "v_rsq_f32 %0, %1", 8)
DEFK(k_rcp, "v_rcp_f32 %0, %1", 8)
DEFK(k_sqrt, "v_sqrt_f32 %0, %1", 8)
DEFK(k_mov, "v_mov_b32 %0, %1", 8)
DEFK(k_cndmask, "v_cndmask_b32 %0, %1, %2, vcc", 8)
DEFK(k_cmp, "v_cmp_lt_f32 vcc, %1, %2", 8)
DEFK(k_cmpx, "v_cmp_lt_f32 s[20:21], %1, %2", 8)
DEFK(k_addu, "v_add_u32 %0, %1, %2", 8)
DEFK(k_subu, "v_sub_u32 %0, %1, %2", 8)
DEFK(k_and, "v_and_b32 %0, %1, %2", 8)
DEFK(k_or, "v_or_b32 %0, %1, %2", 8)
DEFK(k_xor, "v_xor_b32 %0, %1, %2", 8)
DEFK(k_lshl, "v_lshlrev_b32 %0, 4, %1", 8)
DEFK(k_lshl_add, "v_lshl_add_u32 %0, %1, 4, %2", 8)
DEFK(k_add_lshl, "v_add_lshl_u32 %0, %1, %2, 4", 8)
DEFK(k_add3, "v_add3_u32 %0, %0, %1, %2", 8)
DEFK(k_mad24, "v_mad_u32_u24 %0, %1, %2, %0", 8)
DEFK(k_mullo, "v_mul_lo_u32 %0, %1, %2", 8)
DEFK(k_cvt, "v_cvt_f32_i32 %0, %1", 8)
DEFK(k_bfe, "v_bfe_u32 %0, %1, 4, 8", 8)
DEFK(k_max3, "v_max3_f32 %0, %0, %1, %2", 8)
DEFK(k_dpp, "v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf", 8)
DEFK(k_add_dpp, "v_add_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf", 8)
DEFK(k_readlane, "v_readlane_b32 s20, %1, 3", 8)
// dependent chains: the same instruction on 1, 2, 4 accumulators
DEFK(k_fma_dep1, "v_fma_f32 %0, %0, %1, %2", 1)
DEFK(k_fma_dep2, "v_fma_f32 %0, %0, %1, %2", 2)
DEFK(k_fma_dep4, "v_fma_f32 %0, %0, %1, %2", 4)
DEFK(k_mul_dep1, "v_mul_f32 %0, %0, %1", 1)
DEFK(k_max_dep1, "v_max_f32 %0, %0, %1", 1)
DEFK(k_rsq_dep1, "v_rsq_f32 %0, %0", 1)

typedef void (*kern_t)(float *, float, unsigned long long *);

static void run(kern_t k, const char *name, int w, float *d, unsigned long long *clk)
{
    const int blocks = 256 * w;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<<<blocks, 256>>>(d, 1.0001f, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    k<<<blocks, 256>>>(d, 1.0001f, clk);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), clk, 2 * blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0, ravg = 0;
    for (int i = 0; i < blocks; i++) { avg += (double)h[2 * i]; ravg += (double)h[2 * i + 1]; }
    const double ghz = avg / ravg * 0.1;
    const double cyc = ms * 1e-3 * ghz * 1e9 * 1024.0 / ((double)blocks * 4 * ITERS * NCH);
    printf("%-14s w/SIMD=%d  %6.2f cycles/wave-instr/SIMD   (%.2f GHz)\n", name, w, cyc, ghz);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    unsigned long long *clk;
    (void)hipMalloc(&clk, 2 * 256 * 8 * 8);
#define R(K) run(K, #K, w, d, clk)
    for (int w : {1, 4, 8}) {
        R(k_fma); R(k_fma_d3); R(k_fmac); R(k_mul); R(k_add); R(k_sub); R(k_subrev); R(k_max); R(k_min);
        R(k_med3); R(k_max3); R(k_fma_clamp); R(k_mul_clamp); R(k_fma_neg); R(k_fma_abs); R(k_fma_lit);
        R(k_fma_sgpr); R(k_mul_sgpr); R(k_mul_lit); R(k_mul_lit32); R(k_rsq); R(k_rcp); R(k_sqrt);
        R(k_mov); R(k_cndmask); R(k_cmp); R(k_cmpx); R(k_addu); R(k_subu); R(k_and); R(k_or); R(k_xor);
        R(k_lshl); R(k_lshl_add); R(k_add_lshl); R(k_add3); R(k_mad24); R(k_mullo); R(k_cvt); R(k_bfe);
        R(k_dpp); R(k_add_dpp); R(k_readlane);
        R(k_fma_dep1); R(k_fma_dep2); R(k_fma_dep4); R(k_mul_dep1); R(k_max_dep1); R(k_rsq_dep1);
        printf("\n");
    }
    return 0;
}
