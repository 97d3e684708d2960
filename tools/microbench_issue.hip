// microbench_issue.hip -- how the gfx950 SIMD pairs adjacent vector instructions of one wave.
// microbench_valu2 shows two issue classes (about 2.4 and 4.2 cycles per wave64 instruction per
// SIMD at saturation) and that a single dependent chain of "fast" instructions runs at the slow
// rate.  This file times fixed instruction SEQUENCES (one asm block per loop body) to find the
// rule: which neighbours in a wave's stream have to be independent, what a slow-class or
// transcendental instruction between them costs, and what the density / force pair bodies cost
// serial and interleaved.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_issue.hip -o tools/microbench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4096

// BODY: asm text over a0..a7 = %0..%7 (in/out), b0..b7 = %8..%15 (in); NI = instructions in BODY
#define DEFSEQ(NAME, NI, BODY)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, float s, unsigned long long *clk)    \
    {                                                                                            \
        float a[8], b[8];                                                                        \
        for (int i = 0; i < 8; i++) {                                                            \
            a[i] = threadIdx.x * 0.001f + i;                                                     \
            b[i] = a[i] * 0.5f + 1.f;                                                            \
        }                                                                                        \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
        for (int it = 0; it < ITERS; it++) {                                                     \
            asm volatile(BODY                                                                    \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]),           \
                           "+v"(a[5]), "+v"(a[6]), "+v"(a[7])                                    \
                         : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]),     \
                           "v"(b[6]), "v"(b[7])                                                  \
                         : "s20", "s21", "s22", "s23", "vcc", "scc");                            \
        }                                                                                        \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                \
        float r = 0;                                                                             \
        for (int i = 0; i < 8; i++) r += a[i] + b[i];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                          \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; } \
    }                                                                                            \
    static const int NAME##_ni = NI;

#define F(d, x, y, z) "v_fma_f32 " d ", " x ", " y ", " z "\n"
#define MX(d, x, y) "v_max_f32 " d ", " x ", " y "\n"
#define RSQ(d, x) "v_rsq_f32 " d ", " x "\n"

// two chains, alternating: every instruction independent of its predecessor
DEFSEQ(s_alt2, 8, F("%0","%0","%8","%9") F("%1","%1","%8","%9") F("%0","%0","%8","%9") F("%1","%1","%8","%9")
                  F("%0","%0","%8","%9") F("%1","%1","%8","%9") F("%0","%0","%8","%9") F("%1","%1","%8","%9"))
// two chains, in dependent pairs: a0 a0 a1 a1
DEFSEQ(s_pair2, 8, F("%0","%0","%8","%9") F("%0","%0","%8","%9") F("%1","%1","%8","%9") F("%1","%1","%8","%9")
                   F("%0","%0","%8","%9") F("%0","%0","%8","%9") F("%1","%1","%8","%9") F("%1","%1","%8","%9"))
// three chains a0 a0 a0 a1 a1 a1 (one independent boundary in three)
DEFSEQ(s_trip2, 6, F("%0","%0","%8","%9") F("%0","%0","%8","%9") F("%0","%0","%8","%9")
                   F("%1","%1","%8","%9") F("%1","%1","%8","%9") F("%1","%1","%8","%9"))
// fast / slow alternating, all independent
DEFSEQ(s_fast_slow, 8, F("%0","%0","%8","%9") MX("%4","%4","%10") F("%1","%1","%8","%9") MX("%5","%5","%10")
                       F("%2","%2","%8","%9") MX("%6","%6","%10") F("%3","%3","%8","%9") MX("%7","%7","%10"))
// two fast then one slow, all independent
DEFSEQ(s_ffs, 6, F("%0","%0","%8","%9") F("%1","%1","%8","%9") MX("%4","%4","%10")
                 F("%2","%2","%8","%9") F("%3","%3","%8","%9") MX("%5","%5","%10"))
// transcendental among independent fast instructions: 1 rsq + 7 fma, 1 rsq + 3 fma
DEFSEQ(s_rsq7, 8, RSQ("%4","%10") F("%0","%0","%8","%9") F("%1","%1","%8","%9") F("%2","%2","%8","%9")
                  F("%3","%3","%8","%9") F("%0","%0","%8","%9") F("%1","%1","%8","%9") F("%2","%2","%8","%9"))
DEFSEQ(s_rsq3, 8, RSQ("%4","%10") F("%0","%0","%8","%9") F("%1","%1","%8","%9") F("%2","%2","%8","%9")
                  RSQ("%5","%11") F("%3","%3","%8","%9") F("%0","%0","%8","%9") F("%1","%1","%8","%9"))
// same register read by both neighbours / written by one and not read by the other
DEFSEQ(s_alt4, 8, F("%0","%0","%8","%9") F("%1","%1","%10","%11") F("%2","%2","%12","%13") F("%3","%3","%14","%15")
                  F("%0","%0","%8","%9") F("%1","%1","%10","%11") F("%2","%2","%12","%13") F("%3","%3","%14","%15"))

// ---- the density pair body (9 instructions): t = (b0,b1,b2), -1/h2 = b3, candidate = (b4,b5,b6)
// serial, as the compiler emits it today; acc = a0; temporaries a4..a6
#define DPAIR(acc, t0, t1, t2, cx, cy, cz)                                                       \
    "v_sub_f32 " t0 ", %8, " cx "\n"                                                             \
    "v_sub_f32 " t1 ", %9, " cy "\n"                                                             \
    "v_mul_f32 " t0 ", " t0 ", " t0 "\n"                                                         \
    "v_sub_f32 " t2 ", %10, " cz "\n"                                                            \
    "v_fma_f32 " t0 ", " t1 ", " t1 ", " t0 "\n"                                                 \
    "v_fma_f32 " t0 ", " t2 ", " t2 ", " t0 "\n"                                                 \
    "v_fma_f32 " t0 ", " t0 ", %11, 1.0 clamp\n"                                                 \
    "v_mul_f32 " t1 ", " t0 ", " t0 "\n"                                                         \
    "v_fma_f32 " acc ", " t1 ", " t0 ", " acc "\n"
DEFSEQ(s_dens_serial, 18, DPAIR("%0", "%4", "%5", "%6", "%12", "%13", "%14")
                          DPAIR("%0", "%4", "%5", "%6", "%13", "%14", "%15"))
// two pairs interleaved instruction by instruction: accumulators a0, a1; temporaries a2..a7
#define DPAIR2(accA, accB, A0, A1, A2, B0, B1, B2, ax, ay, az, bx, by, bz)                       \
    "v_sub_f32 " A0 ", %8, " ax "\n"     "v_sub_f32 " B0 ", %8, " bx "\n"                        \
    "v_sub_f32 " A1 ", %9, " ay "\n"     "v_sub_f32 " B1 ", %9, " by "\n"                        \
    "v_sub_f32 " A2 ", %10, " az "\n"    "v_sub_f32 " B2 ", %10, " bz "\n"                       \
    "v_mul_f32 " A0 ", " A0 ", " A0 "\n" "v_mul_f32 " B0 ", " B0 ", " B0 "\n"                    \
    "v_fma_f32 " A0 ", " A1 ", " A1 ", " A0 "\n" "v_fma_f32 " B0 ", " B1 ", " B1 ", " B0 "\n"    \
    "v_fma_f32 " A0 ", " A2 ", " A2 ", " A0 "\n" "v_fma_f32 " B0 ", " B2 ", " B2 ", " B0 "\n"    \
    "v_fma_f32 " A0 ", " A0 ", %11, 1.0 clamp\n" "v_fma_f32 " B0 ", " B0 ", %11, 1.0 clamp\n"    \
    "v_mul_f32 " A1 ", " A0 ", " A0 "\n" "v_mul_f32 " B1 ", " B0 ", " B0 "\n"                    \
    "v_fma_f32 " accA ", " A1 ", " A0 ", " accA "\n" "v_fma_f32 " accB ", " B1 ", " B0 ", " accB "\n"
DEFSEQ(s_dens_x2, 18, DPAIR2("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7", "%12", "%13", "%14", "%13", "%14", "%15"))
// the same with max(0,.)^3 written with v_max (what a med3 / max formulation costs)
#define DPAIRM(acc, t0, t1, t2, cx, cy, cz)                                                      \
    "v_sub_f32 " t0 ", %8, " cx "\n"                                                             \
    "v_sub_f32 " t1 ", %9, " cy "\n"                                                             \
    "v_mul_f32 " t0 ", " t0 ", " t0 "\n"                                                         \
    "v_sub_f32 " t2 ", %10, " cz "\n"                                                            \
    "v_fma_f32 " t0 ", " t1 ", " t1 ", " t0 "\n"                                                 \
    "v_fma_f32 " t0 ", " t2 ", " t2 ", " t0 "\n"                                                 \
    "v_sub_f32 " t0 ", %11, " t0 "\n"                                                            \
    "v_max_f32 " t0 ", 0, " t0 "\n"                                                              \
    "v_mul_f32 " t1 ", " t0 ", " t0 "\n"                                                         \
    "v_fma_f32 " acc ", " t1 ", " t0 ", " acc "\n"
DEFSEQ(s_dens_max_serial, 20, DPAIRM("%0", "%4", "%5", "%6", "%12", "%13", "%14")
                              DPAIRM("%0", "%4", "%5", "%6", "%13", "%14", "%15"))

// ---- do scalar instructions take vector issue slots?  8 fma alone, with 8 / 16 s_add between
#define SA(r) "s_add_u32 " r ", " r ", 1\n"
DEFSEQ(s_fma8, 8, F("%0","%0","%8","%9") F("%1","%1","%8","%9") F("%2","%2","%8","%9") F("%3","%3","%8","%9")
                  F("%4","%4","%8","%9") F("%5","%5","%8","%9") F("%6","%6","%8","%9") F("%7","%7","%8","%9"))
DEFSEQ(s_fma8_salu8, 8, F("%0","%0","%8","%9") SA("s20") F("%1","%1","%8","%9") SA("s21") F("%2","%2","%8","%9") SA("s22")
                        F("%3","%3","%8","%9") SA("s23") F("%4","%4","%8","%9") SA("s20") F("%5","%5","%8","%9") SA("s21")
                        F("%6","%6","%8","%9") SA("s22") F("%7","%7","%8","%9") SA("s23"))
DEFSEQ(s_fma8_salu16, 8, F("%0","%0","%8","%9") SA("s20") SA("s21") F("%1","%1","%8","%9") SA("s22") SA("s23")
                         F("%2","%2","%8","%9") SA("s20") SA("s21") F("%3","%3","%8","%9") SA("s22") SA("s23")
                         F("%4","%4","%8","%9") SA("s20") SA("s21") F("%5","%5","%8","%9") SA("s22") SA("s23")
                         F("%6","%6","%8","%9") SA("s20") SA("s21") F("%7","%7","%8","%9") SA("s22") SA("s23"))
// 8 fma + 2 scalar compare-and-branch (never taken)
DEFSEQ(s_fma8_br2, 8, F("%0","%0","%8","%9") F("%1","%1","%8","%9") F("%2","%2","%8","%9") F("%3","%3","%8","%9")
                      "s_cmp_eq_u32 s20, 0x7fffffff\n s_cbranch_scc1 1f\n"
                      F("%4","%4","%8","%9") F("%5","%5","%8","%9") F("%6","%6","%8","%9") F("%7","%7","%8","%9")
                      "s_cmp_eq_u32 s21, 0x7fffffff\n s_cbranch_scc1 1f\n 1:\n")
// 8 fma + s_nop / s_waitcnt that never waits
DEFSEQ(s_fma8_wait4, 8, F("%0","%0","%8","%9") F("%1","%1","%8","%9") "s_waitcnt lgkmcnt(0)\n" F("%2","%2","%8","%9") F("%3","%3","%8","%9") "s_waitcnt lgkmcnt(0)\n"
                        F("%4","%4","%8","%9") F("%5","%5","%8","%9") "s_waitcnt lgkmcnt(0)\n" F("%6","%6","%8","%9") F("%7","%7","%8","%9") "s_waitcnt lgkmcnt(0)\n")
// v_cndmask with vcc / an SGPR pair, and right after the compare that feeds it
DEFSEQ(s_cnd_vcc, 8, "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                     "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n")
DEFSEQ(s_cnd_sgpr, 8, "v_cndmask_b32 %0, %8, %9, s[20:21]\n v_cndmask_b32 %1, %8, %9, s[20:21]\n v_cndmask_b32 %2, %8, %9, s[20:21]\n v_cndmask_b32 %3, %8, %9, s[20:21]\n"
                      "v_cndmask_b32 %4, %8, %9, s[20:21]\n v_cndmask_b32 %5, %8, %9, s[20:21]\n v_cndmask_b32 %6, %8, %9, s[20:21]\n v_cndmask_b32 %7, %8, %9, s[20:21]\n")
DEFSEQ(s_cmp_cnd, 8, "v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %9, vcc\n v_cmp_lt_f32 vcc, %8, %1\n v_cndmask_b32 %1, %8, %9, vcc\n"
                     "v_cmp_lt_f32 vcc, %8, %2\n v_cndmask_b32 %2, %8, %9, vcc\n v_cmp_lt_f32 vcc, %8, %3\n v_cndmask_b32 %3, %8, %9, vcc\n")

typedef void (*kern_t)(float *, float, unsigned long long *);

static void run(kern_t k, const char *name, int ni, int w, float *d, unsigned long long *clk)
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
    const double cyc = ms * 1e-3 * ghz * 1e9 * 1024.0 / ((double)blocks * 4 * ITERS * ni);
    printf("%-18s w/SIMD=%d  %6.2f cycles/wave-instr/SIMD  %7.2f cycles/body  %6.3f ns/body (%.2f GHz)\n",
           name, w, cyc, cyc * ni, cyc * ni / ghz, ghz);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    unsigned long long *clk;
    (void)hipMalloc(&clk, 2 * 256 * 8 * 8);
#define R(K) run(K, #K, K##_ni, w, d, clk)
    for (int w : {1, 2, 4, 8}) {
        R(s_alt2); R(s_pair2); R(s_trip2); R(s_alt4); R(s_fast_slow); R(s_ffs); R(s_rsq7); R(s_rsq3);
        R(s_dens_serial); R(s_dens_x2); R(s_dens_max_serial);
        R(s_fma8); R(s_fma8_salu8); R(s_fma8_salu16); R(s_fma8_br2); R(s_fma8_wait4);
        R(s_cnd_vcc); R(s_cnd_sgpr); R(s_cmp_cnd);
        printf("\n");
    }
    return 0;
}
