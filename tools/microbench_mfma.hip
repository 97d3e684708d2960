// microbench_mfma.hip -- v_mfma_f32_4x4x1_16b_f32 on gfx950: operand layout and how its issue
// shares the SIMD with ordinary fp32 VALU work (the density walk's K = 4 chain per 256 pairs).
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_mfma.hip -o tools/microbench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// layout: D = A x B per block of four lanes
__global__ void k_layout(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int v = 0; v < 4; v++) d[l * 4 + v] = c[v];
}

#define ITERS 8192
// MODE 0: 4 chained MFMAs per iteration (two independent chains); MODE 1: + NV fma; MODE 2: the
// density step body: prep (3 sub, mul, 2 fma), 4 MFMAs, 4 x (mul clamp, mul, fma, alignbit)
template <int MODE, int NV> __global__ __launch_bounds__(256) void k(float *out, float s, unsigned long long *clk,
                                                                    unsigned long long *rclk)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    f4 c0 = {s, s, s, s}, c1 = c0;
    float bx = s, by = s * 0.5f, bz = s * 0.25f, bw = -s;
    float acc = 0.f;
    uint32_t hits = 0;
    float ox = s * 3.f, oy = s * 5.f, oz = s * 7.f;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        if (MODE <= 1) {
            f4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], bx, c0, 0, 0, 0);
            f4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[4], bx, c1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], by, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[5], by, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], bz, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[6], bz, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], bw, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[7], bw, d1, 0, 0, 0);
            // keep the results alive without VALU work
            asm volatile("" ::"v"(d0), "v"(d1));
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < NV; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i & 7]) : "v"(s));
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float x = a[4 * h], y = a[4 * h + 1], z = a[4 * h + 2];
                asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
                const float dx = x - ox, dy = y - oy, dz = z - oz;
                const float n = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(dx, bx, c0, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_4x4x1f32(dy, by, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_4x4x1f32(dz, bz, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_4x4x1f32(n, bw, d, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    float w;
                    asm("v_mul_f32 %0, 1.0, %1 clamp" : "=v"(w) : "v"(d[v]));
                    acc = __builtin_fmaf(w * w, w, acc);
                    hits = __builtin_amdgcn_alignbit(hits, __builtin_bit_cast(uint32_t, d[v]), 31);
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float r = acc + (float)hits + c0[0] + c1[1];
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[blockIdx.x] = t1 - t0; rclk[blockIdx.x] = r1 - r0; }
}

template <int MODE, int NV> void run(const char *name, int wavesPerSimd, float *d, unsigned long long *clk)
{
    int blocks = 256 * wavesPerSimd;
    unsigned long long *rclk = clk + 256 * 8;
    k<MODE, NV><<<blocks, 256>>>(d, 1.0001f, clk, rclk);
    (void)hipDeviceSynchronize();
    k<MODE, NV><<<blocks, 256>>>(d, 1.0001f, clk, rclk);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    // cycles of one SIMD per loop iteration of one wave (= 512 pairs: 8 MFMAs)
    printf("%-34s w/SIMD=%d  %.1f SIMD cycles per iteration (8 MFMAs, 512 pairs)\n", name, wavesPerSimd,
           avg / ITERS / wavesPerSimd);
}

int main()
{
    float *a, *b, *d;
    (void)hipMalloc(&a, 64 * 4); (void)hipMalloc(&b, 64 * 4); (void)hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    std::vector<float> ha(64), hb(64), hd(256);
    for (int i = 0; i < 64; i++) { ha[i] = 1.f + i; hb[i] = 100.f + i; }
    (void)hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(a, b, d);
    (void)hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int v = 0; v < 4; v++) {
            const int blk = l >> 2, j = l & 3;
            const float want = ha[4 * blk + v] * hb[4 * blk + j];
            if (hd[l * 4 + v] != want) bad++;
        }
    printf("layout D[vgpr v][lane 4b+j] = A[lane 4b+v] * B[lane 4b+j]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad)
        for (int l = 0; l < 8; l++) printf("  lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);

    unsigned long long *clk; (void)hipMalloc(&clk, 2 * 256 * 8 * 8);
    for (int w : {1, 2, 4, 5, 8}) {
        run<0, 0>("8 MFMA (2 chains of 4)", w, d, clk);
        run<1, 16>("8 MFMA + 16 v_fma", w, d, clk);
        run<1, 32>("8 MFMA + 32 v_fma", w, d, clk);
        run<1, 44>("8 MFMA + 44 v_fma", w, d, clk);
        run<2, 0>("density step body x2", w, d, clk);
    }
    return 0;
}
