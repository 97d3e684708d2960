// microbench_pairs.hip -- candidate-pair inner loops of the neighbour kernels in isolation:
// AoS records + scalar fp32 (what sph_neighbors.h does) against SoA records + packed fp32
// (v_pk_*_f32, two candidates per lane and instruction).  LDS-resident data, no global traffic.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/microbench_pairs.hip -o tools/microbench_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define NREC 1024
#define ITERS 64

// density, AoS scalar: 8 targets x 8 slices per wave, each lane walks NREC/8 records
__global__ __launch_bounds__(256) void k_aos(const float4 *__restrict__ src, float *out)
{
    __shared__ float4 rec[NREC];
    for (int i = threadIdx.x; i < NREC; i += 256) rec[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, slice = lane & 7;
    const float4 t = rec[lane >> 3];
    const float nih2 = -1.0f;
    float acc = 0.f;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll 4
        for (int k = slice; k < NREC; k += 8) {
            const float4 c = rec[k];
            asm volatile("" ::"v"(c.w));
            const float dx = t.x - c.x, dy = t.y - c.y, dz = t.z - c.z;
            const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            const float w = __builtin_amdgcn_fmed3f(__builtin_fmaf(d2, nih2, 1.0f), 0.f, 1.0f);
            acc = __builtin_fmaf(w * w, w, acc);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// density, SoA packed: each lane handles records (2k, 2k+1)
__global__ __launch_bounds__(256) void k_soa_pk(const float4 *__restrict__ src, float *out)
{
    __shared__ float X[NREC], Y[NREC], Z[NREC];
    for (int i = threadIdx.x; i < NREC; i += 256) {
        const float4 v = src[i];
        X[i] = v.x; Y[i] = v.y; Z[i] = v.z;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, slice = lane & 7;
    const int ti = lane >> 3;
    const f2 tx = {X[ti], X[ti]}, ty = {Y[ti], Y[ti]}, tz = {Z[ti], Z[ti]};
    const f2 nih2 = {-1.0f, -1.0f}, one = {1.0f, 1.0f}, zero = {0.f, 0.f};
    f2 acc = {0.f, 0.f};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll 4
        for (int k = slice * 2; k < NREC; k += 16) {
            const f2 cx = *(const f2 *)&X[k], cy = *(const f2 *)&Y[k], cz = *(const f2 *)&Z[k];
            const f2 dx = tx - cx, dy = ty - cy, dz = tz - cz;
            const f2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            f2 w = __builtin_elementwise_fma(d2, nih2, one);
            w = __builtin_elementwise_min(__builtin_elementwise_max(w, zero), one);
            acc = __builtin_elementwise_fma(w * w, w, acc);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
}

// density, AoS scalar with the norm expansion: d2 = |c|^2 + |t|^2 - 2 t.c  (c.w = |c|^2)
__global__ __launch_bounds__(256) void k_aos_norm(const float4 *__restrict__ src, float *out)
{
    __shared__ float4 rec[NREC];
    for (int i = threadIdx.x; i < NREC; i += 256) {
        float4 v = src[i];
        v.w = v.x * v.x + v.y * v.y + v.z * v.z;
        rec[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, slice = lane & 7;
    const float4 t = rec[lane >> 3];
    const float m2x = -2.f * t.x, m2y = -2.f * t.y, m2z = -2.f * t.z;
    const float nih2 = -1.0f, base = 1.0f - t.w; // 1 - |t|^2/h^2
    float acc = 0.f;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll 4
        for (int k = slice; k < NREC; k += 8) {
            const float4 c = rec[k];
            const float e = __builtin_fmaf(m2z, c.z, __builtin_fmaf(m2y, c.y, __builtin_fmaf(m2x, c.x, c.w)));
            const float w = __builtin_amdgcn_fmed3f(__builtin_fmaf(e, nih2, base), 0.f, 1.0f);
            acc = __builtin_fmaf(w * w, w, acc);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K> double run(K kern, const char *name, const float4 *src, float *out, int blocks)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    kern<<<blocks, 256>>>(src, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    kern<<<blocks, 256>>>(src, out);
    (void)hipEventRecord(b);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    // pairs: per wave 8 targets x NREC candidates x ITERS
    const double pairs = (double)blocks * 4 * 8 * NREC * ITERS;
    printf("%-12s %8.3f ms  %7.1f Gpairs/s\n", name, ms, pairs / ms / 1e6);
    return ms;
}

int main()
{
    std::vector<float4> h(NREC);
    for (int i = 0; i < NREC; i++)
        h[i] = make_float4((i * 37 % 101) * 0.02f, (i * 53 % 103) * 0.02f, (i * 71 % 107) * 0.02f, 1.f);
    float4 *src;
    float *out;
    const int blocks = 256 * 16;
    (void)hipMalloc(&src, NREC * sizeof(float4));
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    (void)hipMemcpy(src, h.data(), NREC * sizeof(float4), hipMemcpyHostToDevice);
    run(k_aos, "aos scalar", src, out, blocks);
    run(k_soa_pk, "soa packed", src, out, blocks);
    run(k_aos_norm, "aos norm", src, out, blocks);
    std::vector<float> r(8);
    (void)hipMemcpy(r.data(), out, 32, hipMemcpyDeviceToHost);
    printf("check %g %g\n", r[0], r[1]);
    return 0;
}
