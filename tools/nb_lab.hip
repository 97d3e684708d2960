// nb_lab.hip -- the candidate loops of the neighbour kernels in isolation, on one synthetic tile
// held in LDS, to price design variants in issue cycles per (cell, 64-lane) unit before they go
// into sph_neighbors.h.  One workgroup = one 4x4x2-cell tile with its 6x6x4 halo, about 8 particles
// per cell (jittered lattice, h = 1, spacing 0.5); lanes = 8 targets x 8 slices as in the product
// kernels.  Every variant walks the nine (dy,dz) row windows of a cell SEPARATELY through a slot
// table built on the host (slot = LDS record index that slice 0 reads); lanes that run past a
// window's end read the records that follow it (cell x+2: farther than h, weight exactly 0) or the
// FAR pad at the end of the row.
//   dens        density, uniform loop: every lane evaluates every slot
//   dens_mask   the same + a hit bit per slot kept in a VGPR, stored per lane
//   force       force, uniform loop
//   force_mask  force from the stored hit masks: every lane walks its own hits only
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/nb_lab.hip -o tools/nb_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HX 6
#define HY 6
#define HZ 4
#define NCELL (HX * HY * HZ)
#define MAXREC 1344
#define MAXSLOT 48
#define NUNIT 32
#define FAR 1.0e18f

struct Tile {
    int nrec;
    int unitSelf[NUNIT];          // LDS record index of the unit's first target
    int unitT[NUNIT];             // targets
    int unitSlots[NUNIT];         // slots
    unsigned short slot[NUNIT][MAXSLOT];
};

__device__ __forceinline__ float dppf(float v, int ctrl);
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <int MODE>
__global__ __launch_bounds__(256) void k_lab(const float4 *__restrict__ gP,
                                                const float4 *__restrict__ gV, const Tile *gT,
                                                float *__restrict__ out, unsigned *__restrict__ masks,
                                                int iters, unsigned long long *clk)
{
    __shared__ float4 recP[MAXREC];
#ifdef LAB_ALIAS // timing only: velocities read from the position image, so that more workgroups fit
    float4 *recV = recP;
#else
    __shared__ float4 recV[(MODE >= 2) ? MAXREC : 1];
#endif
    __shared__ Tile T;
    extern __shared__ int pad[]; // dynamic LDS: only there to limit the workgroups per CU
    if (iters < 0) pad[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < MAXREC; i += 256) {
        recP[i] = gP[i];
#ifndef LAB_ALIAS
        if (MODE >= 2) recV[i] = gV[i];
#endif
    }
    for (int i = threadIdx.x; i < (int)(sizeof(Tile) / 4); i += 256) ((int *)&T)[i] = ((const int *)gT)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u = lane >> 3, slice = lane & 7;
    const int slice16 = slice * 16;
    const char *baseP = (const char *)recP, *baseV = (const char *)recV;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    for (int it = 0; it < iters; it++) {
        for (int unit = w; unit < NUNIT; unit += 4) {
            const int nslots = __builtin_amdgcn_readfirstlane(T.unitSlots[unit]);
            const int self = T.unitSelf[unit];
            const int nT = T.unitT[unit];
            const bool active = u < nT;
            const float4 tp = recP[self + (active ? u : 0)];
            const unsigned short *st = T.slot[unit];
            const size_t mbase = ((size_t)(blockIdx.x * NUNIT + unit) * 64 + lane) * 2;
            if (MODE == 0 || MODE == 1) {
                const float nih2 = -1.0f + tp.w * 0.f; // -1/h2 in a VGPR
                float acc = 0.f;
                unsigned m0 = 0, m1 = 0;
                // hit bit of slot k: m = m + m + (w > 0), so slot k of a word sits at bit (n - 1 - k)
                auto pairD = [&](const float4 &c, unsigned &m) {
                    const float dx = tp.x - c.x, dy = tp.y - c.y, dz = tp.z - c.z;
                    const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    float wv;
                    asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(wv) : "v"(d2), "v"(nih2));
                    acc = __builtin_fmaf(wv * wv, wv, acc);
                    if (MODE == 1)
                        asm("v_cmp_lt_f32 vcc, 0, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(wv) : "vcc");
                };
                auto walk = [&](int k, int kend, unsigned &m) {
                    for (; k + 4 <= kend; k += 4) {
                        float4 c[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) c[q] = *(const float4 *)(baseP + (st[k + q] * 16 + slice16));
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            asm volatile("" ::"v"(c[q].w));
                            pairD(c[q], m);
                        }
                    }
                    for (; k < kend; k++) {
                        const float4 c = *(const float4 *)(baseP + (st[k] * 16 + slice16));
                        asm volatile("" ::"v"(c.w));
                        pairD(c, m);
                    }
                };
                const int n0 = nslots < 32 ? nslots : 32;
                walk(0, n0, m0);
                if (nslots > 32) walk(32, nslots, m1);
                acc += dpp_f<0x101>(acc);
                acc += dpp_f<0x102>(acc);
                acc += dpp_f<0x104>(acc);
                if (active && slice == 0) sink += acc;
                if (MODE == 1 && it == 0) {
                    masks[mbase] = m0;
                    masks[mbase + 1] = m1;
                }
            } else {
                const float4 tv = recV[self + (active ? u : 0)];
                const float negInvH = -1.0f + tp.w * 0.f, tiny = 1.0e-30f + tp.w * 0.f;
                float px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
                auto pair = [&](const float4 &cp, const float4 &cv) {
                    asm volatile("" ::"v"(cp.w), "v"(cv.w), "v"(cv.x)); // whole-record ds_read_b128
                    const float dx = tp.x - cp.x, dy = tp.y - cp.y, dz = tp.z - cp.z;
                    const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, tiny)));
                    const float rinv = __builtin_amdgcn_rsqf(d2);
                    const float r = d2 * rinv;
                    float tt;
                    asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(tt) : "v"(r), "v"(negInvH));
                    const float mt = cp.w * tt;
                    const float ap = mt * tt * (tv.w + cv.w) * rinv;
                    px = __builtin_fmaf(ap, dx, px);
                    py = __builtin_fmaf(ap, dy, py);
                    pz = __builtin_fmaf(ap, dz, pz);
                    vx = __builtin_fmaf(mt, cv.x - tv.x, vx);
                    vy = __builtin_fmaf(mt, cv.y - tv.y, vy);
                    vz = __builtin_fmaf(mt, cv.z - tv.z, vz);
                };
                if (MODE == 2) {
                    int k = 0;
                    for (; k + 2 <= nslots; k += 2) {
                        float4 cp[2], cv[2];
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            const int a = st[k + q] * 16 + slice16;
                            cp[q] = *(const float4 *)(baseP + a);
                            cv[q] = *(const float4 *)(baseV + a);
                        }
#pragma unroll
                        for (int q = 0; q < 2; q++) pair(cp[q], cv[q]);
                    }
                    for (; k < nslots; k++) {
                        const int a = st[k] * 16 + slice16;
                        pair(*(const float4 *)(baseP + a), *(const float4 *)(baseV + a));
                    }
                } else {
                    // every lane walks its own hit slots; the wave runs until the longest list ends
                    unsigned m0 = masks[mbase], m1 = masks[mbase + 1];
                    const int n0 = nslots < 32 ? nslots : 32;
                    const int farA = (MAXREC - 1) * 16;
                    while (__builtin_amdgcn_ballot_w64((m0 | m1) != 0u)) {
                        const bool lo = m0 != 0u;
                        const unsigned m = lo ? m0 : m1;
                        const int b = 31 - __builtin_clz(m | 1u); // highest set bit = earliest slot
                        const bool any = (m0 | m1) != 0u;
                        const int k = lo ? n0 - 1 - b : nslots - 1 - b;
                        if (lo) m0 &= ~(1u << b); else m1 &= ~(1u << b);
                        const int a = any ? st[k & (MAXSLOT - 1 | 63)] * 16 + slice16 : farA;
                        pair(*(const float4 *)(baseP + a), *(const float4 *)(baseV + a));
                    }
                }
                float f[3] = {px + vx, py + vy, pz + vz};
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    f[q] += dpp_f<0x101>(f[q]);
                    f[q] += dpp_f<0x102>(f[q]);
                    f[q] += dpp_f<0x104>(f[q]);
                }
                if (active && slice == 0) sink += f[0] + f[1] + f[2];
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = sink;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static float frand() { return (float)rand() / (float)RAND_MAX; }

int main()
{
    srand(7);
    // records row by row (x fastest), 7 FAR records after every row
    std::vector<float4> P(MAXREC, make_float4(FAR, FAR, FAR, 0.f)), V(MAXREC, make_float4(0, 0, 0, 0));
    int start[NCELL + 1];
    int rowEndPad = 7, n = 0;
    for (int z = 0; z < HZ; z++)
        for (int y = 0; y < HY; y++) {
            for (int x = 0; x < HX; x++) {
                start[(z * HY + y) * HX + x] = n;
                for (int k = 0; k < 8; k++) { // 2x2x2 lattice sites, jitter 0.02; some cells lose/gain one
                    if (k == 7 && frand() < 0.2f) continue;
                    float px = x + 0.25f + 0.5f * (k & 1) + (frand() - 0.5f) * 0.04f;
                    float py = y + 0.25f + 0.5f * ((k >> 1) & 1) + (frand() - 0.5f) * 0.04f;
                    float pz = z + 0.25f + 0.5f * (k >> 2) + (frand() - 0.5f) * 0.04f;
                    P[n] = make_float4(px, py, pz, 125.f);
                    V[n] = make_float4(frand(), frand(), frand(), 0.001f);
                    n++;
                }
                if (frand() < 0.2f) { // a ninth particle
                    P[n] = make_float4(x + frand(), y + frand(), z + frand(), 125.f);
                    V[n] = make_float4(frand(), frand(), frand(), 0.001f);
                    n++;
                }
            }
            n += rowEndPad;
        }
    start[NCELL] = n;
    if (n > MAXREC - 1) { printf("tile too large %d\n", n); return 1; }
    // cell end = next cell's start, except at a row end (pad)
    auto cellEnd = [&](int c) { return (c % HX == HX - 1) ? start[c + 1] - rowEndPad : start[c + 1]; };
    Tile T{};
    T.nrec = n;
    long totalSlots = 0, totalCand = 0;
    int unit = 0;
    for (int z = 1; z <= 2; z++)
        for (int y = 1; y <= 4; y++)
            for (int x = 1; x <= 4; x++, unit++) {
                const int c = (z * HY + y) * HX + x;
                T.unitSelf[unit] = start[c];
                T.unitT[unit] = std::min(8, cellEnd(c) - start[c]);
                int ns = 0;
                for (int dz = -1; dz <= 1; dz++)
                    for (int dy = -1; dy <= 1; dy++) {
                        const int c0 = ((z + dz) * HY + (y + dy)) * HX + x - 1;
                        const int s0 = start[c0], e0 = cellEnd(c0 + 2);
                        totalCand += e0 - s0;
                        for (int s = s0; s < e0; s += 8) T.slot[unit][ns++] = (unsigned short)s;
                    }
                if (ns > MAXSLOT) { printf("too many slots %d\n", ns); return 1; }
                T.unitSlots[unit] = ns;
                totalSlots += ns;
            }
    printf("tile: %d records, %.1f candidates and %.1f slots per unit\n", n, totalCand / 32.0, totalSlots / 32.0);

    const int maxBlocks = 256 * 16, iters = 40;
    float4 *dP, *dV;
    Tile *dT;
    float *out;
    unsigned *masks;
    unsigned long long *clk;
    (void)hipMalloc(&dP, MAXREC * 16);
    (void)hipMalloc(&dV, MAXREC * 16);
    (void)hipMalloc(&dT, sizeof(Tile));
    (void)hipMalloc(&out, maxBlocks * 256 * 4);
    (void)hipMalloc(&masks, (size_t)maxBlocks * NUNIT * 64 * 2 * 4);
    (void)hipMalloc(&clk, maxBlocks * 16);
    (void)hipMemcpy(dP, P.data(), MAXREC * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(dV, V.data(), MAXREC * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(dT, &T, sizeof(Tile), hipMemcpyHostToDevice);
    (void)hipMemset(masks, 0, (size_t)maxBlocks * NUNIT * 64 * 2 * 4);

    auto run = [&](auto kern, const char *name, int wgPerCU) {
        // LDS per workgroup such that exactly wgPerCU fit into the CU's 160 KiB
        const int blocks = 256 * wgPerCU * 2; // two full rounds of resident workgroups
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, (const void *)kern);
        const int want = 160 * 1024 / wgPerCU;
        const int dyn = std::max(0, (want - (int)fa.sharedSizeBytes) / 256 * 256 - 256);
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        kern<<<blocks, 256, dyn>>>(dP, dV, dT, out, masks, iters, clk);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        kern<<<blocks, 256, dyn>>>(dP, dV, dT, out, masks, iters, clk);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks);
        (void)hipMemcpy(h.data(), clk, 2 * blocks * 8, hipMemcpyDeviceToHost);
        double a = 0, r = 0;
        for (int i = 0; i < blocks; i++) { a += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
        const double ghz = a / r * 0.1;
        const double units = (double)blocks * NUNIT * iters;
        const double cyc = ms * 1e-3 * ghz * 1e9 * 1024.0 / units;
        std::vector<float> o(256);
        (void)hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost);
        double s = 0;
        for (float v : o) s += v;
        printf("%-11s wg/CU=%d vgpr=%3d %8.3f ms  %8.1f cycles/unit  %6.2f cycles/slot  (%.2f GHz)  check %.6e\n", name, wgPerCU, fa.numRegs, ms, cyc,
               cyc / (totalSlots / 32.0), ghz, s / iters);
    };
    for (int wg : {2, 3, 4, 6, 8}) {
        run(k_lab<0>, "dens", wg);
        run(k_lab<1>, "dens_mask", wg);
#ifndef LAB_ALIAS
        if (wg <= 3)
#endif
        {
            run(k_lab<1>, "dens_mask", wg); // the masks of this grid size
            run(k_lab<2>, "force", wg);
            run(k_lab<3>, "force_mask", wg);
        }
    }
    return 0;
}
