// sph_neighbors.h -- the two 27-cell neighbour loops: density/pressure and force.
//
// Reference semantics: ComputeDensityAndPressureJob (FluidSPHByCPUSystem.cs:48-104) and
// ComputeForceJob (:126-236); the device counterparts they replace are the HLSL kernels
// ComputeDensityAndPressure / ComputeForce (FluidSolver.compute:128-272).
//
// Mapping (DESIGN.md "neighbour kernels"): one wave owns 64 consecutive sorted particles and
// walks the cells that START inside that span.  For one cell:
//   * lanes 0..26 / 32..58 fetch the (start,end) ranges of the 27 fluid / 27 boundary cells;
//   * the union of those ranges (the candidate list, ~216 particles) is staged once into the
//     wave's LDS tile with coalesced 16-byte loads;
//   * candidates sit ACROSS the 64 lanes while the cell's particles (the targets) are taken eight
//     at a time as wave-uniform scalars, so one LDS read serves eight targets and the inner loop
//     is pure VALU on (SGPR target, VGPR candidate) operands.
// The hit predicates are evaluated exactly as the reference does in fp32 without contraction
// (this file is compiled with -ffp-contract=off):  d2 = (dx*dx + dy*dy) + dz*dz,  density:
// d2 < h2;  force: length < h  <=>  d2 < thrF (Mat::thrF).  Sums are reassociated (lane partial
// sums + butterfly), which the 1e-5 tolerance of BASELINE.json covers; neighbour COUNTS are exact.
#pragma once
#include "sph_device.h"

#define NB_DCAP 512 // candidate slots per staged batch, density kernel (8 KB of LDS)
#define NB_FCAP 320 // candidate slots per staged batch, force kernel (10 KB of LDS)
#define NB_MAXH 128 // per-target hit-list capacity between flushes (force kernel)
#define NB_FAR 1.0e30f

__device__ __forceinline__ int wave_incl_scan_i32(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// ranges of the 27 neighbour cells of the cell with dense key ckey (wave-uniform):
// lanes 0..26 -> fluid table, lanes 32..58 -> boundary table, other lanes empty.
__device__ __forceinline__ void cell_ranges(uint32_t ckey, const GridDesc &g,
                                            const int2 *__restrict__ table,
                                            const int2 *__restrict__ btable, int lane,
                                            int &start, int &len)
{
    const int cx = (int)(ckey % (uint32_t)g.gx);
    const int t = (int)(ckey / (uint32_t)g.gx);
    const int cy = t % g.gy, cz = t / g.gy;
    const int c = lane & 31;
    start = 0;
    len = 0;
    if (c < 27) {
        int x = cx + (c % 3) - 1, y = cy + ((c / 3) % 3) - 1, z = cz + (c / 9) - 1;
        x = x < 0 ? x + g.gx : (x >= g.gx ? x - g.gx : x);
        y = y < 0 ? y + g.gy : (y >= g.gy ? y - g.gy : y);
        z = z < 0 ? z + g.gz : (z >= g.gz ? z - g.gz : z);
        const uint32_t nk = (uint32_t)((z * g.gy + y) * g.gx + x);
        const int2 *tb = (lane >> 5) ? btable : table;
        if (tb) {
            const int2 r = tb[nk];
            start = r.x;
            len = r.y - r.x;
        }
    }
}

// candidate slot q (in the concatenation of the 64 lanes' ranges) -> owning lane and source index
__device__ __forceinline__ void locate_slot(int q, int off, int start, int &rid, int &j)
{
    int lo = 0, hi = 63; // last lane whose exclusive offset is <= q
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const int mid = (lo + hi + 1) >> 1;
        const int o = __shfl(off, mid, 64);
        if (o <= q) lo = mid; else hi = mid - 1;
    }
    rid = lo;
    j = __shfl(start, lo, 64) + (q - __shfl(off, lo, 64));
}

// sums eight per-lane accumulators over the wave; on return the lane holds the total of
// accumulator u = lane >> 3 (bits 5,4,3 of the lane id), replicated over its 8 lanes.
__device__ __forceinline__ float reduce8_transposed(const float (&a)[8], int lane)
{
    float v4[4], v2[2], v;
    const bool h32 = lane & 32, h16 = lane & 16, h8 = lane & 8;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const float keep = h32 ? a[u + 4] : a[u];
        const float send = h32 ? a[u] : a[u + 4];
        v4[u] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const float keep = h16 ? v4[u + 2] : v4[u];
        const float send = h16 ? v4[u] : v4[u + 2];
        v2[u] = keep + __shfl_xor(send, 16, 64);
    }
    {
        const float keep = h8 ? v2[1] : v2[0];
        const float send = h8 ? v2[0] : v2[1];
        v = keep + __shfl_xor(send, 8, 64);
    }
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// density + pressure
template <bool COUNT>
__global__ __launch_bounds__(64) void k_density(
    const float4 *__restrict__ posm, float4 *__restrict__ velq, const uint32_t *__restrict__ mat,
    const uint32_t *__restrict__ skeys, const int2 *__restrict__ table,
    const float4 *__restrict__ bposm, const int2 *__restrict__ btable, GridDesc g,
    const Mat *__restrict__ mats, float *__restrict__ rho, float *__restrict__ prs,
    int *__restrict__ counts, int n)
{
    __shared__ float4 cand[NB_DCAP];
    __shared__ float tacc[64];
    __shared__ int tcnt[64];

    const int lane = threadIdx.x;
    const int base = blockIdx.x * 64;
    const int s = base + lane;
    const uint32_t key = s < n ? skeys[s] : 0xffffffffu;
    bool head = false;
    if (s < n) head = (s == 0) || (key != skeys[s - 1]);
    unsigned long long heads = __ballot(head);

    while (heads) {
        const int b = __builtin_ctzll(heads);
        heads &= heads - 1;
        const uint32_t ckey = (uint32_t)bcast_i((int)key, b);
        const int ts = base + b;
        const int te = uniform_i(table[ckey].y);

        int rstart, rlen;
        cell_ranges(ckey, g, table, btable, lane, rstart, rlen);
        const int incl = wave_incl_scan_i32(rlen, lane);
        const int off = incl - rlen;
        const int total = bcast_i(incl, 63);

        for (int tb = ts; tb < te; tb += 64) {
            const int nt = min(64, te - tb);
            float4 tp = make_float4(0.f, 0.f, 0.f, 0.f);
            uint32_t tm = 0;
            if (lane < nt) {
                tp = posm[tb + lane];
                tm = mat[tb + lane];
            }
            const float th2 = mats[tm].h2;
            tacc[lane] = 0.f;
            if (COUNT) tcnt[lane] = 0;

            for (int bs = 0; bs < total; bs += NB_DCAP) {
                const int bc = min(NB_DCAP, total - bs);
                const int nch = (bc + 63) >> 6;
                if (tb == ts || total > NB_DCAP) {
                    __builtin_amdgcn_wave_barrier();
                    for (int q = lane; q < nch * 64; q += 64) {
                        float4 cd = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
                        int rid, j;
                        locate_slot(bs + q, off, rstart, rid, j);
                        if (q < bc) cd = (rid >= 32 ? bposm : posm)[j];
                        cand[q] = cd;
                    }
                }
                __syncthreads(); // single-wave block: orders the LDS tile, no s_barrier needed

                for (int tg = 0; tg < nt; tg += 8) {
                    const int ntg = min(8, nt - tg);
                    float xi[8], yi[8], zi[8], h2i[8], acc[8];
                    int cn[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int src = min(tg + u, 63);
                        xi[u] = bcast_f(tp.x, src);
                        yi[u] = bcast_f(tp.y, src);
                        zi[u] = bcast_f(tp.z, src);
                        h2i[u] = bcast_f(th2, src);
                        acc[u] = 0.f;
                        cn[u] = 0;
                    }
                    for (int c = 0; c < nch; c++) {
                        const float4 cd = cand[c * 64 + lane];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            if (u < ntg) {
                                const float dx = xi[u] - cd.x;
                                const float dy = yi[u] - cd.y;
                                const float dz = zi[u] - cd.z;
                                const float d2 = (dx * dx + dy * dy) + dz * dz;
                                const float w = fmaxf(h2i[u] - d2, 0.f);
                                acc[u] = __builtin_fmaf(w * w, w, acc[u]);
                                if (COUNT) cn[u] += __popcll(__ballot(d2 < h2i[u]));
                            }
                        }
                    }
                    const float tot = reduce8_transposed(acc, lane);
                    if ((lane & 7) == 0 && (lane >> 3) < ntg) tacc[tg + (lane >> 3)] += tot;
                    if (COUNT) {
                        int mine = 0;
#pragma unroll
                        for (int u = 0; u < 8; u++)
                            if (lane == u) mine = cn[u];
                        if (lane < ntg) tcnt[tg + lane] += mine;
                    }
                }
                __syncthreads();
            }

            if (lane < nt) {
                const Mat m = mats[tm];
                // rho_i = m_i*315/(64*pi*h^9) * sum (h2-d2)^3 ; p_i = max(k (rho_i - rho0), 0)
                const float density = m.poly6 * tacc[lane];
                const float pressure = fmaxf(m.gasConstant * (density - m.restDensity), 0.f);
                rho[tb + lane] = density;
                prs[tb + lane] = pressure;
                velq[tb + lane].w = pressure / (density * density);
                if (COUNT) counts[tb + lane] = tcnt[lane];
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// force: pressure (spiky) + viscosity + gravity
template <bool COUNT>
__global__ __launch_bounds__(64) void k_force(
    const float4 *__restrict__ posm, const float4 *__restrict__ velq,
    const uint32_t *__restrict__ mat, const float *__restrict__ rho,
    const float *__restrict__ prs, const uint32_t *__restrict__ skeys,
    const int2 *__restrict__ table, const float4 *__restrict__ bposm,
    const int2 *__restrict__ btable, GridDesc g, const Mat *__restrict__ mats, float gx, float gy,
    float gz, float4 *__restrict__ force, int *__restrict__ counts, int n)
{
    __shared__ float4 candP[NB_FCAP]; // x, y, z, mass
    __shared__ float4 candV[NB_FCAP]; // vx, vy, vz, q = p/rho^2 (boundary: 0,0,0,0)
    __shared__ unsigned short hits[8][NB_MAXH];
    __shared__ float taccx[64], taccy[64], taccz[64];
    __shared__ int tcnt[64];

    const int lane = threadIdx.x;
    const int base = blockIdx.x * 64;
    const int s = base + lane;
    const uint32_t key = s < n ? skeys[s] : 0xffffffffu;
    bool head = false;
    if (s < n) head = (s == 0) || (key != skeys[s - 1]);
    unsigned long long heads = __ballot(head);
    const int myu = lane & 7, mysl = lane >> 3;

    while (heads) {
        const int b = __builtin_ctzll(heads);
        heads &= heads - 1;
        const uint32_t ckey = (uint32_t)bcast_i((int)key, b);
        const int ts = base + b;
        const int te = uniform_i(table[ckey].y);

        int rstart, rlen;
        cell_ranges(ckey, g, table, btable, lane, rstart, rlen);
        const int incl = wave_incl_scan_i32(rlen, lane);
        const int off = incl - rlen;
        const int total = bcast_i(incl, 63);
        const int selfOff = bcast_i(off, 13); // lane 13 = the cell itself (dx=dy=dz=0), fluid

        for (int tb = ts; tb < te; tb += 64) {
            const int nt = min(64, te - tb);
            // lane t holds target tb+t
            float4 tp = make_float4(0.f, 0.f, 0.f, 0.f), tv = tp;
            float trho = 1.f, tprs = 0.f;
            uint32_t tm = 0;
            if (lane < nt) {
                tp = posm[tb + lane];
                tv = velq[tb + lane];
                trho = rho[tb + lane];
                tprs = prs[tb + lane];
                tm = mat[tb + lane];
            }
            const Mat m = mats[tm];
            const float tPi = tprs / (trho * trho);       // p_i / rho_i^2
            const float tCp = trho * m.spiky;             // rho_i * 45/(pi h^6)
            const float tCv = m.viscosity / trho * m.spiky; // mu_i / rho_i * 45/(pi h^6)
            taccx[lane] = 0.f;
            taccy[lane] = 0.f;
            taccz[lane] = 0.f;
            tcnt[lane] = 0;

            for (int bs = 0; bs < total; bs += NB_FCAP) {
                const int bc = min(NB_FCAP, total - bs);
                const int nch = (bc + 63) >> 6;
                if (tb == ts || total > NB_FCAP) {
                    __builtin_amdgcn_wave_barrier();
                    for (int q = lane; q < nch * 64; q += 64) {
                        float4 cp = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
                        float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                        int rid, j;
                        locate_slot(bs + q, off, rstart, rid, j);
                        if (q < bc) {
                            if (rid >= 32) {
                                cp = bposm[j];
                            } else {
                                cp = posm[j];
                                cv = velq[j];
                            }
                        }
                        candP[q] = cp;
                        candV[q] = cv;
                    }
                }
                __syncthreads();

                for (int tg = 0; tg < nt; tg += 8) {
                    const int ntg = min(8, nt - tg);
                    // phase-2 operands: this lane works for target tg + (lane & 7)
                    const int tsrc = min(tg + myu, 63);
                    const float px = __shfl(tp.x, tsrc, 64), py = __shfl(tp.y, tsrc, 64),
                                pz = __shfl(tp.z, tsrc, 64);
                    const float pvx = __shfl(tv.x, tsrc, 64), pvy = __shfl(tv.y, tsrc, 64),
                                pvz = __shfl(tv.z, tsrc, 64);
                    const float pPi = __shfl(tPi, tsrc, 64), ph = __shfl(m.h, tsrc, 64);
                    const float pCp = __shfl(tCp, tsrc, 64), pCv = __shfl(tCv, tsrc, 64);
                    float fpx = 0.f, fpy = 0.f, fpz = 0.f, fvx = 0.f, fvy = 0.f, fvz = 0.f;

                    // phase-1 operands: wave-uniform targets
                    float xi[8], yi[8], zi[8], thr[8];
                    int cn[8], tot[8], selfRel[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int src = min(tg + u, 63);
                        xi[u] = bcast_f(tp.x, src);
                        yi[u] = bcast_f(tp.y, src);
                        zi[u] = bcast_f(tp.z, src);
                        thr[u] = bcast_f(m.thrF, src);
                        cn[u] = 0;
                        tot[u] = 0;
                        selfRel[u] = selfOff + (tb + tg + u - ts) - bs; // slot of the target itself
                    }

                    auto flush = [&]() {
                        // phase 2: lane (u = lane&7, slice = lane>>3) takes every 8th hit of
                        // target u and evaluates the pair force
                        __syncthreads();
                        int myc = 0, maxc = 0;
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            if (myu == u) myc = cn[u];
                            maxc = max(maxc, cn[u]);
                        }
                        for (int it = 0; it * 8 < maxc; it++) {
                            const int idx = it * 8 + mysl;
                            if (idx < myc) {
                                const int slot = hits[myu][idx];
                                const float4 cp = candP[slot];
                                const float4 cv = candV[slot];
                                const float dx = px - cp.x, dy = py - cp.y, dz = pz - cp.z;
                                const float d2 = (dx * dx + dy * dy) + dz * dz;
                                const float r = __builtin_amdgcn_sqrtf(d2);
                                const float hr = ph - r;
                                // m_j (p_i/rho_i^2 + p_j/rho_j^2) (h-r)^2 / r
                                const float a = cp.w * (pPi + cv.w) * (hr * hr) *
                                                __builtin_amdgcn_rcpf(r);
                                fpx = __builtin_fmaf(a, dx, fpx);
                                fpy = __builtin_fmaf(a, dy, fpy);
                                fpz = __builtin_fmaf(a, dz, fpz);
                                const float bq = cp.w * hr; // m_j (h-r)
                                fvx = __builtin_fmaf(bq, cv.x - pvx, fvx);
                                fvy = __builtin_fmaf(bq, cv.y - pvy, fvy);
                                fvz = __builtin_fmaf(bq, cv.z - pvz, fvz);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++) cn[u] = 0;
                        __syncthreads();
                    };

                    for (int c = 0; c < nch; c++) {
                        const float4 cd = candP[c * 64 + lane];
                        int worst = 0;
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            if (u < ntg) {
                                const float dx = xi[u] - cd.x;
                                const float dy = yi[u] - cd.y;
                                const float dz = zi[u] - cd.z;
                                const float d2 = (dx * dx + dy * dy) + dz * dz;
                                unsigned long long mask = __ballot(d2 < thr[u]);
                                const int sr = selfRel[u] - c * 64;
                                if (sr >= 0 && sr < 64) mask &= ~(1ull << sr); // index != j
                                if (mask) {
                                    const int p = cn[u] + mbcnt64(mask);
                                    if ((mask >> lane) & 1ull)
                                        hits[u][p] = (unsigned short)(c * 64 + lane);
                                    const int k = __popcll(mask);
                                    cn[u] += k;
                                    tot[u] += k;
                                }
                                worst = max(worst, cn[u]);
                            }
                        }
                        if (worst > NB_MAXH - 64) flush();
                    }
                    flush();

                    // fold the 8 slices of every target, scale, accumulate
                    float fx = pCp * fpx + pCv * fvx;
                    float fy = pCp * fpy + pCv * fvy;
                    float fz = pCp * fpz + pCv * fvz;
#pragma unroll
                    for (int d = 8; d < 64; d <<= 1) {
                        fx += __shfl_xor(fx, d, 64);
                        fy += __shfl_xor(fy, d, 64);
                        fz += __shfl_xor(fz, d, 64);
                    }
                    int mine = 0;
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (lane == u) mine = tot[u];
                    if (lane < ntg) {
                        taccx[tg + lane] += fx;
                        taccy[tg + lane] += fy;
                        taccz[tg + lane] += fz;
                        tcnt[tg + lane] += mine;
                    }
                }
                __syncthreads();
            }

            if (lane < nt) {
                // forces = forcePressure + forceViscosity + gravity * rho_i  (:140,:202)
                force[tb + lane] = make_float4(taccx[lane] + gx * trho, taccy[lane] + gy * trho,
                                               taccz[lane] + gz * trho, 0.f);
                if (COUNT) counts[tb + lane] = tcnt[lane];
            }
            __syncthreads();
        }
    }
}
