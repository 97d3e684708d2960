// sph_neighbors.h -- the two 27-cell neighbour loops: density/pressure and force.
//
// Reference semantics: ComputeDensityAndPressureJob (FluidSPHByCPUSystem.cs:48-104) and
// ComputeForceJob (:126-236); the device counterparts they replace are the HLSL kernels
// ComputeDensityAndPressure / ComputeForce (FluidSolver.compute:128-272).
//
// Mapping (DESIGN.md "neighbour kernels").  One wave owns 64 consecutive sorted particles and
// walks the cells that START inside that span.  For one cell with T particles (the targets):
//   * the 64 lanes are split as T targets x S = 64/T candidate slices: lane -> (u, slice);
//     a lane keeps its target in VGPRs (every operand of the inner loop is a VGPR: on gfx950
//     VALU ops with an SGPR/constant operand issue at half rate, tools/microbench_valu.hip);
//   * because the sort key is x-fastest, the three cells (x-1, x, x+1) of one (y, z) row are
//     contiguous in the sorted arrays: 9 fluid + 9 boundary row ranges instead of 54 cell
//     ranges (cells on the periodic seam fall back to 27 + 27 single-cell ranges);
//   * a lane walks every range with stride S, so each load instruction fetches S consecutive
//     16-byte records, each shared by the T lanes of equal slice (one 128-byte line for T = 8);
//   * per-target partial sums are folded over the S slices through LDS once per cell.
// Force kernel: phase 1 is the same walk with the distance test only; hits are compacted per
// target with wave ballots into an LDS tile of 32-byte records, which phase 2 consumes with all
// lanes busy (a hit costs ~10x a miss, and only ~14 % of the candidates are hits).
//
// Predicates: force uses exactly the reference's fp32, non-contracted arithmetic (this file is
// compiled with -ffp-contract=off): d2 = (dx*dx + dy*dy) + dz*dz and  length < h  <=>  d2 < thrF
// (Mat::thrF).  The COUNT variants evaluate d2 < h2 the same way, so neighbour COUNTS are
// bit-exact.  Sums are reassociated (per-slice partial sums), which the 1e-5 tolerance covers.
#pragma once
#include "sph_device.h"

#define NB_FAR 1.0e30f
#define NB_HITCAP 256 // hit records per wave between flushes (force kernel): T * 4S <= 256

// Candidate ranges of the cell with dense key ckey (wave-uniform).  Returns the ballot of the
// lanes that hold a non-empty range [rs, re); lanes >= boundaryLane index the boundary arrays.
__device__ __forceinline__ unsigned long long build_ranges(uint32_t ckey, const GridDesc &g,
                                                           const int2 *__restrict__ table,
                                                           const int2 *__restrict__ btable,
                                                           int lane, int &rs, int &re,
                                                           int &boundaryLane)
{
    const int cx = (int)(ckey % (uint32_t)g.gx);
    const int t = (int)(ckey / (uint32_t)g.gx);
    const int cy = t % g.gy, cz = t / g.gy;
    rs = 0;
    re = 0;
    if (cx >= 1 && cx + 1 < g.gx) {
        // rows: lanes 0..8 fluid, 16..24 boundary; row r covers cells (cx-1..cx+1, y, z)
        boundaryLane = 16;
        const int r = lane & 15, which = lane >> 4;
        const int2 *tb = which == 0 ? table : (which == 1 ? btable : nullptr);
        if (r < 9 && tb) {
            int y = cy + (r % 3) - 1, z = cz + (r / 3) - 1;
            y = y < 0 ? y + g.gy : (y >= g.gy ? y - g.gy : y);
            z = z < 0 ? z + g.gz : (z >= g.gz ? z - g.gz : z);
            const int2 *p = tb + ((z * g.gy + y) * g.gx + cx - 1);
            const int2 a = p[0], b = p[1], c = p[2]; // empty cells hold (0,0)
            rs = a.y > a.x ? a.x : (b.y > b.x ? b.x : c.x);
            re = c.y > c.x ? c.y : (b.y > b.x ? b.y : a.y);
        }
    } else {
        // on the periodic seam: single cells, lanes 0..26 fluid, 32..58 boundary
        boundaryLane = 32;
        const int c = lane & 31;
        const int2 *tb = (lane >> 5) ? btable : table;
        if (c < 27 && tb) {
            int x = cx + (c % 3) - 1, y = cy + ((c / 3) % 3) - 1, z = cz + (c / 9) - 1;
            x = x < 0 ? x + g.gx : (x >= g.gx ? x - g.gx : x);
            y = y < 0 ? y + g.gy : (y >= g.gy ? y - g.gy : y);
            z = z < 0 ? z + g.gz : (z >= g.gz ? z - g.gz : z);
            const int2 a = tb[(z * g.gy + y) * g.gx + x];
            rs = a.x;
            re = a.y;
        }
    }
    return __ballot(re > rs);
}

// lane -> (target slot u, slice) for T targets and S = 64 / T slices
__device__ __forceinline__ void split_lane(int lane, int S, int &u, int &slice)
{
    u = (int)(((float)lane + 0.5f) * (1.0f / (float)S)); // exact for lane < 64
    slice = lane - u * S;
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
    __shared__ float red[64];
    __shared__ int redc[64];

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

        int rs, re, boundaryLane;
        const unsigned long long rmask = build_ranges(ckey, g, table, btable, lane, rs, re,
                                                      boundaryLane);

        for (int tb = ts; tb < te; tb += 64) {
            const int T = min(64, te - tb);
            const int S = 64 / T;
            int u, slice;
            split_lane(lane, S, u, slice);
            const bool active = u < T;
            const int ti = tb + (active ? u : 0);
            const float4 tp = posm[ti];
            const uint32_t tm = mat[ti];
            const float h2 = mats[tm].h2;
            const float negInvH2 = -1.0f / h2;
            float acc = 0.f;
            int cnt = 0;

            unsigned long long rm = rmask;
            while (rm) {
                const int r = __builtin_ctzll(rm);
                rm &= rm - 1;
                const int s0 = bcast_i(rs, r), e0 = bcast_i(re, r);
                const float4 *__restrict__ src = r >= boundaryLane ? bposm : posm;
                for (int k0 = s0; k0 < e0; k0 += 4 * S) {
                    float4 cd[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int k = k0 + q * S + slice;
                        cd[q] = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
                        if (k < e0) cd[q] = src[k];
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (k0 + q * S < e0) { // wave-uniform
                            const float dx = tp.x - cd[q].x;
                            const float dy = tp.y - cd[q].y;
                            const float dz = tp.z - cd[q].z;
                            if (COUNT) {
                                const float d2 = (dx * dx + dy * dy) + dz * dz;
                                cnt += (d2 < h2) ? 1 : 0;
                                const float w = fmaxf(h2 - d2, 0.f);
                                acc = __builtin_fmaf(w * w, w, acc);
                            } else {
                                // w/h2 = clamp(1 - d2/h2) in one full-rate op; out-of-range and
                                // padding candidates (d2 = inf) clamp to 0
                                const float d2 = __builtin_fmaf(
                                    dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                                const float w = __builtin_amdgcn_fmed3f(
                                    __builtin_fmaf(d2, negInvH2, 1.0f), 0.f, 1.0f);
                                acc = __builtin_fmaf(w * w, w, acc);
                            }
                        }
                    }
                }
            }

            // fold the S slices of every target
            red[lane] = acc;
            if (COUNT) redc[lane] = cnt;
            __syncthreads();
            if (active && slice == 0) {
                float sum = 0.f;
                int c = 0;
                for (int q = 0; q < S; q++) {
                    sum += red[lane + q];
                    if (COUNT) c += redc[lane + q];
                }
                const Mat m = mats[tm];
                // rho_i = m_i*315/(64*pi*h^9) * sum (h2-d2)^3 ; p_i = max(k (rho_i - rho0), 0)
                const float scale = COUNT ? m.poly6 : m.poly6 * (h2 * h2 * h2);
                const float density = scale * sum;
                const float pressure = fmaxf(m.gasConstant * (density - m.restDensity), 0.f);
                rho[ti] = density;
                prs[ti] = pressure;
                velq[ti].w = pressure / (density * density);
                if (COUNT) counts[ti] = c;
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
    __shared__ float4 hitP[NB_HITCAP]; // dx, dy, dz, m_j
    __shared__ float4 hitV[NB_HITCAP]; // vx, vy, vz, q_j = p_j/rho_j^2 (boundary: 0,0,0,0)
    __shared__ float redx[64], redy[64], redz[64];
    __shared__ int redc[64];

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

        int rs, re, boundaryLane;
        const unsigned long long rmask = build_ranges(ckey, g, table, btable, lane, rs, re,
                                                      boundaryLane);

        for (int tb = ts; tb < te; tb += 64) {
            const int T = min(64, te - tb);
            const int S = 64 / T;
            const int capU = 4 * S; // hit records per target between flushes
            int u, slice;
            split_lane(lane, S, u, slice);
            const bool active = u < T;
            const int ti = tb + (active ? u : 0);
            const float4 tp = posm[ti];
            const float4 tv = velq[ti];
            const float trho = rho[ti], tprs = prs[ti];
            const Mat m = mats[mat[ti]];
            const float Pi = tprs / (trho * trho); // p_i / rho_i^2
            // lanes that work for the same target: S consecutive lanes starting at u*S
            const unsigned long long mymask =
                active ? ((S >= 64 ? ~0ull : ((1ull << S) - 1ull)) << (u * S)) : 0ull;
            const int listBase = u * capU;
            int cnt = 0, total = 0;
            float fpx = 0.f, fpy = 0.f, fpz = 0.f, fvx = 0.f, fvy = 0.f, fvz = 0.f;

            auto flush = [&]() {
                // phase 2: every lane takes each S-th hit of its own target
                __syncthreads();
                for (int it = 0; __ballot(it * S + slice < cnt) != 0ull; it++) {
                    const int idx = it * S + slice;
                    if (idx < cnt) {
                        const float4 hp = hitP[listBase + idx];
                        const float4 hv = hitV[listBase + idx];
                        const float d2 = (hp.x * hp.x + hp.y * hp.y) + hp.z * hp.z;
                        const float r = __builtin_amdgcn_sqrtf(d2);
                        const float hr = m.h - r;
                        // m_j (p_i/rho_i^2 + p_j/rho_j^2) (h-r)^2 / r
                        const float a = hp.w * (Pi + hv.w) * (hr * hr) * __builtin_amdgcn_rcpf(r);
                        fpx = __builtin_fmaf(a, hp.x, fpx);
                        fpy = __builtin_fmaf(a, hp.y, fpy);
                        fpz = __builtin_fmaf(a, hp.z, fpz);
                        const float bq = hp.w * hr; // m_j (h-r)
                        fvx = __builtin_fmaf(bq, hv.x - tv.x, fvx);
                        fvy = __builtin_fmaf(bq, hv.y - tv.y, fvy);
                        fvz = __builtin_fmaf(bq, hv.z - tv.z, fvz);
                    }
                }
                cnt = 0;
                __syncthreads();
            };

            unsigned long long rm = rmask;
            while (rm) {
                const int r = __builtin_ctzll(rm);
                rm &= rm - 1;
                const int s0 = bcast_i(rs, r), e0 = bcast_i(re, r);
                const bool fluid = r < boundaryLane;
                for (int k0 = s0; k0 < e0; k0 += 2 * S) {
                    float4 cp[2], cv[2];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int k = k0 + q * S + slice;
                        cp[q] = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
                        cv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (k < e0) {
                            if (fluid) {
                                cp[q] = posm[k];
                                cv[q] = velq[k];
                            } else {
                                cp[q] = bposm[k];
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        if (k0 + q * S < e0) { // wave-uniform
                            const int k = k0 + q * S + slice;
                            const float dx = tp.x - cp[q].x;
                            const float dy = tp.y - cp[q].y;
                            const float dz = tp.z - cp[q].z;
                            const float d2 = (dx * dx + dy * dy) + dz * dz;
                            // length(vij) < h  (and index != j for fluid neighbours, :152)
                            const bool hit = active && (d2 < m.thrF) && !(fluid && k == ti);
                            const unsigned long long bal = __ballot(hit);
                            if (bal) {
                                const unsigned long long mine = bal & mymask;
                                if (hit) {
                                    const int p = listBase + cnt + mbcnt64(mine);
                                    hitP[p] = make_float4(dx, dy, dz, cp[q].w);
                                    hitV[p] = cv[q];
                                }
                                const int k2 = __popcll(mine);
                                cnt += k2;
                                total += k2;
                                if (__ballot(cnt > capU - S) != 0ull) flush();
                            }
                        }
                    }
                }
            }
            flush();

            // scale, fold the S slices of every target, add gravity
            const float Cp = trho * m.spiky;               // rho_i * 45/(pi h^6)
            const float Cv = m.viscosity / trho * m.spiky; // mu_i / rho_i * 45/(pi h^6)
            redx[lane] = Cp * fpx + Cv * fvx;
            redy[lane] = Cp * fpy + Cv * fvy;
            redz[lane] = Cp * fpz + Cv * fvz;
            __syncthreads();
            if (active && slice == 0) {
                float fx = 0.f, fy = 0.f, fz = 0.f;
                for (int q = 0; q < S; q++) {
                    fx += redx[lane + q];
                    fy += redy[lane + q];
                    fz += redz[lane + q];
                }
                // forces = forcePressure + forceViscosity + gravity * rho_i  (:140,:202)
                force[ti] = make_float4(fx + gx * trho, fy + gy * trho, fz + gz * trho, 0.f);
                if (COUNT) counts[ti] = total;
            }
            __syncthreads();
        }
    }
}
