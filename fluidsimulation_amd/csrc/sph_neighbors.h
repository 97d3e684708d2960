// sph_neighbors.h -- the two 27-cell neighbour loops: density/pressure and force.
//
// Reference semantics: ComputeDensityAndPressureJob (FluidSPHByCPUSystem.cs:48-104) and
// ComputeForceJob (:126-236); the device counterparts they replace are the HLSL kernels
// ComputeDensityAndPressure / ComputeForce (FluidSolver.compute:128-272), which walk the table
// per thread and gather every neighbour from global memory.
//
// Mapping (DESIGN.md "neighbour kernels").  One 256-thread workgroup owns one TILE of 4x4x2
// cells (the sort key groups particles by tile, sph_device.h):
//   1. threads 0..143 look up the (start,end) of the 6x6x4 halo of cells around the tile in the
//      fluid and boundary tables; a block scan turns the counts into LDS offsets;
//   2. the four waves copy the halo's particles into the LDS tile with coalesced 16-byte loads,
//      cell by cell, x-runs contiguous: each particle is read from L2/HBM ~4.5 times per pass
//      instead of 27 times;
//   3. each wave takes 8 cells of the tile.  For a cell with T particles (the targets) the 64
//      lanes split as T targets x S = 64/T candidate slices; a lane keeps its target in VGPRs
//      (every inner-loop operand is a VGPR: VALU ops with an SGPR or constant operand issue at
//      half rate on gfx950, tools/microbench_valu.hip) and walks the 9 LDS row windows
//      (x-1..x+1 of one (y,z)) with stride S; one ds_read_b128 feeds S candidates to T lanes;
//   4. per-target partial sums are folded over the S slices through a wave-private LDS strip.
// A tile whose halo does not fit the LDS tile (pile-ups) takes the same walk straight from
// global memory (tile_fallback), so any particle distribution is handled.
//
// Arithmetic.  The production kernels are branch-free: density uses w/h2 = clamp(1 - d2/h2),
// force uses (h-r)/h = clamp(1 - r/h), so candidates outside the support contribute exactly 0
// and no per-pair predicate is evaluated (pairs within an ulp of the support radius contribute
// O(ulp^2), far below the 1e-5 tolerance of BASELINE.json).  The COUNT variants additionally
// evaluate the reference's predicates exactly -- fp32, non-contracted (this file is compiled
// with -ffp-contract=off): d2 = (dx*dx + dy*dy) + dz*dz; density d2 < h2; force
// length < h  <=>  d2 < thrF (Mat::thrF) and index != j -- so neighbour COUNTS are bit-exact.
#pragma once
#include "sph_device.h"

#define NB_FAR 1.0e18f // padding coordinate: d2 ~ 3e36 stays finite, every kernel weight clamps to 0
#define NB_HX (SPH_TX + 2)
#define NB_HY (SPH_TY + 2)
#define NB_HZ (SPH_TZ + 2)
#define NB_HCELLS (NB_HX * NB_HY * NB_HZ) // 144
#define NB_THREADS 256
#define NB_WAVES 4
#define NB_CELLS_PER_WAVE (SPH_TCELLS / NB_WAVES)
#define NB_DCAP 2048 // LDS records per tile, density kernel (32 KB)
#define NB_FCAP 1792 // LDS records per tile, force kernel (2 x 28 KB)
#define NB_TCAP 512  // tile's own particles whose constants are cached in LDS
#define NB_NQ 4      // candidate slots per row window issued back to back

struct TileShared {
    int fs[NB_HCELLS];       // fluid start (global sorted index) of the halo cell
    int fc[NB_HCELLS];       // fluid count
    int bs[NB_HCELLS];       // boundary start
    int bc[NB_HCELLS];       // boundary count
    int pref[NB_HCELLS + 1]; // LDS offset of the halo cell's records (fluid first, then boundary)
    int scan[NB_WAVES];
    float red[NB_WAVES][4][64];
    float4 tconst[NB_TCAP];  // per-target constants of the tile's own particles
    float tthr[NB_TCAP];     // COUNT variants: the exact predicate threshold (h2 / thrF)
};

__device__ __forceinline__ int wrapi(int c, int g) { return c < 0 ? c + g : (c >= g ? c - g : c); }

// lane -> (target slot u, slice) for T targets and S = 64 / T slices
__device__ __forceinline__ void split_lane(int lane, int S, int &u, int &slice)
{
    u = (int)(((float)lane + 0.5f) * (1.0f / (float)S)); // exact for lane < 64
    slice = lane - u * S;
}

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in order; this only stops the compiler from moving
    // them across the point where other lanes' data is consumed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// step 1: halo table look-ups + offsets.  Returns the number of records the halo holds.
__device__ __forceinline__ int tile_setup(TileShared &sh, int tile, const GridDesc &g,
                                          const int2 *__restrict__ table,
                                          const int2 *__restrict__ btable)
{
    const int tid = threadIdx.x;
    const int tx = tile % g.ntx, t2 = tile / g.ntx, ty = t2 % g.nty, tz = t2 / g.nty;
    int cnt = 0;
    if (tid < NB_HCELLS) {
        const int hx = tid % NB_HX, hy = (tid / NB_HX) % NB_HY, hz = tid / (NB_HX * NB_HY);
        const int x = wrapi(tx * SPH_TX + hx - 1, g.gx), y = wrapi(ty * SPH_TY + hy - 1, g.gy),
                  z = wrapi(tz * SPH_TZ + hz - 1, g.gz);
        const uint32_t k = tiled_key(x, y, z, g);
        const int2 f = table[k];
        int2 b = make_int2(0, 0);
        if (btable) b = btable[k];
        sh.fs[tid] = f.x;
        sh.fc[tid] = f.y - f.x;
        sh.bs[tid] = b.x;
        sh.bc[tid] = b.y - b.x;
        cnt = (f.y - f.x) + (b.y - b.x);
    }
    // block exclusive scan
    const int lane = tid & 63, w = tid >> 6;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) sh.scan[w] = incl;
    __syncthreads();
    int basew = 0, total = 0;
#pragma unroll
    for (int i = 0; i < NB_WAVES; i++) {
        const int t = sh.scan[i];
        if (i < w) basew += t;
        total += t;
    }
    if (tid < NB_HCELLS) sh.pref[tid] = basew + incl - cnt;
    if (tid == NB_HCELLS) sh.pref[NB_HCELLS] = total;
    __syncthreads();
    return total;
}

// step 2: copy the halo's particles into LDS.  One thread per record; the owning halo cell is
// found by bisection over the 145 offsets, so all loads of a thread are independent and the
// compiler can keep several in flight.
__device__ __forceinline__ int tile_locate(const TileShared &sh, int q)
{
    int lo = 0, hi = NB_HCELLS - 1; // last halo cell whose offset is <= q
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int mid = (lo + hi + 1) >> 1;
        if (sh.pref[mid] <= q) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// LDS-DMA: global_load_lds_dwordx4 writes 16 bytes per lane to  M0 base + lane*16  with no VGPR
// in between, so one wave-instruction moves 64 consecutive records (1 KiB) of the tile image.
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <bool WITH_V>
__device__ __forceinline__ void tile_stage(const TileShared &sh, int total, float4 *recP,
                                           float4 *recV, const float4 *__restrict__ posm,
                                           const float4 *__restrict__ velq,
                                           const float4 *__restrict__ bposm,
                                           const float4 *__restrict__ zeros)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int q0 = 0; q0 < total; q0 += NB_THREADS) {
        const int qw = q0 + w * 64; // first record of this wave-instruction (wave-uniform)
        const int q = qw + lane;
        if (q < total) {
            const int h = tile_locate(sh, q);
            const int k = q - sh.pref[h], fc = sh.fc[h];
            const float4 *gp, *gv;
            if (k < fc) {
                gp = posm + (sh.fs[h] + k);
                gv = velq + (sh.fs[h] + k);
            } else {
                gp = bposm + (sh.bs[h] + (k - fc));
                gv = zeros; // boundary: v = 0, q = 0
            }
            __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(recP + qw), 16, 0, 0);
            if (WITH_V) __builtin_amdgcn_global_load_lds((gptr_t)gv, (lptr_t)(recV + qw), 16, 0, 0);
        }
    }
}

// halo index of the centre cell `lc` (0..31) of the tile and of the first cell of its row window
__device__ __forceinline__ int halo_center(int lc)
{
    const int lx = lc % SPH_TX, ly = (lc / SPH_TX) % SPH_TY, lz = lc / (SPH_TX * SPH_TY);
    return ((lz + 1) * NB_HY + (ly + 1)) * NB_HX + (lx + 1);
}
__device__ __forceinline__ int halo_row(int hc, int r) // r = 0..8 -> (dy, dz); returns hx-1 cell
{
    return hc + ((r / 3) - 1) * (NB_HX * NB_HY) + ((r % 3) - 1) * NB_HX - 1;
}

// ------------------------------------------------------------------------------------------
// per-pair arithmetic shared by the LDS walk and the global-memory fallback

struct DensityTarget {
    float x, y, z, h2, negInvH2;
};
template <bool COUNT>
__device__ __forceinline__ void density_pair(const DensityTarget &t, const float4 &c, float &acc,
                                             int &cnt)
{
    const float dx = t.x - c.x, dy = t.y - c.y, dz = t.z - c.z;
    if (COUNT) {
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        cnt += (d2 < t.h2) ? 1 : 0;
        const float w = fmaxf(t.h2 - d2, 0.f);
        acc = __builtin_fmaf(w * w, w, acc);
    } else {
        const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        const float w =
            __builtin_amdgcn_fmed3f(__builtin_fmaf(d2, t.negInvH2, 1.0f), 0.f, 1.0f); // w / h2
        acc = __builtin_fmaf(w * w, w, acc);
    }
}

struct ForceTarget {
    float x, y, z, vx, vy, vz;
    float Pi;       // p_i / rho_i^2
    float h, negInvH, thrF;
};
struct ForceAcc {
    float px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
    int cnt = 0;
};
// branch-free pair force with t = (h - r)/h clamped to [0,1]:
//   pressure  += m_j (P_i + q_j) t^2 / r * (dx,dy,dz)      (scaled by rho_i*45/(pi h^6) * h^2 later)
//   viscosity += m_j t (v_j - v_i)                          (scaled by mu_i/rho_i*45/(pi h^6) * h)
// self (d2 == 0): rsq = inf is capped so that a*0 = 0.
template <bool COUNT>
__device__ __forceinline__ void force_pair(const ForceTarget &t, const float4 &cp,
                                           const float4 &cv, bool notSelf, ForceAcc &a)
{
    const float dx = t.x - cp.x, dy = t.y - cp.y, dz = t.z - cp.z;
    float d2;
    if (COUNT) {
        d2 = (dx * dx + dy * dy) + dz * dz;
        a.cnt += (notSelf && d2 < t.thrF) ? 1 : 0;
    } else {
        d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    }
    const float rinv = fminf(__builtin_amdgcn_rsqf(d2), 1.0e18f);
    const float r = d2 * rinv;
    const float tt = __builtin_amdgcn_fmed3f(__builtin_fmaf(r, t.negInvH, 1.0f), 0.f, 1.0f);
    const float mt = cp.w * tt;                        // m_j t
    const float ap = mt * tt * (t.Pi + cv.w) * rinv;   // m_j t^2 (P_i + q_j) / r
    a.px = __builtin_fmaf(ap, dx, a.px);
    a.py = __builtin_fmaf(ap, dy, a.py);
    a.pz = __builtin_fmaf(ap, dz, a.pz);
    a.vx = __builtin_fmaf(mt, cv.x - t.vx, a.vx);
    a.vy = __builtin_fmaf(mt, cv.y - t.vy, a.vy);
    a.vz = __builtin_fmaf(mt, cv.z - t.vz, a.vz);
}

// ------------------------------------------------------------------------------------------
// wave-level pieces of the LDS walk

// (T targets) x (S slices): S = floor(64 / T) without an integer division
__device__ __forceinline__ int slices_for(int T)
{
    return (int)(64.5f * __builtin_amdgcn_rcpf((float)T));
}

// lanes 0..8 fetch the LDS window [rs, re) of the nine (dy,dz) rows of the cell with halo index hc
__device__ __forceinline__ void row_windows(const TileShared &sh, int hc, int lane, int &rs,
                                            int &re)
{
    rs = 0;
    re = 0;
    if (lane < 9) {
        const int h0 = hc + ((lane / 3) - 1) * (NB_HX * NB_HY) + ((lane % 3) - 1) * NB_HX - 1;
        rs = sh.pref[h0];
        re = sh.pref[h0 + 3];
    }
}

// keeps all four components of an LDS record live so that the read is one ds_read_b128
// (4 LDS cycles per wave) and not a ds_read_b96 (8 cycles)
__device__ __forceinline__ void keep_b128(const float4 &c) { asm volatile("" ::"v"(c.w)); }

// ------------------------------------------------------------------------------------------
// density + pressure
template <bool COUNT>
__global__ __launch_bounds__(NB_THREADS) void k_density(
    const float4 *__restrict__ posm, float4 *__restrict__ velq, const uint32_t *__restrict__ mat,
    const int2 *__restrict__ table, const int2 *__restrict__ ttable,
    const float4 *__restrict__ bposm, const int2 *__restrict__ btable, GridDesc g,
    const Mat *__restrict__ mats, float *__restrict__ rho, float *__restrict__ prs,
    int *__restrict__ counts, int cap)
{
    __shared__ TileShared sh;
    __shared__ float4 recP[NB_DCAP + 1]; // + the FAR record that padding lanes read

    const int tile = blockIdx.x;
    const int2 trange = ttable[tile];
    if (trange.y <= trange.x) return; // no fluid particle in this tile
    const int ntile = trange.y - trange.x;
    const int total = tile_setup(sh, tile, g, table, btable);
    const bool inLds = total <= cap && ntile <= NB_TCAP;
    if (inLds) {
        tile_stage<false>(sh, total, recP, nullptr, posm, nullptr, bposm, nullptr);
        // constants of the tile's own particles: -1/h2, scale, k, rho0
        for (int i = threadIdx.x; i < ntile; i += NB_THREADS) {
            const Mat m = mats[mat[trange.x + i]];
            const float scale = COUNT ? m.poly6 : m.poly6 * (m.h2 * m.h2 * m.h2);
            sh.tconst[i] = make_float4(-1.0f / m.h2, scale, m.gasConstant, m.restDensity);
            if (COUNT) sh.tthr[i] = m.h2;
        }
        if (threadIdx.x == 0) recP[NB_DCAP] = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *red = sh.red[w][0];
    int *redc = (int *)sh.red[w][1];
    for (int ci = 0; ci < NB_CELLS_PER_WAVE; ci++) {
        const int hc = halo_center(w * NB_CELLS_PER_WAVE + ci);
        const int ncell = uniform_i(sh.fc[hc]);
        const int cstart = uniform_i(sh.fs[hc]);
        if (ncell == 0) continue;
        if (inLds) {
            const int clds = uniform_i(sh.pref[hc]);
            int rs, re;
            row_windows(sh, hc, lane, rs, re);
            for (int tb = 0; tb < ncell; tb += 64) {
                const int T = min(64, ncell - tb);
                const int S = slices_for(T);
                int u, slice;
                split_lane(lane, S, u, slice);
                const bool active = u < T;
                const int tloc = tb + (active ? u : 0);
                const float4 tp = recP[clds + tloc];
                const float4 tc = sh.tconst[cstart - trange.x + tloc];
                DensityTarget t;
                t.x = tp.x; t.y = tp.y; t.z = tp.z;
                t.negInvH2 = tc.x;
                t.h2 = COUNT ? sh.tthr[cstart - trange.x + tloc] : 0.f;
                float acc = 0.f;
                int cnt = 0;
#pragma unroll
                for (int r = 0; r < 9; r++) {
                    const int s0 = bcast_i(rs, r), e0 = bcast_i(re, r);
                    float4 c[NB_NQ];
#pragma unroll
                    for (int q = 0; q < NB_NQ; q++) {
                        const int k = s0 + q * S + slice;
                        c[q] = recP[k < e0 ? k : NB_DCAP];
                        keep_b128(c[q]);
                    }
#pragma unroll
                    for (int q = 0; q < NB_NQ; q++) density_pair<COUNT>(t, c[q], acc, cnt);
                    for (int k0 = s0 + NB_NQ * S; k0 < e0; k0 += S) { // long windows
                        const int k = k0 + slice;
                        const float4 cd = recP[k < e0 ? k : NB_DCAP];
                        density_pair<COUNT>(t, cd, acc, cnt);
                    }
                }
                // fold the S slices of every target (wave-private strip)
                wave_lds_fence();
                red[lane] = acc;
                if (COUNT) redc[lane] = cnt;
                wave_lds_fence();
                if (active && slice == 0) {
                    float sum = 0.f;
                    int c2 = 0;
                    for (int q = 0; q < S; q++) {
                        sum += red[lane + q];
                        if (COUNT) c2 += redc[lane + q];
                    }
                    // rho_i = m_i*315/(64*pi*h^9) * sum (h2-d2)^3 ; p_i = max(k (rho_i - rho0), 0)
                    const float density = tc.y * sum;
                    const float pressure = fmaxf(tc.z * (density - tc.w), 0.f);
                    const int ti = cstart + tloc;
                    rho[ti] = density;
                    prs[ti] = pressure;
                    velq[ti].w = pressure / (density * density);
                    if (COUNT) counts[ti] = c2;
                }
            }
        } else {
            // global-memory walk (halo does not fit the LDS tile)
            for (int tb = 0; tb < ncell; tb += 64) {
                const int T = min(64, ncell - tb);
                const int S = slices_for(T);
                int u, slice;
                split_lane(lane, S, u, slice);
                const bool active = u < T;
                const int ti = cstart + tb + (active ? u : 0);
                const float4 tp = posm[ti];
                const Mat m = mats[mat[ti]];
                DensityTarget t;
                t.x = tp.x; t.y = tp.y; t.z = tp.z;
                t.h2 = m.h2;
                t.negInvH2 = -1.0f / m.h2;
                float acc = 0.f;
                int cnt = 0;
                for (int r = 0; r < 9; r++) {
                    const int h0 = halo_row(hc, r);
                    for (int c3 = 0; c3 < 3; c3++) {
                        const int h = h0 + c3;
                        for (int pass = 0; pass < 2; pass++) {
                            const float4 *__restrict__ src = pass ? bposm : posm;
                            const int s0 = uniform_i(pass ? sh.bs[h] : sh.fs[h]);
                            const int e0 = s0 + uniform_i(pass ? sh.bc[h] : sh.fc[h]);
                            for (int k0 = s0; k0 < e0; k0 += S) {
                                const int k = k0 + slice;
                                float4 cd = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
                                if (k < e0) cd = src[k];
                                density_pair<COUNT>(t, cd, acc, cnt);
                            }
                        }
                    }
                }
                wave_lds_fence();
                red[lane] = acc;
                if (COUNT) redc[lane] = cnt;
                wave_lds_fence();
                if (active && slice == 0) {
                    float sum = 0.f;
                    int c2 = 0;
                    for (int q = 0; q < S; q++) {
                        sum += red[lane + q];
                        if (COUNT) c2 += redc[lane + q];
                    }
                    const float scale = COUNT ? m.poly6 : m.poly6 * (m.h2 * m.h2 * m.h2);
                    const float density = scale * sum;
                    const float pressure = fmaxf(m.gasConstant * (density - m.restDensity), 0.f);
                    rho[ti] = density;
                    prs[ti] = pressure;
                    velq[ti].w = pressure / (density * density);
                    if (COUNT) counts[ti] = c2;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// force: pressure (spiky) + viscosity + gravity
template <bool COUNT>
__global__ __launch_bounds__(NB_THREADS) void k_force(
    const float4 *__restrict__ posm, const float4 *__restrict__ velq,
    const uint32_t *__restrict__ mat, const float *__restrict__ rho,
    const float *__restrict__ prs, const int2 *__restrict__ table,
    const int2 *__restrict__ ttable, const float4 *__restrict__ bposm,
    const int2 *__restrict__ btable, GridDesc g, const Mat *__restrict__ mats, float gx, float gy,
    float gz, float4 *__restrict__ force, int *__restrict__ counts, int cap,
    const float4 *__restrict__ zeros)
{
    __shared__ TileShared sh;
    __shared__ float4 recP[NB_FCAP + 1]; // x, y, z, m_j   (+ FAR record)
    __shared__ float4 recV[NB_FCAP + 1]; // vx, vy, vz, q_j = p_j/rho_j^2 (boundary: 0,0,0,0)

    const int tile = blockIdx.x;
    const int2 trange = ttable[tile];
    if (trange.y <= trange.x) return;
    const int ntile = trange.y - trange.x;
    const int total = tile_setup(sh, tile, g, table, btable);
    const bool inLds = total <= cap && ntile <= NB_TCAP;
    if (inLds) {
        tile_stage<true>(sh, total, recP, recV, posm, velq, bposm, zeros);
        // constants of the tile's own particles: -1/h, Cp, Cv, rho
        for (int i = threadIdx.x; i < ntile; i += NB_THREADS) {
            const Mat m = mats[mat[trange.x + i]];
            const float d = rho[trange.x + i];
            sh.tconst[i] = make_float4(-1.0f / m.h, d * m.spiky * (m.h * m.h),
                                       m.viscosity / d * m.spiky * m.h, d);
            if (COUNT) sh.tthr[i] = m.thrF;
        }
        if (threadIdx.x == 0) {
            recP[NB_FCAP] = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
            recV[NB_FCAP] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *redx = sh.red[w][0], *redy = sh.red[w][1], *redz = sh.red[w][2];
    int *redc = (int *)sh.red[w][3];
    for (int ci = 0; ci < NB_CELLS_PER_WAVE; ci++) {
        const int hc = halo_center(w * NB_CELLS_PER_WAVE + ci);
        const int ncell = uniform_i(sh.fc[hc]);
        const int cstart = uniform_i(sh.fs[hc]);
        if (ncell == 0) continue;
        const int clds = uniform_i(sh.pref[hc]);
        int rs = 0, re = 0;
        if (inLds) row_windows(sh, hc, lane, rs, re);
        for (int tb = 0; tb < ncell; tb += 64) {
            const int T = min(64, ncell - tb);
            const int S = slices_for(T);
            int u, slice;
            split_lane(lane, S, u, slice);
            const bool active = u < T;
            const int tloc = tb + (active ? u : 0);
            const int ti = cstart + tloc;
            ForceTarget t;
            ForceAcc a;
            float Cp, Cv, trho;
            if (inLds) {
                const int selfK = clds + tloc; // LDS slot of the target itself
                const float4 tp = recP[selfK], tv = recV[selfK];
                const float4 tc = sh.tconst[cstart - trange.x + tloc];
                t.x = tp.x; t.y = tp.y; t.z = tp.z;
                t.vx = tv.x; t.vy = tv.y; t.vz = tv.z;
                t.Pi = tv.w; // q_i = p_i / rho_i^2, written by the density kernel
                t.negInvH = tc.x;
                t.h = 0.f;
                t.thrF = COUNT ? sh.tthr[cstart - trange.x + tloc] : 0.f;
                Cp = tc.y; Cv = tc.z; trho = tc.w;
#pragma unroll
                for (int r = 0; r < 9; r++) {
                    const int s0 = bcast_i(rs, r), e0 = bcast_i(re, r);
                    float4 cp[NB_NQ], cv[NB_NQ];
#pragma unroll
                    for (int q = 0; q < NB_NQ; q++) {
                        const int k = s0 + q * S + slice;
                        const int kk = k < e0 ? k : NB_FCAP;
                        cp[q] = recP[kk];
                        cv[q] = recV[kk];
                    }
#pragma unroll
                    for (int q = 0; q < NB_NQ; q++)
                        force_pair<COUNT>(t, cp[q], cv[q], s0 + q * S + slice != selfK, a);
                    for (int k0 = s0 + NB_NQ * S; k0 < e0; k0 += S) { // long windows
                        const int k = k0 + slice;
                        const int kk = k < e0 ? k : NB_FCAP;
                        force_pair<COUNT>(t, recP[kk], recV[kk], k != selfK, a);
                    }
                }
            } else {
                // global-memory walk (halo does not fit the LDS tile)
                const float4 tp = posm[ti], tv = velq[ti];
                const Mat m = mats[mat[ti]];
                trho = rho[ti];
                t.x = tp.x; t.y = tp.y; t.z = tp.z;
                t.vx = tv.x; t.vy = tv.y; t.vz = tv.z;
                t.Pi = prs[ti] / (trho * trho);
                t.h = m.h;
                t.negInvH = -1.0f / m.h;
                t.thrF = m.thrF;
                Cp = trho * m.spiky * (m.h * m.h);       // rho_i * 45/(pi h^6) * h^2
                Cv = m.viscosity / trho * m.spiky * m.h; // mu_i/rho_i * 45/(pi h^6) * h
                for (int r = 0; r < 9; r++) {
                    const int h0 = halo_row(hc, r);
                    for (int c3 = 0; c3 < 3; c3++) {
                        const int h = h0 + c3;
                        { // fluid neighbours (index != j, :152)
                            const int s0 = uniform_i(sh.fs[h]), e0 = s0 + uniform_i(sh.fc[h]);
                            for (int k0 = s0; k0 < e0; k0 += S) {
                                const int k = k0 + slice;
                                float4 cp = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
                                float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                                if (k < e0) {
                                    cp = posm[k];
                                    cv = velq[k];
                                }
                                force_pair<COUNT>(t, cp, cv, k != ti, a);
                            }
                        }
                        { // boundary neighbours: rho_j = rho0_j, p_j = 0, v_j = 0 (:184-189)
                            const int s0 = uniform_i(sh.bs[h]), e0 = s0 + uniform_i(sh.bc[h]);
                            for (int k0 = s0; k0 < e0; k0 += S) {
                                const int k = k0 + slice;
                                float4 cp = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
                                if (k < e0) cp = bposm[k];
                                force_pair<COUNT>(t, cp, make_float4(0.f, 0.f, 0.f, 0.f), true, a);
                            }
                        }
                    }
                }
            }

            // scale, fold the S slices of every target, add gravity
            wave_lds_fence();
            redx[lane] = Cp * a.px + Cv * a.vx;
            redy[lane] = Cp * a.py + Cv * a.vy;
            redz[lane] = Cp * a.pz + Cv * a.vz;
            if (COUNT) redc[lane] = a.cnt;
            wave_lds_fence();
            if (active && slice == 0) {
                float fx = 0.f, fy = 0.f, fz = 0.f;
                int c2 = 0;
                for (int q = 0; q < S; q++) {
                    fx += redx[lane + q];
                    fy += redy[lane + q];
                    fz += redz[lane + q];
                    if (COUNT) c2 += redc[lane + q];
                }
                // forces = forcePressure + forceViscosity + gravity * rho_i  (:140,:202)
                force[ti] = make_float4(fx + gx * trho, fy + gy * trho, fz + gz * trho, 0.f);
                if (COUNT) counts[ti] = c2;
            }
        }
    }
}
