// sph_neighbors.h -- the two 27-cell neighbour loops: density/pressure and force.
//
// Reference semantics: ComputeDensityAndPressureJob (FluidSPHByCPUSystem.cs:48-104) and
// ComputeForceJob (:126-236); the device counterparts they replace are the HLSL kernels
// ComputeDensityAndPressure / ComputeForce (FluidSolver.compute:128-272), which walk the table
// per thread and gather every neighbour from global memory.
//
// Mapping (DESIGN.md "neighbour kernels").  One 256-thread workgroup owns one TILE of 4x4x2
// cells (the sort key groups particles by tile, sph_device.h) and works through it in BLOCKS:
// the whole tile when its halo fits the LDS tile, else its eight 2x2x1 sub-blocks, else single
// cells (the fluid compresses several-fold at the bottom of a deep tank), else -- for a cell whose
// 27-cell neighbourhood still overflows -- the same walk straight from global memory.  Per block:
//   1. one thread per cell of the block's halo (block + one cell on every side) turns the table
//      entries fetched at kernel start into LDS offsets with a block scan;
//   2. the halo's particles are copied into the LDS tile by LDS-DMA (global_load_lds_dwordx4: no
//      VGPR staging, 64 records per wave-instruction), fluid then boundary per cell, x-runs
//      contiguous; a particle is read from L2/HBM ~4.5 times per pass instead of 27 times; the
//      per-target constants of the block's own particles are cached in LDS at the same time, so
//      the compute phase touches no global memory until its stores;
//   3. the block's cells are dealt to the four waves.  For a cell with T particles (the targets)
//      the 64 lanes split as T targets x S = 64/T candidate slices; a lane keeps its target in
//      VGPRs (every inner-loop operand is a VGPR: VALU ops with an SGPR or constant operand issue
//      at about half rate on gfx950, tools/microbench_valu.hip) and walks the 9 LDS row windows
//      (x-1..x+1 of one (y,z)) with stride S, four ds_read_b128 issued ahead of their use;
//   4. per-target partial sums are folded over the S slices through a wave-private LDS strip.
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
#ifndef NB_DTHREADS
#define NB_DTHREADS 256 // workgroup size of the density kernel: 4 workgroups (16 waves) per CU
#endif
#ifndef NB_FTHREADS
#define NB_FTHREADS 512 // workgroup size of the force kernel: 3 workgroups (24 waves) per CU
#endif
#define NB_MAXWAVES 8
#ifndef NB_DG
#define NB_DG 4 // slots per loop iteration (records in flight per lane), density
#endif
#ifndef NB_FG
#define NB_FG 2 // same, force
#endif
#ifndef NB_DCAP
#define NB_DCAP 1440 // LDS records per block, density kernel (23 KB): 40 KB per workgroup in all
#endif
#ifndef NB_FCAP
#define NB_FCAP 1184 // LDS records per block, force kernel (2 x 18.5 KB): 53 KB per workgroup in all
#endif
#define NB_TCAP 384      // targets per block (per-target loop constant and material index in LDS)
#ifndef NB_CHUNK
#define NB_CHUNK 8 // tiles per work-queue request
#endif
#define NB_MAXUNITS 64   // (cell, part) units per block
#define NB_SLOTCAP 1408  // slot-table entries per block
#ifndef NB_DLB
#define NB_DLB 4 // waves per SIMD the density kernel's register allocation must allow
#endif
#ifndef NB_FLB
#define NB_FLB 6 // same, force
#endif

// Per block, in LDS.  A UNIT is (cell, part): up to 32 targets of one cell, T of them x S = 64/T
// candidate slices over the lanes of one wave.  A SLOT is one step of that wave through the
// candidates: lane (target u, slice s) reads the record at  slot entry + 16 s.  The nine (dy,dz)
// row windows of a cell (cells x-1..x+1, contiguous in the image) are walked one after the other.
// In a window's last slot the slices past its end read the records that FOLLOW the window: when
// those are still records of the same image row (cell x+2 and beyond: farther than one cell from
// every target, so their weight clamps to exactly 0 as long as no support radius exceeds the cell
// size) the slot is an ordinary one (list A); otherwise the slot goes to list B, whose entries
// carry the number of valid slices and whose other lanes read the FAR record.
struct UnitDesc {
    int tS;      // targets | slices << 8
    int nAB;     // slots of list A | slots of list B << 16
    int gi0;     // global (sorted) index of the first target
    int self;    // LDS record index of the first target
    int slotA;   // slot-table index of list A
    int slotB;   // slot-table index of list B
    int tbase;   // ordinal of the first target in the block (index into tk / tmat)
    int pad;
};
struct TileShared {
    int fs[NB_HCELLS];        // tile halo: fluid start (global sorted index) of the cell
    int fc[NB_HCELLS];        //            fluid count
    int bs[NB_HCELLS];        //            boundary start
    int bc[NB_HCELLS];        //            boundary count
    int hmap[NB_HCELLS];      // block halo cell -> tile halo cell
    int spref[NB_HCELLS + 1]; // LDS record index of the block halo cell's records (fluid, then boundary)
    int4 cinfo[SPH_TCELLS];   // per block cell: targets, ordinal of first target, first unit, first slot
    int4 cpart[SPH_TCELLS];   //                 parts, targets per part, slices, slots reserved
    int2 cellf[SPH_TCELLS];   // per block cell: halo index in the tile, halo index in the block
    UnitDesc unit[NB_MAXUNITS];
    __attribute__((aligned(16))) uint32_t slottab[NB_SLOTCAP];
    int scan[4 * NB_MAXWAVES];
    float tk[NB_TCAP];                // per target: -1/h2 (density) or -1/h (force)
    unsigned short tmat[NB_TCAP];     // per target: material index
    unsigned char tcell[NB_TCAP];                               // target -> cell of the block
};

// kernel arguments shared by both passes
struct NbArgs {
    const float4 *posm;
    float4 *velq;
    const uint32_t *mat;
    const float *rhoIn;
    const float *prsIn;
    const int2 *table, *ttable, *btable;
    const float4 *bposm;
    const Mat *mats;
    const float4 *zeros;
    float *rho, *prs;
    float4 *force;
    int *counts;
    GridDesc g;
    float gx, gy, gz;
    int cap;
    int overread; // 1: every support radius <= cell size, window tails may read on (list A)
    int nmats;
    const uint32_t *tileList; // non-empty tiles (k_map_table)
    uint32_t *tileCtl;        // [0] number of tiles in the list, [queue] next list entry to take
    int queue;
};

__device__ __forceinline__ int wrapi(int c, int g) { return c < 0 ? c + g : (c >= g ? c - g : c); }

// lane -> (target slot u, slice) for T targets and S = 64 / T slices
__device__ __forceinline__ void split_lane(int lane, int S, int &u, int &slice)
{
    u = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)S)); // exact for lane < 64
    slice = lane - u * S;
}
// S = floor(64 / T) without an integer division
__device__ __forceinline__ int slices_for(int T)
{
    return (int)(64.5f * __builtin_amdgcn_rcpf((float)T));
}

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in order; this only stops the compiler from moving
    // them across the point where other lanes' data is consumed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive scan inside each row of 16 lanes (DPP row_shr, zeros shifted in)
__device__ __forceinline__ int row16_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true); // row_shr:8
    return v;
}

// Sum over the S consecutive lanes of a slice group, valid in the group's first lane (slice 0).
// Groups of 2/4/8/16 lanes never straddle a DPP row of 16, larger or odd groups go through
// ds_bpermute.  The order of the additions depends only on S.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int N> __device__ __forceinline__ void fold_slices(float (&v)[N], int &c, int S, int slice)
{
    if ((S & (S - 1)) == 0 && S <= 16) { // row_shl:d, zeros shifted in
        if (S >= 2) { for (int k = 0; k < N; k++) v[k] += dpp_f<0x101>(v[k]); c += dpp_i<0x101>(c); }
        if (S >= 4) { for (int k = 0; k < N; k++) v[k] += dpp_f<0x102>(v[k]); c += dpp_i<0x102>(c); }
        if (S >= 8) { for (int k = 0; k < N; k++) v[k] += dpp_f<0x104>(v[k]); c += dpp_i<0x104>(c); }
        if (S >= 16) { for (int k = 0; k < N; k++) v[k] += dpp_f<0x108>(v[k]); c += dpp_i<0x108>(c); }
    } else {
        for (int d = 1; d < S; d <<= 1) {
            const bool in = slice + d < S;
            float x[N];
            for (int k = 0; k < N; k++) x[k] = __shfl_down(v[k], d, 64);
            const int xc = __shfl_down(c, d, 64);
            for (int k = 0; k < N; k++) v[k] += in ? x[k] : 0.f;
            c += in ? xc : 0;
        }
    }
}

// exclusive scan of four values per thread over the workgroup; t = the four totals
__device__ __forceinline__ void block_scan4(TileShared &sh, const int (&v)[4], int (&e)[4], int (&t)[4])
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int i4[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = __shfl_up(i4[k], d, 64);
            if (lane >= d) i4[k] += x;
        }
    }
    __syncthreads(); // previous users of sh.scan are done
    if (lane == 63)
#pragma unroll
        for (int k = 0; k < 4; k++) sh.scan[4 * w + k] = i4[k];
    __syncthreads();
    int base[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = 0;
#pragma unroll
    for (int i = 0; i < NB_MAXWAVES; i++) {
        if (i >= nw) break;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = sh.scan[4 * i + k];
            if (i < w) base[k] += x;
            t[k] += x;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) e[k] = base[k] + i4[k] - v[k];
}

// table look-ups for the 6x6x4 halo of the tile (once per workgroup)
__device__ __forceinline__ void tile_lookup(TileShared &sh, int tile, const NbArgs &A)
{
    const GridDesc &g = A.g;
    const int tid = threadIdx.x;
    if (tid < NB_HCELLS) {
        const int tx = tile % g.ntx, t2 = tile / g.ntx, ty = t2 % g.nty, tz = t2 / g.nty;
        const int hx = tid % NB_HX, hy = (tid / NB_HX) % NB_HY, hz = tid / (NB_HX * NB_HY);
        const int x = wrapi(tx * SPH_TX + hx - 1, g.gx), y = wrapi(ty * SPH_TY + hy - 1, g.gy),
                  z = wrapi(tz * SPH_TZ + hz - 1, g.gz);
        const uint32_t k = tiled_key(x, y, z, g);
        const int2 f = A.table[k];
        int2 b = make_int2(0, 0);
        if (A.btable) b = A.btable[k];
        sh.fs[tid] = f.x;
        sh.fc[tid] = f.y - f.x;
        sh.bs[tid] = b.x;
        sh.bc[tid] = b.y - b.x;
    }
    __syncthreads();
}

__device__ __forceinline__ int tile_halo_index(int x, int y, int z) // tile-local cell + 1
{
    return (z * NB_HY + y) * NB_HX + x;
}

// LDS-DMA: global_load_lds_dwordx4 writes 16 bytes per lane to  M0 base + lane*16  with no VGPR
// in between, so one wave-instruction moves 64 consecutive records (1 KiB) of the tile image.
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// keeps all four components of an LDS record live so that the read is one ds_read_b128
// (4 LDS cycles per wave) and not a ds_read_b96 (8 cycles)
__device__ __forceinline__ void keep_b128(const float4 &c) { asm volatile("" ::"v"(c.w)); }
__device__ __forceinline__ void keep_b128x(const float4 &c) { asm volatile("" ::"v"(c.x), "v"(c.w)); }

// ------------------------------------------------------------------------------------------
// per-pair arithmetic

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
    float Pi; // p_i / rho_i^2
    float negInvH, thrF;
    float tiny; // 1e-30 in a VGPR (a literal operand would slow the instruction down)
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
    float d2, rinv;
    if (COUNT) {
        d2 = (dx * dx + dy * dy) + dz * dz;
        a.cnt += (notSelf && d2 < t.thrF) ? 1 : 0;
        rinv = fminf(__builtin_amdgcn_rsqf(d2), 1.0e18f);
    } else {
        // tiny = 1e-30 keeps rsq finite for the pair (i, i) without a min: r = 1e-15, t = 1 and
        // both terms are multiplied by d = 0 / v_j - v_i = 0
        d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, t.tiny)));
        rinv = __builtin_amdgcn_rsqf(d2);
    }
    const float r = d2 * rinv;
    const float tt = __builtin_amdgcn_fmed3f(__builtin_fmaf(r, t.negInvH, 1.0f), 0.f, 1.0f);
    const float mt = cp.w * tt;                      // m_j t
    const float ap = mt * tt * (t.Pi + cv.w) * rinv; // m_j t^2 (P_i + q_j) / r
    a.px = __builtin_fmaf(ap, dx, a.px);
    a.py = __builtin_fmaf(ap, dy, a.py);
    a.pz = __builtin_fmaf(ap, dz, a.pz);
    a.vx = __builtin_fmaf(mt, cv.x - t.vx, a.vx);
    a.vy = __builtin_fmaf(mt, cv.y - t.vy, a.vy);
    a.vz = __builtin_fmaf(mt, cv.z - t.vz, a.vz);
}

// per-target accumulators / finalisation, one type per pass -------------------------------------
// Finishing constants of a target (fc4), from its material (and, for the force pass, its density):
//   density: -, m_i*315/(64 pi h^9) [* h2^3 in the production variant], k, rho0
//   force:   -, rho_i*45/(pi h^6)*h^2, mu_i/rho_i*45/(pi h^6)*h, rho_i
template <bool FORCE, bool COUNT> struct Pass;

template <bool COUNT> struct Pass<false, COUNT> {
    DensityTarget t;
    float acc = 0.f;
    int cnt = 0;
    __device__ __forceinline__ void pair(const float4 &cp, const float4 &, bool)
    {
        density_pair<COUNT>(t, cp, acc, cnt);
    }
    // folds the slices and writes rho, p, q for target ti
    __device__ __forceinline__ void finish(const NbArgs &A, int S, int slice, bool writer, int ti,
                                           const float4 &fc4)
    {
        float v[1] = {acc};
        int c2 = cnt;
        fold_slices<1>(v, c2, S, slice);
        if (writer) {
            const float sum = v[0];
            // rho_i = m_i*315/(64*pi*h^9) * sum (h2-d2)^3 ; p_i = max(k (rho_i - rho0), 0)
            const float density = fc4.y * sum;
            const float pressure = fmaxf(fc4.z * (density - fc4.w), 0.f);
            A.rho[ti] = density;
            A.prs[ti] = pressure;
            // q = p / rho^2 through v_rcp: 1 ulp, it only enters the force sums
            A.velq[ti].w = pressure * __builtin_amdgcn_rcpf(density * density);
            if (COUNT) A.counts[ti] = c2;
        }
    }
};

template <bool COUNT> struct Pass<true, COUNT> {
    ForceTarget t;
    ForceAcc a;
    __device__ __forceinline__ void pair(const float4 &cp, const float4 &cv, bool notSelf)
    {
        force_pair<COUNT>(t, cp, cv, notSelf, a);
    }
    __device__ __forceinline__ void finish(const NbArgs &A, int S, int slice, bool writer, int ti,
                                           const float4 &fc4)
    {
        float v[3] = {fc4.y * a.px + fc4.z * a.vx, fc4.y * a.py + fc4.z * a.vy, fc4.y * a.pz + fc4.z * a.vz};
        int c2 = a.cnt;
        fold_slices<3>(v, c2, S, slice);
        if (writer) {
            const float fx = v[0], fy = v[1], fz = v[2];
            // forces = forcePressure + forceViscosity + gravity * rho_i  (:140,:202)
            const float trho = fc4.w;
            A.force[ti] =
                make_float4(fx + A.gx * trho, fy + A.gy * trho, fz + A.gz * trho, 0.f);
            if (COUNT) A.counts[ti] = c2;
        }
    }
};

// loop constant (-1/h2 or -1/h) and exact-predicate threshold (h2 or thrF) of a material
template <bool FORCE> __device__ __forceinline__ void loop_constants(const Mat &m, float &k, float &thr)
{
    k = FORCE ? m.negInvH : m.negInvH2;
    thr = FORCE ? m.thrF : m.h2;
}
// finishing constants of target gi with material index mi (see Pass)
template <bool FORCE, bool COUNT>
__device__ __forceinline__ float4 finish_constants(const NbArgs &A, int gi, uint32_t mi)
{
    const Mat *m = A.mats + mi;
    if (FORCE) {
        const float d = A.rhoIn[gi];
        const float h = m->h, spiky = m->spiky, mu = m->viscosity;
        return make_float4(0.f, d * spiky * (h * h), mu / d * spiky * h, d);
    }
    return make_float4(0.f, COUNT ? m->poly6 : m->poly6h6, m->gasConstant, m->restDensity);
}

// ------------------------------------------------------------------------------------------
#ifdef NB_STAMPS // experimental builds: cycles per phase, summed over workgroups (thread 0)
static __device__ unsigned long long g_stamps[16];
#define NB_STAMP(i)                                                                              \
    do {                                                                                         \
        if (threadIdx.x == 0) {                                                                  \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                        \
            atomicAdd(&g_stamps[(FORCE ? 8 : 0) + (i)], now_ - *stamp);                          \
            *stamp = now_;                                                                       \
        }                                                                                        \
    } while (0)
#else
#define NB_STAMP(i)
#endif

// floor(a / b) for 0 <= a < 2^20, 1 <= b <= 64 without an integer division: (a + 0.5) / b is at
// least 0.5/64 away from every integer, far more than the error of v_rcp
__device__ __forceinline__ int div_small(int a, int b)
{
    return (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)b));
}

// one block (bx0.., sx x sy x sz cells, tile-local) through the LDS tile.  Returns false, having
// done nothing, when the block's halo, targets, units or slots do not fit.
template <bool FORCE, bool COUNT, int sx, int sy, int sz>
__device__ __forceinline__ bool process_block(TileShared &sh, float4 *recP, float4 *recV, float *tthr,
                                              unsigned char *rcell, const NbArgs &A, int bx0,
                                              int by0, int bz0, unsigned long long *stamp = nullptr)
{
    constexpr int NB_THREADS = FORCE ? NB_FTHREADS : NB_DTHREADS, NB_WAVES = NB_THREADS / 64;
    constexpr int hxN = sx + 2, hyN = sy + 2, nh = hxN * hyN * (sz + 2), ncells = sx * sy * sz;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int cap = A.cap;

    // 1. sizes: records per halo cell; targets, parts, slices and a bound on the slots per cell
    int v4[4] = {0, 0, 0, 0}; // records, targets, units, slots
    int P = 0, per = 0, S = 0;
    if (tid < nh) {
        const int hx = tid % hxN, hy = (tid / hxN) % hyN, hz = tid / (hxN * hyN);
        const int f = tile_halo_index(bx0 + hx, by0 + hy, bz0 + hz);
        sh.hmap[tid] = f;
        v4[0] = sh.fc[f] + sh.bc[f];
    }
    if (tid < ncells) {
        const int cx = tid % sx, cy = (tid / sx) % sy, cz = tid / (sx * sy);
        const int f = tile_halo_index(bx0 + cx + 1, by0 + cy + 1, bz0 + cz + 1);
        const int tcnt = sh.fc[f];
        sh.cellf[tid] = make_int2(f, ((cz + 1) * hyN + (cy + 1)) * hxN + (cx + 1));
        v4[1] = tcnt;
        if (tcnt > 0) {
            // at most 32 targets per unit (S >= 2), and at least as many units as waves
            P = max((tcnt + 31) >> 5, min(tcnt, (NB_WAVES + ncells - 1) / ncells));
            if (P == 1) {
                per = tcnt;
            } else {
                per = div_small(tcnt, P);
                while (per * P < tcnt) per++;
                P = div_small(tcnt, per);
                while (P * per < tcnt) P++;
            }
            S = slices_for(per);
            int V = 0; // candidates: the records of the 27 cells
#pragma unroll
            for (int k = 0; k < 27; k++) {
                const int f2 = f + ((k / 9) - 1) * (NB_HX * NB_HY) + (((k / 3) % 3) - 1) * NB_HX + (k % 3) - 1;
                V += sh.fc[f2] + sh.bc[f2];
            }
            // sum over the 9 windows of ceil(len / S) <= (V + 9 (S - 1)) / S; a multiple of 4
            v4[3] = (div_small(V + 9 * (S - 1), S) + 4) & ~3;
            v4[2] = P;
        }
    }
    int e4[4], t4[4];
    NB_STAMP(1); // sizes
    block_scan4(sh, v4, e4, t4);
    NB_STAMP(2); // scan
    const int total = t4[0], ttotal = t4[1], nunits = t4[2];
    if (total > cap || ttotal > NB_TCAP || nunits > NB_MAXUNITS || t4[3] > NB_SLOTCAP) return false;
    if (ttotal == 0) return true;
    // besides the prefix sums, every record and every target gets the index of its cell (a byte
    // each): the staging loops below then look it up instead of bisecting the prefix sums
    if (tid < nh) {
        sh.spref[tid] = e4[0];
#pragma nounroll
        for (int k = e4[0]; k < e4[0] + v4[0]; k++) rcell[k] = (unsigned char)tid;
    }
    if (tid == nh) sh.spref[nh] = total;
    if (tid < ncells) {
        sh.cinfo[tid] = make_int4(v4[1], e4[1], e4[2], e4[3]);
        sh.cpart[tid] = make_int4(P, per, S, v4[3]);
#pragma nounroll
        for (int k = e4[1]; k < e4[1] + v4[1]; k++) sh.tcell[k] = (unsigned char)tid;
    }
    __syncthreads();
    NB_STAMP(3); // offsets and cell maps

    // 2. stage records (LDS-DMA), per-target loop constants, slot tables and unit descriptors
    for (int q0 = 0; q0 < total; q0 += NB_THREADS) {
        const int qw = q0 + w * 64; // first record of this wave-instruction (wave-uniform)
        const int q = qw + lane;
        if (q < total) {
            const int hcell = rcell[q];
            const int f = sh.hmap[hcell];
            const int k = q - sh.spref[hcell], fc = sh.fc[f];
            const float4 *gp, *gv;
            if (k < fc) {
                gp = A.posm + (sh.fs[f] + k);
                gv = A.velq + (sh.fs[f] + k);
            } else {
                gp = A.bposm + (sh.bs[f] + (k - fc));
                gv = A.zeros; // boundary: v = 0, q = 0
            }
            __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(recP + qw), 16, 0, 0);
            if (FORCE) __builtin_amdgcn_global_load_lds((gptr_t)gv, (lptr_t)(recV + qw), 16, 0, 0);
        }
    }
    for (int i = tid; i < ttotal; i += NB_THREADS) {
        const int c = sh.tcell[i];
        const int gi = sh.fs[sh.cellf[c].x] + (i - sh.cinfo[c].y);
        const uint32_t mi = A.mat[gi];
        float k, thr;
        loop_constants<FORCE>(A.mats[mi], k, thr);
        sh.tk[i] = k;
        sh.tmat[i] = (unsigned short)mi;
        if (COUNT) tthr[i] = thr;
    }
#ifdef NB_SKIP_TABLES
    if (A.cap < 0)
#endif
    if (tid < ncells && v4[1] > 0) {
        // the cell's slot lists: A from the front of its reserved range, B from the back
        const int2 cf = sh.cellf[tid];
        const int cx = tid % sx;
        const int sbase = e4[3], send = e4[3] + v4[3];
        int nA = 0, nB = 0;
#pragma nounroll
        for (int wi = 0; wi < 9; wi++) {
            const int h0 = cf.y + ((wi / 3) - 1) * (hxN * hyN) + ((wi % 3) - 1) * hxN - 1;
            const int ws = sh.spref[h0], we = sh.spref[h0 + 3];
            const int rowEnd = sh.spref[h0 - cx + hxN];
            const int len = we - ws;
            const int nfull = div_small(len, S), rem = len - nfull * S;
#pragma nounroll
            for (int j = 0; j < nfull; j++) sh.slottab[sbase + nA++] = (uint32_t)((ws + j * S) * 16);
            if (rem) {
                const int tail = ws + nfull * S;
                if (A.overread && tail + S <= rowEnd) sh.slottab[sbase + nA++] = (uint32_t)(tail * 16);
                else sh.slottab[send - 1 - nB++] = (uint32_t)(tail * 16) | ((uint32_t)rem << 16);
            }
        }
        const int cstart = sh.fs[cf.x], clds = sh.spref[cf.y];
#pragma nounroll
        for (int part = 0; part < P; part++) {
            const int t0 = part * per;
            UnitDesc u;
            u.tS = min(per, v4[1] - t0) | (S << 8);
            u.nAB = nA | (nB << 16);
            u.gi0 = cstart + t0;
            u.self = clds + t0;
            u.slotA = sbase;
            u.slotB = send - nB;
            u.tbase = e4[1] + t0;
            u.pad = 0;
            sh.unit[e4[2] + part] = u;
        }
    }
    if (tid == 0) {
        recP[cap] = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f); // what masked lanes read
        if (FORCE) recV[cap] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    NB_STAMP(4); // thread 0's own staging work issued
    __syncthreads();
    NB_STAMP(5); // waiting for the staging of the others (DMA, tables)

    // 3. the units, dealt to the waves
    const char *baseP = (const char *)recP, *baseV = (const char *)recV;
    const char *slots = (const char *)sh.slottab;
    const int far16 = cap * 16;
#ifdef NB_SKIP_UNITS // experiment: staging cost alone
    if (A.cap >= 0) { __syncthreads(); return true; }
#endif
    for (int unit = w; unit < nunits; unit += NB_WAVES) {
        const UnitDesc ud = sh.unit[unit];
        const int tS = uniform_i(ud.tS), nAB = uniform_i(ud.nAB);
        const int T = tS & 0xff, S2 = tS >> 8, nA = nAB & 0xffff, nB = nAB >> 16;
        int u, slice;
        split_lane(lane, S2, u, slice);
        const int slice16 = slice * 16;
        const bool active = u < T;
        const int tloc = active ? u : 0;
        const int self16 = (ud.self + tloc) * 16; // LDS byte offset of the target itself
        const int ti = ud.gi0 + tloc;
        const bool writer = active && slice == 0;
        Pass<FORCE, COUNT> ps;
        // the target's finishing constants come from global memory (all slices of a target read
        // the same words): loaded now, used after the walk
        const float4 fc4 = finish_constants<FORCE, COUNT>(A, ti, sh.tmat[ud.tbase + tloc]);
        {
            const float4 tp = *(const float4 *)(baseP + self16);
            ps.t.x = tp.x; ps.t.y = tp.y; ps.t.z = tp.z;
        }
        const float tkv = sh.tk[ud.tbase + tloc];
        if constexpr (FORCE) {
            const float4 tv = *(const float4 *)(baseV + self16);
            ps.t.vx = tv.x; ps.t.vy = tv.y; ps.t.vz = tv.z;
            ps.t.Pi = tv.w; // q_i = p_i / rho_i^2, written by the density kernel
            ps.t.negInvH = tkv;
            ps.t.tiny = 1.0e-30f;
            ps.t.thrF = COUNT ? tthr[ud.tbase + tloc] : 0.f;
        } else {
            ps.t.negInvH2 = tkv;
            ps.t.h2 = COUNT ? tthr[ud.tbase + tloc] : 0.f;
        }
        constexpr int G = FORCE ? NB_FG : NB_DG; // slots per iteration
        // list A: lane reads entry + 16 * slice, nothing else per slot
        const char *sp = slots + ud.slotA * 4;
        int k = 0;
        for (; k + G <= nA; k += G, sp += 4 * G) {
            uint32_t e[G];
            if constexpr (G == 4) {
                const uint4 q = *(const uint4 *)sp;
                e[0] = q.x; e[1] = q.y; e[2] = q.z; e[3] = q.w;
            } else if constexpr (G == 2) {
                const uint2 q = *(const uint2 *)sp;
                e[0] = q.x; e[1] = q.y;
            } else {
#pragma unroll
                for (int q = 0; q < G; q++) e[q] = ((const uint32_t *)sp)[q];
            }
            float4 cp[G], cv[G];
            int off[G];
#pragma unroll
            for (int q = 0; q < G; q++) {
                off[q] = (int)e[q] + slice16;
                cp[q] = *(const float4 *)(baseP + off[q]);
                if (FORCE) cv[q] = *(const float4 *)(baseV + off[q]);
            }
#pragma unroll
            for (int q = 0; q < G; q++) {
                keep_b128(cp[q]);
                if (FORCE) keep_b128x(cv[q]);
                ps.pair(cp[q], cv[q], off[q] != self16);
            }
        }
        for (; k < nA; k++, sp += 4) {
            const int off = (int)*(const uint32_t *)sp + slice16;
            const float4 cp = *(const float4 *)(baseP + off);
            float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FORCE) cv = *(const float4 *)(baseV + off);
            keep_b128(cp);
            if (FORCE) keep_b128x(cv);
            ps.pair(cp, cv, off != self16);
        }
        // list B: slices at or past the entry's count read the FAR record
        sp = slots + ud.slotB * 4;
        for (int kb = 0; kb < nB; kb++, sp += 4) {
            const uint32_t e = *(const uint32_t *)sp;
            const int off = slice < (int)(e >> 16) ? (int)(e & 0xffffu) + slice16 : far16;
            const float4 cp = *(const float4 *)(baseP + off);
            float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FORCE) cv = *(const float4 *)(baseV + off);
            keep_b128(cp);
            if (FORCE) keep_b128x(cv);
            ps.pair(cp, cv, off != self16);
        }
        ps.finish(A, S2, slice, writer, ti, fc4);
    }
    NB_STAMP(6); // thread 0's wave: its units
    __syncthreads(); // the LDS tile is reused by the next block
    NB_STAMP(7); // waiting for the other waves' units
    return true;
}

// one cell straight from global memory (its 27-cell neighbourhood overflows the LDS tile)
template <bool FORCE, bool COUNT>
__device__ void process_cell_global(TileShared &sh, const NbArgs &A, int lx, int ly, int lz)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int f = tile_halo_index(lx + 1, ly + 1, lz + 1);
    const int ncell = uniform_i(sh.fc[f]);
    const int cstart = uniform_i(sh.fs[f]);
    constexpr int NB_WAVES = (FORCE ? NB_FTHREADS : NB_DTHREADS) / 64;
    const int per = (ncell + NB_WAVES - 1) / NB_WAVES; // the waves share the targets
    const int t0 = w * per, t1 = min(ncell, t0 + per);
    for (int tb = t0; tb < t1; tb += 64) {
        const int T = min(64, t1 - tb);
        const int S = slices_for(T);
        int u, slice;
        split_lane(lane, S, u, slice);
        const bool active = u < T;
        const int ti = cstart + tb + (active ? u : 0);
        Pass<FORCE, COUNT> ps;
        float tkv, thr;
        const uint32_t mi = A.mat[ti];
        loop_constants<FORCE>(A.mats[mi], tkv, thr);
        const float4 fc4 = finish_constants<FORCE, COUNT>(A, ti, mi);
        const float4 tp = A.posm[ti];
        ps.t.x = tp.x; ps.t.y = tp.y; ps.t.z = tp.z;
        if constexpr (FORCE) {
            const float4 tv = A.velq[ti];
            ps.t.vx = tv.x; ps.t.vy = tv.y; ps.t.vz = tv.z;
            ps.t.Pi = tv.w;
            ps.t.negInvH = tkv;
            ps.t.tiny = 1.0e-30f;
            ps.t.thrF = thr;
        } else {
            ps.t.negInvH2 = tkv;
            ps.t.h2 = thr;
        }
        for (int c27 = 0; c27 < 27; c27++) {
            const int h = f + ((c27 / 9) - 1) * (NB_HX * NB_HY) + (((c27 / 3) % 3) - 1) * NB_HX +
                          (c27 % 3) - 1;
            { // fluid neighbours (index != j, :152)
                const int s0 = uniform_i(sh.fs[h]), e0 = s0 + uniform_i(sh.fc[h]);
                for (int k0 = s0; k0 < e0; k0 += S) {
                    const int k = k0 + slice;
                    float4 cp = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
                    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k < e0) {
                        cp = A.posm[k];
                        if (FORCE) cv = A.velq[k];
                    }
                    ps.pair(cp, cv, k != ti);
                }
            }
            { // boundary neighbours: rho_j = rho0_j, p_j = 0, v_j = 0 (:184-189)
                const int s0 = uniform_i(sh.bs[h]), e0 = s0 + uniform_i(sh.bc[h]);
                for (int k0 = s0; k0 < e0; k0 += S) {
                    const int k = k0 + slice;
                    float4 cp = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
                    if (k < e0) cp = A.bposm[k];
                    ps.pair(cp, make_float4(0.f, 0.f, 0.f, 0.f), true);
                }
            }
        }
        ps.finish(A, S, slice, active && slice == 0, ti, fc4);
    }
}

// ------------------------------------------------------------------------------------------
template <bool FORCE, bool COUNT>
__device__ __forceinline__ void process_tile(TileShared &sh, float4 *recP, float4 *recV, float *tthr,
                                             unsigned char *rcell, const NbArgs &A, int tile)
{
    unsigned long long stamp_ = 0, *stamp = &stamp_;
    (void)stamp;
#ifdef NB_STAMPS
    stamp_ = __builtin_amdgcn_s_memtime();
#endif
    tile_lookup(sh, tile, A);
    NB_STAMP(0); // table look-ups

    // The largest block whose 27-cell halo fits the LDS tile: the whole tile, then halves
    // (4x4x1, 4x2x1, 2x2x1) down to single cells, then straight from global memory.  The 32
    // cells are numbered so that every block is an aligned run: n = z*16 + (y/2)*8 + (x/2)*4 +
    // (y&1)*2 + (x&1).
    // (the whole-tile attempt is a call of its own so that its shape is a compile-time constant)
    if (process_block<FORCE, COUNT, SPH_TX, SPH_TY, SPH_TZ>(sh, recP, recV, tthr, rcell, A, 0, 0, 0, stamp)) return;
    int cursor = 0, size = SPH_TCELLS / 2;
    while (cursor < SPH_TCELLS) {
        const int bx = ((cursor >> 2) & 1) * 2 + (cursor & 1), by = ((cursor >> 3) & 1) * 2 + ((cursor >> 1) & 1),
                  bz = cursor >> 4;
        bool ok; // one instantiation per shape: the shape is a compile-time constant in each
        switch (size) {
        case 16: ok = process_block<FORCE, COUNT, 4, 4, 1>(sh, recP, recV, tthr, rcell, A, bx, by, bz, stamp); break;
        case 8: ok = process_block<FORCE, COUNT, 4, 2, 1>(sh, recP, recV, tthr, rcell, A, bx, by, bz, stamp); break;
        case 4: ok = process_block<FORCE, COUNT, 2, 2, 1>(sh, recP, recV, tthr, rcell, A, bx, by, bz, stamp); break;
        default: ok = process_block<FORCE, COUNT, 1, 1, 1>(sh, recP, recV, tthr, rcell, A, bx, by, bz, stamp); break;
        }
        if (ok) {
            cursor += size;
        } else if (size > 1) {
            size = size == 4 ? 1 : size >> 1;
            continue;
        } else {
            process_cell_global<FORCE, COUNT>(sh, A, bx, by, bz);
            __syncthreads();
            cursor += 1;
        }
        const int low = cursor & -cursor; // largest aligned block at the new cursor
        size = low >= 16 ? 16 : low >= 8 ? 8 : low >= 4 ? 4 : 1;
    }
}

// Persistent workgroups: each takes the next non-empty tile from the list until it is exhausted
// (every wave leaves through the same uniform test).  The grid is sized to what the chip holds, so
// no workgroup is launched for the empty part of the table.
template <bool FORCE, bool COUNT>
// second launch bound = waves per SIMD the register allocation must allow (what the LDS allows)
__global__ __launch_bounds__(FORCE ? NB_FTHREADS : NB_DTHREADS, (FORCE ? NB_FLB : NB_DLB) - (COUNT ? 1 : 0))
void k_neighbors(NbArgs A)
{
    __shared__ TileShared sh;
    __shared__ float4 recP[(FORCE ? NB_FCAP : NB_DCAP) + 1]; // x, y, z, m_j (+ the FAR record)
    __shared__ float4 recV[FORCE ? NB_FCAP + 1 : 1];         // vx, vy, vz, q_j = p_j/rho_j^2
    __shared__ float tthr[COUNT ? NB_TCAP : 1]; // COUNT variants: exact predicate threshold (h2 / thrF)
    __shared__ unsigned char rcell[FORCE ? NB_FCAP : NB_DCAP]; // record -> halo cell of the block
    __shared__ uint32_t next;
    // Tiles are taken from the list in chunks of NB_CHUNK through one device-wide counter: a single
    // word serves about 88 returning atomics per microsecond on this chip, so one atomic per
    // tile (65 536 tiles on C3) would cost 0.75 ms per launch by itself.  The next chunk is
    // requested while the current one is being processed.
    const uint32_t ntiles = A.tileCtl[0];
    uint32_t pending = 0;
    if (threadIdx.x == 0) pending = atomicAdd(&A.tileCtl[A.queue], (uint32_t)NB_CHUNK);
    for (;;) {
        if (threadIdx.x == 0) {
            next = pending;
            if (pending < ntiles) pending = atomicAdd(&A.tileCtl[A.queue], (uint32_t)NB_CHUNK);
        }
        __syncthreads();
        const uint32_t first = next;
        if (first >= ntiles) break;
        const uint32_t last = min(first + (uint32_t)NB_CHUNK, ntiles);
        // every thread has read `next` before thread 0 can write it again: each tile has barriers
        for (uint32_t i = first; i < last; i++)
            process_tile<FORCE, COUNT>(sh, recP, recV, tthr, rcell, A, (int)A.tileList[i]);
    }
}
