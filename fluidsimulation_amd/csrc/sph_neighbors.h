// sph_neighbors.h -- the two 27-cell neighbour loops: density/pressure and force.
//
// Reference semantics: ComputeDensityAndPressureJob (FluidSPHByCPUSystem.cs:48-104) and
// ComputeForceJob (:126-236); the device counterparts they replace are the HLSL kernels
// ComputeDensityAndPressure / ComputeForce (FluidSolver.compute:128-272), which walk the table
// per thread and gather every neighbour from global memory.
//
// Mapping (DESIGN.md "neighbour kernels").  The sort key groups particles by TILE of 4x4x2 cells
// (sph_device.h).  A tile is worked through in BLOCKS: the whole tile when its halo (block + one
// cell on every side) fits the LDS image, else parts (4x2x2, 4x2x1, 2x2x1), else single cells
// (the fluid compresses several-fold at the bottom of a deep tank), else -- for a cell whose
// 27-cell neighbourhood still overflows -- the cell's targets against chunks of candidates streamed
// through the image (process_cell_chunked).
//
// Plan once, execute twice.  All the integer bookkeeping of a block is the same for the density
// and the force pass, and it is latency-bound work (table look-ups, scans, dependent LDS reads)
// that the arithmetic kernels cannot hide.  k_plan does it once per step, one workgroup per
// non-empty tile, and writes for every block a PLAN to global memory:
//   hinfo   per halo cell: LDS record index, fluid start (global sorted index), fluid count,
//           boundary start
//   units   a UNIT is (cell, part): up to 32 targets of one cell, T of them x S = 64/T candidate
//           slices over the lanes of one wave
//   slots   a SLOT is one step of that wave through the candidates: lane (target u, slice s) reads
//           the record at  slot entry + 16 s.  The nine (dy,dz) row windows of a cell (cells
//           x-1..x+1, contiguous in the image) are walked one after the other.  In a window's
//           last slot the slices past its end read the records that FOLLOW the window: when those
//           are still records of the same image row (cell x+2 and beyond: farther than one cell
//           from every target, so their weight clamps to exactly 0 as long as no support radius
//           exceeds the cell size) the slot is an ordinary one (list A); otherwise it goes to
//           list B, whose entries carry the number of valid slices, the other lanes read FAR
//   tmat    material index per target;   rcell  halo cell per record
// k_neighbors<FORCE> (persistent workgroups over the block list) then only moves bytes and
// computes: the plan and the halo's particles come in by LDS-DMA (global_load_lds_dwordx4: no VGPR
// staging, 64 records per wave-instruction; fluid then boundary per cell, x-runs contiguous, so a
// particle is read from L2/HBM ~4.5 times per pass instead of 27 times), the waves take the units,
// per-target partial sums are folded over the S slices with DPP adds.
//
// Arithmetic.  The production kernels are branch-free: density uses w/h2 = clamp(1 - d2/h2),
// force uses (h-r)/h = clamp(1 - r/h), so candidates outside the support contribute exactly 0
// and no per-pair predicate is evaluated (pairs within an ulp of the support radius contribute
// O(ulp^2), far below the 1e-5 tolerance of BASELINE.json).  Every inner-loop operand is a VGPR
// and the loop is made of the vector ALU's fast class (fma / mul / add / sub with clamp, neg and
// inline-constant modifiers: ~2.4 cycles per wave64 instruction per SIMD; min / max / med3 /
// compares / shifts / SGPR operands: ~4.2; scalar instructions take issue slots too --
// profiles/r2_microbench_*.txt).  The COUNT variants additionally evaluate the reference's
// predicates exactly -- fp32, non-contracted (this file is compiled with -ffp-contract=off):
// d2 = (dx*dx + dy*dy) + dz*dz; density d2 < h2; force length < h  <=>  d2 < thrF (Mat::thrF)
// and index != j -- so neighbour COUNTS are bit-exact.
#pragma once
#include "sph_device.h"

#define NB_FAR 1.0e18f // padding coordinate: d2 ~ 3e36 stays finite, every kernel weight clamps to 0
#define NB_HX (SPH_TX + 2)
#define NB_HY (SPH_TY + 2)
#define NB_HZ (SPH_TZ + 2)
#define NB_HCELLS (NB_HX * NB_HY * NB_HZ) // 144
#define NB_PTHREADS 64 // workgroup size of the plan kernel: one wave, no workgroup barriers
#ifndef NB_DTHREADS
#define NB_DTHREADS 256 // workgroup size of the density kernel
#endif
#ifndef NB_FTHREADS
#define NB_FTHREADS 512 // workgroup size of the force kernel
#endif
#define NB_MAXWAVES 8
#ifndef NB_DG
#define NB_DG 8 // slots per loop iteration (records in flight per lane), density (8: -2..4 % against 4)
#endif
#ifndef NB_FG
#define NB_FG 2 // same, force
#endif
#ifndef NB_CAP
#define NB_CAP 1184 // LDS records per block (18.5 KB of positions, as much again of velocities)
#endif
#define NB_TCAP 384      // targets per block
#define NB_MAXUNITS 64   // (cell, part) units per block
#define NB_SLOTCAP 1408  // slot-table entries per block
#define NB_MTAB 64       // materials whose loop constants are cached in LDS
#ifndef NB_POOL_PIECE
#define NB_POOL_PIECE 8192 // 16-byte units of plan pool a plan workgroup takes at a time (128 KB)
#endif
#define NB_LIST_PIECE 16   // most block-list entries a plan workgroup can leave unused
#ifndef NB_PLAN_WAVES
#define NB_PLAN_WAVES 4 // waves per SIMD the plan kernel's register allocation must allow
#endif
#ifndef NB_PLAN_WGS_PER_CU
#define NB_PLAN_WGS_PER_CU 16 // persistent plan workgroups (single waves) per CU
#endif
#ifndef NB_CHUNK
#define NB_CHUNK 4 // blocks per work-queue request
#define NB_GRUN 32   // entries of the crowded-cell list a plan workgroup takes at a time
#define NB_GQUEUE 4  // crowded cells per work-queue request
#define NB_GPIECE 2048 // 16-byte units of pool a neighbour workgroup takes at a time for hit masks of crowded cells
#endif
#ifndef NB_DLB
#define NB_DLB 5 // waves per SIMD the density kernel's register allocation must allow
#endif
#ifndef NB_FLB
#define NB_FLB 6 // same, force
#endif
#define NB_MASKWORDS 4    // hit-mask words per lane and unit (units of up to 128 slots)
// largest plan: hinfo + units + slots
#define NB_PLANMAX (NB_HCELLS * 16 + NB_MAXUNITS * 32 + NB_SLOTCAP * 4)

struct UnitDesc {
    int tS;      // targets | slices << 8
    int nAB;     // slots of list A | slots of list B << 16
    int gi0;     // global (sorted) index of the first target
    int self;    // LDS record index of the first target
    int slotA;   // slot-table index of list A
    int slotB;   // slot-table index of list B (= slotA + slots of list A)
    int maskw;   // first hit-mask word of the unit in the block's mask area
    int pad1;
};
// block list entry: x = plan offset in the pool (16-byte units), y = tile,
// z = halo cells | units << 8 | block origin (tile-local) x << 16 | y << 20 | z << 24,
// w = records | targets << 11 | slot entries << 20
typedef int4 BlockHdr;

// control words shared by the plan and the neighbour kernels (SphCtx::tileCtl)
// NB_CTL_ERROR sits behind the words that sph_map_table clears every step: an overflow in step k of
// a long resident run is still there when the host next looks (sph_synchronize, sph_download)
enum { NB_CTL_TILES = 0, NB_CTL_BLOCKS, NB_CTL_GCELLS, NB_CTL_POOL, NB_CTL_QUEUE0, NB_CTL_ERROR = 64 };

// the 6x6x4 halo of a tile as the tables give it
struct HaloTab {
    int fs[NB_HCELLS]; // fluid start (global sorted index) of the cell
    int fc[NB_HCELLS]; // fluid count
    int bs[NB_HCELLS]; // boundary start
    int bc[NB_HCELLS]; // boundary count
};

// LDS of the plan kernel (one wave per workgroup, one tile at a time)
struct PlanShared {
    HaloTab h;
    int spref[NB_HCELLS + 1]; // LDS record index of the block halo cell's records
    int2 cellf[SPH_TCELLS];   // per block cell: halo index in the tile, halo index in the block
    int4 cslot[SPH_TCELLS];   // per block cell: slices, first slot, first slot past the range, slots of list A
    int2 cwin[SPH_TCELLS][9]; // per cell and window: x = full slots | rest << 16 | tail in list A << 24,
                              //                      y = offset in list A | index in list B << 16
    uint16_t slot16[NB_SLOTCAP]; // the block's slot table as it is built: record index | valid slices << 11
    uint32_t poolAt, poolEnd; // this workgroup's current piece of the pool (16-byte units)
    uint32_t listAt, listEnd; // this workgroup's current run of block-list entries
};

// kernel arguments shared by the plan and both passes
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
    int useMasks; // force pass: the density pass of this plan has written the hit masks
    int nmats;
    const uint32_t *tileList; // non-empty tiles (k_tile_list)
    uint32_t *ctl;            // NB_CTL_* words and the work-queue counters
    int queue;                // index of this launch's work-queue counter (two words: blocks, cells)
    BlockHdr *blockList;
    int blockCap;
    int2 *gcellList;          // crowded cells (process_cell_chunked): tile (-1: unused entry), x | y << 8 | z << 16
    int gcellCap;
    uint4 *pool;              // plans
    uint32_t poolCap;         // in 16-byte units
    uint32_t planGrid;        // workgroups of the k_plan launch that made this plan (static runs / pieces)
    uint32_t *gmaskOff;       // per cell of gcellList: pool offset of its hit masks (0xffffffff: none)
    int cap1;     // the same for blocks of a single cell (denser cells take the streamed path)
};

// The first run of block-list entries and the first piece of the pool of every plan workgroup are
// STATIC -- workgroup b owns entries [b per, (b + 1) per), per = ceil(tiles / workgroups), and pool
// units [b NB_POOL_PIECE, (b + 1) NB_POOL_PIECE) -- so that a launch does not begin with two
// returning atomics per workgroup on two words (8192 of them at 88 per microsecond: ~90 us before
// the last workgroup could write its first block).  What the counters NB_CTL_BLOCKS / NB_CTL_POOL
// hand out lies behind these regions.
__device__ __forceinline__ uint32_t plan_static_run(uint32_t ntiles, uint32_t grid)
{
    return (ntiles + grid - 1) / grid;
}
__device__ __forceinline__ uint32_t plan_list_base(uint32_t ntiles, uint32_t grid)
{
    return min(grid, ntiles) * plan_static_run(ntiles, grid);
}
__device__ __forceinline__ uint32_t plan_pool_base(uint32_t grid) { return grid * (uint32_t)NB_POOL_PIECE; }

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

// Workgroup barrier for LDS hand-overs that does not wait for the wave's global memory operations:
// __syncthreads() is  s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier, and vmcnt counts stores and returning
// atomics too, so every barrier exposes the latency of whatever the wave has written since the last
// one (the hit masks, the work-queue request).  Where only LDS contents change hands, this is enough.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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

// inclusive scan over the 64 lanes of the wave: rows of 16, then the row totals (row_bcast:15 into
// rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v = row16_incl_scan(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
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

#if defined(NB_STAMPS) || defined(NB_PLAN_TRACE)
static __device__ unsigned long long g_stamps[32];
#endif
#ifdef NB_STAMPS // experimental builds: cycles per phase, summed over workgroups (thread 0)
#define NB_STAMP0() unsigned long long stamp_ = __builtin_amdgcn_s_memtime()
#define NB_STAMP(i)                                                                              \
    do {                                                                                         \
        if (threadIdx.x == 0) {                                                                  \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                        \
            atomicAdd(&g_stamps[(i)], now_ - stamp_);                                            \
            stamp_ = now_;                                                                       \
        }                                                                                        \
    } while (0)
#else
#define NB_STAMP0()
#define NB_STAMP(i)
#endif

#ifdef NB_PLAN_TRACE // experimental builds: cycles per phase of the plan kernel, summed per wave in
                     // registers and added to g_stamps once, at the end (no traffic while it runs)
struct PlanTrace {
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
    __device__ __forceinline__ void start() { last = (uint32_t)__builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void mark(int i)
    {
        const uint32_t now = (uint32_t)__builtin_amdgcn_s_memtime();
        acc[i] += now - last;
        last = now;
    }
};
#define PT_PARAM , PlanTrace &pt
#define PT_ARG , pt
#define PT_MARK(i) pt.mark(i)
#else
#define PT_PARAM
#define PT_ARG
#define PT_MARK(i)
#endif

// table look-ups for the 6x6x4 halo of the tile
__device__ __forceinline__ void tile_lookup(HaloTab &h, int tile, const NbArgs &A)
{
    const GridDesc &g = A.g;
    const int tx = tile % g.ntx, t2 = tile / g.ntx, ty = t2 % g.nty, tz = t2 / g.nty;
    for (int c = threadIdx.x; c < NB_HCELLS; c += blockDim.x) {
        const int hx = c % NB_HX, hy = (c / NB_HX) % NB_HY, hz = c / (NB_HX * NB_HY);
        const int x = wrapi(tx * SPH_TX + hx - 1, g.gx), y = wrapi(ty * SPH_TY + hy - 1, g.gy),
                  z = wrapi(tz * SPH_TZ + hz - 1, g.gz);
        const uint32_t k = tiled_key(x, y, z, g);
        const int2 f = A.table[k];
        int2 b = make_int2(0, 0);
        if (A.btable) b = A.btable[k];
        h.fs[c] = f.x;
        h.fc[c] = f.y - f.x;
        h.bs[c] = b.x;
        h.bc[c] = b.y - b.x;
    }
    __syncthreads();
}

// the same look-ups for the single-wave plan kernel, in two halves: the loads of the next tile are
// in flight while the current one is planned (lane l: halo cells l, l + 64, l + 128)
struct HaloRegs {
    int2 f[3], b[3];
};
__device__ __forceinline__ void tile_lookup_load(HaloRegs &r, int tile, const NbArgs &A)
{
    const GridDesc &g = A.g;
    const int tx = tile % g.ntx, t2 = tile / g.ntx, ty = t2 % g.nty, tz = t2 / g.nty;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        int c = (int)threadIdx.x + 64 * i;
        asm volatile("" : "+v"(c)); // (recomputed per tile, see plan_block)
        r.f[i] = r.b[i] = make_int2(0, 0);
        if (c < NB_HCELLS) {
            const int hx = c % NB_HX, hy = (c / NB_HX) % NB_HY, hz = c / (NB_HX * NB_HY);
            const int x = wrapi(tx * SPH_TX + hx - 1, g.gx), y = wrapi(ty * SPH_TY + hy - 1, g.gy),
                      z = wrapi(tz * SPH_TZ + hz - 1, g.gz);
            const uint32_t k = tiled_key(x, y, z, g);
            r.f[i] = A.table[k];
            if (A.btable) r.b[i] = A.btable[k];
        }
    }
}
__device__ __forceinline__ void tile_lookup_store(HaloTab &h, const HaloRegs &r)
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int c = (int)threadIdx.x + 64 * i;
        if (c < NB_HCELLS) {
            h.fs[c] = r.f[i].x;
            h.fc[c] = r.f[i].y - r.f[i].x;
            h.bs[c] = r.b[i].x;
            h.bc[c] = r.b[i].y - r.b[i].x;
        }
    }
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
                                             int &cnt, uint32_t &hits)
{
    const float dx = t.x - c.x, dy = t.y - c.y, dz = t.z - c.z;
    if (COUNT) {
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        cnt += (d2 < t.h2) ? 1 : 0;
        const float w = fmaxf(t.h2 - d2, 0.f);
        acc = __builtin_fmaf(w * w, w, acc);
    } else {
        const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        // w / h2 = clamp(1 - d2/h2); the same fma without the clamp gives the MISS bit (its sign)
        // for the force pass: hits = hits << 1 | miss, the newest slot in bit 0.  Both are
        // full-rate instructions (a compare writing VCC and an add-with-carry cost 1.7x as much)
        const float u = __builtin_fmaf(d2, t.negInvH2, 1.0f);
        float w;
        asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(w) : "v"(d2), "v"(t.negInvH2));
        acc = __builtin_fmaf(w * w, w, acc);
        hits = __builtin_amdgcn_alignbit(hits, __builtin_bit_cast(uint32_t, u), 31);
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
        // two distinct particles at the same place: the reference's math.normalize(0) is 0 * inf,
        // its pressure force -- and so the particle's force -- is NaN (FluidSPHByCPUSystem.cs:222).
        // This variant reproduces that; the production variant returns the finite limit (the
        // pair's pressure term is 0), see DESIGN.md section 2.
        if (notSelf && d2 == 0.f) a.px = a.py = a.pz = __builtin_nanf("");
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
    uint32_t hits = 0; // miss bit of the last 32 slots walked, newest in bit 0 (production variant)
    __device__ __forceinline__ void pair(const float4 &cp, const float4 &, bool)
    {
        density_pair<COUNT>(t, cp, acc, cnt, hits);
    }
    // folds the slices and writes rho, p, q for target ti
    __device__ __forceinline__ float4 finish(const NbArgs &A, int S, int slice, bool writer, int ti,
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
        return make_float4(0.f, 0.f, 0.f, 0.f);
    }
};

template <bool COUNT> struct Pass<true, COUNT> {
    ForceTarget t;
    ForceAcc a;
    uint32_t hits = 0; // (unused: the force pass reads the masks)
    __device__ __forceinline__ void pair(const float4 &cp, const float4 &cv, bool notSelf)
    {
        force_pair<COUNT>(t, cp, cv, notSelf, a);
    }
    // (the force itself is returned, not stored: exec_block stores it one unit later, see there)
    __device__ __forceinline__ float4 finish(const NbArgs &A, int S, int slice, bool writer, int ti,
                                             const float4 &fc4)
    {
        float v[3] = {fc4.y * a.px + fc4.z * a.vx, fc4.y * a.py + fc4.z * a.vy, fc4.y * a.pz + fc4.z * a.vz};
        int c2 = a.cnt;
        fold_slices<3>(v, c2, S, slice);
        const float fx = v[0], fy = v[1], fz = v[2];
        // forces = forcePressure + forceViscosity + gravity * rho_i  (:140,:202)
        const float trho = fc4.w;
        if (COUNT && writer) A.counts[ti] = c2;
        return make_float4(fx + A.gx * trho, fy + A.gy * trho, fz + A.gz * trho, 0.f);
    }
};

// loop constant (-1/h2 or -1/h) and exact-predicate threshold (h2 or thrF) of a material
template <bool FORCE> __device__ __forceinline__ void loop_constants(const Mat &m, float &k, float &thr)
{
    k = FORCE ? m.negInvH : m.negInvH2;
    thr = FORCE ? m.thrF : m.h2;
}
// what the finishing constants of a target are made of, per material (NbShared::mfin):
//   density: -, m*315/(64 pi h^9) [* h2^3 in the production variant], k, rho0;  force: h, spiky, mu, -
template <bool FORCE, bool COUNT> __device__ __forceinline__ float4 finish_row(const Mat *m)
{
    if (FORCE) return make_float4(m->h, m->spiky, m->viscosity, 0.f);
    return make_float4(0.f, COUNT ? m->poly6 : m->poly6h6, m->gasConstant, m->restDensity);
}
// finishing constants of a target (see Pass) from its material's row and, force pass, its density d
template <bool FORCE> __device__ __forceinline__ float4 finish_constants(const float4 &row, float d)
{
    if (FORCE) {
        const float h = row.x, spiky = row.y, mu = row.z;
        return make_float4(0.f, d * spiky * (h * h), mu / d * spiky * h, d);
    }
    return row;
}

// floor(a / b) for 0 <= a < 2^20, 1 <= b <= 64 without an integer division: (a + 0.5) / b is at
// least 0.5/64 away from every integer, far more than the error of v_rcp
__device__ __forceinline__ int div_small(int a, int b)
{
    return (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)b));
}


// ------------------------------------------------------------------------------------------
// PLAN KERNEL

// The workgroup's next run of the block list, requested ahead of need (lane 0's registers: written
// to LDS, the answer of the atomic would be waited for at once).
struct ListAhead {
    uint32_t at = 0, n = 0; // n = 0: nothing requested
    uint32_t base = 0;      // plan_list_base: where the runs the counter hands out begin
    int tilesLeft = 0;      // tiles this workgroup still has to plan after the current one
};

// The plan of one block (bx0.., sx x sy x sz cells, tile-local), by one wave.  Returns false,
// having done nothing, when the block's halo, targets, units or slots do not fit.  hinfo.z = fluid
// count | boundary count << 16.
template <int sx, int sy, int sz>
__device__ __forceinline__ bool plan_block(PlanShared &sh, const NbArgs &A, int tile, int bx0, int by0,
                                           int bz0, ListAhead &la PT_PARAM)
{
    constexpr int hxN = sx + 2, hyN = sy + 2, nh = hxN * hyN * (sz + 2), ncells = sx * sy * sz;
    static_assert(nh <= 3 * 64 && ncells <= 64, "three halo cells and one block cell per lane");
    // (the lane's cell coordinates are recomputed for every block: hoisted out of the tile loop
    // they are spilled, and a scratch reload is a load like any other -- see k_plan)
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));

    // 1. records per halo cell (lane l: cells 3l .. 3l+2) and their places in the image
    int myf[3], rc[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int h = 3 * lane + i;
        myf[i] = 0;
        rc[i] = 0;
        if (h < nh) {
            const int hx = h % hxN, hy = (h / hxN) % hyN, hz = h / (hxN * hyN);
            myf[i] = tile_halo_index(bx0 + hx, by0 + hy, bz0 + hz);
            rc[i] = sh.h.fc[myf[i]] + sh.h.bc[myf[i]];
        }
    }
    const int rsum = rc[0] + rc[1] + rc[2];
    const int rincl = wave_incl_scan(rsum);
    const int total = __builtin_amdgcn_readlane(rincl, 63);
    PT_MARK(2);
    if (total > (ncells == 1 ? A.cap1 : A.cap)) return false;
    const int rex[3] = {rincl - rsum, rincl - rsum + rc[0], rincl - rsum + rc[0] + rc[1]};
#pragma unroll
    for (int i = 0; i < 3; i++)
        if (3 * lane + i < nh) sh.spref[3 * lane + i] = rex[i];
    if (lane == 0) sh.spref[nh] = total;

    // 2. per block cell: targets, parts, slices and the slots of the nine windows
    int tcnt = 0, P = 0, per = 0, S = 0, nA = 0, nB = 0, mw = 0, uw = 0, nsl = 0, f = 0, fb = 0;
    if (lane < ncells) {
        const int cx = lane % sx, cy = (lane / sx) % sy, cz = lane / (sx * sy);
        f = tile_halo_index(bx0 + cx + 1, by0 + cy + 1, bz0 + cz + 1);
        fb = ((cz + 1) * hyN + (cy + 1)) * hxN + (cx + 1);
        tcnt = sh.h.fc[f];
        sh.cellf[lane] = make_int2(f, fb);
        if (tcnt > 0) {
            // at most 32 targets per unit (S >= 2), and at least 8 units per block
            P = max((tcnt + 31) >> 5, min(tcnt, (NB_MAXWAVES + ncells - 1) / ncells));
            if (P == 1) {
                per = tcnt;
            } else {
                per = div_small(tcnt, P);
                while (per * P < tcnt) per++;
                P = div_small(tcnt, per);
                while (P * per < tcnt) P++;
            }
            S = min(slices_for(per), 32); // (a tail's valid slices, < S, fit 5 bits)
#pragma unroll
            for (int wi = 0; wi < 9; wi++) {
                // first cell of the window in the tile's halo (rows of NB_HX cells)
                const int f0 = f + ((wi / 3) - 1) * (NB_HX * NB_HY) + ((wi % 3) - 1) * NB_HX - 1;
                const int len = (sh.h.fc[f0] + sh.h.bc[f0]) + (sh.h.fc[f0 + 1] + sh.h.bc[f0 + 1]) +
                                (sh.h.fc[f0 + 2] + sh.h.bc[f0 + 2]);
                int rest = 0; // records of the same block row behind the window
#pragma unroll
                for (int x = 3; x < hxN; x++)
                    if (cx + x < hxN) rest += sh.h.fc[f0 + x] + sh.h.bc[f0 + x];
                const int nfull = div_small(len, S), rem = len - nfull * S;
                const int tailA = (rem != 0 && A.overread && S - rem <= rest) ? 1 : 0;
                const int tailB = (rem != 0 && !tailA) ? 1 : 0;
                sh.cwin[lane][wi] = make_int2(nfull | (rem << 16) | (tailA << 24), nA | (nB << 16));
                nA += nfull + tailA;
                nB += tailB;
            }
            nsl = (nA + nB + 3) & ~3;
            // units | hit-mask words (one word per 32 slots, lane and unit) << 12; the clamps only
            // act on cells that make the block fail the checks below
            mw = (nA + nB + 31) >> 5;
            uw = min(P, NB_MAXUNITS + 1) | (min(P * mw, 4095) << 12);
        }
    }
    PT_MARK(3);
    const int tincl = wave_incl_scan(tcnt), uincl = wave_incl_scan(uw), sincl = wave_incl_scan(nsl);
    const int ttotal = __builtin_amdgcn_readlane(tincl, 63), ut = __builtin_amdgcn_readlane(uincl, 63),
              nslots = __builtin_amdgcn_readlane(sincl, 63);
    const int nunits = ut & 0xfff, nwords = ut >> 12;
    if (ttotal > NB_TCAP || nunits > NB_MAXUNITS || nslots > NB_SLOTCAP) return false;
    if (ttotal == 0) return true;
    const int uex = uincl - uw, sex = sincl - nsl;

    // Room for the plan (and, behind it, the hit masks: 64 lanes x 4 bytes per word) and a place
    // in the block list.  Both come out of pieces that the workgroup takes from the device-wide
    // counters: one word serves only ~88 returning atomics per microsecond, one atomic per block
    // would cost most of a millisecond per step.
    int off = 0, idx = 0;
    if (lane == 0) {
        const uint32_t units16 = ((uint32_t)(nh * 16 + nunits * 32 + nslots * 4) >> 4) + (uint32_t)nwords * 16;
        if (sh.poolAt + units16 > sh.poolEnd) {
            const uint32_t piece = max(units16, (uint32_t)NB_POOL_PIECE);
            sh.poolAt = plan_pool_base(gridDim.x) + atomicAdd(&A.ctl[NB_CTL_POOL], piece);
            sh.poolEnd = sh.poolAt + piece;
        }
        // The neighbour kernels walk every entry of the list, and an unused one costs them a
        // request with nothing to do: the first run has one entry per tile of this workgroup (most
        // tiles are one block); when tiles fall into several blocks and the entries left no longer
        // cover one per remaining tile, the next run is requested -- while there are still entries,
        // so that its answer is not waited for.
        if (sh.listAt == sh.listEnd) {
            if (la.n == 0) {
                la.n = (uint32_t)la.tilesLeft + 2;
                la.at = la.base + atomicAdd(&A.ctl[NB_CTL_BLOCKS], la.n);
            }
            sh.listAt = la.at;
            sh.listEnd = la.at + la.n;
            la.n = 0;
        }
        const uint32_t o = sh.poolAt, i = sh.listAt;
        const bool ok = o + units16 <= A.poolCap && i < (uint32_t)A.blockCap;
        if (!ok) { // no room: the entry (if it exists) is marked empty, the host sees the flag
            atomicOr(&A.ctl[NB_CTL_ERROR], 1u);
            if (i < (uint32_t)A.blockCap) A.blockList[i] = make_int4(0, 0, 0, 0);
        }
        sh.poolAt += units16;
        sh.listAt += 1;
        const int left = (int)(sh.listEnd - sh.listAt);
        if (la.n == 0 && left < la.tilesLeft) {
            la.n = (uint32_t)min(la.tilesLeft - left + 1 + (la.tilesLeft >> 2), NB_LIST_PIECE);
            la.at = la.base + atomicAdd(&A.ctl[NB_CTL_BLOCKS], la.n);
        }
        off = ok ? (int)o : -1;
        idx = (int)i;
    }
    off = __builtin_amdgcn_readfirstlane(off);
    idx = __builtin_amdgcn_readfirstlane(idx);
    if (lane < ncells && tcnt > 0) sh.cslot[lane] = make_int4(S, sex, sex + nsl, nA);
    wave_lds_fence();
    PT_MARK(4);

    // 3. the plan goes straight to global memory: hinfo | units | slots
    if (off >= 0) {
        int4 *hinfoG = (int4 *)(A.pool + off);
        UnitDesc *unitG = (UnitDesc *)(hinfoG + nh);
        uint32_t *slotG = (uint32_t *)(unitG + nunits);
#pragma unroll
        for (int i = 0; i < 3; i++) { // lane l writes cells l, l + 64, l + 128: whole lines per store
            const int h = lane + 64 * i;
            if (h < nh) {
                const int hx = h % hxN, hy = (h / hxN) % hyN, hz = h / (hxN * hyN);
                const int hf = tile_halo_index(bx0 + hx, by0 + hy, bz0 + hz);
                hinfoG[h] = make_int4(sh.spref[h], sh.h.fs[hf], sh.h.fc[hf] | (sh.h.bc[hf] << 16), sh.h.bs[hf]);
            }
        }
        if (lane < ncells && tcnt > 0) {
            const int cstart = sh.h.fs[f], clds = sh.spref[fb];
#pragma nounroll
            for (int part = 0; part < P; part++) {
                const int t0 = part * per;
                UnitDesc u;
                u.tS = min(per, tcnt - t0) | (S << 8);
                u.nAB = nA | (nB << 16);
                u.gi0 = cstart + t0;
                u.self = clds + t0;
                u.slotA = sex;
                u.slotB = sex + nA;
                u.maskw = (uex >> 12) + part * mw;
                u.pad1 = 0;
                unitG[(uex & 0xfff) + part] = u;
            }
        }
        PT_MARK(5);
        // one task per (cell, window): its slots of list A from the front of the cell's range,
        // its tail slot of list B (if any) behind them.  The table is built in LDS (16 bits an
        // entry) and copied out 16 bytes per lane: as single dwords ~12 bytes apart from lane to
        // lane it cost four times the write requests, which the reorder pass on the other stream
        // -- the pair is bound by HBM bandwidth -- paid for.
        for (int t = lane; t < ncells * 9; t += NB_PTHREADS) {
            const int c = t / 9, wi = t - c * 9;
            const int2 cf = sh.cellf[c];
            if (sh.h.fc[cf.x] == 0) continue;
            const int4 cs = sh.cslot[c];
            const int2 wdv = sh.cwin[c][wi];
            const int nfull = wdv.x & 0xffff, rem = (wdv.x >> 16) & 0xff, tailA = (wdv.x >> 24) & 1;
            const int aoff = wdv.y & 0xffff, bidx = wdv.y >> 16;
            const int h0 = cf.y + ((wi / 3) - 1) * (hxN * hyN) + ((wi % 3) - 1) * hxN - 1;
            const int ws = sh.spref[h0], Sc = cs.x;
            uint16_t *dst = sh.slot16 + cs.y + aoff;
#pragma nounroll
            for (int j = 0; j < nfull; j++) dst[j] = (uint16_t)(ws + j * Sc);
            if (rem) {
                const int tail = ws + nfull * Sc;
                if (tailA) dst[nfull] = (uint16_t)tail;
                else sh.slot16[cs.y + cs.w + bidx] = (uint16_t)(tail | (rem << 11));
            }
        }
        wave_lds_fence();
        for (int q = lane; q * 4 < nslots; q += NB_PTHREADS) {
            const uint2 v = *(const uint2 *)(sh.slot16 + 4 * q);
            const uint32_t e[4] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
            uint4 o;
            o.x = ((e[0] & 0x7ffu) << 4) | ((e[0] >> 11) << 16);
            o.y = ((e[1] & 0x7ffu) << 4) | ((e[1] >> 11) << 16);
            o.z = ((e[2] & 0x7ffu) << 4) | ((e[2] >> 11) << 16);
            o.w = ((e[3] & 0x7ffu) << 4) | ((e[3] >> 11) << 16);
            ((uint4 *)slotG)[q] = o;
        }
        if (lane == 0)
            A.blockList[idx] =
                make_int4(off, tile, nh | (nunits << 8) | (bx0 << 16) | (by0 << 20) | (bz0 << 24),
                          total | (ttotal << 11) | (nslots << 20));
    }
    PT_MARK(6);
    wave_lds_fence(); // the LDS tables are reused by the next block
    PT_MARK(7);
    return true;
}

// one wave per non-empty tile at a time
__global__ __launch_bounds__(NB_PTHREADS) __attribute__((amdgpu_waves_per_eu(NB_PLAN_WAVES, NB_PLAN_WAVES))) void k_plan(NbArgs A)
{
    __shared__ PlanShared sh;
    const uint32_t ntiles = A.ctl[NB_CTL_TILES];
    const int myTiles = blockIdx.x < ntiles ? (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    ListAhead la;
    la.base = plan_list_base(ntiles, gridDim.x);
    if (threadIdx.x == 0) { // the static first piece and run (about one entry per tile)
        const uint32_t per = myTiles > 0 ? plan_static_run(ntiles, gridDim.x) : 0u;
        sh.poolAt = blockIdx.x * (uint32_t)NB_POOL_PIECE;
        sh.poolEnd = sh.poolAt + (uint32_t)NB_POOL_PIECE;
        sh.listAt = blockIdx.x * per;
        sh.listEnd = sh.listAt + per;
    }
    la.tilesLeft = myTiles;
    uint32_t gAt = 0, gEnd = 0; // lane 0: this workgroup's run of gcellList entries
#ifdef NB_PLAN_TRACE
    PlanTrace pt;
    pt.start();
#endif
    // This workgroup's tiles are tileList[blockIdx.x + k * gridDim.x]: lane l of a register holds
    // the tile of round 64 c + l, loaded a chunk ahead, so that the loop has no load whose result
    // it needs at once.  s_waitcnt vmcnt counts stores too (gfx950 has no separate store counter):
    // any wait for a load in the middle of a tile also waits until every store of the tile before
    // has drained, which takes far longer than the load (profiles: 45 % of the kernel).  The only
    // place left where the wave waits for memory is the top of the loop.
    const uint32_t grid = gridDim.x;
    auto load_tiles = [&](uint32_t k0) -> int {
        const uint64_t i = (uint64_t)blockIdx.x + (uint64_t)(k0 + threadIdx.x) * grid;
        return i < ntiles ? (int)A.tileList[i] : 0;
    };
    int chunkCur = load_tiles(0), chunkNext = load_tiles(64);
    HaloRegs hr;
    if (blockIdx.x < ntiles) tile_lookup_load(hr, __builtin_amdgcn_readlane(chunkCur, 0), A);
    uint32_t k = 0;
    for (uint32_t ti = blockIdx.x; ti < ntiles; ti += grid, k++) {
        tile_lookup_store(sh.h, hr);
        wave_lds_fence();
        PT_MARK(0);
        const int tileCur = __builtin_amdgcn_readlane(chunkCur, (int)(k & 63));
        la.tilesLeft--;
        if (ti + grid < ntiles) { // the next tile's look-ups
            if (((k + 1) & 63) == 0) {
                chunkCur = chunkNext;
                chunkNext = load_tiles(k + 1 + 64);
            }
            tile_lookup_load(hr, __builtin_amdgcn_readlane(chunkCur, (int)((k + 1) & 63)), A);
        }
        PT_MARK(1);
        // The largest blocks whose 27-cell halo fits the LDS image: the whole tile (halo 144
        // cells for 32), else per y-half a 4x2x2 block (96 for 16), else per layer 4x2x1 (72 for
        // 8), 2x2x1 (48 for 4), single cells (27 for 1), else the crowded-cell path.  Blocks stay
        // long in x, the direction of the row windows.  One instantiation per shape.
        if (plan_block<SPH_TX, SPH_TY, SPH_TZ>(sh, A, tileCur, 0, 0, 0, la PT_ARG)) continue;
        for (int yh = 0; yh < 2; yh++) {
            if (plan_block<4, 2, 2>(sh, A, tileCur, 0, 2 * yh, 0, la PT_ARG)) continue;
            for (int z = 0; z < 2; z++) {
                if (plan_block<4, 2, 1>(sh, A, tileCur, 0, 2 * yh, z, la PT_ARG)) continue;
                for (int xh = 0; xh < 2; xh++) {
                    if (plan_block<2, 2, 1>(sh, A, tileCur, 2 * xh, 2 * yh, z, la PT_ARG)) continue;
                    for (int q = 0; q < 4; q++) {
                        const int bx = 2 * xh + (q & 1), by = 2 * yh + (q >> 1);
                        if (plan_block<1, 1, 1>(sh, A, tileCur, bx, by, z, la PT_ARG)) continue;
                        if (threadIdx.x == 0 && sh.h.fc[tile_halo_index(bx + 1, by + 1, z + 1)] > 0) {
                            // (entries are taken a run at a time: this is the one place in the
                            // tile loop that waits for memory, see above)
                            if (gAt == gEnd) {
                                gAt = atomicAdd(&A.ctl[NB_CTL_GCELLS], (uint32_t)NB_GRUN);
                                gEnd = gAt + NB_GRUN;
                            }
                            const uint32_t idx = gAt++;
                            if (idx < (uint32_t)A.gcellCap) {
                                A.gcellList[idx] = make_int2(tileCur, bx | (by << 8) | (z << 16));
                                A.gmaskOff[idx] = 0xffffffffu;
                            }
                            else atomicOr(&A.ctl[NB_CTL_ERROR], 2u);
                        }
                    }
                }
            }
        }
        wave_lds_fence(); // sh.h is rewritten by the next tile
    }
#ifdef NB_PLAN_TRACE
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_stamps[i], (unsigned long long)pt.acc[i]);
#endif
    // the unused entries of this workgroup's last run of the block list: empty blocks
    if (threadIdx.x == 0) {
        for (uint32_t i = sh.listAt; i < sh.listEnd; i++)
            if (i < (uint32_t)A.blockCap) A.blockList[i] = make_int4(0, 0, 0, 0);
        for (uint32_t i = la.at; i < la.at + la.n; i++) // a run requested ahead and not needed
            if (i < (uint32_t)A.blockCap) A.blockList[i] = make_int4(0, 0, 0, 0);
        for (uint32_t i = gAt; i < gEnd; i++) // the rest of the last run of cells: no cell
            if (i < (uint32_t)A.gcellCap) A.gcellList[i] = make_int2(-1, 0);
    }
}

// ------------------------------------------------------------------------------------------
// NEIGHBOUR KERNELS

template <bool FORCE, bool COUNT> struct NbShared {
    __attribute__((aligned(16))) unsigned char plan[NB_PLANMAX]; // hinfo | units | slots | tmat | rcell
    float4 recP[NB_CAP + 1];             // x, y, z, m_j (+ the FAR record)
    float4 recV[FORCE ? NB_CAP + 1 : 1]; // vx, vy, vz, q_j = p_j/rho_j^2
    float4 mtab[NB_MTAB];                // per material: -1/h2, -1/h, h2, thrF
    float4 mfin[NB_MTAB];                // per material: finish_row
    BlockHdr hdr[NB_CHUNK];
    uint32_t next;
    uint32_t gnext; // place in the crowded-cell list (its own word: `next` may still be read by a wave that leaves the block loop)
    uint32_t gpoolAt, gpoolEnd; // this workgroup's piece of the pool for hit masks of crowded cells
};
static_assert(sizeof(HaloTab) <= NB_PLANMAX, "the crowded-cell path keeps the tile's halo in the plan area");
static_assert(NB_CAP < 2048, "PlanShared::slot16 holds record indices in 11 bits");

// one planned block.  MULTI: the scene has more than one material (else no target's material index
// is loaded: with a load on any path of the unit loop, the compiler's wait for it at the loop head
// would make every unit wait for the stores of the unit before)
template <bool FORCE, bool COUNT, bool MULTI>
__device__ __forceinline__ void exec_block(NbShared<FORCE, COUNT> &sh, const NbArgs &A, const BlockHdr hdr,
                                           bool request, uint32_t chunk, uint32_t &pending)
{
    constexpr int NT = FORCE ? NB_FTHREADS : NB_DTHREADS, NB_WAVES = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nh = hdr.z & 0xff, nunits = (hdr.z >> 8) & 0xff;
    const int total = hdr.w & 0x7ff, ttotal = (hdr.w >> 11) & 0x1ff, nslots = (hdr.w >> 20) & 0xfff;
    const int oUnits = nh * 16, oSlots = oUnits + nunits * 32, planBytes = oSlots + nslots * 4;
    (void)ttotal;
    uint32_t *maskBlk = (uint32_t *)(A.pool + hdr.x + (planBytes >> 4)); // behind the plan

    NB_STAMP0();
    // 1. the plan: one linear LDS-DMA copy
    {
        const char *src = (const char *)(A.pool + hdr.x);
        for (int o0 = 0; o0 < planBytes; o0 += NT * 16) {
            const int ow = o0 + w * 1024; // wave-uniform
            if (ow + lane * 16 < planBytes)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + ow + lane * 16), (lptr_t)(sh.plan + ow), 16, 0, 0);
        }
    }
    __syncthreads();
    NB_STAMP((FORCE ? 16 : 8) + 0);

    // 2. the records of the halo, fluid then boundary per cell.  First every image slot gets the
    // index of its source record, written into the slot itself (eight lanes per halo cell); then
    // the waves copy the image in pieces of 64 records by LDS-DMA (global_load_lds_dwordx4 takes a
    // source address per lane and writes 1 KiB of LDS from the M0 base: no VGPR staging).
    const int4 *hinfo = (const int4 *)sh.plan;
    const UnitDesc *units = (const UnitDesc *)(sh.plan + oUnits);
    float4 *recP = sh.recP, *recV = sh.recV;
    {
        uint32_t *slot0 = (uint32_t *)recP;
        const int k = lane & 7;
        for (int c = w * 8 + (lane >> 3); c < nh; c += NB_WAVES * 8) {
            const int4 hi = hinfo[c];
            const int fc = hi.z & 0xffff, cnt = fc + (hi.z >> 16);
            for (int k2 = k; k2 < cnt; k2 += 8)
                slot0[4 * (hi.x + k2)] =
                    k2 < fc ? (uint32_t)(hi.y + k2) : 0x80000000u | (uint32_t)(hi.w + (k2 - fc));
        }
    }
    lds_barrier();
    for (int p0 = w * 64; p0 < total; p0 += NT) { // wave-uniform
        if (p0 + lane < total) {
            const uint32_t src = ((const uint32_t *)recP)[4 * (p0 + lane)];
            const bool bnd = (src >> 31) != 0; // boundary: v = 0, q = 0
            const float4 *gp = bnd ? A.bposm + (src & 0x7fffffffu) : A.posm + src;
            __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(recP + p0), 16, 0, 0);
            if (FORCE) {
                const float4 *gv = bnd ? A.zeros : (const float4 *)A.velq + src;
                __builtin_amdgcn_global_load_lds((gptr_t)gv, (lptr_t)(recV + p0), 16, 0, 0);
            }
        }
    }
    if (tid == 0) {
        recP[NB_CAP] = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f); // what masked lanes read
        if (FORCE) recV[NB_CAP] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // What a unit needs from global memory -- the targets' material indices (only with more than
    // one material), force pass: their densities and the lanes' first hit-mask words -- is
    // fetched one unit ahead, the first unit's here: a wait for a load in the middle of a unit
    // would also wait for the unit's stores (vmcnt counts both).
    const bool walkMasks = FORCE && !COUNT && A.useMasks;
    uint32_t miNext = 0, m0Next = 0, m1Next = 0; // (m0, m1: the lane's first two hit-mask words)
    float rhoNext = 0.f;
    if (w < nunits) {
        const UnitDesc u0 = units[w];
        const int T0 = u0.tS & 0xff, S0 = u0.tS >> 8;
        int u, sl;
        split_lane(lane, S0, u, sl);
        const int t0 = u0.gi0 + (u < T0 ? u : 0);
        if constexpr (MULTI) miNext = A.mat[t0];
        if (FORCE) rhoNext = A.rhoIn[t0];
        if (walkMasks) {
            m0Next = maskBlk[(size_t)u0.maskw * 64 + lane];
            if ((u0.nAB & 0xffff) + (u0.nAB >> 16) > 32) m1Next = maskBlk[(size_t)u0.maskw * 64 + 64 + lane];
        }
    }
    NB_STAMP((FORCE ? 16 : 8) + 1);
    __syncthreads();
    NB_STAMP((FORCE ? 16 : 8) + 2);
    // the request for the workgroup's next chunk goes out here, behind the last barrier that waits
    // for global memory (the image's DMA): its answer is needed when the chunk is done
    if (request && threadIdx.x == 0) pending = atomicAdd(&A.ctl[A.queue], chunk) + gridDim.x * chunk;

    // 3. the units, dealt to the waves
    const char *baseP = (const char *)recP, *baseV = (const char *)recV;
    const char *slots = (const char *)sh.plan + oSlots;
    const int far16 = NB_CAP * 16;
    // Force pass: a unit's forces are stored at the head of the NEXT unit, behind that unit's loads
    // for the one after.  The wait for those loads at the head of a unit is vmcnt(0), which waits
    // for stores as well: stored at its end, a unit's forces were what the next unit waited for.
    float4 heldF = make_float4(0.f, 0.f, 0.f, 0.f);
    int heldTi = -1;
    for (int unit = w; unit < nunits; unit += NB_WAVES) {
        const UnitDesc ud = units[unit];
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
        const uint32_t mi = MULTI ? miNext : 0u, m0 = m0Next, m1 = m1Next;
        const float rhoi = rhoNext;
        if (unit + NB_WAVES < nunits) { // the next unit's, used one unit later
            const UnitDesc un = units[unit + NB_WAVES];
            const int Tn = un.tS & 0xff, Sn = un.tS >> 8;
            int u2, sl2;
            split_lane(lane, Sn, u2, sl2);
            const int tn = un.gi0 + (u2 < Tn ? u2 : 0);
            if constexpr (MULTI) miNext = A.mat[tn];
            if (FORCE) rhoNext = A.rhoIn[tn];
            if (walkMasks) {
                m0Next = maskBlk[(size_t)un.maskw * 64 + lane];
                if ((un.nAB & 0xffff) + (un.nAB >> 16) > 32) m1Next = maskBlk[(size_t)un.maskw * 64 + 64 + lane];
            }
        }
        if (FORCE && heldTi >= 0) A.force[heldTi] = heldF;
        float4 mt, mf;
        if (mi < NB_MTAB) {
            mt = sh.mtab[mi];
            mf = sh.mfin[mi];
        } else {
            const Mat *m = A.mats + mi;
            mt = make_float4(m->negInvH2, m->negInvH, m->h2, m->thrF);
            mf = finish_row<FORCE, COUNT>(m);
        }
        const float4 fc4 = finish_constants<FORCE>(mf, rhoi);
        {
            const float4 tp = *(const float4 *)(baseP + self16);
            ps.t.x = tp.x; ps.t.y = tp.y; ps.t.z = tp.z;
        }
        if constexpr (FORCE) {
            const float4 tv = *(const float4 *)(baseV + self16);
            ps.t.vx = tv.x; ps.t.vy = tv.y; ps.t.vz = tv.z;
            ps.t.Pi = tv.w; // q_i = p_i / rho_i^2, written by the density kernel
            ps.t.negInvH = mt.y;
            ps.t.tiny = 1.0e-30f;
            ps.t.thrF = mt.w;
        } else {
            ps.t.negInvH2 = mt.x;
            ps.t.h2 = mt.z;
        }
        constexpr int G = FORCE ? NB_FG : NB_DG; // slots per iteration
        // hit masks of the unit: one word per 32 slots (list A, then list B) and lane
        uint32_t *maskU = maskBlk + (size_t)ud.maskw * 64 + lane;
        if (walkMasks) {
            // Force pass over the hits the density pass found: every lane walks the set bits of
            // its own words (bit i of word j = slot 32 j + i of the unit); lanes that have run
            // out of hits are switched off, the wave goes on until its longest list ends.  A
            // candidate the density pass left out is farther than h: its force terms are 0.
            const int nwords = (nA + nB + 31) >> 5;
            uint32_t m = __builtin_bitreverse32(m0);
            const char *sbase = slots + ud.slotA * 4;
            for (int j = 0; j < nwords; j++, sbase += 128) {
                uint32_t mnext = 0; // (word 1 came with word 0, a unit ahead; units of more than
                                    // 64 slots -- crowded cells -- load the rest here)
                if (j + 1 < nwords) mnext = j == 0 ? m1 : maskU[(j + 1) * 64];
                while (m) {
                    const int bit = __builtin_ctz(m);
                    m &= m - 1;
                    const int off = (int)*(const uint16_t *)(sbase + bit * 4) + slice16;
                    const float4 cp = *(const float4 *)(baseP + off);
                    const float4 cv = *(const float4 *)(baseV + off);
                    keep_b128(cp);
                    keep_b128x(cv);
                    ps.pair(cp, cv, true);
                }
                m = __builtin_bitreverse32(mnext);
            }
        } else {
            int g = 0; // slots walked (wave-uniform)
            // stores the miss bits of the last nbits slots as a word of hit bits, slot 32 j + i
            // in bit 31 - i (the force pass reverses the word)
            auto flush = [&](int nbits) {
                if (!FORCE && !COUNT) maskU[((g - 1) >> 5) * 64] = (~ps.hits) << (32 - nbits);
            };
            // list A: lane reads entry + 16 * slice, nothing else per slot
            const char *sp = slots + ud.slotA * 4;
            int k = 0;
            for (; k + G <= nA; k += G, sp += 4 * G) {
                uint32_t e[G];
                if constexpr (G == 4) {
                    const uint4 q = *(const uint4 *)sp;
                    e[0] = q.x; e[1] = q.y; e[2] = q.z; e[3] = q.w;
                } else if constexpr (G == 8) {
                    const uint4 q0 = *(const uint4 *)sp, q1 = *(const uint4 *)(sp + 16);
                    e[0] = q0.x; e[1] = q0.y; e[2] = q0.z; e[3] = q0.w;
                    e[4] = q1.x; e[5] = q1.y; e[6] = q1.z; e[7] = q1.w;
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
                g += G;
                if ((g & 31) == 0) flush(32);
            }
            for (; k < nA; k++, sp += 4) {
                const int off = (int)*(const uint32_t *)sp + slice16;
                const float4 cp = *(const float4 *)(baseP + off);
                float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (FORCE) cv = *(const float4 *)(baseV + off);
                keep_b128(cp);
                if (FORCE) keep_b128x(cv);
                ps.pair(cp, cv, off != self16);
                g++;
                if ((g & 31) == 0) flush(32);
            }
            // list B (right behind list A): slices at or past the entry's count read the FAR record
            for (int kb = 0; kb < nB; kb++, sp += 4) {
                const uint32_t e = *(const uint32_t *)sp;
                const int off = slice < (int)(e >> 16) ? (int)(e & 0xffffu) + slice16 : far16;
                const float4 cp = *(const float4 *)(baseP + off);
                float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (FORCE) cv = *(const float4 *)(baseV + off);
                keep_b128(cp);
                if (FORCE) keep_b128x(cv);
                ps.pair(cp, cv, off != self16);
                g++;
                if ((g & 31) == 0) flush(32);
            }
            if (g & 31) flush(g & 31);
        }
        heldF = ps.finish(A, S2, slice, writer, ti, fc4);
        heldTi = writer ? ti : -1;
    }
    if (FORCE && heldTi >= 0) A.force[heldTi] = heldF;
    NB_STAMP((FORCE ? 16 : 8) + 3);
    lds_barrier(); // the LDS image and the plan are reused by the next block
    NB_STAMP((FORCE ? 16 : 8) + 4);
}

// A cell whose 27-cell neighbourhood overflows the LDS image (the bottom of a deep tank while the
// column lands: hundreds of particles per cell).  The reference's hash chains have no occupancy
// limit (FluidSPHByCPUSystem.cs:69-75); here the cell's targets stay in registers -- a group of at
// most 32 per wave, 64 / group candidate slices each -- and the candidates of its 27 cells are
// streamed through the same image by LDS-DMA, every wave walking each image for its own targets.  (Round 2 read every
// candidate from global memory per lane: 14-19 ms per step on C3 while the column lands.)
//
// The candidates are ONE stream: the 27 fluid ranges, then the 27 boundary ranges, back to back
// (sh.plan holds the prefix table behind the halo table); an image is NB_GCHUNK consecutive stream
// positions, whatever ranges they come from, so a neighbourhood of 1500 candidates is two images
// and two barriers, not 54.  Slot q of an image is its records q S .. q S + S - 1, one per slice.
//
// Hit masks as in the planned path: the density pass stores, per group of targets (= one wave and
// pass), lane and 32 slots, one word of hit bits, image after image, in pool space it takes
// with one atomic per group (NbArgs::gmaskOff keeps the offset; no space left: no masks, the force
// pass walks every candidate of that group).  The force pass holds the words of 128 slots in four
// registers and walks their set bits; the next four are in flight meanwhile.
#define NB_GCHUNK (NB_CAP - 64) // candidate records per image; 64 FAR records follow them
#define NB_GNOMASK 0xffffffffu
#ifndef NB_GUNROLL
#define NB_GUNROLL 4 // slots per iteration of the density pass's walk (divides 32)
#endif
#ifndef NB_GSCATTER
#define NB_GSCATTER 1
#endif
#ifndef NB_GGROUPMIN
#define NB_GGROUPMIN NB_MAXWAVES
#endif
// shape of a crowded cell's work, the same in both passes and in the audit kernel
__device__ __forceinline__ void gcell_groups(int ncell, int &per, int &ngroups)
{
    const int ngroups0 = max(NB_GGROUPMIN, (ncell + 31) >> 5);
    per = (ncell + ngroups0 - 1) / ngroups0;
    ngroups = (ncell + per - 1) / per;
}
// mask words of one group (sfull slices) for a stream of sl records: per image, one word per 32 slots
__device__ __forceinline__ int gcell_group_words(int sl, int sfull)
{
    const int full = sl / NB_GCHUNK, last = sl - full * NB_GCHUNK;
    return full * ((div_small(NB_GCHUNK + sfull - 1, sfull) + 31) >> 5) + ((div_small(last + sfull - 1, sfull) + 31) >> 5);
}
// the stride K of the scattered order: next to Q / golden ratio, coprime to Q
__device__ __forceinline__ int gcell_stride(int nquads)
{
    int stride = max(1, (int)((float)nquads * 0.6180339887f));
    for (;; stride++) {
        int a = stride, b = nquads;
        while (b) { const int t = a % b; a = b; b = t; }
        if (a == 1) break;
    }
    return stride;
}
// place in the images -> place in the stream
__device__ __forceinline__ int gcell_stream_pos(int at, int stride, int nquads)
{
    return (int)((uint32_t)((at >> 2) * stride) % (uint32_t)nquads) * 4 + (at & 3);
}
// the stream's range table (one wave): rpre[r] = stream position of range r (r = 54: the end, then
// INT_MAX), rsrc[r] = its first record in posm (r < 27) or bposm.  Returns the stream's length.
__device__ __forceinline__ int gcell_range_table(const HaloTab &ht, int f, int lane, int *rpre, int *rsrc)
{
    int n = 0, s = 0;
    if (lane < 54) {
        const int c27 = lane < 27 ? lane : lane - 27;
        const int h = f + ((c27 / 9) - 1) * (NB_HX * NB_HY) + (((c27 / 3) % 3) - 1) * NB_HX + (c27 % 3) - 1;
        s = lane < 27 ? ht.fs[h] : ht.bs[h]; // fluid (index != j, :152), then boundary (:184-189)
        n = lane < 27 ? ht.fc[h] : ht.bc[h];
    }
    const int incl = wave_incl_scan(n);
    rpre[lane] = lane <= 54 ? incl - n : 0x7fffffff;
    rsrc[lane] = s;
    return __builtin_amdgcn_readlane(incl, 63);
}
// the last range that starts at or before stream position pos
__device__ __forceinline__ int gcell_range_of(const int *rpre, int pos)
{
    int r = 0;
#pragma unroll
    for (int step = 32; step; step >>= 1)
        if (rpre[r + step] <= pos) r += step;
    return r;
}

template <bool FORCE, bool COUNT>
__device__ void process_cell_chunked(NbShared<FORCE, COUNT> &sh, const HaloTab &ht, const NbArgs &A,
                                     uint32_t gi, int lx, int ly, int lz)
{
    constexpr int NT = FORCE ? NB_FTHREADS : NB_DTHREADS, NB_WAVES = NT / 64;
    constexpr int CHUNK = NB_GCHUNK;
    constexpr bool MASKS_OUT = !FORCE && !COUNT, MASKS_IN = FORCE && !COUNT;
    static_assert(sizeof(HaloTab) + 128 * sizeof(int) <= NB_PLANMAX, "range table behind the halo table");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int f = tile_halo_index(lx + 1, ly + 1, lz + 1);
    const int ncell = uniform_i(ht.fc[f]), cstart = uniform_i(ht.fs[f]);
    float4 *recP = sh.recP, *recV = sh.recV;
    // groups of at most 32 targets, at least as many as the force kernel has waves (a cell of 55
    // targets keeps eight waves busy with seven targets and nine slices each); the same groups in
    // both passes, whatever their workgroup size: the hit masks are per group
    int per, ngroups;
    gcell_groups(ncell, per, ngroups);
    const int sfull = slices_for(per); // (a last, smaller group has more slices and fewer slots)
    // the stream's range table; rsrc[60] = pool offset of the cell's hit masks
    int *rpre = (int *)(sh.plan + sizeof(HaloTab)), *rsrc = rpre + 64;
    if (w == 0) {
        const int rt = gcell_range_table(ht, f, lane, rpre, rsrc);
        // The pool space of all the cell's groups comes out of the workgroup's current piece (one
        // atomic per group -- a million returning atomics on one word per launch -- took 6 ms of
        // the density pass).  No space left: no masks, the force pass walks every candidate.
        uint32_t off = NB_GNOMASK;
        if (MASKS_OUT) {
            const uint32_t units = (uint32_t)(ngroups * gcell_group_words(((rt + 3) >> 2) * 4, sfull)) * 16u;
            if (lane == 0) {
                uint32_t at = sh.gpoolAt, end = sh.gpoolEnd;
                if (units > end - at) {
                    const uint32_t piece = max(units, (uint32_t)NB_GPIECE);
                    at = plan_pool_base(A.planGrid) + atomicAdd(&A.ctl[NB_CTL_POOL], piece);
                    end = at + piece;
                    // (the counter may run past the capacity: the sum stays far below 2^32)
                    if (at > A.poolCap || piece > A.poolCap - at) at = end = 0;
                }
                if (units <= end - at && units > 0) {
                    off = at;
                    at += units;
                }
                sh.gpoolAt = at;
                sh.gpoolEnd = end;
                A.gmaskOff[gi] = off;
            }
        } else if (MASKS_IN) {
            if (lane == 0 && A.useMasks) off = A.gmaskOff[gi];
        }
        if (lane == 0) rsrc[60] = (int)off;
    }
    lds_barrier();
    const int rtot = uniform_i(rpre[54]);
    // The stream is walked in a scattered order: quad (4 records, 64 bytes) q of the images is quad
    // q K mod Q of the stream, K next to Q / golden ratio and coprime to Q, so that any run of
    // slots holds records of all 27 cells in proportion and every target meets its hits at an
    // even rate -- the mask walk of the force pass lasts as long as its busiest lane.  (The
    // counting variants keep the stream's own order: they need the target's own place in it.)
    const int nquads = (rtot + 3) >> 2, slen = nquads * 4;
    const int stride = (!COUNT && NB_GSCATTER) ? gcell_stride(nquads) : 1;
    const int selfBase = uniform_i(rpre[13]) - cstart; // stream position of sorted index ti: ti + selfBase
    const uint32_t maskBase = (uint32_t)uniform_i(rsrc[60]), groupUnits = (uint32_t)gcell_group_words(slen, sfull) * 16u;
    for (int g0 = 0; g0 < ngroups; g0 += NB_WAVES) { // (workgroup-uniform)
        const int grp = g0 + w, t0 = grp * per;
        const int T = max(0, min(per, ncell - t0)); // wave-uniform; 0: the wave only helps to copy
        const int S = T > 0 ? slices_for(T) : 64;
        int u, slice;
        split_lane(lane, S, u, slice);
        const bool active = T > 0 && u < T;
        const int ti = cstart + (T > 0 ? t0 : 0) + (active ? u : 0);
        Pass<FORCE, COUNT> ps;
        float tkv, thr;
        const uint32_t mi = A.mat[ti];
        loop_constants<FORCE>(A.mats[mi], tkv, thr);
        const float4 fc4 = finish_constants<FORCE>(finish_row<FORCE, COUNT>(A.mats + mi), FORCE ? A.rhoIn[ti] : 0.f);
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
        // the group's mask words: one per image and 32 slots (wave-uniform count)
        uint32_t *maskW = nullptr; // + 64 per word
        if ((MASKS_OUT || MASKS_IN) && T > 0 && maskBase != NB_GNOMASK)
            maskW = (uint32_t *)(A.pool + maskBase + (uint32_t)grp * groupUnits) + lane;
        const bool walkMasks = MASKS_IN && maskW != nullptr; // wave-uniform
        const int self = ti + selfBase;                      // the target's own stream position
        for (int i0 = 0; i0 < slen; i0 += CHUNK) {
            const int cnt = min(CHUNK, slen - i0);
            lds_barrier(); // every wave is done with the image before
            for (int p0 = w * 64; p0 < cnt + 64; p0 += NT) { // wave-uniform
                const int k = p0 + lane;
                const int at = i0 + k; // place in the images -> place in the stream
                const int pos = gcell_stream_pos(at, stride, nquads);
                if (k < cnt && pos < rtot) {
                    const int r = gcell_range_of(rpre, pos);
                    const int src = rsrc[r] + pos - rpre[r];
                    const bool fluid = r < 27;
                    __builtin_amdgcn_global_load_lds((gptr_t)((fluid ? A.posm : A.bposm) + src), (lptr_t)(recP + p0), 16, 0, 0);
                    if (FORCE) {
                        const float4 *gv = fluid ? (const float4 *)A.velq + src : A.zeros;
                        __builtin_amdgcn_global_load_lds((gptr_t)gv, (lptr_t)(recV + p0), 16, 0, 0);
                    }
                } else if (k < cnt + 64) { // what the last slot's slices past the end read
                    recP[k] = make_float4(NB_FAR, NB_FAR, NB_FAR, 0.f);
                    if (FORCE) recV[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            const int nsl = T > 0 ? (int)(((float)(cnt + S - 1) + 0.5f) * __builtin_amdgcn_rcpf((float)S)) : 0;
            uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0; // the hit words of the next 128 slots
            auto fetch = [&](int q0) {                // (wave-uniform conditions)
                const int nw = (min(128, nsl - q0) + 31) >> 5;
                n0 = maskW[0];
                n1 = nw > 1 ? maskW[64] : 0u;
                n2 = nw > 2 ? maskW[128] : 0u;
                n3 = nw > 3 ? maskW[192] : 0u;
                maskW += nw * 64;
            };
            if (walkMasks && nsl > 0) fetch(0);
            __syncthreads(); // (waits for the DMA)
            if (T == 0) continue;
            const int selfK = self - i0;
            if (walkMasks) {
                for (int q0 = 0; q0 < nsl; q0 += 128) {
                    uint32_t w0 = __builtin_bitreverse32(n0), w1 = __builtin_bitreverse32(n1),
                             w2 = __builtin_bitreverse32(n2), w3 = __builtin_bitreverse32(n3);
                    if (q0 + 128 < nsl) fetch(q0 + 128);
                    int kb = q0 * S + slice; // record of bit 0 of w0
                    const int S32 = 32 * S;
                    while (__any((w0 | w1 | w2 | w3) != 0u)) {
                        if (w0 == 0u) {
                            w0 = w1; w1 = w2; w2 = w3; w3 = 0u;
                            kb += S32;
                        }
                        if (w0 != 0u) {
                            const int k = kb + __builtin_ctz(w0) * S;
                            w0 &= w0 - 1;
                            const float4 cp = recP[k];
                            const float4 cv = recV[FORCE ? k : 0];
                            keep_b128(cp);
                            keep_b128x(cv);
                            ps.pair(cp, cv, true);
                        }
                    }
                }
            } else {
                int q = 0; // slots walked in this image (wave-uniform)
                int k0 = 0;
                // NB_GUNROLL slots at a time, their LDS reads in flight together (a word of hit bits is
                // 32 slots: the flush test stays once per group)
                for (; !FORCE && k0 + NB_GUNROLL * S <= cnt; k0 += NB_GUNROLL * S) { // (density; the force pass walks masks)
                    float4 cp[NB_GUNROLL], cv[NB_GUNROLL];
#pragma unroll
                    for (int g = 0; g < NB_GUNROLL; g++) {
                        cp[g] = recP[k0 + g * S + slice];
                        cv[g] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (FORCE) cv[g] = recV[k0 + g * S + slice];
                    }
#pragma unroll
                    for (int g = 0; g < NB_GUNROLL; g++) {
                        keep_b128(cp[g]);
                        if (FORCE) keep_b128x(cv[g]);
                        ps.pair(cp[g], cv[g], k0 + g * S + slice != selfK);
                    }
                    if (MASKS_OUT) {
                        q += NB_GUNROLL;
                        if ((q & 31) == 0 && maskW) {
                            *maskW = ~ps.hits;
                            maskW += 64;
                        }
                    }
                }
                for (; k0 < cnt; k0 += S) {
                    const int k = k0 + slice;
                    const float4 cp = recP[k];
                    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (FORCE) cv = recV[k];
                    keep_b128(cp);
                    if (FORCE) keep_b128x(cv);
                    ps.pair(cp, cv, k != selfK);
                    if (MASKS_OUT) {
                        q++;
                        if ((q & 31) == 0 && maskW) {
                            *maskW = ~ps.hits;
                            maskW += 64;
                        }
                    }
                }
                if (MASKS_OUT && (q & 31) != 0 && maskW) { // slot 32 j + i in bit 31 - i
                    *maskW = (~ps.hits) << (32 - (q & 31));
                    maskW += 64;
                }
            }
        }
        const float4 F = ps.finish(A, S, slice, active && slice == 0, ti, fc4);
        if (FORCE && active && slice == 0) A.force[ti] = F;
    }
}

// Persistent workgroups.  Blocks are taken from the list in chunks of NB_CHUNK: workgroup b starts
// with chunk b, the chunks behind the first gridDim.x come from one device-wide counter (a single
// word serves about 88 returning atomics per microsecond on this chip, so one atomic per block
// would cost most of a millisecond per launch by itself); the next chunk is requested while the
// current one is being processed.  Then the crowded cells (process_cell_chunked), NB_GQUEUE per
// request, dealt the same way.
template <bool FORCE, bool COUNT, bool MULTI>
// second launch bound = waves per SIMD the register allocation must allow (what the LDS allows)
__global__ __launch_bounds__(FORCE ? NB_FTHREADS : NB_DTHREADS)
__attribute__((amdgpu_waves_per_eu((FORCE ? NB_FLB : NB_DLB) - (COUNT ? 1 : 0), (FORCE ? NB_FLB : NB_DLB) - (COUNT ? 1 : 0))))
void k_neighbors(NbArgs A)
{
    __shared__ NbShared<FORCE, COUNT> sh;
    const int tid = threadIdx.x;
    for (int i = tid; i < NB_MTAB && i < A.nmats; i += blockDim.x) {
        const Mat *m = A.mats + i;
        sh.mtab[i] = make_float4(m->negInvH2, m->negInvH, m->h2, m->thrF);
        sh.mfin[i] = finish_row<FORCE, COUNT>(m);
    }
    if (!FORCE && !COUNT && tid == 0) sh.gpoolAt = sh.gpoolEnd = 0;
    const uint32_t nblocks = min(plan_list_base(A.ctl[NB_CTL_TILES], A.planGrid) + A.ctl[NB_CTL_BLOCKS], (uint32_t)A.blockCap);
    // blocks per request: NB_CHUNK when there is plenty of work, fewer when the whole list is only
    // a few blocks per workgroup (small scenes: the work must spread over all the workgroups)
    const uint32_t chunk = min((uint32_t)NB_CHUNK, max(1u, nblocks / (4u * gridDim.x)));
    uint32_t pending = blockIdx.x * chunk;
    for (;;) {
        if (tid == 0) sh.next = pending;
        lds_barrier();
        const uint32_t first = sh.next;
        if (first >= nblocks) break;
        const uint32_t n = min(chunk, nblocks - first);
        if (tid < (int)n) sh.hdr[tid] = A.blockList[first + tid];
        lds_barrier();
        // every thread has read `next` before thread 0 can write it again: each block has barriers.
        // The next chunk is requested inside the first block that is not an unused list entry.
        bool requested = false;
        for (uint32_t i = 0; i < n; i++) {
            const BlockHdr hdr = sh.hdr[i];
            if (hdr.w == 0) continue; // unused list entry
            exec_block<FORCE, COUNT, MULTI>(sh, A, hdr, !requested, chunk, pending);
            requested = true;
        }
        if (!requested && tid == 0) pending = atomicAdd(&A.ctl[A.queue], chunk) + gridDim.x * chunk;
    }
    // cells whose 27-cell neighbourhood overflows the LDS image
    const uint32_t ncells = min(A.ctl[NB_CTL_GCELLS], (uint32_t)A.gcellCap);
    HaloTab &ht = *(HaloTab *)sh.plan;
    // (NB_GQUEUE cells per request, aligned; the place in the list lives in LDS, not in registers)
    if (tid == 0) sh.gnext = blockIdx.x * NB_GQUEUE;
    for (;;) {
        lds_barrier();
        const uint32_t i = sh.gnext;
        if (i >= ncells) break;
        const int2 gc = A.gcellList[i];
        if (gc.x >= 0) { // (else: unused list entry)
            tile_lookup(ht, gc.x, A);
            process_cell_chunked<FORCE, COUNT>(sh, ht, A, i, gc.y & 0xff, (gc.y >> 8) & 0xff, (gc.y >> 16) & 0xff);
        }
        __syncthreads(); // every thread has read `next`
        if (tid == 0) {
            uint32_t n = i + 1;
            if (n % NB_GQUEUE == 0) n = (atomicAdd(&A.ctl[A.queue + 1], 1u) + gridDim.x) * NB_GQUEUE;
            sh.gnext = n;
        }
    }
}

// ------------------------------------------------------------------------------------------
// PARITY AUDIT of the production neighbour set (tests only; no reference counterpart).
// The production density pass decides with the sign of fma(d2, -1/h2, 1) which candidates the
// force pass will see, and leaves that decision as hit-mask words behind every block's plan.
// This kernel walks plan and masks the way the force pass does and gives, per target (sorted
// order), the number of hits and the sum of the hits' ids -- fluid: ECS id, boundary: bndBase +
// index in the caller's boundary array -- so that a test can compare the SET with the reference's
// predicate d2 < h2 (FluidSPHByCPUSystem.cs:97).  One wave per block.  invalid counts bits that
// point outside the block's image (there must be none).
__global__ __launch_bounds__(64) void k_audit_masks(NbArgs A, const uint32_t *__restrict__ oid,
                                                    const uint32_t *__restrict__ bperm, uint32_t bndBase,
                                                    int *__restrict__ counts,
                                                    unsigned long long *__restrict__ idsum,
                                                    uint32_t *__restrict__ invalid)
{
    const int lane = threadIdx.x;
    const uint32_t nblocks = min(plan_list_base(A.ctl[NB_CTL_TILES], A.planGrid) + A.ctl[NB_CTL_BLOCKS], (uint32_t)A.blockCap);
    for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const BlockHdr hdr = A.blockList[b];
        if (hdr.w == 0) continue;
        const int nh = hdr.z & 0xff, nunits = (hdr.z >> 8) & 0xff;
        const int total = hdr.w & 0x7ff, nslots = (hdr.w >> 20) & 0xfff;
        const int4 *hinfo = (const int4 *)(A.pool + hdr.x);
        const UnitDesc *units = (const UnitDesc *)(hinfo + nh);
        const uint32_t *slotTab = (const uint32_t *)(units + nunits);
        const int planBytes = nh * 16 + nunits * 32 + nslots * 4;
        const uint32_t *maskBlk = (const uint32_t *)(A.pool + hdr.x + (planBytes >> 4));
        for (int unit = 0; unit < nunits; unit++) {
            const UnitDesc ud = units[unit];
            const int T = ud.tS & 0xff, S = ud.tS >> 8, nA = ud.nAB & 0xffff, nB = ud.nAB >> 16;
            int u, s;
            split_lane(lane, S, u, s);
            if (u >= T) continue;
            int cnt = 0;
            unsigned long long sum = 0;
            const int nwords = (nA + nB + 31) >> 5;
            for (int j = 0; j < nwords; j++) {
                uint32_t m = __builtin_bitreverse32(maskBlk[(size_t)(ud.maskw + j) * 64 + lane]);
                while (m) {
                    const int k = 32 * j + __builtin_ctz(m);
                    m &= m - 1;
                    int rec = -1;
                    if (k < nA + nB) {
                        const uint32_t e = slotTab[ud.slotA + k];
                        if (k < nA || s < (int)(e >> 16)) rec = (int)((e & 0xffffu) >> 4) + s;
                    }
                    if (rec < 0 || rec >= total) {
                        atomicAdd(invalid, 1u);
                        continue;
                    }
                    int lo = 0, hi = nh - 1; // the last halo cell that starts at or before rec
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (hinfo[mid].x <= rec) lo = mid;
                        else hi = mid - 1;
                    }
                    const int4 hi4 = hinfo[lo];
                    const int k2 = rec - hi4.x, fc = hi4.z & 0xffff, bc = hi4.z >> 16;
                    if (k2 < fc) sum += oid[hi4.y + k2] & 0x3fffffffu;
                    else if (k2 < fc + bc) sum += (unsigned long long)bndBase + bperm[hi4.w + (k2 - fc)];
                    else atomicAdd(invalid, 1u);
                    cnt++;
                }
            }
            atomicAdd(&counts[ud.gi0 + u], cnt);
            atomicAdd(&idsum[ud.gi0 + u], sum);
        }
    }
}

// The same for the crowded cells (process_cell_chunked): their masks are per group of targets, image
// and 32 slots, and a slot's records are found through the scattered order of the stream.  One
// wave per cell.  A cell whose masks found no room in the pool has none: count -1.
__global__ __launch_bounds__(64) void k_audit_gcells(NbArgs A, const uint32_t *__restrict__ oid,
                                                     const uint32_t *__restrict__ bperm, uint32_t bndBase,
                                                     int *__restrict__ counts,
                                                     unsigned long long *__restrict__ idsum,
                                                     uint32_t *__restrict__ invalid)
{
    __shared__ HaloTab ht;
    __shared__ int rpre[64], rsrc[64];
    const int lane = threadIdx.x;
    const uint32_t ncells = min(A.ctl[NB_CTL_GCELLS], (uint32_t)A.gcellCap);
    for (uint32_t i = blockIdx.x; i < ncells; i += gridDim.x) {
        const int2 gc = A.gcellList[i];
        if (gc.x < 0) continue; // unused list entry
        __syncthreads();
        tile_lookup(ht, gc.x, A);
        const int f = tile_halo_index((gc.y & 0xff) + 1, ((gc.y >> 8) & 0xff) + 1, ((gc.y >> 16) & 0xff) + 1);
        const int ncell = ht.fc[f], cstart = ht.fs[f];
        const int rtot = gcell_range_table(ht, f, lane, rpre, rsrc);
        __syncthreads();
        const uint32_t maskBase = A.gmaskOff[i];
        if (maskBase == NB_GNOMASK) {
            for (int t = lane; t < ncell; t += 64) counts[cstart + t] = -1;
            continue;
        }
        int per, ngroups;
        gcell_groups(ncell, per, ngroups);
        const int sfull = slices_for(per);
        const int nquads = (rtot + 3) >> 2, slen = nquads * 4;
        const int stride = NB_GSCATTER ? gcell_stride(nquads) : 1;
        const uint32_t groupUnits = (uint32_t)gcell_group_words(slen, sfull) * 16u;
        for (int grp = 0; grp < ngroups; grp++) {
            const int t0 = grp * per, T = min(per, ncell - t0), S = slices_for(T);
            int u, slice;
            split_lane(lane, S, u, slice);
            if (u >= T) continue;
            const uint32_t *maskW = (const uint32_t *)(A.pool + maskBase + (uint32_t)grp * groupUnits) + lane;
            int cnt = 0;
            unsigned long long sum = 0;
            for (int i0 = 0; i0 < slen; i0 += NB_GCHUNK) {
                const int n = min(NB_GCHUNK, slen - i0), nsl = div_small(n + S - 1, S);
                for (int j = 0; j * 32 < nsl; j++, maskW += 64) {
                    uint32_t m = __builtin_bitreverse32(*maskW);
                    while (m) {
                        const int k = (32 * j + __builtin_ctz(m)) * S + slice;
                        m &= m - 1;
                        const int pos = gcell_stream_pos(i0 + k, stride, nquads);
                        if (k >= n || pos >= rtot) { // (a FAR record: no hit can be there)
                            atomicAdd(invalid, 1u);
                            continue;
                        }
                        const int r = gcell_range_of(rpre, pos), src = rsrc[r] + pos - rpre[r];
                        if (r < 27) sum += oid[src] & 0x3fffffffu;
                        else sum += (unsigned long long)bndBase + bperm[src];
                        cnt++;
                    }
                }
            }
            atomicAdd(&counts[cstart + t0 + u], cnt);
            atomicAdd(&idsum[cstart + t0 + u], sum);
        }
    }
}
