// sph_device.h -- device-side data layout shared by the gfx950 kernels of libsphfluid.so.
//
// Data layout in HBM (DESIGN.md "Data layout"): the fluid state lives in *sorted order* (stable by
// dense cell key, ties by ECS id) and stays there from step to step:
//   posm[N]  float4  x, y, z, mass (= 1f / InverseMass, FluidSPHByCPUSystem.cs:54,165)
//   velq[N]  float4  vx, vy, vz, q = p / (rho*rho)   (q written by the density kernel)
//   mat[N]   u32     index into the material table (per-particle constants, deduplicated)
//   oid[N]   u32     ECS id of the particle in this slot (= the sorted permutation)
//   rho[N], prs[N] f32, force[N] float4
// Boundary particles are static: sorted once into bposm[B] (x, y, z, mass) with their own table.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPH_WAVE 64

// Periodic cell table (see SphGrid in include/sphfluid.h).  Cells are grouped into tiles of
// SPH_TX x SPH_TY x SPH_TZ cells; the dense key is  tile * SPH_TCELLS + local  with the tile index
// x-fastest over tiles and the local index x-fastest inside the tile, so that after the sort the
// particles of one tile are contiguous (one workgroup of the neighbour kernels = one tile) and
// the x-runs of a tile are contiguous too.  gx, gy, gz are multiples of the tile dimensions.
#define SPH_TX 4
#define SPH_TY 4
#define SPH_TZ 2
#define SPH_TCELLS (SPH_TX * SPH_TY * SPH_TZ)
struct GridDesc {
    int ox, oy, oz;
    int gx, gy, gz;
    float cellSize;
    int cells;
    int ntx, nty, ntz, tiles;
};

// Per-material constants, computed once on the host in float32 in the reference's evaluation
// order (FluidSPHByCPUSystem.cs:53-60,133-135), so that the device and the oracle use identical
// thresholds:  h = radius*rate, h2 = h*h (== pow(h,2)),
//   thrF = the least float t with sqrt_rn(t) >= h, so that  length(v) < h  <=>  dot(v,v) < thrF
//   (the force pass's predicate, :156,219, without a per-pair correctly rounded sqrt).
struct Mat {
    float h, h2, thrF, poly6;      // poly6 = mass*315/(64*PI*h^9)
    float restDensity, gasConstant, viscosity, mass;
    float invMass, radius, spiky;  // spiky = 45/(PI*h^6)
    float negInvH;                 // -1/h, -1/h2, poly6*h2^3: per-target constants of the
    float negInvH2, poly6h6;       // neighbour kernels, divided once on the host
    float pad[2];
};

// Box collider in the form the integrate kernel reads (row-major rotation, world = R*local)
#define SPH_BOX_CHUNK 32 // boxes per broadphase chunk (consecutive indices)
#define SPH_MAX_SUBSTEPS 256 // most pieces a fast particle's step is taken in (k_apply_integrate)

struct DevBox {
    float c[3];
    float R[9];
    float he[3];
    float lv[3];
    float av[3];
    float invMass;
    float invI[3];
    float restitution;
    int dynamic;
    float amin[3], amax[3]; // world AABB of the box (k_bodies): broadphase of k_apply_integrate
};

__device__ __forceinline__ int wrap_cell(int c, int g)
{
    int m = c % g;
    return m < 0 ? m + g : m;
}

// FluidSPHUtils.Quantize, :236-243: (int)floor(p / size), true IEEE division
__device__ __forceinline__ int quantize1(float p, float size)
{
    return (int)floorf(p / size);
}

// FluidSPHUtils.Hash, :246-261: FNV-1 over the 12 little-endian bytes of (x,y,z)
__device__ __forceinline__ uint32_t fnv1_cell(int x, int y, int z)
{
    uint32_t hash = 2166136261u;
    const int p[3] = {x, y, z};
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            hash *= 16777619u;
            hash ^= (uint32_t)(p[i] >> (8 * j)) & 0xffu;
        }
    }
    return hash;
}

// key of the cell with wrapped, origin-shifted coordinates 0 <= x < gx etc.
__device__ __forceinline__ uint32_t tiled_key(int x, int y, int z, const GridDesc &g)
{
    const int tile = ((z / SPH_TZ) * g.nty + (y / SPH_TY)) * g.ntx + (x / SPH_TX);
    const int local = ((z % SPH_TZ) * SPH_TY + (y % SPH_TY)) * SPH_TX + (x % SPH_TX);
    return (uint32_t)(tile * SPH_TCELLS + local);
}

__device__ __forceinline__ uint32_t dense_key(int cx, int cy, int cz, const GridDesc &g)
{
    return tiled_key(wrap_cell(cx - g.ox, g.gx), wrap_cell(cy - g.oy, g.gy),
                     wrap_cell(cz - g.oz, g.gz), g);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// popcount of the set bits of mask below this lane
__device__ __forceinline__ int mbcnt64(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// wave-uniform broadcast of lane `src` (src must be wave-uniform)
__device__ __forceinline__ float bcast_f(float v, int src)
{
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
