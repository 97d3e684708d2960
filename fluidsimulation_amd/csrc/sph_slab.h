// sph_slab.h -- slab decomposition across GPUs (the reference is single-device).
// Each rank owns the fluid particles with x in [lo, hi).  Once per step it packs, for each x
// neighbour, the particles that left the slab (ownership transfer) and the particles within
// `halo` of the shared face (ghost copies), and appends what it receives.  Ghosts live for one
// step: they take part in the sort and in both neighbour passes, are integrated like everything
// else and are dropped by the next step's sort (key = cells, the largest).
#pragma once
#include "sph_device.h"

#define SPH_ID_GHOST 0x80000000u // copy of a particle another rank owns
#define SPH_ID_DEAD 0x40000000u  // ghost of a finished step: sorted to the end and cut off
#define SPH_ID_MASK 0x3fffffffu

// 32-byte exchange record = two float4: (x, y, z, bits(material)), (vx, vy, vz, bits(id | flags))

// appends to side[0] (x < lo + halo) / side[1] (x >= hi - halo); counters: records left, right,
// transfers left, right.  One atomic per wave and side.  COUNT_ONLY: the counters alone, nothing
// is written and no record changes state (sph_slab_pack sizes the call with it, so that a buffer
// that is too small is reported before anything has happened).
template <bool COUNT_ONLY>
__global__ __launch_bounds__(256) void k_slab_pack(
    const float4 *__restrict__ posm, const float4 *__restrict__ velq,
    const uint32_t *__restrict__ mat, uint32_t *__restrict__ oid, int n, float lo, float hi,
    float halo, float4 *__restrict__ left, float4 *__restrict__ right, int cap,
    uint32_t *__restrict__ counters,
    uint32_t *__restrict__ keys /* nullable, not COUNT_ONLY: this step's cell keys (saves k_slab_keys) */,
    GridDesc g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool sendL = false, sendR = false, moveL = false, moveR = false;
    uint32_t id = 0;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        id = oid[i];
        if (id & SPH_ID_GHOST) {
            if (!COUNT_ONLY && !(id & SPH_ID_DEAD)) oid[i] = id | SPH_ID_DEAD;
            if (!COUNT_ONLY && keys) keys[i] = (uint32_t)g.cells; // dead: sorted to the end
        } else {
            p = posm[i];
            moveL = p.x < lo;
            moveR = p.x >= hi;
            sendL = p.x < lo + halo;
            sendR = p.x >= hi - halo;
            if (!COUNT_ONLY && (moveL || moveR)) oid[i] = id | SPH_ID_GHOST; // this step's ghost
            if (!COUNT_ONLY && keys)
                keys[i] = dense_key(quantize1(p.x, g.cellSize), quantize1(p.y, g.cellSize),
                                    quantize1(p.z, g.cellSize), g);
        }
    }
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const bool send = side ? sendR : sendL, move = side ? moveR : moveL;
        const unsigned long long m = __ballot(send);
        if (m == 0) continue; // wave-uniform
        const unsigned long long mm = __ballot(move);
        uint32_t base = 0;
        if (lane_id() == 0) {
            base = atomicAdd(&counters[side], (uint32_t)__popcll(m));
            if (mm) atomicAdd(&counters[2 + side], (uint32_t)__popcll(mm));
        }
        base = (uint32_t)bcast_i((int)base, 0);
        if (!COUNT_ONLY && send) {
            const uint32_t slot = base + mbcnt64(m);
            if (slot < (uint32_t)cap) {
                float4 *dst = (side ? right : left) + 2 * (size_t)slot;
                const float4 v = velq[i];
                dst[0] = make_float4(p.x, p.y, p.z, __uint_as_float(mat[i]));
                dst[1] = make_float4(v.x, v.y, v.z, __uint_as_float(move ? id : (id | SPH_ID_GHOST)));
            }
        }
    }
}

__global__ void k_slab_import(const float4 *__restrict__ recs, int count,
                              const Mat *__restrict__ mats, int nmats, float4 *__restrict__ posm,
                              float4 *__restrict__ velq, uint32_t *__restrict__ mat,
                              uint32_t *__restrict__ oid, uint32_t *__restrict__ keys, GridDesc g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4 a = recs[2 * (size_t)i], b = recs[2 * (size_t)i + 1];
    uint32_t m = __float_as_uint(a.w);
    if (m >= (uint32_t)nmats) m = 0; // never indexes outside the table
    posm[i] = make_float4(a.x, a.y, a.z, mats[m].mass);
    velq[i] = make_float4(b.x, b.y, b.z, 0.f);
    mat[i] = m;
    oid[i] = __float_as_uint(b.w) & ~SPH_ID_DEAD;
    keys[i] = dense_key(quantize1(a.x, g.cellSize), quantize1(a.y, g.cellSize),
                        quantize1(a.z, g.cellSize), g);
}

// upload with caller-given ids and material indices
__global__ void k_slab_pack_fluid(const float *__restrict__ pos3, const float *__restrict__ vel3,
                                  const uint32_t *__restrict__ matIn,
                                  const uint32_t *__restrict__ ids, const Mat *__restrict__ mats,
                                  int n, float4 *__restrict__ posm, float4 *__restrict__ velq,
                                  uint32_t *__restrict__ mat, uint32_t *__restrict__ oid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t m = matIn[i];
    posm[i] = make_float4(pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2], mats[m].mass);
    velq[i] = make_float4(vel3[3 * i], vel3[3 * i + 1], vel3[3 * i + 2], 0.f);
    mat[i] = m;
    oid[i] = ids[i] & SPH_ID_MASK;
}

// cell keys with the dead records sent to the end of the sort (a step that was not preceded by a
// pack: the pack and import kernels write the same keys, they hold the positions)
__global__ void k_slab_keys(const float4 *__restrict__ posm, const uint32_t *__restrict__ oid,
                            int n, GridDesc g, uint32_t *__restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = posm[i];
    const uint32_t k = dense_key(quantize1(p.x, g.cellSize), quantize1(p.y, g.cellSize),
                                 quantize1(p.z, g.cellSize), g);
    keys[i] = (oid[i] & SPH_ID_DEAD) ? (uint32_t)g.cells : k;
}
