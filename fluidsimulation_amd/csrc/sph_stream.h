// sph_stream.h -- the HBM-streaming kernels: pack/unpack at the ABI, cell keys, bounds,
// reorder into sorted SoA, ApplyForce + integration + collider response, rigid bodies.
#pragma once
#include "sph_device.h"

// ---- ABI packing ------------------------------------------------------------------------------
// host layout (tightly packed float3 / SphParticle arrays, FluidSPHByGPUSystem.cs:164-181) ->
// device SoA in ECS order
__global__ void k_pack_fluid(const float *__restrict__ pos3, const float *__restrict__ vel3,
                             const uint32_t *__restrict__ matIn, const Mat *__restrict__ mats,
                             int n, float4 *__restrict__ posm, float4 *__restrict__ velq,
                             uint32_t *__restrict__ mat, uint32_t *__restrict__ oid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t m = matIn[i];
    posm[i] = make_float4(pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2], mats[m].mass);
    velq[i] = make_float4(vel3[3 * i], vel3[3 * i + 1], vel3[3 * i + 2], 0.f);
    mat[i] = m;
    oid[i] = (uint32_t)i;
}

// FluidInitializeSystem.CreateParticles (FluidInitializeSystem.cs:63-92) on the device: lattice
// point (ix,iy,iz), x fastest, local = i*2*rad + bounds.Min + rad (:84-86), world =
// MultiplyPoint3x4(localToWorld, local).  l2w is 3x4 row-major.
struct SeedDesc {
    int nx, ny, nz;
    float rad[3], bmin[3], l2w[12];
};
__global__ void k_seed_lattice(SeedDesc d, float mass, uint32_t matIdx, uint32_t idBase, int count,
                               float4 *__restrict__ posm, float4 *__restrict__ velq,
                               uint32_t *__restrict__ mat, uint32_t *__restrict__ oid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int ix = i % d.nx, iy = (i / d.nx) % d.ny, iz = i / (d.nx * d.ny);
    const float two = 2.0f;
    const float lx = (float)ix * two * d.rad[0] + d.bmin[0] + d.rad[0];
    const float ly = (float)iy * two * d.rad[1] + d.bmin[1] + d.rad[1];
    const float lz = (float)iz * two * d.rad[2] + d.bmin[2] + d.rad[2];
    float w[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
        w[r] = (d.l2w[4 * r] * lx + d.l2w[4 * r + 1] * ly) + d.l2w[4 * r + 2] * lz + d.l2w[4 * r + 3];
    posm[i] = make_float4(w[0], w[1], w[2], mass);
    velq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    mat[i] = matIdx;
    oid[i] = idBase + (uint32_t)i;
}

__global__ void k_pack_boundary(const float *__restrict__ pos3, const float *__restrict__ bparams,
                                int n, float4 *__restrict__ bposm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // BoundaryParticleComponent: radius, mass, restDensity (only mass enters the force, :184-189)
    bposm[i] = make_float4(pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2], bparams[3 * i + 1]);
}

// sorted slot -> ECS order, float3-packed (for sph_download / sph_read)
__global__ void k_unpack3(const float4 *__restrict__ src, const uint32_t *__restrict__ oid, int n,
                          float *__restrict__ out3)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float4 v = src[s];
    const size_t o = 3 * (size_t)oid[s];
    out3[o] = v.x;
    out3[o + 1] = v.y;
    out3[o + 2] = v.z;
}
template <typename T>
__global__ void k_unpack1(const T *__restrict__ src, const uint32_t *__restrict__ oid, int n,
                          T *__restrict__ out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    out[oid[s]] = src[s];
}

// ---- HashPosition (FluidSolver.compute:28-45; CPU: FluidSPHUtils.cs:69,236-261) --------------
// dense key for the sort; with HASH also the reference's FNV-1 hash and quantised cell (parity
// outputs, written in ECS order through oid; oid == nullptr means identity).
template <bool HASH>
__global__ void k_keys(const float4 *__restrict__ posm, const uint32_t *__restrict__ oid, int n,
                       GridDesc g, uint32_t *__restrict__ keys, uint32_t *__restrict__ hashes,
                       int *__restrict__ cells, uint32_t *__restrict__ keysEcs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = posm[i];
    const int cx = quantize1(p.x, g.cellSize), cy = quantize1(p.y, g.cellSize),
              cz = quantize1(p.z, g.cellSize);
    const uint32_t k = dense_key(cx, cy, cz, g);
    keys[i] = k;
    if (HASH) {
        const size_t o = oid ? oid[i] : (uint32_t)i;
        hashes[o] = fnv1_cell(cx, cy, cz);
        cells[3 * o] = cx;
        cells[3 * o + 1] = cy;
        cells[3 * o + 2] = cz;
        keysEcs[o] = k;
    }
}

// ---- FindBoundsJob min/max (FluidSPHUtils.cs:59-80) -------------------------------------------
// two-level reduction; partial[6*block + {0..2}] = min xyz, {3..5} = max xyz
__global__ __launch_bounds__(256) void k_bounds_partial(const float4 *__restrict__ p, int n,
                                                        float *__restrict__ partial)
{
    __shared__ float sm[6][4];
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 v = p[i];
        mn[0] = fminf(mn[0], v.x); mn[1] = fminf(mn[1], v.y); mn[2] = fminf(mn[2], v.z);
        mx[0] = fmaxf(mx[0], v.x); mx[1] = fmaxf(mx[1], v.y); mx[2] = fmaxf(mx[2], v.z);
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64));
        }
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0)
        for (int a = 0; a < 3; a++) {
            sm[a][w] = mn[a];
            sm[3 + a][w] = mx[a];
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sm[threadIdx.x][0];
        for (int i = 1; i < 4; i++)
            v = threadIdx.x < 3 ? fminf(v, sm[threadIdx.x][i]) : fmaxf(v, sm[threadIdx.x][i]);
        partial[6 * blockIdx.x + threadIdx.x] = v;
    }
}

// ---- reorder: gather the state into the new sorted order --------------------------------------
// perm[s] = index in the current arrays of the particle that belongs in sorted slot s (stable in
// the CURRENT order).  Ties inside a cell are re-ranked by ECS id so that the resident order is
// always exactly "stable by (cell key, particle id)" whatever the previous order was.
__global__ __launch_bounds__(256) void k_reorder(
    const uint32_t *__restrict__ perm, const uint32_t *__restrict__ skeys,
    const int2 *__restrict__ table, int n, const float4 *__restrict__ posmIn,
    const float4 *__restrict__ velqIn, const uint32_t *__restrict__ matIn,
    const uint32_t *__restrict__ oidIn, float4 *__restrict__ posmOut,
    float4 *__restrict__ velqOut, uint32_t *__restrict__ matOut, uint32_t *__restrict__ oidOut)
{
    __shared__ uint32_t ids[256];
    const int blockBase = blockIdx.x * 256;
    const int s = blockBase + threadIdx.x;
    uint32_t c = 0, id = 0, key = 0;
    if (s < n) {
        c = perm[s];
        key = skeys[s];
        id = oidIn[c];
    }
    // the record itself is fetched now and travels in registers while the rank is worked out (four
    // gathers in flight instead of one dependent load after the other)
    float4 pm = make_float4(0.f, 0.f, 0.f, 0.f), vq = pm;
    uint32_t mt = 0;
    if (s < n) {
        pm = posmIn[c];
        vq = velqIn[c];
        if (matIn) mt = matIn[c]; // (null: one material, both index buffers hold zeros)
    }
    ids[threadIdx.x] = id;
    __syncthreads();
    if (s >= n) return;
    const int2 cell = table[key];
    const uint32_t me = id & 0x3fffffffu; // bits 30/31: slab flags (sph_slab.h)
    int rank = 0;
    // cell mates inside this block: their ids are in LDS
    const int lo = max(cell.x, blockBase), hi = min(cell.y, blockBase + 256);
    for (int t = lo; t < hi; t++) rank += (ids[t - blockBase] & 0x3fffffffu) < me; // (never < itself)
    // cell mates outside it (a cell that straddles a block boundary; every mate but 256 of a
    // crowded cell): two dependent loads each, eight in flight at a time -- one after the other
    // they cost a crowded step 3 ms (hundreds of particles per cell while a deep column lands)
    auto outside = [&](int a, int b) {
        for (int t0 = a; t0 < b; t0 += 8) {
            uint32_t pv[8], iv[8];
#pragma unroll
            for (int q = 0; q < 8; q++) pv[q] = perm[min(t0 + q, b - 1)];
#pragma unroll
            for (int q = 0; q < 8; q++) iv[q] = oidIn[pv[q]];
#pragma unroll
            for (int q = 0; q < 8; q++) rank += (t0 + q < b && (iv[q] & 0x3fffffffu) < me) ? 1 : 0;
        }
    };
    outside(cell.x, min(cell.y, blockBase));
    outside(max(cell.x, blockBase + 256), cell.y);
    const int dst = cell.x + rank;
    posmOut[dst] = pm;
    velqOut[dst] = vq;
    if (matIn) matOut[dst] = mt;
    oidOut[dst] = id;
}

// boundary: plain gather (the input order is the id order, so the stable sort needs no re-rank)
__global__ void k_gather4(const uint32_t *__restrict__ perm, int n,
                          const float4 *__restrict__ in, float4 *__restrict__ out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) out[s] = in[perm[s]];
}

// ---- ApplyForceJob (FluidSPHUtils.cs:105-124) + row I (integration, colliders) ----------------
__device__ __forceinline__ float dot3f(const float *a, const float *b)
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

// SLABCOUNT: also count what the next slab exchange will send (its own instantiation: the plain
// resident loop must not pay for the arguments)
template <bool APPLY, bool INTEGRATE, bool SLABCOUNT>
__global__ __launch_bounds__(256) void k_apply_integrate(
    float4 *__restrict__ posm, float4 *__restrict__ velq, const float4 *__restrict__ force,
    const float *__restrict__ rho, const uint32_t *__restrict__ mat,
    const Mat *__restrict__ mats, int n, float dt, const DevBox *__restrict__ boxes, int nboxes,
    float particleRestitution, float *__restrict__ bodyImpulse,
    const float4 *__restrict__ boxAabb /* 2 per box: min, max */,
    const float4 *__restrict__ chunkAabb /* 2 per chunk of SPH_BOX_CHUNK boxes */,
    const uint32_t *__restrict__ slabFlags /* slab mode: ids with the ghost bit, else nullptr */,
    uint32_t *__restrict__ nextKeys /* nullable: the next step's cell keys (saves k_keys) */,
    uint32_t *__restrict__ slabNext /* nullable, slab mode: the counters of the NEXT exchange (records for the
                                       left / right neighbour, transfers left / right) if its cuts are ... */,
    float slabLo, float slabHi, float slabHalo /* ... these: saves the counting pass of k_slab_pack */,
    GridDesc grid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), vv = pv;
    Mat m = mats[0];
    bool owned = true; // a ghost's push on a body is counted by the rank that owns the particle
    if (live) {
        pv = posm[i];
        vv = velq[i];
        if (mat) m = mats[mat[i]]; // (null: one material)
        if (slabFlags) owned = (slabFlags[i] & 0x80000000u) == 0;
    }
    float x[3] = {pv.x, pv.y, pv.z}, v[3] = {vv.x, vv.y, vv.z};
    if (APPLY && live) {
        const float d = rho[i];
        if (d != 0.f) {
            const float4 f = force[i];
            const float ff[3] = {f.x, f.y, f.z};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                // velocity.ApplyLinearImpulse(mass, F / rho / invM * dt): Linear += impulse * invM
                const float impulse = ff[a] / d / m.invMass * dt;
                v[a] += impulse * m.invMass;
            }
        }
    }
    if (INTEGRATE) {
        const float r = m.radius;
        // Swept: a particle that moves farther than half its radius takes
        // the step in pieces no longer than r/2 and meets the boxes after each, so that it is never
        // carried through a box at least as thick as its radius (a centre that starts outside, or
        // on a face, is then less than half the thickness deep: the nearest face is the one it
        // came in through).  A piece ends at xb + v (t - tb), (xb, tb) = where and when the
        // particle last touched a box: without a contact the end position is x + v dt exactly.
        // The wave runs as many rounds as its lane with the most pieces; a lane takes part in its
        // own pieces only, so every particle sees the same sequence of tests as in a loop of its
        // own (the oracle's).
        float xb[3] = {x[0], x[1], x[2]}, tb = 0.f;
        float xprev[3] = {x[0], x[1], x[2]}; // end of the piece before (start of the step at first)
        // pieces: the rest of the step is cut into ceil(|v| rest / (r/2)) equal parts and the first
        // is taken; after every piece the cut is made afresh with the velocity the particle then has
        // (a moving box can hand a slow particle a large one in the middle of a step)
        const float piece = 0.5f * r;
        float tnow = 0.f;
        auto next_time = [&](int npieces) -> float {
            float ts = dt;
            if (npieces < SPH_MAX_SUBSTEPS - 1) {
                const float rest = dt - tnow, sp = sqrtf(dot3f(v, v)) * rest;
                if (sp > piece) { // (never more parts than pieces are left: then they are longer)
                    const float mm = fminf(ceilf(sp / piece), (float)(SPH_MAX_SUBSTEPS - npieces));
                    if (mm > 1.f) ts = tnow + rest / mm;
                }
            }
            return ts;
        };
        bool several = nboxes > 0 && live && next_time(0) < dt; // more than one piece to take
        int mw = __any(several) ? 2 : 1;
#pragma unroll
        for (int a = 0; a < 3; a++) x[a] = xb[a] + v[a] * dt;
        // Broadphase: the particles of a wave are neighbours in the sorted order (a few cells), so
        // the wave's own bounds (spheres included) are tested against each box's world AABB with
        // scalar compares and the whole wave skips a box it cannot touch.  Exact: a box that is
        // skipped would have changed nothing; the bounds are taken again after any push-out.
        float wmin[3], wmax[3];
        auto wave_bounds = [&](bool act, const float *lo3, const float *hi3) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
                float lo = act ? lo3[a] - r : 3.0e38f, hi = act ? hi3[a] + r : -3.0e38f;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    lo = fminf(lo, __shfl_xor(lo, d, 64));
                    hi = fmaxf(hi, __shfl_xor(hi, d, 64));
                }
                wmin[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lo)));
                wmax[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hi)));
            }
        };
        auto misses = [&](const float4 lo, const float4 hi) { // wave-uniform (scalar operands)
            return wmax[0] < lo.x || wmin[0] > hi.x || wmax[1] < lo.y || wmin[1] > hi.y ||
                   wmax[2] < lo.z || wmin[2] > hi.z;
        };
        if (nboxes <= 0) mw = 0;
        if (mw > 1) {
            // a wave with pieces to take: if nothing lies within the bounds of its particles'
            // whole paths, no piece of any lane can touch a box and the end positions stand
            float plo[3], phi[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                plo[a] = fminf(xb[a], x[a]);
                phi[a] = fmaxf(xb[a], x[a]);
            }
            wave_bounds(live, plo, phi);
            bool any = false;
            for (int k = 0; k < nboxes && !any; k++) {
                if ((k & (SPH_BOX_CHUNK - 1)) == 0 &&
                    misses(chunkAabb[2 * (k / SPH_BOX_CHUNK)], chunkAabb[2 * (k / SPH_BOX_CHUNK) + 1])) {
                    k += SPH_BOX_CHUNK - 1;
                    continue;
                }
                any = !misses(boxAabb[2 * k], boxAabb[2 * k + 1]);
            }
            if (!any) mw = 0;
        }
        for (int npieces = 0; mw > 0; npieces++) {
        const bool act = live && tnow < dt;
        if (!__any(act)) break;
        const float ts = next_time(npieces);
        const float xs[3] = {xprev[0], xprev[1], xprev[2]}; // where the piece starts: outside every static box
        if (act) {
            const float tau = ts - tb;
#pragma unroll
            for (int a = 0; a < 3; a++) x[a] = xb[a] + v[a] * tau;
        }
        bool touchedAny = false;
        wave_bounds(act, x, x);
        // Every box in index order (the order of the boxes is part of the result); then, if that
        // pushed the particle, the STATIC boxes once more: a push out of one box can end inside a
        // box that was tested before it, and a particle may spend a step inside a moving box but
        // never inside a wall (from beyond a wall's mid-plane the way out is the far side).
        // Two levels: chunks of SPH_BOX_CHUNK consecutive boxes by their joint bounds, then the
        // boxes of a chunk.
        for (int pass = 0; pass < 2; pass++) {
        if (pass == 1 && !__any(touchedAny)) break;
        const bool again = pass == 1 && touchedAny; // (lanes the first pass did not push: unchanged)
        for (int k = 0; k < nboxes; k++) {
            if ((k & (SPH_BOX_CHUNK - 1)) == 0) {
                const int ch = k / SPH_BOX_CHUNK;
                if (misses(chunkAabb[2 * ch], chunkAabb[2 * ch + 1])) {
                    k += SPH_BOX_CHUNK - 1;
                    continue;
                }
            }
            if (misses(boxAabb[2 * k], boxAabb[2 * k + 1])) continue;
            const DevBox &bx = boxes[k];
            if (pass == 1 && bx.dynamic) continue;
            float imp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            bool pushed = false, touched = false;
            if (act && (pass == 0 || again)) {
                const float d[3] = {x[0] - bx.c[0], x[1] - bx.c[1], x[2] - bx.c[2]};
                float p[3], delta[3], nl[3] = {0.f, 0.f, 0.f}, pen = 0.f;
                p[0] = (bx.R[0] * d[0] + bx.R[3] * d[1]) + bx.R[6] * d[2];
                p[1] = (bx.R[1] * d[0] + bx.R[4] * d[1]) + bx.R[7] * d[2];
                p[2] = (bx.R[2] * d[0] + bx.R[5] * d[1]) + bx.R[8] * d[2];
#pragma unroll
                for (int a = 0; a < 3; a++)
                    delta[a] = p[a] - fminf(fmaxf(p[a], -bx.he[a]), bx.he[a]);
                const float dist2 = dot3f(delta, delta);
                bool contact = false;
                if (dist2 > 0.f) {
                    if (dist2 < r * r) {
                        const float dist = sqrtf(dist2);
#pragma unroll
                        for (int a = 0; a < 3; a++) nl[a] = delta[a] / dist;
                        pen = r - dist;
                        contact = true;
                    }
                } else {
                    // centre inside the box: out through the nearest face -- but out of a STATIC
                    // box through the face the piece started beyond, if there is one (a push out of
                    // a moving box can carry a particle past a wall's mid-plane, from where the
                    // nearest face is the far one: nothing ever crosses a static box this way)
                    int best = 0;
                    float bestd = bx.he[0] - fabsf(p[0]);
#pragma unroll
                    for (int a = 1; a < 3; a++) {
                        const float da = bx.he[a] - fabsf(p[a]);
                        if (da < bestd) { bestd = da; best = a; }
                    }
                    float sgn = 0.f;
#pragma unroll
                    for (int a = 0; a < 3; a++)
                        if (a == best) sgn = p[a] < 0.f ? -1.f : 1.f;
                    if (!bx.dynamic) {
                        const float ds[3] = {xs[0] - bx.c[0], xs[1] - bx.c[1], xs[2] - bx.c[2]};
                        float ps[3];
                        ps[0] = (bx.R[0] * ds[0] + bx.R[3] * ds[1]) + bx.R[6] * ds[2];
                        ps[1] = (bx.R[1] * ds[0] + bx.R[4] * ds[1]) + bx.R[7] * ds[2];
                        ps[2] = (bx.R[2] * ds[0] + bx.R[5] * ds[1]) + bx.R[8] * ds[2];
                        float beste = 0.f;
#pragma unroll
                        for (int a = 0; a < 3; a++) {
                            const float ex = fabsf(ps[a]) - bx.he[a];
                            if (ex > beste) {
                                beste = ex;
                                best = a;
                                sgn = ps[a] < 0.f ? -1.f : 1.f;
                            }
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 3; a++)
                        if (a == best) {
                            nl[a] = sgn;
                            bestd = bx.he[a] - sgn * p[a];
                        }
                    pen = bestd + r;
                    contact = true;
                }
                touched = contact;
                if (contact) {
                    float nw[3];
                    nw[0] = (bx.R[0] * nl[0] + bx.R[1] * nl[1]) + bx.R[2] * nl[2];
                    nw[1] = (bx.R[3] * nl[0] + bx.R[4] * nl[1]) + bx.R[5] * nl[2];
                    nw[2] = (bx.R[6] * nl[0] + bx.R[7] * nl[1]) + bx.R[8] * nl[2];
                    float rc[3], vb[3], vrel[3];
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        x[a] = x[a] + nw[a] * pen;
                        rc[a] = (x[a] - nw[a] * r) - bx.c[a];
                    }
                    vb[0] = bx.av[1] * rc[2] - bx.av[2] * rc[1];
                    vb[1] = bx.av[2] * rc[0] - bx.av[0] * rc[2];
                    vb[2] = bx.av[0] * rc[1] - bx.av[1] * rc[0];
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        vb[a] = bx.lv[a] + vb[a];
                        vrel[a] = v[a] - vb[a];
                    }
                    const float vn = dot3f(vrel, nw);
                    if (vn < 0.f) {
                        const float e = fmaxf(particleRestitution, bx.restitution);
                        if (!bx.dynamic) {
                            const float sc = (1.f + e) * vn;
#pragma unroll
                            for (int a = 0; a < 3; a++) v[a] = v[a] - sc * nw[a];
                        } else {
                            float rxn[3], loc[3], w[3];
                            rxn[0] = rc[1] * nw[2] - rc[2] * nw[1];
                            rxn[1] = rc[2] * nw[0] - rc[0] * nw[2];
                            rxn[2] = rc[0] * nw[1] - rc[1] * nw[0];
                            loc[0] = ((bx.R[0] * rxn[0] + bx.R[3] * rxn[1]) + bx.R[6] * rxn[2]) * bx.invI[0];
                            loc[1] = ((bx.R[1] * rxn[0] + bx.R[4] * rxn[1]) + bx.R[7] * rxn[2]) * bx.invI[1];
                            loc[2] = ((bx.R[2] * rxn[0] + bx.R[5] * rxn[1]) + bx.R[8] * rxn[2]) * bx.invI[2];
                            w[0] = (bx.R[0] * loc[0] + bx.R[1] * loc[1]) + bx.R[2] * loc[2];
                            w[1] = (bx.R[3] * loc[0] + bx.R[4] * loc[1]) + bx.R[5] * loc[2];
                            w[2] = (bx.R[6] * loc[0] + bx.R[7] * loc[1]) + bx.R[8] * loc[2];
                            const float denom = (m.invMass + bx.invMass) + dot3f(rxn, w);
                            const float j = -((1.f + e) * vn) / denom;
                            const float jp = j * m.invMass;
#pragma unroll
                            for (int a = 0; a < 3; a++) {
                                v[a] = v[a] + jp * nw[a];
                                imp[a] = owned ? -(j * nw[a]) : 0.f;
                                imp[3 + a] = owned ? -(j * rxn[a]) : 0.f;
                            }
                            pushed = owned;
                        }
                    }
                }
            }
            // reaction on the rigid body: wave-level sum, one atomic per wave and component
            if (bx.dynamic && __any(pushed)) {
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    float t = imp[a];
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
                    if (lane_id() == 0) atomicAdd(&bodyImpulse[6 * k + a], t);
                }
            }
            touchedAny = touchedAny || touched;
            if (__any(touched)) wave_bounds(act, x, x); // positions were pushed out of the box
        }
        }
        if (touchedAny) { // (act lanes only can have touched)
#pragma unroll
            for (int a = 0; a < 3; a++) xb[a] = x[a];
            tb = ts;
        }
        if (act) {
#pragma unroll
            for (int a = 0; a < 3; a++) xprev[a] = x[a];
            tnow = ts;
        }
        }
    }
    if (live) {
        if (INTEGRATE) posm[i] = make_float4(x[0], x[1], x[2], pv.w);
        velq[i] = make_float4(v[0], v[1], v[2], vv.w);
        if (INTEGRATE && nextKeys)
            nextKeys[i] = dense_key(quantize1(x[0], grid.cellSize), quantize1(x[1], grid.cellSize),
                                    quantize1(x[2], grid.cellSize), grid);
    }
    if (INTEGRATE && SLABCOUNT) { // (the whole wave comes here)
        // what k_slab_pack<true> would count from the positions just written: owned records only
        // (a ghost is not sent on), same comparisons on the same floats
        const bool cand = live && owned;
        const bool moveL = cand && x[0] < slabLo, moveR = cand && x[0] >= slabHi;
        const bool sendL = cand && x[0] < slabLo + slabHalo, sendR = cand && x[0] >= slabHi - slabHalo;
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const unsigned long long m = __ballot(side ? sendR : sendL);
            if (m == 0) continue; // wave-uniform
            const unsigned long long mm = __ballot(side ? moveR : moveL);
            if (lane_id() == 0) {
                atomicAdd(&slabNext[side], (uint32_t)__popcll(m));
                if (mm) atomicAdd(&slabNext[2 + side], (uint32_t)__popcll(mm));
            }
        }
    }
}

// ---- rigid bodies: one thread per box (K is small) --------------------------------------------
// state: SphBox as 25 floats (include/sphfluid.h).  integrate != 0: apply impulses, gravity,
// damping, integrate the pose; always: refresh the DevBox view, zero impulses.
__device__ __forceinline__ void quat_to_mat(const float *q, float *m)
{
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    m[0] = 1.f - 2.f * (y * y + z * z); m[1] = 2.f * (x * y - z * w); m[2] = 2.f * (x * z + y * w);
    m[3] = 2.f * (x * y + z * w); m[4] = 1.f - 2.f * (x * x + z * z); m[5] = 2.f * (y * z - x * w);
    m[6] = 2.f * (x * z - y * w); m[7] = 2.f * (y * z + x * w); m[8] = 1.f - 2.f * (x * x + y * y);
}

__global__ void k_bodies(float *__restrict__ state, DevBox *__restrict__ dev,
                         float4 *__restrict__ boxAabb, float *__restrict__ bodyImpulse,
                         const float *__restrict__ velIn, int nboxes, int integrate, float dt,
                         float gx, float gy, float gz, int bounded, float bminx, float bminy,
                         float bminz, float bmaxx, float bmaxy, float bmaxz)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nboxes) return;
    float *b = state + 25 * k;
    float *center = b, *rot = b + 3, *he = b + 7, *lv = b + 10, *av = b + 13, *invI = b + 17;
    const float invMass = b[16], linD = b[22], angD = b[23], gfac = b[24];
    const bool dynamic = invMass != 0.f || invI[0] != 0.f || invI[1] != 0.f || invI[2] != 0.f;
    float m[9];
    quat_to_mat(rot, m);
    // integrate: 0 = refresh the device view only, 1 = the whole update, 2 = velocities only
    // (impulses, gravity, damping: before the box-box contact passes), 3 = the rest (velocities
    // from velIn, tank contacts, pose)
    const bool doVel = integrate == 1 || integrate == 2, doRest = integrate == 1 || integrate == 3;
    if (integrate == 3 && velIn && dynamic)
        for (int a = 0; a < 3; a++) {
            lv[a] = velIn[6 * k + a];
            av[a] = velIn[6 * k + 3 + a];
        }
    if (doVel && dynamic) {
        const float *imp = bodyImpulse + 6 * k;
        float loc[3], dw[3];
        const float g[3] = {gx, gy, gz};
        loc[0] = ((m[0] * imp[3] + m[3] * imp[4]) + m[6] * imp[5]) * invI[0];
        loc[1] = ((m[1] * imp[3] + m[4] * imp[4]) + m[7] * imp[5]) * invI[1];
        loc[2] = ((m[2] * imp[3] + m[5] * imp[4]) + m[8] * imp[5]) * invI[2];
        dw[0] = (m[0] * loc[0] + m[1] * loc[1]) + m[2] * loc[2];
        dw[1] = (m[3] * loc[0] + m[4] * loc[1]) + m[5] * loc[2];
        dw[2] = (m[6] * loc[0] + m[7] * loc[1]) + m[8] * loc[2];
        const float ld = fminf(fmaxf(1.f - linD * dt, 0.f), 1.f);
        const float ad = fminf(fmaxf(1.f - angD * dt, 0.f), 1.f);
        for (int a = 0; a < 3; a++) {
            lv[a] = (lv[a] + imp[a] * invMass + g[a] * gfac * dt) * ld;
            av[a] = (av[a] + dw[a]) * ad;
        }
    }
    if (doRest && dynamic) {
        const float bmin[3] = {bminx, bminy, bminz}, bmax[3] = {bmaxx, bmaxy, bmaxz};
        // contacts of the box's eight corners with the six planes of [bmin,bmax] (the tank's
        // walls): four passes of sequential impulses (ConvertedSubScene.unity:235), speculative,
        // inelastic, frictionless (geometric mean with the walls' friction 0, :742-746)
        if (bounded) {
            for (int it = 0; it < 4; it++)
                for (int corner = 0; corner < 8; corner++) {
                    const float hl[3] = {(corner & 1) ? he[0] : -he[0], (corner & 2) ? he[1] : -he[1],
                                         (corner & 4) ? he[2] : -he[2]};
                    float r[3];
                    r[0] = (m[0] * hl[0] + m[1] * hl[1]) + m[2] * hl[2];
                    r[1] = (m[3] * hl[0] + m[4] * hl[1]) + m[5] * hl[2];
                    r[2] = (m[6] * hl[0] + m[7] * hl[1]) + m[8] * hl[2];
                    for (int a = 0; a < 3; a++)
                        for (int side = 0; side < 2; side++) {
                            const float vc[3] = {lv[0] + (av[1] * r[2] - av[2] * r[1]),
                                                 lv[1] + (av[2] * r[0] - av[0] * r[2]),
                                                 lv[2] + (av[0] * r[1] - av[1] * r[0])};
                            const float sgn = side ? -1.0f : 1.0f;
                            const float pc = center[a] + r[a];
                            const float gap = side ? bmax[a] - pc : pc - bmin[a];
                            const float need = -gap / dt - sgn * vc[a];
                            if (need > 0.0f) {
                                float n[3] = {0.0f, 0.0f, 0.0f};
                                n[a] = sgn;
                                const float rxn[3] = {r[1] * n[2] - r[2] * n[1], r[2] * n[0] - r[0] * n[2],
                                                      r[0] * n[1] - r[1] * n[0]};
                                float l2[3], w2[3];
                                l2[0] = ((m[0] * rxn[0] + m[3] * rxn[1]) + m[6] * rxn[2]) * invI[0];
                                l2[1] = ((m[1] * rxn[0] + m[4] * rxn[1]) + m[7] * rxn[2]) * invI[1];
                                l2[2] = ((m[2] * rxn[0] + m[5] * rxn[1]) + m[8] * rxn[2]) * invI[2];
                                w2[0] = (m[0] * l2[0] + m[1] * l2[1]) + m[2] * l2[2];
                                w2[1] = (m[3] * l2[0] + m[4] * l2[1]) + m[5] * l2[2];
                                w2[2] = (m[6] * l2[0] + m[7] * l2[1]) + m[8] * l2[2];
                                const float K = invMass + ((rxn[0] * w2[0] + rxn[1] * w2[1]) + rxn[2] * w2[2]);
                                const float j = need / K;
                                for (int b = 0; b < 3; b++) {
                                    lv[b] = lv[b] + (j * invMass) * n[b];
                                    av[b] = av[b] + j * w2[b];
                                }
                            }
                        }
                }
        }
        for (int a = 0; a < 3; a++) center[a] = center[a] + lv[a] * dt;
        const float hd = 0.5f * dt;
        const float q0 = rot[0], q1 = rot[1], q2 = rot[2], q3 = rot[3];
        float nq[4];
        nq[0] = q0 + hd * ((av[0] * q3 + av[1] * q2) - av[2] * q1);
        nq[1] = q1 + hd * ((av[1] * q3 + av[2] * q0) - av[0] * q2);
        nq[2] = q2 + hd * ((av[2] * q3 + av[0] * q1) - av[1] * q0);
        nq[3] = q3 + hd * -((av[0] * q0 + av[1] * q1) + av[2] * q2);
        float len2 = 0.f;
        for (int a = 0; a < 4; a++) len2 += nq[a] * nq[a];
        const float il = 1.f / sqrtf(len2);
        for (int a = 0; a < 4; a++) rot[a] = nq[a] * il;
        quat_to_mat(rot, m);
        if (bounded) {
            for (int a = 0; a < 3; a++) {
                const float ext = (fabsf(m[3 * a]) * he[0] + fabsf(m[3 * a + 1]) * he[1]) +
                                  fabsf(m[3 * a + 2]) * he[2];
                if (center[a] - ext < bmin[a]) {
                    center[a] = bmin[a] + ext;
                    if (lv[a] < 0.f) lv[a] = 0.f;
                }
                if (center[a] + ext > bmax[a]) {
                    center[a] = bmax[a] - ext;
                    if (lv[a] > 0.f) lv[a] = 0.f;
                }
            }
        }
    }
    DevBox d;
    for (int a = 0; a < 3; a++) {
        d.c[a] = center[a];
        d.he[a] = he[a];
        d.lv[a] = lv[a];
        d.av[a] = av[a];
        d.invI[a] = invI[a];
    }
    for (int a = 0; a < 9; a++) d.R[a] = m[a];
    d.invMass = invMass;
    d.restitution = b[20];
    d.dynamic = dynamic ? 1 : 0;
    for (int a = 0; a < 3; a++) {
        const float ext = (fabsf(m[3 * a]) * he[0] + fabsf(m[3 * a + 1]) * he[1]) + fabsf(m[3 * a + 2]) * he[2];
        d.amin[a] = center[a] - ext;
        d.amax[a] = center[a] + ext;
    }
    dev[k] = d;
    boxAabb[2 * k] = make_float4(d.amin[0], d.amin[1], d.amin[2], 0.f);
    boxAabb[2 * k + 1] = make_float4(d.amax[0], d.amax[1], d.amax[2], 0.f);
    for (int a = 0; a < 6; a++) bodyImpulse[6 * k + a] = 0.f;
}

// joint bounds of every SPH_BOX_CHUNK consecutive boxes (broadphase of k_apply_integrate)
__global__ void k_box_chunks(const float4 *__restrict__ boxAabb, int nboxes,
                             float4 *__restrict__ chunkAabb)
{
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    const int k0 = ch * SPH_BOX_CHUNK;
    if (k0 >= nboxes) return;
    float4 lo = boxAabb[2 * k0], hi = boxAabb[2 * k0 + 1];
    for (int k = k0 + 1; k < min(k0 + SPH_BOX_CHUNK, nboxes); k++) {
        const float4 a = boxAabb[2 * k], b = boxAabb[2 * k + 1];
        lo.x = fminf(lo.x, a.x); lo.y = fminf(lo.y, a.y); lo.z = fminf(lo.z, a.z);
        hi.x = fmaxf(hi.x, b.x); hi.y = fmaxf(hi.y, b.y); hi.z = fmaxf(hi.z, b.z);
    }
    chunkAabb[2 * ch] = lo;
    chunkAabb[2 * ch + 1] = hi;
}

// ---- density field for rendering (the reference's open roadmap item "fluid texture",
// readme_zh.md:303): rho(x) = sum_j m_j W_poly6(|x - x_j|, h_j) over the fluid particles of the 27
// cells around x, at the nodes of a regular grid, x fastest.  One thread per node; the particles
// are the resident sorted records, found through the cell table.
__global__ __launch_bounds__(256) void k_sample_density(
    const float4 *__restrict__ posm, const uint32_t *__restrict__ mat, const Mat *__restrict__ mats,
    const int2 *__restrict__ table, GridDesc g, float ox, float oy, float oz, float spacing, int nx,
    int ny, int nz, float *__restrict__ field)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)nx * ny * nz) return;
    const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((long)nx * ny));
    const float x = ox + (float)ix * spacing, y = oy + (float)iy * spacing, z = oz + (float)iz * spacing;
    const int cx = quantize1(x, g.cellSize), cy = quantize1(y, g.cellSize), cz = quantize1(z, g.cellSize);
    float sum = 0.f;
    for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dz = -1; dz <= 1; dz++) {
                const int2 r = table[dense_key(cx + dx, cy + dy, cz + dz, g)];
                for (int j = r.x; j < r.y; j++) {
                    const float4 p = posm[j];
                    const Mat &m = mats[mat[j]];
                    const float ddx = x - p.x, ddy = y - p.y, ddz = z - p.z;
                    const float d2 = (ddx * ddx + ddy * ddy) + ddz * ddz;
                    if (d2 < m.h2) {
                        const float w = m.h2 - d2;
                        sum += m.poly6 * ((w * w) * w);
                    }
                }
            }
    field[i] = sum;
}

// ---- contacts between boxes (row I, build-defined; Unity Physics / Havok in the reference) ------
// A contact is a corner of box X inside box Y: normal = the face of Y that X's centre lies beyond
// (in units of Y's half extents; out of Y), depth = the corner's distance below that face.  Velocities are corrected by four Jacobi passes (ConvertedSubScene.unity
// :235, four solver iterations): every pass reads the velocities of the pass before, each dynamic
// body sums the impulses of all its contacts -- as corner owner and as container -- computed with
// its inverse mass and inertia scaled by its number of contacts (mass splitting, so that the summed
// corrections do not overshoot).  Per contact: restitution max(e_X, e_Y) (CombineMode 2), a
// penetration bias 0.2 * depth / dt, Coulomb friction sqrt(mu_X * mu_Y) (CombineMode 0: the
// BuoyancyObject's 0.5 against another such object, 0 against the walls; :742-751).
struct BodyRO {
    float c[3], R[9], he[3], invMass, invI[3], e, mu;
    bool dynamic;
};
__device__ __forceinline__ BodyRO body_view(const float *__restrict__ state, int k)
{
    const float *b = state + 25 * k;
    BodyRO v;
    for (int a = 0; a < 3; a++) { v.c[a] = b[a]; v.he[a] = b[7 + a]; v.invI[a] = b[17 + a]; }
    quat_to_mat(b + 3, v.R);
    v.invMass = b[16];
    v.e = b[20];
    v.mu = b[21];
    v.dynamic = v.invMass != 0.f || v.invI[0] != 0.f || v.invI[1] != 0.f || v.invI[2] != 0.f;
    return v;
}
// world-space inverse inertia applied to v:  R diag(invI) R^T v
__device__ __forceinline__ void inv_inertia(const BodyRO &B, const float *v, float *o)
{
    float l[3];
    l[0] = ((B.R[0] * v[0] + B.R[3] * v[1]) + B.R[6] * v[2]) * B.invI[0];
    l[1] = ((B.R[1] * v[0] + B.R[4] * v[1]) + B.R[7] * v[2]) * B.invI[1];
    l[2] = ((B.R[2] * v[0] + B.R[5] * v[1]) + B.R[8] * v[2]) * B.invI[2];
    o[0] = (B.R[0] * l[0] + B.R[1] * l[1]) + B.R[2] * l[2];
    o[1] = (B.R[3] * l[0] + B.R[4] * l[1]) + B.R[5] * l[2];
    o[2] = (B.R[6] * l[0] + B.R[7] * l[1]) + B.R[8] * l[2];
}
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// corner ci of X inside Y?  n: world normal out of Y, rX / rY: the corner relative to the centres
__device__ __forceinline__ bool corner_contact(const BodyRO &X, const BodyRO &Y, int ci, float *n,
                                               float &depth, float *rX, float *rY)
{
    const float hl[3] = {(ci & 1) ? X.he[0] : -X.he[0], (ci & 2) ? X.he[1] : -X.he[1],
                         (ci & 4) ? X.he[2] : -X.he[2]};
    rX[0] = (X.R[0] * hl[0] + X.R[1] * hl[1]) + X.R[2] * hl[2];
    rX[1] = (X.R[3] * hl[0] + X.R[4] * hl[1]) + X.R[5] * hl[2];
    rX[2] = (X.R[6] * hl[0] + X.R[7] * hl[1]) + X.R[8] * hl[2];
    for (int a = 0; a < 3; a++) rY[a] = (X.c[a] + rX[a]) - Y.c[a];
    float q[3];
    q[0] = (Y.R[0] * rY[0] + Y.R[3] * rY[1]) + Y.R[6] * rY[2];
    q[1] = (Y.R[1] * rY[0] + Y.R[4] * rY[1]) + Y.R[7] * rY[2];
    q[2] = (Y.R[2] * rY[0] + Y.R[5] * rY[1]) + Y.R[8] * rY[2];
    for (int a = 0; a < 3; a++)
        if (!(Y.he[a] - fabsf(q[a]) > 0.f)) return false; // outside (or on) the box
    // the face the corner came in through: the one X's centre is farthest outside of, in units of
    // Y's half extents (the nearest face would flip to a side face near an edge)
    const float dc[3] = {X.c[0] - Y.c[0], X.c[1] - Y.c[1], X.c[2] - Y.c[2]};
    float cq[3];
    cq[0] = (Y.R[0] * dc[0] + Y.R[3] * dc[1]) + Y.R[6] * dc[2];
    cq[1] = (Y.R[1] * dc[0] + Y.R[4] * dc[1]) + Y.R[7] * dc[2];
    cq[2] = (Y.R[2] * dc[0] + Y.R[5] * dc[1]) + Y.R[8] * dc[2];
    int best = 0;
    float bests = fabsf(cq[0]) / Y.he[0];
    for (int a = 1; a < 3; a++) {
        const float sa = fabsf(cq[a]) / Y.he[a];
        if (sa > bests) { best = a; bests = sa; }
    }
    const float sgn = cq[best] < 0.f ? -1.f : 1.f;
    const float bestd = Y.he[best] - sgn * q[best];
    n[0] = Y.R[best] * sgn;     // column `best` of R
    n[1] = Y.R[3 + best] * sgn;
    n[2] = Y.R[6 + best] * sgn;
    depth = bestd;
    return true;
}
// impulse J on X (and -J on Y) of one contact, from the velocities vel[6*k..] of the pass before
__device__ __forceinline__ void contact_impulse(const BodyRO &X, const BodyRO &Y, const float *vX,
                                                const float *vY, float nX, float nY, const float *n,
                                                float depth, const float *rX, const float *rY,
                                                float dt, float *J)
{
    float wX[3], wY[3], vrel[3];
    cross3(vX + 3, rX, wX);
    cross3(vY + 3, rY, wY);
    for (int a = 0; a < 3; a++) vrel[a] = (vX[a] + wX[a]) - (vY[a] + wY[a]);
    const float vn = (vrel[0] * n[0] + vrel[1] * n[1]) + vrel[2] * n[2];
    const float e = fmaxf(X.e, Y.e), mu = sqrtf(X.mu * Y.mu);
    const float wanted = fmaxf(vn < 0.f ? -e * vn : 0.f, 0.2f * depth / dt);
    const float dvn = wanted - vn;
    J[0] = J[1] = J[2] = 0.f;
    if (!(dvn > 0.f)) return;
    auto kdir = [&](const float *d) {
        float cx[3], cy[3], ix[3], iy[3];
        cross3(rX, d, cx);
        cross3(rY, d, cy);
        inv_inertia(X, cx, ix);
        inv_inertia(Y, cy, iy);
        return (nX * X.invMass + nY * Y.invMass) +
               (nX * ((cx[0] * ix[0] + cx[1] * ix[1]) + cx[2] * ix[2]) +
                nY * ((cy[0] * iy[0] + cy[1] * iy[1]) + cy[2] * iy[2]));
    };
    const float jn = dvn / kdir(n);
    float t[3];
    for (int a = 0; a < 3; a++) t[a] = vrel[a] - vn * n[a];
    const float vt = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
    float jt = 0.f;
    if (mu > 0.f && vt > 1.0e-6f) {
        for (int a = 0; a < 3; a++) t[a] = t[a] / vt;
        jt = fminf(mu * jn, vt / kdir(t));
    }
    for (int a = 0; a < 3; a++) J[a] = jn * n[a] - jt * t[a];
}
__device__ __forceinline__ bool aabb_overlap(const float4 *__restrict__ aabb, int a, int b)
{
    const float4 al = aabb[2 * a], ah = aabb[2 * a + 1], bl = aabb[2 * b], bh = aabb[2 * b + 1];
    return !(ah.x < bl.x || al.x > bh.x || ah.y < bl.y || al.y > bh.y || ah.z < bl.z || al.z > bh.z);
}

// number of contacts of every body, and its velocities copied to vel (6 floats a body)
__global__ void k_body_contact_count(const float *__restrict__ state, const float4 *__restrict__ aabb,
                                     int nboxes, float *__restrict__ ncontacts, float *__restrict__ vel)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nboxes) return;
    for (int a = 0; a < 6; a++) vel[6 * k + a] = state[25 * k + 10 + a];
    const BodyRO A = body_view(state, k);
    int cnt = 0;
    for (int j = 0; j < nboxes; j++) {
        if (j == k || !aabb_overlap(aabb, k, j)) continue;
        const BodyRO B = body_view(state, j);
        if (!A.dynamic && !B.dynamic) continue;
        float n[3], d, r1[3], r2[3];
        for (int ci = 0; ci < 8; ci++) {
            if (corner_contact(A, B, ci, n, d, r1, r2)) cnt++;
            if (corner_contact(B, A, ci, n, d, r1, r2)) cnt++;
        }
    }
    ncontacts[k] = (float)max(cnt, 1);
}

// one Jacobi pass: vin -> vout
__global__ void k_body_contact_pass(const float *__restrict__ state, const float4 *__restrict__ aabb,
                                    const float *__restrict__ ncontacts, int nboxes, float dt,
                                    const float *__restrict__ vin, float *__restrict__ vout)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nboxes) return;
    float v[6];
    for (int a = 0; a < 6; a++) v[a] = vin[6 * k + a];
    const BodyRO A = body_view(state, k);
    if (A.dynamic) {
        float dl[3] = {0.f, 0.f, 0.f}, da[3] = {0.f, 0.f, 0.f};
        const float nA = ncontacts[k];
        for (int j = 0; j < nboxes; j++) {
            if (j == k || !aabb_overlap(aabb, k, j)) continue;
            const BodyRO B = body_view(state, j);
            const float nB = ncontacts[j];
            float n[3], d, rX[3], rY[3], J[3], tq[3], w[3];
            for (int ci = 0; ci < 8; ci++) {
                if (corner_contact(A, B, ci, n, d, rX, rY)) { // A's corner in B: +J on A
                    contact_impulse(A, B, vin + 6 * k, vin + 6 * j, nA, nB, n, d, rX, rY, dt, J);
                    cross3(rX, J, tq);
                    inv_inertia(A, tq, w);
                    for (int a = 0; a < 3; a++) { dl[a] = dl[a] + J[a] * A.invMass; da[a] = da[a] + w[a]; }
                }
                if (corner_contact(B, A, ci, n, d, rX, rY)) { // B's corner in A: -J on A
                    contact_impulse(B, A, vin + 6 * j, vin + 6 * k, nB, nA, n, d, rX, rY, dt, J);
                    cross3(rY, J, tq);
                    inv_inertia(A, tq, w);
                    for (int a = 0; a < 3; a++) { dl[a] = dl[a] - J[a] * A.invMass; da[a] = da[a] - w[a]; }
                }
            }
        }
        for (int a = 0; a < 3; a++) { v[a] = v[a] + dl[a]; v[3 + a] = v[3 + a] + da[a]; }
    }
    for (int a = 0; a < 6; a++) vout[6 * k + a] = v[a];
}
