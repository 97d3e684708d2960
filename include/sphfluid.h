/*
 * sphfluid.h -- C ABI of libsphfluid.so: the MI355X (gfx950) SPH fluid step.
 *
 * This is the drop-in boundary for the one hot path of Acceyuriko/FluidSimulation: the per-step
 * SPH update.  The reference has no FFI; the seam this library sits behind is the compute-shader
 * binding set of its GPU system (all citations relative to the reference checkout):
 *   Assets/Scripts/FluidSPHByGPUSystem.cs:227-252   upload, six dispatches, blocking readback
 *   Assets/Resources/FluidSolver.compute            the six kernels
 * and the results must equal those of its CPU/Burst path:
 *   Assets/Scripts/FluidSPHByCPUSystem.cs:258-333, Assets/Scripts/FluidSPHUtils.cs.
 * A C# system P/Invokes these entry points (binding stub: INTEGRATION.md).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * SPH_E_* code and never throws; sph_last_error() gives the text.  Host arrays are caller-owned
 * and not retained past the call; all device memory is owned by the opaque SphCtx.  One context
 * per world, one HIP stream per context, calls on one context are not re-entrant.  There is no
 * CPU fallback: if no gfx950 device is usable sph_create fails.
 */
#ifndef SPHFLUID_H
#define SPHFLUID_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_API __attribute__((visibility("default")))

enum {
    SPH_OK = 0,
    SPH_E_INVALID = -1,   /* bad argument / call order */
    SPH_E_NOMEM = -2,     /* device or host allocation failed */
    SPH_E_DEVICE = -3,    /* HIP runtime error */
    SPH_E_NODEVICE = -4,  /* no usable gfx950 device */
    SPH_E_TOOLARGE = -5   /* size beyond a stated limit */
};

/* FluidParticleComponent, Assets/Scripts/FluidParticleComponent.cs:9-12 (= HLSL Particle,
 * FluidSolver.compute:114-120): 16 bytes, blittable. */
typedef struct SphParticle {
    float radius;
    float restDensity;
    float viscosity;
    float gasConstant;
} SphParticle;

/* BoundaryParticleComponent, Assets/Scripts/BoundaryParticleComponent.cs:6-8 (= HLSL
 * BoundaryParticle, FluidSolver.compute:182-187): 12 bytes. */
typedef struct SphBoundaryParticle {
    float radius;
    float mass;
    float restDensity;
} SphBoundaryParticle;

/* Per-step scalars.  kernelRadiusRate: FluidSPHUtils.cs:128 (4).  gridSize: the result of
 * FindGridSizeJob (FluidSPHUtils.cs:28-44; see sph_find_grid_size).  gravity: hard-coded
 * (0,-9.81,0) in the reference (FluidSPHByCPUSystem.cs:140).  dt: 1/FPS (:266,:323). */
typedef struct SphParams {
    float dt;
    float kernelRadiusRate;
    float gravity[3];
    float gridSize;
    float particleRestitution; /* Particle.prefab:142-146 (0.5, combine = max)            */
    int32_t boundBodies;       /* !=0: keep dynamic boxes' AABBs inside [bodyMin,bodyMax]  */
    float bodyMin[3];
    float bodyMax[3];
    int32_t bodyContacts;      /* !=0: contacts between boxes (a corner of one inside another:
                                  dynamic-dynamic and dynamic-static), DESIGN.md section 6       */
} SphParams;

/* Oriented box collider / rigid body (row I; Unity Physics / Havok in the reference, configured
 * by Wall.prefab:99-279 and ConvertedSubScene.unity:704-885).  25 floats = 100 bytes. */
typedef struct SphBox {
    float center[3];
    float rot[4];          /* unit quaternion x,y,z,w, body -> world */
    float halfExtents[3];
    float linVel[3];
    float angVel[3];
    float invMass;         /* 0 and invInertia 0 => static */
    float invInertia[3];   /* body-space diagonal */
    float restitution;
    float friction;        /* carried; particle friction is 0 with combine = min */
    float linDamping;
    float angDamping;
    float gravityFactor;
} SphBox;

typedef struct SphConfig {
    int32_t device;          /* HIP device ordinal */
    int32_t maxCells;        /* cap on the periodic cell table (0 = default 1<<25)  */
    int32_t enableTimings;   /* HIP events per stage (sph_get_timings): 1 every stage (about 1.4 % of a
                                C3 step: each pair of events puts a bubble between two kernels), 2 the two
                                neighbour kernels only */
    int32_t pinHostArrays;   /* Host copies: large arrays (>= 8 MB) that are not registered go through a
                                ring of pinned 16 MB chunks of the library's own, filled / emptied by a few
                                host threads (SPHFLUID_STAGE_THREADS, default 4) while the DMA of the chunk
                                before runs.
                                !=0: caller arrays of sph_upload / sph_download / sph_step* are
                                registered with the driver on first use (hipHostRegister) and stay
                                registered until sph_unpin_host / sph_destroy: use it when the same
                                arrays come back every frame (persistent NativeArrays), and call
                                sph_unpin_host before freeing one */
} SphConfig;

/* Periodic cell table the dense sort key is built on (DESIGN.md "cell keys").  A particle in
 * quantised cell c (FluidSPHUtils.cs:236-243) has table coordinates (x,y,z) = (c - origin) mod
 * dims; cells are grouped in tiles of 4 x 4 x 2 and key = tile * 32 + local, tiles x-fastest over
 * the table, cells x-fastest inside a tile.  dims are rounded up to whole tiles. */
typedef struct SphGrid {
    int32_t origin[3];
    int32_t dims[3];
    int32_t keyBits; /* ceil(log2(dims product)) */
    int32_t cells;   /* dims product */
} SphGrid;

/* stage indices of SphTimings.ms */
enum {
    SPH_T_KEYS = 0, SPH_T_SORT, SPH_T_TABLE, SPH_T_REORDER, SPH_T_DENSITY, SPH_T_FORCE,
    SPH_T_INTEGRATE, SPH_T_BODIES, SPH_T_PLAN /* neighbour plan: overlaps SPH_T_REORDER in sph_step_resident */,
    SPH_T_COUNT
};
typedef struct SphTimings {
    double ms[SPH_T_COUNT];     /* summed device time per stage since sph_reset_timings */
    int64_t launches[SPH_T_COUNT];
    int64_t steps;
} SphTimings;

/* named device buffers readable with sph_read (mirrors the named ComputeBuffers of
 * FluidSPHByGPUSystem.cs:22-54).  "ECS order" = the order of the arrays given to sph_upload. */
enum {
    SPH_BUF_HASHES = 0,      /* u32[N]   FNV-1 hash of the quantised cell, ECS order            */
    SPH_BUF_CELLS,           /* i32[N*3] quantised cell, ECS order                              */
    SPH_BUF_KEYS,            /* u32[N]   dense key, ECS order                                   */
    SPH_BUF_PERM,            /* u32[N]   sorted slot -> ECS id (stable by key, ties by id)      */
    SPH_BUF_SORTED_KEYS,     /* u32[N]                                                          */
    SPH_BUF_TABLE,           /* i32[cells*2] (start,end) per key, (0,0) when empty              */
    SPH_BUF_DENSITIES,       /* f32[N]   ECS order                                              */
    SPH_BUF_PRESSURES,       /* f32[N]   ECS order                                              */
    SPH_BUF_FORCES,          /* f32[N*3] ECS order                                              */
    SPH_BUF_DENSITY_COUNTS,  /* i32[N]   neighbours with d2 < h2 (self + boundary included)     */
    SPH_BUF_FORCE_COUNTS,    /* i32[N]   neighbours with length < h (j != i, boundary included) */
    SPH_BUF_POSITIONS,       /* f32[N*3] ECS order                                              */
    SPH_BUF_VELOCITIES,      /* f32[N*3] ECS order                                              */
    SPH_BUF_BOUNDARY_PERM,   /* u32[B]                                                          */
    SPH_BUF_BOUNDARY_TABLE,  /* i32[cells*2]                                                    */
    /* Parity audit of the PRODUCTION neighbour set: the hit masks the production density pass
     * leaves for the force pass (candidates with fma(d2, -1/h2, 1) >= 0), walked as the force
     * pass walks them.  Per particle, ECS order: number of hits (self and boundary included, the
     * counterpart of SPH_BUF_DENSITY_COUNTS; -1 for a particle of a crowded cell whose masks
     * found no room in the plan pool: its force pass walks every candidate) and the sum of the
     * hits' ids, fluid ids as given to
     * sph_upload, boundary particle k as N + k.  Needs the production density pass (no
     * SPH_DEBUG_COUNTS) on the current ordering. */
    SPH_BUF_MASK_COUNTS,     /* i32[N]                                                          */
    SPH_BUF_MASK_IDSUMS      /* u64[N]                                                          */
};

typedef struct SphCtx SphCtx;

/* debug flags for sph_set_debug: extra parity outputs, off in production */
enum {
    SPH_DEBUG_HASHES = 1, /* hash_position also writes SPH_BUF_HASHES / SPH_BUF_CELLS        */
    SPH_DEBUG_COUNTS = 2, /* density / force kernels also write the neighbour-count buffers  */
    SPH_DEBUG_NO_LDS_TILE = 4 /* neighbour kernels take their crowded-cell path for every cell
                                 (the path of a cell whose 27 cells overflow the LDS image) */
};

/* ---- life cycle --------------------------------------------------------------------------
 * sph_create replaces OnCreate + the ComputeBuffer (re)allocation of
 * FluidSPHByGPUSystem.cs:59-68,96-157; sph_destroy replaces OnDestroy (:369-399).  Device
 * buffers grow geometrically with the particle count. */
SPH_API int sph_create(const SphConfig *cfg, SphCtx **out);
SPH_API int sph_destroy(SphCtx *ctx);
/* ctx may be NULL: text of the last failure of sph_create on this thread */
SPH_API const char *sph_last_error(const SphCtx *ctx);
SPH_API int sph_device_name(SphCtx *ctx, char *name, size_t cap, int32_t *compute_units);
SPH_API int sph_set_debug(SphCtx *ctx, int32_t flags);
SPH_API int sph_synchronize(SphCtx *ctx);

/* ---- host-side helpers --------------------------------------------------------------------
 * FindGridSizeJob (FluidSPHUtils.cs:28-44, start value :172-173): chunk_first[c] is the index
 * of the first particle of ECS chunk c. */
SPH_API float sph_find_grid_size(const SphParticle *params, const int32_t *chunk_first,
                                 int32_t nchunks, float kernelRadiusRate);

/* ---- scene state --------------------------------------------------------------------------
 * sph_set_boundary: BoundaryPositions / BoundaryParticles buffers (FluidSPHByGPUSystem.cs
 * :186-201,232-233).  Static: uploaded and sorted once, not every step. */
SPH_API int sph_set_boundary(SphCtx *ctx, const float *bpos /* B*3 */,
                             const SphBoundaryParticle *bparams /* B */, int32_t B);
SPH_API int sph_set_colliders(SphCtx *ctx, const SphBox *boxes, int32_t K);
SPH_API int sph_get_colliders(SphCtx *ctx, SphBox *boxes_out, int32_t K);
/* explicit periodic cell table (every dim >= 4); without it the table is sized from the
 * particles' bounds at sph_upload */
SPH_API int sph_set_grid(SphCtx *ctx, const SphGrid *grid);
SPH_API int sph_get_grid(SphCtx *ctx, SphGrid *grid_out);

/* Positions / Masses(=inverse mass) / Velocities / Particles buffers
 * (FluidSPHByGPUSystem.cs:164-181,227-230).  The order of these arrays is "ECS order". */
SPH_API int sph_upload(SphCtx *ctx, const float *pos /* N*3 */, const float *vel /* N*3 */,
                       const float *invMass /* N */, const SphParticle *params /* N */,
                       int32_t N, float gridSize);
SPH_API int sph_download(SphCtx *ctx, float *pos_out /* N*3 */, float *vel_out /* N*3 */);
/* SphConfig.pinHostArrays bookkeeping (the reference's ComputeBuffer.SetData / AsyncGPUReadback
 * span, FluidSPHByGPUSystem.cs:227-233,249-252, stages through driver-owned memory; here the
 * caller's arrays themselves are registered).  sph_unpin_host releases the registration of one
 * array (array == NULL: of all): call it BEFORE freeing or reallocating an array the context has
 * seen.  sph_pin_stats: stats[0] = arrays registered now, [1] = registrations that succeeded,
 * [2] = that failed (those arrays are copied from pageable memory), since sph_create; after a
 * failure sph_last_error holds the driver's reason for the last one. */
SPH_API int sph_unpin_host(SphCtx *ctx, const void *array);
SPH_API int sph_pin_stats(SphCtx *ctx, int32_t stats[3]);
/* FluidInitializeSystem.CreateParticles (FluidInitializeSystem.cs:63-92) on the device, so that a
 * scene never crosses PCIe: fills the local box [boundsMin, boundsMin +
 * boundsSize] of an authoring volume with a lattice at spacing 2*radius in world units, x fastest
 * then y then z, mapped through localToWorld (3x4 row-major, as MultiplyPoint3x4), velocity 0,
 * every particle with *params / invMass.  append != 0 adds to the resident fluid (ids continue),
 * else replaces it.  The new particles' ids are [n_before, n_before + *added). */
SPH_API int sph_seed_volume(SphCtx *ctx, const float localToWorld[12], const float boundsMin[3],
                            const float boundsSize[3], float radius, const SphParticle *params,
                            float invMass, float gridSize, int32_t append, int32_t *added);

/* ---- stages on the device-resident state --------------------------------------------------
 * One entry point per dispatch of FluidSPHByGPUSystem.cs:242-247 (same names), plus the two
 * CPU-side jobs that follow it.  Asynchronous on the context's stream. */
SPH_API int sph_hash_position(SphCtx *ctx, const SphParams *p);   /* HashPosition   :269-277 */
SPH_API int sph_radix_sort(SphCtx *ctx);          /* replaces BitonicSort        :279-299 */
SPH_API int sph_map_table(SphCtx *ctx);           /* ClearTable + MapTable       :301-316 */
SPH_API int sph_reorder(SphCtx *ctx);             /* gather into sorted SoA (the reference
                                                     indirects through IndexMap instead,
                                                     FluidSolver.compute:156-166)           */
SPH_API int sph_compute_density_and_pressure(SphCtx *ctx, const SphParams *p); /* :318-331 */
SPH_API int sph_compute_force(SphCtx *ctx, const SphParams *p);                /* :333-348 */
SPH_API int sph_apply_force(SphCtx *ctx, const SphParams *p);  /* ApplyForceJob,
                                                     FluidSPHUtils.cs:105-124              */
SPH_API int sph_integrate(SphCtx *ctx, const SphParams *p);    /* row I: x += v dt, box
                                                     colliders, rigid bodies (Unity Physics
                                                     step in the reference)                */
/* the whole per-frame sequence (FluidSPHByCPUSystem.cs:258-333 + the physics step), nsteps
 * times, without touching the host */
SPH_API int sph_step_resident(SphCtx *ctx, const SphParams *p, int32_t nsteps);

/* ---- one-call host round trips ------------------------------------------------------------
 * sph_step: upload, one full step, download; pos/vel are in/out, densities_out (N) and
 * forces_out (N*3) may be NULL.  Synchronous on return (as WaitForCompletion, :251-252). */
SPH_API int sph_step(SphCtx *ctx, const SphParams *p, float *pos, float *vel,
                     const float *invMass, const SphParticle *params, int32_t N,
                     float *densities_out, float *forces_out);
/* sph_step_compat: exactly the span FluidSPHByGPUSystem.cs:227-252 -- upload, the dispatches up
 * to ComputeForce, readback of Densities and Forces -- so the caller's unmodified ApplyForceJob
 * (:254-262) and physics step finish the frame. */
SPH_API int sph_step_compat(SphCtx *ctx, const SphParams *p, const float *pos, const float *vel,
                            const float *invMass, const SphParticle *params, int32_t N,
                            float *densities_out, float *forces_out);

/* ---- parity / inspection ------------------------------------------------------------------ */
SPH_API int sph_read(SphCtx *ctx, int32_t buffer, void *out, size_t bytes);
/* FindBoundsJob's snapped AABB (FluidSPHUtils.cs:57-93) of the resident fluid + boundary */
SPH_API int sph_find_bounds(SphCtx *ctx, float gridSize, float mn_out[3], float mx_out[3]);
/* Density field for rendering (the reference's open roadmap item "compute a fluid
 * texture so that the particles render as a liquid", readme_zh.md:303 -- it has no counterpart in the
 * reference's code): rho(x) = sum over fluid particles j of m_j * 315/(64 pi h_j^9) (h_j^2 - |x-x_j|^2)^3
 * at the nodes origin + (ix, iy, iz) * spacing of a grid of dims nodes, written x fastest into the
 * host array field_out (dims[0]*dims[1]*dims[2] floats).  Sorts the resident state first if it has
 * moved since the last sort.  Support radii must not exceed the cell size (27-cell search). */
SPH_API int sph_sample_density(SphCtx *ctx, const SphParams *p, const float origin[3], float spacing,
                               const int32_t dims[3], float *field_out);
/* the radix sort on its own: stable ascending sort of ids 0..n-1 by the low key_bits of keys */
SPH_API int sph_sort_pairs(SphCtx *ctx, const uint32_t *keys, int32_t n, int32_t key_bits,
                           uint32_t *perm_out, uint32_t *sorted_keys_out);
/* No counterpart in the reference.  For callers that cannot allocate -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc: sph_sort_pairs with DEVICE
 * pointers for keys and outputs, every intermediate (one ping-pong (key, value) buffer, the
 * digit histograms) in a file-scope device array of the library, so the call makes no hipMalloc /
 * hipFree.  n <= SPH_SORT_NOALLOC_MAX, else SPH_E_TOOLARGE.  keys_dev is not written; the three
 * buffers must not alias.  Asynchronous on the context's stream (sph_synchronize to wait).  All
 * calls on one GPU share that GPU's array (about 8.3 MB, held until the process ends): calls that
 * overlap on one GPU -- two contexts, two threads -- race. */
#define SPH_SORT_NOALLOC_MAX (1 << 20)
SPH_API int sph_sort_pairs_noalloc(SphCtx *ctx, const uint32_t *keys_dev, int32_t n,
                                   int32_t key_bits, uint32_t *perm_out_dev,
                                   uint32_t *sorted_keys_out_dev);
SPH_API int sph_get_timings(SphCtx *ctx, SphTimings *out); /* synchronises */
SPH_API int sph_reset_timings(SphCtx *ctx);

/* ---- slab decomposition across GPUs --------------------------------------------------------
 * No counterpart in the reference (its solver runs on one device, FluidSPHByGPUSystem.cs:59-68);
 * this is SURVEY.md section 8(e): one process per GPU, every rank owns the fluid particles with
 * x in [xlo, xhi) and trades, once per step, with its two x neighbours.  The library packs and
 * unpacks DEVICE buffers; moving them between ranks (RCCL send/recv) is the caller's part
 * (fluidsimulation_amd/slab.py).  Exchange record, 32 bytes:
 *     float pos[3]; uint32 material; float vel[3]; uint32 id;
 * id bit 31 set = ghost copy (the sender keeps the particle), clear = ownership transfer.
 * Ghosts live for exactly one step (they are cut off by the next step's sort), so the halo must
 * be 2 h: a ghost within h of the face then has all of its own neighbours present and its
 * density, which the force pass needs, is computed locally.
 *
 * Per step: sph_slab_pack -> exchange -> sph_slab_import (each neighbour) -> sph_step_resident.
 * Box colliders are replicated on every rank.  With dynamic boxes the step is split so that the
 * pushes of all ranks' particles reach every copy of a body: sph_slab_step_particles ->
 * sph_slab_get_impulses -> sum over ranks (one all-reduce of K*6 floats, the caller's) ->
 * sph_slab_step_bodies.  Ghosts never push a body (their owner's rank counts them). */

/* Like sph_upload, with global particle ids (< 2^30, unique over all ranks) and an explicit
 * material table that must be identical on every rank: materials[5*m] = radius, restDensity,
 * viscosity, gasConstant, inverseMass; materialIndex[i] < M.  capacity = room for records
 * (owned + ghosts of two consecutive steps) allocated up front; an import that needs more grows
 * the buffers (a reallocation and a copy: size it generously where that matters). */
SPH_API int sph_slab_upload(SphCtx *ctx, const float *pos, const float *vel,
                            const uint32_t *materialIndex, const uint32_t *ids, int32_t N,
                            const float *materials, int32_t M, float gridSize, int32_t capacity);
/* Marks last step's ghosts dead and fills the two DEVICE buffers (capRecords records each):
 * left  <- owned particles with x < xlo + halo (x < xlo: transfer, the local copy stays as a
 * ghost for this step), right <- x >= xhi - halo (x >= xhi: transfer).  Pass -INF / +INF on
 * the outer ranks.  counts = records left, records right, transfers left, transfers right.
 * The order of the records is not deterministic; the order of the particles after the next sort
 * is (cell, id).  Synchronises.  If a side needs more than capRecords the call returns
 * SPH_E_TOOLARGE with counts[] = what is needed and NOTHING packed or re-flagged: grow the buffers
 * and call again. */
SPH_API int sph_slab_pack(SphCtx *ctx, float xlo, float xhi, float halo, void *left_dev,
                          void *right_dev, int32_t capRecords, int32_t counts[4]);
/* Appends `count` records from a DEVICE buffer; `transfers` of them carry no ghost bit (the
 * sender's counts[2] or counts[3]).  Asynchronous on the context's stream. */
SPH_API int sph_slab_import(SphCtx *ctx, const void *records_dev, int32_t count,
                            int32_t transfers);
/* one step without the rigid bodies' update / the bodies' update with the summed impulses
 * (linear xyz, angular xyz per box; NULL = this rank's own) */
SPH_API int sph_slab_step_particles(SphCtx *ctx, const SphParams *p);
SPH_API int sph_slab_get_impulses(SphCtx *ctx, float *impulses_out /* K*6 */, int32_t K);
SPH_API int sph_slab_step_bodies(SphCtx *ctx, const SphParams *p, const float *impulses /* K*6 */,
                                 int32_t K);
/* Native exchange: the same step, with the transport inside the library (RCCL ncclSend / ncclRecv
 * between x neighbours over xGMI, opened with dlopen on first use), for hosts that have no
 * communication layer of their own (the C# host).  No counterpart in the reference.
 *   rank 0: sph_slab_comm_unique_id -> 128 bytes, carried to every rank by the launcher (returns
 *           the number of bytes written, or a negative code)
 *   all:    sph_slab_comm_init(rank, world, id) once, after sph_create (collective)
 *   per step: sph_slab_exchange(xlo, xhi, halo) = sph_slab_pack + send/recv with both neighbours +
 *           sph_slab_import, on the context's stream, with ONE host synchronisation (the sizes of
 *           the messages); then sph_step_resident / sph_slab_step_particles as above.
 *           counts = records sent left, sent right, received from left, received from right.
 * Rank r's neighbours are r-1 and r+1; pass -INF / +INF as xlo / xhi on the outer ranks. */
SPH_API int sph_slab_comm_unique_id(uint8_t *id_out, int32_t cap);
SPH_API int sph_slab_comm_init(SphCtx *ctx, int32_t rank, int32_t world, const uint8_t *unique_id,
                               int32_t id_bytes);
SPH_API int sph_slab_exchange(SphCtx *ctx, float xlo, float xhi, float halo, int32_t counts[4]);
/* room for `records` records a side in the library's send / receive buffers, allocated now: the
 * exchanges that follow allocate nothing unless a message is larger (then the buffers grow) */
SPH_API int sph_slab_exchange_reserve(SphCtx *ctx, int32_t records);
SPH_API int sph_slab_comm_destroy(SphCtx *ctx);
/* counts = records resident, owned, live ghosts, dead ghosts */
SPH_API int sph_slab_counts(SphCtx *ctx, int32_t counts[4]);
/* the owned particles, host arrays, in the resident order; density_out is zero-filled unless a
 * density pass has run since the last reorder.  cap = room in particles. */
SPH_API int sph_slab_download(SphCtx *ctx, float *pos_out, float *vel_out, float *density_out,
                              uint32_t *ids_out, int32_t cap, int32_t *count);

#ifdef __cplusplus
}
#endif
#endif /* SPHFLUID_H */
