/*
 * sph_oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by, or called from the
 * product (libsphfluid.so / fluidsimulation_amd).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker / the reported CPU baseline.
 *
 * A plain-C CPU restatement of the reference's Burst job path (all citations relative to
 * /root/reference):
 *   Quantize                 Assets/Scripts/FluidSPHUtils.cs:236-243
 *   Hash (FNV-1 32 bit)      Assets/Scripts/FluidSPHUtils.cs:246-261
 *   FindGridSizeJob          Assets/Scripts/FluidSPHUtils.cs:28-44, 172-173
 *   FindBoundsJob            Assets/Scripts/FluidSPHUtils.cs:46-94
 *   ComputeDensityAndPressureJob   Assets/Scripts/FluidSPHByCPUSystem.cs:48-104
 *   ComputeForceJob          Assets/Scripts/FluidSPHByCPUSystem.cs:126-236
 *   ApplyForceJob            Assets/Scripts/FluidSPHUtils.cs:105-124
 *   OnUpdate sequence        Assets/Scripts/FluidSPHByCPUSystem.cs:258-333
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for this path and
 * cannot be compiled or run here (C# / Unity 2020.3 / Burst; no .NET toolchain).  This file is
 * therefore pinned only by (1) published FNV-1 test vectors, (2) known answers derived by hand
 * from the reference text and (3) analytic two-particle cases; see
 * tests/test_oracle.py.  Third-party arithmetic restated from its published definition:
 *   com.unity.mathematics 1.2.5: dot = (x*x + y*y) + z*z, length = sqrt(dot), rsqrt = 1/sqrt,
 *     normalize(v) = rsqrt(dot(v,v)) * v, math.PI = 3.14159274f, pow -> libm powf (Burst's pow
 *     may differ in the last ulp: tolerance rows only);
 *   com.unity.collections 1.2.3-pre.1 NativeMultiHashMap: chained buckets, insert at chain head
 *     (same-key values come back most recently added first), bucket = (int)key & mask,
 *     bucket count = ceilpow2(2*capacity).  Growth/rehash order is not restated (it only changes
 *     float summation order);
 *   com.unity.physics 0.50 ApplyLinearImpulse: Linear += impulse * InverseMass.
 * Integration + collider response (orc_integrate_*) has NO reference source (Unity Physics /
 * Havok, closed): it restates this repository's own specification in DESIGN.md section "Row I".
 *
 * Floating point: build with -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile); every
 * expression below is evaluated in IEEE binary32 in the reference's left-to-right order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))
#define ORC_PI 3.14159274f /* Unity.Mathematics math.PI (float) = 0x40490FDB */

/* ------------------------------------------------------------------------------------------ */
/* Quantize + Hash: FluidSPHUtils.cs:236-261                                                    */

static inline int32_t quantize1(float p, float size) { return (int32_t)floorf(p / size); }

ORC_API void orc_quantize(const float *pos, int n, float size, int32_t *cells)
{
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++)
            cells[3 * i + a] = quantize1(pos[3 * i + a], size);
}

/* plain FNV-1 over a byte string: multiply, then xor (used to pin against published vectors) */
ORC_API uint32_t orc_fnv1_bytes(const uint8_t *bytes, int len)
{
    uint32_t hash = 2166136261u;
    for (int i = 0; i < len; i++) {
        hash *= 16777619u;
        hash ^= bytes[i];
    }
    return hash;
}

ORC_API uint32_t orc_hash(int32_t x, int32_t y, int32_t z)
{
    int32_t p[3] = {x, y, z};
    uint32_t hash = 2166136261u;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 4; j++) {
            uint8_t b = (uint8_t)(p[i] >> (j * 8)); /* arithmetic shift, truncated to byte */
            hash *= 16777619u;
            hash ^= b;
        }
    }
    return hash;
}

ORC_API void orc_hash_cells(const int32_t *cells, int n, uint32_t *hashes)
{
    for (int i = 0; i < n; i++)
        hashes[i] = orc_hash(cells[3 * i], cells[3 * i + 1], cells[3 * i + 2]);
}

/* FindGridSizeJob, FluidSPHUtils.cs:35-43 with the float.MaxValue start of :172-173.
 * One entry per ECS chunk: the radius of the chunk's FIRST particle. */
ORC_API float orc_find_grid_size(const float *chunk_first_radius, int nchunks, float rate)
{
    float grid = FLT_MAX;
    for (int c = 0; c < nchunks; c++) {
        float radius = chunk_first_radius[c];
        if (radius < grid / rate)
            grid = radius * rate;
    }
    return grid;
}

/* FindBoundsJob min/max part, FluidSPHUtils.cs:57-93 (the snapped AABB; unused on the CPU path) */
ORC_API void orc_find_bounds(const float *pos, int n, const float *bpos, int b, float grid,
                             float *mn, float *mx)
{
    mn[0] = mn[1] = mn[2] = FLT_MAX;
    mx[0] = mx[1] = mx[2] = -FLT_MAX; /* C# float.MinValue */
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            if (pos[3 * i + a] < mn[a]) mn[a] = pos[3 * i + a];
            if (pos[3 * i + a] > mx[a]) mx[a] = pos[3 * i + a];
        }
    for (int i = 0; i < b; i++)
        for (int a = 0; a < 3; a++) {
            if (bpos[3 * i + a] < mn[a]) mn[a] = bpos[3 * i + a];
            if (bpos[3 * i + a] > mx[a]) mx[a] = bpos[3 * i + a];
        }
    for (int a = 0; a < 3; a++) {
        mn[a] = floorf(mn[a] / grid) * grid;
        mx[a] = ceilf(mx[a] / grid) * grid + 1.0f;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* NativeMultiHashMap<uint,int> restatement (insert at head of the bucket chain)               */

typedef struct {
    int count, capacity, mask;
    int *buckets; /* head entry per bucket, -1 = empty */
    int *next;    /* chain link per entry */
    uint32_t *keys;
    int *values;
} MultiMap;

static int ceilpow2(int x)
{
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

static int mm_init(MultiMap *m, int capacity)
{
    if (capacity < 1) capacity = 1;
    m->count = 0;
    m->capacity = capacity;
    int nb = ceilpow2(capacity * 2);
    m->mask = nb - 1;
    m->buckets = (int *)malloc(sizeof(int) * (size_t)nb);
    m->next = (int *)malloc(sizeof(int) * (size_t)capacity);
    m->keys = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)capacity);
    m->values = (int *)malloc(sizeof(int) * (size_t)capacity);
    if (!m->buckets || !m->next || !m->keys || !m->values) return -1;
    memset(m->buckets, 0xff, sizeof(int) * (size_t)nb);
    return 0;
}

static void mm_free(MultiMap *m)
{
    free(m->buckets); free(m->next); free(m->keys); free(m->values);
    memset(m, 0, sizeof(*m));
}

static inline void mm_add(MultiMap *m, uint32_t key, int value)
{
    int idx = m->count++;
    int b = (int)key & m->mask;
    m->keys[idx] = key;
    m->values[idx] = value;
    m->next[idx] = m->buckets[b];
    m->buckets[b] = idx;
}

/* iteration: for (e = mm_first(m,key); e >= 0; e = mm_next(m,key,e)) value = m->values[e] */
static inline int mm_next(const MultiMap *m, uint32_t key, int e)
{
    e = m->next[e];
    while (e >= 0 && m->keys[e] != key) e = m->next[e];
    return e;
}
static inline int mm_first(const MultiMap *m, uint32_t key)
{
    int e = m->buckets[(int)key & m->mask];
    while (e >= 0 && m->keys[e] != key) e = m->next[e];
    return e;
}

/* FindBoundsJob map fill, FluidSPHUtils.cs:59-83: serial, i ascending */
static int build_maps(const float *pos, int n, const float *bpos, int b, float grid,
                      MultiMap *fluid, MultiMap *boundary)
{
    /* the reference sizes BOTH maps by the fluid count (FluidSPHUtils.cs:189-190) and lets the
     * container grow; we size the boundary map so that it never has to (order unaffected) */
    if (mm_init(fluid, n) || mm_init(boundary, b > n ? b : n)) return -1;
    for (int i = 0; i < n; i++)
        mm_add(fluid, orc_hash(quantize1(pos[3 * i], grid), quantize1(pos[3 * i + 1], grid),
                               quantize1(pos[3 * i + 2], grid)), i);
    for (int i = 0; i < b; i++)
        mm_add(boundary, orc_hash(quantize1(bpos[3 * i], grid), quantize1(bpos[3 * i + 1], grid),
                                  quantize1(bpos[3 * i + 2], grid)), i);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* math helpers in the reference's evaluation order                                            */

static inline float dot3(const float *a, const float *b)
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

/* ComputeDensity, FluidSPHByCPUSystem.cs:95-104 */
static inline float compute_density(float poly6, float h2, float d2, int *hit)
{
    if (d2 < h2) {
        *hit += 1;
        return poly6 * powf(h2 - d2, 3.0f);
    }
    return 0.0f;
}

/* ComputeDensityAndPressureJob.Execute, FluidSPHByCPUSystem.cs:48-93 */
static void density_one(int index, const float *pos, const float *invMass, const float *params,
                        const MultiMap *grid, const float *bpos, const MultiMap *bgrid,
                        float gridSize, float rate, float *dens, float *pres, int32_t *ncount)
{
    float density = 0.0f;
    const float *particle = params + 4 * index; /* radius, restDensity, viscosity, gasConstant */
    const float *position = pos + 3 * index;
    float radius = particle[0];
    float mass = 1.0f / invMass[index];
    float gasConstant = particle[3];
    float restDensity = particle[1];
    int32_t gp[3] = {quantize1(position[0], gridSize), quantize1(position[1], gridSize),
                     quantize1(position[2], gridSize)};
    float kernelRadius = radius * rate;
    float kernelRadius2 = powf(kernelRadius, 2.0f);
    float poly6Constant = mass * 315.0f / (64.0f * ORC_PI * powf(kernelRadius, 9.0f));
    int hits = 0;

    for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++)
            for (int z = -1; z <= 1; z++) {
                uint32_t key = orc_hash(gp[0] + x, gp[1] + y, gp[2] + z);
                for (int e = mm_first(grid, key); e >= 0; e = mm_next(grid, key, e)) {
                    const float *pj = pos + 3 * grid->values[e];
                    float d[3] = {position[0] - pj[0], position[1] - pj[1], position[2] - pj[2]};
                    density += compute_density(poly6Constant, kernelRadius2, dot3(d, d), &hits);
                }
                for (int e = mm_first(bgrid, key); e >= 0; e = mm_next(bgrid, key, e)) {
                    const float *pj = bpos + 3 * bgrid->values[e];
                    float d[3] = {position[0] - pj[0], position[1] - pj[1], position[2] - pj[2]};
                    density += compute_density(poly6Constant, kernelRadius2, dot3(d, d), &hits);
                }
            }
    dens[index] = density;
    pres[index] = fmaxf(gasConstant * (density - restDensity), 0.0f);
    if (ncount) ncount[index] = hits;
}

/* ComputeForcePressureAndViscosity, FluidSPHByCPUSystem.cs:205-236 */
static inline int force_pair(float distance, float h, float viscosity, float densityi,
                             float densityj, float pressurei, float pressurej, const float *vi,
                             const float *vj, float massj, const float *vij, float *fp, float *fv)
{
    if (!(distance < h)) return 0;
    float s = -densityi * massj *
              (pressurei / powf(densityi, 2.0f) + pressurej / powf(densityj, 2.0f)) *
              (-45.0f / (ORC_PI * powf(h, 6.0f)) * powf(h - distance, 2.0f));
    float rs = 1.0f / sqrtf(dot3(vij, vij)); /* math.rsqrt */
    float t = viscosity / densityi * massj;
    for (int a = 0; a < 3; a++) {
        fp[a] += s * (rs * vij[a]);
        fv[a] += t * (vj[a] - vi[a]) * 45.0f / (ORC_PI * powf(h, 6.0f)) * (h - distance);
    }
    return 1;
}

/* ComputeForceJob.Execute, FluidSPHByCPUSystem.cs:126-203 */
static void force_one(int index, const float *pos, const float *invMass, const float *vel,
                      const float *params, const MultiMap *grid, const float *bpos,
                      const float *bparams, const MultiMap *bgrid, const float *dens,
                      const float *pres, float gridSize, float rate, float *forces,
                      int32_t *ncount)
{
    float densityi = dens[index], pressurei = pres[index];
    const float *position = pos + 3 * index;
    int32_t gp[3] = {quantize1(position[0], gridSize), quantize1(position[1], gridSize),
                     quantize1(position[2], gridSize)};
    float kernelRadius = params[4 * index] * rate;
    float viscosity = params[4 * index + 2];
    const float *velocityi = vel + 3 * index;
    float fp[3] = {0, 0, 0}, fv[3] = {0, 0, 0};
    float fg[3] = {0.0f * densityi, -9.81f * densityi, 0.0f * densityi};
    const float zero3[3] = {0, 0, 0};
    int hits = 0;

    for (int x = -1; x <= 1; x++)
        for (int y = -1; y <= 1; y++)
            for (int z = -1; z <= 1; z++) {
                uint32_t key = orc_hash(gp[0] + x, gp[1] + y, gp[2] + z);
                for (int e = mm_first(grid, key); e >= 0; e = mm_next(grid, key, e)) {
                    int j = grid->values[e];
                    if (index == j) continue;
                    const float *pj = pos + 3 * j;
                    float vij[3] = {position[0] - pj[0], position[1] - pj[1],
                                    position[2] - pj[2]};
                    hits += force_pair(sqrtf(dot3(vij, vij)), kernelRadius, viscosity, densityi,
                                       dens[j], pressurei, pres[j], velocityi, vel + 3 * j,
                                       1.0f / invMass[j], vij, fp, fv);
                }
                for (int e = mm_first(bgrid, key); e >= 0; e = mm_next(bgrid, key, e)) {
                    int j = bgrid->values[e];
                    const float *pj = bpos + 3 * j;
                    float vij[3] = {position[0] - pj[0], position[1] - pj[1],
                                    position[2] - pj[2]};
                    /* boundary: radius, mass, restDensity (BoundaryParticleComponent.cs:6-8) */
                    hits += force_pair(sqrtf(dot3(vij, vij)), kernelRadius, viscosity, densityi,
                                       bparams[3 * j + 2], pressurei, 0.0f, velocityi, zero3,
                                       bparams[3 * j + 1], vij, fp, fv);
                }
            }
    for (int a = 0; a < 3; a++) forces[3 * index + a] = fp[a] + fv[a] + fg[a];
    if (ncount) ncount[index] = hits;
}

/* ApplyForceJob.Execute, FluidSPHUtils.cs:105-124 (+ ApplyLinearImpulse) */
ORC_API void orc_apply_force(float *vel, const float *forces, const float *dens,
                             const float *invMass, int n, float dt)
{
    for (int i = 0; i < n; i++) {
        if (dens[i] == 0.0f) continue;
        for (int a = 0; a < 3; a++) {
            float impulse = forces[3 * i + a] / dens[i] / invMass[i] * dt;
            vel[3 * i + a] += impulse * invMass[i];
        }
    }
}

static void set_threads(int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}

/* density/pressure pass on its own (maps rebuilt inside) */
ORC_API int orc_density_pressure(const float *pos, const float *invMass, const float *params,
                                 int n, const float *bpos, int b, float gridSize, float rate,
                                 float *dens, float *pres, int32_t *ncount, int nthreads)
{
    MultiMap grid, bgrid;
    if (build_maps(pos, n, bpos, b, gridSize, &grid, &bgrid)) return -1;
    set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 12)
    for (int i = 0; i < n; i++)
        density_one(i, pos, invMass, params, &grid, bpos, &bgrid, gridSize, rate, dens, pres,
                    ncount);
    mm_free(&grid); mm_free(&bgrid);
    return 0;
}

/* force pass on its own, given densities/pressures */
ORC_API int orc_force(const float *pos, const float *invMass, const float *vel,
                      const float *params, int n, const float *bpos, const float *bparams, int b,
                      const float *dens, const float *pres, float gridSize, float rate,
                      float *forces, int32_t *ncount, int nthreads)
{
    MultiMap grid, bgrid;
    if (build_maps(pos, n, bpos, b, gridSize, &grid, &bgrid)) return -1;
    set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 12)
    for (int i = 0; i < n; i++)
        force_one(i, pos, invMass, vel, params, &grid, bpos, bparams, &bgrid, dens, pres,
                  gridSize, rate, forces, ncount);
    mm_free(&grid); mm_free(&bgrid);
    return 0;
}

/* FluidSPHByCPUSystem.OnUpdate, :258-333: maps -> density -> force -> apply (vel updated in
 * place).  dens/pres/forces are caller scratch (n, n, 3n floats) and hold the step's values. */
ORC_API int orc_sph_step(const float *pos, float *vel, const float *invMass, const float *params,
                         int n, const float *bpos, const float *bparams, int b, float gridSize,
                         float rate, float dt, float *dens, float *pres, float *forces,
                         int nthreads)
{
    MultiMap grid, bgrid;
    if (build_maps(pos, n, bpos, b, gridSize, &grid, &bgrid)) return -1; /* serial, as the IJob */
    set_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 12)
    for (int i = 0; i < n; i++)
        density_one(i, pos, invMass, params, &grid, bpos, &bgrid, gridSize, rate, dens, pres, 0);
#pragma omp parallel for schedule(dynamic, 12)
    for (int i = 0; i < n; i++)
        force_one(i, pos, invMass, vel, params, &grid, bpos, bparams, &bgrid, dens, pres,
                  gridSize, rate, forces, 0);
    orc_apply_force(vel, forces, dens, invMass, n, dt);
    mm_free(&grid); mm_free(&bgrid);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Row S contract: stable ascending sort of ids by key + cell (start,end) table */

typedef struct { uint32_t key; uint32_t id; } KeyId;
static int cmp_keyid(const void *a, const void *b)
{
    const KeyId *x = (const KeyId *)a, *y = (const KeyId *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

ORC_API int orc_stable_sort(const uint32_t *keys, int n, uint32_t *perm, uint32_t *sorted_keys)
{
    KeyId *a = (KeyId *)malloc(sizeof(KeyId) * (size_t)(n > 0 ? n : 1));
    if (!a) return -1;
    for (int i = 0; i < n; i++) { a[i].key = keys[i]; a[i].id = (uint32_t)i; }
    qsort(a, (size_t)n, sizeof(KeyId), cmp_keyid); /* (key,id) total order == stable by key */
    for (int i = 0; i < n; i++) { perm[i] = a[i].id; sorted_keys[i] = a[i].key; }
    free(a);
    return 0;
}

/* table[2k] = first sorted slot of key k, table[2k+1] = one past the last; (0,0) if absent.
 * Same content as the reference's ClearTable + MapTable (FluidSolver.compute:87-112). */
ORC_API void orc_cell_table(const uint32_t *sorted_keys, int n, int32_t *table, int ncells)
{
    memset(table, 0, sizeof(int32_t) * 2 * (size_t)ncells);
    for (int i = 0; i < n; i++) {
        uint32_t k = sorted_keys[i];
        if (i == 0 || k != sorted_keys[i - 1]) table[2 * k] = i;
        if (i == n - 1 || k != sorted_keys[i + 1]) table[2 * k + 1] = i + 1;
    }
}

/* Build-defined dense key (DESIGN.md "cell keys"): the quantised cell, shifted by an origin cell
 * and wrapped into a dims[0] x dims[1] x dims[2] periodic table (dims are multiples of the tile
 * 4 x 4 x 2); key = tile * 32 + local, tiles x-fastest over the table, cells x-fastest in a tile. */
static inline int32_t wrap1(int32_t c, int32_t g)
{
    int32_t m = c % g;
    return m < 0 ? m + g : m;
}
ORC_API void orc_dense_keys(const int32_t *cells, int n, const int32_t *origin,
                            const int32_t *dims, uint32_t *keys)
{
    const int TX = 4, TY = 4, TZ = 2;
    const int ntx = dims[0] / TX, nty = dims[1] / TY;
    for (int i = 0; i < n; i++) {
        int32_t x = wrap1(cells[3 * i] - origin[0], dims[0]);
        int32_t y = wrap1(cells[3 * i + 1] - origin[1], dims[1]);
        int32_t z = wrap1(cells[3 * i + 2] - origin[2], dims[2]);
        int32_t tile = ((z / TZ) * nty + (y / TY)) * ntx + (x / TX);
        int32_t local = ((z % TZ) * TY + (y % TY)) * TX + (x % TX);
        keys[i] = (uint32_t)(tile * (TX * TY * TZ) + local);
    }
}

/* Checks the one situation in which a dense-cell neighbour search and the reference's hashed
 * search can differ: two distinct cells within one particle's 27-cell neighbourhood (Chebyshev
 * distance <= 2) with equal FNV-1 hashes (the reference would then visit both cells twice).
 * Returns the number of such colliding cell pairs in the box [lo, hi] (inclusive, cell units). */
ORC_API long orc_count_near_hash_collisions(const int32_t *lo, const int32_t *hi)
{
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int32_t z = lo[2]; z <= hi[2]; z++)
        for (int32_t y = lo[1]; y <= hi[1]; y++)
            for (int32_t x = lo[0]; x <= hi[0]; x++) {
                uint32_t h0 = orc_hash(x, y, z);
                for (int dz = -2; dz <= 2; dz++)
                    for (int dy = -2; dy <= 2; dy++)
                        for (int dx = -2; dx <= 2; dx++) {
                            if (!dx && !dy && !dz) continue;
                            if (orc_hash(x + dx, y + dy, z + dz) == h0) bad++;
                        }
            }
    return bad / 2;
}

/* ------------------------------------------------------------------------------------------ */
/* Row I (build-defined): integration + box-collider response + rigid bodies.       */

typedef struct {
    float center[3];
    float rot[4]; /* unit quaternion x,y,z,w (body -> world) */
    float halfExtents[3];
    float linVel[3];
    float angVel[3];
    float invMass;       /* 0 => static */
    float invInertia[3]; /* body-space diagonal */
    float restitution;
    float friction; /* carried, unused: particle friction is 0 and the combine rule is min */
    float linDamping;
    float angDamping;
    float gravityFactor;
} OrcBox; /* 25 floats = 100 bytes, identical to SphBox in include/sphfluid.h */

static void quat_to_mat(const float *q, float m[9]) /* row-major, world = m * local */
{
    float x = q[0], y = q[1], z = q[2], w = q[3];
    m[0] = 1.0f - 2.0f * (y * y + z * z); m[1] = 2.0f * (x * y - z * w); m[2] = 2.0f * (x * z + y * w);
    m[3] = 2.0f * (x * y + z * w); m[4] = 1.0f - 2.0f * (x * x + z * z); m[5] = 2.0f * (y * z - x * w);
    m[6] = 2.0f * (x * z - y * w); m[7] = 2.0f * (y * z + x * w); m[8] = 1.0f - 2.0f * (x * x + y * y);
}
static inline void mat_mul(const float m[9], const float *v, float *o)
{
    o[0] = (m[0] * v[0] + m[1] * v[1]) + m[2] * v[2];
    o[1] = (m[3] * v[0] + m[4] * v[1]) + m[5] * v[2];
    o[2] = (m[6] * v[0] + m[7] * v[1]) + m[8] * v[2];
}
static inline void mat_tmul(const float m[9], const float *v, float *o)
{
    o[0] = (m[0] * v[0] + m[3] * v[1]) + m[6] * v[2];
    o[1] = (m[1] * v[0] + m[4] * v[1]) + m[7] * v[2];
    o[2] = (m[2] * v[0] + m[5] * v[1]) + m[8] * v[2];
}
static inline void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* particles: x += v*dt, then sphere(radius params[4i]) vs every box in order.
 * Swept: a particle that moves farther than half its radius in the step takes the step in
 * pieces no longer than r/2 (at most ORC_MAX_SUBSTEPS of them) and is tested against the boxes
 * after each, so that it is never carried through a box that is at least as thick as its radius:
 * a piece of r/2 takes a centre that is outside the box (even one that starts ON a face, as the
 * bottom layer of the demo lattice does) less than half the thickness deep, where the nearest face
 * is still the one it came in through.  The position of piece s is
 * xb + v (t_s - tb), (xb, tb) = where and when the particle last touched a box (the start of the
 * step at first): without a contact the end position is x + v dt exactly as without pieces.
 * body_impulse: K x 6 floats (linear impulse, angular impulse about the centre), accumulated. */
#define ORC_MAX_SUBSTEPS 256
static int box_dynamic_(const OrcBox *bx)
{
    return bx->invMass != 0.0f || bx->invInertia[0] != 0.0f || bx->invInertia[1] != 0.0f ||
           bx->invInertia[2] != 0.0f;
}
ORC_API void orc_integrate_particles(float *pos, float *vel, const float *invMass,
                                     const float *params, int n, const OrcBox *boxes, int nboxes,
                                     float dt, float particleRestitution, float *body_impulse)
{
    float *mats = (float *)malloc(sizeof(float) * 9 * (size_t)(nboxes > 0 ? nboxes : 1));
    for (int k = 0; k < nboxes; k++) quat_to_mat(boxes[k].rot, mats + 9 * k);
    if (body_impulse) memset(body_impulse, 0, sizeof(float) * 6 * (size_t)nboxes);
    for (int i = 0; i < n; i++) {
        float x[3], v[3], xb[3], tb = 0.0f;
        float r = params[4 * i];
        for (int a = 0; a < 3; a++) {
            v[a] = vel[3 * i + a];
            xb[a] = x[a] = pos[3 * i + a];
        }
        /* pieces: the rest of the step is cut into ceil(|v| rest / (r/2)) equal parts and the first
         * is taken; after every piece the cut is made afresh with the velocity the particle then
         * has (a moving box can hand a slow particle a large one in the middle of a step) */
        float tnow = 0.0f;
        for (int npieces = 0; tnow < dt; npieces++) {
        float ts = dt;
        if (nboxes > 0 && npieces < ORC_MAX_SUBSTEPS - 1) {
            float rest = dt - tnow, sp = sqrtf(dot3(v, v)) * rest, piece = 0.5f * r;
            if (sp > piece) { /* (never more parts than pieces are left: then they are longer) */
                float m = fminf(ceilf(sp / piece), (float)(ORC_MAX_SUBSTEPS - npieces));
                if (m > 1.0f) ts = tnow + rest / m;
            }
        }
        const float tau = ts - tb;
        int touched = 0;
        const float xs[3] = {x[0], x[1], x[2]}; /* where the piece starts: outside every static box */
        for (int a = 0; a < 3; a++) x[a] = xb[a] + v[a] * tau;
        /* every box in order; then, if that pushed the particle, the static boxes once more: a
         * push out of one box can end inside a box tested before it, and a particle may spend a
         * step inside a moving box but never inside a wall */
        for (int pass = 0; pass < 2; pass++) {
        if (pass == 1 && !touched) break;
        for (int k = 0; k < nboxes; k++) {
            const OrcBox *bx = &boxes[k];
            const float *m = mats + 9 * k;
            if (pass == 1 && box_dynamic_(bx)) continue;
            float d[3] = {x[0] - bx->center[0], x[1] - bx->center[1], x[2] - bx->center[2]};
            float p[3], q[3], delta[3], nl[3] = {0, 0, 0}, pen;
            mat_tmul(m, d, p);
            for (int a = 0; a < 3; a++) {
                q[a] = fminf(fmaxf(p[a], -bx->halfExtents[a]), bx->halfExtents[a]);
                delta[a] = p[a] - q[a];
            }
            float dist2 = dot3(delta, delta);
            if (dist2 > 0.0f) {
                if (!(dist2 < r * r)) continue;
                float dist = sqrtf(dist2);
                for (int a = 0; a < 3; a++) nl[a] = delta[a] / dist;
                pen = r - dist;
            } else { /* centre inside the box: leave through the nearest face; a STATIC box
                      * through the face the piece started beyond, if there is one (a push out of a
                      * moving box can carry a particle past a wall's mid-plane) */
                int best = 0;
                float bestd = bx->halfExtents[0] - fabsf(p[0]);
                for (int a = 1; a < 3; a++) {
                    float da = bx->halfExtents[a] - fabsf(p[a]);
                    if (da < bestd) { bestd = da; best = a; }
                }
                float sgn = p[best] < 0.0f ? -1.0f : 1.0f;
                if (!box_dynamic_(bx)) {
                    float ds[3] = {xs[0] - bx->center[0], xs[1] - bx->center[1], xs[2] - bx->center[2]};
                    float ps[3], beste = 0.0f;
                    mat_tmul(m, ds, ps);
                    for (int a = 0; a < 3; a++) {
                        float ex = fabsf(ps[a]) - bx->halfExtents[a];
                        if (ex > beste) {
                            beste = ex;
                            best = a;
                            sgn = ps[a] < 0.0f ? -1.0f : 1.0f;
                        }
                    }
                }
                nl[best] = sgn;
                bestd = bx->halfExtents[best] - sgn * p[best];
                pen = bestd + r;
            }
            touched = 1;
            float nw[3];
            mat_mul(m, nl, nw);
            for (int a = 0; a < 3; a++) x[a] = x[a] + nw[a] * pen;
            float rc[3], vb[3], vrel[3];
            for (int a = 0; a < 3; a++) rc[a] = (x[a] - nw[a] * r) - bx->center[a];
            cross3(bx->angVel, rc, vb);
            for (int a = 0; a < 3; a++) {
                vb[a] = bx->linVel[a] + vb[a];
                vrel[a] = v[a] - vb[a];
            }
            float vn = dot3(vrel, nw);
            if (vn < 0.0f) {
                float e = fmaxf(particleRestitution, bx->restitution);
                int dynamic = bx->invMass != 0.0f || bx->invInertia[0] != 0.0f ||
                              bx->invInertia[1] != 0.0f || bx->invInertia[2] != 0.0f;
                if (!dynamic) {
                    float sc = (1.0f + e) * vn;
                    for (int a = 0; a < 3; a++) v[a] = v[a] - sc * nw[a];
                } else {
                    float rxn[3], loc[3], w[3];
                    cross3(rc, nw, rxn);
                    mat_tmul(m, rxn, loc);
                    for (int a = 0; a < 3; a++) loc[a] = loc[a] * bx->invInertia[a];
                    mat_mul(m, loc, w);
                    float denom = (invMass[i] + bx->invMass) + dot3(rxn, w);
                    float j = -((1.0f + e) * vn) / denom;
                    float jp = j * invMass[i];
                    for (int a = 0; a < 3; a++) v[a] = v[a] + jp * nw[a];
                    if (body_impulse)
                        for (int a = 0; a < 3; a++) {
                            body_impulse[6 * k + a] -= j * nw[a];
                            body_impulse[6 * k + 3 + a] -= j * rxn[a];
                        }
                }
            }
        }
        }
        if (touched) {
            for (int a = 0; a < 3; a++) xb[a] = x[a];
            tb = ts;
        }
        tnow = ts;
        }
        for (int a = 0; a < 3; a++) {
            pos[3 * i + a] = x[a];
            vel[3 * i + a] = v[a];
        }
    }
    free(mats);
}

/* rigid bodies (row I, build-defined): (1) the particles' reaction, gravity, damping; (2) when
 * contacts != 0, contacts between boxes (below); (3) when bounded != 0 the corner contacts with the
 * planes of [bmin,bmax]; pose integration; a last positional clamp of the world AABB. */
static int box_dynamic(const OrcBox *bx)
{
    return bx->invMass != 0.0f || bx->invInertia[0] != 0.0f || bx->invInertia[1] != 0.0f ||
           bx->invInertia[2] != 0.0f;
}

static void body_velocities(OrcBox *bx, const float *imp6, float dt, const float *gravity)
{
        float m[9], loc[3], dw[3];
        quat_to_mat(bx->rot, m);
        mat_tmul(m, imp6 + 3, loc);
        for (int a = 0; a < 3; a++) loc[a] = loc[a] * bx->invInertia[a];
        mat_mul(m, loc, dw);
        float ld = fminf(fmaxf(1.0f - bx->linDamping * dt, 0.0f), 1.0f);
        float ad = fminf(fmaxf(1.0f - bx->angDamping * dt, 0.0f), 1.0f);
        for (int a = 0; a < 3; a++) {
            bx->linVel[a] = (bx->linVel[a] + imp6[a] * bx->invMass +
                             gravity[a] * bx->gravityFactor * dt) * ld;
            bx->angVel[a] = (bx->angVel[a] + dw[a]) * ad;
        }
}

static void body_rest(OrcBox *bx, float dt, int bounded, const float *bmin, const float *bmax)
{
        float m[9];
        quat_to_mat(bx->rot, m);
        /* contacts of the box's eight corners with the six planes of [bmin,bmax] (the tank's
         * walls), four passes of sequential impulses as the scene's four solver iterations
         * (ConvertedSubScene.unity:235): speculative, inelastic (restitution max(0,0),
         * :747-751, Wall.prefab), frictionless (geometric mean with the walls' 0, :742-746) */
        if (bounded) {
            const float *he = bx->halfExtents;
            for (int it = 0; it < 4; it++)
                for (int corner = 0; corner < 8; corner++) {
                    float hl[3] = {(corner & 1) ? he[0] : -he[0], (corner & 2) ? he[1] : -he[1],
                                   (corner & 4) ? he[2] : -he[2]};
                    float r[3];
                    mat_mul(m, hl, r);
                    for (int a = 0; a < 3; a++)
                        for (int side = 0; side < 2; side++) {
                            float *lv = bx->linVel, *av = bx->angVel;
                            float vc[3] = {lv[0] + (av[1] * r[2] - av[2] * r[1]),
                                           lv[1] + (av[2] * r[0] - av[0] * r[2]),
                                           lv[2] + (av[0] * r[1] - av[1] * r[0])};
                            float sgn = side ? -1.0f : 1.0f;
                            float pc = bx->center[a] + r[a];
                            float gap = side ? bmax[a] - pc : pc - bmin[a];
                            float need = -gap / dt - sgn * vc[a];
                            if (need > 0.0f) {
                                float n[3] = {0.0f, 0.0f, 0.0f};
                                n[a] = sgn;
                                float rxn[3] = {r[1] * n[2] - r[2] * n[1], r[2] * n[0] - r[0] * n[2],
                                                r[0] * n[1] - r[1] * n[0]};
                                float loc2[3], w2[3];
                                mat_tmul(m, rxn, loc2);
                                for (int b = 0; b < 3; b++) loc2[b] = loc2[b] * bx->invInertia[b];
                                mat_mul(m, loc2, w2);
                                float K = bx->invMass + ((rxn[0] * w2[0] + rxn[1] * w2[1]) + rxn[2] * w2[2]);
                                float j = need / K;
                                for (int b = 0; b < 3; b++) {
                                    lv[b] = lv[b] + (j * bx->invMass) * n[b];
                                    av[b] = av[b] + j * w2[b];
                                }
                            }
                        }
                }
        }
        for (int a = 0; a < 3; a++) bx->center[a] = bx->center[a] + bx->linVel[a] * dt;
        /* q += 0.5*dt*(w,0)*q, renormalised */
        float *q = bx->rot, *w = bx->angVel, hd = 0.5f * dt;
        float dq[4] = {hd * ((w[0] * q[3] + w[1] * q[2]) - w[2] * q[1]),
                       hd * ((w[1] * q[3] + w[2] * q[0]) - w[0] * q[2]),
                       hd * ((w[2] * q[3] + w[0] * q[1]) - w[1] * q[0]),
                       hd * -((w[0] * q[0] + w[1] * q[1]) + w[2] * q[2])};
        float nq[4], len2 = 0.0f;
        for (int a = 0; a < 4; a++) { nq[a] = q[a] + dq[a]; len2 += nq[a] * nq[a]; }
        float il = 1.0f / sqrtf(len2);
        for (int a = 0; a < 4; a++) q[a] = nq[a] * il;
        if (bounded) {
            quat_to_mat(bx->rot, m);
            for (int a = 0; a < 3; a++) {
                float ext = (fabsf(m[3 * a]) * bx->halfExtents[0] +
                             fabsf(m[3 * a + 1]) * bx->halfExtents[1]) +
                            fabsf(m[3 * a + 2]) * bx->halfExtents[2];
                if (bx->center[a] - ext < bmin[a]) {
                    bx->center[a] = bmin[a] + ext;
                    if (bx->linVel[a] < 0.0f) bx->linVel[a] = 0.0f;
                }
                if (bx->center[a] + ext > bmax[a]) {
                    bx->center[a] = bmax[a] - ext;
                    if (bx->linVel[a] > 0.0f) bx->linVel[a] = 0.0f;
                }
            }
        }
}

/* ---- contacts between boxes: a corner of X inside Y; normal = the face of Y that X's centre
 * lies beyond (in units of Y's half extents).  This is build-defined (row I).  Four Jacobi
 * passes with mass splitting; restitution max, bias 0.2*depth/dt, friction sqrt(muX*muY). */
static void w_inv_inertia(const OrcBox *B, const float *R, const float *v, float *o)
{
    float l[3];
    mat_tmul(R, v, l);
    for (int a = 0; a < 3; a++) l[a] = l[a] * B->invInertia[a];
    mat_mul(R, l, o);
}
static void cross3f(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static int corner_contact(const OrcBox *X, const float *RX, const OrcBox *Y, const float *RY, int ci,
                          float *n, float *depth, float *rX, float *rY)
{
    float hl[3] = {(ci & 1) ? X->halfExtents[0] : -X->halfExtents[0],
                   (ci & 2) ? X->halfExtents[1] : -X->halfExtents[1],
                   (ci & 4) ? X->halfExtents[2] : -X->halfExtents[2]};
    mat_mul(RX, hl, rX);
    for (int a = 0; a < 3; a++) rY[a] = (X->center[a] + rX[a]) - Y->center[a];
    float q[3];
    mat_tmul(RY, rY, q);
    for (int a = 0; a < 3; a++)
        if (!(Y->halfExtents[a] - fabsf(q[a]) > 0.0f)) return 0; /* outside (or on) the box */
    /* the face the corner came in through: the one X's centre is farthest outside of, in units
     * of Y's half extents (the nearest face would flip to a side face near an edge) */
    float dc[3] = {X->center[0] - Y->center[0], X->center[1] - Y->center[1], X->center[2] - Y->center[2]};
    float cq[3];
    mat_tmul(RY, dc, cq);
    int best = 0;
    float bests = fabsf(cq[0]) / Y->halfExtents[0];
    for (int a = 1; a < 3; a++) {
        float sa = fabsf(cq[a]) / Y->halfExtents[a];
        if (sa > bests) { best = a; bests = sa; }
    }
    float sgn = cq[best] < 0.0f ? -1.0f : 1.0f;
    float bestd = Y->halfExtents[best] - sgn * q[best];
    n[0] = RY[best] * sgn;
    n[1] = RY[3 + best] * sgn;
    n[2] = RY[6 + best] * sgn;
    *depth = bestd;
    return 1;
}
static float k_dir(const OrcBox *X, const float *RX, const OrcBox *Y, const float *RY, float nX,
                   float nY, const float *rX, const float *rY, const float *d)
{
    float cx[3], cy[3], ix[3], iy[3];
    cross3f(rX, d, cx);
    cross3f(rY, d, cy);
    w_inv_inertia(X, RX, cx, ix);
    w_inv_inertia(Y, RY, cy, iy);
    return (nX * X->invMass + nY * Y->invMass) +
           (nX * ((cx[0] * ix[0] + cx[1] * ix[1]) + cx[2] * ix[2]) +
            nY * ((cy[0] * iy[0] + cy[1] * iy[1]) + cy[2] * iy[2]));
}
static void contact_impulse(const OrcBox *X, const float *RX, const OrcBox *Y, const float *RY,
                            const float *vX, const float *vY, float nX, float nY, const float *n,
                            float depth, const float *rX, const float *rY, float dt, float *J)
{
    float wX[3], wY[3], vrel[3];
    cross3f(vX + 3, rX, wX);
    cross3f(vY + 3, rY, wY);
    for (int a = 0; a < 3; a++) vrel[a] = (vX[a] + wX[a]) - (vY[a] + wY[a]);
    float vn = (vrel[0] * n[0] + vrel[1] * n[1]) + vrel[2] * n[2];
    float e = fmaxf(X->restitution, Y->restitution), mu = sqrtf(X->friction * Y->friction);
    float wanted = fmaxf(vn < 0.0f ? -e * vn : 0.0f, 0.2f * depth / dt);
    float dvn = wanted - vn;
    J[0] = J[1] = J[2] = 0.0f;
    if (!(dvn > 0.0f)) return;
    float jn = dvn / k_dir(X, RX, Y, RY, nX, nY, rX, rY, n);
    float t[3];
    for (int a = 0; a < 3; a++) t[a] = vrel[a] - vn * n[a];
    float vt = sqrtf((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
    float jt = 0.0f;
    if (mu > 0.0f && vt > 1.0e-6f) {
        for (int a = 0; a < 3; a++) t[a] = t[a] / vt;
        jt = fminf(mu * jn, vt / k_dir(X, RX, Y, RY, nX, nY, rX, rY, t));
    }
    for (int a = 0; a < 3; a++) J[a] = jn * n[a] - jt * t[a];
}
static void box_aabb(const OrcBox *bx, const float *R, float *lo, float *hi)
{
    for (int a = 0; a < 3; a++) {
        float ext = (fabsf(R[3 * a]) * bx->halfExtents[0] + fabsf(R[3 * a + 1]) * bx->halfExtents[1]) +
                    fabsf(R[3 * a + 2]) * bx->halfExtents[2];
        lo[a] = bx->center[a] - ext;
        hi[a] = bx->center[a] + ext;
    }
}
static void body_contacts(OrcBox *boxes, int nboxes, float dt)
{
    float *R = (float *)malloc(sizeof(float) * 9 * (size_t)nboxes);
    float *lo = (float *)malloc(sizeof(float) * 3 * (size_t)nboxes);
    float *hi = (float *)malloc(sizeof(float) * 3 * (size_t)nboxes);
    float *nc = (float *)malloc(sizeof(float) * (size_t)nboxes);
    float *va = (float *)malloc(sizeof(float) * 6 * (size_t)nboxes);
    float *vb = (float *)malloc(sizeof(float) * 6 * (size_t)nboxes);
    for (int k = 0; k < nboxes; k++) {
        quat_to_mat(boxes[k].rot, R + 9 * k);
        box_aabb(&boxes[k], R + 9 * k, lo + 3 * k, hi + 3 * k);
        for (int a = 0; a < 3; a++) {
            va[6 * k + a] = boxes[k].linVel[a];
            va[6 * k + 3 + a] = boxes[k].angVel[a];
        }
    }
#define OVERLAP(a, b)                                                                            \
    (!(hi[3 * (a)] < lo[3 * (b)] || lo[3 * (a)] > hi[3 * (b)] || hi[3 * (a) + 1] < lo[3 * (b) + 1] ||  \
       lo[3 * (a) + 1] > hi[3 * (b) + 1] || hi[3 * (a) + 2] < lo[3 * (b) + 2] ||                  \
       lo[3 * (a) + 2] > hi[3 * (b) + 2]))
    for (int k = 0; k < nboxes; k++) {
        int cnt = 0;
        for (int j = 0; j < nboxes; j++) {
            if (j == k || !OVERLAP(k, j)) continue;
            if (!box_dynamic(&boxes[k]) && !box_dynamic(&boxes[j])) continue;
            float n[3], d, r1[3], r2[3];
            for (int ci = 0; ci < 8; ci++) {
                cnt += corner_contact(&boxes[k], R + 9 * k, &boxes[j], R + 9 * j, ci, n, &d, r1, r2);
                cnt += corner_contact(&boxes[j], R + 9 * j, &boxes[k], R + 9 * k, ci, n, &d, r1, r2);
            }
        }
        nc[k] = (float)(cnt > 1 ? cnt : 1);
    }
    for (int it = 0; it < 4; it++) {
        const float *vin = (it & 1) ? vb : va;
        float *vout = (it & 1) ? va : vb;
        for (int k = 0; k < nboxes; k++) {
            const OrcBox *A = &boxes[k];
            const float *RA = R + 9 * k;
            float v[6];
            for (int a = 0; a < 6; a++) v[a] = vin[6 * k + a];
            if (box_dynamic(A)) {
                float dl[3] = {0.0f, 0.0f, 0.0f}, da[3] = {0.0f, 0.0f, 0.0f};
                for (int j = 0; j < nboxes; j++) {
                    if (j == k || !OVERLAP(k, j)) continue;
                    const OrcBox *B = &boxes[j];
                    const float *RB = R + 9 * j;
                    float n[3], d, rX[3], rY[3], J[3], tq[3], w[3];
                    for (int ci = 0; ci < 8; ci++) {
                        if (corner_contact(A, RA, B, RB, ci, n, &d, rX, rY)) {
                            contact_impulse(A, RA, B, RB, vin + 6 * k, vin + 6 * j, nc[k], nc[j], n, d, rX, rY, dt, J);
                            cross3f(rX, J, tq);
                            w_inv_inertia(A, RA, tq, w);
                            for (int a = 0; a < 3; a++) { dl[a] = dl[a] + J[a] * A->invMass; da[a] = da[a] + w[a]; }
                        }
                        if (corner_contact(B, RB, A, RA, ci, n, &d, rX, rY)) {
                            contact_impulse(B, RB, A, RA, vin + 6 * j, vin + 6 * k, nc[j], nc[k], n, d, rX, rY, dt, J);
                            cross3f(rY, J, tq);
                            w_inv_inertia(A, RA, tq, w);
                            for (int a = 0; a < 3; a++) { dl[a] = dl[a] - J[a] * A->invMass; da[a] = da[a] - w[a]; }
                        }
                    }
                }
                for (int a = 0; a < 3; a++) { v[a] = v[a] + dl[a]; v[3 + a] = v[3 + a] + da[a]; }
            }
            for (int a = 0; a < 6; a++) vout[6 * k + a] = v[a];
        }
    }
#undef OVERLAP
    for (int k = 0; k < nboxes; k++)
        if (box_dynamic(&boxes[k]))
            for (int a = 0; a < 3; a++) {
                boxes[k].linVel[a] = va[6 * k + a];
                boxes[k].angVel[a] = va[6 * k + 3 + a];
            }
    free(R); free(lo); free(hi); free(nc); free(va); free(vb);
}

ORC_API void orc_integrate_bodies2(OrcBox *boxes, int nboxes, const float *body_impulse, float dt,
                                   const float *gravity, int bounded, const float *bmin,
                                   const float *bmax, int contacts)
{
    int any = 0;
    for (int k = 0; k < nboxes; k++)
        if (box_dynamic(&boxes[k])) {
            body_velocities(&boxes[k], body_impulse + 6 * k, dt, gravity);
            any = 1;
        }
    if (contacts && any && nboxes >= 2) body_contacts(boxes, nboxes, dt);
    for (int k = 0; k < nboxes; k++)
        if (box_dynamic(&boxes[k])) body_rest(&boxes[k], dt, bounded, bmin, bmax);
}

ORC_API void orc_integrate_bodies(OrcBox *boxes, int nboxes, const float *body_impulse, float dt,
                                  const float *gravity, int bounded, const float *bmin,
                                  const float *bmax)
{
    orc_integrate_bodies2(boxes, nboxes, body_impulse, dt, gravity, bounded, bmin, bmax, 0);
}

/* density field for rendering (no counterpart in the reference's code; its roadmap item
 * readme_zh.md:303): rho(x) = sum_j m_j 315/(64 pi h_j^9) (h_j^2 - d^2)^3 over ALL fluid particles
 * with d^2 < h_j^2, brute force, at the nodes origin + (ix,iy,iz)*spacing, x fastest. */
ORC_API void orc_sample_density(const float *pos, const float *inv_mass, const float *params, int n,
                                float rate, const float *origin, float spacing, const int *dims,
                                float *field)
{
    const float PI = 3.14159274f;
    long nodes = (long)dims[0] * dims[1] * dims[2];
#pragma omp parallel for schedule(static)
    for (long i = 0; i < nodes; i++) {
        int ix = (int)(i % dims[0]), iy = (int)((i / dims[0]) % dims[1]),
            iz = (int)(i / ((long)dims[0] * dims[1]));
        float x = origin[0] + (float)ix * spacing, y = origin[1] + (float)iy * spacing,
              z = origin[2] + (float)iz * spacing;
        float sum = 0.0f;
        for (int j = 0; j < n; j++) {
            float h = params[4 * j] * rate, h2 = h * h;
            float dx = x - pos[3 * j], dy = y - pos[3 * j + 1], dz = z - pos[3 * j + 2];
            float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < h2) {
                float mass = 1.0f / inv_mass[j];
                volatile float h9 = powf(h, 9.0f);
                volatile float den = 64.0f * PI * h9;
                float c = mass * 315.0f / den;
                float w = h2 - d2;
                sum += c * ((w * w) * w);
            }
        }
        field[i] = sum;
    }
}

ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
