// SphNative.cs -- P/Invoke binding of libsphfluid.so (include/sphfluid.h), one declaration per
// SPH_API entry point and one [StructLayout] per struct of the header, field for field.
// tests/test_host_cs.py parses this file and checks sizes, field offsets and the export list
// against the header's ctypes mirror (fluidsimulation_amd/native.py); no C# toolchain exists in the
// build image, so this file has been checked that way and not compiled.
//
// Drop it into Assets/Scripts next to FluidSPHByGPUSystem.cs; libsphfluid.so goes to
// Assets/Plugins/x86_64 (Linux player, headless or not).  "Allow unsafe code" must be on.
using System;
using System.Runtime.InteropServices;

public static unsafe class SphNative
{
    const string Lib = "sphfluid";

    public const int SPH_OK = 0, SPH_E_INVALID = -1, SPH_E_NOMEM = -2, SPH_E_DEVICE = -3,
                     SPH_E_NODEVICE = -4, SPH_E_TOOLARGE = -5;
    public const int SPH_T_COUNT = 9;
    public const int SPH_DEBUG_HASHES = 1, SPH_DEBUG_COUNTS = 2, SPH_DEBUG_NO_LDS_TILE = 4;
    public const int SPH_SORT_NOALLOC_MAX = 1 << 20;

    // named device buffers of sph_read
    public enum Buf
    {
        Hashes = 0, Cells, Keys, Perm, SortedKeys, Table, Densities, Pressures, Forces,
        DensityCounts, ForceCounts, Positions, Velocities, BoundaryPerm, BoundaryTable,
        MaskCounts, MaskIdSums
    }

    // FluidParticleComponent (FluidParticleComponent.cs:9-12) has this layout: a
    // NativeArray<FluidParticleComponent> pointer can be passed where SphParticle* is expected.
    [StructLayout(LayoutKind.Sequential)]
    public struct SphParticle
    {
        public float radius;
        public float restDensity;
        public float viscosity;
        public float gasConstant;
    }

    // BoundaryParticleComponent (BoundaryParticleComponent.cs:6-8)
    [StructLayout(LayoutKind.Sequential)]
    public struct SphBoundaryParticle
    {
        public float radius;
        public float mass;
        public float restDensity;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SphParams
    {
        public float dt;
        public float kernelRadiusRate;
        public fixed float gravity[3];
        public float gridSize;
        public float particleRestitution;
        public int boundBodies;
        public fixed float bodyMin[3];
        public fixed float bodyMax[3];
        public int bodyContacts;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SphBox
    {
        public fixed float center[3];
        public fixed float rot[4];
        public fixed float halfExtents[3];
        public fixed float linVel[3];
        public fixed float angVel[3];
        public float invMass;
        public fixed float invInertia[3];
        public float restitution;
        public float friction;
        public float linDamping;
        public float angDamping;
        public float gravityFactor;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SphConfig
    {
        public int device;
        public int maxCells;
        public int enableTimings;   // 0 off, 1 every stage, 2 the two neighbour kernels only
        public int pinHostArrays;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SphGrid
    {
        public fixed int origin[3];
        public fixed int dims[3];
        public int keyBits;
        public int cells;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SphTimings
    {
        public fixed double ms[9];
        public fixed long launches[9];
        public long steps;
    }

    // ---- life cycle
    [DllImport(Lib)] public static extern int sph_create(SphConfig* cfg, out IntPtr ctx);
    [DllImport(Lib)] public static extern int sph_destroy(IntPtr ctx);
    [DllImport(Lib)] public static extern IntPtr sph_last_error(IntPtr ctx);
    [DllImport(Lib)] public static extern int sph_device_name(IntPtr ctx, byte* name, UIntPtr cap, int* computeUnits);
    [DllImport(Lib)] public static extern int sph_set_debug(IntPtr ctx, int flags);
    [DllImport(Lib)] public static extern int sph_synchronize(IntPtr ctx);

    // ---- host-side helpers
    [DllImport(Lib)] public static extern float sph_find_grid_size(void* particles, int* chunkFirst, int nchunks, float kernelRadiusRate);

    // ---- scene state
    [DllImport(Lib)] public static extern int sph_set_boundary(IntPtr ctx, void* bpos, void* bparams, int B);
    [DllImport(Lib)] public static extern int sph_set_colliders(IntPtr ctx, SphBox* boxes, int K);
    [DllImport(Lib)] public static extern int sph_get_colliders(IntPtr ctx, SphBox* boxesOut, int K);
    [DllImport(Lib)] public static extern int sph_set_grid(IntPtr ctx, SphGrid* grid);
    [DllImport(Lib)] public static extern int sph_get_grid(IntPtr ctx, SphGrid* gridOut);
    [DllImport(Lib)] public static extern int sph_upload(IntPtr ctx, void* pos, void* vel, void* invMass, void* particles, int N, float gridSize);
    [DllImport(Lib)] public static extern int sph_download(IntPtr ctx, void* posOut, void* velOut);
    [DllImport(Lib)] public static extern int sph_unpin_host(IntPtr ctx, void* array);
    [DllImport(Lib)] public static extern int sph_pin_stats(IntPtr ctx, int* stats3);
    [DllImport(Lib)] public static extern int sph_seed_volume(IntPtr ctx, float* localToWorld, float* boundsMin, float* boundsSize, float radius, SphParticle* particle, float invMass, float gridSize, int append, int* added);

    // ---- stages (one per dispatch of FluidSPHByGPUSystem.cs:242-247, then the CPU-side jobs)
    [DllImport(Lib)] public static extern int sph_hash_position(IntPtr ctx, SphParams* p);
    [DllImport(Lib)] public static extern int sph_radix_sort(IntPtr ctx);
    [DllImport(Lib)] public static extern int sph_map_table(IntPtr ctx);
    [DllImport(Lib)] public static extern int sph_reorder(IntPtr ctx);
    [DllImport(Lib)] public static extern int sph_compute_density_and_pressure(IntPtr ctx, SphParams* p);
    [DllImport(Lib)] public static extern int sph_compute_force(IntPtr ctx, SphParams* p);
    [DllImport(Lib)] public static extern int sph_apply_force(IntPtr ctx, SphParams* p);
    [DllImport(Lib)] public static extern int sph_integrate(IntPtr ctx, SphParams* p);
    [DllImport(Lib)] public static extern int sph_step_resident(IntPtr ctx, SphParams* p, int nsteps);

    // ---- one-call host round trips
    [DllImport(Lib)] public static extern int sph_step(IntPtr ctx, SphParams* p, void* pos, void* vel, void* invMass, void* particles, int N, void* densitiesOut, void* forcesOut);
    [DllImport(Lib)] public static extern int sph_step_compat(IntPtr ctx, SphParams* p, void* pos, void* vel, void* invMass, void* particles, int N, void* densitiesOut, void* forcesOut);

    // ---- parity / inspection
    [DllImport(Lib)] public static extern int sph_read(IntPtr ctx, int buffer, void* dst, UIntPtr bytes);
    [DllImport(Lib)] public static extern int sph_find_bounds(IntPtr ctx, float gridSize, float* mnOut, float* mxOut);
    [DllImport(Lib)] public static extern int sph_sample_density(IntPtr ctx, SphParams* p, float* origin, float spacing, int* dims, float* fieldOut);
    [DllImport(Lib)] public static extern int sph_sort_pairs(IntPtr ctx, uint* keys, int n, int keyBits, uint* permOut, uint* sortedKeysOut);
    [DllImport(Lib)] public static extern int sph_sort_pairs_noalloc(IntPtr ctx, void* keysDev, int n, int keyBits, void* permOutDev, void* sortedKeysOutDev);
    [DllImport(Lib)] public static extern int sph_get_timings(IntPtr ctx, SphTimings* timingsOut);
    [DllImport(Lib)] public static extern int sph_reset_timings(IntPtr ctx);

    // ---- slab decomposition across GPUs (one process per GPU; include/sphfluid.h, section 8e)
    [DllImport(Lib)] public static extern int sph_slab_upload(IntPtr ctx, void* pos, void* vel, uint* materialIndex, uint* ids, int N, float* materials, int M, float gridSize, int capacity);
    [DllImport(Lib)] public static extern int sph_slab_pack(IntPtr ctx, float xlo, float xhi, float halo, void* leftDev, void* rightDev, int capRecords, int* counts4);
    [DllImport(Lib)] public static extern int sph_slab_import(IntPtr ctx, void* recordsDev, int count, int transfers);
    [DllImport(Lib)] public static extern int sph_slab_step_particles(IntPtr ctx, SphParams* p);
    [DllImport(Lib)] public static extern int sph_slab_get_impulses(IntPtr ctx, float* impulsesOut, int K);
    [DllImport(Lib)] public static extern int sph_slab_step_bodies(IntPtr ctx, SphParams* p, float* impulses, int K);
    [DllImport(Lib)] public static extern int sph_slab_counts(IntPtr ctx, int* counts4);
    [DllImport(Lib)] public static extern int sph_slab_download(IntPtr ctx, void* posOut, void* velOut, void* densityOut, uint* idsOut, int cap, int* count);
    [DllImport(Lib)] public static extern int sph_slab_comm_init(IntPtr ctx, int rank, int world, byte* uniqueId, int uniqueIdBytes);
    [DllImport(Lib)] public static extern int sph_slab_comm_unique_id(byte* uniqueIdOut, int capBytes);
    [DllImport(Lib)] public static extern int sph_slab_exchange(IntPtr ctx, float xlo, float xhi, float halo, int* counts4);
    [DllImport(Lib)] public static extern int sph_slab_exchange_reserve(IntPtr ctx, int records);
    [DllImport(Lib)] public static extern int sph_slab_comm_destroy(IntPtr ctx);

    public static string LastError(IntPtr ctx) => Marshal.PtrToStringAnsi(sph_last_error(ctx));

    public static void Check(int rc, IntPtr ctx, string what)
    {
        if (rc != SPH_OK) throw new InvalidOperationException($"libsphfluid: {what} failed ({rc}): {LastError(ctx)}");
    }
}
