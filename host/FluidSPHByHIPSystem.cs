// FluidSPHByHIPSystem.cs -- the reference's GPU system with its compute-shader span replaced by
// libsphfluid.so (SURVEY.md 8f #1).  What FluidSPHByGPUSystem.OnUpdate does per frame
// (Assets/Scripts/FluidSPHByGPUSystem.cs:70-267) stays: gate on the settings singleton (:72-78),
// FindGridSizeJob (:80-89), packing the ECS components into flat arrays (:164-201).  The upload,
// the six dispatches and the blocking readback (:227-252) become ONE native call:
//
//   compat mode  sph_step_compat  -> Densities + Forces, then the project's own ApplyForceJob
//                (:254-262) and Unity Physics finish the frame exactly as before;
//   full mode    sph_step         -> the library also applies the force, integrates and resolves
//                the box colliders (DESIGN.md section 6); Translation and PhysicsVelocity are
//                written back and the particles are kept out of the physics world (PhysicsExclude).
//
// Gate: SimulationSettings.UseHIP (host/SimulationSettings.cs); FluidSPHByGPUSystem and
// FluidSPHByCPUSystem must test `!settings.UseHIP` beside their own gates (:73, CPU:261).
// Not compiled in the build image (no Unity / .NET there): tests/test_host_cs.py checks the native
// declarations it relies on against include/sphfluid.h.
using Unity.Collections;
using Unity.Collections.LowLevel.Unsafe;
using Unity.Entities;
using Unity.Jobs;
using Unity.Mathematics;
using Unity.Physics;
using Unity.Physics.Systems;
using Unity.Transforms;
using System;

[UpdateInGroup(typeof(FixedStepSimulationSystemGroup))]
[UpdateBefore(typeof(BuildPhysicsWorld))]
public unsafe partial class FluidSPHByHIPSystem : SystemBase
{
    EntityQuery m_Fluid, m_Boundary, m_Boxes, m_Untagged;
    IntPtr m_Ctx = IntPtr.Zero;
    int m_N = -1, m_B = -1, m_BoundaryVersion = -1;

    // flat host arrays, persistent, grown on demand (the reference reallocates on any change, :96-157)
    NativeArray<float3> m_Pos, m_Vel, m_Force, m_BPos;
    NativeArray<float> m_InvMass, m_Density;
    NativeArray<FluidParticleComponent> m_Particle;
    NativeArray<BoundaryParticleComponent> m_BParticle;
    NativeList<SphNative.SphBox> m_BoxList;

    protected override void OnCreate()
    {
        m_Fluid = GetEntityQuery(FluidSPHUtils.ParticleQueryComponentTypes);
        m_Boundary = GetEntityQuery(FluidSPHUtils.BoundaryQueryComponentTypes);
        m_Boxes = GetEntityQuery(new EntityQueryDesc
        {
            All = new ComponentType[] { ComponentType.ReadOnly<PhysicsCollider>(), ComponentType.ReadOnly<Translation>(), ComponentType.ReadOnly<Rotation>() },
            None = new ComponentType[] { typeof(FluidParticleComponent), typeof(BoundaryParticleComponent) },
        });
        m_Untagged = GetEntityQuery(new EntityQueryDesc
        {
            All = FluidSPHUtils.ParticleQueryComponentTypes,
            None = new ComponentType[] { typeof(PhysicsExclude) },
        });
        m_BoxList = new NativeList<SphNative.SphBox>(Allocator.Persistent);
        RequireForUpdate(m_Fluid);
        RequireSingletonForUpdate<SimulationSettings>();
    }

    protected override void OnDestroy()
    {
        if (m_Ctx != IntPtr.Zero) SphNative.sph_destroy(m_Ctx);
        m_Ctx = IntPtr.Zero;
        Free(ref m_Pos); Free(ref m_Vel); Free(ref m_Force); Free(ref m_BPos);
        Free(ref m_InvMass); Free(ref m_Density); Free(ref m_Particle); Free(ref m_BParticle);
        if (m_BoxList.IsCreated) m_BoxList.Dispose();
    }

    static void Free<T>(ref NativeArray<T> a) where T : struct { if (a.IsCreated) a.Dispose(); }

    // The context is opened with pinHostArrays = 1: the library registers the arrays it is given with
    // the driver and keeps them registered.  An array must therefore be released from the context
    // (sph_unpin_host) BEFORE it is disposed: freeing registered memory is undefined, and a new
    // array at the same address has other pages than the ones still pinned.
    void Fit<T>(ref NativeArray<T> a, int n) where T : struct
    {
        if (a.IsCreated && a.Length == n) return;
        if (a.IsCreated)
        {
            if (m_Ctx != IntPtr.Zero) SphNative.sph_unpin_host(m_Ctx, a.GetUnsafeReadOnlyPtr());
            a.Dispose();
        }
        a = new NativeArray<T>(n, Allocator.Persistent, NativeArrayOptions.UninitializedMemory);
    }

    bool OpenDevice()
    {
        if (m_Ctx != IntPtr.Zero) return true;
        var cfg = new SphNative.SphConfig { device = 0, maxCells = 0, enableTimings = 0, pinHostArrays = 1 };
        int rc = SphNative.sph_create(&cfg, out m_Ctx);
        if (rc == SphNative.SPH_OK) return true;
        // no gfx950 device: leave the frame to the project's own GPU / CPU systems
        UnityEngine.Debug.LogWarning("FluidSPHByHIPSystem disabled: " + SphNative.LastError(IntPtr.Zero));
        m_Ctx = IntPtr.Zero;
        return false;
    }

    protected override void OnUpdate()
    {
        var settings = GetSingleton<SimulationSettings>();
        if (!settings.UseHIP || !OpenDevice())
        {
            Enabled = false;
            return;
        }
        World.GetOrCreateSystem<FixedStepSimulationSystemGroup>().Timestep = 1f / settings.FPS;

        // cell size: the project's FindGridSizeJob, unchanged (FluidSPHUtils.cs:28-44)
        var gridSize = new NativeArray<float>(1, Allocator.TempJob);
        gridSize[0] = float.MaxValue;
        var gridJob = new FluidSPHUtils.FindGridSizeJob
        {
            particleTypeHandle = GetComponentTypeHandle<FluidParticleComponent>(true),
            gridSize = gridSize,
            kernelRadiusRate = FluidSPHUtils.KernelRadiusRate,
        }.Schedule(m_Fluid, Dependency);

        int n = m_Fluid.CalculateEntityCount(), b = m_Boundary.CalculateEntityCount();
        if (n != m_N)
        {
            Fit(ref m_Pos, n); Fit(ref m_Vel, n); Fit(ref m_Force, n);
            Fit(ref m_InvMass, n); Fit(ref m_Density, n); Fit(ref m_Particle, n);
            m_N = n;
        }

        // ECS -> flat arrays (the bounds the reference tracks here are not needed: the library
        // sizes its periodic cell table itself)
        var pos = m_Pos; var vel = m_Vel; var invMass = m_InvMass; var particle = m_Particle;
        var pack = Entities.WithAll<FluidParticleComponent>().ForEach(
            (int entityInQueryIndex, in Translation t, in PhysicsMass m, in PhysicsVelocity v, in FluidParticleComponent p) =>
            {
                pos[entityInQueryIndex] = t.Value;
                vel[entityInQueryIndex] = v.Linear;
                invMass[entityInQueryIndex] = m.InverseMass;
                particle[entityInQueryIndex] = p;
            }).ScheduleParallel(Dependency);

        // boundary particles are static: uploaded (and sorted on the device) only when the set changes
        int boundaryVersion = m_Boundary.GetCombinedComponentOrderVersion();
        bool boundaryChanged = b != m_B || boundaryVersion != m_BoundaryVersion;
        JobHandle packB = pack;
        if (boundaryChanged)
        {
            Fit(ref m_BPos, b); Fit(ref m_BParticle, b);
            var bpos = m_BPos; var bparticle = m_BParticle;
            packB = Entities.WithAll<BoundaryParticleComponent>().ForEach(
                (int entityInQueryIndex, in Translation t, in BoundaryParticleComponent p) =>
                {
                    bpos[entityInQueryIndex] = t.Value;
                    bparticle[entityInQueryIndex] = p;
                }).ScheduleParallel(pack);
        }
        Dependency = JobHandle.CombineDependencies(packB, gridJob);
        Dependency.Complete(); // the sync point of FluidSPHByGPUSystem.cs:203-204

        if (boundaryChanged)
        {
            SphNative.Check(SphNative.sph_set_boundary(m_Ctx, m_BPos.GetUnsafeReadOnlyPtr(), m_BParticle.GetUnsafeReadOnlyPtr(), b), m_Ctx, "sph_set_boundary");
            m_B = b; m_BoundaryVersion = boundaryVersion;
        }

        var par = new SphNative.SphParams
        {
            dt = Time.DeltaTime,
            kernelRadiusRate = FluidSPHUtils.KernelRadiusRate,
            gridSize = gridSize[0],
            particleRestitution = 0.5f, // Particle.prefab:142-146
            boundBodies = 0,
            bodyContacts = 1, // full mode: the walls are boxes, the BuoyancyObject rests on them
        };
        par.gravity[0] = 0f; par.gravity[1] = -9.81f; par.gravity[2] = 0f; // FluidSPHByCPUSystem.cs:140

        if (!settings.HIPFullStep)
        {
            // ---- compat: exactly the span :227-252, then the project's own ApplyForceJob
            SphNative.Check(SphNative.sph_step_compat(m_Ctx, &par, m_Pos.GetUnsafeReadOnlyPtr(), m_Vel.GetUnsafeReadOnlyPtr(),
                m_InvMass.GetUnsafeReadOnlyPtr(), m_Particle.GetUnsafeReadOnlyPtr(), n,
                m_Density.GetUnsafePtr(), m_Force.GetUnsafePtr()), m_Ctx, "sph_step_compat");
            Dependency = new FluidSPHUtils.ApplyForceJob
            {
                densities = m_Density,
                forces = m_Force,
                deltaTime = Time.DeltaTime,
                massTypeHandle = GetComponentTypeHandle<PhysicsMass>(true),
                velocityTypeHandle = GetComponentTypeHandle<PhysicsVelocity>(),
            }.ScheduleParallel(m_Fluid, Dependency);
        }
        else
        {
            // ---- full: colliders in, one whole step on the device, poses out
            CollectBoxes();
            SphNative.Check(SphNative.sph_set_colliders(m_Ctx, (SphNative.SphBox*)m_BoxList.GetUnsafePtr(), m_BoxList.Length), m_Ctx, "sph_set_colliders");
            SphNative.Check(SphNative.sph_step(m_Ctx, &par, m_Pos.GetUnsafePtr(), m_Vel.GetUnsafePtr(),
                m_InvMass.GetUnsafeReadOnlyPtr(), m_Particle.GetUnsafeReadOnlyPtr(), n, null, null), m_Ctx, "sph_step");
            SphNative.Check(SphNative.sph_get_colliders(m_Ctx, (SphNative.SphBox*)m_BoxList.GetUnsafePtr(), m_BoxList.Length), m_Ctx, "sph_get_colliders");
            var outPos = m_Pos; var outVel = m_Vel;
            Dependency = Entities.WithAll<FluidParticleComponent>().WithReadOnly(outPos).WithReadOnly(outVel).ForEach(
                (int entityInQueryIndex, ref Translation t, ref PhysicsVelocity v) =>
                {
                    t.Value = outPos[entityInQueryIndex];
                    v.Linear = outVel[entityInQueryIndex];
                }).ScheduleParallel(Dependency);
            WriteBackBoxes();
            // the particles were moved here: keep them out of BuildPhysicsWorld.  A structural
            // change, so only when some particle still lacks the tag (new particles), not per frame.
            if (m_Untagged.CalculateEntityCount() > 0)
            {
                Dependency.Complete();
                EntityManager.AddComponent<PhysicsExclude>(m_Untagged);
            }
        }
        gridSize.Dispose(Dependency);
    }

    // Box colliders of the scene (walls: Wall.prefab:99-107 static; BuoyancyObject:
    // ConvertedSubScene.unity:704-885 dynamic) -> SphBox, in query order.
    void CollectBoxes()
    {
        m_BoxList.Clear();
        var ents = m_Boxes.ToEntityArray(Allocator.Temp);
        foreach (var e in ents)
        {
            var col = EntityManager.GetComponentData<PhysicsCollider>(e);
            if (!col.IsValid || col.Value.Value.Type != ColliderType.Box) continue;
            var geo = ((BoxCollider*)col.ColliderPtr)->Geometry;
            var mat = ((BoxCollider*)col.ColliderPtr)->Material;
            float3 tr = EntityManager.GetComponentData<Translation>(e).Value;
            quaternion rot = EntityManager.GetComponentData<Rotation>(e).Value;
            quaternion q = math.mul(rot, geo.Orientation);
            float3 c = tr + math.mul(rot, geo.Center);
            var box = new SphNative.SphBox();
            box.center[0] = c.x; box.center[1] = c.y; box.center[2] = c.z;
            box.rot[0] = q.value.x; box.rot[1] = q.value.y; box.rot[2] = q.value.z; box.rot[3] = q.value.w;
            box.halfExtents[0] = 0.5f * geo.Size.x; box.halfExtents[1] = 0.5f * geo.Size.y; box.halfExtents[2] = 0.5f * geo.Size.z;
            box.restitution = mat.Restitution;
            box.friction = mat.Friction;
            if (EntityManager.HasComponent<PhysicsVelocity>(e) && EntityManager.HasComponent<PhysicsMass>(e))
            {
                var pv = EntityManager.GetComponentData<PhysicsVelocity>(e);
                var pm = EntityManager.GetComponentData<PhysicsMass>(e);
                float3 w = math.mul(rot, math.mul(pm.InertiaOrientation, pv.Angular)); // world
                box.linVel[0] = pv.Linear.x; box.linVel[1] = pv.Linear.y; box.linVel[2] = pv.Linear.z;
                box.angVel[0] = w.x; box.angVel[1] = w.y; box.angVel[2] = w.z;
                box.invMass = pm.InverseMass;
                box.invInertia[0] = pm.InverseInertia.x; box.invInertia[1] = pm.InverseInertia.y; box.invInertia[2] = pm.InverseInertia.z;
                if (EntityManager.HasComponent<PhysicsDamping>(e))
                {
                    var d = EntityManager.GetComponentData<PhysicsDamping>(e);
                    box.linDamping = d.Linear; box.angDamping = d.Angular;
                }
                box.gravityFactor = EntityManager.HasComponent<PhysicsGravityFactor>(e)
                    ? EntityManager.GetComponentData<PhysicsGravityFactor>(e).Value : 1f;
            }
            m_BoxList.Add(box);
        }
        ents.Dispose();
    }

    // dynamic boxes: pose and velocity back to the ECS (static ones are untouched by the library)
    void WriteBackBoxes()
    {
        var ents = m_Boxes.ToEntityArray(Allocator.Temp);
        int k = 0;
        foreach (var e in ents)
        {
            var col = EntityManager.GetComponentData<PhysicsCollider>(e);
            if (!col.IsValid || col.Value.Value.Type != ColliderType.Box) continue;
            var box = m_BoxList[k++];
            if (box.invMass == 0f) continue;
            var geo = ((BoxCollider*)col.ColliderPtr)->Geometry;
            quaternion q = new quaternion(box.rot[0], box.rot[1], box.rot[2], box.rot[3]);
            quaternion rot = math.mul(q, math.inverse(geo.Orientation));
            float3 c = new float3(box.center[0], box.center[1], box.center[2]);
            EntityManager.SetComponentData(e, new Rotation { Value = rot });
            EntityManager.SetComponentData(e, new Translation { Value = c - math.mul(rot, geo.Center) });
            var pm = EntityManager.GetComponentData<PhysicsMass>(e);
            float3 w = new float3(box.angVel[0], box.angVel[1], box.angVel[2]);
            EntityManager.SetComponentData(e, new PhysicsVelocity
            {
                Linear = new float3(box.linVel[0], box.linVel[1], box.linVel[2]),
                Angular = math.mul(math.inverse(math.mul(rot, pm.InertiaOrientation)), w),
            });
        }
        ents.Dispose();
    }
}
