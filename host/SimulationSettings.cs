// SimulationSettings with the switch for the HIP back-end beside the reference's UseGPU
// (Assets/Scripts/SimulationSettings.cs:6-10).  Replaces that file.
using System;
using Unity.Entities;

[GenerateAuthoringComponent]
[Serializable]
public struct SimulationSettings : IComponentData
{
    public bool UseGPU;
    public float FPS;
    public bool UseHIP;      // FluidSPHByHIPSystem (libsphfluid.so) takes the step
    public bool HIPFullStep; // false: compat span + the project's ApplyForceJob; true: whole step on the device
}
