// checks the DPP wave scan used by k_plan against a serial scan (run once on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../fluidsimulation_amd/csrc/sph_neighbors.h"
__global__ void k(const int *in, int *out) { out[threadIdx.x] = wave_incl_scan(in[threadIdx.x]); }
int main()
{
    int h[64], r[64], *di, *dr;
    for (int i = 0; i < 64; i++) h[i] = (i * 7919 + 13) % 23;
    (void)hipMalloc(&di, 256); (void)hipMalloc(&dr, 256);
    (void)hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dr);
    (void)hipMemcpy(r, dr, 256, hipMemcpyDeviceToHost);
    int s = 0, bad = 0;
    for (int i = 0; i < 64; i++) { s += h[i]; if (r[i] != s) { bad++; printf("lane %d got %d want %d\n", i, r[i], s); } }
    printf(bad ? "SCAN BAD\n" : "SCAN OK\n");
    return bad != 0;
}
