// One thread chases a random cycle through a 1 GiB array (every hop an L2 miss past the Infinity Cache's reach): ticks of the 100 MHz
// real-time counter per hop = the latency of ONE dependent global load, alone or beside whatever else runs on the chip.
// hipcc -O3 -fPIC -shared --offload-arch=gfx950 experiments/r04/loaded_latency.hip -o experiments/r04/loaded_latency.so
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void chase_kernel(const uint32_t* __restrict__ next, uint32_t start, int hops, unsigned long long* __restrict__ ticks, uint32_t* __restrict__ end) {
    uint32_t i = start;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int h = 0; h < hops; ++h) i = __builtin_nontemporal_load(next + i);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    *ticks = t1 - t0;
    *end = i;
}
extern "C" int chase_launch(const void* next, uint32_t start, int hops, void* ticks, void* end, void* stream) {
    hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const uint32_t*)next, start, hops, (unsigned long long*)ticks, (uint32_t*)end);
    return (int)hipGetLastError();
}
