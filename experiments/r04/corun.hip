// What does a kernel that sits on the CUs like the layer-1 contraction TAKE from a kernel that reads like the layer-1 gather?
// gather_like: lane groups of 8 lanes x 16 B fetch 16 random 128-byte granules of a 1 GiB table per unit (all in flight), sum, store 128 B: a stand-in
//              for gather_mean_rows_kernel<8,16,false> (same request shape, no L2 reuse).
// holder<MODE>: 512-thread blocks with the contraction's footprint (>= 160 VGPRs, 117 KB of LDS), one per CU on `blocks` CUs, busy for `iters` rounds with
//              ONE kind of work: 0 nothing (s_sleep), 1 ds_read_b128 at full rate, 2 v_mfma_f32_32x32x16_bf16 back to back, 3 streaming global loads,
//              4 plain VALU (v_fma), 5 LDS reads + MFMA interleaved like a tile loop.
// hipcc -O3 -fPIC -shared --offload-arch=gfx950 experiments/r04/corun.hip -o experiments/r04/corun.so
#include <hip/hip_runtime.h>
#include <stdint.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

extern "C" __global__ __launch_bounds__(256) void gather_like(const float* __restrict__ table, uint32_t granules, float* __restrict__ out, int units) {
    const int lane = threadIdx.x & 63, grp = lane >> 3, gl = lane & 7;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int u0 = wave * 8; u0 < units; u0 += nwaves * 8) {
        const int u = u0 + grp;
        f32x4 t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t g = mix((uint32_t)u * 16u + (uint32_t)j) % granules;
            t[j] = *reinterpret_cast<const f32x4*>(table + (size_t)g * 32 + gl * 4);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += t[j];
        if (u < units) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(out + (size_t)u * 32 + gl * 4));
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void holder(const float* __restrict__ stream_src, size_t stream_floats, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 117 KB requested at launch
    const int tid = threadIdx.x;
    asm volatile("v_mov_b32 v167, 0" ::: "v167");                        // the allocation is by the highest register named: 168 VGPRs per wave, as the contraction
    float keep[96];                                                       // the register footprint of the W planes + accumulators
#pragma unroll
    for (int i = 0; i < 96; ++i) keep[i] = (float)(tid + i);
    for (int i = tid; i < 117 * 256; i += 512) lds[i] = (float)i;
    __syncthreads();
    f32x16 acc = {0.f};
    bf16x8 a = {(__bf16)1.f}, b = {(__bf16)1.f};
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            __builtin_amdgcn_s_sleep(64);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s += *reinterpret_cast<const f32x4*>(lds + ((tid * 4 + r * 2048 + it * 64) % (117 * 256 - 4)));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else if constexpr (MODE == 3) {
            const size_t base = ((size_t)blockIdx.x * iters + it) * 512 * 4 * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(stream_src + (base + (size_t)r * 2048 + tid * 4) % (stream_floats - 4)));
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 64; ++r) keep[r] = __builtin_fmaf(keep[r], 1.0001f, 0.5f);
        } else {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                s += *reinterpret_cast<const f32x4*>(lds + ((tid * 4 + r * 2048 + it * 64) % (117 * 256 - 4)));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
        }
    }
    float r = s[0] + s[1] + s[2] + s[3] + acc[0] + acc[5];
#pragma unroll
    for (int i = 0; i < 96; ++i) r += keep[i];
    if (r == 12345.678f) sink[tid] = r;                                  // keeps everything live
}

extern "C" int launch_gather(const void* table, uint32_t granules, void* out, int units, int blocks, void* st) {
    hipLaunchKernelGGL(gather_like, dim3(blocks), dim3(256), 0, (hipStream_t)st, (const float*)table, granules, (float*)out, units);
    return (int)hipGetLastError();
}
template <int M> static int lh(const void* src, size_t n, void* sink, int iters, int blocks, void* st) {
    static bool cfg = false;
    if (!cfg) { (void)hipFuncSetAttribute((const void*)holder<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 117 * 1024); cfg = true; }
    hipLaunchKernelGGL(holder<M>, dim3(blocks), dim3(512), 117 * 1024, (hipStream_t)st, (const float*)src, n, (float*)sink, iters);
    return (int)hipGetLastError();
}
extern "C" int launch_holder(int mode, const void* src, size_t n, void* sink, int iters, int blocks, void* st) {
    switch (mode) { case 0: return lh<0>(src, n, sink, iters, blocks, st); case 1: return lh<1>(src, n, sink, iters, blocks, st); case 2: return lh<2>(src, n, sink, iters, blocks, st);
                    case 3: return lh<3>(src, n, sink, iters, blocks, st); case 4: return lh<4>(src, n, sink, iters, blocks, st); default: return lh<5>(src, n, sink, iters, blocks, st); }
}
