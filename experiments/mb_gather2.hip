// experiment: cache policy variants for the row gather (not part of the product)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int POLICY, int UNROLL>  // 0 default, 1 nt all, 2 nt for non-hot (bitmap)
__global__ __launch_bounds__(256) void gm(const float* __restrict__ table, const int32_t* __restrict__ nbr,
    const int32_t* __restrict__ cnt, int k, int n, const uint32_t* __restrict__ hot, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < n; r += nwaves) {
        const int c = __builtin_amdgcn_readfirstlane(cnt[r]);
        int myid = lane < c ? nbr[(int64_t)r * k + lane] : 0;
        int myhot = 0;
        if (POLICY == 2) myhot = (hot[myid >> 5] >> (myid & 31)) & 1;
        f4 acc = {0, 0, 0, 0};
        for (int j0 = 0; j0 < c; j0 += UNROLL) {
            f4 t[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int jj = min(j0 + u, c - 1);
                const int id = __builtin_amdgcn_readlane(myid, jj);
                const f4* p = reinterpret_cast<const f4*>(table + (int64_t)id * 256) + lane;
                if (POLICY == 0) t[u] = *p;
                else if (POLICY == 1) t[u] = __builtin_nontemporal_load(p);
                else {
                    const int h = __builtin_amdgcn_readlane(myhot, jj);
                    if (h) t[u] = *p; else t[u] = __builtin_nontemporal_load(p);
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) if (j0 + u < c) acc += t[u];
        }
        const float inv = 1.f / c;
        f4 o = acc * inv;
        __builtin_nontemporal_store(o, reinterpret_cast<f4*>(out + (int64_t)r * 256) + lane);
    }
}
extern "C" void run(int policy, int unroll, int blocks, const float* table, const int32_t* nbr, const int32_t* cnt, int k, int n,
                    const uint32_t* hot, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define L(P, U) hipLaunchKernelGGL((gm<P, U>), dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, hot, out)
    if (unroll == 8) { if (policy == 0) L(0, 8); else if (policy == 1) L(1, 8); else L(2, 8); }
    else { if (policy == 0) L(0, 16); else if (policy == 1) L(1, 16); else L(2, 16); }
}
