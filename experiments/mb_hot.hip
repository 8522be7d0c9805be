// experiment: cache policy per row in the column-sliced gather -- ids with bit 31 set are "cold" (low-degree sources,
// unlikely to be sampled twice in a batch) and are loaded with a different policy so that they do not evict the hub
// rows from the XCD's L2.  Not product code.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ inline void load_row(f4& v, const float* p, bool cold) {
    if (MODE == 0 || !cold) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(v) : "v"(p) : "memory");
    } else if (MODE == 1) {
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(v) : "v"(p) : "memory");
    } else if (MODE == 2) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(v) : "v"(p) : "memory");
    } else if (MODE == 3) {
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "+v"(v) : "v"(p) : "memory");
    } else if (MODE == 4) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "+v"(v) : "v"(p) : "memory");
    } else {
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "+v"(v) : "v"(p) : "memory");
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void gm_hot(const float* __restrict__ table, const int32_t* __restrict__ nbr,
    const int32_t* __restrict__ cnt, int k, int n, float* __restrict__ out) {
    constexpr int SL = 16, NSLICE = 4, NPI = 4, U = 4;
    const int lane = threadIdx.x & 63;
    const int slice = blockIdx.x % NSLICE;
    const int wave = ((blockIdx.x / NSLICE) * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = ((gridDim.x / NSLICE) * blockDim.x) >> 6;
    const int grp = lane / SL, gl = lane % SL;
    const int coff = slice * SL * 4 + gl * 4;
    for (int r = wave; r < n; r += nwaves) {
        const int c = __builtin_amdgcn_readfirstlane(cnt[r]);
        const int myid = lane < c ? nbr[(int64_t)r * k + lane] : 0;
        f4 acc = {0, 0, 0, 0};
        f4 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = u * NPI + grp;
            const int idf = __shfl(myid, j < c ? j : 0, 64);
            const int id = idf & 0x7fffffff;
            t[u] = f4{0, 0, 0, 0};
            if (j < c) load_row<MODE>(t[u], table + (int64_t)id * 256 + coff, idf < 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) acc += t[u];
#pragma unroll
        for (int m = SL; m < 64; m <<= 1) {
            acc.x += __shfl_xor(acc.x, m, 64); acc.y += __shfl_xor(acc.y, m, 64);
            acc.z += __shfl_xor(acc.z, m, 64); acc.w += __shfl_xor(acc.w, m, 64);
        }
        if (grp == 0) *reinterpret_cast<f4*>(out + (int64_t)r * 256 + coff) = acc * (1.f / c);
    }
}

extern "C" void run_hot(int mode, int blocks, const float* table, const int32_t* nbr, const int32_t* cnt, int k, int n, float* out, hipStream_t st) {
    switch (mode) {
        case 0: hipLaunchKernelGGL(gm_hot<0>, dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out); break;
        case 1: hipLaunchKernelGGL(gm_hot<1>, dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out); break;
        case 2: hipLaunchKernelGGL(gm_hot<2>, dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out); break;
        case 3: hipLaunchKernelGGL(gm_hot<3>, dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out); break;
        case 4: hipLaunchKernelGGL(gm_hot<4>, dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out); break;
        default: hipLaunchKernelGGL(gm_hot<5>, dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out); break;
    }
}
