// experiment: feature table stored slice-major ([4][N][64] floats: each XCD pair reads one contiguous 256 MB array) instead
// of row-major [N][256] -- same column-sliced gather, only the address arithmetic differs.  Not product code.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gm_layout(const float* __restrict__ table, int64_t slice_stride, int64_t ld,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int k, int n, float* __restrict__ out, int hot_k) {
    constexpr int SL = 16, NSLICE = 4, NPI = 4, U = 2;
    const int lane = threadIdx.x & 63;
    const int slice = blockIdx.x % NSLICE;
    const int wave = ((blockIdx.x / NSLICE) * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = ((gridDim.x / NSLICE) * blockDim.x) >> 6;
    const int grp = lane / SL, gl = lane % SL;
    const float* base = table + slice * slice_stride + gl * 4;
    const int coff = slice * SL * 4 + gl * 4;
    for (int r = wave; r < n; r += nwaves) {
        const int c = __builtin_amdgcn_readfirstlane(cnt[r]);
        const int myid = lane < c ? nbr[(int64_t)r * k + lane] : 0;
        f4 acc = {0, 0, 0, 0};
        for (int j0 = 0; j0 < c; j0 += NPI * U) {
            f4 t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j0 + u * NPI + grp;
                const int id = __shfl(myid, jj < c ? jj : 0, 64);
                const f4* src = reinterpret_cast<const f4*>(base + (int64_t)id * ld);
                t[u] = f4{0, 0, 0, 0};
                if (jj < c) t[u] = (id >= hot_k) ? __builtin_nontemporal_load(src) : *src;      // degree-sorted ids: id < hot_k = hub row
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += t[u];
        }
#pragma unroll
        for (int m = SL; m < 64; m <<= 1) {
            acc.x += __shfl_xor(acc.x, m, 64); acc.y += __shfl_xor(acc.y, m, 64);
            acc.z += __shfl_xor(acc.z, m, 64); acc.w += __shfl_xor(acc.w, m, 64);
        }
        if (grp == 0) __builtin_nontemporal_store(acc * (1.f / c), reinterpret_cast<f4*>(out + (int64_t)r * 256 + coff));
    }
}
extern "C" void run_layout(const float* table, int64_t slice_stride, int64_t ld, const int32_t* nbr, const int32_t* cnt, int k, int n, float* out, hipStream_t st, int hot_k) {
    hipLaunchKernelGGL(gm_layout, dim3(2048), dim3(256), 0, st, table, slice_stride, ld, nbr, cnt, k, n, out, hot_k);
}
