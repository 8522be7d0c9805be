// experiment: column-sliced gather (each XCD owns a column slice of every row) -- not product code
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
// SL = lanes per neighbour row slice (8 -> 128-B slices, 8 slices; 16 -> 256-B slices, 4 slices)
template <int SL>
__global__ __launch_bounds__(256) void gm_sliced(const float* __restrict__ table, const int32_t* __restrict__ nbr,
    const int32_t* __restrict__ cnt, int k, int n, float* __restrict__ out) {
    constexpr int NSLICE = 64 / SL;          // slices per 256-col row
    constexpr int NPI = 64 / SL;             // neighbours per wave-instruction
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
        f4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = u * NPI + grp;
            const int id = __shfl(myid, j < c ? j : 0, 64);
            t[u] = (j < c && u * NPI < c) ? *reinterpret_cast<const f4*>(table + (int64_t)id * 256 + coff) : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
        // reduce across the NPI lane groups
#pragma unroll
        for (int m = SL; m < 64; m <<= 1) {
            acc.x += __shfl_xor(acc.x, m, 64); acc.y += __shfl_xor(acc.y, m, 64);
            acc.z += __shfl_xor(acc.z, m, 64); acc.w += __shfl_xor(acc.w, m, 64);
        }
        if (grp == 0) {
            const float inv = 1.f / c;
            *reinterpret_cast<f4*>(out + (int64_t)r * 256 + coff) = acc * inv;
        }
    }
}
extern "C" void run_sliced(int sl, int blocks, const float* table, const int32_t* nbr, const int32_t* cnt, int k, int n, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (sl == 8) hipLaunchKernelGGL(gm_sliced<8>, dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out);
    else if (sl == 32) hipLaunchKernelGGL(gm_sliced<32>, dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out);
    else hipLaunchKernelGGL(gm_sliced<16>, dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out);
}
