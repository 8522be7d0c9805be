// experiment: two-phase column-sliced gather.  Phase 1: every wave walks ALL its destination rows gathering only the HOT
// sources (bit 31 of the id clear = top-K by degree), partial sums parked in LDS; phase 2: the same rows again for the cold
// sources.  All blocks are co-resident and start together, so for the first ~40 % of the launch the XCD L2s see only the
// hot set (which fits) instead of a mix in which cold rows keep evicting it.  Not product code.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int JMAX, bool TWO_PHASE>
__global__ __launch_bounds__(256) void gm_phase(const float* __restrict__ table, const int32_t* __restrict__ nbr,
    const int32_t* __restrict__ cnt, int k, int n, float* __restrict__ out) {
    constexpr int SL = 16, NSLICE = 4, NPI = 4, U = 2;
    __shared__ f4 part[TWO_PHASE ? 4 * JMAX * SL : 1];          // [wave][row j][lane in slice]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int slice = blockIdx.x % NSLICE;
    const int wave = ((blockIdx.x / NSLICE) * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = ((gridDim.x / NSLICE) * blockDim.x) >> 6;
    const int grp = lane / SL, gl = lane % SL;
    const int coff = slice * SL * 4 + gl * 4;
    for (int phase = TWO_PHASE ? 0 : 1; phase < 2; ++phase) {
        int j = 0;
        for (int r = wave; r < n; r += nwaves, ++j) {
            const int c = __builtin_amdgcn_readfirstlane(cnt[r]);
            const int myid = lane < c ? nbr[(int64_t)r * k + lane] : 0;
            f4 acc = {0, 0, 0, 0};
            for (int j0 = 0; j0 < c; j0 += NPI * U) {
                f4 t[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int jj = j0 + u * NPI + grp;
                    const int idf = __shfl(myid, jj < c ? jj : 0, 64);
                    const bool cold = idf < 0;
                    const bool take = jj < c && (!TWO_PHASE || j >= JMAX || cold == (phase == 1));
                    const int id = idf & 0x7fffffff;
                    t[u] = take ? *reinterpret_cast<const f4*>(table + (int64_t)id * 256 + coff) : f4{0, 0, 0, 0};
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc += t[u];
            }
            if (TWO_PHASE && phase == 0 && j >= JMAX) continue;       // rows beyond the LDS budget: single phase (phase 1 takes all)
#pragma unroll
            for (int m = SL; m < 64; m <<= 1) {
                acc.x += __shfl_xor(acc.x, m, 64); acc.y += __shfl_xor(acc.y, m, 64);
                acc.z += __shfl_xor(acc.z, m, 64); acc.w += __shfl_xor(acc.w, m, 64);
            }
            if (TWO_PHASE && phase == 0) {
                if (grp == 0) part[(w * JMAX + j) * SL + gl] = acc;
            } else if (grp == 0) {
                if (TWO_PHASE && j < JMAX) acc += part[(w * JMAX + j) * SL + gl];
                *reinterpret_cast<f4*>(out + (int64_t)r * 256 + coff) = acc * (1.f / c);
            }
        }
    }
}

extern "C" void run_phase(int mode, int blocks, const float* table, const int32_t* nbr, const int32_t* cnt, int k, int n, float* out, hipStream_t st) {
    if (mode == 0) hipLaunchKernelGGL((gm_phase<12, false>), dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out);
    else hipLaunchKernelGGL((gm_phase<12, true>), dim3(blocks), dim3(256), 0, st, table, nbr, cnt, k, n, out);
}
