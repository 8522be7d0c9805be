// linear_act: out = act( [self | agg] . W^T ) without materialising the concat.
//
// Replaces encoders.py:49-62 (self feature fetch, torch.cat, weight.mm(combined.t()),
// relu / sigmoid).  General shapes (any dim, any out_dim): the two-launch form's second
// kernel and the fallback for layers the fused kernel does not cover (Cora 1433->50,
// Pubmed 500->50).  fp32-input MFMA v_mfma_f32_32x32x2_f32: exact fp32 products with
// fp32 accumulation, the same arithmetic class as the reference's sgemm.
//
// Tile: 64 rows x 128 outputs per 256-thread block, K streamed in 32-wide panels through
// LDS.  Wave w owns output columns [32w, 32w+32) and two 32x32 accumulators (rows 0-31,
// 32-63).  LDS rows are padded to 33 floats: the MFMA operand read is ds_read_b32 with
// lanes 0-31 on 32 different rows at one k, so stride 33 puts them on 32 distinct banks.
#include "sage_internal.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 64, BN = 128, BK = 32, LDP = BK + 1;

__global__ __launch_bounds__(256) void linear_act_kernel(
    const float* __restrict__ self_tab, int64_t ld_self, const int32_t* __restrict__ self_index,
    const float* __restrict__ agg, int64_t ld_agg, int dim,
    const float* __restrict__ W, int64_t ldw, int out_dim, int act,
    int n, const int32_t* __restrict__ n_dev, float* __restrict__ out, int64_t ldo, int n_off, sage_finish_t fin) {
    __shared__ float smem[(BM + BN) * LDP];
    float* As = smem;
    float* Bs = smem + BM * LDP;
    int nn = n;
    if (n_dev) nn = min(*n_dev + n_off, n);
    const int m0 = blockIdx.x * BM;
    if (m0 >= nn) {
        sage_finish_block(fin, (int)(gridDim.x * gridDim.y));
        return;
    }
    const int nb0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ds = self_tab ? dim : 0;
    const int K = ds + dim;
    const int wave_n0 = nb0 + wave * 32;
    const bool wave_active = wave_n0 < out_dim;

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int kk = tid & 31, rr = tid >> 5;   // this thread's column / first row inside a panel
    for (int k0 = 0; k0 < K; k0 += BK) {
        const int gk = k0 + kk;
#pragma unroll
        for (int i = 0; i < BM / 8; ++i) {
            const int row = rr + 8 * i, gr = m0 + row;
            float v = 0.f;
            if (gr < nn && gk < K) {
                if (gk < ds) {
                    const int64_t sr = self_index ? (int64_t)self_index[gr] : (int64_t)gr;
                    v = self_tab[sr * ld_self + gk];
                } else {
                    v = agg[(int64_t)gr * ld_agg + (gk - ds)];
                }
            }
            As[row * LDP + kk] = v;
        }
#pragma unroll
        for (int i = 0; i < BN / 8; ++i) {
            const int row = rr + 8 * i, gn = nb0 + row;
            Bs[row * LDP + kk] = (gn < out_dim && gk < K) ? W[(int64_t)gn * ldw + gk] : 0.f;
        }
        __syncthreads();
        if (wave_active) {
            const float* a0p = As + (lane & 31) * LDP + (lane >> 5);
            const float* a1p = a0p + 32 * LDP;
            const float* bp = Bs + (wave * 32 + (lane & 31)) * LDP + (lane >> 5);
#pragma unroll
            for (int q = 0; q < BK; q += 2) {
                const float b = bp[q];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0p[q], b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1p[q], b, acc1, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (wave_active) {
        const int col = wave_n0 + (lane & 31);
        if (col < out_dim) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int g0 = m0 + row, g1 = m0 + 32 + row;
                if (g0 < nn) out[(int64_t)g0 * ldo + col] = sage_activate(acc0[reg], act);
                if (g1 < nn) out[(int64_t)g1 * ldo + col] = sage_activate(acc1[reg], act);
            }
        }
    }
    sage_finish_block(fin, (int)(gridDim.x * gridDim.y));
}

}  // namespace

int sage_launch_linear_act(const float* self_tab, int64_t ld_self, const int32_t* self_index, const float* agg, int64_t ld_agg,
                           int32_t dim, const float* weight, int64_t ldw, int32_t out_dim, int32_t act, int32_t n,
                           const int32_t* n_dev, float* out, int64_t ldo, int32_t n_off, sage_finish_t fin, hipStream_t st) {
    if (n == 0) return SAGE_OK;
    dim3 grid(sage_cdiv(n, BM), sage_cdiv(out_dim, BN));
    hipLaunchKernelGGL(linear_act_kernel, grid, dim3(256), 0, st, self_tab, ld_self, self_index, agg, ld_agg, dim, weight, ldw,
                       out_dim, act, n, n_dev, out, ldo, n_off, fin);
    SAGE_CHECK_LAUNCH("linear_act_kernel");
    return SAGE_OK;
}

extern "C" int sage_linear_act(const float* self_tab, int64_t ld_self, const int32_t* self_index, const float* agg,
                               int64_t ld_agg, int32_t dim, const float* weight, int64_t ldw, int32_t out_dim, int32_t act,
                               int32_t n, const int32_t* n_dev, float* out, int64_t ldo, sage_stream_t stream) {
    SAGE_REQUIRE(agg && weight && out, "linear_act: NULL array");
    SAGE_REQUIRE(n >= 0 && dim >= 1 && out_dim >= 1, "linear_act: n = %d, dim = %d, out_dim = %d", n, dim, out_dim);
    SAGE_REQUIRE(ld_agg >= dim && ldo >= out_dim, "linear_act: ld_agg = %lld, ldo = %lld", (long long)ld_agg, (long long)ldo);
    SAGE_REQUIRE(!self_tab || ld_self >= dim, "linear_act: ld_self = %lld < dim", (long long)ld_self);
    SAGE_REQUIRE(ldw >= (self_tab ? 2 : 1) * (int64_t)dim, "linear_act: ldw = %lld too small", (long long)ldw);
    SAGE_REQUIRE(act >= 0 && act <= SAGE_ACT_NONE, "linear_act: act = %d", act);
    return sage_launch_linear_act(self_tab, ld_self, self_index, agg, ld_agg, dim, weight, ldw, out_dim, act, n, n_dev, out, ldo, 0,
                                  sage_finish_t{nullptr, nullptr}, (hipStream_t)stream);
}
