// Backward of linear_act / gather_mean (training harness).
//
// The reference obtains these from torch autograd through mm / relu / cat / div
// (model.py:249 loss.backward()).  Here:
//   dZ      = grad_out * act'(out)                         (fused into the operand loads)
//   grad_x  = dZ . W                      [n, ds+dim]      fp32 MFMA, reduction over out_dim
//   grad_W += dZ^T . [self | agg]         [out_dim, ds+dim] fp32 MFMA, reduction over the n rows split over blocks:
//                                                          * _ws entry (round 3): every split writes its partial tile, one reduce kernel
//                                                            adds them in split order -- bitwise reproducible, as model.py:249 is on a CPU;
//                                                            operands straight from HBM in MFMA register order (bwd_dw_direct_kernel)
//                                                          * legacy entry: fp32 atomics (order of arrival)
//   grad_table[row(nbr[r,j])] += grad_agg[r] / c           legacy: fp32 atomics (mean backward = scatter); the reproducible form
//                                                          (inverted index, sage_backward_det.hip) sums every row's list in (r, j) order
// One generic 64x128x32 MFMA tile kernel with functor operands serves grad_x and the odd-shaped grad_W; the training
// operand loads are guarded scalar loads along the index that is contiguous in memory, transposed into LDS.
#include "sage_internal.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 64, BN = 128, BK = 32, LDP = BK + 1;

struct Dz {   // dZ(m, h) = grad_out[row(m), h] * act'(out[row(m), h]); row(m) = order ? order[m] : m (grad_W's canonical row order)
    const float* out; int64_t ldo; const float* gout; int64_t ldg; int act; const int32_t* order;
    __device__ float operator()(int m, int h) const {
        if (order) m = order[m];
        const float g = gout[(int64_t)m * ldg + h];
        const float y = out[(int64_t)m * ldo + h];
        if (act == SAGE_ACT_RELU) return y > 0.f ? g : 0.f;
        if (act == SAGE_ACT_SIGMOID) return g * y * (1.f - y);
        return g;
    }
};

struct Xcat {  // X(m, j) = [self | agg](row(m), j), the forward's operand (encoders.py:49-56)
    const float* self_tab; int64_t ld_self; const int32_t* self_index; const float* agg; int64_t ld_agg; int ds; const int32_t* order;
    __device__ float operator()(int m, int j) const {
        if (order) m = order[m];
        if (j < ds) {
            const int64_t sr = self_index ? (int64_t)self_index[m] : (int64_t)m;
            return self_tab[sr * ld_self + j];
        }
        return agg[(int64_t)m * ld_agg + (j - ds)];
    }
};

// C[M, N] (+)= sum_k A(m, k) * B(n, k).  MODE 0: A(m,k)=dZ(m,k), B(n,k)=W[k][n]  -> grad_x (store)
//                                         MODE 1: A(m,k)=dZ(k,m), B(n,k)=X(k,n)  -> grad_W (atomic add, split K)
template <int MODE>
__global__ __launch_bounds__(256) void bwd_gemm_kernel(Dz dz, Xcat x, const float* __restrict__ W, int64_t ldw,
                                                       int M, int N, int K, float* __restrict__ C, int64_t ldc, int ksplit,
                                                       const int32_t* __restrict__ rows_dev, int64_t zstride) {
    // zstride != 0 (MODE 1 only): split z STORES its tile into C + z * zstride (a partial sum for dw_reduce_kernel) instead of
    // adding it to C with atomics; a split with no rows stores zeros.
    __shared__ float smem[(BM + BN) * LDP];
    if (rows_dev) {                                   // the number of layer rows lives on the device (frontier size): it bounds the
        if (MODE == 0) M = min(*rows_dev, M);         // output rows of grad_x and the reduction length of grad_W
        else K = min(*rows_dev, K);
    }
    float* As = smem;
    float* Bs = smem + BM * LDP;
    const int m0 = blockIdx.x * BM, nb0 = blockIdx.y * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kper = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
    const int kbeg = blockIdx.z * kper, kend = min(K, kbeg + kper);
    if (kbeg >= kend && zstride == 0) return;
    if (zstride != 0) C += (int64_t)blockIdx.z * zstride;
    const int wave_n0 = nb0 + wave * 32;
    const bool wave_active = wave_n0 < N;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int kk = tid & 31, rr = tid >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        // Operand tiles go to LDS as [row][k]; the LOADS run along whichever index is contiguous in memory (LDP = 33: the
        // transposing stores are conflict-free).  With k across the lanes for every operand (the first version: "layers are
        // small") the grad_W operands -- dZ(k, m) and X(k, n), both row-major in k -- were 4-byte loads one row apart:
        // 90 us for the 23.6 k x 256 x 128 weight gradient of config 3.
        if constexpr (MODE == 0) {
            const int gk = k0 + kk;
#pragma unroll
            for (int i = 0; i < BM / 8; ++i) {                       // dZ(m, k): k contiguous
                const int row = rr + 8 * i, gm = m0 + row;
                float v = 0.f;
                if (gm < M && gk < kend) v = dz(gm, gk);
                As[row * LDP + kk] = v;
            }
#pragma unroll
            for (int i = 0; i < BN * BK / 256; ++i) {                // W[k][n]: n contiguous
                const int idx = tid + 256 * i, nl = idx % BN, kl = idx / BN, gn = nb0 + nl, gk2 = k0 + kl;
                float v = 0.f;
                if (gn < N && gk2 < kend) v = W[(int64_t)gk2 * ldw + gn];
                Bs[nl * LDP + kl] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < BM * BK / 256; ++i) {                // dZ(k, m): m contiguous
                const int idx = tid + 256 * i, ml = idx % BM, kl = idx / BM, gm = m0 + ml, gk2 = k0 + kl;
                float v = 0.f;
                if (gm < M && gk2 < kend) v = dz(gk2, gm);
                As[ml * LDP + kl] = v;
            }
#pragma unroll
            for (int i = 0; i < BN * BK / 256; ++i) {                // X(k, n): n contiguous
                const int idx = tid + 256 * i, nl = idx % BN, kl = idx / BN, gn = nb0 + nl, gk2 = k0 + kl;
                float v = 0.f;
                if (gn < N && gk2 < kend) v = x(gk2, gn);
                Bs[nl * LDP + kl] = v;
            }
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
        if (col < N) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int g0 = m0 + row, g1 = m0 + 32 + row;
                if (MODE == 0) {
                    if (g0 < M) C[(int64_t)g0 * ldc + col] = acc0[reg];
                    if (g1 < M) C[(int64_t)g1 * ldc + col] = acc1[reg];
                } else if (zstride != 0) {
                    if (g0 < M) C[(int64_t)g0 * ldc + col] = acc0[reg];
                    if (g1 < M) C[(int64_t)g1 * ldc + col] = acc1[reg];
                } else {
                    if (g0 < M) atomicAdd(&C[(int64_t)g0 * ldc + col], acc0[reg]);
                    if (g1 < M) atomicAdd(&C[(int64_t)g1 * ldc + col], acc1[reg]);
                }
            }
        }
    }
}

// ---- weight gradient, operands straight from HBM in MFMA register order (round 3) -------------------------------------------------
// grad_W[m][n] = sum_k dZ(k, m) X(k, n): the reduction index k is the ROW index of both operands, so lane (i = l & 31, h = l >> 5)
// of v_mfma_f32_32x32x2_f32 wants A = dZ[k0 + h][m], B = X[k0 + h][n] -- row-major operands need no transpose at all.  With 8-byte
// loads a lane holds two consecutive m (two consecutive n): two A and two B fragments, i.e. a 64 x 64 output tile per wave whose rows
// are simply permuted (fragment e owns rows m0 + 2 i + e), undone when the tile is stored.  512-thread blocks: 2 (M halves) x 4
// (64-column groups) = a 128 x 256 tile per block; a block owns a contiguous range of k and STORES its partial tile; dw_reduce_kernel
// adds the partials in block order, so the result does not depend on which block finished first (the generic kernel's atomics did:
// captured and eager steps agreed to 1e-4 only).  PF k-pairs (3 x PF loads per lane) are in flight per trip, two waves per SIMD
// take turns on the matrix pipe.  Measured at config 3 (23.6 k x 128 x 256): see DESIGN.md.
struct DwArgs {
    const float* gout; int64_t ldg; const float* out; int64_t ldo; int act;     // dZ(k, m) = gout[k][m] * act'(out[k][m])
    const float* x; int64_t ldx; const int32_t* index; int x_rows;              // X(k, n) = x[index ? index[k] : k][n]
    const int32_t* order;                                                       // nullable: the k-th term of the sum is row order[k]
    int M, Nx, n_rows; const int32_t* rows_dev;
    float* partial; int64_t ldp; int col_off; int nsplit;                       // partial[split][m][col_off + n]
};

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <bool INDEXED>
__global__ __launch_bounds__(512) void bwd_dw_direct_kernel(const DwArgs a) {
    constexpr int PF = 4;
    int nrows = a.n_rows;
    if (a.rows_dev) nrows = min(*a.rows_dev, nrows);
    nrows = max(nrows, 0);
    const int per = (((nrows + a.nsplit - 1) / a.nsplit) + 1) & ~1;       // rows per split, even (whole k-pairs)
    const int kbeg = min((int)blockIdx.x * per, nrows), kend = min(nrows, kbeg + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.z * 128 + (wave >> 2) * 64, n0 = blockIdx.y * 256 + (wave & 3) * 64;
    if (m0 >= a.M || n0 >= a.Nx) return;                                  // no barrier in this kernel
    const int i = lane & 31, h = lane >> 5;
    const int mc = m0 + 2 * i, nc = n0 + 2 * i;
    const bool mok = mc < a.M, nok = nc < a.Nx;                           // M, Nx even (host-checked)
    const float* gp = a.gout + min(mc, a.M - 2);
    const float* yp = a.out + min(mc, a.M - 2);
    const float* xp = a.x + min(nc, a.Nx - 2);
    f32x16 acc[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += 2 * PF) {
        f32x2 g[PF], y[PF], xv[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) {                                    // rows past the range: clamped address, zeroed below
            int rc = min(k0 + 2 * p + h, kend - 1);
            if (a.order) rc = a.order[rc];
            int64_t xr = rc;
            if (INDEXED) xr = min(max(a.index[rc], 0), a.x_rows - 1);
            g[p] = *reinterpret_cast<const f32x2*>(gp + (int64_t)rc * a.ldg);
            y[p] = *reinterpret_cast<const f32x2*>(yp + (int64_t)rc * a.ldo);
            xv[p] = *reinterpret_cast<const f32x2*>(xp + xr * a.ldx);
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const bool live = mok && (k0 + 2 * p + h) < kend;
            float dz[2], xx[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float v = g[p][e];
                if (a.act == SAGE_ACT_RELU) v = y[p][e] > 0.f ? v : 0.f;
                else if (a.act == SAGE_ACT_SIGMOID) v = v * y[p][e] * (1.f - y[p][e]);
                dz[e] = live ? v : 0.f;
                xx[e] = nok ? xv[p][e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz[e], xx[f], acc[e][f], 0, 0, 0);
        }
    }
    float* P = a.partial + (int64_t)blockIdx.x * a.M * a.ldp + a.col_off;
    if (nok) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + 2 * ((reg & 3) + 8 * (reg >> 2) + 4 * h) + e;
                if (m < a.M) *reinterpret_cast<f32x2*>(P + (int64_t)m * a.ldp + nc) = f32x2{acc[e][0][reg], acc[e][1][reg]};
            }
    }
}

// ---- grad_x = dZ . W, operands straight from HBM (round 3) ------------------------------------------------------------------------
// grad_x[m][n] = sum_h dZ(m, h) W[h][n]: M = the layer's rows, reduction over the layer's (small) output width.  A-operand lanes
// (i = l & 31, hh = l >> 5) load 16 bytes of THEIR row of grad_out / out -- dZ(m0 + i, h0 + 4 hh + e), e = 0..3 -- and the four MFMAs
// of a step take the reduction pairs {h0 + e, h0 + 4 + e}; the B operand W[h0 + 4 hh + e][n0 + j] is a coalesced row piece for any
// pairing.  A wave owns 32 rows x 128 columns (four accumulators), a 256-thread block 128 rows.  No LDS, no barrier; every output
// element is one fixed-order sum, so the result is reproducible.  The generic tile kernel took 35 us for the [4096, 128] x [128, 128]
// product of config 3's layer 2; this form is bound by its 2 + 2 + 2 MB of traffic.
struct DxArgs {
    const float* gout; int64_t ldg; const float* out; int64_t ldo; int act;
    const float* W; int64_t ldw; int H, K;             // W [H, K]: H = out_dim (reduction), K = ds + dim (grad_x's width)
    int n_rows; const int32_t* rows_dev;
    float* gx; int64_t ldgx;
};

__global__ __launch_bounds__(256) void bwd_dx_direct_kernel(const DxArgs a) {
    int nrows = a.n_rows;
    if (a.rows_dev) nrows = min(*a.rows_dev, nrows);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (blockIdx.x * 4 + wave) * 32, n0 = blockIdx.y * 128;
    if (m0 >= nrows) return;
    const int i = lane & 31, hh = lane >> 5;
    const int row = min(m0 + i, nrows - 1);
    const bool rok = m0 + i < nrows;
    const float* gp = a.gout + (int64_t)row * a.ldg;
    const float* yp = a.out + (int64_t)row * a.ldo;
    f32x16 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    for (int h0 = 0; h0 < a.H; h0 += 8) {                              // H % 4 == 0 (host-checked)
        const int hc = h0 + 4 * hh;
        const bool hok = hc < a.H;
        const int hcl = min(hc, a.H - 4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gp + hcl);
        const f32x4 y = *reinterpret_cast<const f32x4*>(yp + hcl);
        float wv[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int n = n0 + 32 * f + i;
                wv[e][f] = (hok && n < a.K) ? a.W[(int64_t)(hcl + e) * a.ldw + n] : 0.f;
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = g[e];
            if (a.act == SAGE_ACT_RELU) v = y[e] > 0.f ? v : 0.f;
            else if (a.act == SAGE_ACT_SIGMOID) v = v * y[e] * (1.f - y[e]);
            const float dz = (rok && hok) ? v : 0.f;
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz, wv[e][f], acc[f], 0, 0, 0);
        }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int n = n0 + 32 * f + i;
        if (n < a.K) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                if (m < nrows) a.gx[(int64_t)m * a.ldgx + n] = acc[f][reg];
            }
        }
    }
}

// grad_W[m][n] += partial[0][m][n] + partial[1][m][n] + ... in split order (one thread per element: coalesced along n)
__global__ __launch_bounds__(256) void dw_reduce_kernel(const float* __restrict__ partial, int nsplit, int M, int K, int64_t ldp,
                                                       float* __restrict__ gw, int64_t ldgw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * K) return;
    const int m = idx / K, n = idx % K;
    const float* p = partial + (int64_t)m * ldp + n;
    const int64_t stride = (int64_t)M * ldp;
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= nsplit; z += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + (int64_t)(z + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < nsplit; ++z) s += __builtin_nontemporal_load(p + (int64_t)z * stride);
    gw[(int64_t)m * ldgw + n] += s;
}

// mean backward: one wave per destination row, lanes over columns
__global__ __launch_bounds__(256) void gather_mean_bwd_kernel(const float* __restrict__ gagg, int64_t ldg, int dim,
                                                              const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int k,
                                                              int n, const int32_t* __restrict__ n_dev,
                                                              const int32_t* __restrict__ slot_rows, const int32_t* __restrict__ self_row,
                                                              float* __restrict__ gtab, int table_rows, int64_t ld) {
    int nn = n;
    if (n_dev) nn = min(*n_dev, n);
    const int lane = sage_lane();
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    for (int r = wave; r < nn; r += nwaves) {
        const int c = __builtin_amdgcn_readfirstlane(cnt[r]);
        int s = -1;
        if (self_row) {
            s = self_row[r];
            if (slot_rows && s >= 0) s = slot_rows[s];
            s = __builtin_amdgcn_readfirstlane(s);
        }
        bool extra = s >= 0;
        if (extra) {
            for (int base = 0; base < c; base += kWave) {
                int id = (base + lane < c) ? nbr[(int64_t)r * k + base + lane] : -1;
                if (slot_rows && id >= 0) id = slot_rows[id];
                if (__any(id == s)) extra = false;
            }
        }
        const int ceff = c + (extra ? 1 : 0);
        if (ceff == 0) continue;
        const float inv = 1.0f / (float)ceff;
        for (int j = 0; j < ceff; ++j) {
            int id;
            if (j < c) {
                id = nbr[(int64_t)r * k + j];
                if (slot_rows) id = slot_rows[max(id, 0)];
            } else {
                id = s;
            }
            if (id < 0 || id >= table_rows) continue;
            for (int col = lane; col < dim; col += kWave)
                atomicAdd(&gtab[(int64_t)id * ld + col], gagg[(int64_t)r * ldg + col] * inv);
        }
    }
}

}  // namespace

namespace {
// ---- layer-1 weight gradient of the two-layer stack, summed over the OUTER EDGES (round 3) -------------------------------------------
// d loss / d W1 = sum over the layer-1 rows t of dZ1[t] (x) X1[t], dZ1[t] = act'(h1[t]) . grad_h1[t], and grad_h1[t] is the mean
// backward of layer 2: the sum over the outer samples e = (seed r, slot j) that point at row t of grad_agg2[r] / c_r.  Written as a sum
// over ROWS this needs grad_h1 (a scatter: atomics, or an inverted index = a device-wide sort per step) and, for reproducible bits,
// a canonical order of the frontier's rows, which sit in arbitrary order (a second sort).  Re-associated as a sum over the EDGES,
//     d W1 = sum_r [ concat: act'(h1[r]) . g_self[r] (x) X1[r] ]  +  sum_r sum_{j < c_r} (act'(h1[t_rj]) . g_agg[r] / c_r) (x) X1[t_rj],
// the terms come in (r, j) order -- fixed by the seeds and the sampler key, whatever the frontier's layout -- and neither grad_h1, nor
// the scatter, nor any sort exists: 1.7x the matrix work of the row form (an edge per term instead of a row per term) against five
// launches and ~140 us less at config 3.  Bitwise reproducible: a block owns a fixed range of seeds, compacts their live terms in
// slot order (ballot prefix in LDS), stages 16 terms at a time -- dZ [16 x 128] and X1 [16 x 256], gathered by 16-byte pieces,
// double-buffered in LDS, the next trip's rows in flight during the MFMA loop -- and STORES its partial tile for dw_reduce_kernel.
struct EdgeArgs {
    const float* gx2; int64_t ldgx; int g_self_off, g_agg_off;      // layer 2's grad_x: columns of the self part (concat) / the mean part
    const int32_t* row2; const int32_t* cnt2; int k2; const int32_t* self_row2; int batch;
    const float* h1; int64_t ldh; int M; int act;                    // layer-1 output rows (M = its width) and activation
    const float* agg1; int64_t lda; int d0;
    const float* table; int64_t table_ld; const int32_t* s1_nodes; int concat;
    float* partial; int64_t ldp; int rows_per_chunk;
};

constexpr int EK_KT = 16, EK_TMAX = 512, EK_LDZ = 128 + 4, EK_LDX = 256 + 4, EK_ROWS = 64;

__global__ __launch_bounds__(512) void bwd_dw_edges_kernel(const EdgeArgs a) {
    __shared__ int term_t[EK_TMAX + EK_KT], term_r[EK_TMAX + EK_KT], term_g[EK_TMAX + EK_KT], term_s[EK_TMAX + EK_KT];
    __shared__ float term_w[EK_TMAX + EK_KT];
    __shared__ __attribute__((aligned(16))) float dzs[2][EK_KT][EK_LDZ];
    __shared__ __attribute__((aligned(16))) float xs[2][EK_KT][EK_LDX];
    __shared__ int rowc[EK_ROWS], rowx[EK_ROWS], wcount[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int SPR = a.k2 + 2;                                       // slots per seed: own row (concat), k2 samples, the self-loop row
    const int m0b = blockIdx.z * 128, n0b = blockIdx.y * 256;
    const int mrel = (wave >> 2) * 64, nrel = (wave & 3) * 64;
    const int ds = a.concat ? a.d0 : 0, K1 = ds + a.d0;
    const int i = lane & 31, h = lane >> 5;
    const bool wave_live = m0b + mrel < a.M && n0b + nrel < K1;
    f32x16 acc[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;
    // staging duty of this thread: one 16-byte piece of dZ (term tid / 32, columns 4 (tid % 32)), two of X1
    const int zk = tid >> 5, zc = m0b + 4 * (tid & 31);
    const bool zok = zc < a.M;
    const int RB = a.rows_per_chunk;
    const int nchunks = (a.batch + RB - 1) / RB;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int r_beg = chunk * RB, nrows = min(RB, a.batch - r_beg);
        __syncthreads();                                            // the previous chunk's last trip has been read
        if (tid < nrows) {
            const int r = r_beg + tid;
            const int c = min(max(a.cnt2[r], 0), a.k2);
            int s = a.self_row2 ? a.self_row2[r] : -1;
            if (s >= 0)
                for (int j = 0; j < c; ++j)
                    if (a.row2[(int64_t)r * a.k2 + j] == s) { s = -1; break; }      // aggregators.py:50-51: set union (the forward's rule)
            rowc[tid] = c;
            rowx[tid] = s;
        }
        __syncthreads();
        int base = 0;
        for (int s0 = 0; s0 < nrows * SPR; s0 += 512) {              // slots -> live terms, in slot order
            const int sl = s0 + tid;
            bool valid = false;
            int t = 0, r = r_beg, g = a.g_agg_off;
            float w = 0.f;
            if (sl < nrows * SPR) {
                const int lr = sl / SPR, j = sl % SPR;
                r = r_beg + lr;
                const int c = rowc[lr], ex = rowx[lr];
                const float inv = 1.0f / (float)max(c + (ex >= 0 ? 1 : 0), 1);
                if (j == 0) { if (a.concat) { valid = true; t = r; w = 1.f; g = a.g_self_off; } }
                else if (j <= a.k2) { if (j - 1 < c) { valid = true; t = a.row2[(int64_t)r * a.k2 + j - 1]; w = inv; } }
                else if (ex >= 0) { valid = true; t = ex; w = inv; }
                if (t < 0) valid = false;
            }
            const unsigned long long b = __ballot(valid);
            if (lane == 0) wcount[wave] = __popcll(b);
            __syncthreads();
            int off = base;
            for (int q = 0; q < wave; ++q) off += wcount[q];
            int total = 0;
            for (int q = 0; q < 8; ++q) total += wcount[q];
            if (valid) {
                const int pos = off + __popcll(b & ((1ull << lane) - 1ull));
                term_t[pos] = t; term_r[pos] = r; term_g[pos] = g; term_w[pos] = w;
                term_s[pos] = (a.concat && a.s1_nodes) ? a.s1_nodes[t] : t;
            }
            base += total;
            __syncthreads();
        }
        const int nterm = (base + EK_KT - 1) / EK_KT * EK_KT;
        if (tid < nterm - base) {                                   // pad the last trip with weight-0 terms on a valid row
            const int pos = base + tid;
            term_t[pos] = 0; term_r[pos] = r_beg; term_g[pos] = a.g_agg_off; term_w[pos] = 0.f; term_s[pos] = (a.concat && a.s1_nodes) ? a.s1_nodes[0] : 0;
        }
        __syncthreads();
        f32x4 gq, yq, xq[2];
        auto request = [&](int trip) {                               // global -> registers, no wait
            const int kk = trip * EK_KT + zk;
            const int t = term_t[kk], r = term_r[kk];
            const int zcl = min(zc, a.M - 4);
            gq = *reinterpret_cast<const f32x4*>(a.gx2 + (int64_t)r * a.ldgx + term_g[kk] + zcl);
            yq = *reinterpret_cast<const f32x4*>(a.h1 + (int64_t)t * a.ldh + zcl);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = tid + 512 * u, xk = trip * EK_KT + (idx >> 6);
                const int n = min(n0b + 4 * (idx & 63), K1 - 4);
                const float* src = (n < ds) ? a.table + (int64_t)term_s[xk] * a.table_ld + n : a.agg1 + (int64_t)term_t[xk] * a.lda + (n - ds);
                xq[u] = *reinterpret_cast<const f32x4*>(src);
            }
        };
        auto stage = [&](int buf, int trip) {                        // registers -> LDS
            const float w = term_w[trip * EK_KT + zk];
            f32x4 dz;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = gq[e];
                if (a.act == SAGE_ACT_RELU) v = yq[e] > 0.f ? v : 0.f;
                else if (a.act == SAGE_ACT_SIGMOID) v = v * yq[e] * (1.f - yq[e]);
                dz[e] = (zok && w != 0.f) ? v * w : 0.f;             // select: a padded term may sit on a row holding Inf / NaN
            }
            *reinterpret_cast<f32x4*>(&dzs[buf][zk][4 * (tid & 31)]) = dz;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = tid + 512 * u;
                const bool nok = n0b + 4 * (idx & 63) < K1;
                f32x4 xv = xq[u];
                if (!nok) xv = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&xs[buf][idx >> 6][4 * (idx & 63)]) = xv;
            }
        };
        const int ntrip = nterm / EK_KT;
        int buf = 0;
        if (ntrip > 0) request(0);
        for (int trip = 0; trip < ntrip; ++trip, buf ^= 1) {
            stage(buf, trip);
            __syncthreads();
            if (trip + 1 < ntrip) request(trip + 1);                 // in flight during the MFMA loop
            if (wave_live) {
#pragma unroll
                for (int kp = 0; kp < EK_KT / 2; ++kp) {
                    const f32x2 av = *reinterpret_cast<const f32x2*>(&dzs[buf][2 * kp + h][mrel + 2 * i]);
                    const f32x2 bv = *reinterpret_cast<const f32x2*>(&xs[buf][2 * kp + h][nrel + 2 * i]);
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int f = 0; f < 2; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[f], acc[e][f], 0, 0, 0);
                }
            }
        }
    }
    if (wave_live) {
        float* P = a.partial + (int64_t)blockIdx.x * a.M * a.ldp;
        const int nc = n0b + nrel + 2 * i;
        if (nc < K1) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int m = m0b + mrel + 2 * ((reg & 3) + 8 * (reg >> 2) + 4 * h) + e;
                    if (m < a.M) *reinterpret_cast<f32x2*>(P + (int64_t)m * a.ldp + nc) = f32x2{acc[e][0][reg], acc[e][1][reg]};
                }
        }
    }
}

int edges_rows_per_chunk(int batch, int k2) {
    const int cap = max(1, min(EK_ROWS, EK_TMAX / (k2 + 2)));       // a chunk's slots fit the LDS term list (61 KB of LDS per block in all)
    return max(min(cap, 4), min(cap, sage_cdiv(batch, sage_tunables().bwd_direct_blocks)));   // small batches: >= 4 seeds per chunk (whole trips)
}
int edges_splits(int batch, int k2) { return max(1, min(sage_tunables().bwd_direct_blocks, sage_cdiv(batch, edges_rows_per_chunk(batch, k2)))); }

}  // namespace

namespace {
// splits of the reduction over the rows: the direct kernel's row ranges / the generic kernel's K splits
int dw_direct_splits(int n) { return max(1, min(sage_tunables().bwd_direct_blocks, sage_cdiv(n, 64))); }
int dw_generic_splits(int n, int out_dim, int K) {
    const int tiles = sage_cdiv(out_dim, BM) * sage_cdiv(K, BN);
    return max(1, min(sage_cdiv(n, 4 * BK), sage_tunables().bwd_blocks / tiles));
}

int linear_act_backward_impl(const float* self_tab, int64_t ld_self, const int32_t* self_index, const float* agg,
                             int64_t ld_agg, int32_t dim, const float* weight, int64_t ldw, int32_t out_dim,
                             int32_t act, const float* out, int64_t ldo, const float* grad_out, int64_t ldg, int32_t n,
                             const int32_t* n_dev, float* grad_weight, int64_t ldgw, float* grad_x, int64_t ldgx,
                             const int32_t* row_order, void* workspace, size_t workspace_bytes, bool reproducible, sage_stream_t stream) {
    SAGE_REQUIRE(agg && weight && out && grad_out, "linear_act_backward: NULL array");
    SAGE_REQUIRE(n >= 0 && dim >= 1 && out_dim >= 1, "linear_act_backward: n = %d, dim = %d, out_dim = %d", n, dim, out_dim);
    SAGE_REQUIRE(act >= 0 && act <= SAGE_ACT_NONE, "linear_act_backward: act = %d", act);
    const int ds = self_tab ? dim : 0, K = ds + dim;
    SAGE_REQUIRE(ld_agg >= dim && ldo >= out_dim && ldg >= out_dim && ldw >= K, "linear_act_backward: leading dimensions");
    SAGE_REQUIRE(!self_tab || ld_self >= dim, "linear_act_backward: ld_self");
    SAGE_REQUIRE(!grad_weight || ldgw >= K, "linear_act_backward: ldgw = %lld < %d", (long long)ldgw, K);
    SAGE_REQUIRE(!grad_x || ldgx >= K, "linear_act_backward: ldgx = %lld < %d", (long long)ldgx, K);
    if (reproducible && grad_weight) {
        const size_t need = sage_linear_act_backward_workspace_bytes(n, dim, self_tab ? 1 : 0, out_dim);
        SAGE_REQUIRE(workspace && sage_aligned(workspace, 16) && workspace_bytes >= need,
                     "linear_act_backward_ws: workspace %zu bytes (need %zu, 16-byte aligned)", workspace_bytes, need);
    }
    if (n == 0) return SAGE_OK;
    hipStream_t st = (hipStream_t)stream;
    const Dz dz{out, ldo, grad_out, ldg, act, nullptr};
    const Xcat x{self_tab, ld_self, self_index, agg, ld_agg, ds, nullptr};
    const Dz dzo{out, ldo, grad_out, ldg, act, row_order};          // grad_W only: rows in the caller's canonical order
    const Xcat xo{self_tab, ld_self, self_index, agg, ld_agg, ds, row_order};
    if (grad_x) {
        // 16-byte pieces of grad_out / out rows: out_dim % 4 == 0, leading dimensions % 4 == 0, 16-byte aligned bases
        if (out_dim % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && sage_aligned(grad_out, 16) && sage_aligned(out, 16)) {
            const DxArgs a{grad_out, ldg, out, ldo, act, weight, ldw, out_dim, K, n, n_dev, grad_x, ldgx};
            hipLaunchKernelGGL(bwd_dx_direct_kernel, dim3(sage_cdiv(n, 128), sage_cdiv(K, 128)), dim3(256), 0, st, a);
            SAGE_CHECK_LAUNCH("bwd_dx_direct_kernel");
        } else {
            dim3 grid(sage_cdiv(n, BM), sage_cdiv(K, BN), 1);
            hipLaunchKernelGGL(bwd_gemm_kernel<0>, grid, dim3(256), 0, st, dz, x, weight, ldw, n, K, out_dim, grad_x, ldgx, 1, n_dev, (int64_t)0);
            SAGE_CHECK_LAUNCH("bwd_gemm_kernel<grad_x>");
        }
    }
    if (!grad_weight) return SAGE_OK;
    if (!reproducible) {
        const int ksplit = dw_generic_splits(n, out_dim, K);
        dim3 grid(sage_cdiv(out_dim, BM), sage_cdiv(K, BN), ksplit);
        hipLaunchKernelGGL(bwd_gemm_kernel<1>, grid, dim3(256), 0, st, dz, x, weight, ldw, out_dim, K, n, grad_weight, ldgw, ksplit, n_dev, (int64_t)0);
        SAGE_CHECK_LAUNCH("bwd_gemm_kernel<grad_w>");
        return SAGE_OK;
    }
    float* partial = (float*)workspace;
    const int64_t ldp = K;
    int nsplit;
    // 8-byte operand loads: even widths and leading dimensions, 8-byte aligned bases (the engine's padded widths always are)
    const bool direct = out_dim % 2 == 0 && dim % 2 == 0 && ldg % 2 == 0 && ldo % 2 == 0 && ld_agg % 2 == 0 && (!self_tab || ld_self % 2 == 0) &&
                        sage_aligned(grad_out, 8) && sage_aligned(out, 8) && sage_aligned(agg, 8) && (!self_tab || sage_aligned(self_tab, 8));
    if (direct) {
        nsplit = dw_direct_splits(n);
        DwArgs a{grad_out, ldg, out, ldo, act, agg, ld_agg, nullptr, n, row_order, out_dim, dim, n, n_dev, partial, ldp, ds, nsplit};
        dim3 grid(nsplit, sage_cdiv(dim, 256), sage_cdiv(out_dim, 128));
        hipLaunchKernelGGL(bwd_dw_direct_kernel<false>, grid, dim3(512), 0, st, a);       // the neighbour means: columns [ds, ds + dim)
        SAGE_CHECK_LAUNCH("bwd_dw_direct_kernel<agg>");
        if (self_tab) {                                                                    // the nodes' own rows: columns [0, dim)
            a.x = self_tab; a.ldx = ld_self; a.col_off = 0;
            if (self_index) {
                a.index = self_index; a.x_rows = 1 << 30;       // the caller vouches for the index range, as in the forward
                hipLaunchKernelGGL(bwd_dw_direct_kernel<true>, grid, dim3(512), 0, st, a);
            } else {
                hipLaunchKernelGGL(bwd_dw_direct_kernel<false>, grid, dim3(512), 0, st, a);
            }
            SAGE_CHECK_LAUNCH("bwd_dw_direct_kernel<self>");
        }
    } else {
        nsplit = dw_generic_splits(n, out_dim, K);
        dim3 grid(sage_cdiv(out_dim, BM), sage_cdiv(K, BN), nsplit);
        hipLaunchKernelGGL(bwd_gemm_kernel<1>, grid, dim3(256), 0, st, dzo, xo, weight, ldw, out_dim, K, n, partial, ldp, nsplit, n_dev,
                           (int64_t)out_dim * ldp);
        SAGE_CHECK_LAUNCH("bwd_gemm_kernel<grad_w partials>");
    }
    hipLaunchKernelGGL(dw_reduce_kernel, dim3(sage_cdiv((int64_t)out_dim * K, 256)), dim3(256), 0, st, partial, nsplit, out_dim, K, ldp, grad_weight, ldgw);
    SAGE_CHECK_LAUNCH("dw_reduce_kernel");
    return SAGE_OK;
}
}  // namespace

extern "C" size_t sage_linear_act_backward_workspace_bytes(int32_t n, int32_t dim, int32_t has_self, int32_t out_dim) {
    if (n <= 0 || dim <= 0 || out_dim <= 0) return 16;
    const int K = (has_self ? 2 : 1) * dim;
    const int splits = max(dw_direct_splits(n), dw_generic_splits(n, out_dim, K));
    return (size_t)splits * out_dim * K * sizeof(float) + 16;
}

extern "C" int sage_linear_act_backward(const float* self_tab, int64_t ld_self, const int32_t* self_index, const float* agg,
                                        int64_t ld_agg, int32_t dim, const float* weight, int64_t ldw, int32_t out_dim,
                                        int32_t act, const float* out, int64_t ldo, const float* grad_out, int64_t ldg, int32_t n,
                                        const int32_t* n_dev, float* grad_weight, int64_t ldgw, float* grad_x, int64_t ldgx,
                                        sage_stream_t stream) {
    return linear_act_backward_impl(self_tab, ld_self, self_index, agg, ld_agg, dim, weight, ldw, out_dim, act, out, ldo, grad_out, ldg, n,
                                    n_dev, grad_weight, ldgw, grad_x, ldgx, nullptr, nullptr, 0, false, stream);
}

extern "C" int sage_linear_act_backward_ws(const float* self_tab, int64_t ld_self, const int32_t* self_index, const float* agg,
                                           int64_t ld_agg, int32_t dim, const float* weight, int64_t ldw, int32_t out_dim,
                                           int32_t act, const float* out, int64_t ldo, const float* grad_out, int64_t ldg, int32_t n,
                                           const int32_t* n_dev, float* grad_weight, int64_t ldgw, float* grad_x, int64_t ldgx,
                                           const int32_t* row_order, void* workspace, size_t workspace_bytes, sage_stream_t stream) {
    return linear_act_backward_impl(self_tab, ld_self, self_index, agg, ld_agg, dim, weight, ldw, out_dim, act, out, ldo, grad_out, ldg, n,
                                    n_dev, grad_weight, ldgw, grad_x, ldgx, row_order, workspace, workspace_bytes, true, stream);
}

extern "C" int sage_gather_mean_backward(const float* grad_agg, int64_t ldg, int32_t dim, const int32_t* nbr, const int32_t* cnt,
                                         int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                                         const int32_t* self_row, float* grad_table, int64_t table_rows, int64_t ld,
                                         sage_stream_t stream) {
    SAGE_REQUIRE(grad_agg && nbr && cnt && grad_table, "gather_mean_backward: NULL array");
    SAGE_REQUIRE(n >= 0 && k >= 1 && dim >= 1 && ldg >= dim && ld >= dim, "gather_mean_backward: n=%d k=%d dim=%d", n, k, dim);
    SAGE_REQUIRE(table_rows >= 1 && table_rows < (1ll << 31), "gather_mean_backward: table_rows = %lld", (long long)table_rows);
    if (n == 0) return SAGE_OK;
    const int blocks = min(sage_cdiv(n, 4), kNumCU * 8);
    hipLaunchKernelGGL(gather_mean_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grad_agg, ldg, dim, nbr, cnt, k, n, n_dev,
                       slot_rows, self_row, grad_table, (int)table_rows, ld);
    SAGE_CHECK_LAUNCH("gather_mean_bwd_kernel");
    return SAGE_OK;
}

extern "C" size_t sage_two_hop_grad_w1_workspace_bytes(int32_t batch, int32_t k2, int32_t d0, int32_t concat, int32_t h1) {
    if (batch <= 0 || k2 <= 0 || d0 <= 0 || h1 <= 0) return 16;
    return (size_t)edges_splits(batch, k2) * h1 * ((concat ? 2 : 1) * (size_t)d0) * sizeof(float) + 16;
}

extern "C" int sage_two_hop_grad_w1(const float* grad_x2, int64_t ldgx, const int32_t* row2, const int32_t* cnt2, int32_t k2,
                                    const int32_t* self_row2, int32_t batch, const float* h1, int64_t ldh, int32_t h1_dim, int32_t act1,
                                    const float* agg1, int64_t lda, int32_t d0, int32_t concat, const float* table, int64_t table_ld,
                                    const int32_t* s1_nodes, float* grad_w1, int64_t ldgw, void* workspace, size_t workspace_bytes,
                                    sage_stream_t stream) {
    SAGE_REQUIRE(grad_x2 && row2 && cnt2 && h1 && agg1 && grad_w1 && workspace, "two_hop_grad_w1: NULL array");
    SAGE_REQUIRE(!concat || (table && s1_nodes), "two_hop_grad_w1: the concat encoder needs the table and the layer's node ids");
    SAGE_REQUIRE(batch >= 1 && k2 >= 1 && k2 <= SAGE_MAX_FANOUT && act1 >= 0 && act1 <= SAGE_ACT_NONE, "two_hop_grad_w1: batch = %d, k2 = %d", batch, k2);
    const int mult = concat ? 2 : 1;
    SAGE_REQUIRE(h1_dim >= 4 && h1_dim % 4 == 0 && d0 >= 4 && d0 % 4 == 0 && ldgx % 4 == 0 && ldh % 4 == 0 && lda % 4 == 0 &&
                 (!concat || table_ld % 4 == 0) && ldgx >= mult * h1_dim && ldh >= h1_dim && lda >= d0 && ldgw >= mult * d0,
                 "two_hop_grad_w1: widths and leading dimensions must be multiples of 4 (h1 = %d, d0 = %d)", h1_dim, d0);
    SAGE_REQUIRE(sage_aligned(grad_x2, 16) && sage_aligned(h1, 16) && sage_aligned(agg1, 16) && (!concat || sage_aligned(table, 16)) &&
                 sage_aligned(workspace, 16), "two_hop_grad_w1: 16-byte alignment");
    const size_t need = sage_two_hop_grad_w1_workspace_bytes(batch, k2, d0, concat, h1_dim);
    if (workspace_bytes < need) {
        sage_set_error("two_hop_grad_w1: workspace %zu bytes < %zu needed", workspace_bytes, need);
        return SAGE_ENOSPACE;
    }
    const int K1 = mult * d0, nsplit = edges_splits(batch, k2);
    hipStream_t st = (hipStream_t)stream;
    const EdgeArgs a{grad_x2, ldgx, 0, concat ? h1_dim : 0, row2, cnt2, k2, self_row2, batch, h1, ldh, h1_dim, act1, agg1, lda, d0,
                     table, table_ld, s1_nodes, concat ? 1 : 0, (float*)workspace, (int64_t)K1, edges_rows_per_chunk(batch, k2)};
    dim3 grid(nsplit, sage_cdiv(K1, 256), sage_cdiv(h1_dim, 128));
    hipLaunchKernelGGL(bwd_dw_edges_kernel, grid, dim3(512), 0, st, a);
    SAGE_CHECK_LAUNCH("bwd_dw_edges_kernel");
    hipLaunchKernelGGL(dw_reduce_kernel, dim3(sage_cdiv((int64_t)h1_dim * K1, 256)), dim3(256), 0, st, (const float*)workspace, nsplit, h1_dim, K1,
                       (int64_t)K1, grad_w1, ldgw);
    SAGE_CHECK_LAUNCH("dw_reduce_kernel");
    return SAGE_OK;
}
