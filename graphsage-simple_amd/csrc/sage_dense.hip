// dense_layer: out = act([self_tab[self_index] | x] . W^T) for an [n, dim] operand that is already in HBM
// (the column-sliced gather wrote the per-destination means there).  MFMA-bound: 2*n*K*H flop against
// 4*n*(K+H) bytes is ~40 flop/byte at K = 256, H = 128 (ridge ~20), so unlike the gather this kernel is
// built around keeping the matrix pipes fed:
//   * persistent 256-thread blocks; wave w owns output columns [32w, 32w+32) and keeps its W slice
//     ([32, KP] fp32 = KP/2 VGPRs) in registers for the whole kernel;
//   * 32-row tiles, double-buffered in LDS; the NEXT tile's rows are requested from HBM/L2 (into VGPRs)
//     before the MFMA loop of the CURRENT tile starts, so their latency hides under ~4 us of MFMA work;
//   * one barrier per tile (the double buffer makes the second one unnecessary);
//   * operands as in sage_fused.hip: lane (i = l&31, h = l>>5) supplies A[i][8q+4h+t] / W[n0+i][8q+4h+t]
//     to MFMA 4q+t; LDS rows padded by one ds_read_b128 width.
#include "sage_internal.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct DenseArgs {
    const float* x; int64_t ldx; int dim;
    int n; const int32_t* n_dev; int n_off;
    const float* self_tab; int64_t ld_self; int self_rows; const int32_t* self_index;
    const int32_t* cnt; const int32_t* any_nonempty;    // nullable: rows with cnt == 0 become NaN when *any_nonempty (reference 0/0)
    const float* W; int64_t ldw; int out_dim; int act;
    float* out; int64_t ldo;
    sage_finish_t fin;
    const uint4* wsplit;      // nullable: W already split into bf16 planes in register order (sage_prepare_weights)
    // a launch that contracts ONE K chunk of the concat layer (dense_bf16x3_kernel's SRC / EPI forms): the chunk's pass inside the prepared
    // planes, the number of passes in the buffer (the "W holds a huge value" trailer sits behind them), and the chunk's first W column
    int wpass, wpasses, woff;
};

// Concat encoder (K = 2*dim): with KP <= 128 a wave keeps BOTH chunks of its W slice in registers; at KP = 256 the
// block grows to 8 waves, waves 4-7 own the second K chunk (their own W slice in registers, their own partial
// accumulator) and hand their partial sums to waves 0-3 through LDS -- W is never re-read per tile.
template <int KP, bool CONCAT>
__global__ __launch_bounds__((CONCAT && KP == 256) ? 512 : 256, 2) void dense_layer_kernel(const DenseArgs a) {
    constexpr bool KSPLIT = CONCAT && KP == 256;       // split K across two wave groups
    constexpr int M = 32, WAVES = KSPLIT ? 8 : 4;
    constexpr int LDA = KP + 4;
    constexpr int CHUNKS = CONCAT ? 2 : 1;
    constexpr int WCH = KSPLIT ? 1 : CHUNKS;           // K chunks one wave contracts
    constexpr int LG = KP / 4, RPP = 64 / LG, RPW = M / WAVES, PASSES = RPW / RPP;
    constexpr int BUF = CHUNKS * M * LDA;                   // floats per LDS buffer
    static_assert(RPW % RPP == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][CHUNKS][M][LDA] (+ [4][16][64] partials when KSPLIT)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nn = a.n;
    if (a.n_dev) nn = min(*a.n_dev + a.n_off, a.n);
    const int ntiles = (nn + M - 1) / M;
    if ((int)blockIdx.x < ntiles) {
        const bool nan_rule = (a.cnt && a.any_nonempty) ? (*a.any_nonempty != 0) : false;
        const int i32 = lane & 31, h = lane >> 5;
        const int n0 = (wave & 3) * 32;
        const int kgroup = wave >> 2;                   // 0, or 1 for the second K chunk (KSPLIT)
        const bool mfma_wave = n0 < a.out_dim;
        const int lg = lane & (LG - 1), sg = lane / LG;
        const int c0 = lg * 4;
        const bool col_ok = c0 < a.dim;
        const bool wrow_ok = mfma_wave && (n0 + i32) < a.out_dim;
        const float* wrow = a.W + (int64_t)min(n0 + i32, a.out_dim - 1) * a.ldw;

        float breg[WCH][KP / 2];
#pragma unroll
        for (int wc = 0; wc < WCH; ++wc) {
            const int chunk = KSPLIT ? kgroup : wc;
#pragma unroll
            for (int q = 0; q < KP / 8; ++q) {
                const int kc = 8 * q + 4 * h;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (wrow_ok && kc < a.dim) v = *reinterpret_cast<const f32x4*>(wrow + (int64_t)chunk * a.dim + kc);
                breg[wc][4 * q + 0] = v[0]; breg[wc][4 * q + 1] = v[1]; breg[wc][4 * q + 2] = v[2]; breg[wc][4 * q + 3] = v[3];
            }
        }

        f32x4 xr[PASSES], sr[CONCAT ? PASSES : 1];
        auto request_tile = [&](int tile) {                  // global -> VGPRs, no wait
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int g = tile * M + wave * RPW + p * RPP + sg;
                const bool valid = g < nn && col_ok;
                xr[p] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (valid) {
                    xr[p] = *reinterpret_cast<const f32x4*>(a.x + (int64_t)g * a.ldx + c0);
                    if (nan_rule && a.cnt[g] == 0) { const float q = __builtin_nanf(""); xr[p] = f32x4{q, q, q, q}; }
                }
                if (CONCAT) {
                    sr[p] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (valid) {
                        const int64_t s = a.self_index ? (int64_t)min(max(a.self_index[g], 0), a.self_rows - 1)
                                                       : (int64_t)min(g, a.self_rows - 1);
                        sr[p] = *reinterpret_cast<const f32x4*>(a.self_tab + s * a.ld_self + c0);
                    }
                }
            }
        };
        auto stage_tile = [&](float* buf) {                   // VGPRs -> LDS
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int r = wave * RPW + p * RPP + sg;
                *reinterpret_cast<f32x4*>(buf + ((CHUNKS - 1) * M + r) * LDA + c0) = xr[p];
                if (CONCAT) *reinterpret_cast<f32x4*>(buf + r * LDA + c0) = sr[p];
            }
        };

        int tile = blockIdx.x, b = 0;
        request_tile(tile);
        for (; tile < ntiles; tile += gridDim.x, b ^= 1) {
            float* buf = lds + b * BUF;
            stage_tile(buf);
            __syncthreads();
            const int next = tile + gridDim.x;
            if (next < ntiles) request_tile(next);            // in flight during the MFMA loop below
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            if (mfma_wave) {
#pragma unroll
                for (int wc = 0; wc < WCH; ++wc) {
                    const int chunk = KSPLIT ? kgroup : wc;
                    const float* abase = buf + (chunk * M + i32) * LDA + 4 * h;
#pragma unroll
                    for (int q = 0; q < KP / 8; ++q) {
                        const f32x4 av = *reinterpret_cast<const f32x4*>(abase + 8 * q);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], breg[wc][4 * q + t], acc, 0, 0, 0);
                    }
                }
            }
            if constexpr (KSPLIT) {                          // waves 4-7 -> LDS -> waves 0-3
                float* red = lds + 2 * BUF + (wave & 3) * 16 * 64;
                if (kgroup == 1) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[e * 64 + lane] = acc[e];
                }
                __syncthreads();
                if (kgroup == 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] += red[e * 64 + lane];
                }
            }
            if (mfma_wave && kgroup == 0) {
                const int col = n0 + i32;
                if (col < a.out_dim) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int g = tile * M + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                        if (g < nn) a.out[(int64_t)g * a.ldo + col] = sage_activate(acc[reg], a.act);
                    }
                }
            }
        }
    }
    sage_finish_block(a.fin, (int)gridDim.x);
}

template <int KP, bool CONCAT>
int launch(const DenseArgs& a, hipStream_t st) {
    constexpr bool KSPLIT = CONCAT && KP == 256;
    constexpr size_t lds = ((size_t)2 * (CONCAT ? 2 : 1) * 32 * (KP + 4) + (KSPLIT ? 4 * 16 * 64 : 0)) * sizeof(float);
    static bool configured = false;
    if (!configured) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)dense_layer_kernel<KP, CONCAT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)lds) != hipSuccess) {
            sage_set_error("layer_dense: cannot reserve %zu bytes of LDS", lds);
            return SAGE_ELAUNCH;
        }
        configured = true;
    }
    // one persistent block per CU (24.9 us) beat two (26.3 us) on the config-3 contraction: the prefetch only pays
    // when a block owns >= 2 tiles, and one block per CU leaves wave slots for another batch's kernels
#ifndef SAGE_DENSE_PER_CU
#define SAGE_DENSE_PER_CU 1
#endif
    const int grid = min(sage_cdiv(a.n, 32), SAGE_DENSE_PER_CU * kNumCU);
    hipLaunchKernelGGL((dense_layer_kernel<KP, CONCAT>), dim3(grid), dim3(KSPLIT ? 512 : 256), lds, st, a);
    SAGE_CHECK_LAUNCH("dense_layer_kernel");
    return SAGE_OK;
}

// ---- the same contraction on the bf16 matrix pipe, fp32-accurate (split operands) ----------------------------------
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (1/16 of bf16 MFMA), and with it this kernel spends ~8200
// matrix-pipe cycles per 32-row tile.  An fp32 value is exactly hi + mid + lo with three bf16 terms (8 + 8 + 8
// significant bits, round-to-nearest each time, the remainders are exact in fp32), and a bf16 x bf16 product is exact
// in the MFMA's fp32 accumulator, so
//     x * w  =  hi*hi + (hi*mid + mid*hi) + (mid*mid + hi*lo + lo*hi)  +  O(2^-24 |x||w|)
// -- six v_mfma_f32_32x32x16_bf16 (32 cycles each for 16 k) per k-step instead of eight 64-cycle fp32 MFMAs:
// 192 vs 512 cycles, with an error of the order of the fp32 rounding the reference's sgemm makes anyway (the 1e-5
// parity gate is relative to the row maximum; measured max error 1e-6, tests/test_gpu_ops.py, bench.py parity gate).
// Structure: 8 waves per block; wave w owns output columns [32(w&3), +32) of K half (w>>2); its W slice is split once,
// at kernel start, into three bf16 planes that stay in VGPRs (96 at K = 256).  A tile's rows are split once by the
// threads that stage them (v_cvt_pk_bf16_f32 / v_pk_add_f32, ~5 VALU per element) into three bf16 LDS planes whose rows
// are padded by 16 B, so an A operand is ONE ds_read_b128.  The two K halves meet through LDS.  Double-buffered tiles,
// next tile's rows in flight during the MFMA loop, as above.
#ifndef SAGE_MP_TG
#define SAGE_MP_TG 1
#endif
#ifdef SAGE_DENSE_STAMPS     // diagnostic build (experiments/): where a persistent block's time goes; never in the product library
__device__ unsigned long long g_dense_stamps[512 * 40];
extern "C" int sage_debug_dense_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dense_stamps), sizeof(g_dense_stamps)) == hipSuccess ? 0 : -1;
}
#define STAMP(i) do { if (threadIdx.x == 0 && (i) < 40) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_dense_stamps[blockIdx.x * 40 + (i)] = t_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f32x8 = __attribute__((ext_vector_type(8))) float;

// |x| >= 2^127, +-Inf or NaN (exponent field 254 or 255): the three-term split is not exact there -- RNE to bf16 can
// round the first term up to Inf, and Inf - Inf poisons the remainders -- so a tile (or a weight slice) that holds such a
// value is recomputed by exact_row_dot below, a plain fp32 fma chain with torch.mm's Inf / NaN behaviour.
__device__ inline bool huge4(const f32x4 x) {
    bool h = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) h |= (__float_as_uint(x[e]) & 0x7F800000u) >= 0x7F000000u;
    return h;
}

__device__ inline void split3(const f32x4 x, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
    hi = __builtin_convertvector(x, bf16x4);
    const f32x4 r1 = x - __builtin_convertvector(hi, f32x4);
    mid = __builtin_convertvector(r1, bf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(mid, f32x4);
    lo = __builtin_convertvector(r2, bf16x4);
}

// Block barrier for data exchanged through LDS only.  __syncthreads() carries a workgroup-scope fence, and on gfx9 a release
// fence is `s_waitcnt vmcnt(0)`: it DRAINS every global load in flight -- the W slice (24 KiB per wave) requested in the
// prologue, the next tile's rows requested before the MFMA loop -- at each of the two barriers per tile.  Here only the LDS
// queue is waited for; the compiler still waits for a load where its value is used.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// out[g][col] before the activation as an fp32 fma chain over k (the [self | agg] order of encoders.py:54): the slow, exact
// form for tiles that hold |x| >= 2^127 / Inf / NaN (never taken on ordinary data: one LDS word per tile decides)
__device__ inline float exact_row_dot(const DenseArgs& a, bool concat, int g, int col, bool nan_rule) {
    const float* wrow = a.W + (int64_t)col * a.ldw;
    float acc = 0.f;
    int koff = 0;
    if (concat) {
        const int64_t s = a.self_index ? (int64_t)min(max(a.self_index[g], 0), a.self_rows - 1) : (int64_t)min(g, a.self_rows - 1);
        const float* sr = a.self_tab + s * a.ld_self;
        for (int k = 0; k < a.dim; ++k) acc = fmaf(sr[k], wrow[k], acc);
        koff = a.dim;
    }
    if (nan_rule && a.cnt[g] == 0) return __builtin_nanf("");
    const float* xr = a.x + (int64_t)g * a.ldx;
    for (int k = 0; k < a.dim; ++k) acc = fmaf(xr[k], wrow[koff + k], acc);
    return acc;
}

__device__ inline bool row_is_huge(const DenseArgs& a, bool concat, int g, bool nan_rule) {
    bool h = nan_rule && a.cnt[g] == 0;
    const float* xr = a.x + (int64_t)g * a.ldx;
    for (int k = 0; k < a.dim; ++k) h |= (__float_as_uint(xr[k]) & 0x7F800000u) >= 0x7F000000u;
    if (concat) {
        const int64_t s = a.self_index ? (int64_t)min(max(a.self_index[g], 0), a.self_rows - 1) : (int64_t)min(g, a.self_rows - 1);
        const float* sr = a.self_tab + s * a.ld_self;
        for (int k = 0; k < a.dim; ++k) h |= (__float_as_uint(sr[k]) & 0x7F800000u) >= 0x7F000000u;
    }
    return h;
}

// The exact fp32 fma chain of exact_row_dot for ONE K chunk of the concat layer: the nodes' own rows (self, W columns [0, dim), the chain
// starts at 0 and its partial sum is stored as it is) or the neighbour means (agg, W columns [woff, woff + dim), the chain CONTINUES from
// the partial sum the self launch left in `out`): over both launches bit for bit the chain over [self | agg].
__device__ inline float chunk_exact_dot(const DenseArgs& a, bool self_chunk, int g, int col, bool nan_rule) {
    const float* wrow = a.W + (int64_t)col * a.ldw;
    if (self_chunk) {
        const int64_t s = a.self_index ? (int64_t)min(max(a.self_index[g], 0), a.self_rows - 1) : (int64_t)min(g, a.self_rows - 1);
        const float* sr = a.self_tab + s * a.ld_self;
        float acc = 0.f;
        for (int k = 0; k < a.dim; ++k) acc = fmaf(sr[k], wrow[k], acc);
        return acc;
    }
    if (nan_rule && a.cnt[g] == 0) return __builtin_nanf("");
    float acc = a.out[(int64_t)g * a.ldo + col];
    const float* xr = a.x + (int64_t)g * a.ldx;
    for (int k = 0; k < a.dim; ++k) acc = fmaf(xr[k], wrow[a.woff + k], acc);
    return acc;
}

__device__ inline bool chunk_row_is_huge(const DenseArgs& a, bool self_chunk, int g, bool nan_rule) {
    bool h = false;
    if (self_chunk) {
        const int64_t s = a.self_index ? (int64_t)min(max(a.self_index[g], 0), a.self_rows - 1) : (int64_t)min(g, a.self_rows - 1);
        const float* sr = a.self_tab + s * a.ld_self;
        for (int k = 0; k < a.dim; ++k) h |= (__float_as_uint(sr[k]) & 0x7F800000u) >= 0x7F000000u;
    } else {
        h = nan_rule && a.cnt[g] == 0;
        const float* xr = a.x + (int64_t)g * a.ldx;
        for (int k = 0; k < a.dim; ++k) h |= (__float_as_uint(xr[k]) & 0x7F800000u) >= 0x7F000000u;
    }
    return h;
}

// MP (KP = 256 only): rows wider than 256 -- every K chunk takes ceil(dim / 256) passes (Pubmed 500, Cora 1433+3 pad).
// PREP: W arrives as the planes of sage_prepare_weights (a.wsplit; one-pass shapes only) -- a compile-time property, so that no
// join of two W paths stands between the W loads and their first use.
// SRC / EPI (round 4; CONCAT = false, prepared planes only): ONE K chunk of the 512-deep concat layer per launch, so that the chunk of the
// nodes' own rows -- which needs the sampling only -- can run BESIDE the gather (role pipeline, SAGE_STAGE_CONTRACT1_SELF) and the means'
// chunk finishes the layer: SRC 1 = rows come from self_tab through self_index, 0 = from x;  EPI 1 = out := partial sums (no activation),
// EPI 2 = out := act(out + sums), 0 = out := act(sums).  Compile-time, for the reason PREP is (no join of two paths between a load and its use).
template <int KP, bool CONCAT, bool MP, bool PREP = false, int SRC = 0, int EPI = 0>
__global__ __launch_bounds__(512) void dense_bf16x3_kernel(const DenseArgs a) {
    static_assert((SRC == 0 && EPI == 0) || (!CONCAT && !MP && PREP), "chunk launches: non-concat one-pass kernel on prepared planes");
    constexpr int M = 32, WAVES = 8;
    constexpr int CHUNKS = CONCAT ? 2 : 1;
    // The 512-deep concat layer (two 256-wide chunks) is contracted in two K PASSES so that a wave's W slice stays at
    // 96 VGPRs: the accumulators of a GROUP of up to TG tiles stay in registers while pass 0 (the nodes' own rows) and
    // pass 1 (the neighbour means) run over the group, so W is fetched and split once per pass and group, not per tile.
    static_assert(!MP || KP == 256, "multi-pass rows use 256-wide passes");
    constexpr int PCH = (CONCAT && KP < 256) ? 2 : 1;    // K chunks staged per pass
    constexpr bool MULTI = MP || (CONCAT && KP == 256);  // more than one pass: group accumulators
    constexpr int KPASS = PCH * KP;                      // K columns per pass (<= 256)
    constexpr int KH = KPASS / 2, STEPS = KH / 16;
    constexpr int TG = MP ? SAGE_MP_TG : (MULTI ? 4 : 1);
    constexpr int LDB = KPASS + 8;                       // bf16 elements per LDS row (+16 B: conflict-free ds_read_b128)
    constexpr int PL = M * LDB;                          // elements per plane
    constexpr int LG = KP / 4, RPP = 64 / LG, RPW = M / WAVES, PASSES = RPW / RPP;
    static_assert(RPW % RPP == 0 && KH % 16 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int PLD = 128 + 4;                         // floats per row of a partial-sum plane (+16 B: conflict-free b128 reads)
    __bf16* lds = reinterpret_cast<__bf16*>(lds_raw);                                   // [2][3][M][LDB]
    float* part = reinterpret_cast<float*>(lds_raw + (size_t)2 * 3 * PL * sizeof(__bf16));  // [2 K halves][M][PLD]
    constexpr int kBadListCap = 28;
    int* flags = reinterpret_cast<int*>(part + 2 * M * PLD);     // [0], [1]: the tile staged in buffer b holds a huge value; [2]: W does;
                                                                 // [3]: tiles to redo exactly, [4..31]: their indices (all of the block's if more);
                                                                 // [32 + 32 b + r]: row r of the tile staged in buffer b holds a huge value

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nn = a.n;
    if (a.n_dev) nn = min(*a.n_dev + a.n_off, a.n);
    const int ntiles = (nn + M - 1) / M;
    if ((int)blockIdx.x < ntiles) {
        const bool nan_rule = (a.cnt && a.any_nonempty) ? (*a.any_nonempty != 0) : false;
        const int i32 = lane & 31, h = lane >> 5;
        const int n0 = (wave & 3) * 32;
        const int kgroup = wave >> 2;
        const bool mfma_wave = n0 < a.out_dim;
        const int lg = lane & (LG - 1), sg = lane / LG;
        const int c0 = lg * 4;
        const int stride = (int)gridDim.x;
        const bool vec_store = (a.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
        const int ppc = MP ? (a.dim + KP - 1) / KP : 1;          // passes per K chunk
        const int npass = (CHUNKS / PCH) * ppc;

        f32x4 xr[PCH][PASSES];
        int xc[PASSES];                                       // neighbour counts of the requested rows (0/0 rule only)
        // concat encoder: rows of the nodes' own features are fetched through an index (s1_nodes).  Index and row in one request
        // are two DEPENDENT round trips in front of every tile's MFMA loop; the indices of a block's next tile are requested
        // with the current tile's rows instead and are there when that tile's rows are asked for.
        // (Two-pass 512-deep layer only: same-box A/B, concat forward at config 3 86.6 -> 83.7 us.  In the one-pass concat kernel
        // (KP <= 128) the four index registers cross an occupancy step, 167 -> 172 VGPRs, and although the kernel alone gains --
        // 28.8 -> 26.3 us at config 5 -- the pipeline loses: 82.5 -> 86.0 us, three runs each.)
        constexpr bool SELF_AHEAD = (CONCAT && !MP && KP == 256) || SRC == 1;
        int sidx[SELF_AHEAD ? PASSES : 1];
        auto request_self_index = [&](int tile) {
            if constexpr (SELF_AHEAD) {
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const int g = min(tile * M + wave * RPW + p * RPP + sg, nn - 1);
                    sidx[p] = a.self_index ? a.self_index[g] : g;
                }
            }
        };
        // global -> VGPRs, no wait.  No lane-dependent branch anywhere near these loads: rows past the end and columns past
        // the row width are requested from clamped addresses and masked when the tile is staged.  (`x = 0; if (valid) x = load`
        // compiles to a divergent branch whose join COPIES the loaded registers -- a use, so the compiler waited for the loads
        // right where they were issued and nothing was in flight during the MFMA loop.)
        auto request_tile = [&](int tile, int pass) {
#pragma unroll
            for (int pc = 0; pc < PCH; ++pc) {
                const bool is_agg = SRC == 0 && ((pass / ppc) * PCH + pc) == CHUNKS - 1;      // the last K chunk is the neighbour mean (block-uniform)
                const int coff = min((pass % ppc) * KP + c0, a.dim - 4);         // this lane's first column of the chunk (dim % 4 == 0)
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const int g = min(tile * M + wave * RPW + p * RPP + sg, nn - 1);
                    if (is_agg) {
                        // streaming load: every row of the means is read once, and left in L2 it would evict the gather's hub rows
                        // of the NEXT batch, which runs beside this kernel (same-box A/B with the sampler's nt loads: -0.8 us per forward)
                        if constexpr (SAGE_AGG_LOAD == 1) xr[pc][p] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.x + (int64_t)g * a.ldx + coff));
                        else xr[pc][p] = *reinterpret_cast<const f32x4*>(a.x + (int64_t)g * a.ldx + coff);
                        if (nan_rule) xc[p] = a.cnt[g];
                    } else if constexpr (SELF_AHEAD) {
                        // the node's own row: its index was requested one tile ago (sidx), so this load does not wait for it
                        xr[pc][p] = *reinterpret_cast<const f32x4*>(a.self_tab + (int64_t)min(max(sidx[p], 0), a.self_rows - 1) * a.ld_self + coff);
                    } else {
                        const int64_t s = a.self_index ? (int64_t)min(max(a.self_index[g], 0), a.self_rows - 1)
                                                       : (int64_t)min(g, a.self_rows - 1);
                        xr[pc][p] = *reinterpret_cast<const f32x4*>(a.self_tab + s * a.ld_self + coff);
                    }
                }
                if constexpr (SELF_AHEAD)
                    if (!is_agg) request_self_index(tile + stride);       // a block's tiles come in this order in every pass structure but MP
            }
        };
        // EPI 2: the partial sums the self launch left in `out` for the two rows this thread finishes, requested before the tile's MFMA loop
        f32x4 prev[M / (WAVES * 2)];
        auto request_prev = [&](int tile) {
            if constexpr (EPI == 2) {
#pragma unroll
                for (int it = 0; it < M / (WAVES * 2); ++it) {
                    const int g = min(tile * M + wave * (M / WAVES) + 2 * it + (lane >> 5), nn - 1);
                    prev[it] = *reinterpret_cast<const f32x4*>(a.out + (int64_t)g * a.ldo + min((lane & 31) * 4, a.out_dim - 4));
                }
            }
        };
        int stage_seq = 0;                                    // stagings so far (block-uniform): the tag a "huge value" mark carries,
                                                              // so that marks never have to be cleared (a clear would race the next staging)
        auto stage_tile = [&](__bf16* buf, int bsel, int tile, int pass) {   // VGPRs -> mask -> split -> three bf16 LDS planes
            ++stage_seq;
#pragma unroll
            for (int pc = 0; pc < PCH; ++pc) {
                const bool is_agg = SRC == 0 && ((pass / ppc) * PCH + pc) == CHUNKS - 1;
                const bool col_ok = (pass % ppc) * KP + c0 < a.dim;
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const bool valid = col_ok && tile * M + wave * RPW + p * RPP + sg < nn;
                    const bool nanrow = nan_rule && is_agg && xc[p] == 0;         // aggregators.py:60-61 (0/0 rows of a batch that has non-empty ones)
                    const float q = __builtin_nanf("");
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[pc][p][e] = valid ? (nanrow ? q : xr[pc][p][e]) : 0.f;
                }
            }
            bool huge = false;
#pragma unroll
            for (int pc = 0; pc < PCH; ++pc)
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const bool hp = huge4(xr[pc][p]);
                    if (hp) flags[32 + 32 * bsel + wave * RPW + p * RPP + sg] = stage_seq;     // same value from every lane of the row
                    huge |= hp;
                }
            if (__any(huge) && lane == 0) flags[bsel] = stage_seq;
#pragma unroll
            for (int pc = 0; pc < PCH; ++pc)
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const int r = wave * RPW + p * RPP + sg;
                    bf16x4 hi, mid, lo;
                    split3(xr[pc][p], hi, mid, lo);
                    __bf16* dst = buf + r * LDB + pc * KP + c0;
                    *reinterpret_cast<bf16x4*>(dst) = hi;
                    *reinterpret_cast<bf16x4*>(dst + PL) = mid;
                    *reinterpret_cast<bf16x4*>(dst + 2 * PL) = lo;
                }
        };

        if (tid < 96) flags[tid] = 0;                         // ordered before the first staging by the barrier below
        lds_barrier();
        if constexpr (PREP)                                   // sage_prepare_weights left "W holds |w| >= 2^127 / Inf / NaN" behind the planes
            if (tid == 0 && a.wsplit[(size_t)(a.wpasses > 0 ? a.wpasses : npass) * WAVES * STEPS * 3 * 64].x != 0) flags[2] = 1;   // read after the first staging's barrier
        STAMP(0);
        request_self_index((int)blockIdx.x);
        request_tile((int)blockIdx.x, 0);                     // the first tile's rows travel while W is fetched and split

        // W slice -> three bf16 planes in VGPRs: bw[st][plane] = W[n0+i][kk .. kk+7], kk = pass*KPASS + kgroup*KH + 16 st + 8 h
        // in the [self | agg] K index space (chunk kk / KP, column kk % KP).  All loads of the slice are in flight at once.
        // (Staging W through LDS with row-contiguous loads was measured: 23.1 vs 21.1 us -- the four extra barriers cost
        // more than the uncoalesced but L2-resident 16-B loads.)
        const bool wrow_ok = mfma_wave && (n0 + i32) < a.out_dim;
        const float* wrow = a.W + (int64_t)min(n0 + i32, a.out_dim - 1) * a.ldw;
        bf16x8 bw[STEPS][3];
        auto load_w = [&](int pass) {
            if constexpr (PREP) {
                {
                    // planes prepared by sage_prepare_weights: [wave][step][plane][lane] x 16 B, so every load is one
                    // fully coalesced 1-KiB wave-instruction and nothing is split here (the strided fp32 loads + 16 split3
                    // per lane below took 8000 cycles per wave and ~15000 until the block's slowest wave had its slice:
                    // a third of the kernel, in-kernel s_memtime stamps)
                    // wave-uniform base (SGPR pair) + one lane offset: 24 per-lane 64-bit addresses (the planes span 24 KiB, beyond a
                    // load's immediate offset) cost 48 VGPRs and, in the 512-deep kernel, spills
                    const uint4* wp = a.wsplit + (size_t)((pass + a.wpass) * WAVES + __builtin_amdgcn_readfirstlane(wave)) * STEPS * 3 * 64;
#pragma unroll
                    for (int st = 0; st < STEPS; ++st)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            const uint4 v = wp[(st * 3 + pl) * 64 + lane];
                            bw[st][pl] = __builtin_bit_cast(bf16x8, v);   // no use of v here: the loads stay in flight (the "W holds a
                        }                                                 // huge value" mark comes from the buffer's trailer, see below)
                    return;
                }
            }
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const int kk = (pass % ppc) * KPASS + kgroup * KH + 16 * st + 8 * h;       // column inside the pass's first chunk ...
                const int chunk = (pass / ppc) * PCH + (MP ? 0 : kk / KP), kc = MP ? kk : kk % KP;   // ... or, two chunks per pass, inside its own
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
                if (wrow_ok && kc < a.dim) v0 = *reinterpret_cast<const f32x4*>(wrow + (int64_t)chunk * a.dim + kc);
                if (wrow_ok && kc + 4 < a.dim) v1 = *reinterpret_cast<const f32x4*>(wrow + (int64_t)chunk * a.dim + kc + 4);
                if (__any(huge4(v0) || huge4(v1)) && lane == 0) flags[2] = 1;
                bf16x4 h0, m0, l0, h1, m1, l1;
                split3(v0, h0, m0, l0);
                split3(v1, h1, m1, l1);
                bw[st][0] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                bw[st][1] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
                bw[st][2] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        if (!MULTI) load_w(0);
        STAMP(1);
        int stamp_i = 2;
        (void)stamp_i;

        int b = 0;
        // One group of up to TG tiles.  Called once ahead of the loop (the block's first group, straight-line code) and then from
        // the loop: at a loop header the compiler waits for EVERY load in flight (s_waitcnt vmcnt(0)), which in the first
        // iteration meant the whole W slice; peeled, the first tile is staged while W travels (only its own rows are waited
        // for: they were requested first) and its MFMA steps start as their W registers arrive.
        auto do_group = [&](const int t0) __attribute__((always_inline)) {
            f32x16 acc[TG];
            int bad[TG];                                      // block-uniform: some value of tile t (any pass) or of W is huge
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                bad[t] = 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
            }
            auto do_pass = [&](int pass) {
                if (MULTI) {
                    // keep the scheduler from hoisting these loads above the previous pass's MFMAs: two live copies of
                    // the W slice (2 x 96 VGPRs) spilled ~100 registers to scratch
                    __builtin_amdgcn_sched_barrier(0);
                    load_w(pass);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    const int tile = t0 + t * stride;
                    if (tile >= ntiles) continue;             // block-uniform
                    __bf16* buf = lds + b * 3 * PL;
                    stage_tile(buf, b, tile, pass);
                    lds_barrier();
                    {   // bit 0: the tile (goes on the redo list); bits 1, 2: the two rows THIS thread stores in the epilogue; bit 3: W
                        const int re = wave * (M / WAVES) + (lane >> 5);
                        bad[t] |= (flags[b] == stage_seq ? 1 : 0) | (flags[32 + 32 * b + re] == stage_seq ? 2 : 0) |
                                  (flags[32 + 32 * b + re + 2] == stage_seq ? 4 : 0) | (flags[2] != 0 ? 9 : 0);
                    }
                    STAMP(stamp_i); ++stamp_i;
                    // the next work item's rows are in flight during the MFMA loop below
                    request_prev(tile);
                    if (t + 1 < TG && tile + stride < ntiles) request_tile(tile + stride, pass);
                    else if (pass + 1 < npass) request_tile(t0, pass + 1);
                    else if (t0 + TG * stride < ntiles) request_tile(t0 + TG * stride, 0);
                    if (mfma_wave) {
                        const __bf16* abase = buf + i32 * LDB + kgroup * KH + 8 * h;
#pragma unroll
                        for (int st = 0; st < STEPS; ++st) {
                            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(abase + 16 * st);
                            const bf16x8 am = *reinterpret_cast<const bf16x8*>(abase + PL + 16 * st);
                            const bf16x8 al = *reinterpret_cast<const bf16x8*>(abase + 2 * PL + 16 * st);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bw[st][0], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bw[st][2], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bw[st][1], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bw[st][0], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bw[st][1], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bw[st][0], acc[t], 0, 0, 0);
                        }
                    }
                    STAMP(stamp_i); ++stamp_i;
                    b ^= 1;
                }
            
            };
            if constexpr (MP) {
                for (int pass = 0; pass < npass; ++pass) do_pass(pass);
            } else {
#pragma unroll
                for (int pass = 0; pass < CHUNKS / PCH; ++pass) do_pass(pass);
            }
            // Epilogue: both K halves put their 32 x 32 partial sums into LDS as [row][column] planes, then ALL eight waves
            // add the halves, apply the activation and store whole rows: 64 lanes x 16 B = two 512-B rows per instruction, two
            // instructions per wave and tile.  (Before: K half 1 -> LDS -> K half 0, whose four waves then issued 16 scalar
            // 4-byte stores per lane: 5300 of a tile's 9000 cycles, in-kernel stamps.)  Sum order unchanged: half 0 + half 1.
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int tile = t0 + t * stride;
                if (tile >= ntiles) continue;
                if (TG > 1 && t > 0) lds_barrier();           // the previous tile's partial sums have been read
                if (mfma_wave) {
                    float* mine = part + (size_t)kgroup * M * PLD + n0 + i32;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) mine[((reg & 3) + 8 * (reg >> 2) + 4 * h) * PLD] = acc[t][reg];
                }
                lds_barrier();
                if (tid == 0 && (bad[t] & 1)) { const int i = flags[3]++; if (i < kBadListCap) flags[4 + i] = tile; }   // only thread 0 touches these
#pragma unroll
                for (int it = 0; it < M / (WAVES * 2); ++it) {
                    const int row = wave * (M / WAVES) + 2 * it + (lane >> 5);
                    const int col = (lane & 31) * 4;
                    const int g = tile * M + row;
                    if (g < nn && col < a.out_dim) {
                        const f32x4 p0 = *reinterpret_cast<const f32x4*>(part + row * PLD + col);
                        const f32x4 p1 = *reinterpret_cast<const f32x4*>(part + (M + row) * PLD + col);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (EPI == 1) v[e] = p0[e] + p1[e];
                            else if constexpr (EPI == 2) v[e] = sage_activate(prev[it][e] + (p0[e] + p1[e]), a.act);
                            else v[e] = sage_activate(p0[e] + p1[e], a.act);
                        }
                        if (bad[t] & (8 | (2 << it))) continue;    // a row with a huge value (or huge W): redone exactly below; other rows of
                                                                   // the tile keep the MFMA result, so a row never depends on its tile mates
                        float* dst = a.out + (int64_t)g * a.ldo + col;
                        if (col + 3 < a.out_dim && vec_store) {
                            if constexpr (EPI == 1) *reinterpret_cast<f32x4*>(dst) = v;          // re-read by the means' launch: not a streaming store
                            else sage_store_stream<SAGE_H1_STORE>(reinterpret_cast<f32x4*>(dst), v);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col + e < a.out_dim) __builtin_nontemporal_store(v[e], dst + e);
                        }
                    }
                }
                STAMP(stamp_i); ++stamp_i;
            }
        };
        if constexpr (!MULTI) {
            do_group((int)blockIdx.x);
            for (int t0 = (int)blockIdx.x + TG * stride; t0 < ntiles; t0 += TG * stride) do_group(t0);
        } else {                                              // W is (re)loaded inside every pass: nothing to keep in flight across the
            for (int t0 = (int)blockIdx.x; t0 < ntiles; t0 += TG * stride) do_group(t0);   // loop header, and a second copy of the
        }                                                     // body only adds register pressure (spills at 256 VGPRs)
        STAMP(39);
        // Tiles that held |x| >= 2^127 / Inf / NaN (or all tiles, when W does): the exact fp32 fma chain, outside the loop above
        // so that it costs the ordinary path no register.  Block-uniform; zero iterations on ordinary data.
        lds_barrier();
        const int nbad = flags[3];
        if (nbad > 0) {
            const bool all = nbad > kBadListCap || flags[2] != 0;
            const int per = M * a.out_dim;
            for (int tb = (int)blockIdx.x, li = 0; all ? (tb < ntiles) : (li < nbad); tb += stride, ++li) {
                const int tile = all ? tb : flags[4 + li];
                for (int idx = tid; idx < per; idx += (int)blockDim.x) {
                    const int g = tile * M + idx / a.out_dim, col = idx % a.out_dim;
                    if constexpr (SRC != 0 || EPI != 0) {
                        if (g < nn && (flags[2] != 0 || chunk_row_is_huge(a, SRC == 1, g, nan_rule))) {
                            const float d = chunk_exact_dot(a, SRC == 1, g, col, nan_rule);
                            a.out[(int64_t)g * a.ldo + col] = EPI == 1 ? d : sage_activate(d, a.act);
                        }
                    } else if (g < nn && (flags[2] != 0 || row_is_huge(a, CONCAT, g, nan_rule)))
                        a.out[(int64_t)g * a.ldo + col] = sage_activate(exact_row_dot(a, CONCAT, g, col, nan_rule), a.act);
                }
            }
        }
    }
    sage_finish_block(a.fin, (int)gridDim.x);
}

template <int KP, bool CONCAT, bool MP = false, bool PREP = false, int SRC = 0, int EPI = 0>
int launch_bf16x3(const DenseArgs& a, hipStream_t st) {
    if constexpr (!PREP)
        if (a.wsplit) return launch_bf16x3<KP, CONCAT, MP, true>(a, st);
    constexpr int KPASS = (CONCAT && KP < 256) ? 2 * KP : KP;
    constexpr size_t lds = (size_t)2 * 3 * 32 * (KPASS + 8) * 2 + (size_t)2 * 32 * (128 + 4) * sizeof(float) + 384;
    static bool configured = false;
    if (!configured) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)dense_bf16x3_kernel<KP, CONCAT, MP, PREP, SRC, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)lds) != hipSuccess) {
            sage_set_error("layer_dense: cannot reserve %zu bytes of LDS", lds);
            return SAGE_ELAUNCH;
        }
        configured = true;
    }
    // Persistent blocks on 3/4 of the CUs: alone that costs 1-2 us (4 tile rounds instead of 3), with a second batch in
    // flight it is worth 3 us per forward -- a block holds 344 of a SIMD's 512 VGPRs and 117 KB of LDS for its whole life,
    // and the other batch's latency-bound kernels get the remaining CUs to themselves (same-box A/B: 184-224 blocks
    // 81.3-81.9 us, 256 blocks 84.2, 160 blocks 83.1).
    const int grid = min(sage_cdiv(a.n, 32), sage_tunables().dense_blocks);
    hipLaunchKernelGGL((dense_bf16x3_kernel<KP, CONCAT, MP, PREP, SRC, EPI>), dim3(grid), dim3(512), lds, st, a);
    SAGE_CHECK_LAUNCH("dense_bf16x3_kernel");
    return SAGE_OK;
}

// ---- the same contraction with PRODUCER and CONSUMER waves (round 3) --------------------------------------------------------------
// dense_bf16x3_kernel runs its eight waves in lock step: everybody stages a tile (global -> split -> LDS), barrier, everybody
// issues MFMAs, barrier, everybody runs the epilogue.  In-kernel stamps (round 2): of a tile's 8.1 k cycles the matrix pipes work
// 2.5-3 k; they idle during staging (2.3 k) and the epilogue (3.2 k).  Here the eight waves of a 512-thread block have ROLES, one
// wave of each role per SIMD:
//   waves 0-3   CONSUMERS: wave w keeps the W slice of output columns [32 w, +32) over the WHOLE K range (<= 256: up to 192 VGPRs of
//               prepared planes) and does nothing but ds_read_b128 + v_mfma_f32_32x32x16_bf16 on the tile the producers staged one
//               phase earlier (one accumulation chain of 6 * K / 16 MFMAs: no K halves to add); its 32 x 32 sums go to a [32][128]
//               fp32 plane in LDS;
//   waves 4-7   PRODUCERS: split tile p (its rows were requested one phase ago) into the three bf16 planes of the other LDS buffer,
//               request the rows of tile p + 1, and finish tile p - 2: read the result plane row-wise, (add the partial sums of an
//               earlier K chunk,) activate, store 512-B rows.
// ONE barrier per tile; per phase a SIMD's matrix pipe has 6 * K / 16 MFMAs (3072 cycles at K = 256) of work and the producer's
// vector instructions ride in the 24 of every 32 cycles in which an MFMA leaves the vector issue free (MI355X_MICROARCH.md).
// (First form, measured and replaced: twelve waves, eight consumers that each kept one K HALF in 96 VGPRs and ADDED their sums into the
// plane with ds_add_f32 -- an LDS float atomic costs ~200 cycles per wave-instruction and blocks the LDS pipe for everybody: 128 of
// them per tile made a phase 27 k cycles, in-kernel stamps of experiments/r03/pc_stamps.py.)
// The 512-deep concat layer (2 x 256 columns do not fit the register file) is TWO launches of this kernel instead of two passes that
// reload W per group of tiles: `self` chunk -> partial sums into `out` (epi 1; it depends on the sampling only and runs beside
// the gather), then `agg` chunk with out = act(out + acc) (epi 2): every block keeps ONE W slice for its whole life.
#ifdef SAGE_DENSE_STAMPS
#define STAMP_T(t, i) do { if (threadIdx.x == (t) && (i) < 40) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_dense_stamps[blockIdx.x * 40 + (i)] = t_; } } while (0)
#else
#define STAMP_T(t, i) do { } while (0)
#endif
struct PcMode {
    int src;        // PCH == 1: 0 = the means (a.x), 1 = the nodes' own rows (a.self_tab through a.self_index); PCH == 2: both
    int epi;        // 0: out = act(acc)    1: out = acc (partial sums of a K chunk)    2: out = act(out + acc)
    int wpass;      // the pass of the prepared planes this launch contracts
    int wpasses;    // passes in the prepared buffer (the "W holds a huge value" trailer sits behind them)
    int agg_woff;   // first W column of the means' chunk (exact cold path): a.dim with a self chunk in front, else 0
};

// the exact fp32 fma chain of exact_row_dot, restricted to the chunk(s) this launch contracts and continuing from the partial sum
// an earlier launch left in `out` (epi 2): bit for bit the chain over [self | agg]
__device__ inline float pc_exact_dot(const DenseArgs& a, const PcMode& md, bool use_self, bool use_agg, int g, int col, bool nan_rule) {
    const float* wrow = a.W + (int64_t)col * a.ldw;
    float acc = md.epi == 2 ? a.out[(int64_t)g * a.ldo + col] : 0.f;
    if (use_self) {
        const int64_t s = a.self_index ? (int64_t)min(max(a.self_index[g], 0), a.self_rows - 1) : (int64_t)min(g, a.self_rows - 1);
        const float* sr = a.self_tab + s * a.ld_self;
        for (int k = 0; k < a.dim; ++k) acc = fmaf(sr[k], wrow[k], acc);
    }
    if (use_agg) {
        if (nan_rule && a.cnt[g] == 0) return __builtin_nanf("");
        const float* xr = a.x + (int64_t)g * a.ldx;
        for (int k = 0; k < a.dim; ++k) acc = fmaf(xr[k], wrow[md.agg_woff + k], acc);
    }
    return acc;
}

__device__ inline bool pc_row_is_huge(const DenseArgs& a, bool use_self, bool use_agg, int g, bool nan_rule) {
    bool h = false;
    if (use_agg) {
        h = nan_rule && a.cnt[g] == 0;
        const float* xr = a.x + (int64_t)g * a.ldx;
        for (int k = 0; k < a.dim; ++k) h |= (__float_as_uint(xr[k]) & 0x7F800000u) >= 0x7F000000u;
    }
    if (use_self) {
        const int64_t s = a.self_index ? (int64_t)min(max(a.self_index[g], 0), a.self_rows - 1) : (int64_t)min(g, a.self_rows - 1);
        const float* sr = a.self_tab + s * a.ld_self;
        for (int k = 0; k < a.dim; ++k) h |= (__float_as_uint(sr[k]) & 0x7F800000u) >= 0x7F000000u;
    }
    return h;
}

// SRC and EPI are md.src / md.epi as COMPILE-TIME values: a uniform run-time branch around a load is still a branch, and at its join the
// compiler waits for every load in flight (s_waitcnt vmcnt(0) in front of each of a tile's eight row requests: 8 k cycles per phase,
// in-kernel stamps).  For the same reason the optional index / count loads below are unconditional loads from a harmless address.
template <int KP, int PCH, int SRC, int EPI>
__global__ __launch_bounds__(512) void dense_pc_kernel(const DenseArgs a, const PcMode md) {
    constexpr int M = 32, PW = 4;                        // rows per tile, producer waves
    constexpr int KPASS = PCH * KP, KH = KPASS / 2, STEPS = KH / 16;     // STEPS: k-steps of one K half (the prepared planes' unit)
    constexpr int LDB = KPASS + 8, PL = M * LDB;         // bf16 elements per LDS row (+16 B) and per plane
    constexpr int PLD = 128 + 8;                         // floats per row of the result plane
    constexpr int LG = KP / 4, RPP = 64 / LG, RPW = M / PW, PASSES = RPW / RPP;
    constexpr int kListCap = 28;
    static_assert(KPASS <= 256 && KH % 16 == 0 && RPW % RPP == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __bf16* planes = reinterpret_cast<__bf16*>(lds_raw);                                           // [2][3][M][LDB]
    float* part = reinterpret_cast<float*>(lds_raw + (size_t)2 * 3 * PL * sizeof(__bf16));         // [2][M][PLD]
    int* flags = reinterpret_cast<int*>(part + 2 * M * PLD);   // [0]: W holds a huge value; [1]: tiles to redo exactly; [2 ..]: their indices

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int nn = a.n;
    if (a.n_dev) nn = min(*a.n_dev + a.n_off, a.n);
    const int ntiles = (nn + M - 1) / M;
    const int stride = (int)gridDim.x;
    constexpr bool use_self = PCH == 2 || SRC == 1, use_agg = PCH == 2 || SRC == 0;
    const bool nan_rule = (use_agg && a.cnt && a.any_nonempty) ? (*a.any_nonempty != 0) : false;
    if ((int)blockIdx.x < ntiles) {
        const int nt = (ntiles - 1 - (int)blockIdx.x) / stride + 1;      // this block's tiles: blockIdx.x + j * stride, j < nt
        if (tid == 0) { flags[0] = 0; flags[1] = 0; }
        STAMP_T(0, 0);
        STAMP_T(256, 20);
        lds_barrier();
        if (wave < 4) {
            // ------------------------------------------------------------------ consumer: output columns [32 wave, +32), the whole K range
            const int i32 = lane & 31, h = lane >> 5;
            const int n0 = wave * 32;
            const bool mfma_wave = n0 < a.out_dim;
            bf16x8 bw[2 * STEPS][3];                      // the planes of both K halves (prepared per half: "waves" w and w + 4 of the lock-step kernel)
            {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const uint4* wp = a.wsplit + (size_t)(md.wpass * 8 + half * 4 + wave) * STEPS * 3 * 64;
#pragma unroll
                    for (int st = 0; st < STEPS; ++st)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) bw[half * STEPS + st][pl] = __builtin_bit_cast(bf16x8, wp[(st * 3 + pl) * 64 + lane]);
                }
            }
            if (tid == 0 && a.wsplit[(size_t)md.wpasses * 8 * STEPS * 3 * 64].x != 0) flags[0] = 1;
            STAMP_T(0, 1);
            lds_barrier();                                            // phase 0: the producers staged tile 0
            STAMP_T(0, 2);
            for (int j = 0; j < nt; ++j) {
                const __bf16* buf = planes + (j & 1) * 3 * PL;
                if (mfma_wave) {
                    // Two accumulators, one per K half, added at the end: the sums (and their rounding) of the lock-step kernel, whose two
                    // wave groups own a K half each -- bit for bit the same result, and half the length of an accumulation chain.
                    // A operands one k-step ahead of their MFMAs and no further: left alone the scheduler hoists every ds_read_b128 to the
                    // top (6 * STEPS * 4 VGPRs on top of the 192 of W).
                    f32x16 acc[2];
#pragma unroll
                    for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
                    const __bf16* abase = buf + i32 * LDB + 8 * h;
                    bf16x8 av[2][3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) av[0][pl] = *reinterpret_cast<const bf16x8*>(abase + pl * PL);
#pragma unroll
                    for (int st = 0; st < 2 * STEPS; ++st) {
                        if (st + 1 < 2 * STEPS) {
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) av[(st + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(abase + pl * PL + 16 * (st + 1));
                        }
                        const bf16x8 ah = av[st & 1][0], am = av[st & 1][1], al = av[st & 1][2];
                        f32x16& ac = acc[st / STEPS];
#ifdef PC_NO_MFMA
                        ac[0] += (float)ah[0] + (float)am[0] + (float)al[0];
                        continue;
#endif
                        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bw[st][0], ac, 0, 0, 0);
                        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bw[st][2], ac, 0, 0, 0);
                        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bw[st][1], ac, 0, 0, 0);
                        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bw[st][0], ac, 0, 0, 0);
                        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bw[st][1], ac, 0, 0, 0);
                        ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bw[st][0], ac, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    float* mine = part + (size_t)(j & 1) * M * PLD + n0 + i32;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) mine[((reg & 3) + 8 * (reg >> 2) + 4 * h) * PLD] = acc[0][reg] + acc[1][reg];
                }
                STAMP_T(0, 3 + 2 * j);
                lds_barrier();
                STAMP_T(0, 4 + 2 * j);
            }
        } else {
            // ------------------------------------------------------------------ producer
#ifdef PC_PRIO
            __builtin_amdgcn_s_setprio(PC_PRIO);
#endif
            const int pw = wave - 4;
            const int lg = lane & (LG - 1), sg = lane / LG;
            const int c0 = lg * 4;
            const bool col_ok = c0 < a.dim;
            const int coff = min(c0, a.dim - 4);
            const bool vec_out = (a.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
            f32x4 x0[PCH][PASSES];                       // the rows of the tile in flight (requested right after the previous tile was staged)
            int c0n[PASSES];                             // their neighbour counts (0/0 rule)
            int sidx[PASSES];                            // own-row indices of the NEXT tile to be requested
            f32x4 prev[4];                               // epi 2: the earlier chunk's partial sums of the tile being finished
            const bool has_index = a.self_index != nullptr;
            const int32_t* index_or_any = has_index ? a.self_index : reinterpret_cast<const int32_t*>(a.x);   // [nn] ints are readable at either
            const int32_t* cnt_or_any = nan_rule ? a.cnt : reinterpret_cast<const int32_t*>(a.x);
            auto request_index = [&](int j) {
                if constexpr (use_self) {
                    const int tile = (int)blockIdx.x + min(j, nt - 1) * stride;          // past the block's last tile: a harmless repeat
#pragma unroll
                    for (int p = 0; p < PASSES; ++p) {
                        const int g = min(tile * M + pw * RPW + p * RPP + sg, nn - 1);
                        const int v = index_or_any[g];
                        sidx[p] = has_index ? v : g;
                    }
                }
            };
            // global -> VGPRs, no wait, no lane-dependent branch near the loads (rows past the end and columns past the row width
            // come from clamped addresses and are masked when the tile is staged)
            auto request = [&](f32x4 (&xr)[PCH][PASSES], int (&xc)[PASSES], int j) {
                const int tile = (int)blockIdx.x + j * stride;
#pragma unroll
                for (int pc = 0; pc < PCH; ++pc) {
                    const bool is_agg = PCH == 2 ? pc == 1 : SRC == 0;                    // compile-time after unrolling
#pragma unroll
                    for (int p = 0; p < PASSES; ++p) {
                        const int g = min(tile * M + pw * RPW + p * RPP + sg, nn - 1);
                        if (is_agg) {
                            xr[pc][p] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.x + (int64_t)g * a.ldx + coff));
                            xc[p] = cnt_or_any[g];
                        } else {
                            xr[pc][p] = *reinterpret_cast<const f32x4*>(a.self_tab + (int64_t)min(max(sidx[p], 0), a.self_rows - 1) * a.ld_self + coff);
                        }
                    }
                }
            };
            // VGPRs -> mask -> split -> three bf16 LDS planes; returns the wave's rows that hold a huge value (bit = row - 8 pw)
            auto stage = [&](f32x4 (&xr)[PCH][PASSES], int (&xc)[PASSES], int j) -> unsigned {
                const int tile = (int)blockIdx.x + j * stride;
                __bf16* buf = planes + (j & 1) * 3 * PL;
                unsigned rows_bad = 0;
#pragma unroll
                for (int pc = 0; pc < PCH; ++pc) {
                    const bool is_agg = PCH == 2 ? pc == 1 : SRC == 0;
#pragma unroll
                    for (int p = 0; p < PASSES; ++p) {
                        const bool valid = col_ok && tile * M + pw * RPW + p * RPP + sg < nn;
                        const bool nanrow = nan_rule && is_agg && xc[p] == 0;          // aggregators.py:60-61
                        const float q = __builtin_nanf("");
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = valid ? (nanrow ? q : xr[pc][p][e]) : 0.f;
                        const unsigned long long hb = __ballot(huge4(v));
#pragma unroll
                        for (int s = 0; s < RPP; ++s) {
                            const unsigned long long grp = LG == 64 ? ~0ull : (((1ull << (LG & 63)) - 1ull) << (s * (LG & 63)));
                            if (hb & grp) rows_bad |= 1u << (p * RPP + s);
                        }
                        bf16x4 hi, mid, lo;
#ifdef PC_NO_SPLIT
                        hi = __builtin_convertvector(v, bf16x4); mid = hi; lo = hi;
#else
                        split3(v, hi, mid, lo);
#endif
                        __bf16* dst = buf + (pw * RPW + p * RPP + sg) * LDB + pc * KP + c0;
                        *reinterpret_cast<bf16x4*>(dst) = hi;
                        *reinterpret_cast<bf16x4*>(dst + PL) = mid;
                        *reinterpret_cast<bf16x4*>(dst + 2 * PL) = lo;
                    }
                }
                if (rows_bad != 0 && lane == 0) {                         // the tile goes on the exact-redo list (duplicates are harmless)
                    const int i = atomicAdd(&flags[1], 1);
                    if (i < kListCap) flags[2 + i] = tile;
                }
                return rows_bad;
            };
            auto request_prev = [&](int j) {                             // epi 2: the partial sums the earlier chunk's launch left in `out`
                if constexpr (EPI == 2) {
                    const int tile = (int)blockIdx.x + j * stride;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int g = min(tile * M + pw * RPW + 2 * it + (lane >> 5), nn - 1);
                        const int col = min((lane & 31) * 4, max(a.out_dim - 4, 0));
                        prev[it] = *reinterpret_cast<const f32x4*>(a.out + (int64_t)g * a.ldo + col);
                    }
                }
            };
            auto finish = [&](int j, unsigned rows_bad) {                // tile j: result plane -> (+ prev) -> activation -> out
                const int tile = (int)blockIdx.x + j * stride;
                float* pl = part + (size_t)(j & 1) * M * PLD;
                const bool whuge = flags[0] != 0;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rl = 2 * it + (lane >> 5), row = pw * RPW + rl;
                    const int col = (lane & 31) * 4;
                    const int g = tile * M + row;
                    f32x4 v = *reinterpret_cast<const f32x4*>(pl + row * PLD + col);
                    if (g < nn && col < a.out_dim) {
                        if constexpr (EPI == 2) v += prev[it];
                        if constexpr (EPI != 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = sage_activate(v[e], a.act);
                        }
                        if (whuge || ((rows_bad >> rl) & 1u)) continue;      // redone exactly below; other rows keep the MFMA result
                        float* dst = a.out + (int64_t)g * a.ldo + col;
                        if (col + 3 < a.out_dim && vec_out) {
                            sage_store_stream<SAGE_H1_STORE>(reinterpret_cast<f32x4*>(dst), v);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col + e < a.out_dim) __builtin_nontemporal_store(v[e], dst + e);
                        }
                    }
                }
            };

            // phase 0: tile 0's rows are requested, the sum planes zeroed, tile 0 staged, tile 1 requested
            unsigned bad0 = 0, bad1 = 0, bad2 = 0;                       // rows_bad of the tiles staged in this / the previous two phases
            request_index(0);
            request(x0, c0n, 0);
            request_index(1);
            bad0 = stage(x0, c0n, 0);
            STAMP_T(256, 21);
            if (nt > 1) { request(x0, c0n, 1); request_index(2); }
            lds_barrier();
            STAMP_T(256, 22);
            // phases 1 .. nt: stage tile p (requested one phase ago: its rows travelled while this wave finished tile p - 3 and waited
            // for the consumers), request tile p + 1, finish tile p - 2
            for (int p = 1; p <= nt; ++p) {
                bad2 = bad1; bad1 = bad0; bad0 = 0;
#ifdef SAGE_DENSE_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP_T(256, 22 + 4 * (p - 1) + 1);        // loads of tile p have landed
#endif
                if (p < nt) bad0 = stage(x0, c0n, p);
                STAMP_T(256, 22 + 4 * (p - 1) + 2);
                if (p + 1 < nt) { request(x0, c0n, p + 1); request_index(p + 2); }
                if (p >= 2) { request_prev(p - 2); finish(p - 2, bad2); }
                STAMP_T(256, 22 + 4 * (p - 1) + 3);
                lds_barrier();
                STAMP_T(256, 22 + 4 * (p - 1) + 4);
            }
            // phase nt + 1: the last tile (staged in phase nt - 1: its rows_bad is bad1 now -- phase nt staged nothing)
            request_prev(nt - 1);
            finish(nt - 1, bad1);
            STAMP_T(256, 38);
        }
        // Tiles that held |x| >= 2^127 / Inf / NaN (or all tiles, when W does): the exact fp32 fma chain.  Block-uniform; zero
        // iterations on ordinary data.
        __syncthreads();
        STAMP_T(0, 39);
        const int nbad = flags[1];
        if (nbad > 0 || flags[0] != 0) {
            const bool all = nbad > kListCap || flags[0] != 0;
            const int per = M * a.out_dim;
            for (int tb = (int)blockIdx.x, li = 0; all ? (tb < ntiles) : (li < nbad); tb += stride, ++li) {
                const int tile = all ? tb : flags[2 + li];
                for (int idx = tid; idx < per; idx += (int)blockDim.x) {
                    const int g = tile * M + idx / a.out_dim, col = idx % a.out_dim;
                    if (g < nn && (flags[0] != 0 || pc_row_is_huge(a, use_self, use_agg, g, nan_rule))) {
                        const float v = pc_exact_dot(a, md, use_self, use_agg, g, col, nan_rule);
                        a.out[(int64_t)g * a.ldo + col] = EPI == 1 ? v : sage_activate(v, a.act);
                    }
                }
            }
        }
    }
    sage_finish_block(a.fin, (int)gridDim.x);
}

template <int KP, int PCH, int SRC = 0, int EPI = 0>
int launch_pc(const DenseArgs& a, const PcMode& md, hipStream_t st) {
    constexpr int KPASS = PCH * KP;
    constexpr size_t lds = (size_t)2 * 3 * 32 * (KPASS + 8) * 2 + (size_t)2 * 32 * (128 + 8) * sizeof(float) + 256;
    static bool configured = false;
    if (!configured) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)dense_pc_kernel<KP, PCH, SRC, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            sage_set_error("layer_dense: cannot reserve %zu bytes of LDS", lds);
            return SAGE_ELAUNCH;
        }
        configured = true;
    }
    const int grid = min(sage_cdiv(a.n, 32), sage_tunables().dense_blocks);
    hipLaunchKernelGGL((dense_pc_kernel<KP, PCH, SRC, EPI>), dim3(grid), dim3(512), lds, st, a, md);
    SAGE_CHECK_LAUNCH("dense_pc_kernel");
    return SAGE_OK;
}

// W [out_dim, CHUNKS * dim] fp32 -> three bf16 planes in the register order of dense_bf16x3_kernel<KP, CONCAT, false, true>:
// prepared[(((pass * 8 + wave) * STEPS + st) * 3 + plane) * 64 + lane] = 8 bf16 = plane(W[32 (wave & 3) + (lane & 31)][column(kk) .. + 7]),
// kk = (wave >> 2) * KPASS/2 + 16 st + 8 (lane >> 5) inside the pass; column(kk) = chunk * dim + kk % KP in the [self | agg]
// layout of encoders.py:54 (chunk = pass for the two-pass 512-deep layer, kk / KP when both chunks share one pass, 0 without
// concat); zeros outside [out_dim, dim].  One thread per (pass, wave, st, lane); the trailer's .x = "W holds a huge value".
// MP (KP = 256): rows wider than 256 floats -- ceil(dim / 256) passes per chunk, column(kk) = chunk * dim + (pass % ppc) * 256 + kk.
template <int KP, bool CONCAT, bool MP = false>
__global__ void prepare_weights_kernel(const float* __restrict__ W, int64_t ldw, int dim, int out_dim, uint4* __restrict__ prepared) {
    constexpr int PCH = (CONCAT && KP < 256) ? 2 : 1, KPASS = PCH * KP;
    constexpr int KH = KPASS / 2, STEPS = KH / 16;
    const int ppc = MP ? (dim + KP - 1) / KP : 1;
    const int npass = ((CONCAT ? 2 : 1) / PCH) * ppc;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npass * 8 * STEPS * 64) return;
    uint4* trailer = prepared + (size_t)npass * 8 * STEPS * 3 * 64;   // zeroed by the launcher
    const int lane = idx & 63, st = (idx >> 6) % STEPS, wave = (idx / (64 * STEPS)) % 8, pass = idx / (64 * STEPS * 8);
    const int row = 32 * (wave & 3) + (lane & 31);
    const int kk = (wave >> 2) * KH + 16 * st + 8 * (lane >> 5);
    const int chunk = MP ? pass / ppc : pass * PCH + kk / KP, kc = MP ? (pass % ppc) * KP + kk : kk % KP;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (row < out_dim) {
        const float* wr = W + (int64_t)row * ldw + (int64_t)chunk * dim;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (kc + e < dim) v0[e] = wr[kc + e];
            if (kc + 4 + e < dim) v1[e] = wr[kc + 4 + e];
        }
    }
    if (huge4(v0) || huge4(v1)) atomicOr(&trailer->x, 1u);
    bf16x4 h0, m0, l0, h1, m1, l1;
    split3(v0, h0, m0, l0);
    split3(v1, h1, m1, l1);
    const bf16x8 ph = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    const bf16x8 pm = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
    const bf16x8 pl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    uint4* dst = prepared + ((size_t)((pass * 8 + wave) * STEPS + st) * 3) * 64 + lane;
    dst[0] = __builtin_bit_cast(uint4, ph);
    dst[64] = __builtin_bit_cast(uint4, pm);
    dst[128] = __builtin_bit_cast(uint4, pl);
}

// planes of the whole K range: 8 waves x (K / 32) steps x 3 planes x 64 lanes x 16 B, K = KP or (concat) 2 KP
static size_t prepared_plane_bytes(int kp, bool concat, int dim) {
    const int k_total = (concat ? 2 : 1) * (dim > 256 ? sage_cdiv(dim, 256) * 256 : kp);
    return (size_t)8 * (k_total / 32) * 3 * 64 * 16;
}

int prepared_kp(int32_t dim, int32_t out_dim) {
    if (!sage_layer_dense_supported(dim, out_dim)) return 0;
    return dim <= 64 ? 64 : dim <= 128 ? 128 : 256;
}

}  // namespace

// Prepared weights exist for every shape the contraction kernel takes; 0 = not a shape of that kernel.
extern "C" size_t sage_prepared_weight_bytes(int32_t dim, int32_t out_dim, int32_t concat) {
    const int kp = prepared_kp(dim, out_dim);
    return kp ? prepared_plane_bytes(kp, concat != 0, dim) + 16 : 0;      // the planes + a 16-byte trailer (huge-value mark)
}

extern "C" int sage_prepare_weights(const float* weight, int64_t ldw, int32_t dim, int32_t out_dim, int32_t concat, void* prepared,
                                    size_t prepared_bytes, sage_stream_t stream) {
    SAGE_REQUIRE(weight && prepared, "prepare_weights: NULL argument");
    const size_t need = sage_prepared_weight_bytes(dim, out_dim, concat);
    if (need == 0) {
        sage_set_error("prepare_weights: no prepared form for dim=%d out_dim=%d concat=%d", dim, out_dim, concat);
        return SAGE_EUNSUPPORTED;
    }
    SAGE_REQUIRE(prepared_bytes >= need, "prepare_weights: buffer %zu bytes < %zu", prepared_bytes, need);
    SAGE_REQUIRE(ldw >= (concat ? 2 : 1) * (int64_t)dim && sage_aligned(prepared, 16), "prepare_weights: ldw = %lld, buffer alignment", (long long)ldw);
    const int kp = prepared_kp(dim, out_dim);
    const int threads = (int)((need - 16) / (3 * 16)), blocks = sage_cdiv(threads, 256);      // one thread per (pass, wave, step, lane)
    hipStream_t st = (hipStream_t)stream;
    if (int rc = sage_fill_u32((char*)prepared + need - 16, 0u, 4, st)) return rc;          // the "W holds a huge value" trailer (a kernel, not a memset: sage_api.hip)
    uint4* out = (uint4*)prepared;
    if (dim > 256) {
        if (concat) hipLaunchKernelGGL((prepare_weights_kernel<256, true, true>), dim3(blocks), dim3(256), 0, st, weight, ldw, dim, out_dim, out);
        else hipLaunchKernelGGL((prepare_weights_kernel<256, false, true>), dim3(blocks), dim3(256), 0, st, weight, ldw, dim, out_dim, out);
    } else if (concat) {
        if (kp == 64) hipLaunchKernelGGL((prepare_weights_kernel<64, true>), dim3(blocks), dim3(256), 0, st, weight, ldw, dim, out_dim, out);
        else if (kp == 128) hipLaunchKernelGGL((prepare_weights_kernel<128, true>), dim3(blocks), dim3(256), 0, st, weight, ldw, dim, out_dim, out);
        else hipLaunchKernelGGL((prepare_weights_kernel<256, true>), dim3(blocks), dim3(256), 0, st, weight, ldw, dim, out_dim, out);
    } else {
        if (kp == 64) hipLaunchKernelGGL((prepare_weights_kernel<64, false>), dim3(blocks), dim3(256), 0, st, weight, ldw, dim, out_dim, out);
        else if (kp == 128) hipLaunchKernelGGL((prepare_weights_kernel<128, false>), dim3(blocks), dim3(256), 0, st, weight, ldw, dim, out_dim, out);
        else hipLaunchKernelGGL((prepare_weights_kernel<256, false>), dim3(blocks), dim3(256), 0, st, weight, ldw, dim, out_dim, out);
    }
    SAGE_CHECK_LAUNCH("prepare_weights_kernel");
    return SAGE_OK;
}

bool sage_layer_dense_supported(int32_t dim, int32_t out_dim) {
    return dim >= 4 && dim % 4 == 0 && out_dim >= 1 && out_dim <= 128;
}

// The 512-deep concat layer as two launches of the producer / consumer kernel (self chunk, then the means' chunk): possible when
// the prepared planes exist and the output rows can be re-read 16 bytes at a time.
bool sage_layer_dense_two_launches(int32_t dim, int32_t out_dim, int32_t concat, const void* weight_prepared, const float* out, int64_t ldo) {
    return (sage_dense_pc_enabled() != 0 || sage_dense_two_enabled() != 0) && concat && weight_prepared && dim > 128 && dim <= 256 &&
           sage_layer_dense_supported(dim, out_dim) && ldo % 4 == 0 && sage_aligned(out, 16) && out_dim % 4 == 0;
}

// parts: SAGE_DENSE_PART_SELF (the self chunk's partial sums -> out), SAGE_DENSE_PART_AGG (out = act(out + means' chunk)), or both
// (= the whole layer; the only value accepted when sage_layer_dense_two_launches() is false)
int sage_launch_layer_dense(const float* x, int64_t ldx, int32_t dim, int32_t n, const int32_t* n_dev, int32_t concat,
                            const float* self_tab, int64_t ld_self, int64_t self_rows, const int32_t* self_index,
                            const int32_t* cnt, const int32_t* any_nonempty,
                            const float* weight, int64_t ldw, int32_t out_dim, int32_t act, float* out, int64_t ldo, int32_t n_off,
                            sage_finish_t fin, const void* weight_prepared, hipStream_t st, int parts) {
    if (!sage_layer_dense_supported(dim, out_dim) || ldx % 4 != 0 || ldw % 4 != 0 || !sage_aligned(x, 16) ||
        !sage_aligned(weight, 16) || (concat && (ld_self % 4 != 0 || !sage_aligned(self_tab, 16)))) {
        sage_set_error("layer_dense: unsupported shape dim=%d out_dim=%d", dim, out_dim);
        return SAGE_EUNSUPPORTED;
    }
    if (n == 0) return SAGE_OK;
    const DenseArgs a{x, ldx, dim, n, n_dev, n_off, concat ? self_tab : x, concat ? ld_self : ldx, concat ? (int)self_rows : n,
                      self_index, cnt, any_nonempty, weight, ldw, out_dim, act, out, ldo, fin,
                      (const uint4*)weight_prepared};
    const int kp = dim <= 64 ? 64 : dim <= 128 ? 128 : 256;
    const bool two = sage_layer_dense_two_launches(dim, out_dim, concat, weight_prepared, out, ldo);
    if (parts != SAGE_DENSE_PART_ALL && !two) {
        sage_set_error("layer_dense: this shape is one launch (parts = %d)", parts);
        return SAGE_EINVAL;
    }
    if (dim > 256) return concat ? launch_bf16x3<256, true, true>(a, st) : launch_bf16x3<256, false, true>(a, st);
#ifndef SAGE_DENSE_FP32
    if (weight_prepared && sage_dense_pc_enabled() != 0) {
        // producer / consumer waves (dense_pc_kernel).  mode: {src, epi, wpass, wpasses, agg_woff}
        if (!concat) {
            const PcMode md{0, 0, 0, 1, 0};
            return kp == 64 ? launch_pc<64, 1>(a, md, st) : kp == 128 ? launch_pc<128, 1>(a, md, st) : launch_pc<256, 1>(a, md, st);
        }
        if (kp < 256) {
            const PcMode md{0, 0, 0, 1, dim};
            return kp == 64 ? launch_pc<64, 2>(a, md, st) : launch_pc<128, 2>(a, md, st);
        }
        if (two) {
            if (parts & SAGE_DENSE_PART_SELF) {
                DenseArgs as = a;
                as.fin = sage_finish_t{nullptr, nullptr};
                if (!(parts & SAGE_DENSE_PART_AGG)) as.fin = fin;
                const PcMode md{1, 1, 0, 2, dim};
                if (int rc = launch_pc<256, 1, 1, 1>(as, md, st)) return rc;
            }
            if (parts & SAGE_DENSE_PART_AGG) {
                const PcMode md{0, 2, 1, 2, dim};
                if (int rc = launch_pc<256, 1, 0, 2>(a, md, st)) return rc;
            }
            return SAGE_OK;
        }
    }
    if (two) {
        // the lock-step kernel, one K chunk per launch (round 4): the nodes' own rows -> partial sums in `out`, then out = act(out + means' chunk)
        if (parts & SAGE_DENSE_PART_SELF) {
            DenseArgs as = a;
            as.fin = (parts & SAGE_DENSE_PART_AGG) ? sage_finish_t{nullptr, nullptr} : fin;
            as.cnt = nullptr;                          // the 0/0 rule belongs to the means' chunk
            as.wpass = 0; as.wpasses = 2; as.woff = 0;
            if (int rc = launch_bf16x3<256, false, false, true, 1, 1>(as, st)) return rc;
        }
        if (parts & SAGE_DENSE_PART_AGG) {
            DenseArgs ag = a;
            ag.wpass = 1; ag.wpasses = 2; ag.woff = dim;
            if (int rc = launch_bf16x3<256, false, false, true, 0, 2>(ag, st)) return rc;
        }
        return SAGE_OK;
    }
    if (!concat) {
        if (kp == 64) return launch_bf16x3<64, false>(a, st);
        if (kp == 128) return launch_bf16x3<128, false>(a, st);
        return launch_bf16x3<256, false>(a, st);
    }
    if (kp == 64) return launch_bf16x3<64, true>(a, st);
    if (kp == 128) return launch_bf16x3<128, true>(a, st);
    return launch_bf16x3<256, true>(a, st);
#else
    if (!concat) {
        if (kp == 64) return launch<64, false>(a, st);
        if (kp == 128) return launch<128, false>(a, st);
        return launch<256, false>(a, st);
    }
    if (kp == 64) return launch<64, true>(a, st);
    if (kp == 128) return launch<128, true>(a, st);
    return launch<256, true>(a, st);
#endif
}
