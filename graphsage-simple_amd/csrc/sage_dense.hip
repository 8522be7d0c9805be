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
#include <atomic>
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
    static std::atomic<bool> configured{false};           // role threads (and the express lane's thread) may launch the same kernel concurrently
    if (!configured.load(std::memory_order_acquire)) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)dense_layer_kernel<KP, CONCAT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)lds) != hipSuccess) {
            sage_set_error("layer_dense: cannot reserve %zu bytes of LDS", lds);
            return SAGE_ELAUNCH;
        }
        configured.store(true, std::memory_order_release);
    }
    // one persistent block per CU (24.9 us) beat two (26.3 us) on the config-3 contraction: the prefetch only pays
    // when a block owns >= 2 tiles, and one block per CU leaves wave slots for another batch's kernels
#ifndef SAGE_DENSE_PER_CU
#define SAGE_DENSE_PER_CU 1
#endif
    const int grid = min(sage_cdiv(a.n, 32), SAGE_DENSE_PER_CU * kNumCU);
    SAGE_LAUNCH_TAIL((dense_layer_kernel<KP, CONCAT>), dim3(grid), dim3(KSPLIT ? 512 : 256), lds, st, a);
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
// Stamps are 100 MHz real-time ticks (s_memrealtime: one clock for the whole device, so block starts on different XCDs compare).  Only the
// launch selected with sage_debug_dense_select() is stamped (launch = blocks started so far / grid: launches of a stream never overlap),
// so that a launch in the MIDDLE of a running pipeline can be looked at, not only the last one.
__device__ unsigned long long g_dense_stamps[512 * 40];
__device__ unsigned int g_dense_blocks;
__device__ int g_dense_target = -1;                      // -1: every launch (the last one wins)
extern "C" int sage_debug_dense_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dense_stamps), sizeof(g_dense_stamps)) == hipSuccess ? 0 : -1;
}
extern "C" int sage_debug_dense_select(int launch) {
    const unsigned int zero = 0;
    static unsigned long long zeros[512 * 40];
    return (hipMemcpyToSymbol(HIP_SYMBOL(g_dense_target), &launch, sizeof(int)) == hipSuccess &&
            hipMemcpyToSymbol(HIP_SYMBOL(g_dense_blocks), &zero, sizeof(zero)) == hipSuccess &&
            hipMemcpyToSymbol(HIP_SYMBOL(g_dense_stamps), zeros, sizeof(zeros)) == hipSuccess) ? 0 : -1;
}
#define STAMP_DECL __shared__ int stamp_on_; if (threadIdx.x == 0) { const unsigned int o_ = atomicAdd(&g_dense_blocks, 1u); stamp_on_ = (g_dense_target < 0 || (int)(o_ / gridDim.x) == g_dense_target) ? 1 : 0; }
#define STAMP(i) do { if (threadIdx.x == 0 && (i) < 40 && stamp_on_) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_dense_stamps[blockIdx.x * 40 + (i)] = t_; } } while (0)
#else
#define STAMP_DECL
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

// MP (KP = 256 only): rows wider than 256 -- every K chunk takes ceil(dim / 256) passes (Pubmed 500, Cora 1433+3 pad).
// PREP: W arrives as the planes of sage_prepare_weights (a.wsplit; one-pass shapes only) -- a compile-time property, so that no
// join of two W paths stands between the W loads and their first use.
template <int KP, bool CONCAT, bool MP, bool PREP = false>
__global__ __launch_bounds__(512) void dense_bf16x3_kernel(const DenseArgs a) {
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
    STAMP_DECL
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
        constexpr bool SELF_AHEAD = CONCAT && !MP && KP == 256;
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
                const bool is_agg = ((pass / ppc) * PCH + pc) == CHUNKS - 1;      // the last K chunk is the neighbour mean (block-uniform)
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
        int stage_seq = 0;                                    // stagings so far (block-uniform): the tag a "huge value" mark carries,
                                                              // so that marks never have to be cleared (a clear would race the next staging)
        auto stage_tile = [&](__bf16* buf, int bsel, int tile, int pass) {   // VGPRs -> mask -> split -> three bf16 LDS planes
            ++stage_seq;
#pragma unroll
            for (int pc = 0; pc < PCH; ++pc) {
                const bool is_agg = ((pass / ppc) * PCH + pc) == CHUNKS - 1;
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
            if (tid == 0 && a.wsplit[(size_t)npass * WAVES * STEPS * 3 * 64].x != 0) flags[2] = 1;   // read after the first staging's barrier
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
                    const uint4* wp = a.wsplit + (size_t)(pass * WAVES + __builtin_amdgcn_readfirstlane(wave)) * STEPS * 3 * 64;
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
                        for (int e = 0; e < 4; ++e) v[e] = sage_activate(p0[e] + p1[e], a.act);
                        if (bad[t] & (8 | (2 << it))) continue;    // a row with a huge value (or huge W): redone exactly below; other rows of
                                                                   // the tile keep the MFMA result, so a row never depends on its tile mates
                        float* dst = a.out + (int64_t)g * a.ldo + col;
                        if (col + 3 < a.out_dim && vec_store) {
                            sage_store_stream<SAGE_H1_STORE>(reinterpret_cast<f32x4*>(dst), v);
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
                    if (g < nn && (flags[2] != 0 || row_is_huge(a, CONCAT, g, nan_rule)))
                        a.out[(int64_t)g * a.ldo + col] = sage_activate(exact_row_dot(a, CONCAT, g, col, nan_rule), a.act);
                }
            }
        }
    }
    sage_finish_block(a.fin, (int)gridDim.x);
}

template <int KP, bool CONCAT, bool MP = false, bool PREP = false>
int launch_bf16x3(const DenseArgs& a, hipStream_t st) {
    if constexpr (!PREP)
        if (a.wsplit) return launch_bf16x3<KP, CONCAT, MP, true>(a, st);
    constexpr int KPASS = (CONCAT && KP < 256) ? 2 * KP : KP;
    constexpr size_t lds = (size_t)2 * 3 * 32 * (KPASS + 8) * 2 + (size_t)2 * 32 * (128 + 4) * sizeof(float) + 384;
    static std::atomic<bool> configured{false};           // role threads (and the express lane's thread) may launch the same kernel concurrently
    if (!configured.load(std::memory_order_acquire)) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)dense_bf16x3_kernel<KP, CONCAT, MP, PREP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)lds) != hipSuccess) {
            sage_set_error("layer_dense: cannot reserve %zu bytes of LDS", lds);
            return SAGE_ELAUNCH;
        }
        configured.store(true, std::memory_order_release);
    }
    // Persistent blocks on 3/4 of the CUs: alone that costs 1-2 us (4 tile rounds instead of 3), with a second batch in
    // flight it is worth 3 us per forward -- a block holds 344 of a SIMD's 512 VGPRs and 117 KB of LDS for its whole life,
    // and the other batch's latency-bound kernels get the remaining CUs to themselves (same-box A/B: 184-224 blocks
    // 81.3-81.9 us, 256 blocks 84.2, 160 blocks 83.1).
    const int grid = min(sage_cdiv(a.n, 32), sage_tunables().dense_blocks);
    SAGE_LAUNCH_TAIL((dense_bf16x3_kernel<KP, CONCAT, MP, PREP>), dim3(grid), dim3(512), lds, st, a);
    SAGE_CHECK_LAUNCH("dense_bf16x3_kernel");
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

int sage_launch_layer_dense(const float* x, int64_t ldx, int32_t dim, int32_t n, const int32_t* n_dev, int32_t concat,
                            const float* self_tab, int64_t ld_self, int64_t self_rows, const int32_t* self_index,
                            const int32_t* cnt, const int32_t* any_nonempty,
                            const float* weight, int64_t ldw, int32_t out_dim, int32_t act, float* out, int64_t ldo, int32_t n_off,
                            sage_finish_t fin, const void* weight_prepared, hipStream_t st) {
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
    if (dim > 256) return concat ? launch_bf16x3<256, true, true>(a, st) : launch_bf16x3<256, false, true>(a, st);
#ifndef SAGE_DENSE_FP32
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
