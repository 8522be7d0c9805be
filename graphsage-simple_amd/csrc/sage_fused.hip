// One-launch GraphSAGE layer: sample lists -> gather-mean tile in LDS -> fp32 MFMA -> act.
//
// Replaces one whole Encoder.forward after sampling (aggregators.py:52-74 + encoders.py:49-62)
// without the [n, dim] aggregate ever leaving the CU.  Structure (256 threads = 4 waves):
//
//   phase A  each wave gathers M/4 destination rows: lane l owns columns 4l..4l+3 (one
//            global_load_dwordx4 = one 1 KiB row per wave-instruction at dim = 256), neighbour
//            ids broadcast by v_readlane, 8 rows in flight per wave; the mean (and, for the
//            concat encoder, the node's own row) is written to an fp32 LDS tile
//            A[chunk][M][KP+4]   (+4 floats = one ds_read_b128 width: conflict-free operand reads)
//   phase B  wave w owns output columns [32w, 32w+32).  Its slice of W ([32, KP] fp32) lives in
//            VGPRs for the whole kernel (persistent blocks, loaded once when the layer has a
//            single K chunk), so the MFMA loop reads only A from LDS: per 8-deep k-step one
//            ds_read_b128 per 32-row block feeds 4 v_mfma_f32_32x32x2_f32.
//            Lane (i = l&31, h = l>>5) supplies A[i][8q+4h+t] and W[n0+i][8q+4h+t] to MFMA 4q+t:
//            any k-pairing is valid as long as A and B agree, and this one makes both operands
//            16-byte contiguous per lane.
//   epilogue act() and store out[row, n0 + (l&31)] (128-B segments per accumulator register).
//
// HBM-bound: the gather moves ~14 KiB per destination row against 65 kFLOP of MFMA work, so the
// matrix pipe is idle most of the time; two blocks per CU (LDS 66.5 KiB each at M=64, KP=256;
// <= 256 VGPRs) let one block's MFMA phase hide under the other's gather.
#include <atomic>
#include "sage_internal.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct FusedArgs {
    const float* table; int table_rows; int64_t ld; int dim;
    const int32_t* nbr; const int32_t* cnt; int k; int n; const int32_t* n_dev;
    const int32_t* slot_rows; const int32_t* self_row; const int32_t* any_nonempty;
    const int32_t* self_index;
    const float* self_tab; int64_t ld_self; int self_rows;     // concat encoder: where the node's own row comes from
    const float* W; int64_t ldw; int out_dim; int act;
    float* out; int64_t ldo;
    int n_off; sage_finish_t fin;
    int32_t* wipe_keys; int32_t* rows_out; int32_t* self_rows_out;     // slot -> row resolve duties (sage_slot_resolve_t), nullable
};

// KP: padded K per chunk; M: rows per tile; WAVES: waves per block (all gather; waves (w&3, w>>2)
// own output columns [32(w&3), +32) of 32-row block(s) w>>2); WREG: keep the wave's W slice in
// VGPRs for the whole kernel (big layers) or stream it from L2 inside the MFMA loop (small
// layers, where 16 waves per block buy gather parallelism and cap the VGPR budget at 128).
template <int KP, int M, int WAVES, bool WREG, bool CONCAT, int INFLIGHT = 8>
__global__ __launch_bounds__(WAVES * 64, (WAVES >= 16 || !WREG) ? 4 : 2) void layer_fused_kernel(const FusedArgs a) {
    constexpr int LDA = KP + 4;                 // floats per LDS row
    constexpr int CHUNKS = CONCAT ? 2 : 1;
    constexpr int MB = M / 32;                  // 32-row MFMA blocks per tile
    constexpr int MGROUPS = WAVES / 4;          // wave groups along M
    constexpr int MBW = (MB + MGROUPS - 1) / MGROUPS;   // 32-row blocks per wave
    constexpr int RPW = M / WAVES;              // rows gathered per wave
    constexpr int LG = KP / 4;                  // lanes that cover one row (16-B loads)
    constexpr int RPP = 64 / LG;                // rows a wave gathers at once
    constexpr int PASSES = (RPW + RPP - 1) / RPP;
    static_assert(M % WAVES == 0 && M % 32 == 0 && WAVES % 4 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [CHUNKS][M][LDA]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nn = a.n;
    if (a.n_dev) nn = min(*a.n_dev + a.n_off, a.n);
    const int ntiles = (nn + M - 1) / M;
    if ((int)blockIdx.x >= ntiles) {
        sage_finish_block(a.fin, (int)gridDim.x);
        return;
    }
    const bool nan_rule = a.any_nonempty ? (*a.any_nonempty != 0) : false;
    const int last_row = a.table_rows - 1;
    const int i32 = lane & 31, h = lane >> 5;
    const int n0 = (wave & 3) * 32;
    const int mb0 = (wave >> 2) * MBW;
    const bool mfma_wave = n0 < a.out_dim && mb0 < MB;
    const int lg = lane & (LG - 1), sg = lane / LG;   // lane inside its row group, row group inside the wave
    const int c0 = lg * 4;                      // this lane's columns in phase A
    const bool col_ok = c0 < a.dim;             // dim % 4 == 0 (host-checked)
    const bool col_pad = c0 < KP;               // columns [dim, KP) are zero padding
    const bool wrow_ok = mfma_wave && (n0 + i32) < a.out_dim;
    const float* wrow = a.W + (int64_t)min(n0 + i32, a.out_dim - 1) * a.ldw;

    // ---- W slice -> registers (per K chunk): breg[4q+t] = W[n0+i][chunk*dim + 8q + 4h + t]
    constexpr bool WALL = WREG && CHUNKS == 2 && KP <= 128;   // concat, narrow: both K chunks of the slice stay in registers
    float breg[WREG ? (WALL ? KP : KP / 2) : 4];
    auto load_wq = [&](int chunk, int q) {
        const int kc = 8 * q + 4 * h;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (wrow_ok && kc < a.dim) v = *reinterpret_cast<const f32x4*>(wrow + (int64_t)chunk * a.dim + kc);
        return v;
    };
    auto load_w = [&](int chunk) {
        if constexpr (WREG) {
            const int o = WALL ? chunk * (KP / 2) : 0;
#pragma unroll
            for (int q = 0; q < KP / 8; ++q) {
                const f32x4 v = load_wq(chunk, q);
                breg[o + 4 * q + 0] = v[0]; breg[o + 4 * q + 1] = v[1]; breg[o + 4 * q + 2] = v[2]; breg[o + 4 * q + 3] = v[3];
            }
        }
    };
    if (CHUNKS == 1) load_w(0);
    if (WALL) { load_w(0); load_w(1); }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * M;
        // ------------------------------------------------------------ phase A: gather-mean -> LDS
        if constexpr (RPP == 1) {
            // 256-wide rows: the whole wave is on one row, ids are wave-uniform (v_readlane -> scalar address)
            for (int rr = 0; rr < RPW; ++rr) {
                const int r = wave + WAVES * rr;        // row inside the tile
                const int g = row0 + r;
                if (g >= nn) continue;                  // wave-uniform; rows past nn are never stored
                const int c = __builtin_amdgcn_readfirstlane(a.cnt[g]);
                int s = -1;
                if (a.self_row) {
                    s = a.self_row[g];
                    if (a.slot_rows && s >= 0) {
                        if (a.wipe_keys && lane == 0) a.wipe_keys[s] = -1;
                        s = a.slot_rows[s];
                    }
                    s = __builtin_amdgcn_readfirstlane(s);
                    if (a.self_rows_out && lane == 0) a.self_rows_out[g] = s;
                }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                bool extra = s >= 0;
                if (a.rows_out && c == 0 && lane < a.k) a.rows_out[(int64_t)g * a.k + lane] = -1;
                for (int base = 0; base < c; base += kWave) {
                    const int m = min(kWave, c - base);
                    int myid = (lane < m) ? a.nbr[(int64_t)g * a.k + base + lane] : 0;
                    if (a.slot_rows) {
                        if (a.wipe_keys && lane < m) a.wipe_keys[max(myid, 0)] = -1;
                        myid = a.slot_rows[max(myid, 0)];
                        if (a.rows_out && base + lane < a.k) a.rows_out[(int64_t)g * a.k + base + lane] = lane < m ? myid : -1;
                    }
                    if (extra && __any(lane < m && myid == s)) extra = false;      // aggregators.py:50-51: set union
                    myid = min(max(myid, 0), last_row);
                    for (int j0 = 0; j0 < m; j0 += INFLIGHT) {
                        f32x4 t[INFLIGHT];
#pragma unroll
                        for (int u = 0; u < INFLIGHT; ++u) {
                            const int id = __builtin_amdgcn_readlane(myid, min(j0 + u, m - 1));
                            if (col_ok) t[u] = *reinterpret_cast<const f32x4*>(a.table + (int64_t)id * a.ld + c0);
                            else t[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
#pragma unroll
                        for (int u = 0; u < INFLIGHT; ++u)
                            if (j0 + u < m) acc += t[u];
                    }
                }
                if (extra && col_ok) acc += *reinterpret_cast<const f32x4*>(a.table + (int64_t)min(s, last_row) * a.ld + c0);
                const int ceff = c + (extra ? 1 : 0);
                f32x4 mean;
                if (ceff > 0) mean = acc * (1.0f / (float)ceff);
                else { const float fill = nan_rule ? __builtin_nanf("") : 0.f; mean = f32x4{fill, fill, fill, fill}; }
                if (!col_ok) mean = f32x4{0.f, 0.f, 0.f, 0.f};
                if (col_pad) *reinterpret_cast<f32x4*>(lds + ((CHUNKS - 1) * M + r) * LDA + c0) = mean;
                if (CONCAT) {
                    f32x4 sv = {0.f, 0.f, 0.f, 0.f};
                    if (col_ok) {
                        const int64_t sr = a.self_index ? (int64_t)min(max(a.self_index[g], 0), a.self_rows - 1) : (int64_t)min(g, a.self_rows - 1);
                        sv = *reinterpret_cast<const f32x4*>(a.self_tab + sr * a.ld_self + c0);
                    }
                    if (col_pad) *reinterpret_cast<f32x4*>(lds + r * LDA + c0) = sv;
                }
            }
        } else {
            // narrow rows (KP <= 128): LG = KP/4 lanes cover one row, so a wave works on RPP = 64/LG rows at
            // once -- the dependent chain cnt -> ids -> rows is paid once per pass, not once per row.
            // Ids are exchanged inside the lane group with a width-LG shuffle.
            for (int p = 0; p < PASSES; ++p) {
                const int rsub = p * RPP + sg;               // this lane group's row among the wave's RPW rows
                const int r = wave * RPW + rsub;
                const int g = row0 + r;
                const bool valid = rsub < RPW && g < nn;
                const int gq = valid ? g : 0;
                const int c = valid ? a.cnt[gq] : 0;
                int s = -1;
                if (valid && a.self_row) {
                    s = a.self_row[gq];
                    if (a.slot_rows && s >= 0) {
                        if (a.wipe_keys && lg == 0) a.wipe_keys[s] = -1;
                        s = a.slot_rows[s];
                    }
                    if (a.self_rows_out && lg == 0) a.self_rows_out[gq] = s;
                }
                if (valid && a.rows_out && c == 0 && lg < a.k) a.rows_out[(int64_t)gq * a.k + lg] = -1;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                bool extra = s >= 0;
                for (int base = 0; __any(base < c); base += LG) {
                    const int m = max(0, min(LG, c - base));
                    int myid = (lg < m) ? a.nbr[(int64_t)gq * a.k + base + lg] : 0;
                    if (a.slot_rows) {
                        if (a.wipe_keys && lg < m) a.wipe_keys[max(myid, 0)] = -1;
                        myid = a.slot_rows[max(myid, 0)];
                        if (valid && a.rows_out && base + lg < a.k) a.rows_out[(int64_t)gq * a.k + base + lg] = lg < m ? myid : -1;
                    }
                    const unsigned long long hit = __ballot(extra && lg < m && myid == s);
                    if ((hit >> (lane - lg)) & ((1ull << LG) - 1ull)) extra = false;
                    myid = min(max(myid, 0), last_row);
                    for (int j0 = 0; __any(j0 < m); j0 += INFLIGHT) {
                        f32x4 t[INFLIGHT];
#pragma unroll
                        for (int u = 0; u < INFLIGHT; ++u) {
                            const int id = __shfl(myid, max(min(j0 + u, m - 1), 0), LG);
                            if (col_ok) t[u] = *reinterpret_cast<const f32x4*>(a.table + (int64_t)id * a.ld + c0);
                            else t[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
#pragma unroll
                        for (int u = 0; u < INFLIGHT; ++u)
                            if (j0 + u < m) acc += t[u];
                    }
                }
                if (extra && col_ok) acc += *reinterpret_cast<const f32x4*>(a.table + (int64_t)min(s, last_row) * a.ld + c0);
                const int ceff = c + (extra ? 1 : 0);
                f32x4 mean;
                if (ceff > 0) mean = acc * (1.0f / (float)ceff);
                else { const float fill = nan_rule ? __builtin_nanf("") : 0.f; mean = f32x4{fill, fill, fill, fill}; }
                if (!col_ok) mean = f32x4{0.f, 0.f, 0.f, 0.f};
                if (valid) *reinterpret_cast<f32x4*>(lds + ((CHUNKS - 1) * M + r) * LDA + c0) = mean;
                if (CONCAT) {
                    f32x4 sv = {0.f, 0.f, 0.f, 0.f};
                    if (valid && col_ok) {
                        const int64_t sr = a.self_index ? (int64_t)min(max(a.self_index[gq], 0), a.self_rows - 1) : (int64_t)min(gq, a.self_rows - 1);
                        sv = *reinterpret_cast<const f32x4*>(a.self_tab + sr * a.ld_self + c0);
                    }
                    if (valid) *reinterpret_cast<f32x4*>(lds + r * LDA + c0) = sv;
                }
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ phase B: [M, K] x W^T on the matrix pipe
        if (mfma_wave) {
            f32x16 acc[MBW];
#pragma unroll
            for (int b = 0; b < MBW; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
#pragma unroll
            for (int chunk = 0; chunk < CHUNKS; ++chunk) {
                if (CHUNKS > 1 && !WALL) load_w(chunk);
                const int bo = WALL ? chunk * (KP / 2) : 0;
                const float* abase = lds + (chunk * M + mb0 * 32 + i32) * LDA + 4 * h;
#pragma unroll
                for (int q = 0; q < KP / 8; ++q) {
                    f32x4 av[MBW];
#pragma unroll
                    for (int b = 0; b < MBW; ++b) av[b] = *reinterpret_cast<const f32x4*>(abase + b * 32 * LDA + 8 * q);
                    f32x4 bv;
                    if constexpr (WREG) bv = f32x4{breg[bo + 4 * q], breg[bo + 4 * q + 1], breg[bo + 4 * q + 2], breg[bo + 4 * q + 3]};
                    else bv = load_wq(chunk, q);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int b = 0; b < MBW; ++b)
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][t], bv[t], acc[b], 0, 0, 0);
                }
            }
            const int col = n0 + i32;
            if (col < a.out_dim) {
#pragma unroll
                for (int b = 0; b < MBW; ++b)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int g = row0 + (mb0 + b) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                        if (g < nn) a.out[(int64_t)g * a.ldo + col] = sage_activate(acc[b][reg], a.act);
                    }
            }
        }
        __syncthreads();     // the next tile's gather overwrites the LDS tile
    }
    sage_finish_block(a.fin, (int)gridDim.x);
}

// ---- small layers (layer 2 of the 2-hop forward: a few thousand destination rows) ---------------------------------
// A layer this small is bound by (a) what ONE CU can pull through its L1 -- the 32-row tiles above put 4096 rows on
// only 128 of the 256 CUs -- and (b) dependent round trips, each of which costs several microseconds while another
// batch's gather is saturating the fabric (16 us alone became 40 us with a second batch in flight).
// Here a tile is 16 rows (v_mfma_f32_16x16x4_f32), so 4096 rows fill all 256 CUs; a wave owns ONE row, its LG-lane
// groups fetch different neighbours (the sliced gather's trick) with the whole neighbour list in flight, and the
// groups meet in xor-shuffles: counts+ids -> rows -> LDS -> MFMA is two dependent trips instead of seven.
// Waves 0..7 own the 16-column output tiles; their W slices are requested before the gather and stay in VGPRs.
// The concat layer holds a 2 x KP-deep W slice (64 VGPRs at KP = 128) next to the gather's rows in flight: under the 128-VGPR cap of
// "4 waves per SIMD" it spilled (64 bytes of scratch per lane); with room for 172 it does not, and nothing shares a CU with the
// concat layers' kernels anyway -- concat forward at config 3 101.5 -> 89 us, config 5 90 -> 83 us (same-box A/B).
#ifndef SAGE_T16_CONCAT_MIN_WAVES
#define SAGE_T16_CONCAT_MIN_WAVES 2
#endif
template <int KP, bool CONCAT, int INFLIGHT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, CONCAT ? SAGE_T16_CONCAT_MIN_WAVES : 4) void layer_tile16_kernel(const FusedArgs a) {
    // WAVES = 16: one row per wave (fastest alone).  WAVES = 8: two rows per wave, 512-thread blocks -- half the wave
    // slots and VGPRs per block, so the block finds room on a CU that other batches' kernels already share (sage_pipe.hip:
    // a 1024-thread block needs 16 free wave slots and 384 VGPRs per SIMD on ONE CU and waited for whole gathers to drain).
    constexpr int M = 16;
    constexpr int RW = M / WAVES;                  // rows per wave
    constexpr int CHUNKS = CONCAT ? 2 : 1;
    constexpr int LDA = KP + 4;
    constexpr int LG = KP / 4, NPI = 64 / LG;      // lanes per row, neighbours per wave-instruction
    static_assert(WAVES == 16 || WAVES == 8, "tile16: 16 or 8 waves (at most two rows per wave)");
    __shared__ __attribute__((aligned(16))) float lds[CHUNKS * M * LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nn = a.n;
    if (a.n_dev) nn = min(*a.n_dev + a.n_off, a.n);
    const int ntiles = (nn + M - 1) / M;
    if ((int)blockIdx.x >= ntiles) {
        sage_finish_block(a.fin, (int)gridDim.x);
        return;
    }
    sage_finish_regs fin_regs;
    sage_finish_begin(a.fin, fin_regs);
    const bool nan_rule = a.any_nonempty ? (*a.any_nonempty != 0) : false;
    const int last_row = a.table_rows - 1;
    const int i16 = lane & 15, h = lane >> 4;
    const int n0 = wave * 16;
    const bool mfma_wave = n0 < a.out_dim;
    const bool wrow_ok = mfma_wave && (n0 + i16) < a.out_dim;
    const float* wrow = a.W + (int64_t)min(n0 + i16, a.out_dim - 1) * a.ldw;
    // breg[chunk][4q+t] = W[n0+i][chunk*dim + 16q + 4h + t]: k-step 4q+t of the MFMA loop takes lane group h's
    // k index from column 16q+4h+t (any pairing is valid as long as A and B agree; this one is 16-B contiguous per lane)
    float breg[CHUNKS * (KP / 4)];
#pragma unroll
    for (int chunk = 0; chunk < CHUNKS; ++chunk)
#pragma unroll
        for (int q = 0; q < KP / 16; ++q) {
            const int kc = 16 * q + 4 * h;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (wrow_ok && kc < a.dim) v = *reinterpret_cast<const f32x4*>(wrow + (int64_t)chunk * a.dim + kc);
#pragma unroll
            for (int t = 0; t < 4; ++t) breg[chunk * (KP / 4) + 4 * q + t] = v[t];
        }
    const int lg = lane & (LG - 1), grp = lane / LG;
    const int c0 = lg * 4;
    const bool col_ok = c0 < a.dim;               // dim % 4 == 0 (host-checked); columns [dim, KP) are zero padding

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * M;
        // the first 64 ids of every row of the wave are requested together with the counts, not after them
        int first_ids[RW], cs[RW], ss[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int g = row0 + wave + WAVES * rr;
            first_ids[rr] = 0; cs[rr] = 0; ss[rr] = -1;
            if (g < nn) {
                first_ids[rr] = (lane < a.k) ? a.nbr[(int64_t)g * a.k + lane] : 0;
                cs[rr] = a.cnt[g];
                if (a.self_row) ss[rr] = a.self_row[g];
            }
        }
        // rows of a wave one after the other (not unrolled: two rows' loads in flight at once cost 32 more VGPRs, and
        // this kernel's latency hides under another batch's gather; the register footprint does not)
#pragma unroll 1
        for (int rr = 0; rr < RW; ++rr) {
            const int rt = wave + WAVES * rr;          // row inside the tile
            const int g = row0 + rt;
            if (g >= nn) continue;                     // wave-uniform; rows past nn are never stored
            const int c = __builtin_amdgcn_readfirstlane(RW == 1 || rr == 0 ? cs[0] : cs[RW - 1]);
            const int ids0 = (RW == 1 || rr == 0) ? first_ids[0] : first_ids[RW - 1];
            int s = (RW == 1 || rr == 0) ? ss[0] : ss[RW - 1];
            if (a.slot_rows && s >= 0) {
                if (a.wipe_keys && lane == 0) a.wipe_keys[s] = -1;
                s = a.slot_rows[s];
            }
            s = __builtin_amdgcn_readfirstlane(s);
            if (a.self_rows_out && lane == 0) a.self_rows_out[g] = s;
            f32x4 sv = {0.f, 0.f, 0.f, 0.f};
            if (CONCAT && grp == NPI - 1 && col_ok) {
                const int64_t sr = a.self_index ? (int64_t)min(max(a.self_index[g], 0), a.self_rows - 1) : (int64_t)min(g, a.self_rows - 1);
                sv = *reinterpret_cast<const f32x4*>(a.self_tab + sr * a.ld_self + c0);
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            bool extra = s >= 0;
            if (a.rows_out && c == 0 && lane < a.k) a.rows_out[(int64_t)g * a.k + lane] = -1;
            for (int base = 0; base < c; base += kWave) {
                const int m = min(kWave, c - base);
                int myid = ids0;
                if (base > 0) myid = (lane < m) ? a.nbr[(int64_t)g * a.k + base + lane] : 0;
                if (lane >= m) myid = 0;
                if (a.slot_rows) {
                    if (a.wipe_keys && lane < m) a.wipe_keys[max(myid, 0)] = -1;          // this forward's key: gone for the next one
                    myid = a.slot_rows[max(myid, 0)];
                    if (a.rows_out && base + lane < a.k) a.rows_out[(int64_t)g * a.k + base + lane] = lane < m ? myid : -1;
                }
                if (extra && __any(lane < m && myid == s)) extra = false;          // aggregators.py:50-51: set union
                myid = min(max(myid, 0), last_row);
                for (int j0 = 0; j0 < m; j0 += NPI * INFLIGHT) {
                    f32x4 t[INFLIGHT];
#pragma unroll
                    for (int u = 0; u < INFLIGHT; ++u) {
                        const int j = j0 + u * NPI + grp;
                        const int id = __shfl(myid, min(j, m - 1), kWave);
                        if (col_ok && j < m) t[u] = *reinterpret_cast<const f32x4*>(a.table + (int64_t)id * a.ld + c0);
                        else t[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < INFLIGHT; ++u) acc += t[u];
                }
            }
            if (extra && col_ok && grp == 0) acc += *reinterpret_cast<const f32x4*>(a.table + (int64_t)min(s, last_row) * a.ld + c0);
#pragma unroll
            for (int x = LG; x < kWave; x <<= 1) {
                acc[0] += __shfl_xor(acc[0], x, kWave);
                acc[1] += __shfl_xor(acc[1], x, kWave);
                acc[2] += __shfl_xor(acc[2], x, kWave);
                acc[3] += __shfl_xor(acc[3], x, kWave);
            }
            const int ceff = c + (extra ? 1 : 0);
            f32x4 mean;
            if (ceff > 0) mean = acc * (1.0f / (float)ceff);
            else { const float fill = nan_rule ? __builtin_nanf("") : 0.f; mean = f32x4{fill, fill, fill, fill}; }
            if (!col_ok) mean = f32x4{0.f, 0.f, 0.f, 0.f};
            if (grp == 0) *reinterpret_cast<f32x4*>(lds + ((CHUNKS - 1) * M + rt) * LDA + c0) = mean;
            if (CONCAT && grp == NPI - 1) *reinterpret_cast<f32x4*>(lds + rt * LDA + c0) = sv;
        }
        __syncthreads();
        if (mfma_wave) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int chunk = 0; chunk < CHUNKS; ++chunk) {
                const float* abase = lds + (chunk * M + i16) * LDA + 4 * h;
#pragma unroll
                for (int q = 0; q < KP / 16; ++q) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(abase + 16 * q);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], breg[chunk * (KP / 4) + 4 * q + t], acc, 0, 0, 0);
                }
            }
            const int col = n0 + i16;
            if (col < a.out_dim) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gr = row0 + 4 * h + j;
                    if (gr < nn) a.out[(int64_t)gr * a.ldo + col] = sage_activate(acc[j], a.act);
                }
            }
        }
        __syncthreads();     // the next tile's gather overwrites the LDS tile
    }
    sage_finish_block(a.fin, (int)gridDim.x, fin_regs);
}

template <int KP, bool CONCAT>
int launch_tile16(const FusedArgs& a, hipStream_t st) {
    // neighbour lists up to NPI x INFLIGHT entries are fetched in one trip: 14 (KP = 128) / 28 (KP = 64).  13 in flight (the
    // whole 25-entry list of config 3 in ONE trip, 110 VGPRs) is no faster alone and 1.4 us per forward slower with a second
    // batch in flight than 7 (two trips, 86 VGPRs): the smaller block shares a CU more easily (same-box A/B, 3 x 3 runs)
#ifndef SAGE_T16_INFLIGHT
#define SAGE_T16_INFLIGHT 7
#endif
#ifndef SAGE_T16_INFLIGHT_CONCAT
#define SAGE_T16_INFLIGHT_CONCAT 7
#endif
    constexpr int INFLIGHT = CONCAT ? SAGE_T16_INFLIGHT_CONCAT : SAGE_T16_INFLIGHT;
    const int tiles = sage_cdiv(a.n, 16);
    const int grid = min(tiles, sage_tunables().tile16_grid);
    if (sage_tunables().tile16_waves == 8)
        SAGE_LAUNCH_TAIL((layer_tile16_kernel<KP, CONCAT, INFLIGHT, 8>), dim3(grid), dim3(512), 0, st, a);
    else
        SAGE_LAUNCH_TAIL((layer_tile16_kernel<KP, CONCAT, INFLIGHT, 16>), dim3(grid), dim3(1024), 0, st, a);
    SAGE_CHECK_LAUNCH("layer_tile16_kernel");
    return SAGE_OK;
}

template <int KP, int M, int WAVES, bool WREG, bool CONCAT, int INFLIGHT = 8>
int launch(const FusedArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)(CONCAT ? 2 : 1) * M * (KP + 4) * sizeof(float);
    static std::atomic<bool> configured{false};           // role threads (and the express lane's thread) may launch the same kernel concurrently
    if (!configured.load(std::memory_order_acquire)) {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute((const void*)layer_fused_kernel<KP, M, WAVES, WREG, CONCAT, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            sage_set_error("layer_forward: cannot reserve %zu bytes of LDS", lds);
            return SAGE_ELAUNCH;
        }
        configured.store(true, std::memory_order_release);
    }
    const int tiles = sage_cdiv(a.n, M);
    const int per_cu = WAVES >= 16 ? 2 : (WAVES == 8 ? 1 : (WREG ? 2 : 4));
    const int grid = min(tiles, per_cu * kNumCU);
    SAGE_LAUNCH_TAIL((layer_fused_kernel<KP, M, WAVES, WREG, CONCAT, INFLIGHT>), dim3(grid), dim3(WAVES * 64), lds, st, a);
    SAGE_CHECK_LAUNCH("layer_fused_kernel");
    return SAGE_OK;
}

// Tile shape by layer size (measured on MI355X, config-3 layer 1, 23.5k rows x 256 -> 128):
//   64-row tiles, W slice in VGPRs, 2 blocks/CU ............ 82 us
//   32-row tiles, W slice in VGPRs, 2 blocks/CU ............ 80 us
//   32-row tiles, W streamed from L2, 4 blocks/CU (58 VGPR)  77 us   <- big layers
// The gather is bound by the CU's rate of beyond-L2 row fetches, so neither more loads in flight
// per wave (8 -> 16) nor more waves moved it; what the last form buys is four blocks per CU whose
// MFMA phases interleave with the others' gathers.  A small layer (layer 2: 4096 rows) is latency
// bound instead: 16 gathering waves per 32-row tile, and the W slice preloaded into VGPRs BEFORE
// the gather so the MFMA loop never waits on L2 (a streamed W cost ~1 us per k-step there).
template <int KP, bool CONCAT>
int launch_by_rows(const FusedArgs& a, hipStream_t st) {
    constexpr bool kWide = CONCAT && KP == 256;           // two 256-wide chunks: 32-row tiles to fit LDS
    if (a.n >= 8192) {
        if constexpr (kWide) return launch<KP, 32, 4, true, CONCAT>(a, st);
        else if constexpr (KP == 256) return launch<KP, 32, 4, false, CONCAT>(a, st);
        else return launch<KP, 64, 4, true, CONCAT>(a, st);
    }
    // (A dedicated 2-trip kernel for this case -- ids+counts in one load, all rows in flight, W through LDS, 133 KB of
    // LDS per 1024-thread block -- ran 1 us faster alone and 7 % SLOWER with a second batch in flight: its footprint
    // keeps other kernels off the CU.  What shares the chip well beats what is fastest alone.)
#ifdef SAGE_NO_TILE16
    if constexpr (KP <= 128 && !CONCAT) return launch<KP, 32, 16, true, CONCAT, 6>(a, st);    // 6 in flight: inside the 128-VGPR budget of a 16-wave block
    else return launch<KP, 32, 8, true, CONCAT>(a, st);
#else
    if constexpr (KP <= 128) return launch_tile16<KP, CONCAT>(a, st);
    else return launch<KP, 32, 8, true, CONCAT>(a, st);
#endif
}

}  // namespace

bool sage_layer_fused_supported(int32_t dim, int32_t out_dim, int32_t concat) {
    (void)concat;
    return dim >= 4 && dim <= 256 && dim % 4 == 0 && out_dim >= 1 && out_dim <= 128;
}

int sage_launch_layer_fused(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                            const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                            const int32_t* self_row, const int32_t* any_nonempty, int32_t concat, const int32_t* self_index,
                            const float* weight, int64_t ldw, int32_t out_dim, int32_t act, float* out, int64_t ldo,
                            int32_t n_off, sage_finish_t fin, hipStream_t st, const sage_slot_resolve_t* resolve) {
    if (!sage_layer_fused_supported(dim, out_dim, concat) || ld % 4 != 0 || ldw % 4 != 0 || !sage_aligned(table, 16) ||
        !sage_aligned(weight, 16)) {
        sage_set_error("layer_forward: no fused kernel for dim=%d out_dim=%d ld=%lld (needs dim%%4==0, dim<=256, out_dim<=128, 16-B rows)",
                       dim, out_dim, (long long)ld);
        return SAGE_EUNSUPPORTED;
    }
    if (n == 0) return SAGE_OK;
    const FusedArgs a{table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, self_index,
                      table, ld, (int)table_rows, weight, ldw, out_dim, act, out, ldo, n_off, fin,
                      resolve ? resolve->wipe_keys : nullptr, resolve ? resolve->rows_out : nullptr, resolve ? resolve->self_rows_out : nullptr};
    const int kp = dim <= 64 ? 64 : dim <= 128 ? 128 : 256;
    if (!concat) {
        if (kp == 64) return launch_by_rows<64, false>(a, st);
        if (kp == 128) return launch_by_rows<128, false>(a, st);
        return launch_by_rows<256, false>(a, st);
    }
    if (kp == 64) return launch_by_rows<64, true>(a, st);
    if (kp == 128) return launch_by_rows<128, true>(a, st);
    return launch_by_rows<256, true>(a, st);
}

extern "C" int sage_layer_forward_supported(int32_t dim, int32_t out_dim, int32_t concat) {
    return sage_layer_fused_supported(dim, out_dim, concat) ? 1 : 0;
}

extern "C" int sage_layer_forward(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                                  const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                                  const int32_t* self_row, const int32_t* any_nonempty, int32_t concat, const int32_t* self_index,
                                  const float* weight, int64_t ldw, int32_t out_dim, int32_t act, float* out, int64_t ldo,
                                  sage_stream_t stream) {
    SAGE_REQUIRE(table && nbr && cnt && weight && out, "layer_forward: NULL array");
    SAGE_REQUIRE(n >= 0 && k >= 1 && dim >= 1 && out_dim >= 1, "layer_forward: n=%d k=%d dim=%d out_dim=%d", n, k, dim, out_dim);
    SAGE_REQUIRE(ld >= dim && ldo >= out_dim && ldw >= (concat ? 2 : 1) * (int64_t)dim, "layer_forward: leading dimensions");
    SAGE_REQUIRE(table_rows >= 1 && table_rows < (1ll << 31), "layer_forward: table_rows = %lld", (long long)table_rows);
    SAGE_REQUIRE(act >= 0 && act <= SAGE_ACT_NONE, "layer_forward: act = %d", act);
    return sage_launch_layer_fused(table, table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, concat,
                                   self_index, weight, ldw, out_dim, act, out, ldo, 0, sage_finish_t{nullptr, nullptr},
                                   (hipStream_t)stream, nullptr);
}
