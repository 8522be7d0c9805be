// Column-sliced gather block body (shared by gather_mean_sliced_kernel, sage_gather.hip, and the fused gather + next-batch
// outer sample launch, sage_pipeline.hip): expressed in (bid, nblk) instead of blockIdx / gridDim.
#pragma once
#include "sage_internal.h"

namespace sage_gather_detail {

// ---- column-sliced gather (wide rows, large layers) ------------------------------------------------------
// Per-XCD L2s are private: with one wave per destination ROW every XCD ends up caching its own copy of the
// hub rows, the L2 hit rate is ~20 % and 85 % of the gathered bytes come from beyond L2.
// Here a block owns one SLICE of SL*4 columns (SL lanes x 16 B) of every row it touches, and consecutive
// blocks -- which the dispatcher deals round-robin over the 8 XCDs -- own different slices.  An XCD then only
// ever caches its slice of the hot rows (4x more rows per L2 at 256-B slices), which is what lifts the hit rate.
// Inside a wave, lane group g = lane / SL fetches neighbour g's slice, so one wave-instruction still moves
// 1 KiB (64/SL neighbours x SL*16 B); the groups' partial sums are combined with xor-shuffles (the wavefront
// reduction), and group 0 writes the mean.  Placement is a speed assumption only: any block -> XCD map is correct.
// Measured (config-3 layer 1, 338 k row gathers of 1 KiB): row-per-wave 65 us, 128-B slices 54 us, 256-B slices 45 us.
template <int SL>
__device__ __forceinline__ void gather_sliced_block(
    const float* __restrict__ table, int table_rows, int64_t ld, int dim,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int k, int n, const int32_t* __restrict__ n_dev,
    const int32_t* __restrict__ slot_rows, const int32_t* __restrict__ self_row, const int32_t* __restrict__ any_nonempty,
    float* __restrict__ out, int64_t ldo, int n_off, int nslice, const int bid, const int nblk) {
    using V = __attribute__((ext_vector_type(4))) float;
    constexpr int NPI = kWave / SL;             // neighbours per wave-instruction
    // wave-instructions in flight: 8 neighbours per trip.  Deeper (4 x 4 neighbours) is no faster alone -- 32 waves per
    // CU already cover the latency -- and costs the OTHER batch's latency-bound kernels 3.5 us per forward: every
    // request queued here is latency added to their dependent round trips (same-box A/B, two batches in flight).
#ifdef SAGE_G_INFLIGHT
    constexpr int U = SAGE_G_INFLIGHT;
#else
    constexpr int U = 8 / NPI;
#endif
    int nn = n;
    if (n_dev) nn = min(*n_dev + n_off, n);
    const int lane = sage_lane();
    const int slice = (int)(bid % nslice);
    const int wave = (int)(((bid / nslice) * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)(((nblk / nslice) * blockDim.x) >> 6);
    const int grp = lane / SL, gl = lane % SL;
    const int c0 = slice * SL * 4 + gl * 4;     // this lane's columns
    const bool ok = c0 < dim;                   // dim % 4 == 0 (host-checked)
    const bool nan_rule = any_nonempty ? (*any_nonempty != 0) : false;
    const int last_row = table_rows - 1;
    for (int r = wave; r < nn; r += nwaves) {
        const int c = min(__builtin_amdgcn_readfirstlane(cnt[r]), kWave);   // k <= 64 (host-checked)
        int s = -1;
        if (self_row) {
            s = self_row[r];
            if (slot_rows && s >= 0) s = slot_rows[s];
            s = __builtin_amdgcn_readfirstlane(s);
        }
        int myid = (lane < c) ? nbr[(int64_t)r * k + lane] : 0;
        if (slot_rows) myid = slot_rows[max(myid, 0)];
        bool extra = s >= 0;
        if (extra && __any(lane < c && myid == s)) extra = false;           // aggregators.py:50-51: set union
        myid = min(max(myid, 0), last_row);
        V acc = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < c; j0 += NPI * U) {
            V t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + u * NPI + grp;
                const int id = __shfl(myid, min(j, c - 1), kWave);
                if (ok && j < c) t[u] = *reinterpret_cast<const V*>(table + (int64_t)id * ld + c0);
                else t[u] = V{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += t[u];
        }
        if (extra && ok && grp == 0) acc += *reinterpret_cast<const V*>(table + (int64_t)min(s, last_row) * ld + c0);
#pragma unroll
        for (int m = SL; m < kWave; m <<= 1) {
            acc[0] += __shfl_xor(acc[0], m, kWave);
            acc[1] += __shfl_xor(acc[1], m, kWave);
            acc[2] += __shfl_xor(acc[2], m, kWave);
            acc[3] += __shfl_xor(acc[3], m, kWave);
        }
        if (grp == 0 && ok) {
            const int ceff = c + (extra ? 1 : 0);
            V res;
            if (ceff > 0) res = acc * (1.0f / (float)ceff);
            else { const float fill = nan_rule ? __builtin_nanf("") : 0.f; res = V{fill, fill, fill, fill}; }
#ifndef SAGE_NO_NT_STORES   // streaming stores: the next kernel reads these rows from other XCDs anyway, and dirty lines left in L2 are
                            // written back at the kernel boundary, on the critical path (gather 49.4 -> 48.3 us, contraction 22.2 -> 21.3)
            __builtin_nontemporal_store(res, reinterpret_cast<V*>(out + (int64_t)r * ldo + c0));
#else
            *reinterpret_cast<V*>(out + (int64_t)r * ldo + c0) = res;
#endif
        }
    }
}


}  // namespace sage_gather_detail
