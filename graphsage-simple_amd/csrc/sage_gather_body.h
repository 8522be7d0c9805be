// Column-sliced gather block bodies (the kernels of sage_gather.hip), expressed in (bid, nblk) instead of blockIdx / gridDim.
#pragma once
#include "sage_internal.h"

namespace sage_gather_detail {

// ---- column-sliced gather (wide rows, large layers) ------------------------------------------------------
// Per-XCD L2s are private: with one wave per destination ROW every XCD ends up caching its own copy of the
// hub rows, the L2 hit rate is ~20 % and 85 % of the gathered bytes come from beyond L2 (DESIGN_HISTORY.md section 3).
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
    float* __restrict__ out, int64_t ldo, int n_off, int nslice, const int bid, const int nblk, const int64_t slice_stride = 0, const int act = SAGE_ACT_NONE) {
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
    // slice-major table (slice_stride != 0): slice s of every row is one contiguous [rows, SL * 4] array at table + s * slice_stride
    const float* __restrict__ tcol = table + (slice_stride ? (int64_t)slice * slice_stride + gl * 4 : (int64_t)c0);
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
                if (ok && j < c) t[u] = *reinterpret_cast<const V*>(tcol + (int64_t)id * ld);
                else t[u] = V{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += t[u];
        }
        if (extra && ok && grp == 0) acc += *reinterpret_cast<const V*>(tcol + (int64_t)min(s, last_row) * ld);
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
            // streaming stores: the next kernel reads these rows from other XCDs anyway, and dirty lines left in L2 are
            // written back at the kernel boundary, on the critical path (gather 49.4 -> 48.3 us, contraction 22.2 -> 21.3)
            if (act != SAGE_ACT_NONE) { res[0] = sage_activate(res[0], act); res[1] = sage_activate(res[1], act); res[2] = sage_activate(res[2], act); res[3] = sage_activate(res[3], act); }
            sage_store_stream<SAGE_AGG_STORE>(reinterpret_cast<V*>(out + (int64_t)r * ldo + c0), res);
        }
    }
}

// ---- the same gather, software-pipelined over rows ---------------------------------------------------------------
// The kernel above walks cnt/ids -> first 8 neighbours -> remaining neighbours as three DEPENDENT round trips per
// destination row, so a wave has row data in flight only ~2/3 of the time and the kernel needs all 32 waves of a CU to
// cover the fabric's latency -- which leaves no wave slot for any other kernel (rocprofv3 timeline, two batches in flight:
// the 10-us outer sampler ran 45 us beside a gather, the 24-us contraction 60 us; both were waiting for SLOTS).
// Here the count and ids of the wave's NEXT rows are requested before the current row's data is waited for, and every
// neighbour of a row (up to U*NPI) is requested in ONE trip, so a wave keeps U wave-instructions in flight all the time
// and the same bytes-in-flight per CU need half (or a quarter) of the waves: the rest of the CU is free for the samplers,
// the contraction and layer 2 of other batches (sage_pipe.hip).  Results are bit-identical to gather_sliced_block
// only when the summation order matches, which it does: same lane-group partial sums, same xor-shuffle tree.
template <int SL, int U, int R>
__device__ __forceinline__ void gather_sliced_block_pipelined(
    const float* __restrict__ table, int table_rows, int64_t ld, int dim,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int k, int n, const int32_t* __restrict__ n_dev,
    const int32_t* __restrict__ slot_rows, const int32_t* __restrict__ self_row, const int32_t* __restrict__ any_nonempty,
    float* __restrict__ out, int64_t ldo, int n_off, int nslice, const int bid, const int nblk, const int64_t slice_stride = 0, const int act = SAGE_ACT_NONE) {
    // R rows of the wave are in flight together (R x U wave-instructions): narrow slices (SL = 8: 128 B, one slice per
    // XCD, no hub row cached twice on the chip) put only 2 KiB of a row into one trip, too little to cover the fabric's
    // latency with the waves a shared CU can spare.
    using V = __attribute__((ext_vector_type(4))) float;
    constexpr int NPI = kWave / SL;             // neighbours per wave-instruction
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
    const float* __restrict__ tcol = table + (slice_stride ? (int64_t)slice * slice_stride + gl * 4 : (int64_t)c0);   // slice-major: see above

    int r0 = wave;                              // the group's rows: r0 + i * nwaves, i < R
    int c_cur[R], id_cur[R], s_cur[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int r = r0 + i * nwaves;
        c_cur[i] = 0; id_cur[i] = 0; s_cur[i] = -1;
        if (r < nn) {
            c_cur[i] = cnt[r];
            id_cur[i] = (lane < k) ? nbr[(int64_t)r * k + lane] : 0;
            if (self_row) s_cur[i] = self_row[r];
        }
    }
    while (r0 < nn) {
        const int rn0 = r0 + R * nwaves;
        // ---- request the NEXT group's counts / ids / self rows (consumed in the next iteration)
        int c_nxt[R], id_nxt[R], s_nxt[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = rn0 + i * nwaves;
            c_nxt[i] = 0; id_nxt[i] = 0; s_nxt[i] = -1;
            if (r < nn) {
                c_nxt[i] = cnt[r];
                id_nxt[i] = (lane < k) ? nbr[(int64_t)r * k + lane] : 0;
                if (self_row) s_nxt[i] = self_row[r];
            }
        }
        // ---- this group: every row's first trip is requested before any of them is reduced
        int cc[R], ss[R], ids[R];
        bool extra[R];
        V t[R][U];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            cc[i] = min(__builtin_amdgcn_readfirstlane(c_cur[i]), kWave);   // k <= 64 (host-checked); 0 for rows past nn
            int s = s_cur[i];
            if (slot_rows && s >= 0) s = slot_rows[s];
            ss[i] = __builtin_amdgcn_readfirstlane(s);
            int myid = (lane < cc[i]) ? id_cur[i] : 0;
            if (slot_rows) myid = slot_rows[max(myid, 0)];
            extra[i] = ss[i] >= 0;
            if (extra[i] && __any(lane < cc[i] && myid == ss[i])) extra[i] = false;   // aggregators.py:50-51: set union
            ids[i] = min(max(myid, 0), last_row);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = u * NPI + grp;
                const int id = __shfl(ids[i], min(j, max(cc[i] - 1, 0)), kWave);
                if (ok && j < cc[i]) t[i][u] = *reinterpret_cast<const V*>(tcol + (int64_t)id * ld);
                else t[i][u] = V{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = r0 + i * nwaves;
            if (r >= nn) continue;                                          // wave-uniform
            const int c = cc[i];
            V acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < U; ++u) acc += t[i][u];
            for (int j0 = NPI * U; j0 < c; j0 += NPI * U) {                  // lists longer than one trip
                V w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = j0 + u * NPI + grp;
                    const int id = __shfl(ids[i], min(j, c - 1), kWave);
                    if (ok && j < c) w[u] = *reinterpret_cast<const V*>(tcol + (int64_t)id * ld);
                    else w[u] = V{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc += w[u];
            }
            if (extra[i] && ok && grp == 0) acc += *reinterpret_cast<const V*>(tcol + (int64_t)min(ss[i], last_row) * ld);
#pragma unroll
            for (int m = SL; m < kWave; m <<= 1) {
                acc[0] += __shfl_xor(acc[0], m, kWave);
                acc[1] += __shfl_xor(acc[1], m, kWave);
                acc[2] += __shfl_xor(acc[2], m, kWave);
                acc[3] += __shfl_xor(acc[3], m, kWave);
            }
            if (grp == 0 && ok) {
                const int ceff = c + (extra[i] ? 1 : 0);
                V res;
                if (ceff > 0) res = acc * (1.0f / (float)ceff);
                else { const float fill = nan_rule ? __builtin_nanf("") : 0.f; res = V{fill, fill, fill, fill}; }
                if (act != SAGE_ACT_NONE) { res[0] = sage_activate(res[0], act); res[1] = sage_activate(res[1], act); res[2] = sage_activate(res[2], act); res[3] = sage_activate(res[3], act); }
            sage_store_stream<SAGE_AGG_STORE>(reinterpret_cast<V*>(out + (int64_t)r * ldo + c0), res);
            }
        }
        r0 = rn0;
#pragma unroll
        for (int i = 0; i < R; ++i) { c_cur[i] = c_nxt[i]; id_cur[i] = id_nxt[i]; s_cur[i] = s_nxt[i]; }
    }
}

// ---- sliced gather, one destination ROW per lane group ("rows" form) -------------------------------------------------
// In the two kernels above a wave works on ONE destination row: its lane groups fetch different NEIGHBOURS of it, ids
// travel by ds_bpermute and the groups' partial sums meet in xor-shuffles -- ~150 issued instructions per (row, slice)
// unit.  That is invisible at 256-B slices (memory time dominates) but it is what made 128-B slices slow (189 k units:
// 53 us at full occupancy whatever the loads in flight), although 128-B slices -- one slice per XCD, no hub row cached
// twice on the chip -- cut the bytes that leave L2 from 214 to 171 MB.  Here lane group g owns destination row
// (block of 64/SL rows) + g and walks ITS neighbour list: a wave-instruction fetches the j-th neighbour slice of 64/SL
// different rows, every lane reads its ids with plain loads (L1 hits after the first lane), sums its own 4 columns in
// registers and stores them: no cross-lane traffic at all, ~10x fewer instructions per unit, and TRIP neighbours of every
// row (64/SL x TRIP KiB per wave) in flight at once, so a few waves per CU cover the fabric's latency.
// Summation order: neighbours in list order, one fp32 add per neighbour (as the reference's mask.mm row sum does per column).
template <int SL, int TRIP, bool SLOT>
__device__ __forceinline__ void gather_sliced_block_rows(
    const float* __restrict__ table, int table_rows, int64_t ld, int dim,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int k, int n, const int32_t* __restrict__ n_dev,
    const int32_t* __restrict__ slot_rows, const int32_t* __restrict__ self_row, const int32_t* __restrict__ any_nonempty,
    float* __restrict__ out, int64_t ldo, int n_off, int nslice, const int bid, const int nblk, const int64_t slice_stride = 0, const int act = SAGE_ACT_NONE) {
    using V = __attribute__((ext_vector_type(4))) float;
    constexpr int NG = kWave / SL;              // destination rows per wave-instruction
    int nn = n;
    if (n_dev) nn = min(*n_dev + n_off, n);
    const int lane = sage_lane();
    const int slice = (int)(bid % nslice);
    const int wave = (int)(((bid / nslice) * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)(((nblk / nslice) * blockDim.x) >> 6);
    const int grp = lane / SL, gl = lane % SL;
    const int c0 = min(slice * SL * 4 + gl * 4, dim - 4);    // this lane's columns (lanes past the row width re-read its last
    const bool ok = slice * SL * 4 + gl * 4 < dim;           // 16 bytes and store nothing); dim % 4 == 0 (host-checked)
    const bool nan_rule = any_nonempty ? (*any_nonempty != 0) : false;
    const int last_row = table_rows - 1;
    const float* __restrict__ tcol = table + (slice_stride ? (int64_t)slice * slice_stride + min(gl * 4, SL * 4 - 4) : (int64_t)c0);
    const int nblocks_rows = (nn + NG - 1) / NG;
    for (int rb = wave; rb < nblocks_rows; rb += nwaves) {
        const int r = rb * NG + grp;
        const bool valid = r < nn;
        const int rq = valid ? r : nn - 1;                           // lanes past the end shadow the last row (L1 hits, no store)
        const int c = min(cnt[rq], k);
        int s = -1;
        if (self_row) {
            s = self_row[rq];
            if (SLOT && s >= 0) s = slot_rows[s];
        }
        bool extra = s >= 0;
        const int32_t* __restrict__ myn = nbr + (int64_t)rq * k;
        V acc = {0.f, 0.f, 0.f, 0.f};
        // No branch and no wait inside a trip: every id load and every row load is unconditional -- slots past the
        // list's end re-read the list's LAST row (an L1 hit, nothing leaves the CU) and get weight 0 -- so all TRIP
        // rows of all 64/SL destinations are in flight together.  (A predicated form compiled to one exec branch and one
        // s_waitcnt vmcnt(0) PER LOAD: 54 us at 4 blocks per CU.)
        for (int j0 = 0; j0 < k; j0 += TRIP) {                      // wave-uniform bounds; one trip when k <= TRIP
            int id[TRIP];
#pragma unroll
            for (int u = 0; u < TRIP; ++u) id[u] = myn[min(min(j0 + u, c - 1), k - 1) < 0 ? 0 : min(min(j0 + u, c - 1), k - 1)];
            V t[TRIP];
#pragma unroll
            for (int u = 0; u < TRIP; ++u) {
                int x = id[u];
                if (SLOT) x = slot_rows[max(x, 0)];
                extra = extra && !(j0 + u < c && x == s);           // aggregators.py:50-51: set union
                x = min(max(x, 0), last_row);
                t[u] = *reinterpret_cast<const V*>(tcol + (int64_t)x * ld);
            }
#pragma unroll
            for (int u = 0; u < TRIP; ++u) {
                const float w = (j0 + u < c) ? 1.f : 0.f;
                // select, not multiply: a slot past the end may hold Inf / NaN of a row that is not in the set
                acc[0] += (j0 + u < c) ? t[u][0] : 0.f; acc[1] += (j0 + u < c) ? t[u][1] : 0.f;
                acc[2] += (j0 + u < c) ? t[u][2] : 0.f; acc[3] += (j0 + u < c) ? t[u][3] : 0.f;
                (void)w;
            }
        }
        if (extra) {
            const V sv = *reinterpret_cast<const V*>(tcol + (int64_t)min(s, last_row) * ld);
            acc += sv;
        }
        if (valid && ok) {
            const int ceff = c + (extra ? 1 : 0);
            V res;
            if (ceff > 0) res = acc * (1.0f / (float)ceff);
            else { const float fill = nan_rule ? __builtin_nanf("") : 0.f; res = V{fill, fill, fill, fill}; }
            if (act != SAGE_ACT_NONE) { res[0] = sage_activate(res[0], act); res[1] = sage_activate(res[1], act); res[2] = sage_activate(res[2], act); res[3] = sage_activate(res[3], act); }
            sage_store_stream<SAGE_AGG_STORE>(reinterpret_cast<V*>(out + (int64_t)r * ldo + c0), res);
        }
    }
}

}  // namespace sage_gather_detail
