// Sampler block body (sample_kernel and sample_fused_kernel, sage_sample.hip): everything is expressed in (bid, nblk)
// instead of blockIdx / gridDim.
#pragma once
#include "sage_internal.h"

namespace sage_sample_detail {


// Where a launch takes its node list / sampler key from: the call arguments, or the current
// descriptor of a device-side batch queue (graph replay).
struct BatchSrc {
    const sage_batch_t* queue;
    const int32_t* cursor;
    int len;
    int nodes_from_batch;     // outer hop: nodes = descriptor seeds; inner hop: only the key
    int32_t* nodes_copy;      // nullable: nodes[r] is also written here (concat: seeds head S1)
    int cursor_off;           // descriptor = queue[(*cursor + cursor_off) % len]  (pipelined forwards sample one batch ahead)
    uint64_t* key_slot;       // nullable: the outer hop leaves the sampler key here, the inner hop takes it from here
                              // instead of the queue (it then never reads the cursor, which another batch's last kernel advances)
    const int32_t* seed_map;  // nullable: outer hop only -- nodes[r] is a CALLER id, seed_map[nodes[r]] the internal one
    int num_nodes;            // ids outside [0, num_nodes) are treated as isolated nodes (degree 0): the reference raises
                              // IndexError for them (nn.Embedding lookup); a device kernel must not walk rowptr[] with them.
                              // 0 = no check (callers that produced the ids themselves)
};

struct FrontierDev {
    int32_t* keys;
    int32_t* rows;
    uint32_t mask;
    int32_t* nodes;
    int32_t* count;
    int32_t max_nodes;
    int32_t row_off;      // rows handed out are row_off + (counter value); forward2 keeps a zero-based counter
};

// Side job for the inner-hop launch of forward2 (its grid is sized for the worst-case frontier, so
// most of its threads are idle): turn the outer hop's hash SLOTS into frontier ROWS for layer 2 and
// wipe the used hash keys, which leaves the table clean for the next forward without a reset pass.
struct ResolveJob {
    const int32_t* slots;      // [n_slots] nbr_slot of the outer hop (-1 = padding)
    int32_t* rows_out;         // [n_slots] frontier row of each slot
    int n_slots;
    const int32_t* self_slots; // nullable [n_self]
    int32_t* self_rows_out;
    int n_self;
    const int32_t* hash_rows;
    int32_t* hash_keys;
};

// The ids a block was FIRST to insert into the frontier, with the rows they got (LDS, filled by sample_block): what the
// fused sampler kernel walks for the inner hop, and the seeds of the block (concat encoder: their own layer-1 samples).
struct WinList {
    int32_t* ids;        // [>= THREADS + THREADS / G]  ids[i] sits in frontier row *base + i
    int* count;          // ids in the list
    int* base;           // frontier row of ids[0]
    int32_t* seeds;      // [THREADS / G] the block's own nodes (internal ids, -1 past the end)
    uint32_t* key;       // [2] the sampler key the block resolved (queue / key slot)
};

// G lanes per node (k <= G).  SAMPLE: draw from the CSR row; otherwise ids come from
// (in_nbr, in_cnt).  FRONTIER: also insert the ids into the hash and reserve frontier rows.
template <int G, int THREADS, bool SAMPLE, bool FRONTIER>
__device__ __forceinline__ void sample_block(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ nodes, int n, const int32_t* __restrict__ n_dev,
    int k, uint32_t key0, uint32_t key1, uint32_t tag, int tag_self_rows, uint32_t tag_self,
    const int32_t* __restrict__ in_nbr, const int32_t* __restrict__ in_cnt,
    int32_t* __restrict__ nbr, int32_t* __restrict__ cnt, int32_t* __restrict__ any_nonempty,
    FrontierDev f, int insert_self, int32_t* __restrict__ nbr_slot, int32_t* __restrict__ self_slot, BatchSrc bs,
    int n_off, ResolveJob rj, const int bid, const int nblk, WinList wl = WinList{nullptr, nullptr, nullptr, nullptr, nullptr}) {
    __shared__ int blk[2];                 // [0] rows claimed by this block, [1] their base row
    constexpr int LT = FRONTIER ? 4 * THREADS : 1;      // block-local dedupe table (ids per block <= THREADS + THREADS/G)
    __shared__ int32_t lkeys[LT];
    __shared__ int32_t lvals[LT];
    constexpr int GPB = THREADS / G;
    const int tid = threadIdx.x;
    const int gl = tid & (G - 1);
    const int lane = tid & (kWave - 1);
    const int r = bid * GPB + tid / G;
    if (FRONTIER) {
        if (tid == 0) blk[0] = 0;
        for (int e = tid; e < LT; e += THREADS) lkeys[e] = -1;
        __syncthreads();
    }
    int nn = n;
    if (n_dev) nn = min(*n_dev + n_off, n);
    if (rj.slots) {
        const int stride = (int)(nblk * THREADS);
        for (int e = (int)(bid * THREADS) + tid; e < rj.n_slots; e += stride) {
            const int sl = rj.slots[e];
            int row = -1;
            if (sl >= 0) { row = rj.hash_rows[sl]; rj.hash_keys[sl] = -1; }
            rj.rows_out[e] = row;
        }
        if (rj.self_slots) {
            for (int e = (int)(bid * THREADS) + tid; e < rj.n_self; e += stride) {
                const int sl = rj.self_slots[e];
                int row = -1;
                if (sl >= 0) { row = rj.hash_rows[sl]; rj.hash_keys[sl] = -1; }
                rj.self_rows_out[e] = row;
            }
        }
    }
    if (bs.queue) {
        if (bs.key_slot && !bs.nodes_from_batch) {
            const uint64_t kq = *bs.key_slot;
            key0 = (uint32_t)kq;
            key1 = (uint32_t)(kq >> 32);
        } else {
            const sage_batch_t b = bs.queue[(uint32_t)(*bs.cursor + bs.cursor_off) % (uint32_t)bs.len];
            key0 = (uint32_t)b.seed;
            key1 = (uint32_t)(b.seed >> 32);
            if (bs.nodes_from_batch) {
                nodes = b.seeds;
                if (bs.key_slot && bid == 0 && tid == 0) *bs.key_slot = b.seed;
            }
        }
    }
    const bool active = r < nn;
    int32_t v = -1, id = -1;
    int c = 0;
    if (wl.key && tid == 0) { wl.key[0] = key0; wl.key[1] = key1; }
    if (SAMPLE) {
        int64_t s = 0, deg = 0;
        if (active) {
            v = nodes[r];
            if (bs.seed_map) v = ((uint32_t)v < (uint32_t)bs.num_nodes) ? bs.seed_map[v] : -1;
            if (bs.nodes_copy && gl == 0) bs.nodes_copy[r] = v;
            if (wl.seeds && gl == 0) wl.seeds[tid / G] = v;
            if (bs.num_nodes == 0 || (uint32_t)v < (uint32_t)bs.num_nodes) {
                s = rowptr[v];
                deg = rowptr[v + 1] - s;
            }
            c = (int)min(deg, (int64_t)k);
        }
        if (wl.seeds && !active && gl == 0) wl.seeds[tid / G] = -1;
        const bool floyd = active && deg > (int64_t)k;
        const uint32_t pos = sage_group_positions<G>(floyd, deg, k, v, (r < tag_self_rows) ? tag_self : tag, key0, key1, gl, lane);
        if (active) {
            if (gl < c) id = __builtin_nontemporal_load(col + s + (int64_t)pos);   // one 4-byte read per drawn position: streaming
            if (gl < k) nbr[(int64_t)r * k + gl] = id;
            if (gl == 0) cnt[r] = c;
        }
    } else {
        if (active) {
            if (nodes && (insert_self || bs.nodes_copy)) {
                v = nodes[r];
                if (bs.nodes_copy && gl == 0) bs.nodes_copy[r] = v;
            }
            c = min(in_cnt[r], k);
            if (gl < c) id = in_nbr[(int64_t)r * k + gl];
        }
    }
    if (any_nonempty) {
        // ONE flag word for the whole launch, so it must not be touched per wave: 6000 waves each doing an atomicOr
        // on it cost 67 us, and even an L1-bypassing load + store per wave cost 27 us per forward (same-address
        // requests queue in one L2 channel).  The block ORs its waves in LDS and thread 0 alone looks at the word
        // (plain, L1-cacheable load) and writes it if it still reads 0.
        __shared__ int blk_any;
        if (tid == 0) blk_any = 0;
        __syncthreads();
        if (__any(c > 0) && lane == 0) blk_any = 1;
        __syncthreads();
        if (tid == 0 && blk_any && *any_nonempty == 0) *any_nonempty = 1;
    }
    if constexpr (FRONTIER) {
        // Two-level insert.  A hub id occurs ~1000 times among the 10^5 ids of a batch; 1000 CAS on one global word
        // serialise at ~12 ns each (the insert took 11-28 us depending on the batch's hubs).  Each block therefore
        // dedupes its own <= THREADS + THREADS/G ids in an LDS hash first (LDS atomics are cheap), only the block's
        // first occurrence of an id goes to the global table, and the others read the slot it got.
        constexpr uint32_t LMASK = LT - 1;
        auto lds_insert = [&](int32_t key, bool& first) -> int {
            uint32_t ls = sage_hash_slot((uint32_t)key, LMASK);
            first = false;
            for (uint32_t probe = 0; probe <= LMASK; ++probe) {
                const int32_t seen = atomicCAS(&lkeys[ls], -1, key);
                if (seen == -1) { first = true; return (int)ls; }
                if (seen == key) return (int)ls;
                ls = (ls + 1) & LMASK;
            }
            return -1;
        };
        bool won = false, selfwon = false, lfirst = false, sfirst = false;
        int slot = -1, sslot = -1, ls = -1, sls = -1;
        if (active && gl < c) ls = lds_insert(id, lfirst);
        if (active && insert_self && gl == 0) sls = lds_insert(v, sfirst);
        if (lfirst) { slot = sage_hash_insert(f.keys, f.mask, id, won); lvals[ls] = slot; }
        if (sfirst) { sslot = sage_hash_insert(f.keys, f.mask, v, selfwon); lvals[sls] = sslot; }
        __syncthreads();
        if (ls >= 0 && !lfirst) slot = lvals[ls];
        if (sls >= 0 && !sfirst) sslot = lvals[sls];
        if (active) {
            if (gl < k) nbr_slot[(int64_t)r * k + gl] = slot;
            if (insert_self && gl == 0) self_slot[r] = sslot;
        }
        const unsigned long long wb = __ballot(won), sb = __ballot(selfwon);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int wcount = __popcll(wb) + __popcll(sb);
        int wbase = 0;
        if (lane == 0 && wcount) wbase = atomicAdd(&blk[0], wcount);
        wbase = __builtin_amdgcn_readfirstlane(wbase);
        __syncthreads();
        if (tid == 0) blk[1] = blk[0] ? atomicAdd(f.count, blk[0]) : 0;
        __syncthreads();
        const int base = f.row_off + blk[1] + wbase;
        if (wl.ids && tid == 0) { *wl.count = blk[0]; *wl.base = f.row_off + blk[1]; }
        if (won) {
            const int row = base + __popcll(wb & below);
            if (row < f.max_nodes) f.nodes[row] = id;
            f.rows[slot] = row;
            if (wl.ids) wl.ids[row - (f.row_off + blk[1])] = id;
        }
        if (selfwon) {
            const int row = base + __popcll(wb) + __popcll(sb & below);
            if (row < f.max_nodes) f.nodes[row] = v;
            f.rows[sslot] = row;
            if (wl.ids) wl.ids[row - (f.row_off + blk[1])] = v;
        }
    }
}


// Inner hop of the fused sampler: `nitems` nodes, item i = (id ids[i], frontier row row0 + i), G1 lanes per node.
// Same draw as sample_block<G1, ..., true, false> makes for that node (Philox counter (v, tag, .), Floyd), so the sets are
// the ones oracle/sampler_ref.c states, whatever kernel drew them.  Every lane of the block must call it.
template <int G1, int THREADS>
__device__ __forceinline__ void sample_inner_items(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int num_nodes, const int32_t* ids, int nitems, int row0, int row_limit,
    int k, uint32_t key0, uint32_t key1, uint32_t tag, int32_t* __restrict__ nbr, int32_t* __restrict__ cnt, bool& any) {
    const int tid = threadIdx.x, gl = tid & (G1 - 1), lane = tid & (kWave - 1);
    constexpr int GPB = THREADS / G1;
    for (int base = 0; base < nitems; base += GPB) {                    // block-uniform trip count
        const int it = base + tid / G1;
        const int row = row0 + it;
        const bool active = it < nitems && row < row_limit;
        int32_t v = -1, id = -1;
        int64_t s = 0, deg = 0;
        int c = 0;
        if (active) {
            v = ids[it];
            if (v >= 0 && (num_nodes == 0 || v < num_nodes)) {
                s = rowptr[v];
                deg = rowptr[v + 1] - s;
            }
            c = (int)min(deg, (int64_t)k);
        }
        const bool floyd = active && deg > (int64_t)k;
        const uint32_t pos = sage_group_positions<G1>(floyd, deg, k, v, tag, key0, key1, gl, lane);
        if (active) {
            if (gl < c) id = __builtin_nontemporal_load(col + s + (int64_t)pos);   // one 4-byte read per drawn position: streaming
            if (gl < k) nbr[(int64_t)row * k + gl] = id;
            if (gl == 0) cnt[row] = c;
        }
        any |= c > 0;
    }
}

}  // namespace sage_sample_detail
