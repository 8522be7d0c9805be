// Fixed-fanout neighbour sampler + frontier (distinct-id) construction.
//
// Replaces, for one hop (include/sage355.h cites the same lines):
//   encoders.py:47          to_neighs = [adj_lists[int(n)] for n in nodes]
//   aggregators.py:42-48    k distinct uniform neighbours, or all if deg < k
//   aggregators.py:52-53    unique_nodes_list / unique_nodes (frontier + id->row)
//
// Shape of the work: tiny and latency bound (B*k2 = 1e5 ids, |S1|*k1 = 3.5e5 ids at
// BASELINE config 3) -- integer work, a chain of dependent HBM round trips per node
// (rowptr -> col -> hash CAS).  A GROUP of G = 8/16/32/64 lanes owns one node and lane i
// owns sample slot i, so every memory step of the chain is one wave-wide instruction
// (k col reads, k hash CAS in flight at once); a first version with one thread per node
// serialised its k CAS round trips and took 52 us for 4096 seeds.  Floyd's subset
// algorithm is inherently sequential in i, but each step only asks "is t_i among the
// i positions already chosen": one group-wide compare + ballot, so the k steps cost
// ~8 instructions each for the whole group.  Frontier rows are reserved per BLOCK
// (wave ballot -> LDS counter -> one global atomic per block): a single counter word
// saturates at ~88 atomics/us on this part.
#include "sage_sample_body.h"

namespace {

using namespace sage_sample_detail;

// One hop's arguments (what sample_kernel used to take as 24 scalars), so that ONE launch can serve several batches (sample_multi_kernel)
struct SampleArgs {
    const int64_t* rowptr; const int32_t* col; const int32_t* nodes; int n; const int32_t* n_dev;
    int k; uint32_t key0, key1, tag; int tag_self_rows; uint32_t tag_self;
    const int32_t* in_nbr; const int32_t* in_cnt; int32_t* nbr; int32_t* cnt; int32_t* any_nonempty;
    FrontierDev f; int insert_self; int32_t* nbr_slot; int32_t* self_slot; BatchSrc bs; int n_off; ResolveJob rj;
};

// Blocks [bid, bid + nblk, ...) of one hop: the launch may be SMALLER than the node list's upper bound (the inner hop's list is
// sized for the worst-case frontier, 4.5 x what a batch at BASELINE config 3 fills: 6656 blocks of which 1470 find a row; the other
// 5186 were dispatched, read the row count and left -- 20 k waves per launch queueing for the slots the gather's waves hold).
// A capped grid walks the list in strides; a block whose first chunk is already past the live rows leaves after the resolve job.
template <int G, int THREADS, bool SAMPLE, bool FRONTIER>
__device__ __forceinline__ void sample_strided(const SampleArgs& a, const int bid, const int nblk) {
    constexpr int GPB = THREADS / G;
    int nn = a.n;
    if (a.n_dev) nn = min(*a.n_dev + a.n_off, a.n);
    for (int b = bid;; b += nblk) {
        sample_block<G, THREADS, SAMPLE, FRONTIER>(a.rowptr, a.col, a.nodes, a.n, a.n_dev, a.k, a.key0, a.key1, a.tag, a.tag_self_rows, a.tag_self,
                                                   a.in_nbr, a.in_cnt, a.nbr, a.cnt, a.any_nonempty, a.f, a.insert_self, a.nbr_slot, a.self_slot,
                                                   a.bs, a.n_off, b == bid ? a.rj : ResolveJob{}, b, nblk);
        if ((int64_t)(b + nblk) * GPB >= (int64_t)nn) break;
        __syncthreads();                   // the block's LDS scratch (dedupe table, counters, flag) is reused by the next chunk
    }
}

// Scalar parameters, not `const SampleArgs a`: with the struct as ONE by-value argument the compiler keeps every field in SGPRs for the
// kernel's whole life (106 SGPRs, 36-42 VGPRs against 35-62 / 19-21 with scalars, where each instantiation drops the arguments it does not
// use): fewer blocks per CU, and the inner hop alone went from 12.2 to 16.0 us (round 4, same-box).  The multi-batch kernel below has no
// choice (its items are picked by blockIdx) and pays that.
template <int G, int THREADS, bool SAMPLE, bool FRONTIER>
__global__ __launch_bounds__(THREADS) void sample_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ nodes, int n, const int32_t* __restrict__ n_dev,
    int k, uint32_t key0, uint32_t key1, uint32_t tag, int tag_self_rows, uint32_t tag_self,
    const int32_t* __restrict__ in_nbr, const int32_t* __restrict__ in_cnt,
    int32_t* __restrict__ nbr, int32_t* __restrict__ cnt, int32_t* __restrict__ any_nonempty,
    FrontierDev f, int insert_self, int32_t* __restrict__ nbr_slot, int32_t* __restrict__ self_slot, BatchSrc bs,
    int n_off, ResolveJob rj) {
    constexpr int GPB = THREADS / G;
    const int bid = (int)blockIdx.x, nblk = (int)gridDim.x;
    sample_block<G, THREADS, SAMPLE, FRONTIER>(rowptr, col, nodes, n, n_dev, k, key0, key1, tag, tag_self_rows, tag_self, in_nbr, in_cnt, nbr, cnt,
                                               any_nonempty, f, insert_self, nbr_slot, self_slot, bs, n_off, rj, bid, nblk);
#ifndef SAGE_S_NOLOOP      /* A/B build without the strided tail (then only valid with SAGE_SI_GRID=0) */
    if constexpr (!FRONTIER) {
        // capped grid (blocks_for): the chunks of the node list past the first pass, if the live rows reach that far
        if ((int64_t)nblk * GPB < (int64_t)n) {
            int nn = n;
            if (n_dev) nn = min(*n_dev + n_off, n);
            for (int b = bid + nblk; (int64_t)b * GPB < (int64_t)nn; b += nblk) {
                __syncthreads();               // the block's LDS scratch (flag word) is reused by the next chunk
                sample_block<G, THREADS, SAMPLE, FRONTIER>(rowptr, col, nodes, n, n_dev, k, key0, key1, tag, tag_self_rows, tag_self, in_nbr, in_cnt,
                                                           nbr, cnt, any_nonempty, f, insert_self, nbr_slot, self_slot, bs, n_off, ResolveJob{}, b, nblk);
            }
        }
    }
#endif
}

// The same hop for up to kSampleMulti BATCHES in one launch (VERDICT r3 #1a): item i's blocks are blockIdx.x % count == i, so the
// blocks that find rows come first for every item.  The hops are latency-bound (4-6 dependent round trips per node, 2 MB of ids):
// twice the rows in flight cost the chain once, and stream S pays one kernel boundary per hop and PAIR of batches instead of one per
// batch.  Every item has a workspace (frontier, counters) of its own: per-call dedupe as aggregators.py:52, results bit-identical
// to one launch per batch.
constexpr int kSampleMulti = 4;
struct SampleMulti { SampleArgs item[kSampleMulti]; int count; };

template <int G, int THREADS, bool SAMPLE, bool FRONTIER>
__global__ __launch_bounds__(THREADS) void sample_multi_kernel(const SampleMulti m) {
    const int it = (int)blockIdx.x % m.count;
    sample_strided<G, THREADS, SAMPLE, FRONTIER>(m.item[it], (int)blockIdx.x / m.count, (int)gridDim.x / m.count);
}

// Outer hop + frontier + inner hop in ONE launch, no grid-wide barrier: the inner hop needs no frontier of its own
// (layer-1 neighbour sets are not deduplicated, aggregators.py:52 works per call), only the ROW its node got, and a block
// knows the rows of the ids it was first to insert as soon as its own counter add returns.  So each block samples the
// outer neighbours of its seeds, inserts them, and then walks the ids it won (LDS list) with lane groups of G1, drawing
// their inner neighbours into nbr1 / cnt1 at those rows; the concat encoder's seed rows (second enc1 call, encoders.py:49-52)
// are drawn by the block that owns the seeds.  What the inner-hop launch used to do besides -- translate the outer hop's hash
// SLOTS into frontier ROWS and wipe the used keys -- has moved into the layer-2 kernel, the only consumer of the rows.
// VERDICT r1 #1 ("merge outer+inner sample (+resolve) into one launch").  Measured: 24.3 us against 10.3 + 11.4 us + a boundary for the
// two launches -- a block's winners (~185 of its 800 ids) take three dependent rounds of its 64 lane groups where the separate inner
// launch spreads 23.6 k nodes over 1500 blocks in one round -- and 66.8 vs 66.2 us per forward in the role pipeline.  Kept as a
// tested option (SAGE_SAMPLE_FUSED=1); the default stays two launches.
template <int G2, int G1, int THREADS>
__global__ __launch_bounds__(THREADS) void sample_fused_kernel(
    const int64_t* __restrict__ rowptr2, const int32_t* __restrict__ col2, const int64_t* __restrict__ rowptr1, const int32_t* __restrict__ col1,
    const int32_t* __restrict__ nodes, int n, int k2, int k1, uint32_t key0, uint32_t key1,
    int32_t* __restrict__ nbr2, int32_t* __restrict__ cnt2, int32_t* __restrict__ any2, FrontierDev f, int insert_self,
    int32_t* __restrict__ nbr_slot, int32_t* __restrict__ self_slot, BatchSrc bs,
    int32_t* __restrict__ nbr1, int32_t* __restrict__ cnt1, int32_t* __restrict__ any1, int seed_rows) {
    constexpr int GPB2 = THREADS / G2;
    __shared__ int32_t wl_ids[THREADS + GPB2];
    __shared__ int32_t wl_seeds[GPB2];
    __shared__ int wl_count, wl_base, blk_any1;
    __shared__ uint32_t wl_key[2];
    const WinList wl{wl_ids, &wl_count, &wl_base, wl_seeds, wl_key};
    if (threadIdx.x == 0) { wl_count = 0; wl_base = 0; blk_any1 = 0; }
    sample_block<G2, THREADS, true, true>(rowptr2, col2, nodes, n, nullptr, k2, key0, key1, SAGE_TAG_OUTER, 0, SAGE_TAG_OUTER, nullptr, nullptr,
                                          nbr2, cnt2, any2, f, insert_self, nbr_slot, self_slot, bs, 0, ResolveJob{}, (int)blockIdx.x,
                                          (int)gridDim.x, wl);
    __syncthreads();
    const uint32_t q0 = wl_key[0], q1 = wl_key[1];
    bool any = false;
    sample_inner_items<G1, THREADS>(rowptr1, col1, bs.num_nodes, wl_ids, wl_count, wl_base, f.max_nodes, k1, q0, q1, SAGE_TAG_INNER, nbr1, cnt1, any);
    if (seed_rows) {
        const int r0 = (int)blockIdx.x * GPB2;
        sample_inner_items<G1, THREADS>(rowptr1, col1, bs.num_nodes, wl_seeds, min(GPB2, n - r0), r0, seed_rows, k1, q0, q1, SAGE_TAG_INNER_SELF,
                                        nbr1, cnt1, any);
    }
    if (any1) {
        if (__any(any) && (threadIdx.x & (kWave - 1)) == 0) blk_any1 = 1;
        __syncthreads();
        if (threadIdx.x == 0 && blk_any1 && *any1 == 0) *any1 = 1;
    }
}

__global__ void frontier_reset_kernel(int32_t* __restrict__ keys, int cap, int32_t* __restrict__ count, int first_row) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = gridDim.x * blockDim.x;
    // cap is a power of two >= 4 and keys is 16-byte aligned (host-checked)
    int4* k4 = reinterpret_cast<int4*>(keys);
    for (int q = i; q < cap / 4; q += stride) k4[q] = make_int4(-1, -1, -1, -1);
    if (i == 0) *count = first_row;
}

int check_frontier(const sage_frontier_t* f, int64_t inserts) {
    SAGE_REQUIRE(f->keys && f->rows && f->nodes && f->count, "frontier: NULL member");
    SAGE_REQUIRE(f->capacity >= 4 && (f->capacity & (f->capacity - 1)) == 0, "frontier: capacity %d not a power of two >= 4", f->capacity);
    SAGE_REQUIRE((int64_t)f->capacity >= 2 * inserts, "frontier: capacity %d < 2 x %lld possible ids", f->capacity, (long long)inserts);
    SAGE_REQUIRE(f->max_nodes > 0, "frontier: max_nodes %d", f->max_nodes);
    SAGE_REQUIRE(sage_aligned(f->keys, 16), "frontier: keys not 16-byte aligned");
    return SAGE_OK;
}

// grid of one item: the whole list, or (no frontier = the inner hop's worst-case list) capped at SAGE_SI_GRID blocks
template <int G, int THREADS, bool FRONTIER>
int blocks_for(int n) {
    const int all = sage_cdiv(n, THREADS / G);
    if (FRONTIER) return all;
    const int cap = sage_tunables().sample_inner_grid;
    return cap > 0 ? min(all, cap) : all;
}

template <int G, int THREADS, bool SAMPLE, bool FRONTIER>
void launch_one(const SampleMulti& m, hipStream_t st) {
    const int blocks = blocks_for<G, THREADS, FRONTIER>(m.item[0].n);
    if (m.count == 1) {
        const SampleArgs& a = m.item[0];
        hipLaunchKernelGGL((sample_kernel<G, THREADS, SAMPLE, FRONTIER>), dim3(blocks), dim3(THREADS), 0, st, a.rowptr, a.col, a.nodes, a.n, a.n_dev, a.k,
                           a.key0, a.key1, a.tag, a.tag_self_rows, a.tag_self, a.in_nbr, a.in_cnt, a.nbr, a.cnt, a.any_nonempty, a.f, a.insert_self,
                           a.nbr_slot, a.self_slot, a.bs, a.n_off, a.rj);
    } else
        hipLaunchKernelGGL((sample_multi_kernel<G, THREADS, SAMPLE, FRONTIER>), dim3(blocks * m.count), dim3(THREADS), 0, st, m);
}

template <int T, bool SAMPLE, bool FRONTIER>
void launch_by_fanout_t(int k, const SampleMulti& m, hipStream_t st) {
    if (k <= 8) launch_one<8, T, SAMPLE, FRONTIER>(m, st);
    else if (k <= 16) launch_one<16, T, SAMPLE, FRONTIER>(m, st);
    else if (k <= 32) launch_one<32, T, SAMPLE, FRONTIER>(m, st);
    else launch_one<64, T, SAMPLE, FRONTIER>(m, st);
}

template <bool SAMPLE, bool FRONTIER>
void launch_by_fanout(int k, const SampleMulti& m, hipStream_t st) {
    // frontier variants use 1024-thread blocks: one global counter atomic per 1024/G nodes (256- and 512-thread
    // blocks were measured 3-5 % slower end to end)
    if constexpr (FRONTIER) {
        const int so = sage_tunables().outer_threads;
        if (so == 256) launch_by_fanout_t<256, SAMPLE, FRONTIER>(k, m, st);
        else if (so == 512) launch_by_fanout_t<512, SAMPLE, FRONTIER>(k, m, st);
        else launch_by_fanout_t<1024, SAMPLE, FRONTIER>(k, m, st);
    }
#ifndef SAGE_SI_THREADS
#define SAGE_SI_THREADS 256
#endif
    else launch_by_fanout_t<SAGE_SI_THREADS, SAMPLE, FRONTIER>(k, m, st);
}

// items of one launch must agree in everything that picks the kernel and its grid
int launch_items(const SampleMulti& m, bool sample, hipStream_t st) {
    const SampleArgs& a0 = m.item[0];
    const bool frontier = a0.f.keys != nullptr;
    for (int i = 1; i < m.count; ++i) {
        const SampleArgs& a = m.item[i];
        SAGE_REQUIRE(a.k == a0.k && a.n == a0.n && (a.f.keys != nullptr) == frontier, "sample (multi): items differ in fanout / list size / frontier");
    }
    if (sample && frontier) launch_by_fanout<true, true>(a0.k, m, st);
    else if (sample) launch_by_fanout<true, false>(a0.k, m, st);
    else launch_by_fanout<false, true>(a0.k, m, st);
    SAGE_CHECK_LAUNCH("sample_kernel");
    return SAGE_OK;
}

}  // namespace

// Internal launcher shared with sage_forward.hip (tag_self_rows: rows [0, tag_self_rows)
// draw from stream `tag_self` -- the concat encoder's second enc1 call on the seeds).
// Collector (sage_internal.h): while a thread has one open, sage_launch_sample APPENDS its hop instead of launching it
struct sage_sample_batch { SampleMulti m; bool open; };
static thread_local sage_sample_batch t_collect{{}, false};

void sage_sample_collect_begin() { t_collect.m.count = 0; t_collect.open = true; }
int sage_sample_collect_count() { return t_collect.open ? t_collect.m.count : 0; }
int sage_sample_collect_launch(hipStream_t st) {
    t_collect.open = false;
    if (t_collect.m.count == 0) return SAGE_OK;
    return launch_items(t_collect.m, true, st);
}
void sage_sample_collect_abort() { t_collect.open = false; t_collect.m.count = 0; }

int sage_launch_sample(const int64_t* rowptr, const int32_t* col, int64_t num_nodes, const int32_t* nodes, int32_t n, const int32_t* n_dev,
                       int32_t k, uint64_t seed, uint32_t tag, int32_t tag_self_rows, uint32_t tag_self,
                       int32_t* nbr, int32_t* cnt, int32_t* any_nonempty, const sage_frontier_t* frontier,
                       int32_t insert_self, int32_t* nbr_slot, int32_t* self_slot, const sage_model_t* qm, int nodes_from_batch,
                       int32_t* nodes_copy, int32_t n_off, int32_t frontier_row_off, const sage_resolve_t* resolve,
                       int32_t cursor_off, uint64_t* key_slot, const int32_t* seed_map, hipStream_t st) {
    if (n == 0) return SAGE_OK;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const BatchSrc bs{qm ? qm->queue : nullptr, qm ? qm->queue_cursor : nullptr, qm ? qm->queue_len : 0, nodes_from_batch, nodes_copy,
                      cursor_off, key_slot, (qm && nodes_from_batch) ? qm->seed_map : seed_map, (int)num_nodes};
    ResolveJob rj{};
    if (resolve) rj = ResolveJob{resolve->slots, resolve->rows_out, resolve->n_slots, resolve->self_slots, resolve->self_rows_out,
                                 resolve->n_self, resolve->hash_rows, resolve->hash_keys};
    FrontierDev fd{};
    if (frontier)
        fd = FrontierDev{frontier->keys, frontier->rows, (uint32_t)frontier->capacity - 1u,
                         frontier->nodes, frontier->count, frontier->max_nodes, frontier_row_off};
    const SampleArgs a{rowptr, col, nodes, n, n_dev, k, k0, k1, tag, tag_self_rows, tag_self, nullptr, nullptr, nbr, cnt, any_nonempty,
                       fd, frontier ? insert_self : 0, frontier ? nbr_slot : nullptr, frontier ? self_slot : nullptr, bs, n_off, rj};
    if (t_collect.open) {
        SAGE_REQUIRE(t_collect.m.count < kSampleMulti, "sample (multi): more than %d hops collected for one launch", kSampleMulti);
        t_collect.m.item[t_collect.m.count++] = a;
        return SAGE_OK;
    }
    SampleMulti m;
    m.item[0] = a;
    m.count = 1;
    return launch_items(m, true, st);
}

namespace {
template <int G2, int G1, int T, typename... A>
void launch_fused_t(int n, hipStream_t st, A... args) {
    hipLaunchKernelGGL((sample_fused_kernel<G2, G1, T>), dim3(sage_cdiv(n, T / G2)), dim3(T), 0, st, args...);
}
template <int G2, int G1, typename... A>
void launch_fused_by_threads(int n, hipStream_t st, A... args) {
    if (sage_tunables().outer_threads >= 1024) launch_fused_t<G2, G1, 1024>(n, st, args...);
    else launch_fused_t<G2, G1, 512>(n, st, args...);
}
template <int G2, typename... A>
void launch_fused_by_k1(int k1, int n, hipStream_t st, A... args) {
    if (k1 <= 16) launch_fused_by_threads<G2, 16>(n, st, args...);
    else if (k1 <= 32) launch_fused_by_threads<G2, 32>(n, st, args...);
    else launch_fused_by_threads<G2, 64>(n, st, args...);
}
}  // namespace

// Both hops of a forward as one launch (see sample_fused_kernel).  `seed_rows` = batch for the concat encoder (rows [0, batch) of
// S1 are the seeds themselves), else 0.
int sage_launch_sample_fused(const sage_model_t* m, const int32_t* seeds, int32_t batch, uint64_t seed, int32_t* nbr2, int32_t* cnt2,
                             int32_t* any2, const sage_frontier_t* frontier, int32_t insert_self, int32_t* nbr_slot, int32_t* self_slot,
                             int queued, int32_t* nodes_copy, int32_t frontier_row_off, int32_t* nbr1, int32_t* cnt1, int32_t* any1,
                             int32_t seed_rows, hipStream_t st) {
    if (batch == 0) return SAGE_OK;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const BatchSrc bs{queued ? m->queue : nullptr, queued ? m->queue_cursor : nullptr, queued ? m->queue_len : 0, 1, nodes_copy, 0, nullptr,
                      m->seed_map, (int)m->num_nodes};
    const FrontierDev fd{frontier->keys, frontier->rows, (uint32_t)frontier->capacity - 1u, frontier->nodes, frontier->count,
                         frontier->max_nodes, frontier_row_off};
    if (m->k2 <= 16)
        launch_fused_by_k1<16>(m->k1, batch, st, m->rowptr2, m->col2, m->rowptr1, m->col1, seeds, batch, m->k2, m->k1, k0, k1, nbr2, cnt2, any2, fd,
                               insert_self, nbr_slot, self_slot, bs, nbr1, cnt1, any1, seed_rows);
    else if (m->k2 <= 32)
        launch_fused_by_k1<32>(m->k1, batch, st, m->rowptr2, m->col2, m->rowptr1, m->col1, seeds, batch, m->k2, m->k1, k0, k1, nbr2, cnt2, any2, fd,
                               insert_self, nbr_slot, self_slot, bs, nbr1, cnt1, any1, seed_rows);
    else
        launch_fused_by_k1<64>(m->k1, batch, st, m->rowptr2, m->col2, m->rowptr1, m->col1, seeds, batch, m->k2, m->k1, k0, k1, nbr2, cnt2, any2, fd,
                               insert_self, nbr_slot, self_slot, bs, nbr1, cnt1, any1, seed_rows);
    SAGE_CHECK_LAUNCH("sample_fused_kernel");
    return SAGE_OK;
}

extern "C" int sage_frontier_reset(const sage_frontier_t* f, int32_t first_row, sage_stream_t stream) {
    SAGE_REQUIRE(f, "frontier_reset: NULL frontier");
    if (int rc = check_frontier(f, 0)) return rc;
    SAGE_REQUIRE(first_row >= 0 && first_row <= f->max_nodes, "frontier_reset: first_row %d outside [0, %d]", first_row, f->max_nodes);
    const int blocks = min(sage_cdiv(f->capacity / 4, 256), 1024);
    hipLaunchKernelGGL(frontier_reset_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, f->keys, f->capacity, f->count, first_row);
    SAGE_CHECK_LAUNCH("frontier_reset_kernel");
    return SAGE_OK;
}

extern "C" int sage_sample_neighbors(const int64_t* rowptr, const int32_t* col, int64_t num_nodes, const int32_t* nodes,
                                     int32_t n, const int32_t* n_dev, int32_t k, uint64_t seed, uint32_t tag, int32_t* nbr,
                                     int32_t* cnt, int32_t* any_nonempty, const sage_frontier_t* frontier, int32_t insert_self,
                                     int32_t* nbr_slot, int32_t* self_slot, sage_stream_t stream) {
    SAGE_REQUIRE(rowptr && col && nodes && nbr && cnt, "sample_neighbors: NULL array");
    SAGE_REQUIRE(n >= 0, "sample_neighbors: n = %d", n);
    SAGE_REQUIRE(k >= 1 && k <= SAGE_MAX_FANOUT, "sample_neighbors: k = %d outside [1, %d]", k, SAGE_MAX_FANOUT);
    SAGE_REQUIRE(num_nodes > 0 && num_nodes < (1ll << 31), "sample_neighbors: num_nodes = %lld", (long long)num_nodes);
    if (frontier) {
        if (int rc = check_frontier(frontier, (int64_t)n * (k + (insert_self ? 1 : 0)))) return rc;
        SAGE_REQUIRE(nbr_slot, "sample_neighbors: frontier given but nbr_slot is NULL");
        SAGE_REQUIRE(!insert_self || self_slot, "sample_neighbors: insert_self needs self_slot");
    }
    return sage_launch_sample(rowptr, col, num_nodes, nodes, n, n_dev, k, seed, tag, 0, tag, nbr, cnt, any_nonempty, frontier, insert_self,
                              nbr_slot, self_slot, nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int sage_frontier_insert(const int32_t* nbr, const int32_t* cnt, int32_t k, const int32_t* self_nodes, int32_t n,
                                    const int32_t* n_dev, const sage_frontier_t* frontier, int32_t* nbr_slot, int32_t* self_slot,
                                    sage_stream_t stream) {
    SAGE_REQUIRE(nbr && cnt && frontier && nbr_slot, "frontier_insert: NULL array");
    SAGE_REQUIRE(n >= 0, "frontier_insert: n = %d", n);
    SAGE_REQUIRE(k >= 1 && k <= SAGE_MAX_FANOUT, "frontier_insert: k = %d outside [1, %d]", k, SAGE_MAX_FANOUT);
    SAGE_REQUIRE(!self_nodes || self_slot, "frontier_insert: self_nodes needs self_slot");
    if (int rc = check_frontier(frontier, (int64_t)n * (k + (self_nodes ? 1 : 0)))) return rc;
    if (n == 0) return SAGE_OK;
    const FrontierDev fd{frontier->keys, frontier->rows, (uint32_t)frontier->capacity - 1u,
                         frontier->nodes, frontier->count, frontier->max_nodes, 0};
    SampleMulti m;
    m.item[0] = SampleArgs{nullptr, nullptr, self_nodes, n, n_dev, k, 0u, 0u, 0u, 0, 0u, nbr, cnt, nullptr, nullptr, nullptr, fd, self_nodes ? 1 : 0,
                           nbr_slot, self_slot, BatchSrc{nullptr, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0}, 0, ResolveJob{}};
    m.count = 1;
    return launch_items(m, false, (hipStream_t)stream);
}
