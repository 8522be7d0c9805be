// Fixed-fanout neighbour sampler + frontier (distinct-id) construction.
//
// Replaces, for one hop (include/sage355.h cites the same lines):
//   encoders.py:47          to_neighs = [adj_lists[int(n)] for n in nodes]
//   aggregators.py:42-48    k distinct uniform neighbours, or all if deg < k
//   aggregators.py:52-53    unique_nodes_list / unique_nodes (frontier + id->row)
//
// Shape of the work: tiny and latency bound (B*k2 = 1e5 ids, |S1|*k1 = 3.5e5 ids at
// BASELINE config 3) -- integer work, three dependent HBM round trips per node
// (rowptr -> col -> hash CAS).  One THREAD per node keeps all 64 lanes of a wave busy
// on different nodes (a wave-per-node Floyd loop would idle 63 lanes in its sequential
// part); the k chosen positions of a thread live in LDS, laid out [slot][thread] so a
// wave's accesses hit 64 distinct banks.  Frontier rows are reserved per BLOCK (one
// global atomic per 128 nodes): a single counter word saturates at ~88 atomics/us on
// this part, so per-thread reservation would cost hundreds of us.
#include "sage_common.h"

namespace {

constexpr int kThreads = 128;

struct FrontierDev {
    int32_t* keys;
    int32_t* rows;
    uint32_t mask;
    int32_t* nodes;
    int32_t* count;
    int32_t max_nodes;
};

// SAMPLE: draw from the CSR row; otherwise ids come from (in_nbr, in_cnt).
// FRONTIER: also insert the ids into the hash and reserve frontier rows.
template <bool SAMPLE, bool FRONTIER>
__global__ __launch_bounds__(kThreads) void sample_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ nodes, int n, const int32_t* __restrict__ n_dev,
    int k, uint32_t key0, uint32_t key1, uint32_t tag, int tag_self_rows, uint32_t tag_self,
    const int32_t* __restrict__ in_nbr, const int32_t* __restrict__ in_cnt,
    int32_t* __restrict__ nbr, int32_t* __restrict__ cnt, int32_t* __restrict__ any_nonempty,
    FrontierDev f, int insert_self, int32_t* __restrict__ nbr_slot, int32_t* __restrict__ self_slot) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    constexpr int T = kThreads;
    const int tid = threadIdx.x;
    const int r = blockIdx.x * T + tid;
    int nn = n;
    if (n_dev) nn = min(*n_dev, n);
    const bool active = r < nn;
    int32_t* ids = lds;                       // [k][T]
    int c = 0;
    int32_t v = -1;
    if (active) {
        if (SAMPLE) {
            v = nodes[r];
            const int64_t s = rowptr[v];
            const int64_t deg = rowptr[v + 1] - s;
            if (deg <= (int64_t)k) {
                c = (int)deg;
                for (int j = 0; j < c; ++j) ids[j * T + tid] = col[s + j];
            } else {
                // Floyd: a uniform k-subset of positions [0, deg) in k steps.
                c = k;
                const uint32_t base = (uint32_t)(deg - (int64_t)k);
                const uint32_t t_ = (r < tag_self_rows) ? tag_self : tag;
                for (int i0 = 0; i0 < k; i0 += 4) {
                    const Philox4 rnd = philox4x32_10((uint32_t)v, t_, (uint32_t)(i0 >> 2), 0u, key0, key1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = i0 + q;
                        if (i < k) {
                            const uint32_t j = base + (uint32_t)i;
                            const uint32_t t = sage_bounded(rnd.v[q], j + 1u);
                            bool dup = false;
                            for (int m = 0; m < i; ++m) dup |= (ids[m * T + tid] == (int32_t)t);
                            ids[i * T + tid] = (int32_t)(dup ? j : t);
                        }
                    }
                }
                for (int i = 0; i < k; ++i) ids[i * T + tid] = col[s + (int64_t)(uint32_t)ids[i * T + tid]];
            }
            cnt[r] = c;
            for (int j = 0; j < k; ++j) nbr[(int64_t)r * k + j] = (j < c) ? ids[j * T + tid] : -1;
        } else {
            v = nodes ? nodes[r] : -1;
            c = min(in_cnt[r], k);
            for (int j = 0; j < c; ++j) ids[j * T + tid] = in_nbr[(int64_t)r * k + j];
        }
    }
    if (any_nonempty) {
        if (__any(c > 0) && sage_lane() == 0) atomicOr(any_nonempty, 1);
    }
    if constexpr (FRONTIER) {
        int32_t* slots = lds + k * T;          // [k+1][T]
        int32_t* blk = lds + (2 * k + 1) * T;  // [0] rows claimed by this block, [1] their base row
        if (tid == 0) blk[0] = 0;
        __syncthreads();
        unsigned long long wonmask = 0ull;
        bool selfwon = false;
        int nwon = 0;
        if (active) {
            for (int j = 0; j < c; ++j) {
                bool won;
                const int slot = sage_hash_insert(f.keys, f.mask, ids[j * T + tid], won);
                slots[j * T + tid] = slot;
                nbr_slot[(int64_t)r * k + j] = slot;
                if (won) { wonmask |= 1ull << j; ++nwon; }
            }
            for (int j = c; j < k; ++j) nbr_slot[(int64_t)r * k + j] = -1;
            if (insert_self) {
                const int slot = sage_hash_insert(f.keys, f.mask, v, selfwon);
                slots[k * T + tid] = slot;
                self_slot[r] = slot;
                if (selfwon) ++nwon;
            }
        }
        const int first = nwon ? atomicAdd(&blk[0], nwon) : 0;
        __syncthreads();
        if (tid == 0) blk[1] = blk[0] ? atomicAdd(f.count, blk[0]) : 0;
        __syncthreads();
        int row = blk[1] + first;
        for (int j = 0; j < c; ++j) {
            if ((wonmask >> j) & 1ull) {
                if (row < f.max_nodes) f.nodes[row] = ids[j * T + tid];
                f.rows[slots[j * T + tid]] = row;
                ++row;
            }
        }
        if (selfwon) {
            if (row < f.max_nodes) f.nodes[row] = v;
            f.rows[slots[k * T + tid]] = row;
        }
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

}  // namespace

// Internal launcher shared with sage_forward.hip (tag_self_rows: rows [0, tag_self_rows)
// draw from stream `tag_self` -- the concat encoder's second enc1 call on the seeds).
int sage_launch_sample(const int64_t* rowptr, const int32_t* col, const int32_t* nodes, int32_t n, const int32_t* n_dev,
                       int32_t k, uint64_t seed, uint32_t tag, int32_t tag_self_rows, uint32_t tag_self,
                       int32_t* nbr, int32_t* cnt, int32_t* any_nonempty, const sage_frontier_t* frontier,
                       int32_t insert_self, int32_t* nbr_slot, int32_t* self_slot, hipStream_t st) {
    if (n == 0) return SAGE_OK;
    const int blocks = sage_cdiv(n, kThreads);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    FrontierDev fd{};
    if (frontier) {
        fd = FrontierDev{frontier->keys, frontier->rows, (uint32_t)frontier->capacity - 1u,
                         frontier->nodes, frontier->count, frontier->max_nodes};
        const size_t lds = ((size_t)(2 * k + 1) * kThreads + 2) * sizeof(int32_t);
        hipLaunchKernelGGL((sample_kernel<true, true>), dim3(blocks), dim3(kThreads), lds, st, rowptr, col, nodes, n, n_dev, k,
                           k0, k1, tag, tag_self_rows, tag_self, nullptr, nullptr, nbr, cnt, any_nonempty, fd, insert_self,
                           nbr_slot, self_slot);
    } else {
        const size_t lds = (size_t)k * kThreads * sizeof(int32_t);
        hipLaunchKernelGGL((sample_kernel<true, false>), dim3(blocks), dim3(kThreads), lds, st, rowptr, col, nodes, n, n_dev, k,
                           k0, k1, tag, tag_self_rows, tag_self, nullptr, nullptr, nbr, cnt, any_nonempty, fd, 0, nullptr,
                           nullptr);
    }
    SAGE_CHECK_LAUNCH("sample_kernel");
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
    return sage_launch_sample(rowptr, col, nodes, n, n_dev, k, seed, tag, 0, tag, nbr, cnt, any_nonempty, frontier, insert_self,
                              nbr_slot, self_slot, (hipStream_t)stream);
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
                         frontier->nodes, frontier->count, frontier->max_nodes};
    const size_t lds = ((size_t)(2 * k + 1) * kThreads + 2) * sizeof(int32_t);
    hipLaunchKernelGGL((sample_kernel<false, true>), dim3(sage_cdiv(n, kThreads)), dim3(kThreads), lds, (hipStream_t)stream,
                       nullptr, nullptr, self_nodes, n, n_dev, k, 0u, 0u, 0u, 0, 0u, nbr, cnt, nullptr, nullptr, nullptr, fd,
                       self_nodes ? 1 : 0, nbr_slot, self_slot);
    SAGE_CHECK_LAUNCH("frontier_insert_kernel");
    return SAGE_OK;
}
