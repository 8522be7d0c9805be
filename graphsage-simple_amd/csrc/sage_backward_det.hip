// Reproducible mean backward (round 3): grad_table[t] = sum over the (r, j) with row(nbr[r, j]) == t of grad_agg[r] / c_r,
// every row's terms added in ascending (r, j) order -- the gradient of aggregators.py:60-74, which torch autograd computes
// deterministically on the reference's CPU (model.py:249).  The legacy kernel (sage_backward.hip: gather_mean_bwd_kernel) scatters
// with fp32 atomics, i.e. in order of arrival: two runs of one schedule differed in the last bits, and a captured step matched the
// eager one to 1e-4 only.
//
// Inverted index per call, all on the device, no host synchronisation, everything in the caller's workspace:
//   1. expand   every (r, j) slot -> key = the table row it points at (or the sentinel `table_rows`: padding, ids out of range,
//               rows past the live count), value = r; 1 / c_r per row, with the forward's set-union rule for the self row
//   2. sort     hipcub::DeviceRadixSort::SortPairs on the key bits that matter -- a stable sort, so equal keys keep the slot
//               order (r ascending, then j)
//   3. heads    start[t] / end[t] of every key's run in the sorted list
//   4. sum      one lane group per table row walks its run and STORES the sum (zeros for a row nobody points at), so the
//               caller need not zero grad_table either
// Integer / index work throughout except step 4; nothing here is reshaped into a GEMM.
#include <hipcub/hipcub.hpp>

#include "sage_internal.h"

namespace {

struct DetLayout {
    size_t keys_in, keys_out, vals_in, vals_out, inv_c, start, end, cub, total;
    size_t cub_bytes;
    int slots, bits;
};

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

int key_bits(int64_t table_rows) {
    int bits = 1;
    while ((1ll << bits) <= table_rows) ++bits;         // keys are in [0, table_rows] (the sentinel included)
    return bits;
}

// The radix sort's temporary size is a host-side query (no launch, no allocation).
hipError_t sort_temp_bytes(int slots, int bits, size_t* bytes) {
    *bytes = 0;
    return hipcub::DeviceRadixSort::SortPairs(nullptr, *bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                                              (int32_t*)nullptr, slots, 0, bits, (hipStream_t)0);
}

bool det_layout(int n, int k, int64_t table_rows, DetLayout* L) {
    L->slots = n * (k + 1);
    L->bits = key_bits(table_rows);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    L->keys_in = take((size_t)L->slots * 4);
    L->keys_out = take((size_t)L->slots * 4);
    L->vals_in = take((size_t)L->slots * 4);
    L->vals_out = take((size_t)L->slots * 4);
    L->inv_c = take((size_t)n * 4);
    L->start = take((size_t)table_rows * 4);
    L->end = take((size_t)table_rows * 4);
    if (sort_temp_bytes(L->slots, L->bits, &L->cub_bytes) != hipSuccess) return false;
    L->cub = take(L->cub_bytes + 16);
    L->total = off;
    return true;
}

// one wave per destination row r: its k + 1 slots (k neighbours + the self row)
__global__ __launch_bounds__(256) void det_expand_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int k, int n,
                                                         const int32_t* __restrict__ n_dev, const int32_t* __restrict__ slot_rows,
                                                         const int32_t* __restrict__ self_row, int table_rows, int32_t* __restrict__ keys,
                                                         int32_t* __restrict__ vals, float* __restrict__ inv_c) {
    int nn = n;
    if (n_dev) nn = min(*n_dev, n);
    const int lane = sage_lane();
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    for (int r = wave; r < n; r += nwaves) {
        int32_t* kr = keys + (int64_t)r * (k + 1);
        int32_t* vr = vals + (int64_t)r * (k + 1);
        if (r >= nn) {                                               // past the live count: nothing but sentinels
            for (int j = lane; j <= k; j += kWave) { kr[j] = table_rows; vr[j] = r; }
            if (lane == 0) inv_c[r] = 0.f;
            continue;
        }
        const int c = min(__builtin_amdgcn_readfirstlane(cnt[r]), k);
        int s = -1;
        if (self_row) {
            s = self_row[r];
            if (slot_rows && s >= 0) s = slot_rows[s];
            s = __builtin_amdgcn_readfirstlane(s);
        }
        bool extra = s >= 0;
        for (int base = 0; base < k; base += kWave) {
            const int j = base + lane;
            int id = -1;
            if (j < c) {
                id = nbr[(int64_t)r * k + j];
                if (slot_rows) id = slot_rows[max(id, 0)];
            }
            if (extra && __any(j < c && id == s)) extra = false;     // aggregators.py:50-51: set union (the forward's rule)
            if (j < k) {
                kr[j] = (j < c && id >= 0 && id < table_rows) ? id : table_rows;
                vr[j] = r;
            }
        }
        if (lane == 0) {
            kr[k] = (extra && s < table_rows) ? s : table_rows;      // wave-uniform `extra`: every lane saw every ballot
            vr[k] = r;
            const int ceff = c + (extra ? 1 : 0);
            inv_c[r] = ceff > 0 ? 1.0f / (float)ceff : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void det_heads_kernel(const int32_t* __restrict__ keys, int slots, int table_rows,
                                                        int32_t* __restrict__ start, int32_t* __restrict__ end) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    const int key = keys[i];
    if (key >= table_rows) return;                                   // the sentinel's run (the tail of the list)
    if (i == 0 || keys[i - 1] != key) start[key] = i;
    if (i == slots - 1 || keys[i + 1] != key) end[key] = i + 1;
}

// LG lanes (x 16 B) per table row: a row's run of the sorted list, terms in list order
template <int LG>
__global__ __launch_bounds__(256) void det_sum_kernel(const float* __restrict__ gagg, int64_t ldg, int dim, const int32_t* __restrict__ vals,
                                                      const float* __restrict__ inv_c, const int32_t* __restrict__ start,
                                                      const int32_t* __restrict__ end, int table_rows, const int32_t* __restrict__ rows_dev,
                                                      float* __restrict__ gtab, int64_t ld) {
    using V = __attribute__((ext_vector_type(4))) float;
    int live = table_rows;
    if (rows_dev) live = min(*rows_dev, table_rows);
    const int gid = (int)((blockIdx.x * blockDim.x + threadIdx.x) / LG), gl = (int)(threadIdx.x % LG);
    const int ngroups = (int)((gridDim.x * blockDim.x) / LG);
    for (int t = gid; t < live; t += ngroups) {
        const int a = start[t], b = end[t];
        for (int c0 = gl * 4; c0 < dim; c0 += LG * 4) {              // dim % 4 == 0 (host-checked)
            V acc = {0.f, 0.f, 0.f, 0.f};
            for (int i = a; i < b; ++i) {
                const int r = vals[i];
                const float w = inv_c[r];
                const V g = *reinterpret_cast<const V*>(gagg + (int64_t)r * ldg + c0);
                acc[0] += g[0] * w; acc[1] += g[1] * w; acc[2] += g[2] * w; acc[3] += g[3] * w;
            }
            *reinterpret_cast<V*>(gtab + (int64_t)t * ld + c0) = acc;
        }
    }
}

}  // namespace

// ---- canonical order of a layer's rows ---------------------------------------------------------------------------------------------
// The frontier's rows are in ARBITRARY order (whichever lane won the hash slot first, aggregators.py:52: a Python set's order is
// arbitrary too), so a sum over the layer's rows -- the weight gradient -- adds in a different order run after run.  Nothing else
// depends on the order.  order[] lists the live rows canonically: rows [0, first_row) (the concat encoder's own seeds, in seed order)
// as they are, then the frontier rows by ascending node id (ids are distinct there).  One key per row + a stable radix sort.
namespace {
__global__ __launch_bounds__(256) void order_keys_kernel(const int32_t* __restrict__ nodes, int n, const int32_t* __restrict__ n_dev,
                                                         int first_row, int32_t* __restrict__ keys, int32_t* __restrict__ vals) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int nn = n;
    if (n_dev) nn = min(*n_dev, n);
    const int64_t key = r < first_row ? (int64_t)r : (int64_t)first_row + (int64_t)max(nodes[r], 0);
    keys[r] = r < nn ? (int32_t)min(key, (int64_t)0x7FFFFFFE) : 0x7FFFFFFF;      // dead rows sort to the end
    vals[r] = r;
}
struct OrderLayout { size_t keys_in, keys_out, vals_in, cub, total, cub_bytes; };
bool order_layout(int n, OrderLayout* L) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align256(off + bytes); return o; };
    L->keys_in = take((size_t)n * 4);
    L->keys_out = take((size_t)n * 4);
    L->vals_in = take((size_t)n * 4);
    if (sort_temp_bytes(n, 31, &L->cub_bytes) != hipSuccess) return false;
    L->cub = take(L->cub_bytes + 16);
    L->total = off;
    return true;
}
}  // namespace

extern "C" size_t sage_row_order_workspace_bytes(int32_t n) {
    if (n <= 0) return 0;
    OrderLayout L;
    return order_layout(n, &L) ? L.total : 0;
}

extern "C" int sage_row_order(const int32_t* nodes, int32_t n, const int32_t* n_dev, int32_t first_row, int32_t* order,
                              void* workspace, size_t workspace_bytes, sage_stream_t stream) {
    SAGE_REQUIRE(nodes && order && workspace && sage_aligned(workspace, 256), "row_order: NULL / unaligned argument");
    SAGE_REQUIRE(n >= 1 && first_row >= 0 && first_row <= n, "row_order: n = %d, first_row = %d", n, first_row);
    OrderLayout L;
    SAGE_REQUIRE(order_layout(n, &L), "row_order: radix sort size query failed");
    if (L.total > workspace_bytes) {
        sage_set_error("row_order: workspace %zu bytes < %zu needed", workspace_bytes, L.total);
        return SAGE_ENOSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int32_t* keys_in = (int32_t*)(ws + L.keys_in);
    int32_t* keys_out = (int32_t*)(ws + L.keys_out);
    int32_t* vals_in = (int32_t*)(ws + L.vals_in);
    hipLaunchKernelGGL(order_keys_kernel, dim3(sage_cdiv(n, 256)), dim3(256), 0, st, nodes, n, n_dev, first_row, keys_in, vals_in);
    SAGE_CHECK_LAUNCH("order_keys_kernel");
    size_t cub_bytes = L.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub_bytes, (const int32_t*)keys_in, keys_out, (const int32_t*)vals_in, order, n, 0, 31, st) !=
        hipSuccess) {
        sage_set_error("row_order: radix sort failed");
        return SAGE_ELAUNCH;
    }
    return SAGE_OK;
}

extern "C" size_t sage_gather_mean_backward_workspace_bytes(int32_t n, int32_t k, int64_t table_rows) {
    if (n <= 0 || k <= 0 || table_rows <= 0 || table_rows >= (1ll << 31) || (int64_t)n * (k + 1) >= (1ll << 31)) return 0;
    DetLayout L;
    return det_layout(n, k, table_rows, &L) ? L.total : 0;
}

extern "C" int sage_gather_mean_backward_ws(const float* grad_agg, int64_t ldg, int32_t dim, const int32_t* nbr, const int32_t* cnt,
                                            int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                                            const int32_t* self_row, float* grad_table, int64_t table_rows, const int32_t* table_rows_dev,
                                            int64_t ld, void* workspace, size_t workspace_bytes, sage_stream_t stream) {
    SAGE_REQUIRE(grad_agg && nbr && cnt && grad_table && workspace, "gather_mean_backward_ws: NULL array");
    SAGE_REQUIRE(n >= 0 && k >= 1 && dim >= 4 && dim % 4 == 0 && ldg >= dim && ld >= dim && ldg % 4 == 0 && ld % 4 == 0,
                 "gather_mean_backward_ws: n=%d k=%d dim=%d (rows of 16-byte pieces)", n, k, dim);
    SAGE_REQUIRE(sage_aligned(grad_agg, 16) && sage_aligned(grad_table, 16) && sage_aligned(workspace, 256), "gather_mean_backward_ws: alignment");
    SAGE_REQUIRE(table_rows >= 1 && table_rows < (1ll << 31) && (int64_t)n * (k + 1) < (1ll << 31), "gather_mean_backward_ws: table_rows = %lld",
                 (long long)table_rows);
    if (n == 0) return SAGE_OK;
    DetLayout L;
    SAGE_REQUIRE(det_layout(n, k, table_rows, &L), "gather_mean_backward_ws: radix sort size query failed");
    if (L.total > workspace_bytes) {
        sage_set_error("gather_mean_backward_ws: workspace %zu bytes < %zu needed", workspace_bytes, L.total);
        return SAGE_ENOSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int32_t* keys_in = (int32_t*)(ws + L.keys_in);
    int32_t* keys_out = (int32_t*)(ws + L.keys_out);
    int32_t* vals_in = (int32_t*)(ws + L.vals_in);
    int32_t* vals_out = (int32_t*)(ws + L.vals_out);
    float* inv_c = (float*)(ws + L.inv_c);
    int32_t* start = (int32_t*)(ws + L.start);
    int32_t* end = (int32_t*)(ws + L.end);
    hipLaunchKernelGGL(det_expand_kernel, dim3(min(sage_cdiv(n, 4), kNumCU * 8)), dim3(256), 0, st, nbr, cnt, k, n, n_dev, slot_rows, self_row,
                       (int)table_rows, keys_in, vals_in, inv_c);
    SAGE_CHECK_LAUNCH("det_expand_kernel");
    size_t cub_bytes = L.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(ws + L.cub, cub_bytes, (const int32_t*)keys_in, keys_out, (const int32_t*)vals_in, vals_out, L.slots, 0,
                                           L.bits, st) != hipSuccess) {
        sage_set_error("gather_mean_backward_ws: radix sort failed");
        return SAGE_ELAUNCH;
    }
    // start == end == 0 for rows nobody points at (start and end are adjacent in the workspace: one memset)
    if (int rc = sage_fill_u32(start, 0u, ((size_t)((char*)end - (char*)start) + (size_t)table_rows * 4) / 4, st)) return rc;
    hipLaunchKernelGGL(det_heads_kernel, dim3(sage_cdiv(L.slots, 256)), dim3(256), 0, st, (const int32_t*)keys_out, L.slots, (int)table_rows, start, end);
    SAGE_CHECK_LAUNCH("det_heads_kernel");
    const int lg = dim >= 256 ? 64 : dim >= 128 ? 32 : dim >= 64 ? 16 : 8;
    const int64_t groups = table_rows;
    const int blocks = (int)min((int64_t)kNumCU * 8, (groups * lg + 255) / 256);
#define SAGE_DET_SUM(LGV) hipLaunchKernelGGL(det_sum_kernel<LGV>, dim3(blocks), dim3(256), 0, st, grad_agg, ldg, dim, (const int32_t*)vals_out, \
                                             (const float*)inv_c, (const int32_t*)start, (const int32_t*)end, (int)table_rows, table_rows_dev, grad_table, ld)
    if (lg == 64) SAGE_DET_SUM(64);
    else if (lg == 32) SAGE_DET_SUM(32);
    else if (lg == 16) SAGE_DET_SUM(16);
    else SAGE_DET_SUM(8);
#undef SAGE_DET_SUM
    SAGE_CHECK_LAUNCH("det_sum_kernel");
    return SAGE_OK;
}
