// gather_mean: per-row mean of gathered feature rows (two-launch form).
//
// Replaces aggregators.py:54-74 -- the dense [B,U] 0/1 mask, its two normalisation
// passes, the frontier feature fetch and mask.mm(embed_matrix) -- by reading exactly
// the sampled rows.  HBM-bound: algorithmic bytes per destination row = 4*dim*cnt
// gathered + 4*dim written.  One wavefront per destination row; a lane
// owns 4 consecutive columns (16-B loads, 1 KiB per wave-instruction = one 256-float
// row), neighbour ids are broadcast from a VGPR with v_readlane so every row address
// is scalar; the j-loop is unrolled so 8 row loads are in flight per wave.
#include <hip/hip_ext.h>

#include "sage_gather_body.h"

// Measurement hook (bench.py's dominant-kernel duration): when set by the calling thread, the column-sliced gather is launched through
// hipExtLaunchKernelGGL with these two TIMING events as the launch's own start / stop events, i.e. the interval is the kernel's
// execution (what rocprofv3 reports) and not the distance between two marker packets around it in a busy queue.
thread_local const sage_ext_launch_t* sage_ext_launch = nullptr;
thread_local void* sage_tail_event = nullptr;

namespace {

template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<1> { using type = float; };

__device__ inline void vadd(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ inline void vadd(float& a, const float& b) { a += b; }
__device__ inline float4 vscale(const float4& a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ inline float vscale(const float& a, float s) { return a * s; }
__device__ inline void vfill(float4& a, float s) { a = make_float4(s, s, s, s); }
__device__ inline void vfill(float& a, float s) { a = s; }

template <int VEC>
__global__ __launch_bounds__(256) void gather_mean_kernel(
    const float* __restrict__ table, int table_rows, int64_t ld, int dim,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int k, int n, const int32_t* __restrict__ n_dev,
    const int32_t* __restrict__ slot_rows, const int32_t* __restrict__ self_row, const int32_t* __restrict__ any_nonempty,
    float* __restrict__ out, int64_t ldo, int n_off) {
    using V = typename VecT<VEC>::type;
    int nn = n;
    if (n_dev) nn = min(*n_dev + n_off, n);
    const int lane = sage_lane();
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    const bool nan_rule = any_nonempty ? (*any_nonempty != 0) : false;
    const int last_row = table_rows - 1;

    for (int r = wave; r < nn; r += nwaves) {
        const int c = __builtin_amdgcn_readfirstlane(cnt[r]);
        int s = -1;
        if (self_row) {
            s = self_row[r];
            if (slot_rows && s >= 0) s = slot_rows[s];
            s = __builtin_amdgcn_readfirstlane(s);
        }
        // is the self row already among the sampled ones?  (aggregators.py:50-51: set union)
        bool extra = s >= 0;
        if (extra) {
            for (int base = 0; base < c; base += kWave) {
                int id = (base + lane < c) ? nbr[(int64_t)r * k + base + lane] : -1;
                if (slot_rows && id >= 0) id = slot_rows[id];
                if (__any(id == s)) extra = false;
            }
        }
        const int ceff = c + (extra ? 1 : 0);
        const float inv = 1.0f / (float)ceff;

        for (int cb = 0; cb < dim; cb += kWave * VEC) {
            const int c0 = cb + lane * VEC;
            const bool ok = c0 < dim;
            V acc;
            vfill(acc, 0.f);
            for (int base = 0; base < c; base += kWave) {
                const int m = min(kWave, c - base);
                int myid = (lane < m) ? nbr[(int64_t)r * k + base + lane] : 0;
                if (slot_rows) myid = slot_rows[max(myid, 0)];
                myid = min(max(myid, 0), last_row);      // never fault on a bad id
                // 8 row loads in flight per wave; the tail re-reads row m-1 (an L1 hit) with weight 0
                for (int j0 = 0; j0 < m; j0 += 8) {
                    V t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int id = __builtin_amdgcn_readlane(myid, min(j0 + u, m - 1));
                        if (ok) t[u] = *reinterpret_cast<const V*>(table + (int64_t)id * ld + c0);
                        else vfill(t[u], 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + u < m) vadd(acc, t[u]);
                }
            }
            if (extra && ok) vadd(acc, *reinterpret_cast<const V*>(table + (int64_t)min(s, last_row) * ld + c0));
            if (ok) {
                V res;
                if (ceff > 0) res = vscale(acc, inv);
                else vfill(res, nan_rule ? __builtin_nanf("") : 0.f);
                *reinterpret_cast<V*>(out + (int64_t)r * ldo + c0) = res;
            }
        }
    }
}

using sage_gather_detail::gather_sliced_block;

template <int SL>
__global__ __launch_bounds__(256) void gather_mean_sliced_kernel(
    const float* __restrict__ table, int table_rows, int64_t ld, int dim,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int k, int n, const int32_t* __restrict__ n_dev,
    const int32_t* __restrict__ slot_rows, const int32_t* __restrict__ self_row, const int32_t* __restrict__ any_nonempty,
    float* __restrict__ out, int64_t ldo, int n_off, int nslice, int64_t slice_stride, int act) {
    gather_sliced_block<SL>(table, table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice,
                            (int)blockIdx.x, (int)gridDim.x, slice_stride, act);
}

using sage_gather_detail::gather_sliced_block_pipelined;

template <int SL, int U, int R>
__global__ __launch_bounds__(256) void gather_mean_sliced_pipe_kernel(
    const float* __restrict__ table, int table_rows, int64_t ld, int dim,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int k, int n, const int32_t* __restrict__ n_dev,
    const int32_t* __restrict__ slot_rows, const int32_t* __restrict__ self_row, const int32_t* __restrict__ any_nonempty,
    float* __restrict__ out, int64_t ldo, int n_off, int nslice, int64_t slice_stride, int act) {
    gather_sliced_block_pipelined<SL, U, R>(table, table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo,
                                         n_off, nslice, (int)blockIdx.x, (int)gridDim.x, slice_stride, act);
}

using sage_gather_detail::gather_sliced_block_rows;

template <int SL, int TRIP, bool SLOT>
__global__ __launch_bounds__(256) void gather_mean_rows_kernel(
    const float* __restrict__ table, int table_rows, int64_t ld, int dim,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt, int k, int n, const int32_t* __restrict__ n_dev,
    const int32_t* __restrict__ slot_rows, const int32_t* __restrict__ self_row, const int32_t* __restrict__ any_nonempty,
    float* __restrict__ out, int64_t ldo, int n_off, int nslice, int64_t slice_stride, int act) {
    gather_sliced_block_rows<SL, TRIP, SLOT>(table, table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo,
                                       n_off, nslice, (int)blockIdx.x, (int)gridDim.x, slice_stride, act);
}

// one launch, plain or (measurement hook) with the launch's own start / stop events
#define SAGE_LAUNCH_G(kernel, ...)                                                                                                   \
    do {                                                                                                                             \
        if (const sage_ext_launch_t* x_ = sage_ext_launch)                                                                           \
            hipExtLaunchKernelGGL((kernel), dim3(blocks), dim3(256), 0, st, (hipEvent_t)x_->start, (hipEvent_t)x_->stop, 0u, __VA_ARGS__); \
        else                                                                                                                         \
            SAGE_LAUNCH_TAIL((kernel), dim3(blocks), dim3(256), 0, st, __VA_ARGS__);                                                 \
    } while (0)

template <int SL, typename... A>
void launch_rows(int blocks, bool slot, hipStream_t st, A... args) {
    if (slot) SAGE_LAUNCH_G((gather_mean_rows_kernel<SL, 8, true>), args...);
    else if (sage_tunables().gather_trip >= 16) SAGE_LAUNCH_G((gather_mean_rows_kernel<SL, 16, false>), args...);
    else SAGE_LAUNCH_G((gather_mean_rows_kernel<SL, 8, false>), args...);
}

template <int SL, int U, typename... A>
void launch_pipe(int blocks, hipStream_t st, A... args) {
    const int rows = sage_tunables().gather_rows_in_flight;
    if (rows >= 4) SAGE_LAUNCH_G((gather_mean_sliced_pipe_kernel<SL, U, 4>), args...);
    else if (rows >= 2) SAGE_LAUNCH_G((gather_mean_sliced_pipe_kernel<SL, U, 2>), args...);
    else SAGE_LAUNCH_G((gather_mean_sliced_pipe_kernel<SL, U, 1>), args...);
}

}  // namespace

bool sage_gather_is_sliced(int32_t dim, int64_t ld, int64_t ldo, const float* table, const float* out, int32_t n, int32_t k) {
    const bool vec4 = (dim % 4 == 0) && (ld % 4 == 0) && (ldo % 4 == 0) && sage_aligned(table, 16) && sage_aligned(out, 16);
    return vec4 && k <= kWave && ((dim >= SAGE_SPLIT_MIN_DIM && n >= 8192) || dim > 256);   // rows wider than 256 have no one-launch kernel
}

int sage_launch_gather_mean(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                            const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                            const int32_t* self_row, const int32_t* any_nonempty, float* out, int64_t ldo, int32_t n_off,
                            hipStream_t st, int64_t slice_stride, int act) {
    if (n == 0) return SAGE_OK;
    if (slice_stride != 0 || sage_gather_is_sliced(dim, ld, ldo, table, out, n, k)) {
        // 256-B slices (16 lanes; 512-B rows: 29.1 us as two slices vs 32.5 us as one).  A narrow row that does not end
        // on a slice boundary is ONE slice of 32 lanes instead (two neighbours per wave-instruction): 400-B rows
        // (config 5) 40.5 us vs 45.5 us as a 256-B + a 144-B slice
#ifdef SAGE_SLICE_LANES
        constexpr bool kForce = true;
        const int sl = SAGE_SLICE_LANES;
#else
        constexpr bool kForce = false;
        const int forced = sage_tunables().gather_slice_lanes;
        int sl = forced ? forced : ((dim <= 128 && dim % 64 != 0) ? 32 : 16);
        if (sl == 64 && sage_tunables().gather_variant != 1) sl = 16;      // whole-row slices exist in the pipelined form only (row-major tables)
        if (slice_stride != 0) sl = (int)(ld / 4);                         // a slice-major table: ld IS the slice width (32 / 64 / 128 floats)
        // which kernel: with a slice-major table the one-row-per-lane-group form wins (128-B slices: 60.6 vs 71.7 us per forward)
        const int variant = slice_stride != 0 ? sage_tunables().gather_variant_sliced : sage_tunables().gather_variant;
#endif
        (void)kForce;
        const int nslice = sage_cdiv(dim, sl * 4);
        const int blocks = nslice * max(1, kNumCU * sage_tunables().gather_blocks_per_cu / nslice);   // >= one block per slice (very wide rows)
        if (variant == 2) {
            // one destination row per lane group (see sage_gather_body.h)
            if (sl == 8) launch_rows<8>(blocks, slot_rows != nullptr, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
            else if (sl == 32) launch_rows<32>(blocks, slot_rows != nullptr, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
            else launch_rows<16>(blocks, slot_rows != nullptr, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
            SAGE_CHECK_LAUNCH("gather_mean_rows_kernel");
            return SAGE_OK;
        }
        if (variant == 1) {
            // rows software-pipelined: every neighbour of a row in ONE trip (U wave-instructions of 64/sl neighbours), the
            // next row's ids requested meanwhile.  U by fanout; lists longer than U x 64/sl take further trips.
            const int per = kWave / sl, need = sage_cdiv(k, per);
            if (sl == 64) {     // whole 1-KiB rows, one neighbour per wave-instruction: graphs with little reuse (every row read once)
                if (need <= 8) launch_pipe<64, 8>(blocks, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
                else launch_pipe<64, 16>(blocks, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
            } else if (sl == 8) {
                if (need <= 2) launch_pipe<8, 2>(blocks, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
                else if (need <= 4) launch_pipe<8, 4>(blocks, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
                else launch_pipe<8, 8>(blocks, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
            } else if (sl == 32) {
                if (need <= 4) launch_pipe<32, 4>(blocks, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
                else launch_pipe<32, 8>(blocks, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
            } else {
                if (need <= 2) launch_pipe<16, 2>(blocks, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
                else if (need <= 4) launch_pipe<16, 4>(blocks, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
                else launch_pipe<16, 8>(blocks, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
            }
            SAGE_CHECK_LAUNCH("gather_mean_sliced_pipe_kernel");
            return SAGE_OK;
        }
        if (sl == 32)
            SAGE_LAUNCH_G(gather_mean_sliced_kernel<32>, table, (int)table_rows, ld, dim, nbr, cnt, k,
                               n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
        else if (sl == 8)
            SAGE_LAUNCH_G(gather_mean_sliced_kernel<8>, table, (int)table_rows, ld, dim, nbr, cnt, k,
                               n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
        else
            SAGE_LAUNCH_G(gather_mean_sliced_kernel<16>, table, (int)table_rows, ld, dim, nbr, cnt, k,
                               n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off, nslice, slice_stride, act);
        SAGE_CHECK_LAUNCH("gather_mean_sliced_kernel");
        return SAGE_OK;
    }
    if (act != SAGE_ACT_NONE) { sage_set_error("gather_mean: an activation in the epilogue exists in the column-sliced forms only"); return SAGE_EUNSUPPORTED; }
    const int blocks = min(sage_cdiv(n, 4), kNumCU * 8);
    const bool vec4 = (dim % 4 == 0) && (ld % 4 == 0) && (ldo % 4 == 0) && sage_aligned(table, 16) && sage_aligned(out, 16);
    if (vec4)
        hipLaunchKernelGGL(gather_mean_kernel<4>, dim3(blocks), dim3(256), 0, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n,
                           n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off);
    else
        hipLaunchKernelGGL(gather_mean_kernel<1>, dim3(blocks), dim3(256), 0, st, table, (int)table_rows, ld, dim, nbr, cnt, k, n,
                           n_dev, slot_rows, self_row, any_nonempty, out, ldo, n_off);
    SAGE_CHECK_LAUNCH("gather_mean_kernel");
    return SAGE_OK;
}

extern "C" int sage_gather_mean(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                                const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                                const int32_t* self_row, const int32_t* any_nonempty, float* out, int64_t ldo,
                                sage_stream_t stream) {
    SAGE_REQUIRE(table && nbr && cnt && out, "gather_mean: NULL array");
    SAGE_REQUIRE(n >= 0 && k >= 1, "gather_mean: n = %d, k = %d", n, k);
    SAGE_REQUIRE(dim >= 1 && ld >= dim && ldo >= dim, "gather_mean: dim = %d, ld = %lld, ldo = %lld", dim, (long long)ld, (long long)ldo);
    SAGE_REQUIRE(table_rows >= 1 && table_rows < (1ll << 31), "gather_mean: table_rows = %lld", (long long)table_rows);
    return sage_launch_gather_mean(table, table_rows, ld, dim, nbr, cnt, k, n, n_dev, slot_rows, self_row, any_nonempty, out, ldo, 0,
                                   (hipStream_t)stream);
}
