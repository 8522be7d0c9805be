// Internal (non-ABI) launchers shared between translation units of libsage355.
#pragma once
#include "sage_common.h"

// forward2 plumbing shared by the launchers -------------------------------------------------
struct sage_resolve_t {          // see ResolveJob in sage_sample.hip
    const int32_t* slots; int32_t* rows_out; int32_t n_slots;
    const int32_t* self_slots; int32_t* self_rows_out; int32_t n_self;
    const int32_t* hash_rows; int32_t* hash_keys;
};
// With the fused sampler the layer-2 kernel receives hash SLOTS (nbr, self_row) and turns them into frontier rows itself
// (slot_rows = the hash's rows array); on the way it wipes the keys it reads (which leaves the table clean for the next forward) and
// leaves the rows in rows_out / self_rows_out for read-back (tests, the training step's backward).
struct sage_slot_resolve_t {
    int32_t* wipe_keys;          // [capacity] hash keys; keys[slot] := -1 for every slot read
    int32_t* rows_out;           // [n, k]  frontier row of every neighbour slot (-1 = padding)
    int32_t* self_rows_out;      // nullable [n]
};
// The LAST kernel of a forward: its last-finishing block zeroes the forward's device counters
// (counters[0..6]; counters[7] is the ticket) and advances the batch-queue cursor, so the next
// forward needs no memset / reset / advance launches.
struct sage_finish_t {
    int32_t* counters;           // nullable = no finish duty
    int32_t* cursor;             // nullable
};
// Device-side row count = min(*n_dev + n_off, n).

int sage_launch_sample(const int64_t* rowptr, const int32_t* col, int64_t num_nodes, const int32_t* nodes, int32_t n, const int32_t* n_dev,
                       int32_t k, uint64_t seed, uint32_t tag, int32_t tag_self_rows, uint32_t tag_self,
                       int32_t* nbr, int32_t* cnt, int32_t* any_nonempty, const sage_frontier_t* frontier,
                       int32_t insert_self, int32_t* nbr_slot, int32_t* self_slot, const sage_model_t* queue_model,
                       int nodes_from_batch, int32_t* nodes_copy, int32_t n_off, int32_t frontier_row_off,
                       const sage_resolve_t* resolve, int32_t cursor_off, uint64_t* key_slot, const int32_t* seed_map, hipStream_t st);

int sage_launch_sample_fused(const sage_model_t* m, const int32_t* seeds, int32_t batch, uint64_t seed, int32_t* nbr2, int32_t* cnt2,
                             int32_t* any2, const sage_frontier_t* frontier, int32_t insert_self, int32_t* nbr_slot, int32_t* self_slot,
                             int queued, int32_t* nodes_copy, int32_t frontier_row_off, int32_t* nbr1, int32_t* cnt1, int32_t* any1,
                             int32_t seed_rows, hipStream_t st);

int sage_launch_gather_mean(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                            const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                            const int32_t* self_row, const int32_t* any_nonempty, float* out, int64_t ldo, int32_t n_off,
                            hipStream_t st, int64_t slice_stride = 0, int act = SAGE_ACT_NONE /* column-sliced forms only */);
bool sage_layer_dense_supported(int32_t dim, int32_t out_dim);
bool sage_gather_is_sliced(int32_t dim, int64_t ld, int64_t ldo, const float* table, const float* out, int32_t n, int32_t k);

int sage_launch_linear_act(const float* self_tab, int64_t ld_self, const int32_t* self_index, const float* agg, int64_t ld_agg,
                           int32_t dim, const float* weight, int64_t ldw, int32_t out_dim, int32_t act, int32_t n,
                           const int32_t* n_dev, float* out, int64_t ldo, int32_t n_off, sage_finish_t fin, hipStream_t st);

// Fused layer (sage_fused.hip).  Returns SAGE_EUNSUPPORTED when no instantiation fits.
int sage_launch_layer_fused(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                            const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                            const int32_t* self_row, const int32_t* any_nonempty, int32_t concat, const int32_t* self_index,
                            const float* weight, int64_t ldw, int32_t out_dim, int32_t act, float* out, int64_t ldo,
                            int32_t n_off, sage_finish_t fin, hipStream_t st, const sage_slot_resolve_t* resolve = nullptr);
bool sage_layer_fused_supported(int32_t dim, int32_t out_dim, int32_t concat);
int sage_launch_layer_dense(const float* agg, int64_t ld_agg, int32_t dim, int32_t n, const int32_t* n_dev, int32_t concat,
                            const float* self_tab, int64_t ld_self, int64_t self_rows, const int32_t* self_index,
                            const int32_t* cnt, const int32_t* any_nonempty,
                            const float* weight, int64_t ldw, int32_t out_dim, int32_t act, float* out, int64_t ldo, int32_t n_off,
                            sage_finish_t fin, const void* weight_prepared, hipStream_t st);

// Measurement hook (sage_gather.hip): set by the thread that is about to launch the layer-1 gather, cleared right after.
struct sage_ext_launch_t { void* start; void* stop; };
extern thread_local const sage_ext_launch_t* sage_ext_launch;

// Tail event (round 4).  A stage's hand-off event can ride on the stage's LAST kernel as that dispatch's own completion signal
// (hipExtLaunchKernel stopEvent) instead of being a packet of its own behind the kernel (hipEventRecord): one barrier packet less between
// two kernels of a role stream (experiments/r04/handoff.hip: 7.3 against 8.5 us per hand-off; in the pipeline the record was one of the two
// packets that separate consecutive kernels of a stream).  One-shot and thread-local: whoever is about to call the launcher of a stage's
// last kernel sets it; the first launch made through SAGE_LAUNCH_TAIL consumes it; a launcher that does not know it leaves it set, and the
// setter then records the event explicitly (sage_forward.hip).  Never while a stream is capturing.
extern thread_local void* sage_tail_event;
#ifdef __HIPCC__
#include <hip/hip_ext.h>
#define SAGE_LAUNCH_TAIL(kernel, grid, block, lds, st, ...)                                                               \
    do {                                                                                                                  \
        if (sage_tail_event) {                                                                                            \
            hipEvent_t tail_ = (hipEvent_t)sage_tail_event;                                                               \
            sage_tail_event = nullptr;                                                                                    \
            hipExtLaunchKernelGGL((kernel), (grid), (block), (lds), (st), nullptr, tail_, 0u, __VA_ARGS__);               \
        } else {                                                                                                          \
            hipLaunchKernelGGL((kernel), (grid), (block), (lds), (st), __VA_ARGS__);                                      \
        }                                                                                                                 \
    } while (0)
#endif

// The launches of one forward, by stage (sage_pipe.hip enqueues each stage on its role stream)
#define SAGE_STAGE_SAMPLE_OUTER 1
#define SAGE_STAGE_SAMPLE_INNER 2
#define SAGE_STAGE_GATHER1      4
#define SAGE_STAGE_CONTRACT1    8     /* whole layer 1 when it is a one-launch layer */
#define SAGE_STAGE_LAYER2       16
#define SAGE_STAGE_ALL          31
int sage_forward2_launch_stages(const sage_model_t* m, void* workspace, size_t workspace_bytes, const int32_t* seeds, int32_t batch,
                                uint64_t seed, float* out, int64_t ldo, int32_t stages, hipStream_t stream, void* tail_event = nullptr);

// Launch-shape tunables, read ONCE from the environment (A/B runs on one box without rebuilding; defaults are the
// measured optima recorded in DESIGN.md).  Every value is clamped to a safe range.
struct sage_tunables_t {
    int gather_blocks_per_cu;     // SAGE_G_PER_CU        sliced gather: 256-thread blocks per CU (1..8), default 6
    int gather_slice_lanes;       // SAGE_G_SLICE_LANES   0 = by row width (16 lanes = 256-B slices; 32 for narrow odd rows), or 8 / 16 / 32 / 64 (64: variant 1 only)
    int gather_rows_in_flight;    // SAGE_G_ROWS          pipelined gather: rows of a wave in flight together (1 / 2 / 4), default 1
    int gather_trip;              // SAGE_G_TRIP          rows form: neighbours of a row requested per trip (8 / 16), default 16
    int gather_variant;           // SAGE_G_VARIANT       0 = three-trip rows, 1 = rows software-pipelined (default), 2 = one row per lane group
    int gather_variant_sliced;    // SAGE_G_VARIANT_SM    the variant used with a slice-major table (sage_model_t.table_sliced): 2 (default) / 1 / 0
    int dense_blocks;             // SAGE_DENSE_BLOCKS    split-bf16 contraction: persistent 512-thread blocks (32..512), default 224
    int bwd_blocks;               // SAGE_BWD_BLOCKS      weight-gradient GEMM: blocks over (tiles x K splits), default 2 per CU (every split adds its tile with fp32 atomics)
    int bwd_direct_blocks;        // SAGE_BWD_DIRECT_BLOCKS  reproducible weight gradient: row ranges = 512-thread blocks = partial tiles (16..1024), default 256
    int outer_threads;            // SAGE_SO_THREADS      outer-hop sampler block size (256 / 512 / 1024), default 512 (1024 until round 3)
    int tile16_grid;              // SAGE_T16_GRID        layer-2 tile16 kernel: max blocks (64..1024), default 512
    int sample_fused;             // SAGE_SAMPLE_FUSED    1: both hops in one launch when layer 2 is a one-launch layer; 0 (default): two launches
                                  //                      (measured: 24.3 us fused vs 10.3 + 11.4: the inner hop of a block's own winners is
                                  //                      three dependent rounds on 128-256 blocks instead of one round on 1500; pipeline 66.8 vs 66.2 us)
    int tile16_waves;             // SAGE_T16_WAVES       layer-2 tile16 kernel: 16 (1024-thread blocks) or 8 (512-thread blocks, default: 1.5 us per forward in the pipeline)
};
// n_words 32-bit words := v, as a kernel (hipMemsetAsync misbehaves inside replayed hipGraphs on ROCm 7.2: sage_api.hip)
int sage_fill_u32(void* p, uint32_t v, size_t n_words, hipStream_t st);
const sage_tunables_t& sage_tunables();

// Narrowest layer that takes the split form (column-sliced gather + dense contraction) instead of the one-launch layer.
#ifndef SAGE_SPLIT_MIN_DIM
#define SAGE_SPLIT_MIN_DIM 64
#endif

#ifdef __HIPCC__
// Finish duty of the forward's last kernel: the block that draws the last ticket zeroes the counters (keeping a
// read-back copy for tests / byte counting) and advances the batch-queue cursor.  Split in two so that the loads
// the duty needs are in flight while the block does its real work: sage_finish_begin() at kernel start (thread 0;
// the counters are final before this kernel starts), sage_finish_block() by EVERY block after its last use of them.
struct sage_finish_regs { int32_t c[7]; int32_t cur; };

__device__ inline void sage_finish_begin(const sage_finish_t& fin, sage_finish_regs& r) {
    if (!fin.counters || threadIdx.x != 0) return;
#pragma unroll
    for (int i = 0; i < 7; ++i) r.c[i] = fin.counters[i];
    r.cur = fin.cursor ? *fin.cursor : 0;
}

__device__ inline void sage_finish_block(const sage_finish_t& fin, int total_blocks, const sage_finish_regs& r) {
    if (!fin.counters) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        // No __threadfence() here: on gfx950 it is `buffer_wbl2 sc1` + `buffer_inv sc1` -- a write-back and an
        // invalidate of the XCD's whole L2, once per block, which also throws out the other batch's cached hub rows
        // (measured: layer 2 16.7 -> 10.7 us alone, forward 98 -> 90 us with two batches in flight).
        // It is not needed: every block has consumed the counters it read (its row count decides its whole loop)
        // before the barrier above, the ticket is taken after that barrier, and the block that draws the last ticket
        // is the only writer; the kernel boundary publishes its stores to the next launch.
        if (__hip_atomic_fetch_add(&fin.counters[7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total_blocks - 1) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                fin.counters[8 + i] = r.c[i];             // read-back copy (tests, byte counting)
                fin.counters[i] = 0;
            }
            fin.counters[7] = 0;
            if (fin.cursor) *fin.cursor = r.cur + 1;
        }
    }
}

// one-call form (kernels whose blocks have nothing to overlap the loads with)
__device__ inline void sage_finish_block(const sage_finish_t& fin, int total_blocks) {
    sage_finish_regs r;
    sage_finish_begin(fin, r);
    sage_finish_block(fin, total_blocks, r);
}
#endif
