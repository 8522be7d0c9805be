// Internal (non-ABI) launchers shared between translation units of libsage355.
#pragma once
#include "sage_common.h"

// forward2 plumbing shared by the launchers -------------------------------------------------
struct sage_resolve_t {          // see ResolveJob in sage_sample.hip
    const int32_t* slots; int32_t* rows_out; int32_t n_slots;
    const int32_t* self_slots; int32_t* self_rows_out; int32_t n_self;
    const int32_t* hash_rows; int32_t* hash_keys;
};
// The LAST kernel of a forward: its last-finishing block zeroes the forward's device counters
// (counters[0..6]; counters[7] is the ticket) and advances the batch-queue cursor, so the next
// forward needs no memset / reset / advance launches.
struct sage_finish_t {
    int32_t* counters;           // nullable = no finish duty
    int32_t* cursor;             // nullable
};
// Device-side row count = min(*n_dev + n_off, n).

int sage_launch_sample(const int64_t* rowptr, const int32_t* col, const int32_t* nodes, int32_t n, const int32_t* n_dev,
                       int32_t k, uint64_t seed, uint32_t tag, int32_t tag_self_rows, uint32_t tag_self,
                       int32_t* nbr, int32_t* cnt, int32_t* any_nonempty, const sage_frontier_t* frontier,
                       int32_t insert_self, int32_t* nbr_slot, int32_t* self_slot, const sage_model_t* queue_model,
                       int nodes_from_batch, int32_t* nodes_copy, int32_t n_off, int32_t frontier_row_off,
                       const sage_resolve_t* resolve, hipStream_t st);

int sage_launch_gather_mean(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                            const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                            const int32_t* self_row, const int32_t* any_nonempty, float* out, int64_t ldo, int32_t n_off,
                            hipStream_t st);
bool sage_gather_is_sliced(int32_t dim, int64_t ld, int64_t ldo, const float* table, const float* out, int32_t n, int32_t k);

int sage_launch_linear_act(const float* self_tab, int64_t ld_self, const int32_t* self_index, const float* agg, int64_t ld_agg,
                           int32_t dim, const float* weight, int64_t ldw, int32_t out_dim, int32_t act, int32_t n,
                           const int32_t* n_dev, float* out, int64_t ldo, int32_t n_off, sage_finish_t fin, hipStream_t st);

// Fused layer (sage_fused.hip).  Returns SAGE_EUNSUPPORTED when no instantiation fits.
int sage_launch_layer_fused(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                            const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                            const int32_t* self_row, const int32_t* any_nonempty, int32_t concat, const int32_t* self_index,
                            const float* weight, int64_t ldw, int32_t out_dim, int32_t act, float* out, int64_t ldo,
                            int32_t n_off, sage_finish_t fin, hipStream_t st);
bool sage_layer_fused_supported(int32_t dim, int32_t out_dim, int32_t concat);
int sage_launch_layer_dense(const float* agg, int64_t ld_agg, int32_t dim, int32_t n, const int32_t* n_dev, int32_t concat,
                            const float* self_tab, int64_t ld_self, int64_t self_rows, const int32_t* self_index,
                            const int32_t* cnt, const int32_t* any_nonempty,
                            const float* weight, int64_t ldw, int32_t out_dim, int32_t act, float* out, int64_t ldo, int32_t n_off,
                            sage_finish_t fin, hipStream_t st);

#ifdef __HIPCC__
// Called by EVERY block of the forward's last kernel, after its last use of the counters.
__device__ inline void sage_finish_block(const sage_finish_t& fin, int total_blocks) {
    if (!fin.counters) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&fin.counters[7], 1) == total_blocks - 1) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                fin.counters[8 + i] = fin.counters[i];    // read-back copy (tests, byte counting)
                fin.counters[i] = 0;
            }
            fin.counters[7] = 0;
            if (fin.cursor) *fin.cursor += 1;
        }
    }
}
#endif
