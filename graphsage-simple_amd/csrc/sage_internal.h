// Internal (non-ABI) launchers shared between translation units of libsage355.
#pragma once
#include "sage_common.h"

int sage_launch_sample(const int64_t* rowptr, const int32_t* col, const int32_t* nodes, int32_t n, const int32_t* n_dev,
                       int32_t k, uint64_t seed, uint32_t tag, int32_t tag_self_rows, uint32_t tag_self,
                       int32_t* nbr, int32_t* cnt, int32_t* any_nonempty, const sage_frontier_t* frontier,
                       int32_t insert_self, int32_t* nbr_slot, int32_t* self_slot, hipStream_t st);

int sage_launch_gather_mean(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                            const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                            const int32_t* self_row, const int32_t* any_nonempty, float* out, int64_t ldo, hipStream_t st);

int sage_launch_linear_act(const float* self_tab, int64_t ld_self, const int32_t* self_index, const float* agg, int64_t ld_agg,
                           int32_t dim, const float* weight, int64_t ldw, int32_t out_dim, int32_t act, int32_t n,
                           const int32_t* n_dev, float* out, int64_t ldo, hipStream_t st);

// Fused layer (sage_fused.hip).  Returns SAGE_EUNSUPPORTED when no instantiation fits.
int sage_launch_layer_fused(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr,
                            const int32_t* cnt, int32_t k, int32_t n, const int32_t* n_dev, const int32_t* slot_rows,
                            const int32_t* self_row, const int32_t* any_nonempty, int32_t concat, const int32_t* self_index,
                            const float* weight, int64_t ldw, int32_t out_dim, int32_t act, float* out, int64_t ldo,
                            hipStream_t st);
bool sage_layer_fused_supported(int32_t dim, int32_t out_dim, int32_t concat);
