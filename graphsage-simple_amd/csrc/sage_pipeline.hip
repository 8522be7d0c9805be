// One launch for two independent pieces of work of consecutive queued batches: the layer-1 gather of batch i
// (HBM/fabric-bound, 46 us) and the outer-hop sample of batch i+1 (a chain of dependent round trips that takes 10 us
// alone and 25-30 us while a gather saturates the fabric).  The sampler's blocks come first in the grid, so they are
// resident from the start, and the gather gives up two of its eight blocks per CU for them; nothing in either body
// knows about the other (sage_sample_body.h, sage_gather_body.h take their block index as an argument).  Saves one
// graph node per forward and takes the outer sampler off the critical path.  Used by sage_forward2_gather_sample
// (two workspaces: the sample of batch i+1 must not touch the buffers batch i is still being computed from).
#include "sage_gather_body.h"
#include "sage_sample_body.h"

namespace {

using namespace sage_sample_detail;
using sage_gather_detail::gather_sliced_block;

struct GatherArgs {
    const float* table; int table_rows; int64_t ld; int dim;
    const int32_t* nbr; const int32_t* cnt; int k; int n; const int32_t* n_dev;
    const int32_t* self_row; const int32_t* any_nonempty; float* out; int64_t ldo; int n_off; int nslice;
};
struct SampleArgs {
    const int64_t* rowptr; const int32_t* col; int n; int k; uint32_t tag;
    int32_t* nbr; int32_t* cnt; int32_t* any_nonempty; FrontierDev f; int insert_self; int32_t* nbr_slot; int32_t* self_slot;
    BatchSrc bs;
};

template <int SL, int G>
__global__ __launch_bounds__(256) void gather_plus_sample_kernel(const GatherArgs ga, const SampleArgs sa, const int nsb) {
    if ((int)blockIdx.x < nsb) {
        sample_block<G, 256, true, true>(sa.rowptr, sa.col, nullptr, sa.n, nullptr, sa.k, 0u, 0u, sa.tag, 0, sa.tag, nullptr, nullptr, sa.nbr,
                                         sa.cnt, sa.any_nonempty, sa.f, sa.insert_self, sa.nbr_slot, sa.self_slot, sa.bs, 0, ResolveJob{},
                                         (int)blockIdx.x, nsb);
    } else {
        gather_sliced_block<SL>(ga.table, ga.table_rows, ga.ld, ga.dim, ga.nbr, ga.cnt, ga.k, ga.n, ga.n_dev, nullptr, ga.self_row,
                                ga.any_nonempty, ga.out, ga.ldo, ga.n_off, ga.nslice, (int)blockIdx.x - nsb, (int)gridDim.x - nsb);
    }
}

template <int SL, int G>
void launch(const GatherArgs& ga, const SampleArgs& sa, int nsb, int ngb, hipStream_t st) {
    hipLaunchKernelGGL((gather_plus_sample_kernel<SL, G>), dim3(nsb + ngb), dim3(256), 0, st, ga, sa, nsb);
}

template <int SL>
void launch_by_fanout(int k, const GatherArgs& ga, const SampleArgs& sa, int ngb, hipStream_t st) {
    auto nsb = [&](int g) { return sage_cdiv(sage_cdiv(sa.n, 256 / g), 8) * 8; };   // multiple of 8: the gather's slice -> XCD map is kept
    if (k <= 8) launch<SL, 8>(ga, sa, nsb(8), ngb, st);
    else if (k <= 16) launch<SL, 16>(ga, sa, nsb(16), ngb, st);
    else if (k <= 32) launch<SL, 32>(ga, sa, nsb(32), ngb, st);
    else launch<SL, 64>(ga, sa, nsb(64), ngb, st);
}

}  // namespace

// gather: as sage_launch_gather_mean (sliced form only); sample: as the outer hop of forward2 (queued, frontier insert).
// Returns SAGE_EUNSUPPORTED when the gather would not take the sliced kernel: the caller then launches the two separately.
int sage_launch_gather_plus_sample(const float* table, int64_t table_rows, int64_t ld, int32_t dim, const int32_t* nbr1,
                                   const int32_t* cnt1, int32_t k1, int32_t n1, const int32_t* n1_dev, const int32_t* self_row,
                                   const int32_t* any1, float* agg, int64_t ldo, int32_t n_off,
                                   const int64_t* rowptr2, const int32_t* col2, int32_t batch, int32_t k2, uint32_t tag,
                                   int32_t* nbr2, int32_t* cnt2, int32_t* any2, const sage_frontier_t* frontier, int32_t insert_self,
                                   int32_t* slot2, int32_t* self_slot2, const sage_model_t* qm, int32_t* nodes_copy,
                                   int32_t frontier_row_off, int32_t cursor_off, uint64_t* key_slot, hipStream_t st) {
    if (!qm || !qm->queue || !sage_gather_is_sliced(dim, ld, ldo, table, agg, n1, k1)) return SAGE_EUNSUPPORTED;
    const int sl = (dim <= 128 && dim % 64 != 0) ? 32 : 16;
    const int nslice = sage_cdiv(dim, sl * 4);
    const int ngb = nslice * (kNumCU * 6 / nslice);          // 6 gather blocks per CU + 2 sampler blocks = 32 waves
    const GatherArgs ga{table, (int)table_rows, ld, dim, nbr1, cnt1, k1, n1, n1_dev, self_row, any1, agg, ldo, n_off, nslice};
    const FrontierDev fd{frontier->keys, frontier->rows, (uint32_t)frontier->capacity - 1u, frontier->nodes, frontier->count,
                         frontier->max_nodes, frontier_row_off};
    const BatchSrc bs{qm->queue, qm->queue_cursor, qm->queue_len, 1, nodes_copy, cursor_off, key_slot, qm->seed_map, (int)qm->num_nodes};
    const SampleArgs sa{rowptr2, col2, batch, k2, tag, nbr2, cnt2, any2, fd, insert_self, slot2, self_slot2, bs};
    if (sl == 32) launch_by_fanout<32>(k2, ga, sa, ngb, st);
    else launch_by_fanout<16>(k2, ga, sa, ngb, st);
    SAGE_CHECK_LAUNCH("gather_plus_sample_kernel");
    return SAGE_OK;
}
