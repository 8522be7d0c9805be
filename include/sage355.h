/*
 * sage355.h -- C ABI of libsage355.so: the GraphSAGE sample -> gather -> masked
 * mean -> (concat) -> W.x -> act hot path as hand-written HIP kernels for
 * MI355X (gfx950).
 *
 * The reference (zjzijielu/graphsage-simple) is pure Python and has no FFI;
 * the "operators" of its hot path are the statement groups of
 *   graphsage/aggregators.py:34-76   MeanAggregator.forward
 *   graphsage/encoders.py:40-62      Encoder.forward
 * Each entry point below replaces one such group and cites it.  A maintainer
 * binds them with ctypes (INTEGRATION.md shows the stub); this repo's own host
 * side (graphsage-simple_amd/sage355/) does exactly that.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer (HBM) unless the name ends in _host.
 *    The library never allocates, frees or synchronises: the caller owns all
 *    memory (so a caching allocator and hipGraph capture both work).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *  - Node ids are int32 (N < 2^31), CSR row pointers int64, features fp32
 *    row-major with an explicit leading dimension in ELEMENTS.
 *  - Row counts come in pairs (n, n_dev): `n` is the host-side upper bound used
 *    to size the launch; if `n_dev` is non-NULL the kernels read the actual
 *    count from it (min(*n_dev, n)), so a frontier whose size is only known on
 *    the device needs no host round trip.
 *  - Return value: SAGE_OK or a negative SAGE_E* code; sage_last_error() gives
 *    the message for the calling thread.  Arguments are validated on the host
 *    BEFORE any launch (an out-of-range kernel access can reset the whole
 *    node), so a call either enqueues all of its kernels or none.
 */
#ifndef SAGE355_H
#define SAGE355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAGE_ABI_VERSION 6

#define SAGE_OK            0
#define SAGE_EINVAL       -1   /* bad argument (NULL, size, alignment, range) */
#define SAGE_EUNSUPPORTED -2   /* valid request this build has no kernel for  */
#define SAGE_ELAUNCH      -3   /* hipLaunch / runtime error                   */
#define SAGE_ENOSPACE     -4   /* workspace too small                         */

#define SAGE_ACT_RELU    0     /* encoders.py:61; NaN propagates like torch.relu */
#define SAGE_ACT_SIGMOID 1     /* encoders.py:58-59 (node_degree/shared/pagerank) */
#define SAGE_ACT_NONE    2

#define SAGE_MAX_FANOUT  64    /* device sampler limit on k (BASELINE uses <= 25) */

/* RNG stream tags: which sampling call a draw belongs to (see oracle/sampler_ref.c). */
#define SAGE_TAG_INNER      1  /* layer-1 samples of the frontier            */
#define SAGE_TAG_OUTER      2  /* layer-2 samples of the seeds               */
#define SAGE_TAG_INNER_SELF 3  /* concat encoder: layer-1 samples of the seeds (2nd enc1 call) */

typedef void* sage_stream_t;

int         sage_abi_version(void);
const char* sage_last_error(void);
/* Name of the code-object architecture the library was built for ("gfx950"). */
const char* sage_build_arch(void);

/* ---------------------------------------------------------------------------
 * Frontier: the set of distinct ids one hop produces, plus id -> row lookup.
 * Replaces `unique_nodes_list = list(set.union(*samp_neighs))` and the
 * `unique_nodes` dict (aggregators.py:52-53).  Open-addressing hash in HBM:
 * keys[slot] = node id (or -1), rows[slot] = position of that id in nodes[].
 * The ORDER of nodes[] is arbitrary (as a Python set's is); everything
 * computed from it is order independent.
 * ------------------------------------------------------------------------- */
typedef struct {
    int32_t* keys;        /* [capacity]  -1 = empty; reset with sage_frontier_reset */
    int32_t* rows;        /* [capacity]  row of keys[slot] in nodes[]               */
    int32_t  capacity;    /* power of two, >= 2 * max distinct ids                   */
    int32_t* nodes;       /* [max_nodes] the distinct ids, arbitrary order           */
    int32_t* count;       /* [1] device counter: number of rows in nodes[] so far    */
    int32_t  max_nodes;
} sage_frontier_t;

/* keys[] := -1, *count := first_row (rows below first_row are reserved by the caller). */
int sage_frontier_reset(const sage_frontier_t* f, int32_t first_row, sage_stream_t stream);

/* ---------------------------------------------------------------------------
 * sage_sample_neighbors -- encoders.py:47 (adjacency fetch) + aggregators.py:42-48
 * (fixed-fanout sample) [+ aggregators.py:52-53 when `frontier` is given].
 *
 * For node v = nodes[r], deg = rowptr[v+1]-rowptr[v]:
 *   deg >  k : k DISTINCT uniform positions of the CSR row (Floyd's subset
 *              algorithm, O(k); Philox4x32-10 keyed by `seed`, counter
 *              (v, tag, draw/4)) -> cnt[r] = k
 *   deg <= k : the whole row, in CSR order              -> cnt[r] = deg
 * nbr[r*k + j], j < cnt[r], are global node ids; entries j >= cnt[r] are -1.
 * The draw for (seed, tag, v) does not depend on r, n or the launch shape.
 *
 * frontier != NULL: every sampled id (and v itself when `insert_self`) is
 * inserted; nbr_slot[r*k+j] receives its hash slot and self_slot[r] the slot of
 * v.  rows[slot] is valid once this call's kernels have completed (stream
 * order), so a consumer reads row = frontier->rows[nbr_slot[e]].
 * any_nonempty (nullable): set to 1 if any cnt[r] > 0 (reference NaN rule).
 * ------------------------------------------------------------------------- */
int sage_sample_neighbors(const int64_t* rowptr, const int32_t* col, int64_t num_nodes,
                          const int32_t* nodes, int32_t n, const int32_t* n_dev,
                          int32_t k, uint64_t seed, uint32_t tag,
                          int32_t* nbr, int32_t* cnt, int32_t* any_nonempty,
                          const sage_frontier_t* frontier, int32_t insert_self,
                          int32_t* nbr_slot, int32_t* self_slot,
                          sage_stream_t stream);

/* Insert already-sampled ids (e.g. sets injected by a caller, the reference's
 * num_sample=None path) into a frontier.  Same outputs as above. */
int sage_frontier_insert(const int32_t* nbr, const int32_t* cnt, int32_t k,
                         const int32_t* self_nodes /* nullable */,
                         int32_t n, const int32_t* n_dev,
                         const sage_frontier_t* frontier,
                         int32_t* nbr_slot, int32_t* self_slot, sage_stream_t stream);

/* ---------------------------------------------------------------------------
 * sage_gather_mean -- aggregators.py:54-74: mask build, row normalise, feature
 * fetch and mask.mm(embed_matrix), without the dense mask:
 *     out[r, :] = (1/c) * sum_{j<cnt[r]} table[row(nbr[r*k+j]), :]
 * row(x) = x, or slot_rows[x] when `slot_rows` is given (nbr then holds hash
 * slots).  self_row (nullable): aggregators.py:50-51 intended semantics -- the
 * node's own row joins the mean unless it is already among the sampled ones.
 * cnt[r]==0 (and no self row): NaN row if *any_nonempty != 0 (the reference's
 * 0/0 inside a mixed batch), else zeros (its all-empty batch); with
 * any_nonempty == NULL the row is zeros.
 * ------------------------------------------------------------------------- */
int sage_gather_mean(const float* table, int64_t table_rows, int64_t ld, int32_t dim,
                     const int32_t* nbr, const int32_t* cnt, int32_t k,
                     int32_t n, const int32_t* n_dev,
                     const int32_t* slot_rows, const int32_t* self_row,
                     const int32_t* any_nonempty,
                     float* out, int64_t ldo, sage_stream_t stream);

/* ---------------------------------------------------------------------------
 * sage_linear_act -- encoders.py:49-62 without materialising the concat:
 *     out[r, :] = act( W[:, 0:ds] . self(r) + W[:, ds:ds+dim] . agg[r, :] )
 * self(r) = self_tab[self_index ? self_index[r] : r, 0:dim]; ds = dim when
 * self_tab != NULL (concat encoder, gcn=False) else 0 (gcn=True).
 * W is [out_dim, ds+dim] row-major (ldw), exactly the reference Parameter.
 * out is [n, out_dim] (the module returns its transpose view, encoders.py:62).
 * fp32 MFMA (v_mfma_f32_32x32x2_f32): exact-fp32 products, fp32 accumulate.
 * ------------------------------------------------------------------------- */
int sage_linear_act(const float* self_tab, int64_t ld_self, const int32_t* self_index,
                    const float* agg, int64_t ld_agg, int32_t dim,
                    const float* weight, int64_t ldw, int32_t out_dim, int32_t act,
                    int32_t n, const int32_t* n_dev,
                    float* out, int64_t ldo, sage_stream_t stream);

/* ---------------------------------------------------------------------------
 * sage_layer_forward -- one Encoder.forward (encoders.py:47-62) in ONE launch:
 * gather-mean rows are staged through an LDS tile and contracted with W by
 * fp32 MFMA without the [n, dim] round trip through HBM.  Arguments as the two
 * calls above.  SAGE_EUNSUPPORTED if (dim, out_dim) has no fused kernel; the
 * caller then uses the two-launch form.
 * ------------------------------------------------------------------------- */
int sage_layer_forward(const float* table, int64_t table_rows, int64_t ld, int32_t dim,
                       const int32_t* nbr, const int32_t* cnt, int32_t k,
                       int32_t n, const int32_t* n_dev,
                       const int32_t* slot_rows, const int32_t* self_row,
                       const int32_t* any_nonempty,
                       int32_t concat, const int32_t* self_index,
                       const float* weight, int64_t ldw, int32_t out_dim, int32_t act,
                       float* out, int64_t ldo, sage_stream_t stream);
int sage_layer_forward_supported(int32_t dim, int32_t out_dim, int32_t concat);

/* ---------------------------------------------------------------------------
 * Backward of the two operators (autograd of encoders.py:58-62 and
 * aggregators.py:60-74; the reference gets these from torch autograd through
 * mm / relu / cat / div).
 *
 * sage_linear_act_backward: dZ = grad_out * act'(out);
 *   grad_weight[out_dim, ds+dim] += dZ^T . [self | agg]   (ACCUMULATES: caller zeroes)
 *   grad_x[n, ds+dim]             = dZ . W                 (written; nullable)
 * sage_gather_mean_backward: grad_table[row(nbr[r,j]), :] += grad_agg[r, :] / c
 *   (fp32 atomics, caller zeroes grad_table; same slot_rows / self_row rules
 *   as the forward).
 * ------------------------------------------------------------------------- */
/* n_dev (nullable): the live row count on the device, min(*n_dev, n) rows take part (rows past it are neither read nor written). */
int sage_linear_act_backward(const float* self_tab, int64_t ld_self, const int32_t* self_index,
                             const float* agg, int64_t ld_agg, int32_t dim,
                             const float* weight, int64_t ldw, int32_t out_dim, int32_t act,
                             const float* out, int64_t ldo, const float* grad_out, int64_t ldg,
                             int32_t n, const int32_t* n_dev,
                             float* grad_weight, int64_t ldgw, float* grad_x, int64_t ldgx,
                             sage_stream_t stream);
int sage_gather_mean_backward(const float* grad_agg, int64_t ldg, int32_t dim,
                              const int32_t* nbr, const int32_t* cnt, int32_t k,
                              int32_t n, const int32_t* n_dev,
                              const int32_t* slot_rows, const int32_t* self_row,
                              float* grad_table, int64_t table_rows, int64_t ld,
                              sage_stream_t stream);

/* Reproducible forms (ABI 3).  torch autograd on the reference's CPU (model.py:249) gives the same bits run after run; the two
 * entry points above add in order of arrival (fp32 atomics).  These take a caller-owned workspace instead and are bitwise
 * reproducible for given inputs and launch tunables:
 *  - sage_linear_act_backward_ws: the reduction over the n rows is cut into row ranges, every range STORES its partial
 *    [out_dim, ds+dim] tile in the workspace, one kernel adds the partials in range order to grad_weight (still "+=": the
 *    caller zeroes).  grad_x as above.  Even widths / leading dimensions and 8-byte aligned arrays take the fast kernel (operands
 *    straight from HBM in MFMA register order); anything else the generic tile kernel, also through partials.
 *  - sage_gather_mean_backward_ws: an inverted index (expand -> stable radix sort by table row -> run heads) built per call in
 *    the workspace; every table row's terms are then summed in ascending (r, j) order and the row is STORED (zeros where
 *    nobody points): rows [0, min(*table_rows_dev, table_rows)) of grad_table are written, the caller zeroes nothing.
 *    Needs dim % 4 == 0, ld % 4 == 0, 16-byte aligned arrays.
 * The *_workspace_bytes queries are host-side arithmetic (0 = shape out of range / query failed); workspaces 256-byte aligned. */
size_t sage_linear_act_backward_workspace_bytes(int32_t n, int32_t dim, int32_t has_self, int32_t out_dim);
int sage_linear_act_backward_ws(const float* self_tab, int64_t ld_self, const int32_t* self_index,
                                const float* agg, int64_t ld_agg, int32_t dim,
                                const float* weight, int64_t ldw, int32_t out_dim, int32_t act,
                                const float* out, int64_t ldo, const float* grad_out, int64_t ldg,
                                int32_t n, const int32_t* n_dev,
                                float* grad_weight, int64_t ldgw, float* grad_x, int64_t ldgx,
                                const int32_t* row_order /* nullable */,
                                void* workspace, size_t workspace_bytes, sage_stream_t stream);
/* The frontier's rows are in arbitrary order (aggregators.py:52: so is a Python set), and the weight gradient is a sum over the
 * layer's rows: for the same bits run after run it has to add them in an order that does not depend on the layout.  sage_row_order
 * writes that order -- rows [0, first_row) as they are (the concat encoder's own seeds), then the live rows [first_row, *n_dev) by
 * ascending node id (distinct there), dead rows last -- and sage_linear_act_backward_ws(row_order = it) sums its k-th term from row
 * row_order[k] (grad_weight only; grad_x is per row).  nodes: int32[n], the layer's node ids (sage_ws_layout_t.s1_nodes). */
size_t sage_row_order_workspace_bytes(int32_t n);
int sage_row_order(const int32_t* nodes, int32_t n, const int32_t* n_dev, int32_t first_row, int32_t* order,
                   void* workspace, size_t workspace_bytes, sage_stream_t stream);
/* Layer-1 weight gradient of the TWO-LAYER stack (model.py:219-222 differentiated, model.py:249), summed over the outer samples
 * instead of over the layer-1 rows: grad_w1 [h1, d0 or 2*d0] += sum over the seeds r, in order, of
 *   (concat)  act1'(h1[r]) . grad_x2[r, 0:h1]  (x)  [table[s1_nodes[r]] | agg1[r]]                    (the seed's own layer-1 row)
 *   for j < cnt2[r] (and the self-loop row):  act1'(h1[t]) . grad_x2[r, off:off+h1] / c_r  (x)  [table[s1_nodes[t]] |] agg1[t],  t = row2[r, j]
 * (off = h1 for the concat encoder, else 0; c_r as in the forward).  Mathematically the chain mean-backward -> relu' -> dZ^T.X; in
 * this form no grad_h1 is scattered and the terms come in (seed, slot) order, which does not depend on the frontier's arbitrary row
 * order: bitwise reproducible without an inverted index or a canonical row order (csrc/sage_backward.hip).  grad_x2 is what
 * sage_linear_act_backward(_ws) returned for layer 2; row2 / cnt2 / self_row2 / h1 / agg1 / s1_nodes are the forward's
 * intermediates (sage_ws_layout_t).  Widths and leading dimensions multiples of 4, arrays 16-byte aligned. */
size_t sage_two_hop_grad_w1_workspace_bytes(int32_t batch, int32_t k2, int32_t d0, int32_t concat, int32_t h1);
int sage_two_hop_grad_w1(const float* grad_x2, int64_t ldgx,
                         const int32_t* row2, const int32_t* cnt2, int32_t k2, const int32_t* self_row2, int32_t batch,
                         const float* h1, int64_t ldh, int32_t h1_dim, int32_t act1,
                         const float* agg1, int64_t lda, int32_t d0,
                         int32_t concat, const float* table, int64_t table_ld, const int32_t* s1_nodes,
                         float* grad_w1, int64_t ldgw,
                         void* workspace, size_t workspace_bytes, sage_stream_t stream);
size_t sage_gather_mean_backward_workspace_bytes(int32_t n, int32_t k, int64_t table_rows);
int sage_gather_mean_backward_ws(const float* grad_agg, int64_t ldg, int32_t dim,
                                 const int32_t* nbr, const int32_t* cnt, int32_t k,
                                 int32_t n, const int32_t* n_dev,
                                 const int32_t* slot_rows, const int32_t* self_row,
                                 float* grad_table, int64_t table_rows, const int32_t* table_rows_dev, int64_t ld,
                                 void* workspace, size_t workspace_bytes, sage_stream_t stream);

/* ---------------------------------------------------------------------------
 * Two-layer forward: model.py:219-222 wiring of two Encoders, i.e. the
 * "2-hop forward" the headline metric counts.
 * ------------------------------------------------------------------------- */
/* A batch descriptor in DEVICE memory: which seeds to embed and the sampler key to use.
 * A ring of them + a device-side cursor lets a captured hipGraph of sage_forward2 be
 * replayed for batch after batch with NO per-replay host work: every kernel takes its
 * seeds pointer and key from queue[*cursor % queue_len], and the last node of the
 * forward advances the cursor. */
typedef struct {
    const int32_t* seeds;   /* [batch] device pointer */
    uint64_t       seed;    /* sampler key for this batch */
} sage_batch_t;

typedef struct {
    /* Each Encoder holds its own adjacency (encoders.py:21); model.py passes the
     * same dict to both.  Injecting pre-sampled sets (the reference's
     * num_sample=None switch, aggregators.py:47-48) = pointing a layer at a CSR
     * whose rows ARE the sampled sets, with k >= the longest row. */
    const int64_t* rowptr1;     /* [num_nodes+1]  enc1.adj_lists (inner hop)       */
    const int32_t* col1;
    const int64_t* rowptr2;     /* [num_nodes+1]  enc2.adj_lists (outer hop)       */
    const int32_t* col2;
    int64_t        num_nodes;
    const float*   table;       /* [num_nodes, d0] raw features (model.py:214-215) */
    int64_t        table_ld;
    int32_t        d0;
    const float*   w1;          /* [h1, d0 or 2*d0]   enc1.weight                  */
    int32_t        h1;
    const float*   w2;          /* [h2, h1 or 2*h1]   enc2.weight                  */
    int32_t        h2;
    int32_t        k1;          /* enc1.num_sample (inner hop)                     */
    int32_t        k2;          /* enc2.num_sample (outer hop, on the seeds)       */
    int32_t        concat;      /* 1: encoder gcn=False (self || agg); 0: gcn=True */
    int32_t        agg_self_loop;/* aggregator gcn flag (aggregators.py:50-51)     */
    int32_t        act1, act2;  /* SAGE_ACT_*                                     */
    int32_t        nan_empty;   /* 1: reference NaN rule for empty sets; 0: zeros  */
    int32_t        fused;       /* 1: one-launch layers when supported; 0: never   */
    int32_t        ws_batch;    /* batch size the workspace was laid out and initialised for (sage_forward2_init);
                                   forwards may use any batch <= ws_batch.  0 = the call's own batch.          */
    /* optional batch queue (all NULL/0 = take `seeds` and `seed` from the call arguments) */
    const sage_batch_t* queue;  /* [queue_len] descriptors in device memory            */
    int32_t        queue_len;
    int32_t*       queue_cursor;/* [1] device counter, advanced by one per forward      */
    /* optional (ABI 2): enc1.weight already split into bf16 planes by sage_prepare_weights (NULL = the kernel
     * splits its slice itself at every launch).  Must be refreshed whenever w1 changes (an optimizer step). */
    const void*    w1_prepared;
    /* optional (ABI 2): caller id -> internal id, int32[num_caller_ids].  An engine that keeps graph and table in an
     * order of its own (sage355.engine: rows sorted by descending degree, so that the most-gathered feature rows are
     * neighbours in HBM) translates every seed with it INSIDE the outer-hop kernel; NULL = ids are used as they are. */
    const int32_t* seed_map;
    /* optional (ABI 3): a second copy of the table in SLICE-MAJOR order, float[d0 / W][num_nodes][W], W = table_slice_floats (d0 % W == 0): the
     * column-sliced layer-1 gather then reads one contiguous array per 256-byte slice.  NULL = gather from `table`. */
    const float*   table_sliced;
    int32_t        table_slice_floats;  /* floats per slice of table_sliced: 64 (256-byte slices; 0 means 64) or 32 / 128; d0 % it == 0 */
    /* optional (ABI 4): the caller DECLARES that w1 is the identity matrix [h1, h1] (d0 == h1, gcn encoder) -- serving on a pre-transformed
     * table, Y = X . W1^T computed once per weight update (sage355.engine.pretransform_table): layer 1 is then act1(mean(Y[nbrs])).  When
     * layer 1 runs in its split form the contraction is skipped and the column-sliced gather applies act1 and writes h1 itself; everywhere
     * else the flag is ignored (the contraction with an identity W1 returns its operand exactly, so the result is the same bit for bit).
     * w1 must still point at a real identity matrix. */
    int32_t        w1_is_identity;
} sage_model_t;

/* Where the intermediates of one forward live inside the caller's workspace
 * (byte offsets).  Tests and the benchmark's parity gate read the sampled sets
 * back through this; h1 is what layer 2 consumes. */
typedef struct {
    size_t  total_bytes;
    size_t  counters;     /* int32[8]: [0] frontier rows claimed (|S1| = first row + this), [1] any_nonempty
                             outer, [2] any_nonempty inner, [7] completion ticket; all zero between forwards.
                             int32[8..15]: copy of [0..7] as the LAST forward left them (read-back only)          */
    size_t  hash_keys, hash_rows; int32_t hash_capacity;
    size_t  s1_nodes;     /* int32[max_s1]  layer-1 node ids; concat: rows [0,B) are the seeds */
    int32_t max_s1;
    size_t  nbr2, slot2, cnt2, self_slot2;   /* int32 [B,k2] [B,k2] [B] [B]  */
    size_t  row2, self_row2;                 /* int32 [B,k2] [B]: frontier ROW of every outer sample */
    size_t  nbr1, cnt1;                      /* int32 [max_s1,k1] [max_s1]   */
    size_t  agg1;         /* float [max_s1, d0]   (two-launch form only)     */
    size_t  h1;           /* float [max_s1, h1]                               */
    size_t  agg2;         /* float [B, h1]        (two-launch form only)     */
    int32_t layer1_split; /* 1: layer 1 runs as column-sliced gather -> agg1 -> dense contraction;
                             0: one fused launch (or the generic two-launch form when unsupported)   */
} sage_ws_layout_t;

int sage_forward2_layout(const sage_model_t* m, int32_t max_batch, sage_ws_layout_t* layout_host);

/* The workspace is self-cleaning: each forward wipes the hash keys it used and its last
 * kernel zeroes the device counters, so a forward is 4-5 launches with no memset / reset.
 * sage_forward2_init puts a freshly allocated workspace into that state (call it once,
 * with the LARGEST batch the workspace will see, and use one batch size per workspace
 * thereafter: the layout, hence the key array, depends on it). */
int sage_forward2_init(const sage_model_t* m, void* workspace, size_t workspace_bytes,
                       int32_t max_batch, sage_stream_t stream);

/* seeds int32[batch]; out float[batch, h2] (ldo).  Enqueues the whole forward
 * on `stream`; no host synchronisation.  The sampled sets are a pure function
 * of (seed, node id, hop tag).  With m->queue set, `seeds` and `seed` are ignored
 * (seeds may be NULL) and the call is safe to capture into a hipGraph. */
int sage_forward2(const sage_model_t* m, void* workspace, size_t workspace_bytes,
                  const int32_t* seeds, int32_t batch, uint64_t seed,
                  float* out, int64_t ldo, sage_stream_t stream);

/* Same forward with hipEvent_t markers around its stages, for in-situ kernel timing
 * (bench.py roofline).  stage_events: 2*SAGE_NUM_STAGES hipEvent_t handles
 * (begin, end for interval 0..4 = outer sample, inner sample, layer-1 gather (split layers
 * only, else empty), layer-1 contraction or fused layer 1, layer 2); NULL entries are skipped.
 * Events are recorded on `stream`. */
#define SAGE_NUM_STAGES 5
int sage_forward2_profiled(const sage_model_t* m, void* workspace, size_t workspace_bytes,
                           const int32_t* seeds, int32_t batch, uint64_t seed,
                           float* out, int64_t ldo, sage_stream_t stream,
                           void* const* stage_events);

/* ---------------------------------------------------------------------------
 * Weight preparation for the layer-1 contraction (encoders.py:58-61, `self.weight.mm(combined.t())`).
 * The contraction runs on the bf16 matrix pipe with fp32 accuracy: x.w = sum over the products of the three
 * bf16 terms of x and of w (csrc/sage_dense.hip).  Splitting W [out_dim, dim] into its three bf16 planes,
 * laid out in the kernel's register order, depends only on W, so it can be done once per weight update instead
 * of by every block of every launch (a third of the contraction's time at BASELINE config 3).
 * With concat != 0, W is [out_dim, 2 * dim] in the [self | agg] order of encoders.py:54.
 * sage_prepared_weight_bytes: size of the prepared form (planes + a 16-byte trailer that marks a W holding
 * |w| >= 2^127 / Inf / NaN), 0 if the shape is not one of the contraction kernel's (out_dim > 128, dim % 4 != 0).
 * Results are bit-identical with and without the prepared form.
 * ------------------------------------------------------------------------- */
size_t sage_prepared_weight_bytes(int32_t dim, int32_t out_dim, int32_t concat);
int sage_prepare_weights(const float* weight, int64_t ldw, int32_t dim, int32_t out_dim, int32_t concat,
                         void* prepared, size_t prepared_bytes, sage_stream_t stream);

/* ---------------------------------------------------------------------------
 * Role pipeline: consecutive forwards software-pipelined over ROLE STREAMS (csrc/sage_pipe.hip).
 * Each stage of the forward -- S: outer + inner sample, G: layer-1 gather, D: layer-1 contraction,
 * L: layer 2 -- is enqueued on a HIP stream of its own and consecutive batches move through the
 * stages over `depth` workspaces; hipEvents carry S(b) -> G(b) -> D(b) -> L(b) -> S(b + depth).
 * In steady state one stage (the gather) is on the critical path instead of the sum of five.
 * Results are bit-identical to sage_forward2 on the same (seeds, key).  The pipe owns only its
 * hipEvents; workspaces (each laid out by sage_forward2_layout and initialised by
 * sage_forward2_init for m->ws_batch / `batch`), streams and outputs belong to the caller.
 * streams[4] = {S, G, D, L}; entries may coincide (the event between two roles on one stream is
 * skipped).  All calls are host-side enqueues (eleven stream operations per batch); sage_pipe_fork / sage_pipe_join
 * order the pipe against the caller's own stream.  The whole pattern -- hipStreamBeginCapture(stream), sage_pipe_fork(stream),
 * submits (the first `depth` of them on free workspaces: segment_start, or sage_pipe_reset before), sage_pipe_join(stream),
 * hipStreamEndCapture -- can be captured into ONE hipGraph (ABI 3): while the role streams are capturing, the workspace-release
 * edge L(b) -> S(b + depth) is added as an explicit node dependency, because waiting for it by event crashes
 * hipStreamEndCapture on ROCm 7.2 (csrc/sage_pipe.hip, experiments/r03/capture_repro.cpp).  The captured graph embeds the seeds
 * pointers and keys of the batches it was captured with.  After a capture, call sage_pipe_reset before eager submission.
 * Host enqueue threads (ABI 4): sage_pipe_set_threads(p, 1, window) starts one host thread per role; a submit then only POSTS the
 * batch and each role's thread makes that role's HIP calls on its stream (thirteen HIP calls per batch are 40-60 us on one host
 * thread -- as long as the GPU's period), ordered across threads as hipEvent semantics need.  sage_pipe_flush returns when every
 * posted batch has been enqueued (and reports a role thread's error); join / fork / reset / update_weights / destroy flush first.
 * A caller that synchronises the DEVICE or the role streams itself must flush first.  Needs four distinct role streams.
 * Not thread safe per pipe (one submitting thread).
 * Replaces nothing in the reference (model.py:240-252 runs one batch at a time).
 * ------------------------------------------------------------------------- */
#define SAGE_PIPE_MAX_DEPTH 8
typedef struct sage_pipe sage_pipe_t;
int sage_pipe_create(const sage_model_t* m, int32_t batch, int32_t depth, void* const* workspaces,
                     size_t workspace_bytes, const sage_stream_t* streams, sage_pipe_t** out);
int sage_pipe_destroy(sage_pipe_t* p);
/* The pipe keeps a COPY of *m; after an optimizer step (new pointers and / or freshly prepared planes; the caller
 * orders the role streams behind whatever wrote them, e.g. with sage_pipe_fork): */
int sage_pipe_update_weights(sage_pipe_t* p, const float* w1, const float* w2, const void* w1_prepared);
/* One batch: seeds int32[batch] and out float[batch, h2] must stay valid until the batch has left stream L. */
int sage_pipe_submit(sage_pipe_t* p, const int32_t* seeds, uint64_t key, float* out, int64_t ldo);
/* sage_pipe_submit + two caller-owned hipEvent_t (gather_events[0], [1]) recorded on stream G around the layer-1 gather. */
int sage_pipe_submit_profiled(sage_pipe_t* p, const int32_t* seeds, uint64_t key, float* out, int64_t ldo,
                              void* const* gather_events);
/* n batches from one host loop: batch i reads seeds + i*seed_stride (elements), keys_host[i] (HOST array) and
 * writes out + (i % out_slots)*out_stride.  segment_start != 0: the first `depth` batches of this call find
 * their workspaces free (the caller has joined everything submitted before). */
int sage_pipe_submit_many(sage_pipe_t* p, const int32_t* seeds, int64_t seed_stride, const uint64_t* keys_host,
                          int32_t n, float* out, int64_t ldo, int64_t out_stride, int32_t out_slots,
                          int32_t segment_start);
/* `stream` waits for everything submitted so far / every role stream waits for `stream`. */
int sage_pipe_join(sage_pipe_t* p, sage_stream_t stream);
int sage_pipe_fork(sage_pipe_t* p, sage_stream_t stream);
/* Forget every submit (the next `depth` submits find their workspaces free); the caller has joined / synchronised all of them. */
int sage_pipe_reset(sage_pipe_t* p);
/* on != 0: start the four host enqueue threads (on == 0: drain and stop them).  window > 0: role S enqueues batch b only once batch
 * b - window has left the GPU, which bounds how far the host runs ahead (0 = unbounded; window < 32). */
int sage_pipe_set_threads(sage_pipe_t* p, int32_t on, int32_t window);
/* Every posted batch has been enqueued on the role streams (not: has run).  Returns the first error a role thread met. */
int sage_pipe_flush(sage_pipe_t* p);
/* Express lane (ABI 6).  A batch submitted to an IDLE pipe -- every earlier batch has left layer 2, found with one hipEventQuery -- is
 * enqueued whole on stream L, like sage_forward2, instead of being handed from stream to stream: four record + wait pairs of ~11 us
 * each on an otherwise empty GPU (141 -> ~90 us to the first output of a region; the batches behind it start on the role streams at
 * once).  Same kernels and workspace, bit-identical results; never inside a stream capture; needs four distinct role streams;
 * SAGE_PIPE_EXPRESS=0 (environment, read once) turns it off.  Returns how many batches of this pipe took it so far (-1: NULL pipe). */
int64_t sage_pipe_express_count(const sage_pipe_t* p);

/* (ABI 5: sage_set_option is gone with the only option it carried -- the producer / consumer contraction kernel of round 3 was measured
 * slower inside the pipeline and deleted in round 4.) */

#ifdef __cplusplus
}
#endif
#endif /* SAGE355_H */
