// Two-layer ("2-hop") forward: graphsage/model.py:219-222 wiring of two Encoders,
// enqueued on one stream with no host synchronisation.
//
//   seeds --sample k2 (enc2.adj)--> nbr2 --hash--> frontier U2 = layer-1 node list S1
//   S1    --sample k1 (enc1.adj)--> nbr1
//   layer 1 on S1 : h1 = act1( [table[S1] |] mean(table[nbr1]) . W1^T )     (encoders.py:47-62)
//   layer 2 on B  : out = act2( [h1[seed] |] mean(h1[row(nbr2)]) . W2^T )
//
// Concat encoder (gcn=False): the reference evaluates layer 1 a SECOND time on the seeds
// (self_feats = features(nodes), encoders.py:49-52) with samples of its own; those B rows
// sit at the head of S1 (rows [0,B)) and draw from RNG stream SAGE_TAG_INNER_SELF, the
// frontier rows follow from row B.
#include <string.h>

#include "sage_internal.h"

namespace {

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int next_pow2(int64_t x) {
    int64_t p = 4;
    while (p < x) p <<= 1;
    return (int)p;
}

int check_model(const sage_model_t* m) {
    SAGE_REQUIRE(m, "forward2: NULL model");
    SAGE_REQUIRE(m->num_nodes > 0 && m->num_nodes < (1ll << 31), "forward2: num_nodes = %lld", (long long)m->num_nodes);
    SAGE_REQUIRE(m->d0 >= 1 && m->h1 >= 1 && m->h2 >= 1, "forward2: dims d0=%d h1=%d h2=%d", m->d0, m->h1, m->h2);
    SAGE_REQUIRE(m->k1 >= 1 && m->k1 <= SAGE_MAX_FANOUT && m->k2 >= 1 && m->k2 <= SAGE_MAX_FANOUT,
                 "forward2: fanouts k1=%d k2=%d outside [1, %d]", m->k1, m->k2, SAGE_MAX_FANOUT);
    SAGE_REQUIRE(m->table_ld >= m->d0, "forward2: table_ld = %lld < d0", (long long)m->table_ld);
    SAGE_REQUIRE(m->act1 >= 0 && m->act1 <= SAGE_ACT_NONE && m->act2 >= 0 && m->act2 <= SAGE_ACT_NONE, "forward2: bad activation");
    return SAGE_OK;
}

}  // namespace

extern "C" int sage_forward2_layout(const sage_model_t* m, int32_t max_batch, sage_ws_layout_t* L) {
    if (int rc = check_model(m)) return rc;
    SAGE_REQUIRE(L, "forward2_layout: NULL layout");
    SAGE_REQUIRE(max_batch >= 1 && (int64_t)max_batch * (m->k2 + 1) < (1ll << 30), "forward2_layout: max_batch = %d", max_batch);
    memset(L, 0, sizeof(*L));
    const int64_t B = max_batch;
    const int64_t max_s1 = B * m->k2 + B;   // frontier of B*k2 ids + B self rows (concat or self-loop)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L->counters = take(32 * sizeof(int32_t));    // 16 counters + a spare 64-bit slot (sampler key of a stage-wise forward)
    L->hash_capacity = next_pow2(2 * B * (m->k2 + 1));
    L->hash_keys = take((size_t)L->hash_capacity * 4);
    L->hash_rows = take((size_t)L->hash_capacity * 4);
    L->max_s1 = (int32_t)max_s1;
    L->s1_nodes = take((size_t)max_s1 * 4);
    L->nbr2 = take((size_t)B * m->k2 * 4);
    L->slot2 = take((size_t)B * m->k2 * 4);
    L->cnt2 = take((size_t)B * 4);
    L->self_slot2 = take((size_t)B * 4);
    L->row2 = take((size_t)B * m->k2 * 4);
    L->self_row2 = take((size_t)B * 4);
    L->nbr1 = take((size_t)max_s1 * m->k1 * 4);
    L->cnt1 = take((size_t)max_s1 * 4);
    L->agg1 = take((size_t)max_s1 * m->d0 * 4);
    L->h1 = take((size_t)max_s1 * m->h1 * 4);
    L->agg2 = take((size_t)B * m->h1 * 4);
    L->total_bytes = off;
    L->layer1_split = (m->fused && sage_layer_dense_supported(m->d0, m->h1) && m->table_ld % 4 == 0 && m->k1 <= 64 &&
                       ((m->d0 >= SAGE_SPLIT_MIN_DIM && max_s1 >= 8192) || m->d0 > 256)) ? 1 : 0;
#ifdef SAGE_FORCE_FUSED1
    L->layer1_split = 0;
#endif
    return SAGE_OK;
}

namespace {
// ev: NULL, or 2*SAGE_NUM_STAGES hipEvent_t (begin, end per stage; NULL entries skipped)
#define SAGE_EV(i)                                                                         \
    do {                                                                                   \
        if (ev && ev[i] && hipEventRecord((hipEvent_t)ev[i], st) != hipSuccess) {          \
            sage_set_error("forward2: hipEventRecord failed");                             \
            return SAGE_ELAUNCH;                                                           \
        }                                                                                  \
    } while (0)

int forward2_impl(const sage_model_t* m, void* workspace, size_t workspace_bytes, const int32_t* seeds, int32_t batch,
                  uint64_t seed, float* out, int64_t ldo, sage_stream_t stream, void* const* ev, int stages = SAGE_STAGE_ALL,
                  int cursor_off = 0, bool key_in_ws = false, void* tail_event = nullptr) {
    if (int rc = check_model(m)) return rc;
    SAGE_REQUIRE(m->rowptr1 && m->col1 && m->rowptr2 && m->col2 && m->table && m->w1 && m->w2, "forward2: NULL model array");
    const bool queued = m->queue != nullptr;
    SAGE_REQUIRE(!queued || (m->queue_len >= 1 && m->queue_cursor), "forward2: batch queue without length / cursor");
    SAGE_REQUIRE(workspace && (seeds || queued) && (out || !(stages & SAGE_STAGE_LAYER2)), "forward2: NULL argument");
    SAGE_REQUIRE(batch >= 1, "forward2: batch = %d", batch);
    SAGE_REQUIRE(ldo >= m->h2 || !(stages & SAGE_STAGE_LAYER2), "forward2: ldo = %lld < h2", (long long)ldo);
    SAGE_REQUIRE(sage_aligned(workspace, 256), "forward2: workspace not 256-byte aligned");
    SAGE_REQUIRE(m->ws_batch == 0 || batch <= m->ws_batch, "forward2: batch %d > ws_batch %d", batch, m->ws_batch);
    sage_ws_layout_t L;
    if (int rc = sage_forward2_layout(m, m->ws_batch ? m->ws_batch : batch, &L)) return rc;
    if (L.total_bytes > workspace_bytes) {
        sage_set_error("forward2: workspace %zu bytes < %zu needed for batch %d", workspace_bytes, L.total_bytes, batch);
        return SAGE_ENOSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    // tail_event (sage_pipe.hip): to be recorded behind the LAST launch of `stages` -- as that launch's own completion signal where its
    // launcher knows SAGE_LAUNCH_TAIL (arm() right before the stage's final launcher call), by hipEventRecord otherwise (settle())
    int last_stage = 0;
    for (int bit = SAGE_STAGE_LAYER2; bit >= 1; bit >>= 1)
        if (tail_event && (stages & bit)) { last_stage = bit; break; }
    struct TailGuard { ~TailGuard() { sage_tail_event = nullptr; } } tail_guard;      // an error return must not leave it armed on this thread
    bool armed = false;
    auto arm = [&](int stage) { if (stage == last_stage) { sage_tail_event = tail_event; armed = true; } };
    auto settle = [&](int stage) -> int {
        if (stage != last_stage) return SAGE_OK;
        if (armed && !sage_tail_event) return SAGE_OK;                                // the stage's last launch carried it
        sage_tail_event = nullptr;                                                    // nothing launched, or by a launcher that does not know it
        if (hipEventRecord((hipEvent_t)tail_event, st) != hipSuccess) { sage_set_error("forward2: hipEventRecord failed"); return SAGE_ELAUNCH; }
        return SAGE_OK;
    };
    char* ws = (char*)workspace;
    int32_t* counters = (int32_t*)(ws + L.counters);
    int32_t* s1_count = counters + 0;      // frontier rows claimed so far (zero based; rows start at first_row)
    int32_t* any2 = counters + 1;
    int32_t* any1 = counters + 2;
    int32_t* s1_nodes = (int32_t*)(ws + L.s1_nodes);
    int32_t* nbr2 = (int32_t*)(ws + L.nbr2);
    int32_t* slot2 = (int32_t*)(ws + L.slot2);
    int32_t* cnt2 = (int32_t*)(ws + L.cnt2);
    int32_t* self_slot2 = (int32_t*)(ws + L.self_slot2);
    int32_t* row2 = (int32_t*)(ws + L.row2);
    int32_t* self_row2 = (int32_t*)(ws + L.self_row2);
    int32_t* nbr1 = (int32_t*)(ws + L.nbr1);
    int32_t* cnt1 = (int32_t*)(ws + L.cnt1);
    float* agg1 = (float*)(ws + L.agg1);
    float* h1 = (float*)(ws + L.h1);
    float* agg2 = (float*)(ws + L.agg2);
    const sage_frontier_t fr{(int32_t*)(ws + L.hash_keys), (int32_t*)(ws + L.hash_rows), L.hash_capacity, s1_nodes, s1_count, L.max_s1};
    const int first_row = m->concat ? batch : 0;
    const int self_loop = m->agg_self_loop ? 1 : 0;
    const sage_model_t* qm = queued ? m : nullptr;
    const sage_finish_t no_fin{nullptr, nullptr};
    const sage_finish_t fin{counters, queued ? m->queue_cursor : nullptr};

    // The workspace is self-cleaning (sage_forward2_init once, then every forward leaves the hash
    // keys wiped and the counters zero), so a forward is exactly 4 launches (5 when layer 1 is split into
    // gather + contraction, 6 with the generic two-launch layers).
    const int64_t ldw1 = (int64_t)m->d0 * (m->concat ? 2 : 1);
    const int64_t ldw2 = (int64_t)m->h1 * (m->concat ? 2 : 1);
    const int32_t* nan1 = m->nan_empty ? any1 : nullptr;
    // outer hop: some neighbour was sampled  <=>  the frontier counter is non-zero, so layer 2 needs no flag at all
    const int32_t* nan2 = m->nan_empty ? (self_loop ? any2 : s1_count) : nullptr;
    const bool fuse1 = m->fused && sage_layer_fused_supported(m->d0, m->h1, m->concat) && m->table_ld % 4 == 0 &&
                       sage_aligned(m->table, 16) && sage_aligned(m->w1, 16);
    // wide + large layer 1: column-sliced gather (cross-XCD L2 partitioning) into agg1, then a dense contraction;
    // otherwise the one-launch fused layer; otherwise the generic two-launch form
    const bool split1 = m->fused && L.layer1_split && sage_aligned(m->table, 16) && sage_aligned(m->w1, 16) &&
                        sage_gather_is_sliced(m->d0, m->table_ld, m->d0, m->table, agg1, L.max_s1, m->k1);
    uint64_t* key_slot = key_in_ws ? (uint64_t*)(counters + 16) : nullptr;    // in the 256-B slot of the counters, past the 16 ints in use
    // Layer 2 as a one-launch layer resolves hash slots itself, so both hops can be sampled by ONE launch (sage_sample.hip:
    // sample_fused_kernel).  The choice depends on the model only, so every call on a workspace agrees on who resolves the slots.
    const bool fuse2 = m->fused && sage_layer_fused_supported(m->h1, m->h2, m->concat) && sage_aligned(m->w2, 16);
    const bool sfused = fuse2 && sage_tunables().sample_fused != 0 && m->k1 <= 64 && m->k2 <= 64;
    const int both = SAGE_STAGE_SAMPLE_OUTER | SAGE_STAGE_SAMPLE_INNER;
    if (sfused && (stages & both)) {
        SAGE_REQUIRE((stages & both) == both, "forward2: with the fused sampler the two sampling stages are one launch: pass "
                                                "SAGE_STAGE_SAMPLE_OUTER | SAGE_STAGE_SAMPLE_INNER together");
        SAGE_EV(0);
        if (int rc = sage_launch_sample_fused(m, seeds, batch, seed, nbr2, cnt2, (m->nan_empty && self_loop) ? any2 : nullptr, &fr, self_loop, slot2,
                                              self_slot2, queued ? 1 : 0, m->concat ? s1_nodes : nullptr, first_row, nbr1, cnt1,
                                              m->nan_empty ? any1 : nullptr, m->concat ? batch : 0, st))
            return rc;
        SAGE_EV(1);
        SAGE_EV(2);
        SAGE_EV(3);
        if (last_stage == SAGE_STAGE_SAMPLE_INNER || last_stage == SAGE_STAGE_SAMPLE_OUTER)
            if (int rc = settle(last_stage)) return rc;
    }
    if (!sfused && (stages & SAGE_STAGE_SAMPLE_OUTER)) {
    // 1. outer hop: seeds -> nbr2, hash insert -> frontier rows [first_row, ...)
    SAGE_EV(0);
    arm(SAGE_STAGE_SAMPLE_OUTER);
    if (int rc = sage_launch_sample(m->rowptr2, m->col2, m->num_nodes, seeds, batch, nullptr, m->k2, seed, SAGE_TAG_OUTER, 0, SAGE_TAG_OUTER, nbr2,
                                    cnt2, (m->nan_empty && self_loop) ? any2 : nullptr, &fr, self_loop, slot2, self_slot2, qm, 1, m->concat ? s1_nodes : nullptr, 0, first_row,
                                    nullptr, cursor_off, key_slot, m->seed_map, st))
        return rc;
    if (int rc = settle(SAGE_STAGE_SAMPLE_OUTER)) return rc;
    SAGE_EV(1);
    }
    if (!sfused && (stages & SAGE_STAGE_SAMPLE_INNER)) {
    // 2. inner hop: S1 -> nbr1 (raw table rows; duplicates are served by L2 / Infinity Cache).  Its spare
    //    threads turn the outer hop's hash slots into frontier rows and wipe the used keys.
    //    (Drawing these samples inside the layer-1 gather instead was measured: the gather went from 48 to
    //    100 us, its per-row dependent chain growing from 2 to 5 round trips.)
    const sage_resolve_t resolve{slot2, row2, batch * m->k2, self_loop ? self_slot2 : nullptr, self_row2, batch, fr.rows, fr.keys};
    SAGE_EV(2);
    arm(SAGE_STAGE_SAMPLE_INNER);
    if (int rc = sage_launch_sample(m->rowptr1, m->col1, m->num_nodes, s1_nodes, L.max_s1, s1_count, m->k1, seed, SAGE_TAG_INNER, first_row,
                                    SAGE_TAG_INNER_SELF, nbr1, cnt1, m->nan_empty ? any1 : nullptr, nullptr, 0, nullptr, nullptr, qm, 0, nullptr, first_row, 0,
                                    &resolve, cursor_off, key_slot, nullptr, st))
        return rc;
    if (int rc = settle(SAGE_STAGE_SAMPLE_INNER)) return rc;
    SAGE_EV(3);
    }
    // serving on a pre-transformed table (sage_model_t.w1_is_identity): the split layer's gather applies act1 and writes h1; no contraction
    const bool gather_only1 = split1 && m->w1_is_identity != 0 && !m->concat && m->d0 == m->h1 &&
                              sage_gather_is_sliced(m->d0, m->table_ld, m->h1, m->table, h1, L.max_s1, m->k1);
    // 3. layer 1 on S1: the HBM-bound gather ...
    if (stages & SAGE_STAGE_GATHER1) {
    // (sage_forward2_profiled: events 4 / 5 are the gather launch's own start / stop events when it takes a column-sliced form)
    const sage_ext_launch_t gx{ev ? ev[4] : nullptr, ev ? ev[5] : nullptr};
    const bool ext_g = ev && ev[4] && ev[5] && (gather_only1 || split1) && sage_ext_launch == nullptr;
    if (ext_g) sage_ext_launch = &gx; else SAGE_EV(4);
    struct ClearHook { bool on; ~ClearHook() { if (on) sage_ext_launch = nullptr; } } clear_hook{ext_g};
    arm(SAGE_STAGE_GATHER1);               // (a launch that carries the measurement hook's events leaves it armed: settle() records)
    if (gather_only1) {
        const int sw = m->table_slice_floats ? m->table_slice_floats : 64;
        const bool sm = m->table_sliced != nullptr && (sw == 32 || sw == 64 || sw == 128) && m->d0 % sw == 0 && sage_aligned(m->table_sliced, 16);
        if (int rc = sage_launch_gather_mean(sm ? m->table_sliced : m->table, m->num_nodes, sm ? sw : m->table_ld, m->d0, nbr1, cnt1, m->k1, L.max_s1,
                                             s1_count, nullptr, self_loop ? s1_nodes : nullptr, nan1, h1, m->h1, first_row, st,
                                             sm ? m->num_nodes * (int64_t)sw : 0, m->act1))
            return rc;
    } else if (split1) {
        // optional slice-major copy of the table ([d0 / 64][num_nodes][64]): every XCD pair reads ONE contiguous array
        const int sw = m->table_slice_floats ? m->table_slice_floats : 64;
        const bool sm = m->table_sliced != nullptr && (sw == 32 || sw == 64 || sw == 128) && m->d0 % sw == 0 && sage_aligned(m->table_sliced, 16);
        if (int rc = sage_launch_gather_mean(sm ? m->table_sliced : m->table, m->num_nodes, sm ? sw : m->table_ld, m->d0, nbr1, cnt1, m->k1, L.max_s1,
                                             s1_count, nullptr, self_loop ? s1_nodes : nullptr, nan1, agg1, m->d0, first_row, st,
                                             sm ? m->num_nodes * (int64_t)sw : 0))
            return rc;
    }
    if (int rc = settle(SAGE_STAGE_GATHER1)) return rc;
    if (!ext_g) SAGE_EV(5);
    }
    // ... and its contraction (one launch with the gather unless the layer is split); then layer 2
    if (stages & SAGE_STAGE_CONTRACT1) {
    SAGE_EV(6);
    if (gather_only1) {
        // nothing: the gather wrote h1
    } else if (split1) {
        arm(SAGE_STAGE_CONTRACT1);
        if (int rc = sage_launch_layer_dense(agg1, m->d0, m->d0, L.max_s1, s1_count, m->concat, m->table, m->table_ld, m->num_nodes,
                                             s1_nodes, nullptr, nullptr, m->w1, ldw1, m->h1, m->act1, h1, m->h1, first_row, no_fin, m->w1_prepared, st))
            return rc;
    } else if (fuse1) {
        arm(SAGE_STAGE_CONTRACT1);
        if (int rc = sage_launch_layer_fused(m->table, m->num_nodes, m->table_ld, m->d0, nbr1, cnt1, m->k1, L.max_s1, s1_count, nullptr,
                                             self_loop ? s1_nodes : nullptr, nan1, m->concat, s1_nodes, m->w1, ldw1, m->h1, m->act1,
                                             h1, m->h1, first_row, no_fin, st))
            return rc;
    } else {
        if (int rc = sage_launch_gather_mean(m->table, m->num_nodes, m->table_ld, m->d0, nbr1, cnt1, m->k1, L.max_s1, s1_count, nullptr,
                                             self_loop ? s1_nodes : nullptr, nan1, agg1, m->d0, first_row, st))
            return rc;
        if (int rc = sage_launch_linear_act(m->concat ? m->table : nullptr, m->table_ld, s1_nodes, agg1, m->d0, m->d0, m->w1, ldw1,
                                            m->h1, m->act1, L.max_s1, s1_count, h1, m->h1, first_row, no_fin, st))
            return rc;
    }
    if (int rc = settle(SAGE_STAGE_CONTRACT1)) return rc;
    SAGE_EV(7);
    }
    if (stages & SAGE_STAGE_LAYER2) {
    // 4. layer 2 on the seeds; its last block zeroes the counters and advances the batch queue
    SAGE_EV(8);
    if (fuse2 && sfused) {
        // slots in, rows resolved (and left in row2 / self_row2), keys wiped: the duties the inner-hop launch had
        const sage_slot_resolve_t rs{fr.keys, row2, self_loop ? self_row2 : nullptr};
        arm(SAGE_STAGE_LAYER2);
        if (int rc = sage_launch_layer_fused(h1, L.max_s1, m->h1, m->h1, slot2, cnt2, m->k2, batch, nullptr, fr.rows,
                                             self_loop ? self_slot2 : nullptr, nan2, m->concat, nullptr, m->w2, ldw2, m->h2, m->act2,
                                             out, ldo, 0, fin, st, &rs))
            return rc;
    } else if (fuse2) {
        arm(SAGE_STAGE_LAYER2);
        if (int rc = sage_launch_layer_fused(h1, L.max_s1, m->h1, m->h1, row2, cnt2, m->k2, batch, nullptr, nullptr,
                                             self_loop ? self_row2 : nullptr, nan2, m->concat, nullptr, m->w2, ldw2, m->h2, m->act2,
                                             out, ldo, 0, fin, st))
            return rc;
    } else {
        if (int rc = sage_launch_gather_mean(h1, L.max_s1, m->h1, m->h1, row2, cnt2, m->k2, batch, nullptr, nullptr,
                                             self_loop ? self_row2 : nullptr, nan2, agg2, m->h1, 0, st))
            return rc;
        if (int rc = sage_launch_linear_act(m->concat ? h1 : nullptr, m->h1, nullptr, agg2, m->h1, m->h1, m->w2, ldw2, m->h2, m->act2,
                                            batch, nullptr, out, ldo, 0, fin, st))
            return rc;
    }
    if (int rc = settle(SAGE_STAGE_LAYER2)) return rc;
    SAGE_EV(9);
    }
    return SAGE_OK;
}
}  // namespace

extern "C" int sage_forward2(const sage_model_t* m, void* workspace, size_t workspace_bytes, const int32_t* seeds, int32_t batch,
                             uint64_t seed, float* out, int64_t ldo, sage_stream_t stream) {
    return forward2_impl(m, workspace, workspace_bytes, seeds, batch, seed, out, ldo, stream, nullptr);
}

// A subset of the forward's launches with the seeds and the sampler key taken from the call (sage_pipe.hip: one call per role stream)
int sage_forward2_launch_stages(const sage_model_t* m, void* workspace, size_t workspace_bytes, const int32_t* seeds, int32_t batch,
                                uint64_t seed, float* out, int64_t ldo, int32_t stages, hipStream_t stream, void* tail_event) {
    return forward2_impl(m, workspace, workspace_bytes, seeds, batch, seed, out, ldo, (sage_stream_t)stream, nullptr, stages, 0, false, tail_event);
}

extern "C" int sage_forward2_profiled(const sage_model_t* m, void* workspace, size_t workspace_bytes, const int32_t* seeds,
                                      int32_t batch, uint64_t seed, float* out, int64_t ldo, sage_stream_t stream,
                                      void* const* stage_events) {
    return forward2_impl(m, workspace, workspace_bytes, seeds, batch, seed, out, ldo, stream, stage_events);
}

// Put a fresh (or dirty) workspace into the state every forward leaves behind: counters zero,
// hash keys empty.  Call once after allocating the workspace (and after an aborted stream).
extern "C" int sage_forward2_init(const sage_model_t* m, void* workspace, size_t workspace_bytes, int32_t max_batch,
                                  sage_stream_t stream) {
    if (int rc = check_model(m)) return rc;
    SAGE_REQUIRE(workspace && sage_aligned(workspace, 256), "forward2_init: workspace NULL or not 256-byte aligned");
    sage_ws_layout_t L;
    if (int rc = sage_forward2_layout(m, max_batch, &L)) return rc;
    if (L.total_bytes > workspace_bytes) {
        sage_set_error("forward2_init: workspace %zu bytes < %zu needed for batch %d", workspace_bytes, L.total_bytes, max_batch);
        return SAGE_ENOSPACE;
    }
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    // smaller batches use a prefix of the key array with a smaller power-of-two capacity: wipe the largest
    if (int rc = sage_fill_u32(ws + L.counters, 0u, 16, st)) return rc;
    if (int rc = sage_fill_u32(ws + L.hash_keys, 0xFFFFFFFFu, (size_t)L.hash_capacity, st)) return rc;
    return SAGE_OK;
}
