// Role pipeline: consecutive 2-hop forwards software-pipelined over ROLE STREAMS.
//
// One forward is a chain of five dependent launches (sage_forward.hip); enqueued on one stream it pays a kernel
// boundary between every pair and leaves the latency-bound stages (samplers, layer 2) alone on the chip.  Here each
// STAGE of the forward has a HIP stream of its own and consecutive batches flow through them like through an
// assembly line, over `depth` workspaces:
//
//     stream S :  sample(b+2)        outer hop + frontier, inner hop          (dependent round trips)
//     stream G :  gather(b+1)        layer-1 column-sliced gather             (fabric / HBM bound: the pacemaker)
//     stream D :  contract(b)        layer-1 split-bf16 MFMA contraction      (matrix pipe)
//     stream L :  layer2(b-1)        layer-2 gather + MFMA, workspace release
//
// so that what is on the critical path is one stage, not the sum of five, and every stream's own boundary
// (kernel fill/drain, L2 write-back) is covered by the other streams' kernels.  Dependencies are hipEvents:
// S(b) -> G(b) -> D(b) -> L(b) -> S(b + depth) (workspace reuse).  Streams may coincide (an event between two
// roles on the same stream is skipped), so {S,G,D,L} = one stream degenerates to sage_forward2.
// Results are bit-identical to sage_forward2 on the same (seeds, key): same kernels, same workspaces' layout.
// The reference has no counterpart: model.py:240-252 runs one batch at a time on the host.
//
// Stream capture (round 3).  fork -> submits -> join can be captured into ONE hipGraph (hipStreamBeginCapture on the caller's
// stream; the role streams join the capture through the fork event) with ONE exception, found with a stand-alone program of
// trivial kernels (experiments/r03/capture_repro.cpp, patterns 0-17 on ROCm 7.2): hipStreamEndCapture segfaults when stream S
// waits, by hipStreamWaitEvent, for the event L recorded for an EARLIER batch -- the workspace-release edge L(b) -> S(b + depth);
// fresh events per record, relaxed capture mode, system-fence flags and a relay through a helper stream change nothing, the same
// pattern without that wait (or with at most `depth` batches, which never need it) captures and replays correctly, and so do
// two-stream patterns with a wait on an event that is no longer its stream's tail.  Expressed WITHOUT an event -- the node L(b)
// left as the tail of stream L is read back with hipStreamGetCaptureInfo_v2 right after it is enqueued, and added to S's next node
// with hipStreamUpdateCaptureDependencies -- the very same graph captures, instantiates and replays correctly (patterns 15, 17).
// submit_one does exactly that while the role streams are capturing; eager submission keeps the event.
#include <stdlib.h>

#include <new>

#include "sage_internal.h"

struct sage_pipe {
    sage_model_t model;
    int32_t batch;
    int32_t depth;
    void* ws[SAGE_PIPE_MAX_DEPTH];
    size_t ws_bytes;
    hipStream_t st[4];                                  // S, G, D, L
    hipEvent_t ev[4][SAGE_PIPE_MAX_DEPTH];              // [role][slot]: role's work on the slot's batch is enqueued
    hipEvent_t ev_fork;
    uint64_t submitted;
    // while capturing: the graph node(s) layer 2 of the slot's last batch left as the tail of stream L, and the capture they belong to
    hipGraphNode_t cap_nodes[SAGE_PIPE_MAX_DEPTH][4];
    int cap_count[SAGE_PIPE_MAX_DEPTH];
    unsigned long long cap_id[SAGE_PIPE_MAX_DEPTH];
};

namespace {
enum { RS = 0, RG = 1, RD = 2, RL = 3 };
// no timing, and no system-scope fence when an event completes: the hand-offs are device-to-device (agent scope is what the
// next kernel's launch acquires anyway); a system fence per record is an L2 write-back on the stage's critical path
static unsigned event_flags() {
    const char* v = getenv("SAGE_PIPE_SYSFENCE");      // diagnostic: keep the system-scope fence on event completion
    return (v && *v == '1') ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
}
#define kEventFlags event_flags()

int wait_on(sage_pipe* p, int consumer, int producer, int slot) {
    if (p->st[consumer] == p->st[producer]) return SAGE_OK;          // stream order already says it
    if (hipStreamWaitEvent(p->st[consumer], p->ev[producer][slot], 0) != hipSuccess) {
        sage_set_error("pipe: hipStreamWaitEvent failed");
        return SAGE_ELAUNCH;
    }
    return SAGE_OK;
}
// 0 = not capturing; otherwise the id of the capture the stream belongs to
int capture_id(hipStream_t st, unsigned long long* id) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    *id = 0;
    if (hipStreamGetCaptureInfo_v2(st, &status, id, nullptr, nullptr, nullptr) != hipSuccess) {
        sage_set_error("pipe: hipStreamGetCaptureInfo_v2 failed");
        return SAGE_ELAUNCH;
    }
    if (status != hipStreamCaptureStatusActive) *id = 0;
    return SAGE_OK;
}
int record(sage_pipe* p, int role, int slot, bool needed) {
    if (!needed) return SAGE_OK;
    if (hipEventRecord(p->ev[role][slot], p->st[role]) != hipSuccess) {
        sage_set_error("pipe: hipEventRecord failed");
        return SAGE_ELAUNCH;
    }
    return SAGE_OK;
}
}  // namespace

extern "C" int sage_pipe_create(const sage_model_t* m, int32_t batch, int32_t depth, void* const* workspaces, size_t workspace_bytes,
                                const sage_stream_t* streams, sage_pipe_t** out) {
    SAGE_REQUIRE(m && workspaces && streams && out, "pipe_create: NULL argument");
    SAGE_REQUIRE(!m->queue, "pipe_create: the pipeline takes seeds and keys per submit, not from a batch queue");
    SAGE_REQUIRE(depth >= 1 && depth <= SAGE_PIPE_MAX_DEPTH, "pipe_create: depth = %d outside [1, %d]", depth, SAGE_PIPE_MAX_DEPTH);
    SAGE_REQUIRE(batch >= 1 && (m->ws_batch == 0 || batch <= m->ws_batch), "pipe_create: batch = %d", batch);
    sage_ws_layout_t L;
    if (int rc = sage_forward2_layout(m, m->ws_batch ? m->ws_batch : batch, &L)) return rc;
    if (L.total_bytes > workspace_bytes) {
        sage_set_error("pipe_create: workspace %zu bytes < %zu needed", workspace_bytes, L.total_bytes);
        return SAGE_ENOSPACE;
    }
    for (int i = 0; i < depth; ++i) {
        SAGE_REQUIRE(workspaces[i] && sage_aligned(workspaces[i], 256), "pipe_create: workspace %d NULL or not 256-byte aligned", i);
        for (int j = 0; j < i; ++j) SAGE_REQUIRE(workspaces[i] != workspaces[j], "pipe_create: workspaces %d and %d coincide", i, j);
    }
    sage_pipe* p = new (std::nothrow) sage_pipe();
    SAGE_REQUIRE(p, "pipe_create: out of host memory");
    p->model = *m;
    p->batch = batch;
    p->depth = depth;
    p->ws_bytes = workspace_bytes;
    p->submitted = 0;
    for (int i = 0; i < SAGE_PIPE_MAX_DEPTH; ++i) { p->cap_count[i] = 0; p->cap_id[i] = 0; }
    for (int i = 0; i < depth; ++i) p->ws[i] = workspaces[i];
    for (int r = 0; r < 4; ++r) p->st[r] = (hipStream_t)streams[r];
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < depth; ++i) p->ev[r][i] = nullptr;
    p->ev_fork = nullptr;
    if (hipEventCreateWithFlags(&p->ev_fork, kEventFlags) != hipSuccess) {
        sage_set_error("pipe_create: hipEventCreate failed");
        delete p;
        return SAGE_ELAUNCH;
    }
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < depth; ++i)
            if (hipEventCreateWithFlags(&p->ev[r][i], kEventFlags) != hipSuccess) {
                sage_set_error("pipe_create: hipEventCreate failed");
                sage_pipe_destroy(p);
                return SAGE_ELAUNCH;
            }
    *out = p;
    return SAGE_OK;
}

extern "C" int sage_pipe_destroy(sage_pipe_t* p) {
    if (!p) return SAGE_OK;
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < p->depth; ++i)
            if (p->ev[r][i]) (void)hipEventDestroy(p->ev[r][i]);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    delete p;
    return SAGE_OK;
}

extern "C" int sage_pipe_update_weights(sage_pipe_t* p, const float* w1, const float* w2, const void* w1_prepared) {
    SAGE_REQUIRE(p && w1 && w2, "pipe_update_weights: NULL argument");
    p->model.w1 = w1;
    p->model.w2 = w2;
    p->model.w1_prepared = w1_prepared;
    return SAGE_OK;
}

// One batch through the four role streams.  `fresh_slot`: no earlier submit of this pipe is outstanding on the slot (a fresh
// pipe, or the first `depth` submits after the caller joined and synchronised everything before): the workspace-release wait
// is skipped.
static int submit_one(sage_pipe* p, const int32_t* seeds, uint64_t key, float* out, int64_t ldo, bool fresh_slot,
                      void* const* gather_events = nullptr) {
    const int slot = (int)(p->submitted % (uint64_t)p->depth);
    const sage_model_t* m = &p->model;
    void* ws = p->ws[slot];
    // S: outer + inner sample.  Needs the slot's previous batch to have left layer 2 (its last block zeroes the counters).
    unsigned long long cap = 0;
    if (int rc = capture_id(p->st[RS], &cap)) return rc;
    if (!fresh_slot && p->st[RS] != p->st[RL]) {
        if (cap != 0) {
            // captured: the release edge as an explicit node dependency (an event wait here crashes hipStreamEndCapture, see above)
            SAGE_REQUIRE(p->cap_id[slot] == cap && p->cap_count[slot] > 0,
                         "pipe: inside a stream capture the first `depth` submits must find their workspaces free: join everything "
                         "submitted before, begin the capture, and pass segment_start (sage_pipe_submit_many) / call sage_pipe_reset");
            if (hipStreamUpdateCaptureDependencies(p->st[RS], p->cap_nodes[slot], (size_t)p->cap_count[slot], hipStreamAddCaptureDependencies) != hipSuccess) {
                sage_set_error("pipe: hipStreamUpdateCaptureDependencies failed");
                return SAGE_ELAUNCH;
            }
        } else {
            SAGE_REQUIRE(p->cap_id[slot] == 0, "pipe: this slot's last batch was submitted inside a stream capture: call sage_pipe_reset "
                                               "(after synchronising) before submitting eagerly again");
            if (int rc = wait_on(p, RS, RL, slot)) return rc;
        }
    }
#ifndef SAGE_PIPE_SKIP_S   // diagnostic builds only (experiments/ab_build.sh).  _G and _D may be skipped alone (stale data downstream); _S and _L
                           // only together with everything else ("events only"): the samplers fill and layer 2 wipes the frontier hash, and
                           // one without the other leaves a full table behind (an outer sampler probing it took 67 ms per batch)
    if (int rc = sage_forward2_launch_stages(m, ws, p->ws_bytes, seeds, p->batch, key, nullptr, 0,
                                             SAGE_STAGE_SAMPLE_OUTER | SAGE_STAGE_SAMPLE_INNER, p->st[RS]))
        return rc;
#endif
    if (int rc = record(p, RS, slot, p->st[RG] != p->st[RS])) return rc;
    // G: the layer-1 gather (nothing to launch when layer 1 is a one-launch layer; D then waits on S through G's stream order)
    if (int rc = wait_on(p, RG, RS, slot)) return rc;
    if (gather_events && gather_events[0] && hipEventRecord((hipEvent_t)gather_events[0], p->st[RG]) != hipSuccess) { sage_set_error("pipe: hipEventRecord failed"); return SAGE_ELAUNCH; }
#ifndef SAGE_PIPE_SKIP_G   // diagnostic builds (experiments/ab_build.sh): the pipeline without one of its stages' kernels, stale data downstream
    if (int rc = sage_forward2_launch_stages(m, ws, p->ws_bytes, seeds, p->batch, key, nullptr, 0, SAGE_STAGE_GATHER1, p->st[RG])) return rc;
#endif
    if (gather_events && gather_events[1] && hipEventRecord((hipEvent_t)gather_events[1], p->st[RG]) != hipSuccess) { sage_set_error("pipe: hipEventRecord failed"); return SAGE_ELAUNCH; }
    if (int rc = record(p, RG, slot, p->st[RD] != p->st[RG])) return rc;
    // D: the contraction (or the whole fused layer 1)
    if (int rc = wait_on(p, RD, RG, slot)) return rc;
#ifndef SAGE_PIPE_SKIP_D
    if (int rc = sage_forward2_launch_stages(m, ws, p->ws_bytes, seeds, p->batch, key, nullptr, 0, SAGE_STAGE_CONTRACT1, p->st[RD])) return rc;
#endif
    if (int rc = record(p, RD, slot, p->st[RL] != p->st[RD])) return rc;
    // L: layer 2; afterwards the workspace is clean again
    if (int rc = wait_on(p, RL, RD, slot)) return rc;
#ifndef SAGE_PIPE_SKIP_L
    if (int rc = sage_forward2_launch_stages(m, ws, p->ws_bytes, seeds, p->batch, key, out, ldo, SAGE_STAGE_LAYER2, p->st[RL])) return rc;
#endif
    if (int rc = record(p, RL, slot, p->st[RS] != p->st[RL])) return rc;
    p->cap_id[slot] = 0;
    p->cap_count[slot] = 0;
    if (cap != 0 && p->st[RS] != p->st[RL]) {               // remember the node layer 2 left as the tail of stream L
        hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
        unsigned long long id = 0;
        const hipGraphNode_t* deps = nullptr;
        size_t ndeps = 0;
        if (hipStreamGetCaptureInfo_v2(p->st[RL], &status, &id, nullptr, &deps, &ndeps) != hipSuccess || status != hipStreamCaptureStatusActive ||
            ndeps < 1 || ndeps > 4) {
            sage_set_error("pipe: cannot read the tail of stream L during capture (%zu nodes)", ndeps);
            return SAGE_ELAUNCH;
        }
        for (size_t i = 0; i < ndeps; ++i) p->cap_nodes[slot][i] = deps[i];
        p->cap_count[slot] = (int)ndeps;
        p->cap_id[slot] = id;
    }
    ++p->submitted;
    return SAGE_OK;
}

// Forget every submit: the next `depth` submits find their workspaces free.  The caller has synchronised (or joined) everything
// submitted before -- e.g. after a stream capture ended, before eager submission resumes on the same pipe.
extern "C" int sage_pipe_reset(sage_pipe_t* p) {
    SAGE_REQUIRE(p, "pipe_reset: NULL pipe");
    p->submitted = 0;
    for (int i = 0; i < SAGE_PIPE_MAX_DEPTH; ++i) { p->cap_count[i] = 0; p->cap_id[i] = 0; }
    return SAGE_OK;
}

extern "C" int sage_pipe_submit(sage_pipe_t* p, const int32_t* seeds, uint64_t key, float* out, int64_t ldo) {
    SAGE_REQUIRE(p && seeds && out, "pipe_submit: NULL argument");
    SAGE_REQUIRE(ldo >= p->model.h2, "pipe_submit: ldo = %lld < h2", (long long)ldo);
    return submit_one(p, seeds, key, out, ldo, p->submitted < (uint64_t)p->depth);
}

// The same with two caller-owned hipEvent_t recorded on stream G right before and right after the layer-1 gather launch
// (bench.py: the dominant kernel's duration inside the running pipeline).
extern "C" int sage_pipe_submit_profiled(sage_pipe_t* p, const int32_t* seeds, uint64_t key, float* out, int64_t ldo, void* const* gather_events) {
    SAGE_REQUIRE(p && seeds && out && gather_events, "pipe_submit_profiled: NULL argument");
    SAGE_REQUIRE(ldo >= p->model.h2, "pipe_submit_profiled: ldo = %lld < h2", (long long)ldo);
    return submit_one(p, seeds, key, out, ldo, p->submitted < (uint64_t)p->depth, gather_events);
}

// n batches in one call (one host loop, no per-batch crossing of the language boundary): batch i takes
// seeds + i*seed_stride, keys[i] and writes out + (i % out_slots) * out_stride.
// segment_start != 0: treat the first `depth` batches as having fresh slots (after sage_pipe_join + a synchronisation
// of everything submitted before).
extern "C" int sage_pipe_submit_many(sage_pipe_t* p, const int32_t* seeds, int64_t seed_stride, const uint64_t* keys_host, int32_t n,
                                     float* out, int64_t ldo, int64_t out_stride, int32_t out_slots, int32_t segment_start) {
    SAGE_REQUIRE(p && seeds && keys_host && out, "pipe_submit_many: NULL argument");
    SAGE_REQUIRE(n >= 0 && out_slots >= 1 && seed_stride >= 0 && out_stride >= 0, "pipe_submit_many: n = %d, out_slots = %d", n, out_slots);
    SAGE_REQUIRE(ldo >= p->model.h2, "pipe_submit_many: ldo = %lld < h2", (long long)ldo);
    SAGE_REQUIRE(out_slots >= p->depth || out_stride == 0, "pipe_submit_many: %d output slots for %d batches in flight", out_slots, p->depth);
    const uint64_t base = segment_start ? p->submitted : 0;
    for (int32_t i = 0; i < n; ++i) {
        const bool fresh = p->submitted - base < (uint64_t)p->depth && (segment_start || p->submitted < (uint64_t)p->depth);
        if (int rc = submit_one(p, seeds + (int64_t)i * seed_stride, keys_host[i], out + (int64_t)(i % out_slots) * out_stride, ldo, fresh)) return rc;
    }
    return SAGE_OK;
}

// Make `stream` wait for everything submitted so far (all four roles).
extern "C" int sage_pipe_join(sage_pipe_t* p, sage_stream_t stream) {
    SAGE_REQUIRE(p, "pipe_join: NULL pipe");
    if (p->submitted == 0) return SAGE_OK;
    hipStream_t st = (hipStream_t)stream;
    for (int r = 0; r < 4; ++r) {
        if (p->st[r] == st) continue;
        bool seen = false;
        for (int q = 0; q < r; ++q) seen |= p->st[q] == p->st[r];
        if (seen) continue;
        hipEvent_t e = p->ev[r][(p->submitted - 1) % (uint64_t)p->depth];
        // the role's last record may have been skipped (same stream as its consumer): record a fresh marker
        if (hipEventRecord(e, p->st[r]) != hipSuccess || hipStreamWaitEvent(st, e, 0) != hipSuccess) {
            sage_set_error("pipe_join: event record / wait failed");
            return SAGE_ELAUNCH;
        }
    }
    return SAGE_OK;
}

// Fork: make every role stream wait for `stream` ("the inputs written there are ready").
extern "C" int sage_pipe_fork(sage_pipe_t* p, sage_stream_t stream) {
    SAGE_REQUIRE(p, "pipe_fork: NULL pipe");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e = p->ev_fork;
    bool recorded = false;
    for (int r = 0; r < 4; ++r) {
        if (p->st[r] == st) continue;
        bool seen = false;
        for (int q = 0; q < r; ++q) seen |= p->st[q] == p->st[r];
        if (seen) continue;
        if (!recorded && hipEventRecord(e, st) != hipSuccess) { sage_set_error("pipe_fork: hipEventRecord failed"); return SAGE_ELAUNCH; }
        recorded = true;
        if (hipStreamWaitEvent(p->st[r], e, 0) != hipSuccess) { sage_set_error("pipe_fork: hipStreamWaitEvent failed"); return SAGE_ELAUNCH; }
    }
    return SAGE_OK;
}
