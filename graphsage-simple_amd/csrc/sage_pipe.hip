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
//
// Host enqueue threads (round 3, sage_pipe_set_threads).  One batch is thirteen HIP calls (five launches, four records, four
// waits): 40-60 us of host time on one thread, next to a 59 us period on the GPU -- the host was as good as the pacemaker, and in a
// short timed region (20 steps) the GPU chased the host through the whole region.  With the threads on, sage_pipe_submit only POSTS a
// descriptor (seeds, key, out, slot); role r's own thread makes role r's calls on role r's stream, so the four streams are fed in
// parallel.  HIP's event semantics fix the order of the HOST calls across threads: hipStreamWaitEvent(ev) means "the most recent
// hipEventRecord(ev) made before this call", so thread r may touch batch b only after thread r-1 has recorded for batch b (and S(b)
// after L(b - depth)): one atomic counter per role (`done[r]` = batches whose calls role r has made), acquire / release.  The same
// chain guarantees that an event is re-recorded (batch b + depth) only after its consumer's wait for batch b has been made.
// Everything that looks at the streams from outside (join, fork, reset, update_weights, destroy) first drains the posted batches
// (sage_pipe_flush).  Needs four distinct role streams; not available inside a stream capture.
//
// Express lane (round 4).  A hand-off between two role streams is a record + wait packet pair, ~11 us from the producer's end to the
// consumer's start (rocprofv3 trace).  In steady state the other streams' kernels run in those gaps; a batch submitted to
// an IDLE pipeline pays all four of them with the chip empty: its output arrives after ~141 us where the five kernels need 86 back to back,
// and the batches behind it inherit the delay.  So a batch that finds every earlier batch gone (the last batch's layer-2 event has
// completed -- one hipEventQuery) is enqueued whole on stream L, like sage_forward2; the next batch starts on the role streams at once,
// beside it.  Same kernels, same workspace, same arguments: bit-identical.  Not while capturing (an event query is not capturable);
// needs four distinct role streams; SAGE_PIPE_EXPRESS=0 turns it off (A/B).
#include <sched.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>

#include "sage_internal.h"

namespace {
enum { kRing = 256, kDoneEvents = 32 };
struct pipe_desc {                                      // one posted batch
    const int32_t* seeds;
    uint64_t key;
    float* out;
    int64_t ldo;
    int slot;
    bool fresh;
    bool express;                                       // an IDLE pipeline's batch: all five launches on stream L, no hand-off (see submit_one)
    void* gev[2];
};
}  // namespace

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
    int64_t express_count;                              // batches that took the express lane (submitting thread only)
    // host enqueue threads
    int device;
    bool threaded;
    int window;                                         // > 0: S(b) is enqueued only once batch b - window has LEFT the GPU (host run-ahead bound)
    std::thread th[4];
    std::atomic<uint64_t> posted{0};                    // batches posted by the caller (threaded mode: == submitted)
    std::atomic<uint64_t> done[4];                      // batches whose calls role r's thread has made
    std::atomic<int> stop{0};
    std::atomic<int> sleepers{0};
    std::atomic<int> worker_rc{0};
    std::mutex mu;
    std::condition_variable cv;
    char worker_err[512];
    pipe_desc ring[kRing];
    hipEvent_t ev_done[kDoneEvents];                    // window > 0: recorded on stream L behind batch b (index b % kDoneEvents)
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

// A producer that has already FINISHED needs no wait packet in the consumer's queue: the event is asked first (its most recent record
// is the producer's record for this batch: the host order of the calls guarantees it) and a completed one is skipped.  In steady state
// that is every S -> G hand-off (the samplers run batches ahead of the gather): one barrier packet less per batch in the gather's queue.
// Never while capturing (an event query is not a capturable operation).  SAGE_PIPE_QUERY=0 turns it off (A/B).
// Measured (experiments/r03/call43.sh): nothing in steady state, where the host runs batches ahead of the GPU and finds no event
// complete (59.1 vs 59.2 us); 64.9 vs 65.8 us in the 20-step form, whose first batches are enqueued on an idle GPU.
// (Making it ALWAYS true -- role threads that enqueue only once their producer has finished, so no queue ever holds a wait packet --
// was measured too: 60.3 vs 59.0 us, call44.sh: the wait packets are not what separates consecutive kernels of a stream.)
static bool query_first() {
    static const bool on = [] { const char* v = getenv("SAGE_PIPE_QUERY"); return !(v && *v == '0'); }();
    return on;
}
int wait_on(sage_pipe* p, int consumer, int producer, int slot, bool capturing = false) {
    if (p->st[consumer] == p->st[producer]) return SAGE_OK;          // stream order already says it
    if (!capturing && query_first() && hipEventQuery(p->ev[producer][slot]) == hipSuccess) return SAGE_OK;
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
    p->express_count = 0;
    p->threaded = false;
    p->window = 0;
    p->device = 0;
    p->worker_err[0] = 0;
    (void)hipGetDevice(&p->device);
    for (int r = 0; r < 4; ++r) p->done[r].store(0);
    for (int i = 0; i < kDoneEvents; ++i) p->ev_done[i] = nullptr;
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

static void stop_threads(sage_pipe* p);

extern "C" int sage_pipe_destroy(sage_pipe_t* p) {
    if (!p) return SAGE_OK;
    stop_threads(p);
    for (int i = 0; i < kDoneEvents; ++i)
        if (p->ev_done[i]) (void)hipEventDestroy(p->ev_done[i]);
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < p->depth; ++i)
            if (p->ev[r][i]) (void)hipEventDestroy(p->ev[r][i]);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    delete p;
    return SAGE_OK;
}

extern "C" int sage_pipe_update_weights(sage_pipe_t* p, const float* w1, const float* w2, const void* w1_prepared) {
    SAGE_REQUIRE(p && w1 && w2, "pipe_update_weights: NULL argument");
    if (int rc = sage_pipe_flush(p)) return rc;          // the role threads read p->model
    p->model.w1 = w1;
    p->model.w2 = w2;
    p->model.w1_prepared = w1_prepared;
    return SAGE_OK;
}

// Role r's calls for one batch, on role r's stream: wait for the producer's event, launch, record.  `fresh`: no earlier submit of
// this pipe is outstanding on the slot (a fresh pipe, or the first `depth` submits after the caller joined and synchronised
// everything before): the workspace-release wait is skipped.  Called for r = S, G, D, L in turn by the caller's thread
// (submit_one), or by role r's own thread.
static int role_enqueue(sage_pipe* p, int r, const pipe_desc& d, unsigned long long cap) {
    const sage_model_t* m = &p->model;
    const int slot = d.slot;
    void* ws = p->ws[slot];
    // the role's hand-off event rides on its stage's last kernel (sage_internal.h: tail event) instead of being recorded behind it -- not while
    // capturing (hipExtLaunchKernel is not a capturable launch), and SAGE_PIPE_TAIL=0 keeps the separate record (A/B)
    static const bool tail_on = [] { const char* v = getenv("SAGE_PIPE_TAIL"); return !(v && *v == '0'); }();
    auto tail = [&](int role, bool needed) -> void* { return (tail_on && cap == 0 && needed) ? (void*)p->ev[role][slot] : nullptr; };
    if (d.express) {
        // the pipeline was idle at submit: no release to wait for, nothing to hand over; roles S, G, D have no calls to make
        if (r != RL) return SAGE_OK;
        const sage_ext_launch_t x{d.gev[0], d.gev[1]};
        if (d.gev[0] && d.gev[1]) sage_ext_launch = &x;
        const int rc = sage_forward2_launch_stages(m, ws, p->ws_bytes, d.seeds, p->batch, d.key, d.out, d.ldo, SAGE_STAGE_ALL, p->st[RL],
                                                   tail(RL, true));
        sage_ext_launch = nullptr;
        if (rc) return rc;
        return tail(RL, true) ? SAGE_OK : record(p, RL, slot, true);
    }
    switch (r) {
    case RS:
        // S: outer + inner sample.  Needs the slot's previous batch to have left layer 2 (its last block zeroes the counters).
        if (!d.fresh && p->st[RS] != p->st[RL]) {
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
                if (int rc = wait_on(p, RS, RL, slot, cap != 0)) return rc;
            }
        }
#ifndef SAGE_PIPE_SKIP_S   // diagnostic builds only (experiments/ab_build.sh).  _G and _D may be skipped alone (stale data downstream); _S and _L
                           // only together with everything else ("events only"): the samplers fill and layer 2 wipes the frontier hash, and
                           // one without the other leaves a full table behind (an outer sampler probing it took 67 ms per batch)
        if (int rc = sage_forward2_launch_stages(m, ws, p->ws_bytes, d.seeds, p->batch, d.key, nullptr, 0,
                                                 SAGE_STAGE_SAMPLE_OUTER | SAGE_STAGE_SAMPLE_INNER, p->st[RS], tail(RS, p->st[RG] != p->st[RS])))
            return rc;
        if (tail(RS, p->st[RG] != p->st[RS])) return SAGE_OK;
#endif
        return record(p, RS, slot, p->st[RG] != p->st[RS]);
    case RG:
        // G: the layer-1 gather (nothing to launch when layer 1 is a one-launch layer; D then waits on S through G's stream order)
        if (int rc = wait_on(p, RG, RS, slot, cap != 0)) return rc;
#ifndef SAGE_PIPE_SKIP_G   // diagnostic builds (experiments/ab_build.sh): the pipeline without one of its stages' kernels, stale data downstream
        {
            // profiled submit: the two caller-owned timing events become the gather launch's OWN start / stop events (sage_gather.hip)
            const sage_ext_launch_t x{d.gev[0], d.gev[1]};
            if (d.gev[0] && d.gev[1]) sage_ext_launch = &x;
            const int rc = sage_forward2_launch_stages(m, ws, p->ws_bytes, d.seeds, p->batch, d.key, nullptr, 0, SAGE_STAGE_GATHER1, p->st[RG],
                                                       tail(RG, p->st[RD] != p->st[RG]));
            sage_ext_launch = nullptr;
            if (rc) return rc;
            if (tail(RG, p->st[RD] != p->st[RG])) return SAGE_OK;
        }
#endif
        return record(p, RG, slot, p->st[RD] != p->st[RG]);
    case RD:
        // D: the contraction (or the whole fused layer 1)
        if (int rc = wait_on(p, RD, RG, slot, cap != 0)) return rc;
#ifndef SAGE_PIPE_SKIP_D
        if (int rc = sage_forward2_launch_stages(m, ws, p->ws_bytes, d.seeds, p->batch, d.key, nullptr, 0, SAGE_STAGE_CONTRACT1, p->st[RD],
                                                 tail(RD, p->st[RL] != p->st[RD])))
            return rc;
        if (tail(RD, p->st[RL] != p->st[RD])) return SAGE_OK;
#endif
        return record(p, RD, slot, p->st[RL] != p->st[RD]);
    default:
        // L: layer 2; afterwards the workspace is clean again
        if (int rc = wait_on(p, RL, RD, slot, cap != 0)) return rc;
#ifndef SAGE_PIPE_SKIP_L
        if (int rc = sage_forward2_launch_stages(m, ws, p->ws_bytes, d.seeds, p->batch, d.key, d.out, d.ldo, SAGE_STAGE_LAYER2, p->st[RL],
                                                 tail(RL, p->st[RS] != p->st[RL])))
            return rc;
        if (tail(RL, p->st[RS] != p->st[RL])) return SAGE_OK;
#endif
        return record(p, RL, slot, p->st[RS] != p->st[RL]);
    }
}

// ---- host enqueue threads ------------------------------------------------------------------------------------------------
namespace {
inline void cpu_relax() { __builtin_ia32_pause(); }

// spin (pause, then yield) until pred() or the pipe stops; false = stopped
template <class Pred>
bool spin_until(sage_pipe* p, Pred pred) {
    for (unsigned n = 0; !pred(); ++n) {
        if (p->stop.load(std::memory_order_relaxed)) return false;
        if (n < 4096) cpu_relax(); else sched_yield();
    }
    return true;
}

void worker_fail(sage_pipe* p, int rc) {
    int expected = 0;
    if (p->worker_rc.compare_exchange_strong(expected, rc)) {
        std::lock_guard<std::mutex> g(p->mu);
        snprintf(p->worker_err, sizeof(p->worker_err), "%s", sage_last_error());
    }
}

// how long an idle role thread spins before it sleeps (SAGE_PIPE_SPIN_US, default 2000, 0 = sleep at once)
int spin_us() {
    static const int v = [] { const char* e = getenv("SAGE_PIPE_SPIN_US"); const int x = e ? atoi(e) : 2000; return x < 0 ? 0 : x > 1000000 ? 1000000 : x; }();
    return v;
}

void role_thread(sage_pipe* p, int r) {
    (void)hipSetDevice(p->device);
    uint64_t b = p->done[r].load(std::memory_order_relaxed);
    for (;;) {
        // a posted batch: spin for spin_us() microseconds (default 2000: a running pipeline posts every few tens of microseconds, and a caller
        // that fences between two regions -- flush, device synchronize, barrier -- is back within a millisecond or two; a thread that went
        // to sleep costs the next region's first batch a ~50 us condition-variable wake-up), then sleep.  Until round 3 the bound was a
        // count of 200 000 `pause` instructions, 3-4 ms on the EPYC hosts, not the "few tens of microseconds" DESIGN claimed.
        bool have = false;
        const auto t_spin = std::chrono::steady_clock::now();
        for (unsigned n = 0;; ++n) {
            if (p->posted.load(std::memory_order_acquire) > b) { have = true; break; }
            if (p->stop.load(std::memory_order_relaxed)) return;
            cpu_relax();
            if ((n & 255u) == 255u && std::chrono::steady_clock::now() - t_spin > std::chrono::microseconds(spin_us())) break;
        }
        if (!have) {
            std::unique_lock<std::mutex> lk(p->mu);
            p->sleepers.fetch_add(1);
            while (p->posted.load(std::memory_order_acquire) <= b && !p->stop.load(std::memory_order_relaxed))
                p->cv.wait_for(lk, std::chrono::milliseconds(2));
            p->sleepers.fetch_sub(1);
            if (p->stop.load(std::memory_order_relaxed)) return;
        }
        const pipe_desc& d = p->ring[b % kRing];
        // the HOST order HIP's event semantics need: the producer's record for this batch has been made
        if (r == RS) {
            if (!d.fresh && !spin_until(p, [&] { return p->done[RL].load(std::memory_order_acquire) + (uint64_t)p->depth > b; })) return;
            if (p->window > 0 && b >= (uint64_t)p->window) {
                // bound the host's run-ahead: batch b - window has left the GPU (its marker on stream L has completed).  Role L records
                // that marker with its calls for batch b - window: wait until those have been made (window < depth: they may not yet)
                if (!spin_until(p, [&] { return p->done[RL].load(std::memory_order_acquire) + (uint64_t)p->window > b; })) return;
                hipEvent_t e = p->ev_done[(b - (uint64_t)p->window) % kDoneEvents];
                if (!spin_until(p, [&] { return hipEventQuery(e) != hipErrorNotReady; })) return;
            }
        } else if (!spin_until(p, [&] { return p->done[r - 1].load(std::memory_order_acquire) > b; })) {
            return;
        }
        if (p->worker_rc.load(std::memory_order_relaxed) == 0) {      // after a failure nothing more is enqueued; the counters still advance
            int rc = role_enqueue(p, r, d, 0);
            if (rc == SAGE_OK && r == RL && p->window > 0 && hipEventRecord(p->ev_done[b % kDoneEvents], p->st[RL]) != hipSuccess) {
                sage_set_error("pipe: hipEventRecord failed");
                rc = SAGE_ELAUNCH;
            }
            if (rc != SAGE_OK) worker_fail(p, rc);
        }
        ++b;
        p->done[r].store(b, std::memory_order_release);
    }
}

int post(sage_pipe* p, const pipe_desc& d) {
    // the ring holds the descriptors the role threads have not finished with
    if (!spin_until(p, [&] { return p->submitted - p->done[RL].load(std::memory_order_acquire) < (uint64_t)kRing - 1; })) return SAGE_ELAUNCH;
    p->ring[p->submitted % kRing] = d;
    ++p->submitted;
    p->posted.store(p->submitted, std::memory_order_seq_cst);
    if (p->sleepers.load(std::memory_order_seq_cst) > 0) {
        { std::lock_guard<std::mutex> g(p->mu); }
        p->cv.notify_all();
    }
    return SAGE_OK;
}
}  // namespace

static void stop_threads(sage_pipe* p) {
    if (!p->threaded) return;
    (void)sage_pipe_flush(p);
    p->stop.store(1);
    { std::lock_guard<std::mutex> g(p->mu); }
    p->cv.notify_all();
    for (int r = 0; r < 4; ++r)
        if (p->th[r].joinable()) p->th[r].join();
    p->stop.store(0);
    p->threaded = false;
}

// Every posted batch has been enqueued on the role streams (NOT: has run).  Returns the first error a role thread met.
extern "C" int sage_pipe_flush(sage_pipe_t* p) {
    SAGE_REQUIRE(p, "pipe_flush: NULL pipe");
    if (!p->threaded) return SAGE_OK;
    for (unsigned n = 0; p->done[RL].load(std::memory_order_acquire) < p->submitted; ++n) {
        if (n < 4096) cpu_relax(); else sched_yield();
    }
    const int rc = p->worker_rc.load();
    if (rc != SAGE_OK) {
        std::lock_guard<std::mutex> g(p->mu);
        sage_set_error("pipe (role thread): %s", p->worker_err);
    }
    return rc;
}

// on != 0: start one host enqueue thread per role; on == 0: drain and stop them.  window > 0 bounds the host's run-ahead: role S
// enqueues batch b only when batch b - window has left the GPU (0 = unbounded).
extern "C" int sage_pipe_set_threads(sage_pipe_t* p, int32_t on, int32_t window) {
    SAGE_REQUIRE(p, "pipe_set_threads: NULL pipe");
    SAGE_REQUIRE(window >= 0 && window < kDoneEvents, "pipe_set_threads: window = %d outside [0, %d)", window, (int)kDoneEvents);
    stop_threads(p);
    if (!on) return SAGE_OK;
    for (int r = 0; r < 4; ++r)
        for (int q = 0; q < r; ++q) SAGE_REQUIRE(p->st[q] != p->st[r], "pipe_set_threads: the host enqueue threads need four distinct role streams");
    unsigned long long cap = 0;
    if (int rc = capture_id(p->st[RS], &cap)) return rc;
    SAGE_REQUIRE(cap == 0, "pipe_set_threads: not inside a stream capture");
    if (window > 0)
        for (int i = 0; i < kDoneEvents; ++i)
            if (!p->ev_done[i] && hipEventCreateWithFlags(&p->ev_done[i], kEventFlags) != hipSuccess) {
                sage_set_error("pipe_set_threads: hipEventCreate failed");
                return SAGE_ELAUNCH;
            }
    p->window = window;
    p->worker_rc.store(0);
    p->posted.store(p->submitted);
    for (int r = 0; r < 4; ++r) p->done[r].store(p->submitted);
    p->threaded = true;
    for (int r = 0; r < 4; ++r) p->th[r] = std::thread(role_thread, p, r);
    return SAGE_OK;
}

// Every batch submitted so far has left layer 2 (and with it the GPU: L is a batch's last stage, stream L runs them in order).
static bool express_on() {
    static const bool on = [] { const char* v = getenv("SAGE_PIPE_EXPRESS"); return !(v && *v == '0'); }();
    return on;
}
static bool pipe_idle(sage_pipe* p) {
    if (!express_on()) return false;
    for (int r = 0; r < 4; ++r)
        for (int q = 0; q < r; ++q)
            if (p->st[q] == p->st[r]) return false;           // coinciding role streams skip records: nothing to ask
    if (p->submitted == 0) return true;                       // a new / reset pipe: the caller has synchronised what came before
    if (p->threaded && p->done[RL].load(std::memory_order_acquire) < p->submitted) return false;   // the last batch's record has not been made yet
    return hipEventQuery(p->ev[RL][(p->submitted - 1) % (uint64_t)p->depth]) == hipSuccess;
}

// One batch through the four role streams.
static int submit_one(sage_pipe* p, const int32_t* seeds, uint64_t key, float* out, int64_t ldo, bool fresh_slot,
                      void* const* gather_events = nullptr) {
    pipe_desc d;
    d.seeds = seeds; d.key = key; d.out = out; d.ldo = ldo;
    d.slot = (int)(p->submitted % (uint64_t)p->depth);
    d.fresh = fresh_slot;
    d.express = false;
    d.gev[0] = gather_events ? gather_events[0] : nullptr;
    d.gev[1] = gather_events ? gather_events[1] : nullptr;
    const int slot = d.slot;
    if (p->threaded) {
        SAGE_REQUIRE(p->cap_id[slot] == 0, "pipe: this slot's last batch was submitted inside a stream capture: call sage_pipe_reset first");
        if (int rc = p->worker_rc.load()) { sage_set_error("pipe: a role thread failed earlier (sage_pipe_flush reports it)"); return rc; }
        d.express = pipe_idle(p);
        p->express_count += d.express ? 1 : 0;
        return post(p, d);
    }
    unsigned long long cap = 0;
    if (int rc = capture_id(p->st[RS], &cap)) return rc;
    d.express = cap == 0 && p->cap_id[slot] == 0 && pipe_idle(p);
    p->express_count += d.express ? 1 : 0;
    for (int r = 0; r < 4; ++r)
        if (int rc = role_enqueue(p, r, d, cap)) return rc;
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

extern "C" int64_t sage_pipe_express_count(const sage_pipe_t* p) { return p ? p->express_count : -1; }

// Forget every submit: the next `depth` submits find their workspaces free.  The caller has synchronised (or joined) everything
// submitted before -- e.g. after a stream capture ended, before eager submission resumes on the same pipe.
extern "C" int sage_pipe_reset(sage_pipe_t* p) {
    SAGE_REQUIRE(p, "pipe_reset: NULL pipe");
    if (int rc = sage_pipe_flush(p)) return rc;
    p->submitted = 0;
    if (p->threaded) {
        // the role threads are idle (everything posted has been enqueued) and wait for posted > done: restart their numbering
        const int window = p->window;
        stop_threads(p);
        for (int i = 0; i < SAGE_PIPE_MAX_DEPTH; ++i) { p->cap_count[i] = 0; p->cap_id[i] = 0; }
        return sage_pipe_set_threads(p, 1, window);
    }
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
    if (int rc = sage_pipe_flush(p)) return rc;
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
    if (int rc = sage_pipe_flush(p)) return rc;
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
