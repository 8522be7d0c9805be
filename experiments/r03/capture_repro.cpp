// Minimal stand-alone pattern of the role pipeline's stream capture: which ingredient makes
// hipStreamEndCapture crash on ROCm 7.2?  No torch, no libsage355: four role streams, trivial kernels, hipEvents.
//
//   hipcc -O2 --offload-arch=gfx950 experiments/r03/capture_repro.cpp -o experiments/r03/capture_repro
//   experiments/r03/capture_repro <pattern>  # one pattern per process (capture_repro.sh loops over them and reports exit status / signal)
//
// Patterns (batches = 8, depth = 4 unless said otherwise; S(b) -> G(b) -> D(b) -> L(b) -> S(b + depth), fork from and join to the
// origin stream, as sage_pipe_fork / submit_many / sage_pipe_join do):
//   0  single stream, 20 kernels                                   (sanity)
//   1  four streams, a FRESH event for every record                (no event is recorded twice inside the capture)
//   2  four streams, events ev[role][slot] re-recorded             (the product's pattern: slot = b % depth)
//   3  pattern 1 + the join re-records an event that was already recorded in the capture (sage_pipe_join)
//   4  pattern 2 with hipEventDisableSystemFence events            (the product's event flags)
//   5  pattern 1 with hipEventDisableSystemFence events
//   6  pattern 2, batches = depth (no event re-recorded, but the wait S(b) <- L(b - depth) never appears)
//   7  pattern 1, relaxed capture mode (hipStreamCaptureModeRelaxed) instead of global
// Second series (after 1-5 and 7 crashed inside hipStreamEndCapture while 0 and 6 passed): what separates 6 from 1?
//   8  pattern 1 WITHOUT the workspace-release wait S(b) <- L(b - depth)    (same 40 kernels: is it the graph's size?)
//   9  pattern 1, batches = depth + 1                                        (exactly one S <- L wait)
//  10  two forked streams A, B: A: k, e1 | B: wait e1, k, e2 | A: wait e2, k | origin joins both   (minimal "wait back" between
//      two forked streams)
//  11  the same with A = the origin stream                                   (the shape of a fork + join, which pattern 6 has too)
//  12  pattern 10, but B's event is also waited for by the origin BEFORE A waits for it
// Third series (8 passed, 9 crashed, 10-12 passed): in 9 the event S waits for, L(0)'s, is no longer the TAIL of stream L's
// captured work (L(1) .. L(3) were enqueued after it); every other wait of every pattern targets the producer's tail.
//  13  two forked streams: A: k1, record e1, k2 | B: wait e1, k        (minimal wait on an INTERIOR event)
//  14  pattern 9, the release relayed through a helper stream: right after L(b) is recorded a fresh stream H_b waits for it and
//      records h_b (H_b gets no more work, so h_b stays its tail); S(b + depth) waits for h_b
//  15  pattern 9, the release edge without an event: hipStreamGetCaptureInfo_v2 on stream L right after L(b) gives its node;
//      before S(b + depth), hipStreamUpdateCaptureDependencies(S, {that node}, add)
//  16  pattern 1 (8 batches) with the relay of 14
//  17  pattern 1 (8 batches) with the explicit dependency of 15
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("  %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);              \
            fflush(stdout);                                                                     \
            return 10;                                                                          \
        }                                                                                       \
    } while (0)

__global__ void bump(int* p, int v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, v);
}

struct Events {
    unsigned flags;
    bool fresh;                     // a new event per record
    hipEvent_t fixed[4][8];
    std::vector<hipEvent_t> pool;
    int init(unsigned f, bool fr) {
        flags = f; fresh = fr;
        for (int r = 0; r < 4; ++r)
            for (int s = 0; s < 8; ++s) CK(hipEventCreateWithFlags(&fixed[r][s], flags));
        return 0;
    }
    hipEvent_t last[4][8];
    int record(int role, int slot, hipStream_t st) {
        hipEvent_t e = fixed[role][slot];
        if (fresh) { CK(hipEventCreateWithFlags(&e, flags)); pool.push_back(e); }
        last[role][slot] = e;
        CK(hipEventRecord(e, st));
        return 0;
    }
};

static int run(int pattern) {
    int* d = nullptr;
    CK(hipMalloc(&d, sizeof(int)));
    CK(hipMemset(d, 0, sizeof(int)));
    hipStream_t origin, st[4];
    CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
    for (int r = 0; r < 4; ++r) CK(hipStreamCreateWithFlags(&st[r], hipStreamNonBlocking));
    hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, origin, d, 0);      // module load outside the capture
    CK(hipDeviceSynchronize());
    const hipStreamCaptureMode mode = pattern == 7 ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeGlobal;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int expect = 0;
    if (pattern == 13) {
        hipEvent_t f, e1, e3, e4;
        for (hipEvent_t* e : {&f, &e1, &e3, &e4}) CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        hipStream_t A = st[0], B = st[1];
        CK(hipStreamBeginCapture(origin, mode));
        CK(hipEventRecord(f, origin));
        CK(hipStreamWaitEvent(A, f, 0));
        CK(hipStreamWaitEvent(B, f, 0));
        hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, A, d, 1); ++expect;
        CK(hipEventRecord(e1, A));
        hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, A, d, 1); ++expect;      // A moves on: e1 is now interior
        CK(hipStreamWaitEvent(B, e1, 0));
        hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, B, d, 1); ++expect;
        CK(hipEventRecord(e3, A)); CK(hipStreamWaitEvent(origin, e3, 0));
        CK(hipEventRecord(e4, B)); CK(hipStreamWaitEvent(origin, e4, 0));
        printf("  end capture ...\n"); fflush(stdout);
        CK(hipStreamEndCapture(origin, &graph));
    } else if (pattern >= 10 && pattern <= 12) {
        hipEvent_t f, e1, e2, e3, e4;
        for (hipEvent_t* e : {&f, &e1, &e2, &e3, &e4}) CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        hipStream_t A = pattern == 11 ? origin : st[0], B = st[1];
        CK(hipStreamBeginCapture(origin, mode));
        CK(hipEventRecord(f, origin));
        if (A != origin) CK(hipStreamWaitEvent(A, f, 0));
        CK(hipStreamWaitEvent(B, f, 0));
        hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, A, d, 1); ++expect;
        CK(hipEventRecord(e1, A));
        CK(hipStreamWaitEvent(B, e1, 0));
        hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, B, d, 1); ++expect;
        CK(hipEventRecord(e2, B));
        if (pattern == 12) CK(hipStreamWaitEvent(origin, e2, 0));
        CK(hipStreamWaitEvent(A, e2, 0));
        hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, A, d, 1); ++expect;
        if (A != origin) { CK(hipEventRecord(e3, A)); CK(hipStreamWaitEvent(origin, e3, 0)); }
        CK(hipEventRecord(e4, B));
        CK(hipStreamWaitEvent(origin, e4, 0));
        printf("  end capture ...\n"); fflush(stdout);
        CK(hipStreamEndCapture(origin, &graph));
    } else if (pattern == 0) {
        CK(hipStreamBeginCapture(origin, mode));
        for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, origin, d, 1); ++expect; }
        printf("  end capture ...\n"); fflush(stdout);
        CK(hipStreamEndCapture(origin, &graph));
    } else {
        const bool fresh = pattern == 1 || pattern == 3 || pattern == 5 || pattern >= 7;   // 14-17 too
        const unsigned flags = (pattern == 4 || pattern == 5) ? (hipEventDisableTiming | hipEventDisableSystemFence) : hipEventDisableTiming;
        const int depth = 4, batches = pattern == 6 ? depth : (pattern == 9 || pattern == 14 || pattern == 15) ? depth + 1 : 8;
        const bool release_wait = pattern != 8;
        const bool relay = pattern == 14 || pattern == 16, explicit_dep = pattern == 15 || pattern == 17;
        hipEvent_t relay_ev[8];
        hipStream_t relay_st[8];
        hipGraphNode_t lnode[8];
        for (int i = 0; i < 8; ++i) { relay_ev[i] = nullptr; relay_st[i] = nullptr; lnode[i] = nullptr; }
        Events ev;
        if (int rc = ev.init(flags, fresh)) return rc;
        hipEvent_t fork_ev;
        CK(hipEventCreateWithFlags(&fork_ev, flags));
        CK(hipStreamBeginCapture(origin, mode));
        CK(hipEventRecord(fork_ev, origin));
        for (int r = 0; r < 4; ++r) CK(hipStreamWaitEvent(st[r], fork_ev, 0));
        for (int b = 0; b < batches; ++b) {
            const int slot = b % depth;
            for (int r = 0; r < 4; ++r) {
                // consumer waits on producer: S <- L of the slot's previous batch, G <- S, D <- G, L <- D
                if (r == 0) {
                    if (b >= depth && release_wait) {
                        if (relay) CK(hipStreamWaitEvent(st[0], relay_ev[b - depth], 0));
                        else if (explicit_dep) CK(hipStreamUpdateCaptureDependencies(st[0], &lnode[b - depth], 1, hipStreamAddCaptureDependencies));
                        else CK(hipStreamWaitEvent(st[0], ev.last[3][slot], 0));
                    }
                }
                else CK(hipStreamWaitEvent(st[r], ev.last[r - 1][slot], 0));
                hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, st[r], d, 1); ++expect;
                if (r == 0) { hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, st[r], d, 1); ++expect; }     // two sampler kernels
                if (int rc = ev.record(r, slot, st[r])) return rc;
                if (r == 3 && relay && b + depth < batches) {          // L(b) is the tail of stream L right now
                    CK(hipStreamCreateWithFlags(&relay_st[b], hipStreamNonBlocking));
                    CK(hipEventCreateWithFlags(&relay_ev[b], flags));
                    CK(hipStreamWaitEvent(relay_st[b], ev.last[3][slot], 0));
                    CK(hipEventRecord(relay_ev[b], relay_st[b]));
                }
                if (r == 3 && explicit_dep) {
                    hipStreamCaptureStatus cs; unsigned long long id = 0; hipGraph_t g = nullptr; const hipGraphNode_t* deps = nullptr; size_t nd = 0;
                    CK(hipStreamGetCaptureInfo_v2(st[3], &cs, &id, &g, &deps, &nd));
                    if (nd != 1) { printf("  stream L has %zu tail nodes\n", nd); return 12; }
                    lnode[b] = deps[0];
                }
            }
        }
        for (int i = 0; i < 8; ++i)                                     // helper streams join the origin too
            if (relay_st[i]) CK(hipStreamWaitEvent(origin, relay_ev[i], 0));
        // join: the origin waits for every role stream's last work
        const int lslot = (batches - 1) % depth;
        for (int r = 0; r < 4; ++r) {
            hipEvent_t e = ev.last[r][lslot];
            if (pattern == 3 || !fresh) CK(hipEventRecord(e, st[r]));     // sage_pipe_join re-records (the role's last record may have been skipped)
            CK(hipStreamWaitEvent(origin, e, 0));
        }
        printf("  end capture ...\n"); fflush(stdout);
        CK(hipStreamEndCapture(origin, &graph));
    }
    printf("  captured; instantiate ...\n"); fflush(stdout);
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    printf("  instantiated; launch x3 ...\n"); fflush(stdout);
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, origin));
    CK(hipStreamSynchronize(origin));
    int got = -1;
    CK(hipMemcpy(&got, d, sizeof(int), hipMemcpyDeviceToHost));
    printf("  counter = %d, expected %d -> %s\n", got, 3 * expect, got == 3 * expect ? "OK" : "WRONG");
    fflush(stdout);
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
    return got == 3 * expect ? 0 : 11;
}

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: capture_repro <pattern 0..17>   (experiments/r03/capture_repro.sh runs them all, one process each)\n"); return 2; }
    return run(atoi(argv[1]));
}
