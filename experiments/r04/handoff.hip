// What does a stream-to-stream hand-off cost between two kernels, by primitive?  A chain of 2N kernels alternating between two streams, each
// waiting for the previous one: (time per kernel) - (the same chain on ONE stream) = the hand-off.
//   0 one stream (baseline)            1 hipEvent, DisableTiming | DisableSystemFence (the product's)     2 hipEvent, DisableTiming only
//   3 hipStreamWriteValue32 / hipStreamWaitValue32 on device memory        4 the same on signal memory (hipMallocSignalMemory)
//   5 one-thread signal / gate kernels on a device counter (round 4's "flags")
//   6 the kernel's OWN completion signal as the event (hipExtLaunchKernelGGL stopEvent) + hipStreamWaitEvent: no record packet      7 the same, events with the system fence
// build: hipcc -O3 --offload-arch=gfx950 handoff.hip -o handoff        run: ./handoff [kernel_us] [n]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void busy(unsigned long long ticks, float* sink) {           // 100 MHz counter
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) sink[blockIdx.x] = (float)ticks;
}
__global__ void sig(unsigned long long* f, unsigned long long v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void gate(const unsigned long long* f, unsigned long long need) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) break;          // 1 s: never hang the queue
    }
}

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const double kus = argc > 1 ? atof(argv[1]) : 20.0;
    const int n = argc > 2 ? atoi(argv[2]) : 200;
    const int blocks = argc > 3 ? atoi(argv[3]) : 256;
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    float* sink; CK(hipMalloc(&sink, 4096 * sizeof(float)));
    unsigned long long* flag; CK(hipMalloc(&flag, 256)); CK(hipMemset(flag, 0, 256));
    uint32_t* val_dev; CK(hipMalloc(&val_dev, 256)); CK(hipMemset(val_dev, 0, 256));
    uint32_t* val_sig = nullptr;
    if (hipExtMallocWithFlags((void**)&val_sig, 8, hipMallocSignalMemory) != hipSuccess) { val_sig = nullptr; (void)hipGetLastError(); }
    int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("kernel %.0f us x %d blocks, chain of %d; hipDeviceAttributeCanUseStreamWaitValue = %d, signal memory %s\n", kus, blocks, 2 * n, can, val_sig ? "ok" : "unavailable");
    const unsigned long long ticks = (unsigned long long)(kus * 100.0);
    hipEvent_t ev[2][2];
    for (int i = 0; i < 2; ++i) {
        CK(hipEventCreateWithFlags(&ev[0][i], hipEventDisableTiming | hipEventDisableSystemFence));
        CK(hipEventCreateWithFlags(&ev[1][i], hipEventDisableTiming));
    }
    const char* names[8] = {"one stream", "hipEvent (no timing, no system fence)", "hipEvent (no timing)", "hipStreamWrite/WaitValue32, device memory",
                            "hipStreamWrite/WaitValue32, signal memory", "signal / gate kernels",
                            "stopEvent of the launch + wait (no fence)", "stopEvent of the launch + wait"};
    double base = 0;
    unsigned long long seq = 0; uint32_t vseq = 0;
    for (int mode = 0; mode < 8; ++mode) {
        if ((mode == 3 || mode == 4) && !can) { printf("%-45s unsupported\n", names[mode]); continue; }
        if (mode == 4 && !val_sig) { printf("%-45s no signal memory\n", names[mode]); continue; }
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 2 * n; ++i) {
                const int me = mode == 0 ? 0 : (i & 1), other = me ^ 1;
                if (i > 0) {                                  // wait for kernel i-1 (on the other stream)
                    if (mode == 1 || mode == 2) CK(hipStreamWaitEvent(s[me], ev[mode - 1][other], 0));
                    else if (mode == 6 || mode == 7) CK(hipStreamWaitEvent(s[me], ev[mode - 6][other], 0));
                    else if (mode == 3) CK(hipStreamWaitValue32(s[me], val_dev, vseq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    else if (mode == 4) CK(hipStreamWaitValue32(s[me], val_sig, vseq, hipStreamWaitValueGte, 0xFFFFFFFFu));
                    else if (mode == 5) hipLaunchKernelGGL(gate, dim3(1), dim3(1), 0, s[me], flag, seq);
                }
                if (mode == 6 || mode == 7) hipExtLaunchKernelGGL(busy, dim3(blocks), dim3(256), 0, s[me], nullptr, ev[mode - 6][me], 0u, ticks, sink);
                else hipLaunchKernelGGL(busy, dim3(blocks), dim3(256), 0, s[me], ticks, sink);
                if (mode == 1 || mode == 2) CK(hipEventRecord(ev[mode - 1][me], s[me]));
                else if (mode == 3) CK(hipStreamWriteValue32(s[me], val_dev, ++vseq, 0));
                else if (mode == 4) CK(hipStreamWriteValue32(s[me], val_sig, ++vseq, 0));
                else if (mode == 5) hipLaunchKernelGGL(sig, dim3(1), dim3(1), 0, s[me], flag, ++seq);
            }
            const auto t1 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            const auto t2 = std::chrono::steady_clock::now();
            const double per = std::chrono::duration<double, std::micro>(t2 - t0).count() / (2 * n);
            const double host = std::chrono::duration<double, std::micro>(t1 - t0).count() / (2 * n);
            if (per < best) best = per;
            if (rep == 3) printf("%-45s %7.2f us per kernel (host enqueue %5.2f)", names[mode], best, host);
        }
        if (mode == 0) base = best;
        printf("   hand-off = %+6.2f us\n", best - base);
    }
    return 0;
}
