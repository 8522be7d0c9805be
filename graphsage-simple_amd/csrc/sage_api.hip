// libsage355: error reporting + version entry points (include/sage355.h).
#include <stdarg.h>
#include <string.h>

#include <stdlib.h>

#include <atomic>

#include "sage_internal.h"

static thread_local char g_err[512] = "";

void sage_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int sage_abi_version(void) { return SAGE_ABI_VERSION; }
extern "C" const char* sage_last_error(void) { return g_err; }
extern "C" const char* sage_build_arch(void) { return "gfx950"; }

static int env_int(const char* name, int dflt, int lo, int hi) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    const long x = strtol(v, nullptr, 10);
    return (int)(x < lo ? lo : x > hi ? hi : x);
}

const sage_tunables_t& sage_tunables() {
    static const sage_tunables_t t = [] {
        sage_tunables_t x;
        x.gather_blocks_per_cu = env_int("SAGE_G_PER_CU", 6, 1, 8);
        { const int sl = env_int("SAGE_G_SLICE_LANES", 0, 0, 64); x.gather_slice_lanes = (sl == 8 || sl == 16 || sl == 32 || sl == 64) ? sl : 0; }
        x.gather_rows_in_flight = env_int("SAGE_G_ROWS", 1, 1, 4);
        x.gather_trip = env_int("SAGE_G_TRIP", 16, 8, 16) >= 16 ? 16 : 8;
        x.gather_variant = env_int("SAGE_G_VARIANT", 1, 0, 2);
        x.gather_variant_sliced = env_int("SAGE_G_VARIANT_SM", getenv("SAGE_G_VARIANT") ? x.gather_variant : 2, 0, 2);
        // 224 since the end of round 3 (256 before): the concat encoder's pipeline, whose pacemaker the contraction is, gains 4-5 % (config 3:
        // 88.9 -> 84.5 us per forward, config 5: 84.5 -> 81.5; three / two interleaved runs, experiments/r03/call61.sh), the gcn one is
        // indifferent (58.4 vs 58.4); alone the kernel loses ~1 us (four tile rounds instead of three)
        x.dense_blocks = env_int("SAGE_DENSE_BLOCKS", 224, 32, 512);
        x.bwd_blocks = env_int("SAGE_BWD_BLOCKS", 512, 16, 4096);
        x.bwd_direct_blocks = env_int("SAGE_BWD_DIRECT_BLOCKS", 256, 16, 1024);
        // 512 since round 3: once the slice-major table took the gather off the critical path the sampler stream showed up --
        // 58.9 vs 61.0 us per forward (200 steps), 65.6 vs 67.2 (20 steps), four interleaved runs each (experiments/r03/call30.sh)
        const int so = env_int("SAGE_SO_THREADS", 512, 256, 1024);
        x.outer_threads = so >= 1024 ? 1024 : so >= 512 ? 512 : 256;
        x.tile16_grid = env_int("SAGE_T16_GRID", 2 * kNumCU, 64, 1024);
        x.sample_fused = env_int("SAGE_SAMPLE_FUSED", 0, 0, 1);
        x.tile16_waves = env_int("SAGE_T16_WAVES", 8, 8, 16) >= 16 ? 16 : 8;
        return x;
    }();
    return t;
}

// Device memory fills as a KERNEL, never as hipMemsetAsync: on ROCm 7.2 a small hipMemsetAsync captured into a hipGraph does what it
// says on the FIRST replay and writes garbage (kernel-argument-like pointers) into its 16 bytes from the second replay on
// (experiments/r03/memset_in_graph.py).  That memset zeroed the "W holds a huge value" word behind the prepared weight planes, so
// from its second replay on a captured training step ran layer 1 through the exact fp32 cold path whenever the garbage's first word was
// not 0 -- which is what made captured steps differ from eager ones in the last bits (and from one another, one process in three).
// n_words 32-bit words of value v at p (4-byte aligned).
__global__ void sage_fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, size_t n_words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int sage_fill_u32(void* p, uint32_t v, size_t n_words, hipStream_t st) {
    if (n_words == 0) return SAGE_OK;
    const int blocks = (int)(n_words < 256 ? 1 : (n_words / 256 < 1024 ? n_words / 256 : 1024));
    hipLaunchKernelGGL(sage_fill_u32_kernel, dim3(blocks), dim3(256), 0, st, (uint32_t*)p, v, n_words);
    SAGE_CHECK_LAUNCH("sage_fill_u32_kernel");
    return SAGE_OK;
}
