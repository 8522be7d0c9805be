// Shared host/device helpers for libsage355 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "sage355.h"

// ---- error plumbing (host) --------------------------------------------------
void sage_set_error(const char* fmt, ...);

#define SAGE_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            sage_set_error(__VA_ARGS__);   \
            return SAGE_EINVAL;            \
        }                                  \
    } while (0)

#define SAGE_CHECK_LAUNCH(what)                                                   \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            sage_set_error("%s: %s", what, hipGetErrorString(e_));                \
            return SAGE_ELAUNCH;                                                  \
        }                                                                         \
    } while (0)

static inline bool sage_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline int sage_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kNumCU = 256;        // MI355X

// ---- Philox4x32-10 (device + host; oracle/sampler_ref.c restates it) -----------
struct Philox4 { uint32_t v[4]; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// uniform integer in [0, bound) from one 32-bit draw (multiply-shift; bias < bound / 2^32)
__host__ __device__ inline uint32_t sage_bounded(uint32_t r, uint32_t bound) {
    return (uint32_t)(((uint64_t)r * (uint64_t)bound) >> 32);
}

// ---- frontier hash ------------------------------------------------------------
__host__ __device__ inline uint32_t sage_hash_slot(uint32_t id, uint32_t mask) {
    uint32_t h = id * 0x9E3779B1u;
    h ^= h >> 15;
    return h & mask;
}

#ifdef __HIPCC__
// Insert `id`; returns the slot holding it and whether THIS call claimed the slot.
// Terminates because capacity >= 2 x max distinct ids (checked on the host).
__device__ inline int sage_hash_insert(int32_t* __restrict__ keys, uint32_t mask, int32_t id, bool& won) {
    uint32_t slot = sage_hash_slot((uint32_t)id, mask);
    won = false;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        // CAS straight away.  (A look before the CAS -- one more dependent round trip for every new id -- paid while a
        // hub id was inserted by ~1000 lanes of a batch; since the samplers dedupe inside the block first, at most one
        // lane per block gets here with a given id, and the extra trip cost 1.5 us alone, 3 us with a second batch in
        // flight, where every trip of this latency-bound kernel is ~3x longer.)
#ifdef SAGE_CAS_LOAD_FIRST
        int32_t seen = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == -1) seen = atomicCAS(&keys[slot], -1, id);
#else
        int32_t seen = atomicCAS(&keys[slot], -1, id);
#endif
        if (seen == -1) { won = true; return (int)slot; }
        if (seen == id) return (int)slot;
        slot = (slot + 1) & mask;
    }
    return -1;  // table full: unreachable with a correctly sized table
}

__device__ inline int sage_lane() { return (int)(threadIdx.x & (kWave - 1)); }

// ---- cache policy of the intermediates that cross a kernel boundary (the layer-1 means, h1) -------------------------
// 16-byte store / load of a row piece another KERNEL consumes (never another workgroup of the same launch).
// Policy (MI355X_MICROARCH.md, "stores of each flavour"): plain and nt stores KEEP the written line in the XCD's L2 --
// 3 MB of means per XCD and launch that evict the gather's hub slices -- while sc1 stores write through and DROP it.
//   0 plain   1 nt (round 1-2 default)   2 sc1   3 sc0 sc1   4 nt sc1
// A/B knobs of experiments/r03 (-DSAGE_AGG_STORE=..., -DSAGE_H1_STORE=..., -DSAGE_AGG_LOAD=...); the defaults are what measured best.
using sage_f32x4 = __attribute__((ext_vector_type(4))) float;
template <int POLICY>
__device__ __forceinline__ void sage_store_stream(sage_f32x4* p, const sage_f32x4 v) {
    if constexpr (POLICY == 0) *p = v;
    else if constexpr (POLICY == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}
#ifndef SAGE_AGG_STORE
#define SAGE_AGG_STORE 1
#endif
#ifndef SAGE_H1_STORE
#define SAGE_H1_STORE 1
#endif
#ifndef SAGE_AGG_LOAD          // 0 plain, 1 nt
#define SAGE_AGG_LOAD 1
#endif

__device__ inline uint32_t sage_philox_word(const Philox4& p, int w) {
    return w == 0 ? p.v[0] : w == 1 ? p.v[1] : w == 2 ? p.v[2] : p.v[3];
}

// Fixed-fanout draw for one node by a group of G lanes (lane gl = sample slot gl, k <= G): returns the lane's
// position in the node's CSR row.  deg <= k: the whole row in CSR order.  deg > k (`floyd`): Floyd's subset
// algorithm -- step i draws t_i in [0, deg-k+i] (Philox4x32-10, counter (v, tag, i/4), word i%4) and takes it
// unless one of the i earlier picks already is t_i, then takes deg-k+i: one group-wide compare + ballot per step.
// Must be called by every lane of the wave (shuffles / ballots); bit-identical to oracle/sampler_ref.c.
template <int G>
__device__ inline uint32_t sage_group_positions(bool floyd, int64_t deg, int k, int32_t v, uint32_t tag, uint32_t key0, uint32_t key1,
                                                int gl, int lane) {
    uint32_t pos = (uint32_t)gl;
    if (__any(floyd)) {
        const uint32_t ji = (uint32_t)(deg - (int64_t)k) + (uint32_t)gl;
        uint32_t ti = 0;
        if (floyd && gl < k) {
            const Philox4 p = philox4x32_10((uint32_t)v, tag, (uint32_t)(gl >> 2), 0u, key0, key1);
            ti = sage_bounded(sage_philox_word(p, gl & 3), ji + 1u);
        }
        uint32_t chosen = ti;
        for (int i = 1; i < k; ++i) {
            const uint32_t t = (uint32_t)__shfl((int)ti, i, G);
            const unsigned long long b = __ballot(floyd && gl < i && chosen == t);
            const unsigned long long gb = (G == kWave) ? b : ((b >> (lane - gl)) & ((1ull << (G & 63)) - 1ull));
            if (gl == i) chosen = gb ? ji : ti;
        }
        if (floyd) pos = chosen;
    }
    return pos;
}

__device__ inline float sage_activate(float v, int act) {
    if (act == SAGE_ACT_RELU) return v < 0.f ? 0.f : v;             // NaN stays NaN (torch.relu)
    if (act == SAGE_ACT_SIGMOID) return 1.f / (1.f + expf(-v));     // accurate expf (not __expf): torch.sigmoid to ~1e-7 over the fp32 range
    return v;
}
#endif
