/*
 * sampler_ref.c -- CPU restatement of the device sampler's INTEGER work.
 *
 * TEST INFRASTRUCTURE (oracle/__init__.py): linked only by tests, smoke() and
 * bench.py's checker legs.
 *
 * What it pins: include/sage355.h sage_sample_neighbors().  The reference draws
 * with Python's global `random.sample` (graphsage/aggregators.py:42-46); a GPU
 * sampler cannot consume that stream, so the contract taken from the reference
 * is distributional -- deg >= k: a uniform k-subset without replacement;
 * deg < k: the whole neighbour set (aggregators.py:45-46) -- and THIS file is
 * the bit-exact statement of how the build realises it:
 *
 *   key     = (seed & 0xffffffff, seed >> 32)
 *   block b = Philox4x32-10(counter = (node, tag, b, 0), key)      [Salmon et al. 2011]
 *   draw i  = word (i & 3) of block (i >> 2)
 *   Floyd's subset algorithm over positions of the CSR row:
 *       for i in 0..k-1:  j = deg-k+i;  t = (draw_i * (j+1)) >> 32
 *                         pos_i = t if t not in {pos_0..pos_{i-1}} else j
 *   nbr[r*k+i] = col[rowptr[v] + pos_i],  cnt[r] = k      (deg >  k)
 *   nbr[r*k+j] = col[rowptr[v] + j],      cnt[r] = deg    (deg <= k), rest -1
 *
 * Parity status of the stream itself: "parity unpinned" by construction (the
 * reference has no counterpart).  tests/test_sampler_kat.py checks the Philox
 * block function against the Random123 known-answer vectors and the Floyd walk
 * against a pure-Python restatement (CPU); tests/test_gpu_ops.py pins the
 * device kernel bit-for-bit against this file and tests its statistics.
 */
#include <stdint.h>

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c[0];
        uint64_t p1 = (uint64_t)M1 * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

/* Known-answer hook: one Philox block, so tests can check the published
 * Random123 test vectors before trusting anything built on it. */
void sage_ref_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

int sage_ref_sample_neighbors(const int64_t* rowptr, const int32_t* col, const int32_t* nodes, int32_t n, int32_t k,
                              uint64_t seed, uint32_t tag, int32_t* nbr, int32_t* cnt) {
    if (k < 1 || k > 64) return -1;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int32_t r = 0; r < n; ++r) {
        const int32_t v = nodes[r];
        const int64_t s = rowptr[v];
        const int64_t deg = rowptr[v + 1] - s;
        int32_t* out = nbr + (int64_t)r * k;
        if (deg <= k) {
            cnt[r] = (int32_t)deg;
            for (int j = 0; j < k; ++j) out[j] = j < deg ? col[s + j] : -1;
            continue;
        }
        uint32_t pos[64];
        uint32_t blk[4] = {0, 0, 0, 0};
        const uint32_t base = (uint32_t)(deg - k);
        for (int i = 0; i < k; ++i) {
            if ((i & 3) == 0) {
                blk[0] = (uint32_t)v; blk[1] = tag; blk[2] = (uint32_t)(i >> 2); blk[3] = 0;
                philox4x32_10(blk, k0, k1);
            }
            const uint32_t j = base + (uint32_t)i;
            const uint32_t t = (uint32_t)(((uint64_t)blk[i & 3] * (uint64_t)(j + 1u)) >> 32);
            int dup = 0;
            for (int m = 0; m < i; ++m) dup |= (pos[m] == t);
            pos[i] = dup ? j : t;
        }
        cnt[r] = k;
        for (int i = 0; i < k; ++i) out[i] = col[s + pos[i]];
    }
    return 0;
}
