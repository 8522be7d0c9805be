"""Known-answer and cross-restatement tests of the sampler oracle (CPU only).

oracle/sampler_ref.c is the bit-exact statement of the device sampler (include/sage355.h
sage_sample_neighbors; the reference's rule is graphsage/aggregators.py:42-46).  Before the GPU tests
trust it: (1) its Philox4x32-10 block function reproduces the published Random123 known-answer
vectors (Salmon et al. 2011, kat_vectors of the Random123 distribution); (2) its Floyd walk agrees
with an independent pure-Python restatement written from the header comment; (3) the contract taken
from the reference holds: k distinct members of the neighbour set, the whole set when deg <= k."""
import numpy as np

from oracle import sampler_ref

M32 = 0xFFFFFFFF
KAT = [   # (counter, key, expected) -- Random123 kat_vectors, philox4x32 10 rounds
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((M32, M32, M32, M32), (M32, M32), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def py_philox(c, k):
    c0, c1, c2, c3 = c
    k0, k1 = k
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return [c0, c1, c2, c3]


def py_sample(rowptr, col, v, k, seed, tag):
    s, deg = int(rowptr[v]), int(rowptr[v + 1] - rowptr[v])
    if deg <= k:
        return [int(x) for x in col[s:s + deg]]
    key = (seed & M32, seed >> 32)
    pos = []
    for i in range(k):
        if i % 4 == 0:
            blk = py_philox((v & M32, tag, i // 4, 0), key)
        j = deg - k + i
        t = (blk[i % 4] * (j + 1)) >> 32
        pos.append(j if t in pos else t)
    return [int(col[s + p]) for p in pos]


def test_philox_known_answer_vectors():
    for ctr, key, want in KAT:
        assert sampler_ref.philox(ctr, key) == list(want)
        assert py_philox(ctr, key) == list(want)


def _random_csr(rng, n, max_deg):
    deg = rng.integers(0, max_deg, n)
    deg[rng.integers(0, n, 3)] = 4 * max_deg          # a few hubs
    rowptr = np.zeros(n + 1, np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = np.concatenate([np.sort(rng.choice(10 * n, d, replace=False)) for d in deg] + [np.zeros(0, np.int64)]).astype(np.int32)
    return rowptr, col


def test_c_oracle_matches_the_python_restatement_and_the_reference_rule():
    rng = np.random.default_rng(5)
    rowptr, col = _random_csr(rng, 200, 40)
    nodes = rng.permutation(200).astype(np.int32)
    for k in (1, 5, 15, 25, 64):
        for seed in (0, 42, 0xDEADBEEFCAFEF00D):
            nbr, cnt = sampler_ref.sample_neighbors(rowptr, col, nodes, k, seed, 2)
            for r, v in enumerate(nodes):
                want = py_sample(rowptr, col, int(v), k, seed, 2)
                got = [int(x) for x in nbr[r, :cnt[r]]]
                assert got == want
                assert all(x == -1 for x in nbr[r, cnt[r]:])
                row = set(int(x) for x in col[rowptr[v]:rowptr[v + 1]])
                assert len(set(got)) == len(got) == min(len(row), k) and set(got) <= row     # aggregators.py:42-46


def test_draw_depends_only_on_seed_tag_and_node():
    rng = np.random.default_rng(6)
    rowptr, col = _random_csr(rng, 64, 60)
    a, _ = sampler_ref.sample_neighbors(rowptr, col, np.arange(64, dtype=np.int32), 10, 7, 1)
    perm = rng.permutation(64).astype(np.int32)
    b, _ = sampler_ref.sample_neighbors(rowptr, col, perm, 10, 7, 1)
    assert np.array_equal(a[perm], b)
    c, _ = sampler_ref.sample_neighbors(rowptr, col, np.arange(64, dtype=np.int32), 10, 7, 3)
    assert not np.array_equal(a, c)
