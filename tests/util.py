"""Shared helpers for the parity tests (fixtures -> tensors, tolerance)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# *_b256: BASELINE configs[0] / configs[1] at their batch size (B = 256 seeds, full feature widths), the reference's own outputs
# (round 4; the table of such a fixture is regenerated from a committed integer seed, see synth_table / full_table)
TWO_LAYER_CASES_SMALL = ["tiny_gcn", "tiny_concat", "tiny_sigmoid", "cora_emb_gcn_5_5", "cora_emb_concat_10_10",
                         "cora_bow_gcn_5_5", "cora_bow_concat_5_5", "pubmed_gcn_10_25", "pubmed_concat_10_25"]
TWO_LAYER_CASES_B256 = ["cora_gcn_10_10_b256", "cora_concat_10_10_b256", "cora_gcn_5_5_b256", "cora_concat_5_5_b256",
                        "pubmed_gcn_10_25_b256", "pubmed_concat_10_25_b256"]
TWO_LAYER_CASES = TWO_LAYER_CASES_SMALL + TWO_LAYER_CASES_B256

# BASELINE.json north_star: "within 1e-5 relative fp32".  SURVEY.md section 7 (Tolerance
# definition): the reference's own fp32 result is only reproducible relative to the row
# maximum, so |a-b| <= RTOL * max|ref row| (+ rtol elementwise) is the gate everywhere.
RTOL = 1e-5


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


TABLE_KINDS = {0: "bow", 1: "tfidf"}


def synth_table(kind, n, d, seed):
    """Feature tables of the batch-size fixtures, a pure function of (kind, n, d, seed) -- shared with tests/golden/make_golden.py, which
    records the sha256 of the bytes it fed to the reference.  bow: Cora-like 0/1 bag of words (18 words of 1433 per paper on average);
    tfidf: SURVEY 8(d)'s Pubmed stand-in, rand * Bernoulli(0.1)."""
    rng = np.random.default_rng(int(seed))
    if kind == "bow":
        return (rng.random((n, d)) < 18.0 / d).astype(np.float32)
    if kind == "tfidf":
        return (rng.random((n, d)) * (rng.random((n, d)) < 0.1)).astype(np.float32)
    raise ValueError(kind)


def full_table(g):
    """Rebuild the [N, D0] table: a small fixture stores the touched rows, a batch-size fixture the generator's (kind, seed) and
    the checksum of the table the reference saw."""
    if "table_kind" in g:
        import hashlib
        t = synth_table(TABLE_KINDS[int(g["table_kind"])], int(g["num_nodes"]), int(g["d0"]), int(g["table_seed"]))
        digest = np.frombuffer(hashlib.sha256(t.tobytes()).digest(), dtype=np.uint8)
        assert np.array_equal(digest, g["table_sha256"]), "regenerated feature table differs from the one the reference was run on"
        return torch.from_numpy(t)
    t = torch.zeros(int(g["num_nodes"]), int(g["d0"]))
    t[torch.from_numpy(g["feat_ids"])] = torch.from_numpy(g["feat_rows"])
    return t


def assert_agg1_close(actual, g, what="agg1_out"):
    """Layer-1 aggregator output against the fixture: all rows (small fixtures), or the recorded sample of rows (batch-size
    fixtures keep 16 of the [|S1|, D0] rows: the whole matrix is 7 MB at Cora's width)."""
    if "agg1_out" in g:
        return assert_close_rowmax(actual, g["agg1_out"], what=what)
    rows = torch.from_numpy(g["agg1_rows"])
    return assert_close_rowmax(torch.as_tensor(np.asarray(actual))[rows], g["agg1_out_rows"], what=what + " (row sample)")


def sets_from_padded(nodes, nbr, cnt):
    return {int(n): set(int(x) for x in nbr[r, :int(cnt[r])]) for r, n in enumerate(nodes)}


def assert_close_rowmax(actual, expected, rtol=RTOL, rows_dim=0, what=""):
    """|a-b| <= rtol * max|expected row|; NaNs must coincide."""
    a = torch.as_tensor(np.asarray(actual), dtype=torch.float64)
    e = torch.as_tensor(np.asarray(expected), dtype=torch.float64)
    assert a.shape == e.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(e.shape)}"
    if rows_dim == 1:
        a, e = a.t(), e.t()
    nan_a, nan_e = torch.isnan(a), torch.isnan(e)
    assert torch.equal(nan_a, nan_e), f"{what}: NaN pattern differs"
    a = torch.nan_to_num(a)
    e = torch.nan_to_num(e)
    if e.numel() == 0:
        return 0.0
    scale = e.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    err = ((a - e).abs() / scale).max().item()
    assert err <= rtol, f"{what}: max |a-b|/rowmax = {err:.3e} > {rtol:.1e}"
    return err


def usable_cores():
    """Host cores this process may actually use: the affinity mask / cgroup quota, not the machine's core count (a GPU box
    reports 256 cores to a 16-core share, and torch's CPU ops on 256 intra-op threads run 20x slower there than on 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n
