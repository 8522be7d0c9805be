"""Shared helpers for the parity tests (fixtures -> tensors, tolerance)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TWO_LAYER_CASES = ["tiny_gcn", "tiny_concat", "tiny_sigmoid", "cora_emb_gcn_5_5", "cora_emb_concat_10_10",
                   "cora_bow_gcn_5_5", "cora_bow_concat_5_5", "pubmed_gcn_10_25", "pubmed_concat_10_25"]

# BASELINE.json north_star: "within 1e-5 relative fp32".  SURVEY.md section 7 (Tolerance
# definition): the reference's own fp32 result is only reproducible relative to the row
# maximum, so |a-b| <= RTOL * max|ref row| (+ rtol elementwise) is the gate everywhere.
RTOL = 1e-5


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def full_table(g):
    """Rebuild the [N, D0] table: only the touched rows are stored in a fixture."""
    t = torch.zeros(int(g["num_nodes"]), int(g["d0"]))
    t[torch.from_numpy(g["feat_ids"])] = torch.from_numpy(g["feat_rows"])
    return t


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
