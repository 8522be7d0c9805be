"""Two CAPTURED runs of the 1024-seed training schedule differ in the last bit in ~1 of 3 processes that ran tests/test_gpu_backward.py
first.  Which tensor differs first, and at which step?  python experiments/r03/repro_flake.py [rounds]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd"), os.path.join(REPO, "tests")]
import numpy as np, pytest, torch
from sage355.graph import rmat_graph
from sage355.train import EngineTrainer

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pytest.main([os.path.join(REPO, "tests", "test_gpu_backward.py"), "-q", "-p", "no:cacheprovider"])
DEV = "cuda"
graph = rmat_graph(14, 300_000, seed=4, accel=None)
gen = torch.Generator().manual_seed(1)
d0, hidden1, b = 256, 128, 1024
table = torch.randn(graph.num_nodes, d0, generator=gen).to(DEV)
rowptr, col = graph.to(DEV)
labels_by_node = torch.from_numpy(np.random.default_rng(3).integers(0, 5, graph.num_nodes)).to(DEV)
cand = np.nonzero(graph.degrees() > 0)[0]
ring = torch.from_numpy(np.stack([np.random.default_rng(10 + i).choice(cand, b, replace=False) for i in range(4)]).astype(np.int32)).to(DEV)
keys = [101, 102, 103, 104]


def make():
    torch.manual_seed(5)
    return EngineTrainer(rowptr, col, table, 5, hidden1=hidden1, hidden2=32, num_sample1=7, num_sample2=15, gcn=True, lr=0.3, max_batch=b, relabel=None)


def snap(tr):
    e = tr.engine
    it = e.intermediates()
    order = torch.argsort(it["s1_nodes"])
    L = e.layout
    n1 = it["n_s1"]
    agg1 = e._view(L.agg1, L.max_s1 * e.d0p, torch.float32).view(L.max_s1, e.d0p)[:n1][order].clone()
    return {"agg1(sorted by node)": agg1, "w1": tr.w1.clone(), "w2": tr.w2.clone(), "w_cls": tr.w_cls.clone(), "h1(sorted by node)": it["h1"][order].clone(),
            "nbr1(sorted)": it["nbr1"][order].clone(), "nbr2": it["nbr2"].clone(), "n_s1": torch.tensor(it["n_s1"])}


ref = make()
ref.engine.forward(ring[0], seed=101)
_it = ref.engine.intermediates()
h1_ref = _it["h1"][torch.argsort(_it["s1_nodes"])].clone()
from sage355.engine import TwoHopEngine
_e2 = TwoHopEngine(rowptr, col, table, ref.w1, ref.w2, 7, 15, max_batch=b, slice_major=False)
_e2.forward(ring[0], seed=101)
_it2 = _e2.intermediates()
h1_rowmajor = _it2["h1"][torch.argsort(_it2["s1_nodes"])].clone()
print("row-major reference == slice-major reference:", bool(torch.equal(h1_ref, h1_rowmajor)), int((h1_ref != h1_rowmajor).sum()))
bad = 0
for rnd in range(rounds):
    trs = [make(), make()]
    e0, e1 = trs[0].engine, trs[1].engine
    print("  before capture: w equal", [bool(torch.equal(a, c)) for a, c in zip(trs[0].parameters(), trs[1].parameters())],
          "split", bool(e0.layout.layer1_split), bool(e1.layout.layer1_split), "sliced", e0._table_sliced is not None, e1._table_sliced is not None)
    if "--embed" in sys.argv:
        o0, o1 = trs[0].embed(ring[0], 101).clone(), trs[1].embed(ring[0], 101).clone()
        print("  eager forward: out equal", bool(torch.equal(o0, o1)))
    losses = [t.capture_step(ring, keys, labels_by_node) for t in trs]
    print("  sliced copies equal:", bool(torch.equal(e0._table_sliced, e1._table_sliced)), "planes equal:", bool(torch.equal(e0._w1prep, e1._w1prep)))
    for e in (e0, e1):
        mq, mc = e._model_q, e._model_c
        print("  after capture: sliced copy", e._table_sliced is not None, "model_c.table_sliced", bool(mc.table_sliced) if mc is not None else None,
              "model_q.table_sliced", bool(mq.table_sliced) if mq is not None else None, "slice floats", getattr(e, "_slice_floats", None),
              "w1prep", e._w1prep is not None, "model_q.w1_prepared", bool(mq.w1_prepared) if mq is not None else None)
    first = None
    for step in range(6):
        snaps = []
        for t, l in zip(trs, losses):
            t.replay_step()
            torch.cuda.synchronize()
            s = snap(t)
            s["loss"] = l.clone()
            snaps.append(s)
        print(f"  step {step}: trailers", [t.engine._w1prep[-16:].tolist() for t in trs])
        if step == 0:
            print("  step 0: h1 == eager reference:", [bool(torch.equal(sn["h1(sorted by node)"], h1_ref)) for sn in snaps],
                  "elements off", [int((sn["h1(sorted by node)"] != h1_ref).sum()) for sn in snaps],
                  "== row-major reference:", [bool(torch.equal(sn["h1(sorted by node)"], h1_rowmajor)) for sn in snaps])
        for k in snaps[0]:
            a, c = snaps[0][k], snaps[1][k]
            if a.shape != c.shape or not torch.equal(a, c):
                nd = int((a != c).sum()) if a.shape == c.shape else -1
                mx = float((a.double() - c.double()).abs().max()) if a.shape == c.shape else float("nan")
                if first is None:
                    first = (step, k)
                print(f"round {rnd} step {step}: {k} differs at {nd} elements, max abs {mx:.3e}")
        if first is not None:
            break
    bad += first is not None
    print(f"round {rnd}: {'first difference at step %d in %s' % first if first else 'identical'}")
print(f"{bad} of {rounds} rounds differ")
