"""Captured vs eager training step: where do the weights differ?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd")]
import numpy as np, torch
from sage355.graph import rmat_graph
from sage355.train import EngineTrainer
DEV = "cuda"
graph = rmat_graph(14, 300_000, seed=4, accel=None)
rowptr, col = graph.to(DEV)
cand = np.nonzero(graph.degrees() > 0)[0]
labels_by_node = torch.from_numpy(np.random.default_rng(3).integers(0, 5, graph.num_nodes)).to(DEV)
d0, hidden1, b = 256, 128, 1024
table = torch.randn(graph.num_nodes, d0, generator=torch.Generator().manual_seed(1)).to(DEV)
ring = torch.from_numpy(np.stack([np.random.default_rng(10 + i).choice(cand, b, replace=False) for i in range(4)]).astype(np.int32)).to(DEV)
keys = [101, 102, 103, 104]
def make():
    torch.manual_seed(5)
    return EngineTrainer(rowptr, col, table, 5, hidden1=hidden1, hidden2=32, num_sample1=7, num_sample2=15, gcn=True, lr=0.3, max_batch=b)
for nsteps in (1, 2, 3, 6, 7):
    tr = make()
    le = [float(tr.step(ring[i % 4], labels_by_node[ring[i % 4].long()], keys[i % 4])) for i in range(nsteps)]
    cap = make()
    loss = cap.capture_step(ring, keys, labels_by_node)
    lc = []
    for i in range(nsteps):
        cap.replay_step(); lc.append(float(loss))
    for name, a, c in zip(("w1", "w2", "w_cls"), cap.parameters(), tr.parameters()):
        d = (a - c).abs()
        print(f"steps {nsteps} {name}: equal {bool(torch.equal(a, c))} max diff {d.max().item():.3e} differing {int((d > 0).sum())} of {d.numel()} rows {int((d.amax(1) > 0).sum())} cols {int((d.amax(0) > 0).sum())}", flush=True)
    print("   losses equal:", le == lc, le[-1], lc[-1], flush=True)
print("--- two captured runs")
caps = []
for rep in range(2):
    cap = make()
    loss = cap.capture_step(ring, keys, labels_by_node)
    for i in range(6):
        cap.replay_step()
    torch.cuda.synchronize()
    caps.append([p.clone() for p in cap.parameters()])
print("captured == captured:", [bool(torch.equal(a, c)) for a, c in zip(*caps)])
print("--- one step, from equal weights: which quantity differs between the eager computation and the captured graph's?")
tr, cap = make(), make()
loss = cap.capture_step(ring, keys, labels_by_node)
tr.step(ring[0], labels_by_node[ring[0].long()], keys[0]); cap.replay_step(); torch.cuda.synchronize()
print("after step 1 equal:", [bool(torch.equal(a, c)) for a, c in zip(tr.parameters(), cap.parameters())])
# eager API on both, step 2's batch: same weights -> same grads?
l_a, g_a = tr.grads(ring[1], labels_by_node[ring[1].long()], keys[1])
cap.engine.invalidate_weights()
l_b, g_b = cap.grads(ring[1], labels_by_node[ring[1].long()], keys[1])
print("eager grads on both trainers equal:", [bool(torch.equal(a, c)) for a, c in zip(g_a, g_b)], float(l_a) == float(l_b))
# now the graph's own step 2 on `cap`, eager step 2 on `tr`
w_before = [p.clone() for p in cap.parameters()]
cap.replay_step(); tr.step(ring[1], labels_by_node[ring[1].long()], keys[1]); torch.cuda.synchronize()
for name, a, c, w0, ga in zip(("w1", "w2", "w_cls"), cap.parameters(), tr.parameters(), w_before, g_a):
    g_graph = (w0 - a) / cap.lr
    print(f"{name}: graph update vs eager grad: max |diff| {(g_graph - ga).abs().max().item():.3e} (grad scale {ga.abs().max().item():.3e}); weights equal {bool(torch.equal(a, c))}")
