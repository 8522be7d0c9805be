"""Graph ingestion: edge lists / dict-of-sets -> CSR (int64 rowptr, int32 col).

Host side of SURVEY.md section 8 row f-3.  The reference keeps adjacency as a
``defaultdict(set)`` filled line by line (graphsage/model.py:303-310 for
``cora.cites``, model.py:449-458 for the Pubmed ``.tab`` file, model.py:135-146
for citeseer); every lookup on the hot path is ``adj_lists[int(node)]``
(graphsage/encoders.py:47).  Here the adjacency is converted ONCE to CSR, the
layout the HIP sampler walks: ``rowptr`` int64 [N+1], ``col`` int32 [nnz], each
row sorted ascending and duplicate free (a Python set has no duplicates and no
order, so sorted is as faithful as any order and makes the layout canonical).

Nothing in this file touches the GPU; ``CSRGraph.to(device)`` is the only torch
call and just moves the two arrays.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class CSRGraph:
    rowptr: "np.ndarray"   # int64 [N+1]
    col: "np.ndarray"      # int32 [nnz], sorted within a row
    num_nodes: int

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def degrees(self):
        return np.diff(self.rowptr)

    def neighbors(self, v):
        return self.col[self.rowptr[v]:self.rowptr[v + 1]]

    def to_adj_lists(self, nodes=None):
        """dict node -> set(neighbours): the reference's adjacency type.  Sets are
        built from the sorted row so that two processes get the same iteration
        order (needed to replay the reference's ``random.sample`` stream)."""
        ids = range(self.num_nodes) if nodes is None else nodes
        return {int(v): set(int(x) for x in self.neighbors(int(v))) for v in ids}

    def to(self, device):
        import torch
        return (torch.from_numpy(self.rowptr).to(device),
                torch.from_numpy(self.col).to(device))


def csr_from_edges(src, dst, num_nodes, symmetric=True, drop_self_loops=False):
    """COO -> canonical CSR.  ``symmetric`` adds the reverse of every edge, as the
    loaders do with the pair of ``add`` calls (model.py:309-310).  Self loops are
    kept by default: the reference keeps them (``adj_lists[p].add(p)``)."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    if drop_self_loops:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    key = np.unique(src * np.int64(num_nodes) + dst)
    src = key // num_nodes
    dst = key % num_nodes
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=num_nodes), out=rowptr[1:])
    return CSRGraph(rowptr, dst.astype(np.int32), int(num_nodes))


def csr_from_adj_lists(adj_lists, num_nodes=None):
    """dict-of-sets (the reference's ``adj_lists``) -> CSR, vectorised per row."""
    if num_nodes is None:
        num_nodes = 0
        for k, v in adj_lists.items():
            num_nodes = max(num_nodes, int(k) + 1, (max(v) + 1) if len(v) else 0)
    deg = np.zeros(num_nodes, dtype=np.int64)
    for k, v in adj_lists.items():
        deg[int(k)] = len(v)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    col = np.empty(int(rowptr[-1]), dtype=np.int32)
    for k, v in adj_lists.items():
        if len(v):
            s = int(rowptr[int(k)])
            col[s:s + len(v)] = np.sort(np.fromiter(v, dtype=np.int64, count=len(v)))
    return CSRGraph(rowptr, col, int(num_nodes))


def _relabel_first_appearance(a, b):
    """String/int paper ids -> 0..N-1 in order of first appearance (the content
    files that define the reference's ``node_map`` are not in the mount, so
    first appearance in the edge list is this build's documented convention)."""
    inter = np.empty(a.size + b.size, dtype=np.result_type(a, b))
    inter[0::2] = a
    inter[1::2] = b
    uniq, first = np.unique(inter, return_index=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    ids = rank[np.searchsorted(uniq, inter)]
    return ids[0::2], ids[1::2], uniq[order]


def read_edge_list(path, fmt="pairs"):
    """Edge-list files the reference's loaders read.

    fmt="pairs":  ``<id> <id>`` per line, tab or space separated
                  (cora/cora.cites, citeseer/citeseer.cites, *.cites.parsed).
    fmt="pubmed": Pubmed-Diabetes.DIRECTED.cites.tab -- two header lines, then
                  ``n<TAB>paper:A<TAB>|<TAB>paper:B`` (model.py:450-456).
    Returns (CSRGraph, original_ids) with ids relabelled by first appearance.
    """
    a, b = [], []
    with open(path) as fp:
        if fmt == "pubmed":
            fp.readline()
            fp.readline()
            for line in fp:
                parts = line.strip().split("\t")
                if len(parts) < 4:
                    continue
                a.append(parts[1].split(":")[1])
                b.append(parts[-1].split(":")[1])
        else:
            for line in fp:
                parts = line.split()
                if len(parts) < 2:
                    continue
                a.append(parts[0])
                b.append(parts[1])
    a = np.array(a)
    b = np.array(b)
    ia, ib, names = _relabel_first_appearance(a, b)
    return csr_from_edges(ia, ib, names.size, symmetric=True), names


def _accel_device(accel, big=True):
    if accel is None:
        return None
    import torch
    if accel == "auto":
        return torch.device("cuda", torch.cuda.current_device()) if (big and torch.cuda.is_available()) else None
    return torch.device(accel)


def rmat_graph(scale, num_edges, a=0.57, b=0.19, c=0.19, seed=0, chunk=1 << 24, cache_dir=None, accel="auto"):
    """Graph500-parameter R-MAT generator: ``num_edges`` directed
    draws on 2**scale nodes, ``numpy.random.default_rng(seed)``, one uniform per
    level choosing the quadrant with probabilities (a, b, c, 1-a-b-c); self loops
    dropped, symmetrised, deduplicated.  Generated in chunks to bound memory;
    ``cache_dir`` keeps the CSR as .npy files for later runs on the same box.
    ``accel``: where the integer work on the (numpy-drawn) uniforms runs -- "auto" = the GPU when there is one and the
    graph is big, None = numpy, or a torch device; the result is the same CSR bit for bit."""
    import os
    tag = f"rmat_s{scale}_e{num_edges}_a{a}_b{b}_c{c}_seed{seed}"
    if cache_dir is not None:
        fr, fc = os.path.join(cache_dir, tag + "_rowptr.npy"), os.path.join(cache_dir, tag + "_col.npy")
        if os.path.exists(fr) and os.path.exists(fc):
            try:
                return CSRGraph(np.load(fr), np.load(fc), 1 << scale)
            except (OSError, ValueError, EOFError):
                pass                               # unreadable cache: regenerate
    rng = np.random.default_rng(seed)
    n = 1 << scale
    t_ab, t_abc = np.float32(a + b), np.float32(a + b + c)
    t_a = np.float32(a)
    dev = _accel_device(accel, big=num_edges >= 4_000_000)
    if dev is not None:
        # Same uniforms from the same numpy stream; only the integer work on them (bit assembly, symmetrise, sort-unique,
        # row pointers) runs as torch ops on `dev` -- a sorted set of keys does not depend on who sorted it, so the CSR is
        # identical to the numpy path's (tests/test_host_logic.py), in seconds instead of minutes at 128 M edges.
        import torch
        keys = []
        done = 0
        while done < num_edges:
            m = min(chunk, num_edges - done)
            src = torch.zeros(m, dtype=torch.int64, device=dev)
            dst = torch.zeros(m, dtype=torch.int64, device=dev)
            for _ in range(scale):
                u = torch.from_numpy(rng.random(m, dtype=np.float32)).to(dev, non_blocking=False)
                sbit = u >= float(t_ab)
                dbit = ((u >= float(t_a)) & ~sbit) | (u >= float(t_abc))
                src = (src << 1) | sbit
                dst = (dst << 1) | dbit
            keep = src != dst
            src, dst = src[keep], dst[keep]
            keys.append(torch.unique(torch.cat([src * n + dst, dst * n + src])))
            del src, dst, keep
            done += m
        key = torch.unique(torch.cat(keys)) if len(keys) > 1 else keys[0]
        del keys
        src = key >> scale
        col_t = (key & (n - 1)).to(torch.int32)
        del key
        rowptr_t = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rowptr_t[1:] = torch.cumsum(torch.bincount(src, minlength=n), 0)
        del src
        rowptr, dst = rowptr_t.cpu().numpy(), col_t.cpu().numpy()
        del rowptr_t, col_t
        if dev.type == "cuda":
            torch.cuda.empty_cache()
    else:
        keys = []
        done = 0
        while done < num_edges:
            m = min(chunk, num_edges - done)
            src = np.zeros(m, dtype=np.int64)
            dst = np.zeros(m, dtype=np.int64)
            for _ in range(scale):
                u = rng.random(m, dtype=np.float32)
                sbit = u >= t_ab
                dbit = ((u >= t_a) & ~sbit) | (u >= t_abc)
                src = (src << 1) | sbit
                dst = (dst << 1) | dbit
            keep = src != dst
            src, dst = src[keep], dst[keep]
            keys.append(np.unique(np.concatenate([src * n + dst, dst * n + src])))
            done += m
        key = np.unique(np.concatenate(keys)) if len(keys) > 1 else keys[0]
        src = key >> scale
        dst = key & (n - 1)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(src, minlength=n), out=rowptr[1:])
    g = CSRGraph(rowptr, dst.astype(np.int32), n)
    if cache_dir is not None:
        # several ranks of one node generate the same graph at the same time (bench.py --gpus N): each writes its own
        # temporary file and renames it into place, so a reader never sees a half-written cache (col first: the
        # reader's test is "both exist", and rowptr appearing last makes the pair complete)
        os.makedirs(cache_dir, exist_ok=True)
        for path, arr in ((fc, g.col), (fr, g.rowptr)):
            tmp = f"{path}.{os.getpid()}.tmp.npy"
            np.save(tmp, arr)
            os.replace(tmp, path)
    return g


def relabel_by_degree(g):
    """The same graph with node ids renumbered by descending degree (id 0 = the biggest hub): the rows of the feature
    table that are gathered most often then sit next to each other in memory.  -> (graph, new_id_of_old)."""
    deg = g.degrees()
    order = np.argsort(-deg, kind="stable")                 # old ids in new order
    new_of_old = np.empty(g.num_nodes, dtype=np.int64)
    new_of_old[order] = np.arange(g.num_nodes)
    src_old = np.repeat(np.arange(g.num_nodes, dtype=np.int64), deg)
    src = new_of_old[src_old]
    dst = new_of_old[g.col.astype(np.int64)]
    key = np.sort(src * np.int64(g.num_nodes) + dst)
    rowptr = np.zeros(g.num_nodes + 1, dtype=np.int64)
    np.cumsum(deg[order], out=rowptr[1:])
    return CSRGraph(rowptr, (key % g.num_nodes).astype(np.int32), g.num_nodes), new_of_old


def truncate_nodes(g, num_nodes):
    """Keep the subgraph induced on ids < num_nodes (ogbn-products-shaped config:
    scale-22 R-MAT truncated to 2.4 M ids)."""
    deg = g.degrees()
    src = np.repeat(np.arange(g.num_nodes, dtype=np.int64), deg)
    keep = (src < num_nodes) & (g.col < num_nodes)
    src = src[keep]
    dst = g.col[keep]
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=num_nodes), out=rowptr[1:])
    return CSRGraph(rowptr, dst.astype(np.int32), int(num_nodes))
