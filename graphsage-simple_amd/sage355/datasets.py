"""Stand-in node-classification data on a REAL citation topology.

The reference's feature/label files (cora/cora.content, pubmed-data/*.NODE.paper.tab) are not in
the mount, so F1 comparisons run on "real topology + synthesised content":
labels = graph-Voronoi cells around random centres (so they correlate with the topology, as topics
do in a citation graph), features = sparse bag-of-words drawn mostly from the label's topic words.
Deterministic in (graph, seed); used identically by tests/golden/make_golden.py (which trains the
REFERENCE on it) and by the GPU training test.
"""
import numpy as np


def graph_voronoi_labels(graph, num_classes, rng):
    n = graph.num_nodes
    labels = np.full(n, -1, dtype=np.int64)
    order = rng.permutation(n)
    centres = order[:num_classes]
    labels[centres] = np.arange(num_classes)
    frontier = list(centres)
    while frontier:
        nxt = []
        for v in frontier:
            for u in graph.neighbors(v):
                if labels[u] < 0:
                    labels[u] = labels[v]
                    nxt.append(int(u))
        frontier = nxt
    missing = labels < 0
    labels[missing] = rng.integers(0, num_classes, size=int(missing.sum()))
    return labels


def standin_citation(graph, num_classes=7, feat_dim=1433, words_per_node=18, topic_words=60, purity=0.7, seed=0):
    """-> (feat_data float32 [N, feat_dim] 0/1, labels int64 [N, 1]) in the reference loaders' layout
    (model.py:267-269: feat_data, labels [N,1])."""
    rng = np.random.default_rng(seed)
    n = graph.num_nodes
    labels = graph_voronoi_labels(graph, num_classes, rng)
    topics = np.stack([rng.choice(feat_dim, topic_words, replace=False) for _ in range(num_classes)])
    feats = np.zeros((n, feat_dim), dtype=np.float32)
    from_topic = rng.random((n, words_per_node)) < purity
    topic_pick = topics[labels[:, None], rng.integers(0, topic_words, size=(n, words_per_node))]
    random_pick = rng.integers(0, feat_dim, size=(n, words_per_node))
    words = np.where(from_topic, topic_pick, random_pick)
    feats[np.arange(n)[:, None], words] = 1.0
    return feats, labels.reshape(-1, 1)
