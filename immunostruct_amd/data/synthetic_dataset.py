"""Synthetic stand-in for ``ImmunoPredDataset`` / ``ImmunoPredDatasetComparative`` (reference
``data/immmunopred_dataloader.py``): the real graph files were never shipped, so the entry scripts can run
end-to-end on graphs drawn by ``immunostruct_amd.synthetic`` with the same per-item tuple layout."""
from __future__ import annotations

from collections import Counter

import torch
from torch.utils.data import Dataset

from .. import synthetic
from ..graph import PackedGraphBatch


def _graphs_of(raw):
    n = int(raw.batch_num_nodes[0])
    out = []
    src, dst = torch.from_numpy(raw.src), torch.from_numpy(raw.dst)
    gid = dst // n
    for i in range(raw.num_graphs):
        sel = gid == i
        g = PackedGraphBatch(src[sel] - i * n, dst[sel] - i * n, n)
        g.ndata["x"] = torch.from_numpy(raw.x[i * n:(i + 1) * n])
        g.edata["edge_attr"] = torch.from_numpy(raw.edge_attr[sel.numpy()])
        g.csr()
        out.append(g)
    return out


class SyntheticImmunoDataset(Dataset):
    """items: (graph, full sequence (283,21), target, property (2,)); ``binary`` picks the target."""

    def __init__(self, num_items, seed=1, binary=False, deg_extra=2):
        raw = synthetic.make_batch(num_items, seed=seed, deg_extra=deg_extra)
        self.graphs = _graphs_of(raw)
        self.seq = torch.from_numpy(raw.one_hot_sequence())
        self.prop = torch.from_numpy(raw.prop)
        self.y = torch.from_numpy(raw.y_bin if binary else raw.y_reg)
        counts = Counter(raw.y_bin.tolist())
        self.class_weights = {0: float(counts.get(0.0, 1)), 1: float(max(counts.get(1.0, 1), 1))}

    def __len__(self):
        return len(self.graphs)

    def __getitem__(self, i):
        return self.graphs[i], self.seq[i], self.y[i], self.prop[i]


class SyntheticPairedDataset(Dataset):
    """cancer / wild-type pairs: every field but the label is a 2-tuple."""

    def __init__(self, num_items, seed=1, binary=True):
        self.c = SyntheticImmunoDataset(num_items, seed=seed, binary=binary)
        self.w = SyntheticImmunoDataset(num_items, seed=seed + 7919, binary=binary)
        self.class_weights = self.c.class_weights

    def __len__(self):
        return len(self.c)

    def __getitem__(self, i):
        gc, sc, y, pc = self.c[i]
        gw, sw, _, pw = self.w[i]
        return (gc, gw), (sc, sw), y, (pc, pw)
