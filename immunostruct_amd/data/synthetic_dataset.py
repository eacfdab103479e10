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
    """items: (graph, sequence, target, property (2,)); ``binary`` picks the target; ``full_sequence``: the sequence is the
    padded HLA + peptide (283, 21), otherwise the padded peptide alone (11, 21) -- the reference's default input
    (``SplitDataset(..., full=config.full_sequence)``, ``data/util_dataloader.py:52-66``)."""

    def __init__(self, num_items, seed=1, binary=False, deg_extra=2, full_sequence=True):
        raw = synthetic.make_batch(num_items, seed=seed, deg_extra=deg_extra)
        self.graphs = _graphs_of(raw)
        self.seq = torch.from_numpy(raw.one_hot_sequence())
        if not full_sequence:
            self.seq = self.seq[:, synthetic.HLA_LEN:].contiguous()
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

    def __init__(self, num_items, seed=1, binary=True, full_sequence=True):
        self.c = SyntheticImmunoDataset(num_items, seed=seed, binary=binary, full_sequence=full_sequence)
        self.w = SyntheticImmunoDataset(num_items, seed=seed + 7919, binary=binary, full_sequence=full_sequence)
        self.class_weights = self.c.class_weights

    def __len__(self):
        return len(self.c)

    def __getitem__(self, i):
        gc, sc, y, pc = self.c[i]
        gw, sw, _, pw = self.w[i]
        return (gc, gw), (sc, sw), y, (pc, pw)
