"""Batch construction with the reference's function names (``data/utils.py``), producing ``PackedGraphBatch``.

``to_packed`` is the ``to_dgl`` equivalent (``data/utils.py:54-67``): a PyG-style object with ``x``,
``edge_index`` and ``num_nodes`` becomes a graph with ``ndata['x']`` and a constant ``edge_attr`` of ones;
``pad_graph`` zero-pads features/coordinates to the dataset-wide node count (``:13-33``);
``collate`` batches (graph, sequence, label, property) samples, single or cancer/wild-type paired
(``:160-176``); ``one_hot_encode_sequence`` / ``RandomRotation`` as in ``:75-89,148-155``.
"""
from __future__ import annotations

import numpy as np
import torch

from ..graph import PackedGraphBatch, batch, graph

__all__ = ["pad_graph", "to_packed", "to_dgl", "one_hot_encode_sequence", "pad_peptide_sequence", "RandomRotation", "collate"]

AMINO_ACIDS = "ACDEFGHIKLMNPQRSTVWY"
PADDING_CHAR = "J"


def pad_graph(g, max_nodes, feature_size, coord_size):
    if g.x.shape[1] != feature_size:
        raise ValueError("`pad_graph`: graph.x shape mismatch.")
    missing = max_nodes - g.num_nodes
    if missing > 0:
        g.x = torch.cat([g.x, torch.zeros(missing, feature_size)], dim=0)
        g.coords = torch.cat([g.coords, torch.zeros(missing, coord_size)], dim=0)
        g.num_nodes = max_nodes
    return g


def to_packed(pyg_graph) -> PackedGraphBatch:
    src, dst = pyg_graph.edge_index
    out = graph((src, dst), num_nodes=pyg_graph.num_nodes)
    out.ndata["x"] = pyg_graph.x
    out.edata["edge_attr"] = torch.ones((src.numel(), 1))
    out.csr()   # build the CSR indices once per graph (DataLoader workers), not once per step
    return out


to_dgl = to_packed  # the reference's name


def pad_peptide_sequence(sequence, max_length=11, padding_char=PADDING_CHAR):
    return sequence.ljust(max_length, padding_char)


def one_hot_encode_sequence(sequence, amino_acids=AMINO_ACIDS, padding_char=PADDING_CHAR):
    index = {ch: i for i, ch in enumerate(amino_acids + padding_char)}
    out = np.zeros((len(sequence), len(index)))
    for pos, ch in enumerate(sequence):
        if ch in index:
            out[pos, index[ch]] = 1
        else:
            print("unknown character: {}", ch)
    return out


class RandomRotation:
    """Random orthogonal transform of the coordinates (QR of a Gaussian matrix)."""

    def __call__(self, x):
        q, _ = np.linalg.qr(np.random.randn(3, 3))
        return x @ q


def _stack(items):
    return torch.stack(list(items), dim=0)


def collate(samples):
    graphs, seqs, labels, props = map(list, zip(*samples))
    if isinstance(graphs[0], PackedGraphBatch):
        return batch(graphs), _stack(seqs), _stack(labels), _stack(props)
    # comparative samples: every field except the label is a (cancer, wild-type) pair
    pair = lambda items, fn: (fn([it[0] for it in items]), fn([it[1] for it in items]))
    return pair(graphs, batch), pair(seqs, _stack), _stack(labels), pair(props, _stack)
