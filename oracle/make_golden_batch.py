"""Golden vectors of BATCH CONSTRUCTION from the reference's own ``data/utils.py`` -> ``tests/golden/batch.npz``.

Run in the build container only (needs ``/root/reference``):

    python -m oracle.make_golden_batch

Ragged PyG-style items (``x`` (n_i, 20), ``coords`` (n_i, 3), ``edge_index`` (2, E_i), ``num_nodes``) are put through the
reference's chain as ``data/preprocess.py:334-349`` runs it -- ``x = cat([x, coords])`` -> ``pad_graph`` (``data/utils.py:13-33``)
-> ``to_dgl`` (``:54-67``) -> ``collate`` / ``collate_amino_acid`` (``:160-196``, i.e. ``dgl.batch``) -- imported unchanged under
``oracle/shims.py`` (``dgl.graph`` / ``dgl.batch`` are ``oracle/graph_ref``'s restatement: DGL is not installable here).  Stored:
the batched graph's ``edges()``, ``ndata['x']``, ``edata['edge_attr']``, ``batch_num_nodes()`` and the stacked sequence / label /
property tensors, for a single batch, a (cancer, wild-type) paired batch and the amino-acid collate.  The ITEMS are not stored:
``ragged_items`` regenerates them from the seed on either side.  Test infrastructure; only arrays are written."""
from __future__ import annotations

import importlib
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FEATURES, COORDS, SEQ_LEN = 20, 3, 17
SINGLE_SEED, PAIR_SEED, COUNT = 41, 43, 10


def ragged_items(seed, count=COUNT):
    """``count`` PyG-style structures with 5 .. 23 nodes: a directed chain, random extra edges (parallel edges and self loops
    included), isolated trailing nodes; item 3 has NO edge, item 5 already has the largest node count (nothing to pad).
    Returns ``[(structure, sequence one-hot, label, property)]`` -- every call builds fresh tensors (``pad_graph`` edits in place)."""
    rng = np.random.RandomState(seed)
    out = []
    for i in range(count):
        n = 23 if i == 5 else int(rng.randint(5, 23))
        real = max(2, n - int(rng.randint(0, 3)))                       # trailing nodes without edges
        chain = np.stack([np.arange(real - 1), np.arange(1, real)])
        extra = rng.randint(0, real, size=(2, int(rng.randint(0, 3 * real))))
        if extra.shape[1] >= 2:
            extra[:, 1] = extra[:, 0]                                   # a parallel edge
            extra[1, 0] = extra[0, 0]                                   # a self loop
        edges = np.concatenate([chain, extra], axis=1)
        edges = edges[:, rng.permutation(edges.shape[1])]               # no particular order
        if i == 3:
            edges = edges[:, :0]
        x = rng.rand(n, FEATURES).astype(np.float32)
        coords = rng.randn(n, COORDS).astype(np.float32)
        s = SimpleNamespace(x=torch.from_numpy(x), coords=torch.from_numpy(coords), num_nodes=n,
                            edge_index=torch.from_numpy(edges.astype(np.int64)), name=f"Immuno{seed}_{i}")
        seq = torch.from_numpy(np.eye(21, dtype=np.float32)[rng.randint(0, 21, size=SEQ_LEN)])
        out.append((s, seq, torch.tensor(float(rng.rand())), torch.from_numpy(rng.rand(2).astype(np.float32))))
    return out


def build_samples(utils, seed, pad_to=None):
    """the per-item part of the chain with the functions of the module ``utils`` (the reference's or the product's):
    ``data/preprocess.py:334-349``"""
    items = ragged_items(seed)
    for s, _, _, _ in items:
        s.x = torch.cat([s.x, s.coords], dim=-1).to(dtype=torch.float32)
    max_nodes = max(s.num_nodes for s, _, _, _ in items) if pad_to is None else pad_to
    return [(utils.to_dgl(utils.pad_graph(s, max_nodes, FEATURES + COORDS, COORDS)), seq, y, prop) for s, seq, y, prop in items]


def amino_of(i):
    return torch.tensor([(7 * i + 3) % 20])


def describe(prefix, g):
    src, dst = g.edges()
    return {f"{prefix}/src": src.numpy(), f"{prefix}/dst": dst.numpy(), f"{prefix}/x": g.ndata["x"].numpy(),
            f"{prefix}/edge_attr": g.edata["edge_attr"].numpy(), f"{prefix}/batch_num_nodes": g.batch_num_nodes().numpy()}


def main():
    from oracle import shims
    shims.install()
    utils = importlib.import_module("data.utils")                       # the reference's data/utils.py
    out = {"single_seed": np.int64(SINGLE_SEED), "pair_seed": np.int64(PAIR_SEED), "count": np.int64(COUNT)}
    single = build_samples(utils, SINGLE_SEED)
    g, seq, y, prop = utils.collate(single)
    out.update(describe("single", g))
    out.update({"single/seq": seq.numpy(), "single/y": y.numpy(), "single/prop": prop.numpy()})
    # (cancer, wild-type) pairs share one padded node count (data/immmunopred_dataloader.py:146-147 pads each side on its own;
    # both sides hold the same structures' lengths, so the counts agree -- here both are padded to the larger)
    cancer, wt = build_samples(utils, SINGLE_SEED, 23), build_samples(utils, PAIR_SEED, 23)
    pairs = [((c[0], w[0]), (c[1], w[1]), c[2], (c[3], w[3])) for c, w in zip(cancer, wt)]
    (gc, gw), (sc, sw), y, (pc, pw) = utils.collate(pairs)
    out.update(describe("pair/cancer", gc))
    out.update(describe("pair/wt", gw))
    out.update({"pair/seq_c": sc.numpy(), "pair/seq_w": sw.numpy(), "pair/y": y.numpy(), "pair/prop_c": pc.numpy(), "pair/prop_w": pw.numpy()})
    with_amino = [s + (amino_of(i),) for i, s in enumerate(build_samples(utils, SINGLE_SEED))]
    g, seq, y, prop, amino = utils.collate_amino_acid(with_amino)
    out.update(describe("amino", g))
    out["amino/amino"] = amino.numpy()
    path = os.path.join(ROOT, "tests", "golden", "batch.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
