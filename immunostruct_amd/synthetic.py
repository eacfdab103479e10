"""Synthetic peptide-MHC residue-graph batches (no network, no shipped graphs).

The reference's graph inputs (``graph_pyg_IEDB`` etc.) were never shipped
(``/root/reference/.MISSING_LARGE_BLOBS``), so benches and tests draw
synthetic batches with the *shape* of the real ones (``SURVEY.md`` section 8d):

* every graph is padded to ``n`` = 190 nodes (``data/preprocess.py:343-349``);
  real node counts 188/189/190 follow the shipped peptide-length histogram
  (9/10/11-mers on 179 HLA residues); padded nodes have zero features, zero
  coordinates and no edges (``data/utils.py:13-33``);
* node feature = 20-d amino-acid one-hot || 3-d C-alpha coordinate
  (``data/preprocess.py:40-41``); coordinates are a 3.8 Angstrom random walk;
* edges = directed chain i->i+1 plus ``deg_extra`` random directed contacts per
  real node (never touching padded nodes, never self loops); ``symmetric=True``
  lists every chain link and contact in BOTH directions, as ``from_networkx`` does
  for the undirected residue graphs Graphein builds (twice the edges);
* edge feature = ones(E, 1) (``data/utils.py:60``) or U(0,1) features when
  ``edge_feats`` > 1 (BASELINE config 5);
* sequence = 283 tokens over 21 symbols, trailing peptide pad symbol index 20
  (``data/utils.py:69-89``), expanded to one-hot float32 (B, 283, 21);
* properties ~ U(0.2, 0.8)^2; regression target ~ U(-1, 1)
  (``data/immmunopred_dataloader.py:67-70``); binary target ~ Bernoulli(0.19).

Pure numpy; nothing here touches a device.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass

import numpy as np

SEQ_LEN = 283
HLA_LEN = 272          # all 27 shipped alleles; the peptide (9..11 residues, padded to 11) follows
SEQ_ALPHABET = 21
NODE_FEATS = 23
AA = 20


@dataclass
class RawBatch:
    """Host-side arrays of one batched graph + its per-graph side inputs."""
    x: np.ndarray               # (B*n, 23) float32
    src: np.ndarray             # (E,) int64, global node ids
    dst: np.ndarray             # (E,) int64
    edge_attr: np.ndarray       # (E, Fe) float32
    batch_num_nodes: np.ndarray  # (B,) int64, all equal n
    seq_tokens: np.ndarray      # (B, 283) uint8
    prop: np.ndarray            # (B, 2) float32
    y_reg: np.ndarray           # (B,) float32
    y_bin: np.ndarray           # (B,) float32

    @property
    def num_graphs(self):
        return int(self.batch_num_nodes.shape[0])

    @property
    def num_nodes(self):
        return int(self.x.shape[0])

    @property
    def num_edges(self):
        return int(self.src.shape[0])

    def one_hot_sequence(self):
        out = np.zeros((self.num_graphs, SEQ_LEN, SEQ_ALPHABET), dtype=np.float32)
        b, p = np.meshgrid(np.arange(self.num_graphs), np.arange(SEQ_LEN), indexing="ij")
        out[b, p, self.seq_tokens] = 1.0
        return out


def _one_graph(rng, n_pad, n_real, deg_extra, edge_feats, symmetric=False):
    x = np.zeros((n_pad, NODE_FEATS), dtype=np.float32)
    aa = rng.randint(0, AA, size=n_real)
    x[np.arange(n_real), aa] = 1.0
    steps = rng.normal(size=(n_real, 3))
    steps /= np.linalg.norm(steps, axis=1, keepdims=True)
    x[:n_real, AA:] = np.cumsum(3.8 * steps, axis=0).astype(np.float32)
    chain_src = np.arange(n_real - 1)
    chain_dst = chain_src + 1
    extra_dst = np.repeat(np.arange(n_real), deg_extra)
    # uniform over the other real nodes: never a self loop
    extra_src = (extra_dst + 1 + rng.randint(0, n_real - 1, size=extra_dst.shape[0])) % n_real
    src = np.concatenate([chain_src, extra_src]).astype(np.int64)
    dst = np.concatenate([chain_dst, extra_dst]).astype(np.int64)
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    order = rng.permutation(src.shape[0])  # edge order is arbitrary in the real data
    src, dst = src[order], dst[order]
    if edge_feats == 1:
        ea = np.ones((src.shape[0], 1), dtype=np.float32)
    else:
        ea = rng.uniform(0.0, 1.0, size=(src.shape[0], edge_feats)).astype(np.float32)
    return x, src, dst, ea


def make_batch(num_graphs, seed=1, n_pad=190, deg_extra=2, edge_feats=1, n_real_choices=(188, 189, 190),
               n_real_probs=(0.49, 0.29, 0.22), symmetric=False) -> RawBatch:
    """Draw one batch.  ``seed`` fully determines it (numpy legacy RandomState)."""
    rng = np.random.RandomState(seed)
    xs, srcs, dsts, eas = [], [], [], []
    tokens = np.full((num_graphs, SEQ_LEN), SEQ_ALPHABET - 1, dtype=np.uint8)
    for g in range(num_graphs):
        n_real = int(rng.choice(n_real_choices, p=n_real_probs))
        n_real = min(n_real, n_pad)
        x, s, d, ea = _one_graph(rng, n_pad, n_real, deg_extra, edge_feats, symmetric)
        xs.append(x)
        srcs.append(s + g * n_pad)
        dsts.append(d + g * n_pad)
        eas.append(ea)
        pep_len = 9 + (n_real - min(n_real_choices)) if n_real >= min(n_real_choices) else 9
        pep_len = int(np.clip(pep_len, 9, 11))
        tokens[g, : HLA_LEN + pep_len] = rng.randint(0, AA, size=HLA_LEN + pep_len)
    return RawBatch(
        x=np.concatenate(xs, axis=0),
        src=np.concatenate(srcs), dst=np.concatenate(dsts),
        edge_attr=np.concatenate(eas, axis=0),
        batch_num_nodes=np.full((num_graphs,), n_pad, dtype=np.int64),
        seq_tokens=tokens,
        prop=rng.uniform(0.2, 0.8, size=(num_graphs, 2)).astype(np.float32),
        y_reg=rng.uniform(-1.0, 1.0, size=(num_graphs,)).astype(np.float32),
        y_bin=(rng.uniform(size=(num_graphs,)) < 0.19).astype(np.float32),
    )


def det_state_dict(shapes, seed=0):
    """Deterministic, machine-independent parameter fill used by tests and goldens.

    ``shapes``: mapping name -> shape.  Matrices (and 1-d vectors) get
    U(-b, b) with b = 1/sqrt(fan_in) (PyTorch's default Linear bound);
    BatchNorm weights/biases keep 1/0.  numpy's legacy RandomState stream is
    stable across versions, so the GPU box regenerates the same weights.
    """
    out = {}
    for name, shape in shapes.items():
        # seeded by the parameter NAME, so the fill does not depend on dict order
        rng = np.random.RandomState((seed * 100003 + zlib.crc32(name.encode())) % (2 ** 32))
        shape = tuple(shape)
        if name.startswith("projector.1."):  # BatchNorm1d affine / running stats
            if name.endswith("weight") or name.endswith("running_var"):
                out[name] = np.ones(shape, dtype=np.float32)
            elif name.endswith("num_batches_tracked"):
                out[name] = np.zeros(shape, dtype=np.int64)
            else:
                out[name] = np.zeros(shape, dtype=np.float32)
            continue
        fan_in = shape[1] if len(shape) == 2 else max(shape[0], 1)
        bound = 1.0 / np.sqrt(fan_in)
        out[name] = rng.uniform(-bound, bound, size=shape).astype(np.float32)
    return out
