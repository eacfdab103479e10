"""Shared test helpers: build oracle / product inputs from the same seeded raw batch."""
import os

import numpy as np
import torch

from immunostruct_amd import synthetic
from immunostruct_amd.graph import PackedGraphBatch
from oracle import graph_ref

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VAE_IN = synthetic.SEQ_LEN * synthetic.SEQ_ALPHABET
RECON_COLS = slice(0, None, 37)
ATTN_ROWS = [0, 57, 189]


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def oracle_graph(raw, dtype=torch.float32):
    g = graph_ref.RefGraph(raw.src, raw.dst, raw.num_nodes, raw.batch_num_nodes)
    g.ndata["x"] = torch.from_numpy(raw.x).to(dtype)
    g.edata["edge_attr"] = torch.from_numpy(raw.edge_attr).to(dtype)
    return g


def product_graph(raw, device):
    return PackedGraphBatch.from_raw(raw, device=device)


def make_eps(seed, batch, latent=32):
    return torch.from_numpy(np.random.RandomState(seed).normal(size=(batch, latent)).astype(np.float32))


def det_sd(shapes, seed):
    return {k: torch.from_numpy(v) for k, v in synthetic.det_state_dict(shapes, seed=seed).items()}


def model_shapes(name, **kw):
    from immunostruct_amd.models import model_map
    m = model_map[name](vae_input_dim=VAE_IN, device="cpu", **kw)
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def egnn_shapes(dins, fe, prefix="GCN_layers."):
    shapes = {}
    for i, din in enumerate(dins):
        p = f"{prefix}{i}."
        shapes.update({p + "edge_mlp.0.weight": (64, 2 * din + 1 + fe), p + "edge_mlp.0.bias": (64,),
                       p + "edge_mlp.2.weight": (64, 64), p + "edge_mlp.2.bias": (64,),
                       p + "node_mlp.0.weight": (64, din + 64), p + "node_mlp.0.bias": (64,),
                       p + "node_mlp.2.weight": (64, 64), p + "node_mlp.2.bias": (64,),
                       p + "coord_mlp.0.weight": (64, 64), p + "coord_mlp.0.bias": (64,),
                       p + "coord_mlp.2.weight": (1, 64)})
    return shapes


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny): error relative to the tensor's scale."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    scale = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / scale


def assert_close(a, b, tol, what=""):
    err = rel_err(a, b)
    assert err <= tol, f"{what}: scaled max error {err:.3e} > {tol:.1e}"
    return err
