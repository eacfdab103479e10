"""Import shims that let the reference's own Python run in the build container.

TEST INFRASTRUCTURE ONLY.  ``dgl``, ``torch_geometric`` and ``wandb`` are not
installed (and cannot be: no network), so the reference's ``models/*.py``
cannot be imported as-is (``models/hybrid_models.py:4-5``).  This module puts
minimal stand-ins into ``sys.modules`` -- backed by ``oracle/graph_ref.py`` --
and puts ``/root/reference/immunostruct`` on ``sys.path`` so that the
reference's *own* model / loss / contrastive code is executed unchanged.

Used by ``oracle/make_golden.py`` (fixture generation) and by
``tests/test_oracle_pins_reference.py`` (skipped where ``/root/reference`` is
absent, e.g. on the GPU box).  Nothing of the reference is copied: it is only
imported from where it lies.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("IMMUNOSTRUCT_REFERENCE", "/root/reference")
REFERENCE_PKG = os.path.join(REFERENCE_ROOT, "immunostruct")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_PKG, "models", "hybrid_models.py"))


def install():
    """Install the shims (idempotent) and make the reference importable."""
    import torch.utils.data

    from . import graph_ref

    if "dgl" not in sys.modules or not getattr(sys.modules["dgl"], "_oracle_shim", False):
        dgl = types.ModuleType("dgl")
        dgl._oracle_shim = True
        dgl.DGLGraph = graph_ref.RefGraph
        dgl.graph = graph_ref.graph
        dgl.batch = graph_ref.batch
        dgl_nn = types.ModuleType("dgl.nn")
        dgl_nn.EGNNConv = graph_ref.EGNNConvRef
        dgl_dl = types.ModuleType("dgl.dataloading")
        dgl_dl.GraphDataLoader = torch.utils.data.DataLoader
        dgl.nn, dgl.dataloading = dgl_nn, dgl_dl
        sys.modules.update({"dgl": dgl, "dgl.nn": dgl_nn, "dgl.dataloading": dgl_dl})

    if "torch_geometric" not in sys.modules:
        pyg = types.ModuleType("torch_geometric")
        pyg_nn = types.ModuleType("torch_geometric.nn")
        pyg_nn.global_mean_pool = graph_ref.global_mean_pool
        pyg_nn.global_max_pool = graph_ref.global_max_pool
        pyg.nn = pyg_nn
        sys.modules.update({"torch_geometric": pyg, "torch_geometric.nn": pyg_nn})

    if "wandb" not in sys.modules:
        wandb = types.ModuleType("wandb")
        wandb.init = lambda *a, **k: None
        wandb.log = lambda *a, **k: None
        sys.modules["wandb"] = wandb

    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_PKG}")
    if REFERENCE_PKG not in sys.path:
        sys.path.insert(0, REFERENCE_PKG)


def load_reference():
    """Return ``(model_map, Losses, PairedContrastiveLoss)`` from the reference itself."""
    install()
    import matplotlib
    matplotlib.use("Agg")
    from models.mapping import model_map  # reference models/mapping.py:7-22
    from utils.contrastive import PairedContrastiveLoss  # reference utils/contrastive.py
    from utils.loss import Losses  # reference utils/loss.py
    return model_map, Losses, PairedContrastiveLoss
