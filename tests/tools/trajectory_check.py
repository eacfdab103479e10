"""Loss trajectory of several Adam steps at the reference's learning rate: HIP path vs the CPU oracle on the same batches,
same initial weights, same reparameterisation noise.  python tests/tools/trajectory_check.py [steps] [batch] [lr]"""
import os, sys
import unittest.mock as mock

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from immunostruct_amd import optim, synthetic  # noqa: E402
from immunostruct_amd.graph import PackedGraphBatch  # noqa: E402
from immunostruct_amd.models import model_map  # noqa: E402
from immunostruct_amd.utils import Losses  # noqa: E402
from oracle import functional_ref as FR  # noqa: E402
from oracle import graph_ref  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
dev = torch.device("cuda:0")
VAE_IN = synthetic.SEQ_LEN * synthetic.SEQ_ALPHABET
torch.manual_seed(0)
model = model_map["HybridModelv2"](vae_input_dim=VAE_IN, device=dev).to(dev)
model.eval()     # no dropout masks: both sides deterministic given eps
sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
opt_h = optim.Adam(model.parameters(), lr=lr)
opt_o = torch.optim.Adam(list(sd.values()), lr=lr)
losses = Losses(VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
torch.set_num_threads(min(16, torch.get_num_threads()))
for s in range(steps):
    raw = synthetic.make_batch(B, seed=500 + s, deg_extra=2)
    eps = torch.from_numpy(np.random.RandomState(900 + s).normal(size=(B, 32)).astype(np.float32))
    seq, prop, y = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop), torch.from_numpy(raw.y_reg)
    # HIP
    g = PackedGraphBatch.from_raw(raw, device=dev)
    opt_h.zero_grad(set_to_none=True)
    with mock.patch("torch.randn_like", lambda t: eps.to(t.device, t.dtype)):
        recon, mu, logvar, final = model(g, seq.to(dev), prop.to(dev))
    lh = losses.regression_loss(recon, seq.to(dev), mu, logvar, final, y.to(dev))
    lh.backward()
    opt_h.step()
    # oracle
    go = graph_ref.RefGraph(raw.src, raw.dst, raw.num_nodes, raw.batch_num_nodes)
    go.ndata["x"], go.edata["edge_attr"] = torch.from_numpy(raw.x), torch.from_numpy(raw.edge_attr)
    opt_o.zero_grad()
    with mock.patch("torch.randn_like", lambda t: eps.to(t.dtype)):
        it = FR.forward("HybridModelv2", sd, go, seq, prop)
    lo = FR.regression_loss(it["recon_x"], seq, it["mu"], it["logvar"], it["final_output"], y, VAE_IN)
    lo.backward()
    opt_o.step()
    lh, lo = float(lh), float(lo)
    print(f"step {s:3d}  hip {lh:.6f}  oracle {lo:.6f}  rel {abs(lh - lo) / max(abs(lo), 1e-30):.2e}  "
          f"max|logvar| hip {float(logvar.abs().max()):.2f} oracle {float(it['logvar'].abs().max()):.2f}", flush=True)
