import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from immunostruct_amd import synthetic, functional as HF
from immunostruct_amd.graph import PackedGraphBatch
from immunostruct_amd.nn import EGNNConv, egnn_stack_forward
dev = torch.device("cuda:0")
raw = synthetic.make_batch(3, seed=1)
g = PackedGraphBatch.from_raw(raw, device=dev)
torch.manual_seed(3)
layers = [EGNNConv(20, 64, 64, 1).to(dev)]
out = {}
for mode in ("v2", "v3"):
    HF.EDGE_FWD = mode
    with torch.no_grad():
        h, x = egnn_stack_forward(layers, g, g.ndata["x"][:, :20], g.ndata["x"][:, 20:], g.edata["edge_attr"])
    out[mode] = (h.cpu(), x.cpu())
dh = (out["v2"][0] - out["v3"][0]).abs(); dx = (out["v2"][1] - out["v3"][1]).abs()
print("dh max", dh.max().item(), "rows differing", (dh.max(1).values > 0).sum().item(), "of", dh.shape[0])
print("dx max", dx.max().item(), "rows differing", (dx.max(1).values > 0).sum().item())
rows = torch.nonzero(dx.max(1).values > 0).flatten()[:10]
print(rows.tolist())
rp = g.csr().rowptr_dst.cpu()
for r in rows[:5].tolist():
    print(r, "deg", int(rp[r+1]-rp[r]), out["v2"][1][r].tolist(), out["v3"][1][r].tolist())
