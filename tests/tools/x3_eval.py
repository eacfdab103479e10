"""Accuracy and speed of the split-bf16 forward edge kernel (v3x) against the fp32 kernel (v3) and an fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from immunostruct_amd import synthetic, functional as HF
from immunostruct_amd.graph import PackedGraphBatch
from immunostruct_amd.nn import EGNNConv, egnn_stack_forward
from oracle import graph_ref
dev = torch.device("cuda:0")
raw = synthetic.make_batch(int(os.environ.get("B", 128)), seed=1)
g = PackedGraphBatch.from_raw(raw, device=dev)
torch.manual_seed(0)
layers = [EGNNConv(20 if i == 0 else 64, 64, 64, 1).to(dev) for i in range(6)]
h0 = g.ndata["x"][:, :20].contiguous(); x0 = g.ndata["x"][:, 20:].contiguous(); ea = g.edata["edge_attr"]
out = {}
for mode in ("v3", "v3x"):
    HF.EDGE_FWD = mode
    with torch.no_grad():
        h, x = egnn_stack_forward(layers, g, h0, x0, ea)
    out[mode] = (h.double().cpu(), x.double().cpu())
    HF.KernelTimer.reset(); HF.KernelTimer.enabled = True
    for _ in range(5):
        with torch.no_grad():
            egnn_stack_forward(layers, g, h0, x0, ea)
    torch.cuda.synchronize()
    print(mode, {k: round(v[1] * 1e3, 1) for k, v in HF.KernelTimer.summary().items()})
    HF.KernelTimer.enabled = False
# fp64 oracle (6 layers, CPU) on a 8-graph slice to keep it quick
n8 = 8 * 190
sel = torch.from_numpy(raw.dst) < n8
src, dst = torch.from_numpy(raw.src)[sel], torch.from_numpy(raw.dst)[sel]
sd = {}
for i, l in enumerate(layers):
    for k, v in l.state_dict().items():
        sd[f"L{i}.{k}"] = v.double().cpu()
h = torch.from_numpy(raw.x[:n8, :20]).double(); x = torch.from_numpy(raw.x[:n8, 20:]).double()
a = torch.from_numpy(raw.edge_attr)[sel].double()
for i in range(6):
    h, x = graph_ref.egnn_conv(sd, f"L{i}.", src, dst, n8, h, x, a)
rel = lambda p, q: float((p - q).abs().max() / q.abs().max())
for mode in ("v3", "v3x"):
    print(mode, "vs fp64 after 6 layers: h", f"{rel(out[mode][0][:n8], h):.2e}", "x", f"{rel(out[mode][1][:n8], x):.2e}")
print("v3x vs v3: h", f"{rel(out['v3x'][0], out['v3'][0]):.2e}", "x", f"{rel(out['v3x'][1], out['v3'][1]):.2e}")
