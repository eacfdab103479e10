"""A/B timing of the two EGNN layer kernels on the 6-layer stack (B from $B, default 128; deg_extra from $DEG, default 2):
per-launch time of egnn_layer_fwd / egnn_layer_bwd (HIP events, 4 launches back to back per pair), the wall time of ITER eager
forward + backward passes of the whole stack, and a digest of the gradients (to compare variants selected through environment
switches such as IMMUNOSTRUCT_SAVE_Z3 or IMMUNOSTRUCT_LIB).   python tools/layer_ab.py [label]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from immunostruct_amd import functional as HF  # noqa: E402
from immunostruct_amd import synthetic  # noqa: E402
from immunostruct_amd.graph import PackedGraphBatch  # noqa: E402
from immunostruct_amd.nn import EGNNConv, egnn_stack_forward  # noqa: E402

HF.LaunchClock.enabled = os.environ.get("CLOCKS", "0") == "1"      # the in-kernel workgroup stamps bench.py switches on
dev = torch.device("cuda:0")
B, DEG, FE = int(os.environ.get("B", 128)), int(os.environ.get("DEG", 2)), int(os.environ.get("FE", 1))
raw = synthetic.make_batch(B, seed=1, deg_extra=DEG, edge_feats=FE)
g = PackedGraphBatch.from_raw(raw, device=dev)
torch.manual_seed(0)
layers = [EGNNConv(20 if i == 0 else 64, 64, 64, FE).to(dev) for i in range(6)]
h0 = g.ndata["x"][:, :20].contiguous()
x0 = g.ndata["x"][:, 20:].contiguous()
ea = g.edata["edge_attr"]


def step():
    for l in layers:
        l.zero_grad(set_to_none=True)
    h, x = egnn_stack_forward(layers, g, h0, x0, ea)
    (h.square().mean() + 1e-4 * x.square().mean()).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
digest = [float(sum(p.grad.double().abs().sum() for p in l.parameters() if p.grad is not None)) for l in layers]
HF.KernelTimer.reset()
HF.KernelTimer.enabled = True
HF.KernelTimer.repeat, HF.KernelTimer.repeat_names = 4, frozenset(["egnn_layer_fwd", "egnn_layer_bwd", "egnn_layer_fwd_nocoord", "egnn_layer_bwd_nocoord"])
for _ in range(6):
    step()
torch.cuda.synchronize()
t = {k: round(v[1] * 1e3, 2) for k, v in HF.KernelTimer.summary().items()}
HF.KernelTimer.enabled = False
iters = 20
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / iters
print(json.dumps({"label": sys.argv[1] if len(sys.argv) > 1 else "", "B": B, "E": raw.num_edges, "kernels_us": t,
                  "eager_step_ms": round(wall * 1e3, 3), "grad_digest": [round(d, 6) for d in digest]}))
