"""Time the EGNN forward stack (6 layers) with / without saving pre-activations, v2 vs v3 edge kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from immunostruct_amd import synthetic, functional as HF
from immunostruct_amd.graph import PackedGraphBatch
from immunostruct_amd.nn import EGNNConv, egnn_stack_forward
dev = torch.device("cuda:0")
raw = synthetic.make_batch(int(os.environ.get("B", 128)), seed=1)
g = PackedGraphBatch.from_raw(raw, device=dev)
layers = [EGNNConv(20 if i == 0 else 64, 64, 64, 1).to(dev) for i in range(6)]
h0 = g.ndata["x"][:, :20].contiguous(); x0 = g.ndata["x"][:, 20:].contiguous(); ea = g.edata["edge_attr"]
def run(grad):
    HF.KernelTimer.reset(); HF.KernelTimer.enabled = True
    for _ in range(5):
        if grad:
            egnn_stack_forward(layers, g, h0, x0, ea)
        else:
            with torch.no_grad():
                egnn_stack_forward(layers, g, h0, x0, ea)
    torch.cuda.synchronize()
    t = HF.KernelTimer.summary()
    HF.KernelTimer.enabled = False
    return {k: round(v[1] * 1e3, 1) for k, v in t.items()}
for mode in ("v2", "v3"):
    HF.EDGE_FWD = mode
    for grad in (False, True):
        run(grad)
        print(mode, "save" if grad else "nosave", run(grad))
