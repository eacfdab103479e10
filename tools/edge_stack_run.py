"""Small fixed workload for PMC passes: 3 x (forward + backward) of the 6-layer EGNN stack, B = 128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from immunostruct_amd import synthetic
from immunostruct_amd.graph import PackedGraphBatch
from immunostruct_amd.nn import EGNNConv, egnn_stack_forward
dev = torch.device("cuda:0")
raw = synthetic.make_batch(int(os.environ.get("B", 128)), seed=1)
g = PackedGraphBatch.from_raw(raw, device=dev)
layers = [EGNNConv(20 if i == 0 else 64, 64, 64, 1).to(dev) for i in range(6)]
h0 = g.ndata["x"][:, :20].contiguous(); x0 = g.ndata["x"][:, 20:].contiguous(); ea = g.edata["edge_attr"]
for _ in range(3):
    for l in layers:
        l.zero_grad()
    h, x = egnn_stack_forward(layers, g, h0, x0, ea)
    (h.sum() + x.sum()).backward()
torch.cuda.synchronize()
print("ok")
