"""Eager-mode GPU time per model section (HIP events), to decide what to fuse next."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from immunostruct_amd import synthetic
from immunostruct_amd.graph import PackedGraphBatch
from immunostruct_amd.models import model_map
from immunostruct_amd.utils import Losses
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
VAE_IN = 5943
raw = synthetic.make_batch(128, seed=1)
g = PackedGraphBatch.from_raw(raw, device=dev)
seq = torch.from_numpy(raw.one_hot_sequence()).to(dev); prop = torch.from_numpy(raw.prop).to(dev); y = torch.from_numpy(raw.y_reg).to(dev)
model = model_map["HybridModelv2"](vae_input_dim=VAE_IN, device=dev).to(dev); model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
losses = Losses(VAE_IN, {0: 81.0, 1: 19.0})
def step():
    opt.zero_grad(set_to_none=False)
    r = model(g, seq, prop); l = losses.regression_loss(r[0], seq, r[1], r[2], r[3], y); l.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t > 0: rows.append((t / 3.0, e.count // 3, e.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total device us/step {tot:.0f}")
for t, c, k in rows[:45]: print(f"{t:9.1f} us  x{c:4d}  {k[:110]}")
