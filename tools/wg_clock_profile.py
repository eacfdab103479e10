"""Per-workgroup start offsets and durations of one forward and one backward layer launch (the in-kernel wall-clock stamps bench.py
uses, 10 ns resolution): how long the dispatch ramp is, how even the workgroups' durations are, which half of the grid (first / second
workgroup of a CU) ends last.   python tools/wg_clock_profile.py   ($B graphs, default 128; $LAYER, default 3)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from immunostruct_amd import functional as HF  # noqa: E402
from immunostruct_amd import synthetic  # noqa: E402
from immunostruct_amd.graph import PackedGraphBatch  # noqa: E402
from immunostruct_amd.nn import EGNNConv, egnn_stack_forward  # noqa: E402

HF.LaunchClock.enabled = True
dev = torch.device("cuda:0")
B, LAYER = int(os.environ.get("B", 128)), int(os.environ.get("LAYER", 3))
raw = synthetic.make_batch(B, seed=1)
g = PackedGraphBatch.from_raw(raw, device=dev)
torch.manual_seed(0)
layers = [EGNNConv(20 if i == 0 else 64, 64, 64, 1).to(dev) for i in range(6)]
h0 = g.ndata["x"][:, :20].contiguous(); x0 = g.ndata["x"][:, 20:].contiguous(); ea = g.edata["edge_attr"]
for rep in range(4):
    for l in layers:
        l.zero_grad(set_to_none=True)
    h, x = egnn_stack_forward(layers, g, h0, x0, ea)
    (h.square().mean() + 1e-4 * x.square().mean()).backward()
    torch.cuda.synchronize()
for key, buf in sorted(HF.LaunchClock.sites.items(), key=str):
    if key[1] != LAYER:
        continue
    t = buf.cpu().double() / HF.LaunchClock.TICKS_PER_US
    s, e = t[:, 0] - t[:, 0].min(), t[:, 1] - t[:, 0].min()
    d = e - s
    q = lambda v, p: float(torch.quantile(v, p))
    print(key, f"span {float(e.max()):.2f} us; start offset p50 {q(s, .5):.2f} p90 {q(s, .9):.2f} max {float(s.max()):.2f};"
          f" duration min {float(d.min()):.2f} p10 {q(d, .1):.2f} p50 {q(d, .5):.2f} p90 {q(d, .9):.2f} max {float(d.max()):.2f};"
          f" end p10 {q(e, .1):.2f} p50 {q(e, .5):.2f} p90 {q(e, .9):.2f}")
    n = t.shape[0]
    for a in range(0, n, max(1, n // 8)):
        b = min(n, a + max(1, n // 8))
        print(f"   wg {a:4d}..{b - 1:4d}: start {float(s[a:b].mean()):6.2f}  duration {float(d[a:b].mean()):6.2f} (max {float(d[a:b].max()):6.2f})  end {float(e[a:b].mean()):6.2f} (max {float(e[a:b].max()):6.2f})")
