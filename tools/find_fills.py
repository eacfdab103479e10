"""Which host lines launch fill kernels in one eager train step (torch.profiler with stacks): tiny launches on the replayed
step's critical chain cost ~5 us each whatever they do.   python tools/find_fills.py"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

args = types.SimpleNamespace(model="HybridModelv2", stage="pretrain", batch=128, deg_extra=2, eager=True, force_pack=False)
dev = torch.device("cuda:0")
w = bench.IedbWorkload(args, dev, 0, 1)
for i in range(3):
    w.step(i)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    w.step(3)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
evs.sort(key=lambda e: e.time_range.start)
for i, ev in enumerate(evs):
    if ev.name in ("aten::fill_", "aten::zero_"):
        ctx = [e.name for e in evs[max(0, i - 6):i + 4]]
        print(ev.name, ev.input_shapes, "thread", ev.thread, "| around:", ctx, "| stack:", list(ev.stack or [])[:6])
