"""Where does a replayed step's time go when nothing in the kernels changed?  (the driver measured
1.83 ms/step where the builder measured 1.39 at identical per-kernel times.)  Replays the bench step in blocks of 20
under different host-side conditions and prints ms/step per block:

    python tools/bench_jitter_probe.py            (needs a GPU)
"""
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from immunostruct_amd import distributed as D, optim  # noqa: E402
from immunostruct_amd.engine import CapturedTrainStep  # noqa: E402
from immunostruct_amd.models import model_map  # noqa: E402
from immunostruct_amd.utils import Losses  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = model_map["HybridModelv2"](vae_input_dim=bench.VAE_IN, device=dev).to(dev).train()
    reducer = D.FlatGradReducer(model.parameters(), world=1)
    opt = optim.Adam(model.parameters(), lr=1e-3)
    losses = Losses(bench.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    pool = bench.build_batches(4, 128, 2, dev, seed0=1000)

    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final = m(g, seq, prop)
        return losses.regression_loss(recon, seq, mu, logvar, final, y)

    b0 = pool[0]
    cap = CapturedTrainStep(model, opt, reducer, forward_loss, (b0["g"], b0["seq"], b0["prop"], b0["y"]),
                            edge_capacity=max(b["raw"].num_edges for b in pool))

    def block(k=20, load=True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            b = pool[i % 4]
            if load:
                cap(b["g"], b["seq"], b["prop"], b["y"])
            else:
                cap.replay()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / k

    for _ in range(5):
        block()
    print("A back-to-back blocks      :", " ".join(f"{block():.3f}" for _ in range(8)), flush=True)
    print("A' replay only (no load)   :", " ".join(f"{block(load=False):.3f}" for _ in range(4)), flush=True)
    for idle in (0.001, 0.005, 0.02, 0.1, 0.5, 2.0):
        out = []
        for _ in range(3):
            torch.cuda.synchronize()
            time.sleep(idle)
            out.append(block())
        print(f"B idle {idle * 1e3:6.0f} ms before block:", " ".join(f"{v:.3f}" for v in out), flush=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    _ = [a.elapsed_time(b) for a, b in zip(ev, ev[1:])]
    print("C after event elapsed_time :", " ".join(f"{block():.3f}" for _ in range(3)), flush=True)
    import gc
    gc.collect()
    print("C' after gc.collect        :", " ".join(f"{block():.3f}" for _ in range(3)), flush=True)
    # D: a monitoring tool polling the GPU next to the run (the driver samples rocm-smi every ~5 s)
    poll = subprocess.Popen(["bash", "-c", "while true; do rocm-smi --showuse --showpower --showclocks --showmeminfo vram --json "
                             "> /dev/null 2>&1; sleep 0.2; done"])
    try:
        time.sleep(1.0)
        vals = [block() for _ in range(40)]
    finally:
        poll.terminate()
        poll.wait()
    print("D rocm-smi polling beside  :", " ".join(f"{v:.3f}" for v in vals), flush=True)
    print("A again                    :", " ".join(f"{block():.3f}" for _ in range(4)), flush=True)


if __name__ == "__main__":
    main()
