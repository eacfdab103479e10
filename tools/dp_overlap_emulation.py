"""Single-GPU emulation of the data-parallel step: the gradient all-reduce is replaced by a stand-in kernel on its own stream
that behaves like RCCL's towards the compute stream -- CHANNELS persistent workgroups of 512 threads (RCCL runs one workgroup per
channel; NCCL_MIN_NCHANNELS / NCCL_MAX_NCHANNELS bound the count) that stream the bucket twice in place (the local HBM
traffic of reduce-scatter + all-gather) and hold their CU slots for the emulated duration (csrc/abi_misc.hip
``is_debug_emulated_collective``).  Round 2's stand-in was ``torch.cuda._sleep`` -- one idle thread, no slots, no bandwidth -- which
made the overlapped form look better than it can be.  For every (duration, channels) the serial form, the two-stage form with 0 and
with R reserved CUs (the layer kernels' grids leave 2 R workgroup slots free, functional.RESERVED_CUS) and the engine's own choice
are timed.

    python tools/dp_overlap_emulation.py [--channels 16,32] [--reserved 0,16,32] [us ...]      -> one JSON line per row"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import immunostruct_amd.distributed as D  # noqa: E402
from immunostruct_amd import _lib, optim, synthetic  # noqa: E402
from immunostruct_amd.engine import CapturedTrainStep  # noqa: E402
from immunostruct_amd.graph import PackedGraphBatch  # noqa: E402
from immunostruct_amd.models import model_map  # noqa: E402
from immunostruct_amd.utils import Losses  # noqa: E402

dev = torch.device("cuda:0")
VAE_IN = synthetic.SEQ_LEN * synthetic.SEQ_ALPHABET
comm_stream = torch.cuda.Stream()
state = {"us": 0.0, "channels": 16, "elapsed": None, "floor_us": 0.0}


class FakeWork:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def fake_all_reduce(t, op=None, async_op=False):
    """a 'collective' of state['us'] microseconds per 25 MB (scaled by the bucket size) on the communication stream"""
    comm_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(comm_stream):
        ticks = int(state["us"] * 100.0 * t.numel() / 6.33e6)
        if state["us"] > 0:
            ticks = max(ticks, int(state["floor_us"] * 100.0))      # a small all-reduce is latency-bound: it has a floor
        if ticks > 0:
            if state["elapsed"] is None:
                state["elapsed"] = torch.zeros(64, dtype=torch.int64, device=t.device)
            big = t.numel() > 3_000_000      # the streaming time of the 24 MB bucket is recorded (the small one is negligible)
            _lib.check(_lib.load().is_debug_emulated_collective(_lib.ptr(t), t.numel(), state["channels"], 2, ticks,
                                                                _lib.ptr(state["elapsed"]) if big else None, _lib.stream_ptr()),
                       "is_debug_emulated_collective")
        ev = torch.cuda.Event()
        ev.record(comm_stream)
    work = FakeWork(ev)
    if not async_op:
        work.wait()
        return None
    return work


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("us", nargs="*", type=float, default=[0.0, 100.0, 200.0, 350.0, 500.0])
    ap.add_argument("--channels", default="16,32")
    ap.add_argument("--reserved", default="0,16,32")
    ap.add_argument("--floor-us", type=float, default=40.0, help="minimum duration of any emulated all-reduce (the 0.7 MB bucket)")
    args = ap.parse_args()
    state["floor_us"] = args.floor_us
    D.dist.all_reduce = fake_all_reduce
    raws = [synthetic.make_batch(128, seed=100 + i, deg_extra=2) for i in range(3)]
    batches = [(PackedGraphBatch.from_raw(r, device=dev), torch.from_numpy(r.one_hot_sequence()).to(dev),
                torch.from_numpy(r.prop).to(dev), torch.from_numpy(r.y_reg).to(dev)) for r in raws]
    losses = Losses(VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)

    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final = m(g, seq, prop)
        return losses.regression_loss(recon, seq, mu, logvar, final, y)

    def run(mode, reserved):
        os.environ["IMMUNOSTRUCT_DP_OVERLAP"] = mode
        os.environ["IMMUNOSTRUCT_DP_RESERVED_CUS"] = reserved
        model = model_map["HybridModelv2"](vae_input_dim=VAE_IN, device=dev).to(dev)
        model.train()
        red = D.FlatGradReducer(model.parameters(), world=2)       # packing + "collectives" (the stand-in above); grads / 2
        red._collective = True
        opt = optim.Adam(model.parameters(), lr=1e-3)
        eng = CapturedTrainStep(model, opt, red, forward_loss, batches[0], edge_capacity=max(r.num_edges for r in raws))
        for i in range(5):
            eng(*batches[i % 3])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(30):
            eng(*batches[i % 3])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 30 * 1e3, eng

    run("0", "0")      # the process's first engine pays one-off costs (allocator, library handles): not a row
    for ch in [int(c) for c in args.channels.split(",")]:
        for us in args.us:
            state["us"], state["channels"] = us, ch
            row = {"allreduce_us_per_25MB": us, "floor_us": args.floor_us, "channels": ch,
                   "split_update": True, "serial_ms": round(run("0", "0")[0], 3)}
            for r in args.reserved.split(","):
                row[f"two_stage_reserved{r}_ms"] = round(run("1", r)[0], 3)
            ms, eng = run("auto", args.reserved)
            row["auto"] = {"ms": round(ms, 3), "form": "two-stage" if eng.two_stage else "serial",
                           "reserved_cus": eng.reserved if eng.two_stage else None}
            if state["elapsed"] is not None:
                row["standin_streaming_us"] = round(float(state["elapsed"][:ch].max()) / 100.0, 1)      # must stay below the duration
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
