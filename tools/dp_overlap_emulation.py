"""Single-GPU emulation of the data-parallel step: the gradient all-reduce is replaced by a kernel of a chosen duration
on its own stream (what an RCCL collective is to the compute stream), so that the serial and the two-stage (overlapped)
forms of engine.CapturedTrainStep can be compared without a second GPU.   python tools/dp_overlap_emulation.py [us ...]"""
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
import immunostruct_amd.distributed as D  # noqa: E402
from immunostruct_amd import optim, synthetic  # noqa: E402
from immunostruct_amd.engine import CapturedTrainStep  # noqa: E402
from immunostruct_amd.graph import PackedGraphBatch  # noqa: E402
from immunostruct_amd.models import model_map  # noqa: E402
from immunostruct_amd.utils import Losses  # noqa: E402

dev = torch.device("cuda:0")
VAE_IN = synthetic.SEQ_LEN * synthetic.SEQ_ALPHABET
comm_stream = torch.cuda.Stream()
CYCLES_PER_US = 2400.0     # torch.cuda._sleep counts shader-clock ticks; calibrated below
state = {"us": 0.0}


class FakeWork:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def fake_all_reduce(t, op=None, async_op=False):
    """a 'collective' of state['us'] microseconds per 25 MB, scaled by the bucket size, on the communication stream"""
    comm_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(comm_stream):
        torch.cuda._sleep(int(state["us"] * CYCLES_PER_US * t.numel() / 6.33e6) + 1)
        ev = torch.cuda.Event()
        ev.record(comm_stream)
    work = FakeWork(ev)
    if not async_op:
        work.wait()
        return None
    return work


def calibrate():
    global CYCLES_PER_US
    torch.cuda._sleep(1000)        # first call: module load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda._sleep(20_000_000)
    torch.cuda.synchronize()
    CYCLES_PER_US = 20_000_000 / ((time.perf_counter() - t0) * 1e6)


def main():
    calibrate()
    D.dist.all_reduce = fake_all_reduce
    raws = [synthetic.make_batch(128, seed=100 + i, deg_extra=2) for i in range(3)]
    batches = [(PackedGraphBatch.from_raw(r, device=dev), torch.from_numpy(r.one_hot_sequence()).to(dev),
                torch.from_numpy(r.prop).to(dev), torch.from_numpy(r.y_reg).to(dev)) for r in raws]
    losses = Losses(VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)

    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final = m(g, seq, prop)
        return losses.regression_loss(recon, seq, mu, logvar, final, y)

    import os
    print(f"sleep calibration: {CYCLES_PER_US:.1f} ticks/us")
    print(f"{'all-reduce of 25 MB':>22} {'serial':>10} {'two-stage':>10} {'auto picks':>12}")
    for us in [float(a) for a in sys.argv[1:]] or [0.0, 100.0, 200.0, 350.0, 500.0]:
        state["us"] = us
        res = {}
        for mode in ("0", "1", "auto"):
            os.environ["IMMUNOSTRUCT_DP_OVERLAP"] = mode
            model = model_map["HybridModelv2"](vae_input_dim=VAE_IN, device=dev).to(dev)
            model.train()
            red = D.FlatGradReducer(model.parameters(), world=2)       # packing + "collectives" (the fake above); grads / 2
            opt = optim.Adam(model.parameters(), lr=1e-3)
            eng = CapturedTrainStep(model, opt, red, forward_loss, batches[0], edge_capacity=max(r.num_edges for r in raws))
            for i in range(5):
                eng(*batches[i % 3])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(30):
                eng(*batches[i % 3])
            torch.cuda.synchronize()
            res[mode] = ((time.perf_counter() - t0) / 30 * 1e3, eng.two_stage)
        print(f"{us:>19.0f} us {res['0'][0]:>8.3f}ms {res['1'][0]:>8.3f}ms {'two-stage' if res['auto'][1] else 'serial':>12} ({res['auto'][0]:.3f} ms)")


if __name__ == "__main__":
    main()
