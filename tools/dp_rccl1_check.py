"""The data-parallel step under a REAL RCCL process group on the one GPU of a test box: a one-rank ``nccl`` group
(``IMMUNOSTRUCT_FORCE_COLLECTIVE=1``; two ranks cannot share a GPU under RCCL).

What runs for real here and nowhere else without a multi-GPU node: ``init_process_group("nccl")``, the gradient buckets'
``all_reduce(async_op=True)`` on RCCL's own stream between captured HIP graphs, ``work.wait()`` handing the bucket back to the
compute stream, ``time_all_reduce`` and the engine's form selection (``IMMUNOSTRUCT_DP_OVERLAP=auto``: every candidate replayed
with the collectives in place).  With one rank the reduced gradient IS the local gradient, so every captured form must leave
the weights of an eager, unpacked, collective-free run after 3 Adam steps (the ``tools/dp_parity_check.py`` assertion).

    IMMUNOSTRUCT_FORCE_COLLECTIVE=1 python tools/dp_rccl1_check.py
"""
import copy
import os
import sys
import unittest.mock as mock

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from immunostruct_amd import distributed as D, optim  # noqa: E402
from immunostruct_amd.data import DeviceResidentDataset, SyntheticImmunoDataset  # noqa: E402
from immunostruct_amd.engine import CapturedTrainStep  # noqa: E402
from immunostruct_amd.utils import Losses  # noqa: E402
from tools.dp_parity_check import LR, VAE_IN, compare, make_model  # noqa: E402


class Counted:
    """counts the collectives issued through torch.distributed and the waits on their work handles"""

    def __init__(self):
        self.calls = self.async_calls = self.waits = 0
        self._all_reduce = dist.all_reduce

    def all_reduce(self, tensor, *a, **k):
        self.calls += 1
        work = self._all_reduce(tensor, *a, **k)
        if k.get("async_op"):
            self.async_calls += 1
            assert work is not None, "async all_reduce returned no work handle"
            outer = self

            class W:
                def wait(self_inner, *aa, **kk):
                    outer.waits += 1
                    return work.wait(*aa, **kk)
            return W()
        return work


def main():
    assert os.environ.get("IMMUNOSTRUCT_FORCE_COLLECTIVE") == "1"
    rank, local_rank, world = D.init_from_env()
    assert world == 1 and dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device("cuda", 0)
    bsz, steps = 32, 3
    ds = SyntheticImmunoDataset(bsz * steps, seed=11)
    dds = DeviceResidentDataset(ds, dev)
    losses = Losses(VAE_IN, ds.class_weights, sequence=True)

    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final = m(g, seq, prop)
        return losses.regression_loss(recon, seq, mu, logvar, final, y)

    ids = lambda s: torch.arange(s * bsz, (s + 1) * bsz, device=dev)
    start = make_model(dev, 7)
    # the yardstick: eager, gradients where autograd leaves them, no bucket, no collective
    ref = copy.deepcopy(start)
    ref.train()
    ropt = optim.Adam(ref.parameters(), lr=LR)
    whole = dds.new_batch(bsz)
    for s in range(steps):
        g, seq, prop, y = dds.gather_into(ids(s), *whole)
        ropt.zero_grad(set_to_none=True)
        forward_loss(ref, g, seq, prop, y).backward()
        ropt.step()
    torch.cuda.synchronize()
    moved = compare(ref, start, "sanity", float("inf"))
    assert moved > 0.5 * LR

    # RCCL itself: an out-of-place collective must move data even with one rank (the in-place all-reduce of one rank may be elided)
    src = torch.randn(1 << 20, device=dev)
    out = torch.empty_like(src)
    dist.all_gather_into_tensor(out, src)
    red = src.clone()
    dist.all_reduce(red)
    torch.cuda.synchronize()
    assert torch.equal(out, src) and torch.equal(red, src)

    from immunostruct_amd import engine
    for mode in ("1", "0", "auto", "graph", "graph2", "graph-auto"):
        # "graph" / "graph2": the step as ONE captured graph with RCCL's collectives inside (serial / two-stage), forced;
        # "graph-auto": every form captured and timed, the fastest kept.  The first three: the multi-graph forms alone
        engine._ONE_GRAPH = {"graph": "1", "graph2": "1", "graph-auto": "auto"}.get(mode, "0")
        os.environ["IMMUNOSTRUCT_DP_OVERLAP"] = {"graph": "0", "graph2": "1", "graph-auto": "auto"}.get(mode, mode)
        model = copy.deepcopy(start)
        model.train()
        opt = optim.Adam(model.parameters(), lr=LR)
        reducer = D.FlatGradReducer(model.parameters(), world=1, always_pack=True)
        assert reducer._collective and reducer.packing
        buf = dds.new_batch(bsz)
        dds.gather_into(ids(0), *buf)
        cnt = Counted()
        with mock.patch.object(dist, "all_reduce", cnt.all_reduce):
            cap = CapturedTrainStep(model, opt, reducer, forward_loss, buf, edge_capacity=bsz * dds.max_edges, warmup=1,
                                    preserve_state=True)
            if mode == "1":
                assert cap.two_stage and len(reducer.buckets) == 2
            if mode == "0":
                assert not cap.two_stage
            if mode == "auto":
                assert cap.dp_times is not None and len(cap.dp_times["two_stage_ms_by_reserved_cus"]) >= 1
            before = (cnt.calls, cnt.async_calls, cnt.waits)
            for s in range(steps):
                dds.gather_into(ids(s), cap.sgraph, cap.seq, cap.prop, cap.y)
                cap.replay()
            torch.cuda.synchronize()
        per_step = [(b - a) / steps for a, b in zip(before, (cnt.calls, cnt.async_calls, cnt.waits))]
        nb = len(reducer.buckets)
        if mode in ("graph", "graph2"):
            assert cap.one_graph_error is None, f"the collectives could not be captured: {cap.one_graph_error}"
            assert cap.one_graph == mode, (cap.one_graph, sorted(cap.graph_c))
        if cap.one_graph:
            assert per_step[0] == 0, f"{per_step[0]} collectives issued from the host per replayed one-graph step"
        else:
            assert per_step[0] == nb, f"{per_step[0]} collectives per replayed step, {nb} buckets"
            if cap.two_stage or cap._split_update():
                assert per_step[1] >= 1 and per_step[2] == per_step[1], f"async collectives {per_step[1]}, waits {per_step[2]} per step"
        worst = compare(model, ref, f"captured data-parallel step (IMMUNOSTRUCT_DP_OVERLAP={mode}) under a 1-rank RCCL group vs eager",
                        2e-2 * LR * steps)
        timing = D.time_all_reduce(reducer)
        assert timing is not None and len(timing) == nb and all(t["ms"] > 0 for t in timing)
        print(f"mode {mode}: form {cap.one_graph or ('two-stage' if cap.two_stage else 'serial')}, one-graph error {cap.one_graph_error}, buckets {[int(b['flat'].numel()) for b in reducer.buckets]}, "
              f"collectives/step {per_step[0]:.0f} (async {per_step[1]:.0f}, waits {per_step[2]:.0f}), max parameter difference {worst:.3e}, "
              f"standalone all-reduce {timing}, tuned {cap.dp_times}", flush=True)
        cap.close()      # the graphs of this form (RCCL's nodes among them) go before the next engine is built
        del cap
    print("RCCL 1-RANK CHECK OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    with mock.patch("torch.randn_like", torch.zeros_like):
        main()
