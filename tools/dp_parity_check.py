"""Data-parallel parity on the HIP model, one process per rank (SURVEY.md section 4 iv), two checks:

``split``: N ranks (2 or 4) x B/N graphs through the CAPTURED two-stage step (bucket 0 all-reduced under the stack backward,
           1/world inside Adam) == 1 rank x B graphs EAGER, same weights after 3 Adam steps.  The whole batch is the
           concatenation of the ranks' halves; the loss is a mean over the batch, so the mean of the two half-batch
           gradients is the full-batch gradient.
``split-auto``: the same with the form chosen by timing (IMMUNOSTRUCT_DP_OVERLAP=auto): every rank must arrive at the same form.
           Both gather the chosen form from all ranks and require the ranks' parameters to be bit-identical after the steps.
``tail`` : ``procedures.train_model_device`` with a shard length that is NOT a multiple of the batch size (three replayed
           steps + one eager trailing step per epoch, which must pack the eager step's own gradients) == an all-eager
           data-parallel loop written out here with plain ``dist.all_reduce`` per parameter.

The reparameterisation noise is replaced by zeros and dropout is switched off on both sides (their random streams differ
between a captured and an eager run); everything else is the product path.

    IMMUNOSTRUCT_DIST_BACKEND=gloo IMMUNOSTRUCT_FORCE_DEVICE=0 python -m torch.distributed.run --nproc-per-node 2 \\
        --master-addr 127.0.0.1 --master-port 29561 tools/dp_parity_check.py split|tail
"""
import copy
import os
import sys
import tempfile
import unittest.mock as mock
from types import SimpleNamespace

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from immunostruct_amd import distributed as D, optim  # noqa: E402
from immunostruct_amd.data import DeviceResidentDataset, SyntheticImmunoDataset  # noqa: E402
from immunostruct_amd.engine import CapturedTrainStep  # noqa: E402
from immunostruct_amd.models import model_map  # noqa: E402
from immunostruct_amd.procedures import train_model_device  # noqa: E402
from immunostruct_amd.utils import Losses  # noqa: E402

VAE_IN = 283 * 21
LR = 1e-3


def make_model(dev, seed):
    torch.manual_seed(seed)
    model = model_map["HybridModelv2"](vae_input_dim=VAE_IN, device=dev).to(dev)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def compare(a, b, what, tol):
    worst = 0.0
    for (k, p), q in zip(a.named_parameters(), b.parameters()):
        moved = float((p.detach() - q.detach()).abs().max())
        worst = max(worst, moved)
        assert moved <= tol, f"{what}: {k} differs by {moved:.3e} (> {tol:.1e})"
    return worst


def check_split(rank, world, dev, overlap="1"):
    total, steps = 32, 3
    per = total // world                            # graphs per rank and step (world 2: 16, world 4: 8)
    assert per * world == total
    ds = SyntheticImmunoDataset(total * steps, seed=11)
    dds = DeviceResidentDataset(ds, dev)
    losses = Losses(VAE_IN, ds.class_weights, sequence=True)
    model = make_model(dev, 7)                      # same seed on every rank
    ref = copy.deepcopy(model)
    start = copy.deepcopy(model)

    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final = m(g, seq, prop)
        return losses.regression_loss(recon, seq, mu, logvar, final, y)

    os.environ["IMMUNOSTRUCT_DP_OVERLAP"] = overlap     # "1": force the two-stage form; "auto": timed on all ranks, one choice for all
    opt = optim.Adam(model.parameters(), lr=LR)
    reducer = D.FlatGradReducer(model.parameters(), world=world)
    ids = lambda s, r: torch.arange(s * total + r * per, s * total + (r + 1) * per, device=dev)
    buf = dds.new_batch(per)
    dds.gather_into(ids(0, rank), *buf)
    model.train()
    cap = CapturedTrainStep(model, opt, reducer, forward_loss, buf, edge_capacity=per * dds.max_edges, warmup=1,
                            preserve_state=True)
    assert len(reducer.buckets) == 2 and (cap.two_stage or overlap == "auto")
    # every rank replays the SAME form (ranks replaying different forms would stop matching their collectives): the choice --
    # forced, or timed with the maximum over the ranks -- is gathered and compared
    forms = [None] * world
    dist.all_gather_object(forms, (bool(cap.two_stage), cap.reserved if cap.two_stage else None, cap.one_graph))
    assert all(f == forms[0] for f in forms), f"the ranks chose different forms: {forms}"
    if overlap == "auto":
        assert cap.dp_times is not None and cap.dp_times["serial_ms"] > 0 and cap.dp_times["two_stage_ms"] > 0
    for s in range(steps):
        dds.gather_into(ids(s, rank), cap.sgraph, cap.seq, cap.prop, cap.y)
        cap.replay()
    torch.cuda.synchronize()
    # the ranks' parameters after the steps are the same BITS (same reduced gradients, same update)
    mine = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    root = mine.clone()
    dist.broadcast(root, src=0)
    assert torch.equal(mine, root), f"rank {rank}: parameters differ from rank 0's after {steps} steps"
    # one rank, whole batch, eager, no collective
    ref.train()
    ropt = optim.Adam(ref.parameters(), lr=LR)
    whole = dds.new_batch(total)
    for s in range(steps):
        g, seq, prop, y = dds.gather_into(torch.cat([ids(s, r) for r in range(world)]), *whole)
        ropt.zero_grad(set_to_none=True)
        forward_loss(ref, g, seq, prop, y).backward()
        ropt.step()
    torch.cuda.synchronize()
    moved = compare(ref, start, "sanity", float("inf"))
    worst = compare(model, ref, f"{world} ranks x B/{world} captured ({'two-stage' if cap.two_stage else 'serial'}) vs 1 rank x B eager", 2e-2 * LR * steps)
    print(f"rank {rank}: split parity ({world} ranks, form {forms[0]}): parameters moved {moved:.3e}, max difference {worst:.3e}", flush=True)
    assert moved > 0.5 * LR
    cap.close()


def check_tail(rank, world, dev):
    n_train, bsz = 104, 16                           # 52 per rank: three full batches + a tail of 4
    ds = SyntheticImmunoDataset(n_train + 8, seed=5)
    dds = DeviceResidentDataset(ds, dev)
    losses = Losses(VAE_IN, ds.class_weights, sequence=True)
    model = make_model(dev, 9)
    ref = copy.deepcopy(model)
    tmp = tempfile.mkdtemp()
    cfg = SimpleNamespace(batch_size=bsz, num_epochs=2, model_save_path_pretrain=os.path.join(tmp, f"m{rank}.pt"),
                          model_save_path_finetune=os.path.join(tmp, f"f{rank}.pt"),
                          step_random=None)      # (the patched torch.randn_like below must be what the captured step draws with)
    opt = optim.Adam(model.parameters(), lr=LR)
    train_model_device(cfg, dev, model, dds, list(range(n_train)), list(range(n_train, n_train + 8)), opt,
                       losses.regression_loss, seed=5)
    # the same schedule, all eager, gradients averaged parameter by parameter
    ropt = optim.Adam(ref.parameters(), lr=LR)
    gen = torch.Generator(device="cpu").manual_seed(5)
    index = torch.arange(n_train, device=dev)
    ref.train()
    bufs = {}
    for _ in range(cfg.num_epochs):
        perm = index[torch.randperm(n_train, generator=gen).to(dev)]
        perm = perm[: (n_train // world) * world][rank::world]
        for at in range(0, perm.numel(), bsz):
            idx = perm[at:at + bsz]
            b = int(idx.numel())
            if b not in bufs:
                bufs[b] = dds.new_batch(b)
            g, seq, prop, y = dds.gather_into(idx, *bufs[b])
            ropt.zero_grad(set_to_none=True)
            recon, mu, logvar, final = ref(g, seq, prop)
            losses.regression_loss(recon, seq, mu, logvar, final, y).backward()
            for p in ref.parameters():
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                dist.all_reduce(p.grad)
                p.grad /= world
            ropt.step()
    torch.cuda.synchronize()
    worst = compare(model, ref, "captured + eager tail vs all-eager data parallel", 2e-2 * LR)
    print(f"rank {rank}: tail parity: max parameter difference {worst:.3e}", flush=True)


def main():
    mode = sys.argv[1]
    rank, local_rank, world = D.init_from_env()
    dev_index = int(os.environ.get("IMMUNOSTRUCT_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    with mock.patch("torch.randn_like", torch.zeros_like):
        {"split": check_split, "split-auto": lambda r, w, d: check_split(r, w, d, overlap="auto"), "tail": check_tail}[mode](rank, world, dev)
    dist.barrier()
    if rank == 0:
        print("DP PARITY OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
