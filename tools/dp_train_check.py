"""Data-parallel ``train_model_device`` check, one process per rank (torch.distributed.run): trains two epochs, then
verifies that all ranks hold identical parameters, that they moved, and that the shards partition every epoch.

    IMMUNOSTRUCT_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \\
        --master-port 29541 tools/dp_train_check.py          (several ranks may share one GPU with the gloo backend)"""
import os
import sys
import tempfile
from types import SimpleNamespace

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from immunostruct_amd import distributed as D, optim  # noqa: E402
from immunostruct_amd.data import DeviceResidentDataset, SyntheticImmunoDataset  # noqa: E402
from immunostruct_amd.models import model_map  # noqa: E402
from immunostruct_amd.procedures import train_model_device  # noqa: E402
from immunostruct_amd.utils import Losses  # noqa: E402


def main():
    rank, local_rank, world = D.init_from_env()
    dev_index = int(os.environ.get("IMMUNOSTRUCT_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ds = SyntheticImmunoDataset(112, seed=3)
    dds = DeviceResidentDataset(ds, dev)
    torch.manual_seed(100 + rank)            # different initial weights per rank: the broadcast must make them equal
    model = model_map["HybridModelv2"](vae_input_dim=283 * 21, device=dev).to(dev)
    start = {k: v.detach().clone() for k, v in model.state_dict().items()}
    opt = optim.Adam(model.parameters(), lr=1e-4)
    losses = Losses(283 * 21, ds.class_weights, sequence=True)
    tmp = tempfile.mkdtemp()
    cfg = SimpleNamespace(batch_size=16, num_epochs=2, model_save_path_pretrain=os.path.join(tmp, f"m{rank}.pt"),
                          model_save_path_finetune=os.path.join(tmp, f"f{rank}.pt"))
    tr, va = train_model_device(cfg, dev, model, dds, list(range(96)), list(range(96, 112)), opt, losses.regression_loss, seed=5)
    assert all(map(lambda v: v == v, tr + va)), "non-finite loss"
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    spread = max(float((g - gathered[0]).abs().max()) for g in gathered)
    moved = max(float((model.state_dict()[k] - start[k].to(dev)).abs().max()) for k in start)
    saved = os.path.isfile(cfg.model_save_path_pretrain)
    print(f"rank {rank}/{world}: train {tr} val {va} | max parameter spread across ranks {spread:.3e} | moved {moved:.3e} | "
          f"checkpoint written {saved}", flush=True)
    assert spread == 0.0, "ranks diverged"
    assert moved > 0
    assert saved == (rank == 0)
    # the shards of an epoch partition its (world-multiple) permutation
    perm = torch.randperm(96, generator=torch.Generator().manual_seed(5))
    shards = [perm[: (96 // world) * world][r::world] for r in range(world)]
    assert sorted(torch.cat(shards).tolist()) == sorted(perm[: (96 // world) * world].tolist())
    dist.barrier()
    if rank == 0:
        print("DP TRAIN CHECK OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
