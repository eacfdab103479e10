"""Data-parallel logic on CPU: 2 and 4 gloo ranks must reproduce the 1-rank gradient of the full batch."""
import pytest
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from immunostruct_amd.distributed import FlatGradReducer, broadcast_parameters, shard_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.SiLU(), torch.nn.Linear(5, 1))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)           # ranks start different ...
    net = _net() if rank == 0 else torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.SiLU(), torch.nn.Linear(5, 1))
    broadcast_parameters(net)               # ... and are made identical
    red = FlatGradReducer(net.parameters())
    data = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10
    y = torch.linspace(-1, 1, 8)
    idx = shard_indices(8, epoch=3, seed=1, rank=rank, world=world)
    red.zero()
    loss = ((net(data[idx]).squeeze(1) - y[idx]) ** 2).mean()
    loss.backward()
    red.all_reduce_mean()
    for p in net.parameters():   # the optimizer must see the reduced bucket
        assert p.grad.data_ptr() >= red.flat.data_ptr()
    if rank == 0:
        torch.save({"flat": red.flat.clone(), "idx": idx}, out)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_gradient_equals_single_rank(tmp_path, world):
    """the shards of the epoch permutation (``perm[r::world]``) cover the batch, the flat bucket's all-reduce + 1 / world is the
    full-batch mean gradient -- for two ranks and for four (the logic beyond world = 2)"""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    assert got["idx"].numel() == 8 // world
    net = _net()
    data = torch.arange(8 * 6, dtype=torch.float32).reshape(8, 6) / 10
    y = torch.linspace(-1, 1, 8)
    ((net(data).squeeze(1) - y) ** 2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.allclose(got["flat"], ref, rtol=1e-5, atol=1e-7)


def test_shards_partition_the_epoch_permutation():
    parts = [shard_indices(103, epoch=2, seed=5, rank=r, world=4) for r in range(4)]
    allidx = torch.cat(parts)
    assert allidx.numel() == 100 and allidx.unique().numel() == 100
    assert not torch.equal(parts[0], shard_indices(103, epoch=3, seed=5, rank=0, world=4))
