"""Data-parallel training: one process per GPU, RCCL all-reduce of ONE flat gradient bucket.

The reference is single-process; graphs are independent units, so the batched-graph
train step shards over ranks with no data-path collective: rank r trains on ``perm[r::world]`` of each
epoch's seeded permutation and the only exchange is the gradient all-reduce (25 MB fp32 for
HybridModelv2).  On MI355X xGMI is point-to-point (7 links per GPU), so the whole model goes out as a
single bucket -- one large collective instead of many small ones.

``torch.distributed`` backend "nccl" IS RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # IMMUNOSTRUCT_DIST_BACKEND=gloo lets the multi-rank code path be exercised on a single GPU
            backend = os.environ.get("IMMUNOSTRUCT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class FlatGradReducer:
    """One flat fp32 gradient bucket per step, all-reduced with a single collective.

    ``zero()`` drops the gradients (autograd then *assigns* fresh ones: no per-parameter add kernels);
    ``all_reduce_mean()`` packs them into the persistent flat bucket with ONE concatenation kernel,
    all-reduces it (RCCL over xGMI: the whole 25 MB model as a single collective) and re-points every
    ``.grad`` at its slice of the bucket, which is what the optimizer then reads.  With one rank and
    ``always_pack=False`` both calls are no-ops apart from dropping the gradients.
    """

    def __init__(self, parameters, world=None, always_pack=False):
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.always_pack = always_pack
        dev, total = self.params[0].device, sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev) if (self.world > 1 or always_pack) else None
        self.sources = None   # gradient tensors to pack (defaults to the parameters' current .grad)

    @property
    def packing(self):
        return self.flat is not None

    def zero(self):
        for p in self.params:
            p.grad = None

    def bind_sources(self):
        """Remember the CURRENT .grad tensors as the pack sources (used with captured graphs, where the
        backward always writes the same buffers while ``.grad`` is re-pointed at the bucket)."""
        self.sources = [p.grad for p in self.params]

    def all_reduce_mean(self, async_op=False):
        if not self.packing:
            return None
        src = self.sources if self.sources is not None else [p.grad for p in self.params]
        pieces = [(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(src, self.params)]
        torch.cat(pieces, out=self.flat)
        work = None
        if self.world > 1:
            self.flat.div_(self.world)
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        return work


def broadcast_parameters(module, src=0):
    """Make all ranks start from rank ``src``'s weights (also after ``load_trained(new_head=True)``)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src)


def shard_indices(num_items, epoch, seed, rank, world, drop_last=True):
    """Seeded epoch permutation, strided over ranks (the DistributedSampler-equivalent)."""
    gen = torch.Generator().manual_seed(seed * 1000003 + epoch)
    perm = torch.randperm(num_items, generator=gen)
    if drop_last:
        perm = perm[: (num_items // world) * world]
    return perm[rank::world]
