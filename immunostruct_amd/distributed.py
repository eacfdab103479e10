"""Data-parallel training: one process per GPU, RCCL all-reduce of one or two flat gradient buckets.

The reference is single-process; graphs are independent units, so the batched-graph
train step shards over ranks with no data-path collective: rank r trains on ``perm[r::world]`` of each
epoch's seeded permutation and the only exchange is the gradient all-reduce (25 MB fp32 for
HybridModelv2).  On MI355X xGMI is point-to-point (7 links per GPU), so the whole model goes out as a
single bucket -- one large collective instead of many small ones.

``torch.distributed`` backend "nccl" IS RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env; returns (rank, local_rank, world).

    ``IMMUNOSTRUCT_FORCE_COLLECTIVE=1`` with one rank creates a ONE-RANK RCCL group (the whole launch path of the data-parallel
    step -- ``init_process_group("nccl")``, asynchronous ``all_reduce`` on RCCL's stream, ``work.wait()`` between captured
    graphs -- on the single GPU of a test box).  ``IMMUNOSTRUCT_DIST_MAX_NCHANNELS=k`` bounds RCCL's channel count (one
    persistent workgroup per channel) -- opt-in: the only measurement behind a bound is the single-GPU emulation
    (tools/dp_overlap_emulation.py), and on a real xGMI node fewer channels can cost all-reduce bandwidth."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    forced = world == 1 and os.environ.get("IMMUNOSTRUCT_FORCE_COLLECTIVE") == "1"
    if (world > 1 or forced) and not dist.is_initialized():
        # the host driver of these boxes supports dmabuf IPC only: without this RCCL / cross-process tensor sharing fails with
        # "hipIpcGetMemHandle: invalid argument" (exported on the build and GPU boxes already; a launcher that scrubs the environment
        # must not lose it).  Set before the first HIP call of the process.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # IMMUNOSTRUCT_DIST_BACKEND=gloo lets the multi-rank code path be exercised on a single GPU
            backend = os.environ.get("IMMUNOSTRUCT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("IMMUNOSTRUCT_FORCE_DEVICE", local_rank)))
            cap = os.environ.get("IMMUNOSTRUCT_DIST_MAX_NCHANNELS")
            if cap:
                os.environ["NCCL_MAX_NCHANNELS"] = str(int(cap))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def collectives_forced():
    """one rank, but the collectives are issued all the same (IMMUNOSTRUCT_FORCE_COLLECTIVE=1 under an initialised group)"""
    return os.environ.get("IMMUNOSTRUCT_FORCE_COLLECTIVE") == "1" and dist.is_initialized()


class FlatGradReducer:
    """Flat fp32 gradient buckets, each all-reduced with a single collective.

    ``zero()`` drops the gradients (autograd then *assigns* fresh ones: no per-parameter add kernels);
    ``all_reduce_mean()`` packs them into the persistent flat bucket(s) with ONE concatenation kernel each,
    all-reduces (RCCL over xGMI: the whole 25 MB model as one or two collectives) and re-points every
    ``.grad`` at its slice of a bucket, which is what the optimizer then reads.  With one rank and
    ``always_pack=False`` all of this is a no-op apart from dropping the gradients.

    ``split(late)`` cuts the parameters into two buckets (everything else / ``late``); ``reduce_bucket(i, async_op)``
    handles one of them -- the engine all-reduces the first bucket (sequence branch + heads: 96 % of the bytes)
    asynchronously while the backward of the graph branch, whose gradients form the second bucket, is still running.
    """

    def __init__(self, parameters, world=None, always_pack=False):
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.always_pack = always_pack
        # issue the collectives even in a one-rank process group (the RCCL launch path on the one GPU of a test box)
        self._collective = self.world > 1 or collectives_forced()
        self._packing = self.world > 1 or always_pack or self._collective
        self.buckets = []
        # True: reduce_bucket divides by the world size (its own pass over the bucket); the engine switches it off when
        # the optimizer applies 1 / world itself (immunostruct_amd.optim: ``grad_scale``)
        self.divide = True
        if self._packing:
            self._make_buckets([self.params])

    def _make_buckets(self, groups):
        dev = self.params[0].device
        self.buckets = []
        for g in groups:
            if not g:
                continue
            flat = torch.zeros(sum(p.numel() for p in g), dtype=torch.float32, device=dev)
            views, off = [], 0
            for p in g:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                # gradient producers that own their output buffer (functional.LinearSmallBatchFn: the two VAE matrices,
                # 96 % of the bytes) write straight into the bucket: no copy for them when the bucket is packed
                p._grad_dest = views[-1]
            self.buckets.append(dict(params=g, flat=flat, views=views, sources=None))

    @property
    def packing(self):
        return self._packing

    @property
    def flat(self):
        """the (first) flat bucket -- kept for callers that inspect it"""
        return self.buckets[0]["flat"] if self.buckets else None

    def split(self, late):
        """two buckets: [parameters not in ``late``] and [``late``]; call before any source binding"""
        if not self._packing:
            return
        late_ids = set(id(p) for p in late)
        self._make_buckets([[p for p in self.params if id(p) not in late_ids], [p for p in self.params if id(p) in late_ids]])

    def zero(self):
        for p in self.params:
            p.grad = None

    def bind_sources(self, bucket=None, packed=False):
        """Remember the CURRENT .grad tensors as the pack sources (used with captured graphs, where the
        backward always writes the same buffers while ``.grad`` is re-pointed at the bucket).  ``packed``: the graph that wrote
        them packs the bucket itself (its last nodes) -- the bit belongs to THESE sources: binding others resets it, so that
        :meth:`reduce_bucket` never skips the pack for gradients of a graph that did not pack."""
        for i, b in enumerate(self.buckets):
            if bucket is None or bucket == i:
                b["sources"] = [p.grad for p in b["params"]]
                b["packed"] = bool(packed)

    def sources(self, value=None):
        """get (a copy of) / set the bound pack sources of all buckets, each WITH its packed bit: [(sources, packed), ...] -- the
        engine switches between captured forms"""
        if value is None:
            return [(list(b["sources"]) if b["sources"] is not None else None, bool(b.get("packed"))) for b in self.buckets]
        for b, v in zip(self.buckets, value):
            b["sources"], b["packed"] = (v if isinstance(v, tuple) else (v, False))

    @contextlib.contextmanager
    def live_gradients(self):
        """an EAGER step between replays of a captured one (a trailing partial batch): pack from the ``.grad`` tensors the
        eager backward produces, not from the captured graph's gradient buffers the sources are bound to"""
        saved = self.sources()
        for b in self.buckets:
            b["sources"], b["packed"] = None, False
        try:
            yield self
        finally:
            self.sources(saved)

    def pack(self, i, from_grad=False):
        """one multi-tensor copy of bucket i's gradients that are not already in place (``from_grad``: from the current ``.grad``
        tensors even when sources are bound -- the engine packs INSIDE the graph it is capturing, whose gradient buffers those are)"""
        b = self.buckets[i]
        src = b["sources"] if (b["sources"] is not None and not from_grad) else [p.grad for p in b["params"]]
        dsts, srcs = [], []
        for g, v in zip(src, b["views"]):
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                dsts.append(v)
                srcs.append(g.view_as(v) if g.is_contiguous() else g.contiguous().view_as(v))
        if srcs:
            torch._foreach_copy_(dsts, srcs)

    def reduce_bucket(self, i, async_op=False, prepacked=False):
        """pack bucket i (unless the captured graph that produced the bound sources packed it itself: ``packed``; or the caller
        just did: ``prepacked`` -- the one-graph forms, whose capture holds pack AND collective), all-reduce
        (SUM; divided by the world size here unless ``divide`` is off), re-point the gradients; returns the work handle"""
        b = self.buckets[i]
        if not prepacked and not (b.get("packed") and b["sources"] is not None):
            self.pack(i)
        work = None
        if self._collective:
            if self.divide:
                b["flat"].div_(self.world)
            work = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=async_op)
        for p, v in zip(b["params"], b["views"]):
            p.grad = v
        return work

    def all_reduce_mean(self, async_op=False):
        work = None
        for i in range(len(self.buckets)):
            work = self.reduce_bucket(i, async_op=async_op)
        return work


def time_all_reduce(reducer, repeats=10):
    """standalone duration (ms, max over ranks) of the all-reduce of every gradient bucket -- nothing beside it on the GPU; what a
    data-parallel bench line needs to explain its own efficiency.  None when the reducer issues no collectives (one rank
    without IMMUNOSTRUCT_FORCE_COLLECTIVE)."""
    if not (dist.is_initialized() and getattr(reducer, "_collective", False)) or not reducer.buckets:
        return None
    import time
    out = []
    for b in reducer.buckets:
        scratch = torch.zeros_like(b["flat"])
        for k in range(2 + repeats):
            if k == 2:
                dist.barrier()
                torch.cuda.synchronize() if scratch.is_cuda else None
                t0 = time.perf_counter()
            dist.all_reduce(scratch, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize() if scratch.is_cuda else None
        t = torch.tensor([(time.perf_counter() - t0) / repeats * 1e3], dtype=torch.float64, device=scratch.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out.append(dict(floats=int(scratch.numel()), ms=round(float(t.item()), 4)))
    return out


def broadcast_parameters(module, src=0):
    """Make all ranks start from rank ``src``'s weights (also after ``load_trained(new_head=True)``)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not collectives_forced()):
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
