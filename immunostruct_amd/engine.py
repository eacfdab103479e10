"""hipGraph-captured train step + the static device-side batch it replays on.

The per-step work of this model is a few hundred short kernels (N = 24k nodes, E ~ 72k edges at
B = 128): launched eagerly the step is bound by host launch overhead, not by the GPU.  The engine
captures ``zero-grad -> forward -> loss -> backward [-> optimizer]`` ONCE into a HIP graph (through
``torch.cuda.CUDAGraph``; every HIP kernel of this package is enqueued on the capturing stream and
allocates only through torch's graph-aware allocator) and replays it per step.

Replay needs fixed addresses and launch geometry:

* ``StaticGraphBatch`` is a ``PackedGraphBatch`` whose node / edge / CSR arrays live in
  fixed-capacity device buffers.  All kernels take their edge ranges from the device-side
  ``rowptr`` arrays and size their grids by the (fixed, padded) node count, so a batch with fewer
  edges than the capacity replays correctly: slots past ``rowptr[N]`` are never touched.
* ``load(batch)`` copies a device-resident batch into those buffers (async D2D on the same stream);
  this is the on-GPU batcher's hand-over point.

With data parallelism the gradient all-reduce (RCCL) runs between the graphs: graph A = zero + forward +
loss + backward + bucket pack, all-reduce of the flat bucket(s), graph B = optimizer step per bucket
(``IMMUNOSTRUCT_DP_ONE_GRAPH``: the collectives inside one graph).
"""
from __future__ import annotations

import os

import torch

from .graph import CSRIndex, PackedGraphBatch


# Captures are thread-local: under a process group, ProcessGroupNCCL's watchdog thread polls the events of outstanding collectives
# (hipEventQuery) at any time -- in the default "global" mode such a call from ANOTHER thread invalidates the capture in progress
# ("operation not permitted when stream is capturing", first met under a one-rank RCCL group in round 4)
_CAPTURE = {"capture_error_mode": "thread_local"}
# The data-parallel step as ONE captured graph (round 5): pack, the all-reduce(s) and the optimizer update are captured together with
# forward and backward -- ProcessGroupNCCL's collectives are stream operations, so RCCL's kernels become nodes of the graph (the
# asynchronous form forks onto the process group's stream and joins at ``work.wait()``): one replay per step, no graph boundaries, no
# host-side wait.  "0" (default since round 6: the form has only ever run under a ONE-rank RCCL group -- no multi-GPU node was
# available to this build -- and a graph that holds RCCL's nodes must be released before the process group, `close()`); "auto":
# captured where the process group is "nccl", timed against the multi-graph forms on all ranks, kept if faster; "1": forced wherever it
# was captured (an error while capturing is kept in ``one_graph_error`` and the multi-graph forms run).
_ONE_GRAPH = os.environ.get("IMMUNOSTRUCT_DP_ONE_GRAPH", "0")
if _ONE_GRAPH not in ("auto", "0", "1"):
    raise ValueError("IMMUNOSTRUCT_DP_ONE_GRAPH must be auto, 0 or 1")


class StaticGraphBatch(PackedGraphBatch):
    """Fixed-capacity device buffers with the PackedGraphBatch surface."""

    def __init__(self, template: PackedGraphBatch, edge_capacity: int):
        dev = template.device
        n = template.num_nodes()
        super().__init__(torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int64, device=dev),
                         n, template._counts)
        self._device = dev
        self.edge_capacity = int(edge_capacity)
        fe = template.edata["edge_attr"].shape[1]
        self.ndata["x"] = torch.zeros_like(template.ndata["x"])
        self.edata["edge_attr"] = torch.zeros(self.edge_capacity, fe, dtype=torch.float32, device=dev)
        csr = object.__new__(CSRIndex)
        csr.rowptr_dst = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        csr.rowptr_src = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        csr.src_sorted = torch.zeros(self.edge_capacity, dtype=torch.int32, device=dev)
        csr.dst_sorted = torch.zeros(self.edge_capacity, dtype=torch.int32, device=dev)
        csr._chunks = {}     # filled on first use from the loaded rowptr, refreshed by every load()
        csr.pos_by_src = torch.zeros(self.edge_capacity, dtype=torch.int32, device=dev)
        csr.eperm = torch.zeros(0, dtype=torch.int64, device=dev)
        csr.num_nodes, csr.num_edges = n, self.edge_capacity
        self._csr = csr
        self._ea_csr = torch.zeros(self.edge_capacity, fe, dtype=torch.float32, device=dev)
        self._seg_ptr = template.seg_ptr().clone()

    @property
    def device(self):
        return self._device

    def num_edges(self):
        return self.edge_capacity

    def edge_feat_csr(self, edge_feat):
        return self._ea_csr

    def copy_pairs(self, g: PackedGraphBatch):
        """(source, destination) tensor pairs that hand a device-resident batch (same node layout,
        E <= capacity) over to the static buffers."""
        e = g.num_edges()
        if e > self.edge_capacity:
            raise ValueError(f"batch has {e} edges, static capacity is {self.edge_capacity}")
        if g.num_nodes() != self._num_nodes or g._counts != self._counts:
            raise ValueError("node layout differs from the captured batch")
        src = g.csr()
        return [(g.ndata["x"], self.ndata["x"]), (src.rowptr_dst, self._csr.rowptr_dst),
                (src.rowptr_src, self._csr.rowptr_src), (src.src_sorted, self._csr.src_sorted[:e]),
                (src.dst_sorted, self._csr.dst_sorted[:e]), (src.pos_by_src, self._csr.pos_by_src[:e]),
                (g.edge_feat_csr(g.edata["edge_attr"]), self._ea_csr[:e])] + \
               [(src.chunks(k), dst) for k, dst in self._csr._chunks.items()]

    def load(self, g: PackedGraphBatch):
        multi_copy(self.copy_pairs(g))

    def refresh_partitions(self):
        """Recompute the edge kernels' work partitions from the rowptr currently in the buffers (after the on-device
        batcher wrote a new batch): vectorised torch ops on the device, no host sync, capturable."""
        from . import _lib
        from .graph import CHUNK_SHARES, FULL_GRID_CHUNKS
        lib = _lib.load()
        for k, dst in self._csr._chunks.items():      # == graph.balanced_node_chunks, one launch each
            _lib.check(lib.is_chunk_partition(_lib.ptr(self._csr.rowptr_dst), self._num_nodes, int(k), int(CHUNK_SHARES == "auto" and int(k) == FULL_GRID_CHUNKS), _lib.ptr(dst),
                                              _lib.stream_ptr()), "is_chunk_partition")


def multi_copy(pairs):
    """All (source, destination) copies of one batch hand-over as ONE kernel launch
    (csrc/segment_ops.hip ``is_multi_copy``) instead of one hipMemcpyAsync per array."""
    import ctypes

    from . import _lib
    jobs = []
    for s_, d_ in pairs:
        if (s_.dtype != d_.dtype or s_.numel() != d_.numel() or not s_.is_contiguous() or not d_.is_contiguous()
                or s_.device != d_.device):
            raise ValueError("batch tensors must be contiguous, on the step's device, and match the captured "
                             f"dtype / shape (got {tuple(s_.shape)} {s_.dtype} for {tuple(d_.shape)} {d_.dtype})")
        nbytes = s_.numel() * s_.element_size()
        if nbytes % 4:
            raise ValueError("is_multi_copy moves 4-byte words")
        if nbytes:
            jobs.append(_lib.CopyJob(s_.data_ptr(), d_.data_ptr(), nbytes))
    lib = _lib.load()
    for at in range(0, len(jobs), 24):
        chunk = jobs[at:at + 24]
        arr = (_lib.CopyJob * len(chunk))(*chunk)
        _lib.check(lib.is_multi_copy(ctypes.cast(arr, ctypes.c_void_p), len(chunk), _lib.stream_ptr()), "is_multi_copy")


def _optimizer_tensors(optimizer):
    out = []
    for p, st in optimizer.state.items():
        for k, v in st.items():
            if torch.is_tensor(v):
                out.append((id(p), k, v))
    for gs in getattr(optimizer, "_groups", {}).values():      # immunostruct_amd.optim: device-side step counter
        out.append((id(gs), "state", gs["state"]))
    return out


def _snapshot(model, optimizer):
    return ({k: v.detach().clone() for k, v in model.state_dict().items()},
            {(a, k): v.detach().clone() for a, k, v in _optimizer_tensors(optimizer)})


def _restore(model, optimizer, snap):
    """in place (the captured graph and the optimizer's chunk table hold these addresses): values from before the
    warm-up; optimizer tensors that the warm-up created (fresh optimizer) go back to zero"""
    msnap, osnap = snap
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(msnap[k])
        for a, k, v in _optimizer_tensors(optimizer):
            if (a, k) in osnap:
                v.copy_(osnap[(a, k)])
            else:
                v.zero_()


class CapturedTrainStep:
    """``step(graph, seq, prop, y) -> loss`` replaying captured HIP graphs (``graph`` / ``seq`` / ``prop`` may be
    (cancer, wild-type) pairs: ``forward_loss`` then receives pairs of static buffers).

    ``forward_loss(model, graph, seq, prop, y) -> scalar loss`` defines the step body (so the same
    engine serves the regression / BCE / comparative stages).  Set ``forward_loss.fused_loss = True`` when the loss reads the
    reconstruction through ``utils.Losses`` only (every in-tree step does): the head then joins the sequence branch early.  Construction runs ``warmup`` (>= 1) REAL
    eager train steps on the template batch (they update the model like any other step), then captures.
    """

    def __init__(self, model, optimizer, reducer, forward_loss, template, edge_capacity, warmup=3, preserve_state=False,
                 step_random=None, random_stream=0):
        if warmup < 1:
            raise ValueError("warmup must be >= 1: optimizer state and BLAS handles have to be created by an eager "
                             "step BEFORE the capture (state created inside a capture is re-initialised on every replay)")
        g, seq, prop, y = template
        self.model, self.optimizer, self.reducer, self.forward_loss = model, optimizer, reducer, forward_loss
        # ``forward_loss.fused_loss = True``: the caller's promise that the loss reads the reconstruction through ``utils.Losses`` /
        # ``functional.vae_loss`` only -- the models may then join the sequence branch at the latent (functional.SpeculativeBackward).
        # A ``forward_loss`` without it (say, ``F.mse_loss`` on ``recon_x``) gets the full join: 1.7 % slower, never a missing edge
        self._fused_loss = bool(getattr(forward_loss, "fused_loss", False))
        # paired (cancer, wild-type) batches: graph / sequence / property are 2-tuples, one static buffer set per member
        self.paired = isinstance(g, (tuple, list))
        if self.paired:
            caps = edge_capacity if isinstance(edge_capacity, (tuple, list)) else (edge_capacity,) * len(g)
            self.sgraph = tuple(StaticGraphBatch(gi, ci) for gi, ci in zip(g, caps))
            self.seq = tuple(torch.zeros_like(t) for t in seq)
            self.prop = tuple(torch.zeros_like(t) for t in prop)
            g = g[0]
        else:
            self.sgraph = StaticGraphBatch(g, edge_capacity)
            self.seq, self.prop = torch.zeros_like(seq), torch.zeros_like(prop)
        self.y = torch.zeros_like(y)
        # step_random: where the step's dropout masks and reparameterisation noise come from.  None (this constructor's default, and
        # what every parity test uses: a ``forward_loss`` that patches the draws would be bypassed otherwise): the models draw them
        # inside the step with torch's generator (every replay of such a graph launches two generator-state fills in front of it).
        # "device" (the device-resident training loops of ``procedures`` and ``bench.py``): one launch of the library's own
        # generator inside the step (functional.StepRandom, is_step_random); ``random_stream`` is mixed into its key -- the index
        # of the run's stage --, ``random_state()`` / ``load_random_state()`` carry key and counter across a checkpoint.
        from .functional import StepRandom
        if step_random not in (None, "device"):
            raise ValueError("step_random must be None or 'device'")
        self._rand = StepRandom(self.y.device, stream_id=random_stream) if step_random else None
        self._load(*template)
        self.fused_optimizer = not reducer.packing   # single rank: optimizer inside the same graph
        if getattr(reducer, "_collective", False) and hasattr(optimizer, "grad_scale") and reducer.divide:
            # the summed gradient bucket is averaged inside the optimizer kernel instead of by a pass of its own
            reducer.divide = False
            optimizer.grad_scale = 1.0 / reducer.world
        # host->device uploads must not happen inside the capture: build the (cached) gradient scatter
        # maps of both layer shapes now, even when no eager warm-up step is requested
        from . import functional as HF
        fe = int(g.edata["edge_attr"].shape[1])
        for din in (20, HF.HIDDEN):
            HF.layer_plan(din, fe, g.device)
        # data-parallel runs: backward in two stages so that the all-reduce of the first gradient bucket (everything
        # above the EGNN stack: 96 % of the bytes) runs under the backward of the stack.  IMMUNOSTRUCT_DP_OVERLAP:
        # "auto" (default) captures both forms, times them on this machine / process group and keeps the faster one;
        # "1" / "0" force the two-stage / the serial form.
        mode = os.environ.get("IMMUNOSTRUCT_DP_OVERLAP", "auto")
        if mode not in ("auto", "0", "1"):
            raise ValueError("IMMUNOSTRUCT_DP_OVERLAP must be auto, 0 or 1")
        self.two_stage = False
        self._want_two_stage = reducer.packing and mode != "0"
        self._late = None
        self.dp_times = None
        snap = _snapshot(model, optimizer) if preserve_state else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body(eager=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if snap is not None:
            _restore(model, optimizer, snap)      # the warm-up steps leave no trace: training starts from the caller's state
        self.graph_a = self.graph_a1 = self.graph_a2 = self.graph_b = None
        self._forms = {}
        # graph A2 (the stack backward) is what runs BESIDE the collective: it is captured once per candidate number of CUs its
        # persistent layer kernels leave free for the collective's workgroups (functional.RESERVED_CUS;
        # IMMUNOSTRUCT_DP_RESERVED_CUS = comma-separated candidates, default "0,16" when collectives are issued, else "0");
        # "auto" times every candidate and keeps the fastest
        default_res = "0,16" if getattr(reducer, "_collective", False) else "0"
        self._reserved_candidates = sorted({max(0, int(v)) for v in os.environ.get("IMMUNOSTRUCT_DP_RESERVED_CUS", default_res).split(",") if v.strip() != ""}) or [0]
        if mode != "auto":
            # nothing is timed: only the first candidate would ever be replayed
            self._reserved_candidates = self._reserved_candidates[:1]
        self._a2 = {}
        self.reserved = self._reserved_candidates[0]
        if self.two_stage:
            self.graph_a1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a1, **_CAPTURE):
                loss = self._stage1()
                self.reducer.pack(0, from_grad=True)      # the bucket's pack as the graph's last node: no eager launch behind the replay
            # a graph always writes the gradient buffers it allocated while capturing: pack from those
            self.reducer.bind_sources(0, packed=True)
            self.reducer.reduce_bucket(0)         # .grad of the first bucket now aliases its persistent flat buffer
            bnd, bnd_grads = self._bnd, self._bnd_grads
            for res in self._reserved_candidates:
                self._bnd, self._bnd_grads = bnd, bnd_grads      # every capture walks the same autograd graph (retained below)
                graph = torch.cuda.CUDAGraph()
                saved = HF.RESERVED_CUS
                HF.RESERVED_CUS = res
                try:
                    with torch.cuda.graph(graph, pool=self.graph_a1.pool(), **_CAPTURE):
                        self._stage2(retain=res != self._reserved_candidates[-1])
                        self.reducer.pack(1, from_grad=True)
                finally:
                    HF.RESERVED_CUS = saved
                self.reducer.bind_sources(1, packed=True)
                self.reducer.reduce_bucket(1)
                self._a2[res] = (graph, self.reducer.sources())
            self.graph_a2, sources = self._a2[self.reserved]
            self._forms[True] = (loss, sources)
        if not self.two_stage or mode == "auto":
            self.graph_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a, **_CAPTURE):
                loss = self._fwd_bwd()
                if self.fused_optimizer:
                    self.optimizer.step()
                    from .functional import Stamps
                    Stamps.mark("optimizer done")
                else:
                    for i in range(len(self.reducer.buckets)):
                        self.reducer.pack(i, from_grad=True)
            if not self.fused_optimizer:
                self.reducer.bind_sources(packed=True)
                self.reducer.all_reduce_mean()    # .grad now aliases the persistent flat bucket(s)
            self._forms[False] = (loss, self.reducer.sources())
        if not self.fused_optimizer:
            if self._split_update():
                # one graph per gradient bucket: the first bucket's parameters (97 % of the bytes) are updated while the second,
                # latency-bound all-reduce is still on the wire
                self.graph_b = []
                for i, b in enumerate(self.reducer.buckets):
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, **({"pool": self.graph_b[0].pool()} if self.graph_b else {}), **_CAPTURE):
                        self.optimizer.step_subset(b["params"], first=i == 0)
                    self.graph_b.append(graph)
            else:
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, **_CAPTURE):
                    self.optimizer.step()
        self.one_graph = False
        self.graph_c = {}
        self.one_graph_error = None
        if not self.fused_optimizer and _ONE_GRAPH != "0" and self._collectives_capturable():
            self._capture_one_graph()
            # every rank replays the same form or the collectives stop matching: a form is a candidate only where EVERY rank
            # captured it (an eager MIN all-reduce of the two success flags; one rank's failure withdraws the form everywhere)
            import torch.distributed as dist
            have = torch.tensor([1.0 if k in self.graph_c else 0.0 for k in ("graph", "graph2")], device=self.y.device)
            dist.all_reduce(have, op=dist.ReduceOp.MIN)
            for k, ok in zip(("graph", "graph2"), have.tolist()):
                if ok < 0.5 and k in self.graph_c:
                    del self.graph_c[k]
                    self.one_graph_error = self.one_graph_error or f"form {k!r} was not captured on every rank"
        if self.graph_c and _ONE_GRAPH == "1":      # forced: no timing may hand the step back to a multi-graph form
            self._use_form("graph2" if "graph2" in self.graph_c and mode != "0" else "graph")
        elif len(self._forms) == 2 or self.graph_c:
            self._choose_form(model, optimizer)
        else:
            self._use_form(self.two_stage)
        self._drop_unused_forms()

    def _collectives_capturable(self):
        """the reducer issues collectives on a backend whose collectives are stream operations (nccl = RCCL); gloo's run on the host"""
        import torch.distributed as dist
        if not getattr(self.reducer, "_collective", False) or not dist.is_initialized():
            return False
        return dist.get_backend() == "nccl"

    def _capture_one_graph(self):
        """graph_c["graph"]: forward, backward, pack, all-reduce of every bucket, update -- the serial form as one graph;
        graph_c["graph2"] (when the model splits at the EGNN stack): stage 1, bucket 0 on the wire (asynchronous: a fork inside the
        graph), stack backward beside it, bucket 1, join, update(s) -- the two-stage form as one graph.  Each form is captured on its
        own: a failure (a stack that cannot capture its collectives) is recorded verbatim in ``one_graph_error`` and withdraws that
        form only; the multi-graph forms stay in charge of whatever is missing."""
        red = self.reducer
        saved = red.sources()
        split = self._split_update()      # (the update in the parts the eager warm-up steps ran it in: their chunk tables exist)

        def serial():
            loss = self._fwd_bwd()
            for i in range(len(red.buckets)):
                red.pack(i, from_grad=True)
                red.reduce_bucket(i, prepacked=True)
                if split:
                    self.optimizer.step_subset(red.buckets[i]["params"], first=i == 0)
            if not split:
                self.optimizer.step()
            return loss

        def two_stage():
            loss = self._stage1()
            red.pack(0, from_grad=True)
            work0 = red.reduce_bucket(0, async_op=True, prepacked=True)
            self._stage2()
            red.pack(1, from_grad=True)
            work1 = red.reduce_bucket(1, async_op=True, prepacked=True)
            if split:
                if work0 is not None:
                    work0.wait()
                self.optimizer.step_subset(red.buckets[0]["params"], first=True)
                if work1 is not None:
                    work1.wait()
                self.optimizer.step_subset(red.buckets[1]["params"], first=False)
            else:
                for w in (work0, work1):
                    if w is not None:
                        w.wait()
                self.optimizer.step()
            return loss

        bodies = [("graph", serial)]
        if self._forms.get(True) is not None and self._late is not None and len(red.buckets) == 2:
            bodies.append(("graph2", two_stage))
        self.graph_c = {}
        for name, body in bodies:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, **_CAPTURE):
                    loss = body()
                self.graph_c[name] = (graph, loss)
            except Exception as exc:      # noqa: BLE001 -- whatever the stack raises is the finding
                self.one_graph_error = (self.one_graph_error + "; " if self.one_graph_error else "") + f"{name}: {type(exc).__name__}: {exc}"
                torch.cuda.synchronize()
            finally:
                red.sources(saved)

    def random_state(self):
        """the state of the step's random-tensor provider (None without one): save it beside the optimizer state to resume a run"""
        return self._rand.state_dict() if self._rand is not None else None

    def load_random_state(self, sd):
        if self._rand is not None and sd is not None:
            self._rand.load_state_dict(sd)

    def _use_form(self, two_stage, reserved=None):
        if isinstance(two_stage, str):      # a one-graph form
            self.one_graph = two_stage
            self.loss = self.graph_c[two_stage][1]
            return
        self.one_graph = False
        self.two_stage = two_stage
        if two_stage and reserved is not None:
            self.reserved = reserved
            self.graph_a2, sources = self._a2[reserved]
            self._forms[True] = (self._forms[True][0], sources)
        self.loss, sources = self._forms[two_stage]
        self.reducer.sources(sources)

    def _choose_form(self, model, optimizer, steps=8):
        """time ``steps`` replays of the two-stage and of the serial form (max over ranks) and keep the faster one; the
        steps taken for this leave no trace in the model or the optimizer"""
        import time
        import torch.distributed as dist
        multi = dist.is_initialized() and dist.get_world_size() > 1
        snap = _snapshot(model, optimizer)
        cands = ([(True, r) for r in self._reserved_candidates] if True in self._forms else []) + \
                ([(False, None)] if False in self._forms else []) + [(k, None) for k in self.graph_c]
        times = []
        for form, res in cands:
            self._use_form(form, res)
            for k in range(2 + steps):
                if k == 2:
                    if multi:
                        dist.barrier()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                self.replay()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        t = torch.tensor(times, dtype=torch.float64, device=self.y.device)
        if multi:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)      # every rank takes the same decision
        t = [1e3 * v / steps for v in t.tolist()]
        two = [i for i, (f, _) in enumerate(cands) if f is True]
        ser = [i for i, (f, _) in enumerate(cands) if f is False]
        self.dp_times = {"two_stage_ms": min((t[i] for i in two), default=None), "serial_ms": t[ser[0]] if ser else None,
                         "two_stage_ms_by_reserved_cus": {str(cands[i][1]): t[i] for i in two},
                         "one_graph_ms": {cands[i][0]: t[i] for i in range(len(cands)) if isinstance(cands[i][0], str)}}
        _restore(model, optimizer, snap)
        best = min(range(len(cands)), key=lambda i: t[i])
        self._use_form(*cands[best])

    def _load(self, g, seq, prop, y):
        if self.paired:
            pairs = [(y, self.y)]
            for sg, gi, si, ss, pi, ps in zip(self.sgraph, g, seq, self.seq, prop, self.prop):
                pairs += sg.copy_pairs(gi) + [(si, ss), (pi, ps)]
            multi_copy(pairs)
        else:
            multi_copy(self.sgraph.copy_pairs(g) + [(seq, self.seq), (prop, self.prop), (y, self.y)])

    def _fwd_bwd(self):
        from . import functional as HF
        from .functional import SpeculativeBackward, Stamps, unit_gradient
        Stamps.mark("step start")
        self.reducer.zero()
        with SpeculativeBackward(self._fused_loss), HF.StepRandom.use(self._rand):      # the backward below is seeded with the unit gradient
            loss = self.forward_loss(self.model, self.sgraph, self.seq, self.prop, self.y)
        Stamps.mark("loss done")
        loss.backward(unit_gradient(loss.device))      # d loss / d loss = 1: recognised by the fused loss (no fill, no scaling)
        Stamps.mark("backward done (main stream)")
        return loss.detach()

    # ---- two-stage backward (data-parallel overlap) ---------------------------------
    def _classify(self, loss, bnd):
        """late = the parameters reachable from the loss ONLY through the EGNN stack outputs ``bnd`` (autograd graph walk)"""
        def leaves(roots, stop):
            seen, todo, out = set(), [r for r in roots if r is not None], set()
            while todo:
                fn = todo.pop()
                if fn in seen or fn in stop:
                    continue
                seen.add(fn)
                if hasattr(fn, "variable"):
                    out.add(id(fn.variable))
                todo.extend(f for f, _ in fn.next_functions if f is not None)
            return out
        cut = set(t.grad_fn for t in bnd if t.grad_fn is not None)
        below = leaves(list(cut), set())
        above = leaves([loss.grad_fn], cut)
        late = [p for p in self.reducer.params if id(p) in below and id(p) not in above]
        if late and len(late) < len(self.reducer.params):
            self._late = late
            ids = set(id(p) for p in late)
            self._early = [p for p in self.reducer.params if id(p) not in ids]
            self.reducer.split(late)
            self.two_stage = True
        else:
            self._want_two_stage = False

    def _stage1(self):
        """forward, loss, and the backward of everything ABOVE the EGNN stack(s): gradients of the stack outputs and
        of the parameters that do not feed the stack (sequence VAE, property MLP, attention values, heads)"""
        from . import functional as HF
        self.reducer.zero()
        HF.StackBoundary.begin()
        try:
            with HF.SpeculativeBackward(self._fused_loss), HF.StepRandom.use(self._rand):
                loss = self.forward_loss(self.model, self.sgraph, self.seq, self.prop, self.y)
        finally:
            bnd = HF.StackBoundary.end()
        if self._late is None:
            self._classify(loss, bnd)      # first eager step
            if not self.two_stage:
                loss.backward(HF.unit_gradient(loss.device))
                self._bnd = self._bnd_grads = None
                return loss.detach()
        outs = torch.autograd.grad(loss, bnd + self._early, grad_outputs=HF.unit_gradient(loss.device), allow_unused=True)
        pairs = [(t, g) for t, g in zip(bnd, outs[:len(bnd)]) if g is not None]
        self._bnd, self._bnd_grads = [t for t, _ in pairs], [g for _, g in pairs]
        for p, g in zip(self._early, outs[len(bnd):]):
            p.grad = g
        return loss.detach()

    def _stage2(self, retain=False):
        """backward of the EGNN stack(s) from the gradients stage 1 left at their outputs"""
        if self._bnd:
            outs = torch.autograd.grad(self._bnd, self._late, grad_outputs=self._bnd_grads, allow_unused=True, retain_graph=retain)
            for p, g in zip(self._late, outs):
                p.grad = g
        self._bnd = self._bnd_grads = None

    def _split_update(self):
        """data-parallel forms with two gradient buckets: update bucket by bucket, each as soon as its all-reduce is done"""
        return not self.fused_optimizer and len(self.reducer.buckets) == 2 and hasattr(self.optimizer, "step_subset")

    def _reduce_and_update(self, stage2, update):
        """the part of a data-parallel step behind the (first-stage) backward: all-reduce bucket by bucket, ``stage2()`` (the stack
        backward of the two-stage form, or None) under the first bucket's all-reduce, ``update(i)`` = the optimizer step of bucket
        i's parameters (i = None: of all parameters)"""
        red = self.reducer
        if not self._split_update():
            if stage2 is not None:
                work = red.reduce_bucket(0, async_op=True)
                stage2()
                red.reduce_bucket(1)
                if work is not None:
                    work.wait()
            else:
                red.all_reduce_mean()
            update(None)
            return
        work0 = red.reduce_bucket(0, async_op=True)
        if stage2 is not None:
            stage2()
        work1 = red.reduce_bucket(1, async_op=True)      # queued behind bucket 0 on the collective's stream
        if work0 is not None:
            work0.wait()
        update(0)
        if work1 is not None:
            work1.wait()
        update(1)

    def _eager_update(self, i):
        if i is None:
            self.optimizer.step()
        else:
            self.optimizer.step_subset(self.reducer.buckets[i]["params"], first=i == 0)

    def _body(self, eager=False):
        if self._want_two_stage:
            loss = self._stage1()
            self._reduce_and_update(self._stage2 if self.two_stage else None, self._eager_update)
            return loss
        loss = self._fwd_bwd()
        if self.fused_optimizer:
            self.reducer.all_reduce_mean()
            self.optimizer.step()
        else:
            self._reduce_and_update(None, self._eager_update)
        return loss

    def __call__(self, g, seq, prop, y):
        self._load(g, seq, prop, y)
        return self.replay()

    def replay(self):
        """Run the captured step on whatever the static buffers (``sgraph``, ``seq``, ``prop``, ``y``) hold -- the
        on-device batcher (``data.DeviceResidentDataset.gather_into``) writes them directly."""
        if hasattr(self.optimizer, "refresh"):
            self.optimizer.refresh()      # learning-rate schedulers: host value -> device copy read by the captured step
        def update(i):
            (self.graph_b if i is None else self.graph_b[i]).replay()
        if self.one_graph:
            self.graph_c[self.one_graph][0].replay()      # forward ... collectives ... update: one graph
        elif self.two_stage:
            self.graph_a1.replay()
            # bucket 0 is in flight while graph A2 runs the stack backward
            self._reduce_and_update(self.graph_a2.replay, update)
        else:
            self.graph_a.replay()
            if self.graph_b is not None:
                self._reduce_and_update(None, update)
        return self.loss

    def _drop_unused_forms(self):
        """the captured forms that lost the choice are released: they hold pool memory and -- the one-graph forms -- RCCL's nodes"""
        if self.one_graph:
            self.graph_c = {self.one_graph: self.graph_c[self.one_graph]}
            self.graph_a = self.graph_a1 = self.graph_a2 = self.graph_b = None
            self._a2, self._forms = {}, {}
        else:
            self.graph_c = {}
            if self.two_stage:
                self.graph_a = None
                self._a2 = {self.reserved: self._a2[self.reserved]}
            else:
                self.graph_a1 = self.graph_a2 = None
                self._a2 = {}
            self._forms = {k: v for k, v in self._forms.items() if k == self.two_stage}      # (their gradient sources go with them)

    def close(self):
        """Release every captured graph (after a synchronize).  Call BEFORE ``torch.distributed.destroy_process_group()``: a graph
        that holds RCCL's collective nodes must not outlive its communicator (an abort at process exit was seen when it did)."""
        torch.cuda.synchronize()
        self.graph_c = {}
        self._a2 = {}
        self.graph_a = self.graph_a1 = self.graph_a2 = self.graph_b = None
        torch.cuda.synchronize()
