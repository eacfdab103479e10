"""``torch.autograd.Function`` wrappers over the C-ABI kernels.

Each Function validates shapes/dtypes in Python (raising ``ValueError`` the
way torch would), allocates outputs with torch (device memory + stream are the
only things torch provides here) and enqueues the HIP kernels on the current
stream.  There is no fallback path: CPU tensors raise ``HipExtensionError``.
"""
from __future__ import annotations

import torch

from . import _lib

import ctypes
import os

HIDDEN = 64
_MAX_BWD_GRID = 256  # one persistent workgroup per CU (MI355X: 256 CUs)
SAVE_Z3 = os.environ.get("IMMUNOSTRUCT_SAVE_Z3", "1") == "1"      # 0: the backward recomputes z3 per tile instead of reading it back (measured: -37 MB per layer pair, +6 us per backward launch)
# workgroups per layer of the batched node weight-gradient launch, by kind (node block: 12 MFMAs per 4-row step; next
# pre-projection: 8).  13 x 56 = 728 <= 3 x 256 resident.  Equal work per workgroup (72 : 48) is SLOWER (82 -> 92 us): the launch
# lasts as long as its longest workgroup, ~2.9 us per 16-row chunk whatever the kind (HISTORY.md, round 3)
WGRAD_GRID_NODE = 56
WGRAD_GRID_PROJ = 56
FWD_CHUNKS_MAX = 2048
FWD_CHUNK_EDGES = 32
# Data-parallel mode of the two persistent layer kernels: their grids normally fill EVERY workgroup slot of the chip (2 per CU),
# and a static partition means that a slot taken by a co-running kernel -- under data parallelism RCCL's all-reduce, which cannot
# be scheduled away -- costs the launch a whole extra round.  With RESERVED_CUS = R the grids stop at 2 * (256 - R) workgroups
# (forward: 8 * (256 - R) wave chunks), so that a collective of up to 2 R workgroup slots (NCCL_MAX_NCHANNELS bounds its footprint)
# fits BESIDE a layer kernel.  Set by engine.CapturedTrainStep for the overlapped (two-stage) form; 0 = single-GPU behaviour.
RESERVED_CUS = 0


def layer_slots():
    """workgroup slots the persistent layer kernels may fill"""
    return 2 * (_MAX_BWD_GRID - max(0, min(RESERVED_CUS, _MAX_BWD_GRID - 1)))


# 1 (default): the backward layer launches run as ONE 512-thread workgroup per CU wherever csrc/egnn_layer_bwd8.hip covers the
# shape and use_paired_bwd's measured rule picks it; 0: always two 256-thread workgroups per CU (csrc/egnn_layer_bwd.hip)
BWD_PAIRED = os.environ.get("IMMUNOSTRUCT_BWD_PAIRED", "1") == "1"
# The paired kernel halves the partial records (reduce_partials_batched - 13 us per step) and lets no co-running kernel take a slot
# beside it (the step's first backward launches are not stretched by the sequence branch's backward); its lockstep costs 4 - 8 % per
# launch.  The fixed gains win while the launches are short; measured on the replayed step (bench.py --batch B, E / N = 3, ms per step,
# paired / 256-thread, two interleaved runs each, round 6): B = 128: 1.020 / 1.028, 160: 1.218 / 1.242, 200: 1.514 / 1.541,
# 224: 1.614 / 1.635 -- and B = 256: 1.812 / 1.788, 320: 2.221 / 2.166, 384: 2.599 / 2.554, 512: 3.400 / 3.315; config 4 (128 pairs = 256
# graphs): 1.915 / 1.892.  The cut sits between 2660 and 3040 node tiles of 16 (5.2 and 5.9 tiles per workgroup slot).
PAIRED_BWD_MAX_TILES = 2816      # 5.5 tiles per slot of the full grid


def use_paired_bwd(num_nodes, fe):
    """whether the backward layer launches of a batch of ``num_nodes`` nodes run as the paired 512-thread kernel"""
    if not (BWD_PAIRED and SAVE_Z3 and bool(_lib.load().is_egnn_layer_bwd_paired_supported(fe))):
        return False
    return (num_nodes + 15) // 16 <= PAIRED_BWD_MAX_TILES * layer_slots() // (2 * _MAX_BWD_GRID)


FWD_NODES_PER_WG = 56      # nodes a forward workgroup should own at most on average: one 64-row pass of its node half, with a margin


def prepare_layer_partitions(csr, fe, forward=True):
    """build (and cache in ``csr``) the work partition the forward layer kernel will ask for -- to be called outside a stream capture:
    a partition built while capturing lives in the capture's memory pool.  (The backward cuts the nodes into plain 16-node tiles
    inside the kernel: nothing to prepare; ``forward=False`` is a no-op kept for the data-parallel engine's call.)"""
    if forward:
        csr.chunks(fwd_chunk_count(csr.num_edges, csr.num_nodes))


def fwd_chunk_count(num_edges, num_nodes=0):
    """number of wave-chunks of the forward layer kernel: ~FWD_CHUNK_EDGES edges per wave and at most FWD_NODES_PER_WG nodes per
    workgroup of 4 waves (its node half runs 64 rows per pass: at E / N = 2, 32 edges per wave are 64.5 nodes per workgroup and
    half of the workgroups ran a second pass -- round 3's sweep: 51 us at 48 k edges against 46 us at 72 k), at most
    FWD_CHUNKS_MAX waves, a multiple of the 4 waves of a workgroup.  A count between one and two workgroups per CU is rounded up
    to the full grid: with 377 workgroups on 256 CUs the launch lasts as long as the CUs that hold two."""
    full = min(FWD_CHUNKS_MAX, 4 * layer_slots())
    k = max((num_edges + FWD_CHUNK_EDGES - 1) // FWD_CHUNK_EDGES, (4 * num_nodes + FWD_NODES_PER_WG - 1) // FWD_NODES_PER_WG)
    if 2 * k > full:
        k = full
    k = max(4, min(full, k))
    return (k + 3) // 4 * 4


class StackBoundary:
    """Collects the outputs of the EGNN stack(s) of one forward pass.  The data-parallel engine cuts the backward
    there: everything above the stack first (its gradients go out on the all-reduce), the stack itself second."""
    active = None

    @classmethod
    def begin(cls):
        cls.active = []

    @classmethod
    def record(cls, *tensors):
        if cls.active is not None:
            cls.active.extend(t for t in tensors if t is not None and t.requires_grad)

    @classmethod
    def end(cls):
        out, cls.active = cls.active or [], None
        return out


COMPOSED_PATHS = {}      # reason -> how often a shape OUTSIDE the HIP kernels' build ran as a device-side torch composition


def composed_path(reason):
    """Note that an operator ran from device-side torch ops because its shape lies outside what the HIP kernels are built for
    (more than 256 nodes per graph or other head counts than 1 / 8 in the node attention, an EGNN width other than 64, more than
    256 contrastive pairs): same arithmetic, same oracle tests, but NOT the measured path.  The first occurrence of every reason
    warns; ``COMPOSED_PATHS`` keeps the counts (``bench.py`` prints them in ``config.composed_paths``: {} for every workload it
    times), so a caller who leaves the envelope sees it instead of a silently slower run."""
    if reason not in COMPOSED_PATHS:
        import warnings
        warnings.warn(f"immunostruct_amd: {reason} -- outside the HIP kernels' build, running as a device-side torch composition "
                      "(correct, tested against the oracle, much slower; functional.COMPOSED_PATHS counts the calls)", RuntimeWarning,
                      stacklevel=3)
    COMPOSED_PATHS[reason] = COMPOSED_PATHS.get(reason, 0) + 1


class Stamps:
    """Debug aid (IMMUNOSTRUCT_STAMPS=1): device wall-clock stamps at named points of a step, also inside a captured
    HIP graph -- the only way to see the schedule of a replayed graph without a profiler's perturbation."""
    enabled = os.environ.get("IMMUNOSTRUCT_STAMPS", "0") == "1"
    names = []
    buf = None

    @classmethod
    def mark(cls, name):
        if not cls.enabled:
            return
        if cls.buf is None:
            cls.buf = torch.zeros(256, dtype=torch.int64, device="cuda")
        if name not in cls.names:
            cls.names.append(name)
        slot = cls.buf[cls.names.index(name):]
        _lib.check(_lib.load().is_debug_timestamp(_lib.ptr(slot), _lib.stream_ptr()), "is_debug_timestamp")

    @classmethod
    def hook(cls, tensor, name):
        """stamp when the gradient of ``tensor`` is produced in backward"""
        if cls.enabled and tensor.requires_grad:
            def _h(g):
                cls.mark(name)
                return g
            tensor.register_hook(_h)
        return tensor

    @classmethod
    def report(cls):
        vals = cls.buf[:len(cls.names)].tolist()
        t0 = min(v for v in vals if v > 0)
        return sorted(((v - t0) / 100.0, n) for v, n in zip(vals, cls.names))     # microseconds (100 MHz clock)


class LaunchClock:
    """In-situ duration of the layer-kernel launches (bench.py's roofline): when enabled, every launch of ``is_egnn_layer_fwd`` /
    ``is_egnn_layer_bwd`` / ``is_gather_segment_sum`` gets a [grid, 2] int64 buffer in which its workgroups store the device wall
    clock at their start and end (``csrc/common.h`` ``wg_clock_start`` / ``_end``); a launch lasted from the smallest start to the
    largest end.  Unlike HIP events this works inside a replayed HIP graph -- the buffers are allocated on the first (eager)
    step, keyed by launch site, and the captured launches keep writing into them -- so the numbers are those of the step whose
    throughput is reported, co-running branches included.  Enable BEFORE the first step of a model."""
    enabled = False
    sites = {}      # (kind, layer index, grid) -> int64 tensor [grid, 2]
    TICKS_PER_US = 100.0      # wall_clock64: 100 MHz

    @classmethod
    def slot(cls, kind, index, grid, device):
        if not cls.enabled:
            return None
        key = (kind, index, grid)
        buf = cls.sites.get(key)
        if buf is None or buf.device != device:
            if torch.cuda.is_current_stream_capturing():
                return None       # never allocate inside a capture: such a launch stays untimed
            buf = torch.zeros(grid, 2, dtype=torch.int64, device=device)
            cls.sites[key] = buf
        return buf

    @classmethod
    def read(cls):
        """{(kind, index, grid): (first workgroup start, last workgroup end) in microseconds of the device clock} for the most
        recent launch at every site -- call after a synchronize"""
        out = {}
        for key, buf in cls.sites.items():
            t = buf.cpu()
            ok = (t[:, 0] > 0) & (t[:, 1] >= t[:, 0])
            if bool(ok.any()):
                out[key] = (float(t[ok, 0].min()) / cls.TICKS_PER_US, float(t[ok, 1].max()) / cls.TICKS_PER_US)
        return out

    @classmethod
    def durations(cls):
        """Per launch kind, the in-situ durations (microseconds) of the most recent step's launches, two ways:
        ``span`` = first workgroup start -> last workgroup end of the launch itself;
        ``slot`` = last workgroup end of the PREVIOUS layer launch on the same chain -> last workgroup end of this one, i.e. the
        launch's share of the step's critical chain, dispatch gap and ramp included (what a kernel trace reports as its
        duration when launches are back to back) -- defined for launches whose predecessor is clocked too (forward layers
        1.., backward layers L-2..0, the layer-0 gather)."""
        raw = cls.read()
        fwd = sorted((k for k in raw if k[0].startswith("fwd")), key=lambda k: k[1])
        bwd = sorted((k for k in raw if k[0].startswith("bwd")), key=lambda k: -k[1]) + [k for k in raw if k[0] == "gather"]
        out = {}
        for chain in (fwd, bwd):
            prev = None
            for k in chain:
                ent = out.setdefault(k[0], {"span": [], "slot": []})
                ent["span"].append(raw[k][1] - raw[k][0])
                if prev is not None and raw[k][1] > raw[prev][1]:
                    ent["slot"].append(raw[k][1] - raw[prev][1])
                prev = k
        return out

    @classmethod
    def reset(cls):
        cls.sites = {}


class KernelTimer:
    """Optional HIP-event bracketing of individual kernel launches (used by bench.py for the roofline).

    Events are recorded on the stream the kernel is launched on (torch's current stream).
    """
    enabled = False
    records = {}
    # names of launches that :meth:`launch` issues ``repeat`` times back to back inside ONE event pair (idempotent kernels
    # only): on an idle GPU an event pair also brackets the host's launch latency (~10 us: a 82 us kernel reads 93 us); with
    # K launches in the pair that gap is paid once, and the GPU runs the kernel at the clocks it has inside a replayed step
    repeat = 1
    repeat_names = frozenset()

    class _Span:
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            if KernelTimer.enabled:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *exc):
            if KernelTimer.enabled:
                self.e1.record()
                KernelTimer.records.setdefault(self.name, []).append((self.e0, self.e1))
            return False

    @classmethod
    def span(cls, name):
        return cls._Span(name)

    @classmethod
    def launch(cls, name, fn):
        """run ``fn`` (one kernel launch) under the timer ``name``; see ``repeat``"""
        k = cls.repeat if (cls.enabled and name in cls.repeat_names) else 1
        if k <= 1:
            with cls.span(name):
                fn()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        cls.records.setdefault(name, []).append((e0, e1, k))

    @classmethod
    def reset(cls):
        cls.records = {}

    @classmethod
    def summary(cls):
        """name -> (launches kept, mean milliseconds); call after torch.cuda.synchronize()."""
        out = {}
        for name, evs in cls.records.items():
            ms = sorted(ev[0].elapsed_time(ev[1]) / (ev[2] if len(ev) > 2 else 1) for ev in evs)
            # eager launches: an event pair also brackets the HOST time between the two records when the GPU is idle, so a
            # garbage-collection pause or an allocator hipMalloc shows up as a 10-100 ms "launch": drop such stalls
            # (> 5x the median) before averaging
            med = ms[len(ms) // 2] if ms else 0.0
            kept = [t for t in ms if t <= 5.0 * med] or ms
            out[name] = (len(kept), sum(kept) / max(len(kept), 1))
        return out


# ---------------------------------------------------------------------------
# EGNN stack
# ---------------------------------------------------------------------------
PARAMS_PER_LAYER = 11
# order of the per-layer parameter tensors handed to EGNNStackFn (the reference's native tensors):
#   edge_mlp.0.weight [64, 2*Din+1+Fe], edge_mlp.0.bias, edge_mlp.2.weight, edge_mlp.2.bias,
#   node_mlp.0.weight [64, Din+64], node_mlp.0.bias, node_mlp.2.weight, node_mlp.2.bias,
#   coord_mlp.0.weight, coord_mlp.0.bias, coord_mlp.2.weight [1, 64]
_EDGE_STRIDE = 8448 + 64 * 8
_NODE_STRIDE = 64 * 128 + 64 * 64 + 128
_PROJ_STRIDE = 128 * 64 + 128
_plan_cache = {}
class _LayerPlan:
    """Offsets of one layer's native gradient tensors in a flat buffer + the scatter maps that send the
    kernels' partial-record layouts there (built once per (Din, Fe, device))."""

    def __init__(self, din, fe, device):
        import numpy as np
        h = HIDDEN
        self.din, self.fe = din, fe
        self.ldw = 2 * din + 1 + fe
        shapes = [(h, self.ldw), (h,), (h, h), (h,), (h, din + h), (h,), (h, h), (h,), (h, h), (h,), (1, h)]
        self.shapes = shapes
        offs, total = [], 0
        for sh in shapes:
            offs.append(total)
            total += int(np.prod(sh))
        self.offsets, self.total = offs, total
        o_w1, o_b1, o_w2, o_b2, o_wn1, o_bn1, o_wn2, o_bn2, o_wc1, o_bc1, o_wc2 = offs
        c = np.arange(h)
        # edge record: dW2 | dWc1 | db2 | dbc1 | dwc2 | dw_r | dW_a[c][8]
        edge = np.full(_EDGE_STRIDE, -1, dtype=np.int32)
        edge[0:4096] = o_w2 + np.arange(4096)
        edge[4096:8192] = o_wc1 + np.arange(4096)
        edge[8192:8256] = o_b2 + c
        edge[8256:8320] = o_bc1 + c
        edge[8320:8384] = o_wc2 + c
        edge[8384:8448] = o_w1 + c * self.ldw + 2 * din
        for f in range(fe):
            edge[8448 + c * 8 + f] = o_w1 + c * self.ldw + 2 * din + 1 + f
        # node record: dWn1 [64][128] (h part padded to 64 | h_neigh part) | dWn2 | dbn1 | dbn2
        node = np.full(_NODE_STRIDE, -1, dtype=np.int32)
        oo, kk = np.meshgrid(c, np.arange(128), indexing="ij")
        dst = np.where(kk < 64, np.where(kk < din, o_wn1 + oo * (din + h) + kk, -1), o_wn1 + oo * (din + h) + din + (kk - 64))
        node[0:8192] = dst.reshape(-1)
        node[8192:12288] = o_wn2 + np.arange(4096)
        node[12288:12352] = o_bn1 + c
        node[12352:12416] = o_bn2 + c
        # proj record: dW1sd [128][64] | db1
        proj = np.full(_PROJ_STRIDE, -1, dtype=np.int32)
        cc, kk = np.meshgrid(np.arange(128), np.arange(64), indexing="ij")
        dst = np.where(kk < din, np.where(cc < 64, o_w1 + cc * self.ldw + kk, o_w1 + (cc - 64) * self.ldw + din + kk), -1)
        proj[0:8192] = dst.reshape(-1)
        proj[8192:8256] = o_b1 + c
        covered = np.zeros(total, dtype=bool)
        for m in (edge, node, proj):
            covered[m[m >= 0]] = True
        assert covered.all(), "gradient scatter maps do not cover the layer's parameters"
        self.edge_map = torch.from_numpy(edge).to(device)
        self.node_map = torch.from_numpy(node).to(device)
        self.proj_map = torch.from_numpy(proj).to(device)

    def views(self, flat):
        return [flat[o:o + int(torch.Size(sh).numel())].view(sh) for o, sh in zip(self.offsets, self.shapes)]


def layer_plan(din, fe, device):
    key = (din, fe, str(device))
    if key not in _plan_cache:
        _plan_cache[key] = _LayerPlan(din, fe, device)
    return _plan_cache[key]


def _grid_for(n_rows, rows_per_wg):
    return max(1, min(_MAX_BWD_GRID, (n_rows + rows_per_wg - 1) // rows_per_wg))


class StackPrologue:
    """Outputs of the stack's prologue launch (``is_stack_prologue``): layer-0 pre-projection ``psd`` [N,128], the operand
    packs of every layer's node half, and the dense coordinates ``x`` [N,3].  A caller may launch it EARLY
    (:func:`launch_stack_prologue`) -- the models do so before they fork the sequence branch onto its side stream, so that
    the graph branch's first kernel is not the one that pays the fork -- and hand the object to :func:`egnn_stack`.

    :func:`launch_stack_forward` goes further and launches the layer kernels too (``layers`` / ``outs`` then hold what
    ``EGNNStackFn.forward`` would have produced): the autograd node is still created by the later :func:`egnn_stack` call, so
    the ORDER IN WHICH AUTOGRAD NODES ARE CREATED (which fixes the order of the backward) is independent of the order in which
    the forward kernels are enqueued (which fixes how a captured HIP graph is laid out over the hardware queues)."""

    __slots__ = ("psd", "packs", "x", "key", "layers", "outs", "need_grad", "final_coords", "ea_key", "fork_event")

    def __init__(self, psd, packs, x, key):
        self.psd, self.packs, self.x, self.key = psd, packs, x, key
        self.layers = self.outs = self.ea_key = self.fork_event = None
        self.need_grad = self.final_coords = False


def _prologue_key(h0, x0, params):
    return (h0.data_ptr(), tuple(h0.shape), x0.data_ptr(), tuple(x0.stride()), tuple(p.data_ptr() for p in params))


def _launch_prologue(h0, ld_h0, din0, x0, params, head, n_layers, n, dev):
    """params: the flat list of contiguous fp32 layer parameters (11 per layer); head: ((Wa, Wb), ba, bb) or None"""
    lib = _lib.load()
    P = PARAMS_PER_LAYER
    f32 = dict(dtype=torch.float32, device=dev)
    # coordinates: the kernels read a dense [N,3] array; the reference keeps them as the last columns of ndata['x'] -- the
    # prologue launch writes the dense copy (no copy launch in front of the stack)
    if x0.dtype != torch.float32:
        raise ValueError(f"expected float32 coordinates, got {x0.dtype}")
    if x0.is_contiguous():
        x, x_src, ld_x0 = x0, None, 0
    else:
        x_src, ld_x0 = _lib.rows_ld(x0)
        x = torch.empty(n, 3, **f32)
    w1_0, b1_0 = params[0], params[1]
    psd = torch.empty(n, 2 * HIDDEN, **f32)
    # layer-0 pre-projection and the lane-ordered operand packs of every layer's node half (forward + backward
    # order): independent, ONE launch
    packs = torch.empty(n_layers, 2, lib.is_node_pack_floats(), **f32)
    jobs = []
    for i in range(n_layers):
        lp = params[i * P:(i + 1) * P]
        if i < n_layers - 1:
            w1n = params[(i + 1) * P]          # next layer's edge_mlp.0.weight: [W1s | W1d | w_r | W_a] column blocks
            wa_p, wb_p, ldn = w1n.data_ptr(), w1n.data_ptr() + 4 * HIDDEN, int(w1n.shape[1])
        elif head is not None:
            wa_p, wb_p, ldn = head[0][0].data_ptr(), head[0][1].data_ptr(), HIDDEN      # [Wq | Wk] where they are
        else:
            wa_p, wb_p, ldn = None, None, 0
        jobs.append(_lib.NodePackJob(lp[4].data_ptr(), lp[6].data_ptr(), wa_p, wb_p,
                                     packs[i, 0].data_ptr(), packs[i, 1].data_ptr(), din0 if i == 0 else HIDDEN, ldn))
    jarr = (_lib.NodePackJob * len(jobs))(*jobs)
    with KernelTimer.span("stack_prologue"):
        _lib.check(lib.is_stack_prologue(ctypes.cast(jarr, ctypes.c_void_p), len(jobs), _lib.ptr(h0), ld_h0, din0,
                                         _lib.ptr(w1_0), int(w1_0.shape[1]), None, _lib.ptr(b1_0), _lib.ptr(psd),
                                         _lib.ptr(x_src), ld_x0, _lib.ptr(x) if x_src is not None else None, n, _lib.stream_ptr()),
                   "is_stack_prologue")
    return psd, packs, x


def launch_stack_prologue(h0, x0, layer_params, head=None):
    """Launch the prologue of :func:`egnn_stack` now, on the current stream; returns the :class:`StackPrologue` to pass on.
    The arguments must be the very tensors the later ``egnn_stack`` call receives (checked there; a mismatch relaunches)."""
    flat = [p for lp in layer_params for p in lp] + (list(head) if head is not None else [])
    _lib.require_device(h0, x0, *flat)
    din0, n = int(h0.shape[1]), int(h0.shape[0])
    if din0 not in (20, HIDDEN) or x0.shape != (n, 3) or not 1 <= len(layer_params) <= 7:
        raise ValueError("launch_stack_prologue: node features must be 20 or 64 wide, coordinates (N, 3), 1-7 layers")
    key = _prologue_key(h0, x0, flat)
    h0c, ld_h0 = _lib.rows_ld(h0)
    params = [_lib.f32c(p) for p in flat]
    hd = None
    if head is not None:
        wa, ba, wb, bb = params[-4:]
        params = params[:-4]
        hd = ((wa, wb), ba, bb)
    with torch.no_grad():
        psd, packs, x = _launch_prologue(h0c, ld_h0, din0, x0, params, hd, len(layer_params), n, h0.device)
    return StackPrologue(psd, packs, x, key)


def launch_stack_forward(h0, x0, ea_csr, csr, layer_params, head=None, final_coords=True, fork_after=None):
    """Enqueue the WHOLE forward of :func:`egnn_stack` now (prologue + one launch per layer), outside autograd; the returned
    :class:`StackPrologue` makes the later ``egnn_stack(..., prologue=...)`` call -- same tensors -- create the autograd node
    without launching anything.  ``fork_after`` = i: an event recorded right behind layer i's launch is left in
    ``.fork_event`` (for a caller that starts another branch on a side stream from that point of the stack)."""
    pro = launch_stack_prologue(h0, x0, layer_params, head=head)
    flat = [p for lp in layer_params for p in lp]
    n_layers = len(layer_params)
    fe = int(ea_csr.shape[1]) if ea_csr is not None else 0
    if fe > 8:
        raise ValueError("edge_feat_size > 8 is not supported by the HIP kernel")
    if h0.shape[0] != csr.num_nodes:
        raise ValueError("node feature rows must equal the number of nodes")
    need_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                for t in [h0, x0] + flat + (list(head) if head is not None else []))
    h0c, ld_h0 = _lib.rows_ld(h0)
    ea = _lib.f32c(ea_csr) if fe else None
    params = [_lib.f32c(p) for p in flat]
    hd = None
    if head is not None:
        wa, ba, wb, bb = (_lib.f32c(t) for t in head)
        hd = ((wa, wb), ba, bb)

    def mark():
        pro.fork_event = torch.cuda.Event()
        pro.fork_event.record()

    with torch.no_grad():
        layers, h_out, x_out, psd = _launch_stack_layers(
            h0c, ld_h0, int(h0.shape[1]), ea, fe, csr, params, hd, n_layers, final_coords, need_grad, pro.psd, pro.packs, pro.x,
            after=(max(0, min(int(fork_after), n_layers - 1)), mark) if fork_after is not None else None)
    pro.layers, pro.outs = layers, (h_out, x_out, psd)
    pro.need_grad, pro.final_coords, pro.ea_key = need_grad, bool(final_coords), _ea_key(ea, csr)
    return pro


def _ea_key(ea, csr):
    return (ea.data_ptr() if ea is not None else 0, id(csr))


def _launch_stack_layers(h0, ld_h0, din0, ea, fe, csr, params, head, n_layers, final_coords, need_grad, psd, packs, x, after=None):
    """the forward layer launches of the stack (one per layer); ``after`` = (i, fn): call fn() right behind layer i's launch.
    -> (per-layer saved tensors, h_L, x_L | None, head projection | None)"""
    lib = _lib.load()
    P = PARAMS_PER_LAYER
    n, e = csr.num_nodes, csr.num_edges
    st = _lib.stream_ptr()
    f32 = dict(dtype=torch.float32, device=x.device)
    layers = []
    kf = fwd_chunk_count(e, n)
    chunks = csr.chunks(kf)
    h_in, ld_h, din = h0, ld_h0, din0
    for i in range(n_layers):
        W1, b1, W2, b2, Wn1, bn1, Wn2, bn2, Wc1, bc1, wc2 = params[i * P:(i + 1) * P]
        ldw = int(W1.shape[1])
        if ldw != 2 * din + 1 + fe:
            raise ValueError(f"layer {i}: edge_mlp.0.weight has {ldw} columns, expected {2 * din + 1 + fe}")
        last = i == n_layers - 1
        no_coords = (not final_coords) and last
        h_neigh = torch.empty(n, HIDDEN, **f32)
        x_out = torch.empty(n, 3, **f32) if not no_coords else None
        z2s = torch.empty(max(e, 16), HIDDEN, **f32) if need_grad else None     # full 16-row tiles are stored
        # IMMUNOSTRUCT_SAVE_Z3=0: the coordinate MLP's pre-activation is not saved -- the backward layer kernel recomputes it (one
        # more product per tile): one [E, 64] store per layer forward and one load per layer backward less, but the backward
        # window is issue-bound and the 64 extra MFMAs cost more than the loads they replace (HISTORY.md), so saving stays default
        z3s = torch.empty(max(e, 16), HIDDEN, **f32) if (need_grad and not no_coords and SAVE_Z3) else None
        # the first edge-MLP activation and its derivative, when the library's backward reads them back (is_layer_saves_m1)
        save_m1 = int(lib.is_layer_saves_m1()) if need_grad else 0      # 1: SiLU(z1) + SiLU'(z1); 2: z1 only
        m1s = torch.empty(max(e, 16), HIDDEN, **f32) if save_m1 else None
        dy1s = torch.empty(max(e, 16), HIDDEN, **f32) if save_m1 == 1 else None
        geos = torch.empty(max(e, 16), 4, **f32) if (need_grad and lib.is_layer_saves_geo()) else None
        zn1 = torch.empty(n, HIDDEN, **f32) if need_grad else None
        h_out = torch.empty(n, HIDDEN, **f32)
        emit = (not last) or head is not None
        psd_next = torch.empty(n, 2 * HIDDEN, **f32) if emit else None
        if last:
            b0n, b1n = (head[1], head[2]) if head is not None else (None, None)
        else:
            b0n, b1n = None, params[(i + 1) * P + 1]
        clk = LaunchClock.slot("fwd_nocoord" if no_coords else "fwd", i, kf // 4, x.device)
        KernelTimer.launch("egnn_layer_fwd_nocoord" if no_coords else "egnn_layer_fwd", lambda: _lib.check(lib.is_egnn_layer_fwd(
            _lib.ptr(psd), _lib.ptr(psd[:, HIDDEN:]), 2 * HIDDEN, _lib.ptr(x), _lib.ptr(ea),
            _lib.ptr(csr.rowptr_dst), _lib.ptr(csr.src_sorted), _lib.ptr(csr.dst_sorted), _lib.ptr(chunks), kf,
            _lib.ptr(W1), ldw, din, _lib.ptr(W2), _lib.ptr(b2), _lib.ptr(Wc1), _lib.ptr(bc1), _lib.ptr(wc2),
            _lib.ptr(h_neigh), HIDDEN, _lib.ptr(x_out), _lib.ptr(z2s), _lib.ptr(z3s), n, e, fe,
            _lib.ptr(h_in), ld_h, _lib.ptr(bn1), _lib.ptr(bn2), _lib.ptr(b0n), _lib.ptr(b1n), _lib.ptr(packs[i, 0]),
            _lib.ptr(zn1), _lib.ptr(h_out), _lib.ptr(psd_next), _lib.ptr(clk), _lib.ptr(m1s), _lib.ptr(dy1s), _lib.ptr(geos), st), "is_egnn_layer_fwd"))
        layers.append(dict(psd=psd, x=x, z2s=z2s, z3s=z3s, m1s=m1s, dy1s=dy1s, geos=geos, h_neigh=h_neigh, zn1=zn1, h_in=h_in, ld_h=ld_h, din=din,
                           h_out=h_out))
        psd, x, h_in, ld_h, din = psd_next, x_out, h_out, HIDDEN, HIDDEN
        if after is not None and after[0] == i:
            after[1]()
    return layers, h_in, x, psd


def _bucket_slice(layer_params, plan):
    """the slice of a flat gradient bucket (distributed.FlatGradReducer: ``p._grad_dest``) that holds this layer's gradients in
    the plan's layout, or None (no reducer, another parameter order, a gradient being accumulated over several backward passes)"""
    d0 = getattr(layer_params[0], "_grad_dest", None)
    if d0 is None or not d0.is_cuda or d0.dtype != torch.float32:
        return None
    base = d0._base if d0._base is not None else d0
    if base.dim() != 1 or not base.is_contiguous():
        return None
    p0 = d0.data_ptr()
    for p, off, sh in zip(layer_params, plan.offsets, plan.shapes):
        d = getattr(p, "_grad_dest", None)
        if (d is None or p.grad is not None or tuple(d.shape) != tuple(sh) or not d.is_contiguous() or d.data_ptr() != p0 + 4 * off
                or (d._base if d._base is not None else d) is not base):
            return None
    so = d0.storage_offset()
    return base[so:so + plan.total]


class EGNNStackFn(torch.autograd.Function):
    """L chained EGNNConv layers on the fused layer kernels: one launch per layer forward (``csrc/egnn_layer_fwd.hip``),
    node data path + edge pass + source gather per layer backward, one batched weight-gradient launch and one batched
    reduction for the whole stack.

    forward(h0 (N, Din0) [row stride may exceed Din0], x0 (N,3), ea_csr (E,Fe) | None, csr, n_layers, final_coords, *params)
      -> (h_L (N,64), x_L (N,3) | None[, head projection (N,128)]).  Layer 0 has Din0 in {20, 64}; later layers 64.
    """

    @staticmethod
    def forward(ctx, h0, x0, ea, csr, n_layers, final_coords, prologue, *params):
        """params = 11 tensors per layer [+ (Wa, ba, Wb, bb) of an optional 128-wide projection head of the final h:
        the node attention's query / key projection, emitted by the last layer's node half].
        ``final_coords=False``: the caller does not use the last layer's coordinates (the reference's models never do:
        ``hybrid_models.py:323-324`` keeps only h) -- that layer's coordinate MLP is then not evaluated, forward or
        backward, and None is returned in place of x."""
        lib = _lib.load()
        _lib.require_device(h0, x0, ea, csr.rowptr_dst, *params)
        ctx.set_materialize_grads(False)      # an unused output (the final coordinates) reaches backward as None, not as zeros
        P = PARAMS_PER_LAYER
        has_head = len(params) == P * n_layers + 4
        if not has_head and len(params) != P * n_layers:
            raise ValueError("expected 11 parameter tensors per layer (+ 4 for the projection head)")
        if not 1 <= n_layers <= 7:
            raise ValueError("the fused stack holds 1 to 7 layers")
        n, e = csr.num_nodes, csr.num_edges
        dev = x0.device
        din0 = int(h0.shape[1])
        if din0 not in (20, HIDDEN):
            raise ValueError(f"node feature width must be 20 or {HIDDEN}, got {din0}")
        if h0.shape[0] != n or x0.shape != (n, 3):
            raise ValueError("node feature / coordinate rows must equal the number of nodes")
        fe = int(ea.shape[1]) if ea is not None else 0
        if fe > 8:
            raise ValueError("edge_feat_size > 8 is not supported by the HIP kernel")
        early = prologue if (prologue is not None and prologue.key == _prologue_key(h0, x0, params)) else None
        h0, ld_h0 = _lib.rows_ld(h0)
        ea = _lib.f32c(ea) if fe else None
        params = [_lib.f32c(p) for p in params]
        head = None
        if has_head:
            wa, ba, wb, bb = params[-4:]
            params = params[:-4]
            if wa.shape != (HIDDEN, HIDDEN) or wb.shape != (HIDDEN, HIDDEN):
                raise ValueError("projection head weights must be (64, 64)")
            head = ((wa, wb), ba, bb)
        need_grad = any(ctx.needs_input_grad)
        st = _lib.stream_ptr()
        f32 = dict(dtype=torch.float32, device=dev)
        layers = []
        if early is not None:
            psd, packs, x = early.psd, early.packs, early.x      # launched by the caller (before its stream fork)
        else:
            psd, packs, x = _launch_prologue(h0, ld_h0, din0, x0, params, head, n_layers, n, dev)
        if early is not None and early.layers is not None and early.need_grad >= need_grad and early.final_coords == bool(final_coords) \
                and early.ea_key == _ea_key(ea, csr):
            layers, (h_in, x, psd) = early.layers, early.outs      # the layer kernels are already enqueued (launch_stack_forward)
        else:
            layers, h_in, x, psd = _launch_stack_layers(h0, ld_h0, din0, ea, fe, csr, params, head, n_layers, final_coords, need_grad,
                                                        psd, packs, x)
        ctx.layers, ctx.params, ctx.csr, ctx.ea, ctx.fe, ctx.n_layers = layers, params, csr, ea, fe, n_layers
        ctx.h0_needs_grad, ctx.x0_needs_grad = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ctx.head = head
        ctx.packs = packs
        if head is not None:
            return h_in, x, psd
        return h_in, x

    @staticmethod
    def backward(ctx, g_h, g_x=None, g_head=None):
        lib = _lib.load()
        layers, params, csr, ea, fe, L = ctx.layers, ctx.params, ctx.csr, ctx.ea, ctx.fe, ctx.n_layers
        n, e = csr.num_nodes, csr.num_edges
        dev = params[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        st = _lib.stream_ptr()
        P = PARAMS_PER_LAYER
        g_hd = _lib.f32c(g_h) if g_h is not None else torch.zeros(n, HIDDEN, **f32)
        plans = [layer_plan(layers[i]["din"], fe, dev) for i in range(L)]
        # data-parallel runs: a layer's eleven gradient tensors lie in the reducer's flat bucket in exactly the plan's order
        # (module parameter order), so the reduction of the partial records writes them THERE -- nothing to pack afterwards
        gflat = [_bucket_slice(params[i * PARAMS_PER_LAYER:(i + 1) * PARAMS_PER_LAYER], plans[i]) for i in range(L)]
        gflat = [g if g is not None else torch.empty(pl.total, **f32) for g, pl in zip(gflat, plans)]
        # no gradient at the final coordinates (unused, or never produced): the last layer's backward skips its
        # coordinate-MLP half (null g_xout; csrc/egnn_edge_bwd16.hip) instead of pushing zeros through it
        g_xc = _lib.f32c(g_x) if g_x is not None else None
        # the PAIRED form (csrc/egnn_layer_bwd8.hip): one 512-thread workgroup per CU = two groups that share the staged weight tiles
        # and write ONE partial record -- half the records per launch; Fe <= 1, z3 read back, and batches of up to ~ 5.5 tiles per
        # workgroup slot (use_paired_bwd: the measured rule).  Both kernels cut the nodes into plain 16-node tiles (the greedy tile
        # list of rounds 2 - 5 lost at every size and density measured and was removed: HISTORY.md 9.6)
        paired = use_paired_bwd(n, fe)
        bwd_entry = lib.is_egnn_layer_bwd_paired if paired else lib.is_egnn_layer_bwd
        if paired:
            grid_e = max(1, min(layer_slots() // 2, ((n + 15) // 16 + 1) // 2))
        else:
            grid_e = max(1, min(layer_slots(), (n + 15) // 16))
        wg_stride, wg_proj = lib.is_egnn_node_wgrad_stride(), lib.is_egnn_node_wgrad_proj_floats()
        grid_wn, grid_wp = max(1, min(WGRAD_GRID_NODE, (n + 15) // 16)), max(1, min(WGRAD_GRID_PROJ, (n + 15) // 16))
        grid_w = max(grid_wn, grid_wp)      # records per partial buffer
        wjobs, rjobs = [], []     # weight-gradient layers / reduction jobs of the two batched launches at the end
        keep = []
        head = ctx.head
        g_psd_next = None
        head_flat = None
        if head is not None:
            # the head's projection is "the next layer's pre-projection" of the last layer
            g_psd_next = _lib.f32c(g_head) if g_head is not None else torch.zeros(n, 2 * HIDDEN, **f32)
            head_flat = torch.empty(_PROJ_STRIDE, **f32)       # [dWa ; dWb] (128 x 64) | db_b | db_a
        above = None      # (dZ1, dD, dx) of the layer above: gathered by source inside the next launch
        for i in reversed(range(L)):
            lay = layers[i]
            W1, b1, W2, b2, Wn1, bn1, Wn2, bn2, Wc1, bc1, wc2 = params[i * P:(i + 1) * P]
            ldw, din = int(W1.shape[1]), lay["din"]
            need_dh = i > 0 or ctx.h0_needs_grad
            d_h = torch.empty(n, HIDDEN, **f32) if need_dh else None
            d_hn = torch.empty(n, HIDDEN, **f32)
            has_psd = g_psd_next is not None
            is_head = has_psd and i == L - 1
            dh_total = torch.empty(n, HIDDEN, **f32) if has_psd else g_hd
            dzn1 = torch.empty(n, HIDDEN, **f32)
            part_e = torch.empty(grid_e * _EDGE_STRIDE, **f32)
            dZ1 = torch.empty(max(e, 1), HIDDEN, **f32)
            dD = torch.empty(max(e, 1), 3, **f32)
            dpsd = torch.empty(n, 2 * HIDDEN, **f32)
            dx = torch.empty(n, 3, **f32)
            psd = lay["psd"]
            dZ1n, dDn, dxn = above if above is not None else (None, None, None)
            gxtot = torch.empty(n, 3, **f32) if above is not None else None      # scratch: dL/dx_out completed by the gather
            # ONE launch: source gather of the layer above (completes g_psd_next[:, :64] and the coordinate gradient) ->
            # node data path (dh = g_h + g_psd W1sd(next), node-MLP backward) -> fused edge pass backward
            # (idempotent: every output is written from inputs the launch does not change -- bench.py times it K times back to back)
            clk = LaunchClock.slot("bwd_nocoord" if (g_xc is None and above is None) else "bwd", i, grid_e, dev)
            KernelTimer.launch("egnn_layer_bwd_nocoord" if (g_xc is None and above is None) else "egnn_layer_bwd", lambda: _lib.check(
                bwd_entry(
                    _lib.ptr(psd), _lib.ptr(psd[:, HIDDEN:]), 2 * HIDDEN, _lib.ptr(lay["x"]), _lib.ptr(ea),
                    _lib.ptr(csr.rowptr_dst), _lib.ptr(csr.src_sorted), _lib.ptr(W1), ldw, din,
                    _lib.ptr(W2), _lib.ptr(Wc1), _lib.ptr(bc1), _lib.ptr(wc2), _lib.ptr(lay["z2s"]), _lib.ptr(lay["z3s"]),
                    _lib.ptr(g_xc) if above is None else None, _lib.ptr(dZ1), _lib.ptr(dD),
                    _lib.ptr(dpsd[:, HIDDEN:]), 2 * HIDDEN, _lib.ptr(dx), _lib.ptr(part_e), grid_e, n, fe,
                    _lib.ptr(dZ1n), _lib.ptr(dDn), _lib.ptr(dxn), _lib.ptr(csr.rowptr_src), _lib.ptr(csr.pos_by_src),
                    _lib.ptr(g_hd), _lib.ptr(g_psd_next), _lib.ptr(lay["zn1"]), _lib.ptr(ctx.packs[i, 1]),
                    _lib.ptr(dh_total) if has_psd else None, _lib.ptr(dzn1), _lib.ptr(d_h), _lib.ptr(d_hn), _lib.ptr(gxtot),
                    _lib.ptr(clk), _lib.ptr(lay["m1s"]), _lib.ptr(lay["dy1s"]), _lib.ptr(lay["geos"]), st),
                "is_egnn_layer_bwd"))
            pw = torch.empty(grid_w * wg_stride, **f32)
            keep.extend([dh_total, dzn1, g_psd_next, g_hd, pw, part_e, d_hn, above, gxtot])
            wjobs.append(_lib.WgradLayer(_lib.ptr(g_psd_next).value if has_psd else None, lay["h_out"].data_ptr(),
                                         dh_total.data_ptr(), lay["zn1"].data_ptr(), dzn1.data_ptr(), lay["h_in"].data_ptr(),
                                         lay["h_neigh"].data_ptr(), pw.data_ptr(), lay["ld_h"], din, HIDDEN, HIDDEN, HIDDEN, 0))
            if is_head:
                rjobs.append((pw, grid_wp, wg_stride, wg_proj, None, head_flat))
            elif has_psd:
                rjobs.append((pw, grid_wp, wg_stride, wg_proj, plans[i + 1].proj_map, gflat[i + 1]))
            rjobs.append((pw[wg_proj:], grid_wn, wg_stride, _NODE_STRIDE, plans[i].node_map, gflat[i]))
            rjobs.append((part_e, grid_e, _EDGE_STRIDE, _EDGE_STRIDE, plans[i].edge_map, gflat[i]))
            above = (dZ1, dD, dx)
            g_hd, g_psd_next, g_xc = d_h, dpsd, dx
        # the gather of layer 0's per-edge gradients completes dL/dpsd_0 (and dL/dx_0): its own launch (no layer below)
        dZ1, dD, dx = above
        with KernelTimer.span("gather_segment_sum"):      # (accumulates into dx: not repeatable)
            _lib.check(lib.is_gather_segment_sum(_lib.ptr(dZ1), _lib.ptr(dD), _lib.ptr(csr.rowptr_src), _lib.ptr(csr.pos_by_src),
                                                 _lib.ptr(g_psd_next), 2 * HIDDEN, _lib.ptr(dx), n,
                                                 _lib.ptr(LaunchClock.slot("gather", 0, (n + 15) // 16, dev)), st), "is_gather_segment_sum")
        # (4) layer-0 pre-projection: weight gradient (+ input-feature gradient when requested)
        lay0 = layers[0]
        W1_0 = params[0]
        dh0 = torch.empty(n, HIDDEN, **f32) if ctx.h0_needs_grad else None
        keep.extend([g_hd, g_psd_next])
        if not ctx.h0_needs_grad:
            # weight gradient of the layer-0 pre-projection as one more (projection-only) job of the batched launch
            pw0 = torch.empty(grid_w * wg_stride, **f32)
            keep.append(pw0)
            wjobs.append(_lib.WgradLayer(g_psd_next.data_ptr(), lay0["h_in"].data_ptr(), None, None, None, None, None,
                                         pw0.data_ptr(), HIDDEN, HIDDEN, HIDDEN, lay0["ld_h"], lay0["din"], 0))
            rjobs.append((pw0, grid_wp, wg_stride, wg_proj, plans[0].proj_map, gflat[0]))
        else:
            grid_n = max(1, min(_MAX_BWD_GRID, (n + 127) // 128))
            part_p = torch.empty(grid_n * _PROJ_STRIDE, **f32)
            keep.append(part_p)
            with KernelTimer.span("node_proj_bwd"):
                _lib.check(lib.is_node_proj_bwd(_lib.ptr(g_hd), _lib.ptr(g_psd_next), _lib.ptr(lay0["h_in"]), lay0["ld_h"], lay0["din"],
                                                _lib.ptr(W1_0), int(W1_0.shape[1]), _lib.ptr(dh0), _lib.ptr(part_p), grid_n, n, st),
                           "is_node_proj_bwd")
            rjobs.append((part_p, grid_n, _PROJ_STRIDE, _PROJ_STRIDE, plans[0].proj_map, gflat[0]))
        arr = (_lib.WgradLayer * len(wjobs))(*wjobs)
        with KernelTimer.span("egnn_node_wgrad_batched"):
            _lib.check(lib.is_egnn_node_wgrad_batched(ctypes.cast(arr, ctypes.c_void_p), len(wjobs), grid_wn, grid_wp, n, st),
                       "is_egnn_node_wgrad_batched")
        split = lib.is_reduce_partials_scratch_floats(1)
        big = torch.empty(sum(split * c for (_, _, _, c, _, _) in rjobs), **f32)
        jobs, off = [], 0
        for (pt, nparts, stride, count, mp, dst) in rjobs:
            jobs.append(_lib.ReduceJob(pt.data_ptr(), mp.data_ptr() if mp is not None else None, dst.data_ptr(),
                                       big[off:].data_ptr(), nparts, stride, count, 0))
            off += split * count
        jarr = (_lib.ReduceJob * len(jobs))(*jobs)
        with KernelTimer.span("reduce_partials_batched"):
            _lib.check(lib.is_reduce_partials_batched(ctypes.cast(jarr, ctypes.c_void_p), len(jobs), st), "is_reduce_partials_batched")
        keep.clear()
        grads = []
        for i in range(L):
            grads.extend(plans[i].views(gflat[i]))
        g_h0 = dh0[:, :lay0["din"]] if dh0 is not None else None
        g_x0 = g_xc if ctx.x0_needs_grad else None
        if head_flat is not None:
            hh = HIDDEN * HIDDEN
            grads.extend([head_flat[0:hh].view(HIDDEN, HIDDEN), head_flat[2 * hh + HIDDEN:2 * hh + 2 * HIDDEN],
                          head_flat[hh:2 * hh].view(HIDDEN, HIDDEN), head_flat[2 * hh:2 * hh + HIDDEN]])
        return (g_h0, g_x0, None, None, None, None, None) + tuple(grads)


class PairLinearFn(torch.autograd.Function):
    """out (N,128) = [h Wa^T + ba | h Wb^T + bb] for h (N,64) -- the fused query/key projection of the
    node attention, on the node pre-projection kernels (``csrc/egnn_node.hip``)."""

    @staticmethod
    def forward(ctx, h, wa, ba, wb, bb):
        lib = _lib.load()
        _lib.require_device(h, wa, ba, wb, bb)
        if h.dim() != 2 or h.shape[1] != HIDDEN or wa.shape != (HIDDEN, HIDDEN) or wb.shape != (HIDDEN, HIDDEN):
            raise ValueError("PairLinearFn expects h (N,64) and two (64,64) weights")
        h, ld_h = _lib.rows_ld(h)
        wpack = torch.cat([wa, wb], dim=1).contiguous()      # [64][128]: columns [Wa | Wb], the W1-style layout
        ba, bb = _lib.f32c(ba), _lib.f32c(bb)
        n = int(h.shape[0])
        out = torch.empty(n, 2 * HIDDEN, dtype=torch.float32, device=h.device)
        with KernelTimer.span("attn_qk_proj_fwd"):
            _lib.check(lib.is_node_proj_fwd(_lib.ptr(h), ld_h, HIDDEN, _lib.ptr(wpack), 2 * HIDDEN, _lib.ptr(ba), _lib.ptr(bb),
                                            _lib.ptr(out), n, _lib.stream_ptr()), "is_node_proj_fwd")
        ctx.ld_h = ld_h
        ctx.save_for_backward(h, wpack)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        h, wpack = ctx.saved_tensors
        n = int(h.shape[0])
        dev = h.device
        f32 = dict(dtype=torch.float32, device=dev)
        g = _lib.f32c(g)
        st = _lib.stream_ptr()
        grid = _grid_for(n, 128)
        part = torch.empty(grid * _PROJ_STRIDE, **f32)
        dh = torch.empty(n, HIDDEN, **f32)
        with KernelTimer.span("attn_qk_proj_bwd"):
            _lib.check(lib.is_node_proj_bwd(None, _lib.ptr(g), _lib.ptr(h), ctx.ld_h, HIDDEN, _lib.ptr(wpack), 2 * HIDDEN,
                                            _lib.ptr(dh), _lib.ptr(part), grid, n, st), "is_node_proj_bwd")
        flat = torch.empty(_PROJ_STRIDE, **f32)
        scratch = torch.empty(lib.is_reduce_partials_scratch_floats(_PROJ_STRIDE), **f32)
        _lib.check(lib.is_reduce_partials(_lib.ptr(part), grid, _PROJ_STRIDE, _PROJ_STRIDE, None, _lib.ptr(flat), _lib.ptr(scratch), st),
                   "is_reduce_partials")
        hh = HIDDEN * HIDDEN
        return (dh, flat[0:hh].view(HIDDEN, HIDDEN), flat[2 * hh + HIDDEN:2 * hh + 2 * HIDDEN],
                flat[hh:2 * hh].view(HIDDEN, HIDDEN), flat[2 * hh:2 * hh + HIDDEN])


def pair_linear(h, wa, ba, wb, bb):
    return PairLinearFn.apply(h, wa, ba, wb, bb)


class AttnColMeanFn(torch.autograd.Function):
    """ctx (B, heads, 64) = sum_j abar_h[j] x_j, abar = column mean of the per-graph attention matrix
    (``csrc/node_attention.hip``).  qk (B*n, 128) = [Q | K], x (B*n, 64)."""

    @staticmethod
    def forward(ctx_, qk, x, b, n, heads):
        lib = _lib.load()
        _lib.require_device(qk, x)
        if qk.shape != (b * n, 2 * HIDDEN) or x.shape != (b * n, HIDDEN):
            raise ValueError("expected qk (B*n,128) and x (B*n,64)")
        if heads not in (1, 8) or n > 256:
            raise NotImplementedError("attention kernel supports 1 or 8 heads and <= 256 nodes per graph")
        qk, x = _lib.f32c(qk), _lib.f32c(x)
        dev = x.device
        out = torch.empty(b, heads, HIDDEN, dtype=torch.float32, device=dev)
        need = any(ctx_.needs_input_grad)
        abar = torch.empty(b, heads, n, dtype=torch.float32, device=dev) if need else None
        # the attention probabilities in accumulator-tile order (148 KB per graph and head at n = 190): read back by
        # the backward instead of recomputing the scores three times
        probs = torch.empty(lib.is_attn_colmean_probs_floats(b, n, heads), dtype=torch.float32, device=dev) if need else None
        with KernelTimer.span("attn_colmean_fwd"):
            _lib.check(lib.is_attn_colmean_fwd(_lib.ptr(qk), _lib.ptr(x), _lib.ptr(out), _lib.ptr(abar), _lib.ptr(probs),
                                               b, n, heads, _lib.stream_ptr()), "is_attn_colmean_fwd")
        ctx_.dims = (b, n, heads)
        ctx_.save_for_backward(qk, x, abar, probs)
        return out

    @staticmethod
    def backward(ctx_, g):
        lib = _lib.load()
        qk, x, abar, probs = ctx_.saved_tensors
        b, n, heads = ctx_.dims
        g = _lib.f32c(g)
        dqk = torch.empty_like(qk)
        dx = torch.empty_like(x)
        with KernelTimer.span("attn_colmean_bwd"):
            _lib.check(lib.is_attn_colmean_bwd(_lib.ptr(qk), _lib.ptr(x), _lib.ptr(abar), _lib.ptr(probs), _lib.ptr(g),
                                               _lib.ptr(dqk), _lib.ptr(dx), b, n, heads, _lib.stream_ptr()), "is_attn_colmean_bwd")
        return dqk, dx, None, None, None


def attn_colmean(qk, x, b, n, heads):
    return AttnColMeanFn.apply(qk, x, b, n, heads)


class AttnPooledTailFn(torch.autograd.Function):
    """Single head: ``AttnColMeanFn`` followed by the pooled tail u = W_c (W_v ctx + b_v) + b_c in the SAME forward launch
    (``is_attn_colmean_fwd_tail``); the backward is the tail's (``is_mlp2_bwd`` + reduction) followed by the attention's."""

    @staticmethod
    def forward(ctx_, qk, x, wv, bv, wc, bc, b, n):
        lib = _lib.load()
        _lib.require_device(qk, x, wv, bv, wc, bc)
        if qk.shape != (b * n, 2 * HIDDEN) or x.shape != (b * n, HIDDEN) or tuple(wv.shape) != (HIDDEN, HIDDEN) \
                or tuple(wc.shape) != (HIDDEN, HIDDEN):
            raise ValueError("expected qk (B*n,128), x (B*n,64) and 64x64 projections")
        if n > 256:
            raise NotImplementedError("attention kernel supports <= 256 nodes per graph")
        qk, x, wv, bv, wc, bc = (_lib.f32c(t) for t in (qk, x, wv, bv, wc, bc))
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        need = any(ctx_.needs_input_grad)
        pooled = torch.empty(b, 1, HIDDEN, **f32)
        y = torch.empty(b, HIDDEN, **f32)
        a1 = torch.empty(b, HIDDEN, **f32) if need else None
        abar = torch.empty(b, 1, n, **f32) if need else None
        probs = torch.empty(lib.is_attn_colmean_probs_floats(b, n, 1), **f32) if need else None
        with KernelTimer.span("attn_colmean_fwd"):
            _lib.check(lib.is_attn_colmean_fwd_tail(_lib.ptr(qk), _lib.ptr(x), _lib.ptr(pooled), _lib.ptr(abar), _lib.ptr(probs),
                                                    _lib.ptr(wv), _lib.ptr(bv), _lib.ptr(wc), _lib.ptr(bc), _lib.ptr(a1), _lib.ptr(y),
                                                    b, n, _lib.stream_ptr()), "is_attn_colmean_fwd_tail")
        ctx_.dims = (b, n)
        ctx_.save_for_backward(qk, x, abar, probs, pooled, wv, wc, a1, y)
        return y

    @staticmethod
    def backward(ctx_, gy):
        lib = _lib.load()
        qk, x, abar, probs, pooled, wv, wc, a1, y = ctx_.saved_tensors
        b, n = ctx_.dims
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        st = _lib.stream_ptr()
        gy = _lib.f32c(gy)
        hid = HIDDEN
        # ONE launch: every graph's workgroup derives g_ctx = W_v^T W_c^T gy itself, an extra workgroup contracts the samples
        # into the tail's parameter gradients (no per-workgroup records, no reduction launch)
        flat = torch.empty(2 * hid * hid + 2 * hid, **f32)
        dqk, dx = torch.empty_like(qk), torch.empty_like(x)
        with KernelTimer.span("attn_colmean_bwd"):
            _lib.check(lib.is_attn_colmean_bwd_tail(_lib.ptr(qk), _lib.ptr(x), _lib.ptr(abar), _lib.ptr(probs), _lib.ptr(gy),
                                                    _lib.ptr(wv), _lib.ptr(wc), _lib.ptr(pooled), _lib.ptr(a1), _lib.ptr(dqk),
                                                    _lib.ptr(dx), _lib.ptr(flat), b, n, st), "is_attn_colmean_bwd_tail")
        o1, o2, o3 = hid * hid, hid * hid + hid, 2 * hid * hid + hid
        return dqk, dx, flat[:o1].view(hid, hid), flat[o1:o2], flat[o2:o3].view(hid, hid), flat[o3:], None, None


def attn_pooled_tail(qk, x, wv, bv, wc, bc, b, n):
    return AttnPooledTailFn.apply(qk, x, wv, bv, wc, bc, b, n)


class CombinedAttentionMeanFn(torch.autograd.Function):
    """z (B,T) = mean over features of MultiHeadAttention(F, 8, input_dim=1) applied to the scalar tokens of a row
    (``csrc/combined_attention.hip``, closed form).  The row is given as 1-4 pieces (B, w_p) laid side by side -- the kernels
    read them where they are (no ``torch.cat`` in front) and the backward returns one contiguous gradient per piece (no
    slice copies behind)."""

    @staticmethod
    def forward(ctx, wq, bq, wk, bk, wv, bv, wc, bc, *pieces):
        lib = _lib.load()
        _lib.require_device(wq, bq, wk, bk, wv, bv, wc, bc, *pieces)
        if not 1 <= len(pieces) <= 4 or any(p.dim() != 2 or p.shape[0] != pieces[0].shape[0] for p in pieces):
            raise ValueError("expected 1 to 4 pieces of shape (batch, tokens_p)")
        b, t = int(pieces[0].shape[0]), sum(int(p.shape[1]) for p in pieces)
        f = int(wc.shape[0])
        if f not in (16, 32) or t > 256:
            raise NotImplementedError("combined attention kernel supports feature_dim 16/32 and <= 256 tokens")
        rows = [(_lib.f32c(p), int(p.shape[1])) for p in pieces]      # dense pieces: one leading dimension serves x and dx
        wq, bq, wk, wv, bv, wc, bc = (_lib.f32c(v) for v in (wq, bq, wk, wv, bv, wc, bc))
        dev = pieces[0].device
        z = torch.empty(b, t, dtype=torch.float32, device=dev)
        need = any(ctx.needs_input_grad)
        stats = torch.empty(lib.is_comb_attn_stats_floats(b, t), dtype=torch.float32, device=dev) if need else None
        parts = (_lib.CaPart * len(rows))(*[_lib.CaPart(x.data_ptr(), None, int(x.shape[1]), ld) for x, ld in rows])
        with KernelTimer.span("comb_attn_fwd"):
            _lib.check(lib.is_comb_attn_fwd(ctypes.cast(parts, ctypes.c_void_p), len(rows), _lib.ptr(wq), _lib.ptr(bq), _lib.ptr(wk),
                                            _lib.ptr(wv), _lib.ptr(bv), _lib.ptr(wc), _lib.ptr(bc), _lib.ptr(z), _lib.ptr(stats), b, t, f,
                                            _lib.stream_ptr()), "is_comb_attn_fwd")
        ctx.dims = (b, t, f)
        ctx.lds = [ld for _, ld in rows]
        ctx.save_for_backward(stats, wq, bq, wk, wv, bv, wc, bc, *[x for x, _ in rows])
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        stats, wq, bq, wk, wv, bv, wc, bc = ctx.saved_tensors[:8]
        xs = ctx.saved_tensors[8:]
        b, t, f = ctx.dims
        dev = dz.device
        dz = _lib.f32c(dz)
        dxs = [torch.empty(b, int(x.shape[1]), dtype=torch.float32, device=dev) for x in xs]
        part = torch.empty(lib.is_comb_attn_partials_floats(b), dtype=torch.float32, device=dev)
        g = torch.empty(lib.is_comb_attn_grad_floats(f), dtype=torch.float32, device=dev)
        parts = (_lib.CaPart * len(xs))(*[_lib.CaPart(x.data_ptr(), d.data_ptr(), int(x.shape[1]), ld)
                                          for x, d, ld in zip(xs, dxs, ctx.lds)])
        with KernelTimer.span("comb_attn_bwd"):
            _lib.check(lib.is_comb_attn_bwd(ctypes.cast(parts, ctypes.c_void_p), len(xs), _lib.ptr(stats), _lib.ptr(dz), _lib.ptr(wq),
                                            _lib.ptr(bq), _lib.ptr(wk), _lib.ptr(wv), _lib.ptr(bv), _lib.ptr(wc), _lib.ptr(bc),
                                            _lib.ptr(part), _lib.ptr(g), b, t, f, _lib.stream_ptr()), "is_comb_attn_bwd")
        col = lambda i: g[i * f:(i + 1) * f]
        return (col(0).view(f, 1), col(1), col(2).view(f, 1), col(3), col(4).view(f, 1), col(5),
                g[6 * f:6 * f + f * f].view(f, f), g[6 * f + f * f:7 * f + f * f]) + tuple(dxs)


def combined_attention_mean(x, mha):
    """``mha``: a models.layers.MultiHeadAttention built with input_dim=1 and 8 heads; ``x``: (batch, tokens) or a list of 1-4
    pieces (batch, tokens_p) that form the row side by side."""
    pieces = list(x) if isinstance(x, (list, tuple)) else [x]
    return CombinedAttentionMeanFn.apply(mha.w_q.weight, mha.w_q.bias, mha.w_k.weight, mha.w_k.bias,
                                         mha.w_v.weight, mha.w_v.bias, mha.w_concat.weight, mha.w_concat.bias, *pieces)


class CombinedAttentionClassifierFn(torch.autograd.Function):
    """``CombinedAttentionMeanFn`` followed by the classifier Linear(T, hid) - ReLU - Dropout - Linear(hid, out) in the SAME
    launches (``is_comb_attn_cls_fwd`` / ``_bwd``): one launch forward, one (+ the attention block's finish launch)
    backward -- the classifier's input gradient never leaves the workgroup, its parameter gradients come from one extra
    workgroup (no partial records, no reduction launch).  ``mask``: the scaled dropout keep-mask (B, hid) or None."""

    @staticmethod
    def forward(ctx, wq, bq, wk, bk, wv, bv, wc, bc, w1, b1, w2, b2, mask, pair_rows, *pieces):
        """``pair_rows`` = B > 0: every tensor of ``pieces`` holds the rows of a (cancer; wild-type) pair stacked, [2B, w]; sample
        i's token row is then [cancer pieces of row i | wild-type pieces of row B + i] (``comparative_models.py:478-480``) -- read
        where the rows are, and the gradients come back as ONE [2B, w] tensor per input (no slice / accumulate launches)."""
        lib = _lib.load()
        _lib.require_device(wq, bq, wk, bk, wv, bv, wc, bc, w1, b1, w2, b2, mask, *pieces)
        pr = int(pair_rows)
        if pr and any(int(p.shape[0]) != 2 * pr for p in pieces):
            raise ValueError("pair_rows: every piece must hold 2 * pair_rows rows")
        b = pr if pr else int(pieces[0].shape[0])
        t = sum(int(p.shape[1]) for p in pieces) * (2 if pr else 1)
        f = int(wc.shape[0])
        hid, out = int(w1.shape[0]), int(w2.shape[0])
        rows = [(_lib.f32c(p), int(p.shape[1])) for p in pieces]
        wq, bq, wk, wv, bv, wc, bc, w1, b1, w2, b2 = (_lib.f32c(v) for v in (wq, bq, wk, wv, bv, wc, bc, w1, b1, w2, b2))
        mask = _lib.f32c(mask) if mask is not None else None
        dev = pieces[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        z = torch.empty(b, t, **f32)
        y = torch.empty(b, out, **f32)
        need = any(ctx.needs_input_grad)
        stats = torch.empty(lib.is_comb_attn_stats_floats(b, t), **f32) if need else None
        a1 = torch.empty(b, hid, **f32) if need else None
        recs = [_lib.CaPart(x.data_ptr(), None, int(x.shape[1]), ld) for x, ld in rows]
        if pr:
            recs += [_lib.CaPart(x.data_ptr() + 4 * pr * ld, None, int(x.shape[1]), ld) for x, ld in rows]
        parts = (_lib.CaPart * len(recs))(*recs)
        with KernelTimer.span("comb_attn_cls_fwd"):
            _lib.check(lib.is_comb_attn_cls_fwd(ctypes.cast(parts, ctypes.c_void_p), len(recs), _lib.ptr(wq), _lib.ptr(bq), _lib.ptr(wk),
                                                _lib.ptr(wv), _lib.ptr(bv), _lib.ptr(wc), _lib.ptr(bc), _lib.ptr(w1), _lib.ptr(b1),
                                                _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(mask), _lib.ptr(z), _lib.ptr(stats), _lib.ptr(a1),
                                                _lib.ptr(y), b, t, f, hid, out, 0, _lib.stream_ptr()), "is_comb_attn_cls_fwd")
        ctx.dims = (b, t, f, hid, out, pr)
        ctx.lds = [ld for _, ld in rows]
        ctx.has_mask = mask is not None
        ctx.save_for_backward(stats, wq, bq, wk, wv, bv, wc, bc, w1, w2, z, a1, y, *([mask] if mask is not None else []), *[x for x, _ in rows])
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        saved = ctx.saved_tensors
        stats, wq, bq, wk, wv, bv, wc, bc, w1, w2, z, a1, y = saved[:13]
        mask = saved[13] if ctx.has_mask else None
        xs = saved[14 if ctx.has_mask else 13:]
        b, t, f, hid, out, pr = ctx.dims
        dev = gy.device
        f32 = dict(dtype=torch.float32, device=dev)
        gy = _lib.f32c(gy)
        dxs = [torch.empty(int(x.shape[0]), int(x.shape[1]), **f32) for x in xs]      # (2B rows for a stacked pair)
        part = torch.empty(lib.is_comb_attn_partials_floats(b), **f32)
        g = torch.empty(lib.is_comb_attn_grad_floats(f), **f32)
        gc = torch.empty(lib.is_comb_attn_cls_grad_floats(t, hid, out), **f32)
        recs = [_lib.CaPart(x.data_ptr(), d.data_ptr(), int(x.shape[1]), ld) for x, d, ld in zip(xs, dxs, ctx.lds)]
        if pr:
            recs += [_lib.CaPart(x.data_ptr() + 4 * pr * ld, d.data_ptr() + 4 * pr * int(x.shape[1]), int(x.shape[1]), ld)
                     for x, d, ld in zip(xs, dxs, ctx.lds)]
        parts = (_lib.CaPart * len(recs))(*recs)
        with KernelTimer.span("comb_attn_cls_bwd"):
            _lib.check(lib.is_comb_attn_cls_bwd(ctypes.cast(parts, ctypes.c_void_p), len(recs), _lib.ptr(stats), _lib.ptr(gy), _lib.ptr(wq),
                                                _lib.ptr(bq), _lib.ptr(wk), _lib.ptr(wv), _lib.ptr(bv), _lib.ptr(wc), _lib.ptr(bc),
                                                _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(mask), _lib.ptr(z), _lib.ptr(a1), _lib.ptr(y),
                                                _lib.ptr(part), _lib.ptr(g), _lib.ptr(gc), b, t, f, hid, out, 0, _lib.stream_ptr()),
                       "is_comb_attn_cls_bwd")
        col = lambda i: g[i * f:(i + 1) * f]
        o1, o2, o3 = hid * t, hid * t + hid, hid * t + hid + out * hid
        return (col(0).view(f, 1), col(1), col(2).view(f, 1), col(3), col(4).view(f, 1), col(5),
                g[6 * f:6 * f + f * f].view(f, f), g[6 * f + f * f:7 * f + f * f],
                gc[:o1].view(hid, t), gc[o1:o2], gc[o2:o3].view(out, hid), gc[o3:], None, None) + tuple(dxs)


def combined_attention_classifier(pieces, mha, classifier, mask="draw", pair_rows=0):
    """The fusion head as one launch: combined attention (``mha``: MultiHeadAttention(F, 8, input_dim=1)) over the row given as
    1-4 pieces, then ``classifier`` = nn.Sequential(Flatten, Linear, ReLU, Dropout, Linear).  Returns None when the modules do
    not have that form / exceed the kernel's limits (the caller then chains the two separate functions)."""
    pieces = list(pieces)
    mods = [m for m in classifier if not isinstance(m, torch.nn.Flatten)]
    if len(mods) != 4 or not all(p.is_cuda and p.dim() == 2 for p in pieces):
        return None
    l1, r1, dr, l2 = mods
    t = sum(int(p.shape[1]) for p in pieces) * (2 if pair_rows else 1)
    nparts = len(pieces) * (2 if pair_rows else 1)
    ok = (isinstance(l1, torch.nn.Linear) and isinstance(r1, torch.nn.ReLU) and isinstance(dr, torch.nn.Dropout)
          and isinstance(l2, torch.nn.Linear) and l1.bias is not None and l2.bias is not None and l1.in_features == t
          and l1.out_features <= 32 and l2.out_features <= 64 and t <= 256 and 1 <= nparts <= 4
          and mha.n_head == 8 and mha.w_q.in_features == 1 and int(mha.w_concat.weight.shape[0]) in (16, 32))
    if not ok:
        return None
    if isinstance(mask, str):
        rows_ = pair_rows if pair_rows else pieces[0].shape[0]
        mask = dropout_mask(rows_, l1.out_features, dr.p, pieces[0].device) if (classifier.training and dr.p > 0) else None
    return CombinedAttentionClassifierFn.apply(mha.w_q.weight, mha.w_q.bias, mha.w_k.weight, mha.w_k.bias, mha.w_v.weight, mha.w_v.bias,
                                               mha.w_concat.weight, mha.w_concat.bias, l1.weight, l1.bias, l2.weight, l2.bias, mask,
                                               int(pair_rows), *pieces)


class PairEmbeddingsFn(torch.autograd.Function):
    """(cancer, wild-type) embeddings [x_gat | z_vae] of a stacked pair (inputs [2B, w] each: cancer rows first), as the paired
    models hand them to the contrastive loss (``comparative_models.py:474-476``).  The backward writes ONE gradient per stacked
    input (two concatenations) -- slicing the stacked tensors instead costs a zero-fill + copy per slice and an add per pair."""

    @staticmethod
    def forward(ctx, xg, zv, b):
        ctx.b, ctx.wx = int(b), int(xg.shape[1])
        ctx.set_materialize_grads(False)
        return torch.cat([xg[:b], zv[:b]], dim=1), torch.cat([xg[b:], zv[b:]], dim=1)

    @staticmethod
    def backward(ctx, gc, gw):
        if gc is None and gw is None:
            return None, None, None
        ref = gc if gc is not None else gw
        gc = torch.zeros_like(ref) if gc is None else gc
        gw = torch.zeros_like(ref) if gw is None else gw
        wx = ctx.wx
        return torch.cat([gc[:, :wx], gw[:, :wx]], dim=0), torch.cat([gc[:, wx:], gw[:, wx:]], dim=0), None


def pair_embeddings(xg, zv, b):
    return PairEmbeddingsFn.apply(xg, zv, b)


def fused_head_available(n_layers):
    """the (Wa, ba, Wb, bb) projection head of the final h rides on the last layer's node half (the batched weight-gradient
    launch holds 8 jobs: the layers, the layer-0 pre-projection and the head)"""
    return n_layers <= 7


def egnn_stack(h0, x0, ea_csr, csr, layer_params, head=None, final_coords=True, prologue=None):
    """layer_params: list (one entry per layer) of the 11 native parameter tensors; ``head`` = optional
    (Wa, ba, Wb, bb), 64x64 each: the call then also returns [h Wa^T + ba | h Wb^T + bb] (N, 128) of the final h.
    ``final_coords=False``: the last layer's coordinates are not needed; None may be returned in their place."""
    flat = [p for lp in layer_params for p in lp]
    if head is not None:
        flat = flat + list(head)
    return EGNNStackFn.apply(h0, x0, ea_csr, csr, len(layer_params), bool(final_coords), prologue, *flat)


class Mlp2Fn(torch.autograd.Function):
    """y = act2(W2 (mask * act1(W1 X + b1)) + b2) per sample (``csrc/mlp_head.hip``): the classifier, the property
    embedding and the pooled attention's W_v / w_concat tail as one launch forward, one (+ reduction) backward.

    x (B, xrow); ``hgroup`` > 0: every group of hgroup hidden units reads its own ``in``-wide slice of the row."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, mask, act1, act2, hgroup):
        lib = _lib.load()
        _lib.require_device(x, w1, b1, w2, b2, mask)
        if x.dim() != 2:
            raise ValueError("expected x (batch, features)")
        x, ld_x = _lib.rows_ld(x)
        w1, b1, w2, b2 = (_lib.f32c(t) for t in (w1, b1, w2, b2))
        mask = _lib.f32c(mask) if mask is not None else None
        b = int(x.shape[0])
        hid, inn = int(w1.shape[0]), int(w1.shape[1])
        out = int(w2.shape[0])
        heads = hid // hgroup if hgroup > 0 else 1
        if int(x.shape[1]) != heads * inn or int(w2.shape[1]) != hid or (mask is not None and tuple(mask.shape) != (b, hid)):
            raise ValueError("Mlp2Fn: inconsistent shapes")
        dev = x.device
        y = torch.empty(b, out, dtype=torch.float32, device=dev)
        need = any(ctx.needs_input_grad)
        a1 = torch.empty(b, hid, dtype=torch.float32, device=dev) if need else None
        with KernelTimer.span("mlp2_fwd"):
            _lib.check(lib.is_mlp2_fwd(_lib.ptr(x), ld_x, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(mask),
                                       _lib.ptr(a1), _lib.ptr(y), b, inn, hid, out, hgroup, act1, act2, _lib.stream_ptr()), "is_mlp2_fwd")
        ctx.cfg = (ld_x, b, inn, hid, out, hgroup, act1, act2, heads)
        ctx.save_for_backward(x, w1, w2, mask, a1, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, w1, w2, mask, a1, y = ctx.saved_tensors
        ld_x, b, inn, hid, out, hgroup, act1, act2, heads = ctx.cfg
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        gy = _lib.f32c(gy)
        st = _lib.stream_ptr()
        nrec, rec = lib.is_mlp2_bwd_records(b), lib.is_mlp2_bwd_record_floats(inn, hid, out)
        part = torch.empty(nrec * rec, **f32)
        gx = torch.empty(b, heads * inn, **f32) if ctx.needs_input_grad[0] else None
        with KernelTimer.span("mlp2_bwd"):
            _lib.check(lib.is_mlp2_bwd(_lib.ptr(x), ld_x, _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(mask), _lib.ptr(a1), _lib.ptr(y),
                                       _lib.ptr(gy), _lib.ptr(gx), _lib.ptr(part), b, inn, hid, out, hgroup, act1, act2, st), "is_mlp2_bwd")
            flat = torch.empty(rec, **f32)
            scratch = torch.empty(lib.is_reduce_partials_scratch_floats(rec), **f32)
            _lib.check(lib.is_reduce_partials(_lib.ptr(part), nrec, rec, rec, None, _lib.ptr(flat), _lib.ptr(scratch), st), "is_reduce_partials")
        o1, o2, o3 = hid * inn, hid * inn + hid, hid * inn + hid + out * hid
        return (gx, flat[:o1].view(hid, inn), flat[o1:o2], flat[o2:o3].view(out, hid), flat[o3:], None, None, None, None)


def mlp2(x, w1, b1, w2, b2, mask=None, act1=0, act2=0, hgroup=0):
    return Mlp2Fn.apply(x, w1, b1, w2, b2, mask, int(act1), int(act2), int(hgroup))


_ones_cache = {}


class StepRandom:
    """The random tensors of a train step -- dropout keep-masks, the reparameterisation noise -- from ONE launch of the library's own
    generator inside the step (``is_step_random``: Philox4x32-10 on a device-resident state seeded from ``torch.cuda.initial_seed()``),
    issued at the step's first draw.

    Why: a captured graph that uses torch's generator makes every ``CUDAGraph.replay()`` launch two fill kernels (the generator's
    seed and offset) in front of the graph -- 10.6 us on the step's critical chain -- and the three draws are launch-bound torch
    kernels inside the step.  With a provider active (``StepRandom.use``) the models get static buffers instead of drawing.  The
    values are not torch's streams (same distributions; reproducible for a given seed and ``stream_id``).  (Drawing with torch's
    generator one step ahead on a helper stream was measured slower and is gone: HISTORY.md 7.8.)"""
    active = None

    def __init__(self, device, stream_id=0):
        """``stream_id`` is mixed into the Philox key: the engines of a run's stages pass their stage's index (``procedures``: pretrain
        0, finetune 1) so that they draw DIFFERENT mask / noise sequences from the same device seed -- a property of the stage, not
        of how many providers the process built before (a resumed process that builds only the finetune engine draws what the full
        run drew there).  :meth:`state_dict` / :meth:`load_state_dict` carry key and step counter across a checkpoint."""
        self.mode = "device"
        self.slots, self.cursor = [], 0
        key = (torch.cuda.initial_seed() + 0x9E3779B97F4A7C15 * int(stream_id)) & 0x7FFFFFFFFFFFFFFF
        self.state = torch.tensor([key, 0, 0], dtype=torch.int64, device=device)

    def state_dict(self):
        """the generator's key and step counter (host copies) -- what a resumed run needs to continue the sequence"""
        return {"mode": "device", "state": [int(v) for v in self.state.tolist()]}

    def load_state_dict(self, sd):
        if sd.get("mode") != "device":
            raise ValueError(f"StepRandom: checkpoint of mode {sd.get('mode')!r}")
        self.state.copy_(torch.tensor(sd["state"], dtype=torch.int64))      # in place: captured steps hold the address

    def _launch(self, slots):
        jobs = [_lib.RandJob(s["buf"].data_ptr(), s["buf"].numel(), 0 if s["kind"] == "randn" else 1, float(s["p"])) for s in slots]
        arr = (_lib.RandJob * len(jobs))(*jobs)
        _lib.check(_lib.load().is_step_random(ctypes.cast(arr, ctypes.c_void_p), len(jobs), _lib.ptr(self.state), _lib.stream_ptr()),
                   "is_step_random")

    class _Use:
        def __init__(self, provider):
            self.provider = provider

        def __enter__(self):
            self.saved, StepRandom.active = StepRandom.active, self.provider
            if self.provider is not None:
                self.provider.cursor = 0
            return self.provider

        def __exit__(self, *exc):
            StepRandom.active = self.saved
            return False

    @classmethod
    def use(cls, provider):
        """the models' draws inside this context come from ``provider`` (None: they draw themselves)"""
        return cls._Use(provider)

    def draw(self, kind, like, p=0.0):
        i, self.cursor = self.cursor, self.cursor + 1
        if i == len(self.slots):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("StepRandom: a draw the eager warm-up step did not make appeared during capture")
            if like.dtype != torch.float32:
                raise RuntimeError("StepRandom: float32 draws only")
            # the first (eager) step learns the step's draws one by one: a launch each
            self.slots.append({"kind": kind, "p": p, "buf": torch.empty(like.shape, dtype=torch.float32, device=like.device)})
            self._launch(self.slots[-1:])
            return self.slots[-1]["buf"]
        s = self.slots[i]
        if s["kind"] != kind or s["buf"].shape != like.shape or s["p"] != p:
            raise RuntimeError(f"StepRandom: draw {i} was {s['kind']} {tuple(s['buf'].shape)} p={s['p']}, now {kind} {tuple(like.shape)} p={p}")
        if i == 0:
            self._launch(self.slots)      # every tensor of the step, on the stream of the step's first draw (the others follow it)
        return s["buf"]


def randn_like(t):
    """``torch.randn_like(t)``, or the step's draw from the engine's provider (:class:`StepRandom`)"""
    prov = StepRandom.active
    return torch.randn_like(t) if prov is None else prov.draw("randn", t)


def dropout_mask(rows, cols, p, device):
    """scaled keep-mask of nn.Dropout(p) in training mode (torch's generator: capturable), as an explicit tensor"""
    key = (rows, cols, str(device))
    if key not in _ones_cache:
        _ones_cache[key] = torch.ones(rows, cols, dtype=torch.float32, device=device)
    prov = StepRandom.active
    if prov is not None:
        return prov.draw("dropout", _ones_cache[key], p)
    return torch.nn.functional.dropout(_ones_cache[key], p=p, training=True)


def sequential_dropout_mask(seq, rows, device):
    """the scaled keep-mask :func:`sequential_mlp2` would draw for ``seq`` on ``rows`` samples (None in eval mode / p = 0) -- so that
    a caller can draw it EARLY, off the critical path (the models draw the classifier's mask on the sequence branch's stream)"""
    mods = [m for m in seq if not isinstance(m, torch.nn.Flatten)]
    if len(mods) < 4 or not isinstance(mods[2], torch.nn.Dropout) or not isinstance(mods[0], torch.nn.Linear):
        return None
    dr = mods[2]
    return dropout_mask(rows, mods[0].out_features, dr.p, device) if (seq.training and dr.p > 0) else None


def sequential_mlp2(seq, x, mask="draw"):
    """Run an ``nn.Sequential`` of the form [Flatten,] Linear, ReLU, Dropout, Linear [, ReLU] (the reference's classifier
    and property embedding) through :func:`mlp2`; returns None when the module does not have that form or the sizes
    exceed the kernel's limits (the caller then uses the module itself)."""
    mods = [m for m in seq if not isinstance(m, torch.nn.Flatten)]
    if len(mods) not in (4, 5) or not x.is_cuda or x.dim() != 2:
        return None
    l1, r1, dr, l2 = mods[:4]
    ok = (isinstance(l1, torch.nn.Linear) and isinstance(r1, torch.nn.ReLU) and isinstance(dr, torch.nn.Dropout)
          and isinstance(l2, torch.nn.Linear) and (len(mods) == 4 or isinstance(mods[4], torch.nn.ReLU)))
    if not ok or l1.in_features > 256 or l1.out_features > 64 or l2.out_features > 64 or l1.bias is None or l2.bias is None:
        return None
    if isinstance(mask, str):      # "draw": draw the keep-mask now; otherwise the caller's (possibly None) pre-drawn mask
        mask = dropout_mask(x.shape[0], l1.out_features, dr.p, x.device) if (seq.training and dr.p > 0) else None
    return mlp2(x, l1.weight, l1.bias, l2.weight, l2.bias, mask=mask, act1=1, act2=1 if len(mods) == 5 else 0)


class PairedContrastiveFn(torch.autograd.Function):
    """``PairedContrastiveLoss`` value and embedding gradients on ``csrc/contrastive.hip`` (5 launches).  The projector
    (w1, gamma, beta, w2) is frozen in the reference (never handed to the optimizer): no parameter gradients."""

    @staticmethod
    def forward(ctx, emb_c, emb_w, pos, w1, gamma, beta, w2, lam, gate=None, scale=1.0):
        """``gate`` (device tensor with one element, or None) and ``scale``: the result is scale * gate * loss, applied inside the
        launches (the reference's early-out as a 0 / 1 factor and the caller's loss coefficient: no multiply launches behind)"""
        lib = _lib.load()
        _lib.require_device(emb_c, emb_w, pos, w1, gamma, beta, w2, gate)
        if emb_c.shape != emb_w.shape or emb_c.dim() != 2:
            raise AssertionError("cancer / wild-type embeddings must have equal (batch, features) shapes")
        b, e = int(emb_c.shape[0]), int(emb_c.shape[1])
        if not (2 <= b <= 256) or e > 256 or tuple(w1.shape) != (128, e) or tuple(w2.shape) != (128, 128):
            raise NotImplementedError("contrastive kernel: 2 <= batch <= 256, embedding <= 256, projector width 128")
        emb_c, emb_w, pos, w1, gamma, beta, w2 = (_lib.f32c(t) for t in (emb_c, emb_w, pos, w1, gamma, beta, w2))
        dev = emb_c.device
        scratch = torch.empty(lib.is_contrastive_scratch_floats(b), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)      # its own buffer: no clone launch for the result
        gate = _lib.f32c(gate.reshape(1)) if gate is not None else None
        with KernelTimer.span("contrastive_fwd"):
            _lib.check(lib.is_contrastive_fwd(_lib.ptr(emb_c), _lib.ptr(emb_w), e, e, _lib.ptr(pos), _lib.ptr(w1), _lib.ptr(gamma),
                                              _lib.ptr(beta), _lib.ptr(w2), float(lam), _lib.ptr(scratch), _lib.ptr(loss),
                                              _lib.ptr(gate), float(scale), b, _lib.stream_ptr()), "is_contrastive_fwd")
        ctx.cfg = (b, e, float(lam), float(scale))
        ctx.has_gate = gate is not None
        ctx.save_for_backward(pos, w1, gamma, w2, scratch, *([gate] if gate is not None else []))
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        pos, w1, gamma, w2, scratch = ctx.saved_tensors[:5]
        gate = ctx.saved_tensors[5] if ctx.has_gate else None
        b, e, lam, scale = ctx.cfg
        dev = scratch.device
        work = torch.empty(lib.is_contrastive_work_floats(b), dtype=torch.float32, device=dev)
        dc = torch.empty(b, e, dtype=torch.float32, device=dev)
        dw = torch.empty(b, e, dtype=torch.float32, device=dev)
        gl = _lib.f32c(g.reshape(1))
        with KernelTimer.span("contrastive_bwd"):
            _lib.check(lib.is_contrastive_bwd(_lib.ptr(pos), _lib.ptr(w1), _lib.ptr(gamma), _lib.ptr(w2), lam, _lib.ptr(scratch),
                                              _lib.ptr(work), _lib.ptr(gl), _lib.ptr(gate), scale, _lib.ptr(dc), _lib.ptr(dw), e, e, b,
                                              _lib.stream_ptr()), "is_contrastive_bwd")
        return dc, dw, None, None, None, None, None, None, None, None


def contrastive_targets(target):
    """(pos, gate): pos = (target > mean) as floats, gate = [exactly two distinct values] as a 0-d float tensor; one launch"""
    lib = _lib.load()
    _lib.require_device(target)
    t = _lib.f32c(target.reshape(-1).to(torch.float32))
    b = int(t.numel())
    if not 1 <= b <= 1024:
        raise NotImplementedError("contrastive_targets: 1 <= batch <= 1024")
    pos = torch.empty(b, dtype=torch.float32, device=t.device)
    gate = torch.empty(1, dtype=torch.float32, device=t.device)
    _lib.check(lib.is_contrastive_targets(_lib.ptr(t), _lib.ptr(pos), _lib.ptr(gate), b, _lib.stream_ptr()), "is_contrastive_targets")
    return pos, gate[0]


def paired_contrastive(emb_c, emb_w, pos, w1, gamma, beta, w2, lam, gate=None, scale=1.0):
    return PairedContrastiveFn.apply(emb_c, emb_w, pos, w1, gamma, beta, w2, lam, gate, scale)


def _linear_forward(x, w, b):
    """``F.linear(x, w, b)`` for a small batch; a long contraction into a small output (vae_fc1) goes through the
    split-contraction kernel (``is_linear_fwd_long``), everything else through the library"""
    n, k = int(w.shape[0]), int(w.shape[1])
    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and w.dtype == torch.float32 and k >= 8 * n and x.shape[0] <= 1024
            and x.is_contiguous() and w.is_contiguous() and (b is None or (b.dtype == torch.float32 and b.is_contiguous()))):
        lib = _lib.load()
        bsz = int(x.shape[0])
        y = torch.empty(bsz, n, dtype=torch.float32, device=x.device)
        scratch = torch.empty(int(lib.is_linear_dgrad_scratch_floats(bsz, k, n)), dtype=torch.float32, device=x.device)
        with KernelTimer.span("linear_fwd_long"):
            _lib.check(lib.is_linear_fwd_long(_lib.ptr(x), k, _lib.ptr(w), k, _lib.ptr(b), _lib.ptr(y), _lib.ptr(scratch), bsz, k, n,
                                              _lib.stream_ptr()), "is_linear_fwd_long")
        return y
    return torch.nn.functional.linear(x, w, b)


class LinearSmallBatchFn(torch.autograd.Function):
    """``F.linear(x, w, b)`` whose weight / bias gradients come from ``csrc/dense.hip`` (contraction over the small batch),
    and whose input gradient does too when the contraction is long and the output small (``is_linear_dgrad``); forward and
    the other input gradients are the library GEMMs."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        # data-parallel runs: the reducer's slice of the flat gradient bucket for this weight (distributed.FlatGradReducer);
        # the weight gradient is then written there directly instead of being copied in when the bucket is packed
        ctx.dest = getattr(w, "_grad_dest", None)
        ctx.fwd_stream = torch.cuda.current_stream(x.device) if x.is_cuda else None
        ctx.spec = None
        return _linear_forward(x, w, b)

    @staticmethod
    def launch_backward(ctx, gy, need_gx):
        """(gx, dw, db) for the upstream gradient ``gy`` on the current stream"""
        lib = _lib.load()
        x, w = ctx.saved_tensors
        gy = _lib.f32c(gy)
        xc = _lib.f32c(x)
        n, k, bsz = int(w.shape[0]), int(w.shape[1]), int(xc.shape[0])
        dest = ctx.dest
        # (not while a gradient is being accumulated over several backward passes: the slice IS the accumulator then)
        direct = (dest is not None and w.grad is None and dest.shape == w.shape and dest.is_contiguous() and dest.device == w.device)
        # (a NEW view object of the slice: autograd's AccumulateGrad adopts a gradient without copying only when nothing else
        #  references the tensor it is handed -- the reducer's own view object would be cloned, 12 MB, and packed back)
        dw = dest.view_as(dest) if direct else torch.empty(n, k, dtype=torch.float32, device=w.device)
        db = torch.empty(n, dtype=torch.float32, device=w.device) if ctx.has_bias else None
        # the input gradient first: the rest of the branch's backward waits for it, nothing waits for the weight gradient
        gx = None
        if need_gx:
            wc = _lib.f32c(w)
            if n >= 8 * k and bsz <= 1024:
                # long contraction, small output (vae_fc4 going backward): split-contraction kernel (csrc/dense.hip)
                gx = torch.empty(bsz, k, dtype=torch.float32, device=w.device)
                scratch = torch.empty(int(lib.is_linear_dgrad_scratch_floats(bsz, n, k)), dtype=torch.float32, device=w.device)
                with KernelTimer.span("linear_dgrad"):
                    _lib.check(lib.is_linear_dgrad(_lib.ptr(gy), n, _lib.ptr(wc), k, _lib.ptr(gx), _lib.ptr(scratch), bsz, n, k,
                                                   _lib.stream_ptr()), "is_linear_dgrad")
            else:
                gx = gy @ w
        with KernelTimer.span("linear_wgrad"):
            _lib.check(lib.is_linear_wgrad(_lib.ptr(gy), n, _lib.ptr(xc), k, _lib.ptr(dw), _lib.ptr(db), bsz, n, k, _lib.stream_ptr()),
                       "is_linear_wgrad")
        return gx, dw, db

    @staticmethod
    def backward(ctx, gy):
        spec, ctx.spec = ctx.spec, None
        if spec is not None and gy.data_ptr() == spec[0].data_ptr() and gy.shape == spec[0].shape:
            # the gradients were launched ahead of time for exactly this upstream gradient (speculate_recon_backward)
            return spec[1], spec[2], spec[3]
        return LinearSmallBatchFn.launch_backward(ctx, gy, ctx.needs_input_grad[0])


def linear_small_batch(x, weight, bias):
    """nn.Linear forward for (batch, features) inputs on a GPU; plain F.linear otherwise"""
    if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32:
        return LinearSmallBatchFn.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


class VaeLatentFn(torch.autograd.Function):
    """The sequence VAE's latent block (``csrc/vae_latent.hip``): from a1 = vae_fc1(x) (pre-activation) to
    (mu, logvar, [z | p], h3 = relu(vae_fc3([z | p]))) in one launch; backward = two launches (data path, weight gradients) -- with ``vae_fc1``'s weight gradient between them
    when that layer runs inside the node (``fc1=``).
    ``eps`` is the caller's ``torch.randn_like`` draw (reference ``hybrid_models.py:301-304``); ``p`` may be None."""

    @staticmethod
    def forward(ctx, a1, w21, b21, w22, b22, eps, p, w3, b3, x=None, w1=None, b1=None):
        lib = _lib.load()
        ctx.fc1 = w1 is not None
        if ctx.fc1:
            # the first layer inside this node (``a1`` is ignored): forward = the library GEMM; backward launches its weight
            # gradient BETWEEN this block's data path and weight pass -- the order the rest of the step's schedule wants
            # (DESIGN.md section 5.1), which separate autograd nodes cannot express
            a1 = _linear_forward(x, w1, b1)
            ctx.fc1_dest = getattr(w1, "_grad_dest", None)
            ctx.fc1_w, ctx.fc1_bias = w1, b1
        _lib.require_device(a1, w21, b21, w22, b22, eps, p, w3, b3)
        a1, w21, b21, w22, b22, eps, w3, b3 = (_lib.f32c(t) for t in (a1, w21, b21, w22, b22, eps, w3, b3))
        p = _lib.f32c(p) if p is not None else None
        b, hd = int(a1.shape[0]), int(a1.shape[1])
        lat = int(w21.shape[0])
        pw = int(p.shape[1]) if p is not None else 0
        if tuple(w21.shape) != (lat, hd) or tuple(w22.shape) != (lat, hd) or tuple(w3.shape) != (hd, lat + pw) or tuple(eps.shape) != (b, lat):
            raise ValueError("VaeLatentFn: inconsistent shapes")
        f32 = dict(dtype=torch.float32, device=a1.device)
        mu, logvar = torch.empty(b, lat, **f32), torch.empty(b, lat, **f32)
        zp, h3 = torch.empty(b, lat + pw, **f32), torch.empty(b, hd, **f32)
        with KernelTimer.span("vae_latent_fwd"):
            _lib.check(lib.is_vae_latent_fwd(_lib.ptr(a1), _lib.ptr(w21), _lib.ptr(b21), _lib.ptr(w22), _lib.ptr(b22), _lib.ptr(eps),
                                             _lib.ptr(p), pw, _lib.ptr(w3), _lib.ptr(b3), _lib.ptr(mu), _lib.ptr(logvar), _lib.ptr(zp),
                                             _lib.ptr(h3), b, hd, lat, _lib.stream_ptr()), "is_vae_latent_fwd")
        ctx.dims = (b, hd, lat, pw)
        ctx.has_p = p is not None
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(a1, w21, w22, eps, w3, logvar, zp, h3, *((x,) if ctx.fc1 else ()))
        return mu, logvar, zp, h3

    @staticmethod
    def backward(ctx, g_mu, g_lv, g_zp, g_h3):
        lib = _lib.load()
        a1, w21, w22, eps, w3, logvar, zp, h3 = ctx.saved_tensors[:8]
        b, hd, lat, pw = ctx.dims
        f32 = dict(dtype=torch.float32, device=a1.device)
        g_mu, g_lv, g_zp, g_h3 = (None if g is None else _lib.f32c(g) for g in (g_mu, g_lv, g_zp, g_h3))
        d_a3, d_a1 = torch.empty(b, hd, **f32), torch.empty(b, hd, **f32)
        dmu, dlv = torch.empty(b, lat, **f32), torch.empty(b, lat, **f32)
        d_p = torch.empty(b, pw, **f32) if pw else None
        wg = torch.empty(lib.is_vae_latent_grad_floats(hd, pw), **f32)
        st = _lib.stream_ptr()
        dw1 = db1 = None
        with KernelTimer.span("vae_latent_bwd"):
            _lib.check(lib.is_vae_latent_bwd_data(_lib.ptr(g_h3), _lib.ptr(h3), _lib.ptr(g_mu), _lib.ptr(g_lv), _lib.ptr(g_zp),
                                                  _lib.ptr(eps), _lib.ptr(logvar), _lib.ptr(a1), _lib.ptr(w21), _lib.ptr(w22), pw,
                                                  _lib.ptr(w3), _lib.ptr(d_a3), _lib.ptr(dmu), _lib.ptr(dlv), _lib.ptr(d_p),
                                                  _lib.ptr(d_a1), b, hd, lat, st), "is_vae_latent_bwd_data")
        if ctx.fc1:
            x, w1 = _lib.f32c(ctx.saved_tensors[8]), ctx.fc1_w
            kin = int(w1.shape[1])
            dest = ctx.fc1_dest
            direct = (dest is not None and w1.grad is None and dest.shape == w1.shape and dest.is_contiguous() and dest.device == w1.device)
            dw1 = dest.view_as(dest) if direct else torch.empty(hd, kin, **f32)      # (a new view object: see LinearSmallBatchFn)
            db1 = torch.empty(hd, **f32) if ctx.fc1_bias is not None else None
            with KernelTimer.span("linear_wgrad"):
                _lib.check(lib.is_linear_wgrad(_lib.ptr(d_a1), hd, _lib.ptr(x), kin, _lib.ptr(dw1), _lib.ptr(db1), b, hd, kin, st),
                           "is_linear_wgrad")
        with KernelTimer.span("vae_latent_bwd_wgrad"):
            _lib.check(lib.is_vae_latent_bwd_wgrad(_lib.ptr(a1), _lib.ptr(dmu), _lib.ptr(dlv), _lib.ptr(zp), _lib.ptr(d_a3), pw,
                                                   _lib.ptr(wg), b, hd, lat, st), "is_vae_latent_bwd_wgrad")
        n2 = lat * hd
        o = 2 * n2 + 2 * lat
        n3 = hd * (lat + pw)
        return (None if ctx.fc1 else d_a1, wg[:n2].view(lat, hd), wg[2 * n2:2 * n2 + lat], wg[n2:2 * n2].view(lat, hd),
                wg[2 * n2 + lat:o], None, d_p if ctx.has_p else None, wg[o:o + n3].view(hd, lat + pw), wg[o + n3:],
                None, dw1, db1)


def vae_latent_supported(a1, latent, p, hidden=None):
    """``a1``: the first layer's output -- or its INPUT x together with ``hidden`` (the first layer's width), for the form of
    :func:`vae_latent` that runs the first layer itself"""
    hd = a1.shape[1] if hidden is None else hidden
    return (a1.is_cuda and a1.dim() == 2 and a1.dtype == torch.float32 and latent == 32 and hd % 16 == 0
            and 16 <= hd <= 2048 and (p is None or (p.dim() == 2 and p.shape[1] <= 16))
            and (hidden is None or not a1.requires_grad))


def vae_latent(a1, w21, b21, w22, b22, eps, p, w3, b3, fc1=None):
    """-> (mu, logvar, [z | p], relu(vae_fc3([z | p]))).  ``fc1`` = (x, weight, bias): compute ``a1 = vae_fc1(x)`` inside the node
    (``a1`` is then only a shape / dtype carrier and may be None); x must not require a gradient."""
    if fc1 is not None:
        x, w1, b1 = fc1
        if x.requires_grad:
            raise ValueError("vae_latent(fc1=...): the input of the first layer must not require a gradient")
        return VaeLatentFn.apply(None, w21, b21, w22, b22, eps, p, w3, b3, x, w1, b1)
    return VaeLatentFn.apply(a1, w21, b21, w22, b22, eps, p, w3, b3, None, None, None)


class SegmentPoolFn(torch.autograd.Function):
    """Per-segment mean and/or max over rows (``csrc/segment_ops.hip``).

    ``mode``: "mean", "max" or "meanmax" (returns mean || max along dim 1).
    """

    @staticmethod
    def forward(ctx, x, seg_ptr, mode):
        lib = _lib.load()
        _lib.require_device(x, seg_ptr)
        if x.dim() != 2:
            raise ValueError("segment pooling expects a 2-d (rows, channels) tensor")
        if seg_ptr.dtype != torch.int32:
            raise ValueError("seg_ptr must be int32")
        x, ld_x = _lib.rows_ld(x)
        s, c = int(seg_ptr.numel()) - 1, int(x.shape[1])
        want_mean, want_max = mode in ("mean", "meanmax"), mode in ("max", "meanmax")
        if not (want_mean or want_max):
            raise ValueError(f"unknown pooling mode {mode!r}")
        o_mean = torch.empty(s, c, dtype=torch.float32, device=x.device) if want_mean else None
        o_max = torch.empty(s, c, dtype=torch.float32, device=x.device) if want_max else None
        code = lib.is_segment_pool_fwd(_lib.ptr(x), ld_x, _lib.ptr(seg_ptr), _lib.ptr(o_mean), _lib.ptr(o_max), s, c,
                                       _lib.stream_ptr())
        _lib.check(code, "is_segment_pool_fwd")
        ctx.mode, ctx.ld_x, ctx.c = mode, ld_x, c
        ctx.save_for_backward(x, seg_ptr, o_max)
        if mode == "mean":
            return o_mean
        if mode == "max":
            return o_max
        return torch.cat([o_mean, o_max], dim=1)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, seg_ptr, o_max = ctx.saved_tensors
        c, mode = ctx.c, ctx.mode
        g = _lib.f32c(g)
        g_mean = g_max = None
        if mode == "mean":
            g_mean = g
        elif mode == "max":
            g_max = g
        else:
            g_mean, g_max = g[:, :c].contiguous(), g[:, c:].contiguous()
        dx = torch.zeros(x.shape[0], c, dtype=torch.float32, device=x.device)
        code = lib.is_segment_pool_bwd(_lib.ptr(x), ctx.ld_x, _lib.ptr(seg_ptr), _lib.ptr(o_max), _lib.ptr(g_mean),
                                       _lib.ptr(g_max), _lib.ptr(dx), c, int(seg_ptr.numel()) - 1, c,
                                       _lib.stream_ptr())
        _lib.check(code, "is_segment_pool_bwd")
        return dx, None, None


def segment_pool(x, seg_ptr, mode="mean"):
    return SegmentPoolFn.apply(x, seg_ptr, mode)


class VaeLossFn(torch.autograd.Function):
    """Fused prediction + reconstruction + KLD loss (``csrc/losses.hip``).

    Value and all input gradients are produced by the forward launch; backward
    scales them by the upstream scalar.
    """

    @staticmethod
    def forward(ctx, recon, x, mu, logvar, logit, y, mode, pos_weight, c_pred, c_mse, c_kld, pre=None):
        lib = _lib.load()
        _lib.require_device(logit, y, recon, x, mu, logvar)
        logit_c = _lib.f32c(logit.reshape(-1))
        y_c = _lib.f32c(y.reshape(-1).to(torch.float32))
        b = int(logit_c.numel())
        if y_c.numel() != b:
            raise ValueError(f"target has {y_c.numel()} elements, prediction {b}")
        dev = logit.device
        seq = recon is not None
        if seq:
            recon_c, x_c = _lib.f32c(recon), _lib.f32c(x.reshape(recon.shape))
            mu_c, lv_c = _lib.f32c(mu), _lib.f32c(logvar)
            d_mu, d_lv = torch.empty_like(mu_c), torch.empty_like(lv_c)
            d_recon = torch.empty_like(recon_c) if pre is None else pre[0]
            rt, lt = recon_c.numel(), mu_c.numel()
        else:
            recon_c = x_c = mu_c = lv_c = d_recon = d_mu = d_lv = None
            rt = lt = 0
        d_logit = torch.empty(b, dtype=torch.float32, device=dev)
        partials = torch.empty(lib.is_loss_partials_floats(), dtype=torch.float32, device=dev) if pre is None else pre[1]
        out = torch.empty(4, dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)      # its own buffer: no clone launch for the differentiable result
        # (pre: stage 1 of the reconstruction term already ran -- recon = NULL tells the library so)
        code = lib.is_vae_loss(_lib.ptr(recon_c) if pre is None else None, _lib.ptr(x_c), _lib.ptr(d_recon), rt, _lib.ptr(mu_c), _lib.ptr(lv_c),
                               _lib.ptr(d_mu), _lib.ptr(d_lv), lt, _lib.ptr(logit_c), _lib.ptr(y_c), _lib.ptr(d_logit),
                               b, int(mode), float(pos_weight), float(c_pred), float(c_mse), float(c_kld),
                               _lib.ptr(partials), _lib.ptr(out), _lib.ptr(total), _lib.stream_ptr())
        _lib.check(code, "is_vae_loss")
        ctx.seq = seq
        ctx.logit_shape = logit.shape
        ctx.set_materialize_grads(False)       # no zero-fill launch for the (non-differentiable) term vector's gradient
        ctx.save_for_backward(d_recon, d_mu, d_lv, d_logit)
        ctx.mark_non_differentiable(out)
        return total, out

    @staticmethod
    def backward(ctx, g, _g_terms):
        d_recon, d_mu, d_lv, d_logit = ctx.saved_tensors
        if g is None:
            return (None,) * 12
        if g is _unit_gradients.get((g.device.type, g.device.index)):
            # the engine seeds the backward with unit_gradient(): d loss / d loss = 1 exactly, no scaling launches
            return d_recon, None, d_mu, d_lv, d_logit.reshape(ctx.logit_shape), None, None, None, None, None, None, None
        if ctx.seq:
            gr, gm, gl, gz = torch._foreach_mul([d_recon, d_mu, d_lv, d_logit], g)     # one multi-tensor launch
        else:
            gr, gm, gl, gz = None, None, None, d_logit * g
        return gr, None, gm, gl, gz.reshape(ctx.logit_shape), None, None, None, None, None, None, None


_unit_gradients = {}


def unit_gradient(device):
    """THE ones scalar to seed ``loss.backward(unit_gradient(dev))`` with: ``VaeLossFn`` recognises it by identity and hands
    its stored gradients on unscaled (no ones_like fill, no multi-tensor multiply of the 3 MB reconstruction gradient)"""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _unit_gradients:
        _unit_gradients[key] = torch.ones((), dtype=torch.float32, device=device)
    return _unit_gradients[key]


class SpeculativeBackward:
    """While enabled (the captured / engine-driven train step, which seeds its backward with :func:`unit_gradient`), the fused
    loss launches the reconstruction term's own gradient chain AHEAD of the prediction: stage 1 of the MSE (which yields
    d loss / d recon for a unit seed), then the producing ``vae_fc4``'s input and weight gradients -- all on the stream ``recon``
    was produced on (the sequence branch's), where they depend on nothing the graph branch computes.  They run beside the EGNN
    forward layers (whose workgroups leave room for them) instead of beside the backward layers (whose do not: DESIGN.md
    section 5.1).  The backward node of ``vae_fc4`` recognises the gradient it was speculated for by its buffer and hands the
    stored results on; any other seed (a scaled loss) makes it launch normally -- the speculation is then wasted, not wrong."""
    enabled = False

    # the caller's promise that the loss consumes the reconstruction through :func:`vae_loss` (``utils.Losses``): only then may the
    # models let the main stream join the sequence branch at the LATENT (models/_core.py EARLY_JOIN) -- a loss that reads ``recon_x``
    # with plain torch ops on the main stream would otherwise have no dependency on the decoder's GEMM (a missing edge
    # inside a captured graph).  engine.CapturedTrainStep passes ``getattr(forward_loss, "fused_loss", False)``.
    early_join = False

    def __init__(self, early_join=False):
        self._early_join = bool(early_join)

    def __enter__(self):
        self._saved = (SpeculativeBackward.enabled, SpeculativeBackward.early_join)
        SpeculativeBackward.enabled = True
        SpeculativeBackward.early_join = self._early_join
        return self

    def __exit__(self, *exc):
        SpeculativeBackward.enabled, SpeculativeBackward.early_join = self._saved
        return False


def _speculate_recon_backward(recon, x, c_mse):
    """-> (d_recon, partials) with stage 1 and the producer's backward launched on the producer's stream, or None"""
    node = recon.grad_fn if torch.is_tensor(recon) else None
    if (not SpeculativeBackward.enabled or node is None or type(node).__name__ != "LinearSmallBatchFnBackward" or not recon.is_cuda
            or recon.dtype != torch.float32 or not recon.is_contiguous() or getattr(node, "fwd_stream", None) is None
            or not torch.is_grad_enabled() or float(c_mse) == 0.0):
        return None
    lib = _lib.load()
    main, side = torch.cuda.current_stream(recon.device), node.fwd_stream
    xs = _lib.f32c(x.reshape(recon.shape))
    with torch.cuda.stream(side):
        d_recon = torch.empty_like(recon)
        partials = torch.empty(lib.is_loss_partials_floats(), dtype=torch.float32, device=recon.device)
        rt = recon.numel()
        _lib.check(lib.is_recon_mse(_lib.ptr(recon), _lib.ptr(xs), _lib.ptr(d_recon), rt, float(c_mse) * 2.0 / rt, _lib.ptr(partials),
                                    _lib.stream_ptr()), "is_recon_mse")
        ready = torch.cuda.Event()
        ready.record()          # the loss (caller's stream) needs stage 1 only, not the gradient chain behind it
        with torch.no_grad():
            gx, dw, db = LinearSmallBatchFn.launch_backward(node, d_recon, node.needs_input_grad[0])
    node.spec = (d_recon, gx, dw, db)
    if side != main:
        main.wait_event(ready)
        for t in (d_recon, partials, gx, dw, db, xs):
            if t is not None:
                t.record_stream(main)
    return d_recon, partials


def vae_loss(recon, x, mu, logvar, logit, y, mode, pos_weight, c_pred, c_mse, c_kld):
    """Returns (total, terms[4] = {total, prediction, recon MSE, KLD})."""
    pre = _speculate_recon_backward(recon, x, c_mse) if recon is not None else None
    if pre is None and torch.is_tensor(recon):
        # no speculation (a reconstruction weight of 0, a non-contiguous or foreign ``recon``): this launch reads ``recon`` on the
        # caller's stream, which under the models' early join has only waited for the latent -- wait for the decoder here
        ev = getattr(recon, "_ready_event", None)
        if ev is not None:
            torch.cuda.current_stream(recon.device).wait_event(ev)
        elif SpeculativeBackward.early_join and recon.is_cuda:
            # a view / slice of the reconstruction lost the attribute (a Python attribute of the tensor OBJECT): wait for everything
            # the sequence branch's stream has been given -- never a missing edge, at worst the full join's cost
            from .models._core import _side_stream
            torch.cuda.current_stream(recon.device).wait_stream(_side_stream(recon.device))
    return VaeLossFn.apply(recon, x, mu, logvar, logit, y, mode, pos_weight, c_pred, c_mse, c_kld, pre)


