"""``torch.autograd.Function`` wrappers over the C-ABI kernels.

Each Function validates shapes/dtypes in Python (raising ``ValueError`` the
way torch would), allocates outputs with torch (device memory + stream are the
only things torch provides here) and enqueues the HIP kernels on the current
stream.  There is no fallback path: CPU tensors raise ``HipExtensionError``.
"""
from __future__ import annotations

import torch

from . import _lib

HIDDEN = 64
_MAX_BWD_GRID = 256  # one persistent workgroup per CU (MI355X: 256 CUs)
_NODES_PER_TILE = 32


class KernelTimer:
    """Optional HIP-event bracketing of individual kernel launches (used by bench.py for the roofline).

    Events are recorded on the stream the kernel is launched on (torch's current stream).
    """
    enabled = False
    records = {}

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
    def reset(cls):
        cls.records = {}

    @classmethod
    def summary(cls):
        """name -> (launches, mean milliseconds); call after torch.cuda.synchronize()."""
        out = {}
        for name, evs in cls.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = (len(ms), sum(ms) / max(len(ms), 1))
        return out


class EGNNEdgeFn(torch.autograd.Function):
    """Fused edge pass of one EGNNConv layer (``csrc/egnn_edge_fwd.hip`` / ``_bwd.hip``).

    inputs : psd (N,128) = [h W1s^T | h W1d^T + b1], x (N,3), ea (E,Fe) in CSR slot
             order, w_r (64,), w_a (64,Fe), W2 (64,64), b2, Wc1 (64,64), bc1, wc2 (64,)
    outputs: h_neigh (N,64) = sum of messages, x_out (N,3) = x + mean coordinate message
    """

    @staticmethod
    def forward(ctx, psd, x, ea, w_r, w_a, W2, b2, Wc1, bc1, wc2, csr):
        lib = _lib.load()
        _lib.require_device(psd, x, ea, w_r, w_a, W2, b2, Wc1, bc1, wc2, csr.rowptr_dst)
        n, e = csr.num_nodes, csr.num_edges
        if psd.shape != (n, 2 * HIDDEN):
            raise ValueError(f"psd must be ({n}, {2 * HIDDEN}), got {tuple(psd.shape)}")
        if x.shape != (n, 3):
            raise ValueError(f"coordinates must be ({n}, 3), got {tuple(x.shape)}")
        fe = int(ea.shape[1]) if ea is not None and ea.dim() == 2 else 0
        if fe > 8:
            raise ValueError("edge_feat_size > 8 is not supported by the HIP kernel")
        if fe and ea.shape[0] != e:
            raise ValueError("edge feature rows must equal the number of edges")
        psd, ld_p = _lib.rows_ld(psd)
        x = _lib.f32c(x)
        ea = _lib.f32c(ea) if fe else None
        w_r, W2, b2, Wc1, bc1, wc2 = (_lib.f32c(t) for t in (w_r, W2, b2, Wc1, bc1, wc2))
        w_a = _lib.f32c(w_a) if fe else None
        need_grad = any(ctx.needs_input_grad)
        h_neigh = torch.empty(n, HIDDEN, dtype=torch.float32, device=x.device)
        x_out = torch.empty(n, 3, dtype=torch.float32, device=x.device)
        z2s = torch.empty(max(e, 1), HIDDEN, dtype=torch.float32, device=x.device) if need_grad else None
        z3s = torch.empty(max(e, 1), HIDDEN, dtype=torch.float32, device=x.device) if need_grad else None
        pd_view = psd[:, HIDDEN:]
        with KernelTimer.span("egnn_edge_fwd"):
            code = lib.is_egnn_edge_fwd(
                _lib.ptr(psd), _lib.ptr(pd_view), ld_p, _lib.ptr(x), _lib.ptr(ea),
                _lib.ptr(csr.rowptr_dst), _lib.ptr(csr.src_sorted), _lib.ptr(w_r), _lib.ptr(w_a),
                _lib.ptr(W2), _lib.ptr(b2), _lib.ptr(Wc1), _lib.ptr(bc1), _lib.ptr(wc2),
                _lib.ptr(h_neigh), HIDDEN, _lib.ptr(x_out), _lib.ptr(z2s), _lib.ptr(z3s), n, fe, _lib.stream_ptr())
        _lib.check(code, "is_egnn_edge_fwd")
        ctx.csr, ctx.fe, ctx.ld_p = csr, fe, ld_p
        ctx.save_for_backward(psd, x, ea, w_r, w_a, W2, Wc1, wc2, z2s, z3s)
        return h_neigh, x_out

    @staticmethod
    def backward(ctx, g_hn, g_xout):
        lib = _lib.load()
        psd, x, ea, w_r, w_a, W2, Wc1, wc2, z2s, z3s = ctx.saved_tensors
        csr, fe, ld_p = ctx.csr, ctx.fe, ctx.ld_p
        n, e = csr.num_nodes, csr.num_edges
        dev = x.device
        if g_hn is None:
            g_hn = torch.zeros(n, HIDDEN, dtype=torch.float32, device=dev)
        if g_xout is None:
            g_xout = torch.zeros(n, 3, dtype=torch.float32, device=dev)
        g_hn, ld_ghn = _lib.rows_ld(g_hn)
        g_xout = _lib.f32c(g_xout)
        dZ1 = torch.empty(max(e, 1), HIDDEN, dtype=torch.float32, device=dev)
        dD = torch.empty(max(e, 1), 3, dtype=torch.float32, device=dev)
        dpsd = torch.empty(n, 2 * HIDDEN, dtype=torch.float32, device=dev)
        dx = torch.empty(n, 3, dtype=torch.float32, device=dev)
        grid = max(1, min(_MAX_BWD_GRID, (n + _NODES_PER_TILE - 1) // _NODES_PER_TILE))
        partials = torch.empty(lib.is_egnn_edge_bwd_partials_floats(grid), dtype=torch.float32, device=dev)
        gW2 = torch.empty_like(W2)
        gWc1 = torch.empty_like(Wc1)
        gb2 = torch.empty(HIDDEN, dtype=torch.float32, device=dev)
        gbc1 = torch.empty_like(gb2)
        gwc2 = torch.empty_like(gb2)
        gw_r = torch.empty_like(gb2)
        gw_a = torch.zeros(HIDDEN, max(fe, 1), dtype=torch.float32, device=dev)
        st = _lib.stream_ptr()
        with KernelTimer.span("egnn_edge_bwd"):
            code = lib.is_egnn_edge_bwd(
                _lib.ptr(psd), _lib.ptr(psd[:, HIDDEN:]), ld_p, _lib.ptr(x), _lib.ptr(ea),
                _lib.ptr(csr.rowptr_dst), _lib.ptr(csr.src_sorted), _lib.ptr(w_r), _lib.ptr(w_a),
                _lib.ptr(W2), _lib.ptr(Wc1), _lib.ptr(wc2), _lib.ptr(z2s), _lib.ptr(z3s),
                _lib.ptr(g_hn), ld_ghn, _lib.ptr(g_xout), _lib.ptr(dZ1), _lib.ptr(dD),
                _lib.ptr(dpsd[:, HIDDEN:]), 2 * HIDDEN, _lib.ptr(dx), _lib.ptr(partials), grid,
                _lib.ptr(gW2), _lib.ptr(gWc1), _lib.ptr(gb2), _lib.ptr(gbc1), _lib.ptr(gwc2), _lib.ptr(gw_r),
                _lib.ptr(gw_a), n, fe, st)
        _lib.check(code, "is_egnn_edge_bwd")
        # source-side scatter-add as a CSR-by-source gather: dPs[u] = sum dz1, dx[u] += sum dD
        with KernelTimer.span("gather_segment_sum"):
            code = lib.is_gather_segment_sum(_lib.ptr(dZ1), _lib.ptr(dD), _lib.ptr(csr.rowptr_src),
                                             _lib.ptr(csr.pos_by_src), _lib.ptr(dpsd), 2 * HIDDEN, _lib.ptr(dx), n, st)
        _lib.check(code, "is_gather_segment_sum")
        g_wa = gw_a if fe else None
        # inputs: psd, x, ea, w_r, w_a, W2, b2, Wc1, bc1, wc2, csr
        return dpsd, dx, None, gw_r, g_wa, gW2, gb2, gWc1, gbc1, gwc2, None


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


def egnn_edge(psd, x, ea, w_r, w_a, W2, b2, Wc1, bc1, wc2, csr):
    return EGNNEdgeFn.apply(psd, x, ea, w_r, w_a, W2, b2, Wc1, bc1, wc2, csr)


def segment_pool(x, seg_ptr, mode="mean"):
    return SegmentPoolFn.apply(x, seg_ptr, mode)


class VaeLossFn(torch.autograd.Function):
    """Fused prediction + reconstruction + KLD loss (``csrc/losses.hip``).

    Value and all input gradients are produced by the forward launch; backward
    scales them by the upstream scalar.
    """

    @staticmethod
    def forward(ctx, recon, x, mu, logvar, logit, y, mode, pos_weight, c_pred, c_mse, c_kld):
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
            d_recon, d_mu, d_lv = torch.empty_like(recon_c), torch.empty_like(mu_c), torch.empty_like(lv_c)
            rt, lt = recon_c.numel(), mu_c.numel()
        else:
            recon_c = x_c = mu_c = lv_c = d_recon = d_mu = d_lv = None
            rt = lt = 0
        d_logit = torch.empty(b, dtype=torch.float32, device=dev)
        partials = torch.empty(lib.is_loss_partials_floats(), dtype=torch.float32, device=dev)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        code = lib.is_vae_loss(_lib.ptr(recon_c), _lib.ptr(x_c), _lib.ptr(d_recon), rt, _lib.ptr(mu_c), _lib.ptr(lv_c),
                               _lib.ptr(d_mu), _lib.ptr(d_lv), lt, _lib.ptr(logit_c), _lib.ptr(y_c), _lib.ptr(d_logit),
                               b, int(mode), float(pos_weight), float(c_pred), float(c_mse), float(c_kld),
                               _lib.ptr(partials), _lib.ptr(out), _lib.stream_ptr())
        _lib.check(code, "is_vae_loss")
        ctx.seq = seq
        ctx.logit_shape = logit.shape
        ctx.save_for_backward(d_recon, d_mu, d_lv, d_logit)
        total = out[0].clone()
        ctx.mark_non_differentiable(out)
        return total, out

    @staticmethod
    def backward(ctx, g, _g_terms):
        d_recon, d_mu, d_lv, d_logit = ctx.saved_tensors
        gr = d_recon * g if ctx.seq else None
        gm = d_mu * g if ctx.seq else None
        gl = d_lv * g if ctx.seq else None
        return gr, None, gm, gl, (d_logit * g).reshape(ctx.logit_shape), None, None, None, None, None, None


def vae_loss(recon, x, mu, logvar, logit, y, mode, pos_weight, c_pred, c_mse, c_kld):
    """Returns (total, terms[4] = {total, prediction, recon MSE, KLD})."""
    return VaeLossFn.apply(recon, x, mu, logvar, logit, y, mode, pos_weight, c_pred, c_mse, c_kld)
