"""Attention blocks with the reference's parameter names.

* ``SelfAttention(feature_dim)`` -- parameters ``query/key/value``; single
  head, no output projection, returns ``(out, weights)``
  (reference ``models/layers.py:6-22``).
* ``MultiHeadAttention(feature_dim, n_head, input_dim=None)`` -- parameters
  ``w_q/w_k/w_v/w_concat``; returns ``(out, weights[b, head, n, n])``; no mask is
  ever passed on this path (reference ``models/layers.py:51-106``).

Both evaluate through one shared routine (:func:`attend`).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def attend(q, k, v, heads):
    """softmax(q k^T / sqrt(d_head)) v for (b, n, d) inputs; returns (out (b,n,d), weights (b,heads,n,n)).  The FULL attention
    output from torch ops: what the max-pooling variant (StructureModelv2, ablation_models.py:296-299) and explicit ``forward``
    calls need -- the HIP kernels produce the pooled mean only (``pooled_mean``), which is all the other models read."""
    if q.is_cuda:
        from .. import functional as HF
        HF.composed_path("full (n x d) node attention output (max pooling / explicit forward; kernels: the pooled mean)")
    b, n, d = q.shape
    dh = d // heads
    q, k, v = (t.reshape(b, n, heads, dh).transpose(1, 2) for t in (q, k, v))
    w = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(dh)), dim=-1)
    out = torch.matmul(w, v).transpose(1, 2).reshape(b, n, d)
    return out, w


def attend_pooled_mean(x, wq, bq, wk, bk, wv, bv, heads, need_weights=False, qk=None, out_proj=None):
    """mean over the n rows of softmax(q k^T / sqrt(d)) v  WITHOUT materialising v or the (n, d) output.

    mean_i sum_j A_ij v_j = sum_j abar_j v_j with abar = column mean of A, and
    sum_j abar_j (W_v x_j + b_v) = W_v (sum_j abar_j x_j) + b_v because sum_j abar_j = 1.
    HIP kernels: fused query/key projection (``PairLinearFn``) and scores -> softmax -> column mean ->
    ctx = abar^T x (``AttnColMeanFn``); the value projection acts on one 64-vector per (graph, head).
    Returns (pooled (b, d_model), weights (b, heads, n, n) or None).
    """
    from .. import functional as HF
    b, n, dm = x.shape
    dh = dm // heads
    x2 = x.reshape(b * n, dm)
    if qk is None:     # else: already produced by the EGNN stack's last node kernel (functional.egnn_stack head)
        qk = HF.pair_linear(x2, wq, bq, wk, bk) if dm == 64 else torch.cat([F.linear(x2, wq, bq), F.linear(x2, wk, bk)], dim=1)
    if (heads == 1 and n <= 256 and dm == 64 and out_proj is not None and x.is_cuda
            and not need_weights):
        # scores -> softmax -> column mean -> ctx -> value projection -> w_concat: ONE forward launch
        return HF.attn_pooled_tail(qk, x2, wv, bv, out_proj.weight, out_proj.bias, b, n), None
    if heads in (1, 8) and n <= 256 and dm == 64:
        ctx = HF.attn_colmean(qk, x2, b, n, heads)                             # (b, heads, dm)
        w = None
        if need_weights:
            with torch.no_grad():
                q5 = qk.view(b, n, 2, heads, dh)
                w = torch.softmax(torch.matmul(q5[:, :, 0].transpose(1, 2), q5[:, :, 1].transpose(1, 2).transpose(-1, -2))
                                  * (1.0 / math.sqrt(dh)), dim=-1)
    else:
        HF.composed_path(f"node attention with {heads} head(s) over {n} nodes per graph at width {dm} (kernels: 1 or 8 heads, <= 256 nodes, width 64)")
        q5 = qk.view(b, n, 2, heads, dh)
        q, k = q5[:, :, 0].transpose(1, 2), q5[:, :, 1].transpose(1, 2)
        w = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(dh)), dim=-1)
        ctx = torch.matmul(w.mean(dim=2), x)
    if out_proj is not None and x.is_cuda and dm == 64 and ctx.is_contiguous():
        # per-head value projection and the output projection (w_concat) of the pooled vector in one HIP launch
        u = HF.mlp2(ctx.reshape(b, heads * dm), wv, bv, out_proj.weight, out_proj.bias, hgroup=dh)
        return u, w
    u = torch.einsum("bhk,hdk->bhd", ctx, wv.view(heads, dh, dm)).reshape(b, dm) + bv
    if out_proj is not None:
        u = out_proj(u)
    return u, w


class SelfAttention(nn.Module):
    def __init__(self, feature_dim):
        super().__init__()
        self.query = nn.Linear(feature_dim, feature_dim)
        self.key = nn.Linear(feature_dim, feature_dim)
        self.value = nn.Linear(feature_dim, feature_dim)

    def forward(self, x):
        out, w = attend(self.query(x), self.key(x), self.value(x), 1)
        return out, w.squeeze(1)

    def qk_head(self):
        """(Wq, bq, Wk, bk) when the query / key projection can ride on the EGNN stack's last node kernel"""
        if self.query.in_features != 64 or self.query.out_features != 64:
            return None
        return self.query.weight, self.query.bias, self.key.weight, self.key.bias

    def pooled_mean(self, x, need_weights=False, qk=None):
        """(mean over rows of the attention output, weights) -- what the models feed to global_mean_pool."""
        u, w = attend_pooled_mean(x, self.query.weight, self.query.bias, self.key.weight, self.key.bias,
                                  self.value.weight, self.value.bias, 1, need_weights, qk=qk)
        return u, (w.squeeze(1) if w is not None else None)


class MultiHeadAttention(nn.Module):
    def __init__(self, feature_dim, n_head, input_dim=None):
        super().__init__()
        if feature_dim % n_head != 0:
            raise AssertionError("Embedding dimension must be 0 modulo number of heads.")
        input_dim = input_dim or feature_dim
        self.n_head = n_head
        self.w_q = nn.Linear(input_dim, feature_dim)
        self.w_k = nn.Linear(input_dim, feature_dim)
        self.w_v = nn.Linear(input_dim, feature_dim)
        self.w_concat = nn.Linear(feature_dim, feature_dim)

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("attention masks are never used on this path")
        out, w = attend(self.w_q(x), self.w_k(x), self.w_v(x), self.n_head)
        return self.w_concat(out), w

    def qk_head(self):
        """(Wq, bq, Wk, bk) when the query / key projection can ride on the EGNN stack's last node kernel"""
        if self.w_q.in_features != 64 or self.w_q.out_features != 64:
            return None
        return self.w_q.weight, self.w_q.bias, self.w_k.weight, self.w_k.bias

    def pooled_mean(self, x, need_weights=False, qk=None):
        """mean over rows of ``forward(x)[0]`` (the output projection commutes with the mean)."""
        if self.w_q.in_features != self.w_q.out_features or self.w_q.out_features != 64:
            if x.is_cuda:
                from .. import functional as HF
                HF.composed_path(f"node attention at width {self.w_q.in_features} -> {self.w_q.out_features} (kernels: width 64)")
            out, w = self.forward(x)
            return out.mean(dim=1), w
        return attend_pooled_mean(x, self.w_q.weight, self.w_q.bias, self.w_k.weight, self.w_k.bias,
                                  self.w_v.weight, self.w_v.bias, self.n_head, need_weights, qk=qk, out_proj=self.w_concat)
