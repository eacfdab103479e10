"""Operator API the reference models call, backed by the HIP kernels.

Drop-in replacements (same names, argument meaning and return values) for

* ``dgl.nn.EGNNConv(in_size, hidden_size, out_size, edge_feat_size)`` and its
  ``forward(graph, node_feat, coord_feat, edge_feat) -> (h, x)`` -- reference
  call sites ``models/hybrid_models.py:29-31,89-90,261-263,323-324``;
* ``torch_geometric.nn.global_mean_pool(x, batch)`` / ``global_max_pool`` --
  reference ``models/hybrid_models.py:97,331``, ``ablation_models.py:296-297``.

``EGNNConv`` keeps DGL's parameter layout (``edge_mlp.{0,2}``,
``node_mlp.{0,2}``, ``coord_mlp.{0,2}``) so reference checkpoints load
unchanged.  How the layer is evaluated differs from DGL by design:

  edge_mlp.0 is split as W1 = [W1s | W1d | w_r | W_a]; the node-level
  projections Ps = h W1s^T and Pd = h W1d^T + b1 are computed once per node so
  the per-edge work is a 256-byte row gather + two 64x64 MFMA layers; gather,
  messages, coordinate messages, the sum/mean reductions, the node MLP and the
  NEXT layer's projections run in ONE fused kernel per layer
  (``csrc/egnn_layer_fwd.hip``).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as HF
from .graph import PackedGraphBatch


class EGNNConv(nn.Module):
    def __init__(self, in_size, hidden_size, out_size, edge_feat_size=0):
        super().__init__()
        # the HIP layer kernels are built for the reference's sizes (hidden = out = 64 = gat_hidden_channels' default,
        # models/hybrid_models.py:247; 20 or 64 input features; <= 8 edge features).  Any other size still runs -- on the device,
        # composed from torch ops with the same fixed summation order (egnn_conv_composed below) -- so that every constructor
        # argument of the reference's models is accepted; it is a slow path and cannot be captured by engine.CapturedTrainStep
        self.native = hidden_size == HF.HIDDEN and out_size == HF.HIDDEN and in_size in (20, HF.HIDDEN) and edge_feat_size <= 8
        self.in_size, self.hidden_size = in_size, hidden_size
        self.out_size, self.edge_feat_size = out_size, edge_feat_size
        act = nn.SiLU()
        self.edge_mlp = nn.Sequential(
            nn.Linear(in_size * 2 + edge_feat_size + 1, hidden_size), act,
            nn.Linear(hidden_size, hidden_size), act)
        self.node_mlp = nn.Sequential(
            nn.Linear(in_size + hidden_size, hidden_size), act,
            nn.Linear(hidden_size, out_size))
        self.coord_mlp = nn.Sequential(
            nn.Linear(hidden_size, hidden_size), act,
            nn.Linear(hidden_size, 1, bias=False))

    def native_parameters(self):
        """The 11 parameter tensors in the order ``functional.EGNNStackFn`` expects."""
        return [self.edge_mlp[0].weight, self.edge_mlp[0].bias, self.edge_mlp[2].weight, self.edge_mlp[2].bias,
                self.node_mlp[0].weight, self.node_mlp[0].bias, self.node_mlp[2].weight, self.node_mlp[2].bias,
                self.coord_mlp[0].weight, self.coord_mlp[0].bias, self.coord_mlp[2].weight]

    def forward(self, graph, node_feat, coord_feat, edge_feat=None):
        h, x = egnn_stack_forward([self], graph, node_feat, coord_feat, edge_feat)
        return h, x


def egnn_stack_prologue(layers, node_feat, coord_feat, head=None):
    """Launch the first kernel of :func:`egnn_stack_forward` (layer-0 pre-projection, operand packs, dense coordinates) now and
    return the handle to pass as ``prologue=`` -- for callers that fork other work onto a side stream before the stack."""
    return HF.launch_stack_prologue(node_feat, coord_feat, [layer.native_parameters() for layer in layers], head=head)


def egnn_stack_prelaunch(layers, graph, node_feat, coord_feat, edge_feat=None, head=None, final_coords=True, fork_after=None):
    """Enqueue the whole forward of :func:`egnn_stack_forward` now, outside autograd, and return the handle to pass as
    ``prologue=`` to the later call with the same arguments (which then only creates the autograd node).  ``fork_after`` = i
    leaves an event recorded behind layer i's launch in ``.fork_event``."""
    _check_stack(layers, graph, edge_feat)
    ea = graph.edge_feat_csr(edge_feat) if layers[0].edge_feat_size > 0 else None
    return HF.launch_stack_forward(node_feat, coord_feat, ea, graph.csr(), [layer.native_parameters() for layer in layers],
                                   head=head, final_coords=final_coords, fork_after=fork_after)


def stack_is_native(layers):
    """True when consecutive layers can run as the fused HIP stack (sizes the kernels are built for, one edge_feat_size)"""
    fe = layers[0].edge_feat_size
    return all(layer.native and layer.edge_feat_size == fe and (i == 0 or layer.in_size == HF.HIDDEN) for i, layer in enumerate(layers))


def egnn_conv_composed(layer, csr, num_edges, h, x, ea_csr, want_coords=True):
    """One EGNNConv layer from device-side torch ops, for sizes the HIP kernels are not built for: the arithmetic of
    ``dgl.nn.EGNNConv`` (x_src - x_dst, squared distance, / (sqrt + 1e-30), [h_src, h_dst, radial, a], sum for h, mean for x) over
    the destination-sorted edge slots, the two aggregations as segment sums over the CSR row pointer (fixed order, no atomics)."""
    src, dst = csr.src_sorted[:num_edges].long(), csr.dst_sorted[:num_edges].long()
    offsets = csr.rowptr_dst.long()
    d = x[src] - x[dst]
    radial = (d * d).sum(dim=1, keepdim=True)
    feats = [h[src], h[dst], radial] + ([ea_csr[:num_edges]] if layer.edge_feat_size > 0 else [])
    m = layer.edge_mlp(torch.cat(feats, dim=1))
    h_neigh = torch.segment_reduce(m, "sum", offsets=offsets, axis=0)
    h_out = layer.node_mlp(torch.cat([h, h_neigh], dim=1))
    if not want_coords:
        return h_out, None
    msg_x = layer.coord_mlp(m) * (d / (radial.sqrt() + 1e-30))
    deg = (offsets[1:] - offsets[:-1]).clamp(min=1).to(x.dtype).unsqueeze(1)
    return h_out, x + torch.segment_reduce(msg_x, "sum", offsets=offsets, axis=0) / deg


def _check_stack(layers, graph, edge_feat):
    if not isinstance(graph, PackedGraphBatch):
        raise TypeError("immunostruct_amd.nn.EGNNConv expects an immunostruct_amd.graph.PackedGraphBatch")
    fe = layers[0].edge_feat_size
    if fe > 0 and edge_feat is None:
        raise ValueError("Edge features must be provided.")
    if edge_feat is not None and edge_feat.requires_grad:
        raise NotImplementedError("gradients w.r.t. edge features are not produced by the HIP kernel")


def egnn_stack_forward(layers, graph, node_feat, coord_feat, edge_feat=None, head=None, final_coords=True, prologue=None):
    """Run consecutive :class:`EGNNConv` layers as one fused HIP stack (what the models do with ``GCN_layers``).

    ``head`` = optional (Wa, ba, Wb, bb): also return the 128-wide projection [h Wa^T + ba | h Wb^T + bb] of the
    final node features (the node attention's query / key projection), computed by the last layer's node kernel.
    ``final_coords=False``: the caller ignores the last layer's coordinates; its coordinate MLP is then skipped and the
    returned x may be None."""
    _check_stack(layers, graph, edge_feat)
    ea = graph.edge_feat_csr(edge_feat) if layers[0].edge_feat_size > 0 else None
    if not stack_is_native(layers):
        if getattr(graph, "edge_capacity", None) is not None:
            raise NotImplementedError("EGNN layers of sizes the HIP kernels are not built for run from torch ops and cannot be "
                                      "captured into a HIP graph (engine.CapturedTrainStep): train them with the eager loops")
        HF._lib.require_device(node_feat, coord_feat)
        HF.composed_path(f"EGNN layers of sizes {[(l.in_size, l.hidden_size, l.out_size, l.edge_feat_size) for l in layers][:2]}... "
                         "(kernels: hidden = out = 64, 20 or 64 inputs, <= 8 edge features)")
        h, x = node_feat, coord_feat
        csr, e = graph.csr(), graph.num_edges()
        for i, layer in enumerate(layers):
            eai = graph.edge_feat_csr(edge_feat) if layer.edge_feat_size > 0 else None
            h, x = egnn_conv_composed(layer, csr, e, h, x, eai, want_coords=final_coords or i + 1 < len(layers))
        if head is not None:
            wa, ba, wb, bb = head
            return h, x, torch.cat([torch.nn.functional.linear(h, wa, ba), torch.nn.functional.linear(h, wb, bb)], dim=1)
        return h, x
    return HF.egnn_stack(node_feat, coord_feat, ea, graph.csr(), [layer.native_parameters() for layer in layers], head=head,
                         final_coords=final_coords, prologue=prologue)


def _seg_ptr_from_batch(batch_index, size=None):
    n = int(batch_index.max()) + 1 if size is None else int(size)
    counts = torch.bincount(batch_index, minlength=n)
    ptr = torch.zeros(n + 1, dtype=torch.int32, device=batch_index.device)
    ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return ptr


def global_mean_pool(x, batch, size=None):
    """PyG signature; ``batch`` must be sorted (block-diagonal batches always are)."""
    return HF.segment_pool(x, _seg_ptr_from_batch(batch, size), "mean")


def global_max_pool(x, batch, size=None):
    return HF.segment_pool(x, _seg_ptr_from_batch(batch, size), "max")
