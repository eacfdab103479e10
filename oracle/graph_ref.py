"""Oracle restatement of the third-party graph operators the hot path calls.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Un-fused, op-for-op
PyTorch on whatever dtype the inputs carry (fp32 / fp64), CPU.

What is restated and where the reference calls it
-------------------------------------------------
* ``dgl.graph`` / ``DGLGraph`` attribute surface used by the models:
  ``ndata['x']``, ``edata['edge_attr']``, ``batch_num_nodes()``, ``.device``,
  ``.to()``  -- reference ``models/hybrid_models.py:316-321``,
  ``data/utils.py:54-67``.
* ``dgl.batch``: concatenate node/edge frames, offset edge endpoints by the
  running node count, keep per-graph node counts -- reference
  ``data/utils.py:160-176``.
* ``dgl.nn.EGNNConv`` (DGL 2.x, not vendored; parity unpinned): call sites
  ``models/hybrid_models.py:29-31,89-90,261-263,323-324``.  Algorithm
  (Satorras et al. 2021, eqs. 3-6, with DGL's concrete choices):

      per edge u->v:  x_diff = x_u - x_v ; radial = |x_diff|^2
                      x_diff = x_diff / (sqrt(radial) + 1e-30)
                      f      = [h_u , h_v , radial , a_uv]
                      msg_h  = SiLU(W2 SiLU(W1 f + b1) + b2)
                      msg_x  = (wc2 . SiLU(Wc1 msg_h + bc1)) * x_diff
      per node v:     h_neigh = sum_in msg_h ; x_neigh = mean_in msg_x (0 if
                      in-degree 0)
                      h' = Wn2 SiLU(Wn1 [h , h_neigh] + bn1) + bn2
                      x' = x + x_neigh

  state-dict layout ``edge_mlp.{0,2}``, ``node_mlp.{0,2}``,
  ``coord_mlp.{0,2}`` (``coord_mlp.2`` has no bias).
* ``torch_geometric.nn.global_mean_pool`` / ``global_max_pool``
  (torch_geometric==2.5.3, not vendored): ``scatter(x, batch, reduce=...)``
  over dim 0 -- reference ``models/hybrid_models.py:97,331``,
  ``models/ablation_models.py:296-297``.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class RefGraph:
    """Minimal stand-in for a (batched) DGLGraph: COO edges + feature frames."""

    def __init__(self, src, dst, num_nodes, batch_num_nodes=None):
        self._src = torch.as_tensor(src, dtype=torch.int64)
        self._dst = torch.as_tensor(dst, dtype=torch.int64)
        self._num_nodes = int(num_nodes)
        self.ndata = {}
        self.edata = {}
        if batch_num_nodes is None:
            batch_num_nodes = [self._num_nodes]
        self._batch_num_nodes = torch.as_tensor(batch_num_nodes, dtype=torch.int64)

    # -- DGLGraph surface the reference touches ---------------------------
    def edges(self):
        return self._src, self._dst

    def num_nodes(self):
        return self._num_nodes

    def num_edges(self):
        return int(self._src.numel())

    def batch_num_nodes(self):
        return self._batch_num_nodes

    @property
    def batch_size(self):
        return int(self._batch_num_nodes.numel())

    @property
    def device(self):
        return self._src.device

    def to(self, device):
        g = RefGraph(self._src.to(device), self._dst.to(device), self._num_nodes,
                     self._batch_num_nodes.to(device))
        g.ndata = {k: v.to(device) for k, v in self.ndata.items()}
        g.edata = {k: v.to(device) for k, v in self.edata.items()}
        return g


def graph(edges, num_nodes=None):
    """``dgl.graph((src, dst), num_nodes=n)`` (reference ``data/utils.py:64``)."""
    src, dst = edges
    src = torch.as_tensor(src, dtype=torch.int64)
    dst = torch.as_tensor(dst, dtype=torch.int64)
    if num_nodes is None:
        num_nodes = int(max(src.max(), dst.max())) + 1 if src.numel() else 0
    return RefGraph(src, dst, num_nodes)


def batch(graphs):
    """``dgl.batch``: block-diagonal union (reference ``data/utils.py:163``)."""
    srcs, dsts, counts = [], [], []
    offset = 0
    for g in graphs:
        s, d = g.edges()
        srcs.append(s + offset)
        dsts.append(d + offset)
        for n in g.batch_num_nodes().tolist():
            counts.append(n)
        offset += g.num_nodes()
    out = RefGraph(torch.cat(srcs) if srcs else torch.zeros(0, dtype=torch.int64),
                   torch.cat(dsts) if dsts else torch.zeros(0, dtype=torch.int64),
                   offset, counts)
    for key in graphs[0].ndata:
        out.ndata[key] = torch.cat([g.ndata[key] for g in graphs], dim=0)
    for key in graphs[0].edata:
        out.edata[key] = torch.cat([g.edata[key] for g in graphs], dim=0)
    return out


def egnn_conv(params, prefix, src, dst, num_nodes, h, x, a=None):
    """One EGNNConv layer, functional form (weights from a state-dict).

    ``params[prefix + 'edge_mlp.0.weight']`` etc.  Returns ``(h', x')``.
    """
    def lin(name, t, bias=True):
        w = params[prefix + name + ".weight"]
        b = params[prefix + name + ".bias"] if bias else None
        return torch.nn.functional.linear(t, w, b)

    silu = torch.nn.functional.silu
    x_diff = x[src] - x[dst]
    radial = x_diff.square().sum(dim=1, keepdim=True)
    x_diff = x_diff / (radial.sqrt() + 1e-30)
    pieces = [h[src], h[dst], radial]
    if a is not None:
        pieces.append(a)
    f = torch.cat(pieces, dim=-1)
    msg_h = silu(lin("edge_mlp.2", silu(lin("edge_mlp.0", f))))
    msg_x = lin("coord_mlp.2", silu(lin("coord_mlp.0", msg_h)), bias=False) * x_diff

    h_neigh = torch.zeros(num_nodes, msg_h.shape[1], dtype=h.dtype, device=h.device)
    h_neigh = h_neigh.index_add(0, dst, msg_h)
    x_sum = torch.zeros(num_nodes, x.shape[1], dtype=x.dtype, device=x.device)
    x_sum = x_sum.index_add(0, dst, msg_x)
    deg = torch.bincount(dst, minlength=num_nodes).clamp(min=1).to(x.dtype)
    x_neigh = x_sum / deg.unsqueeze(1)

    h_out = lin("node_mlp.2", silu(lin("node_mlp.0", torch.cat([h, h_neigh], dim=-1))))
    return h_out, x + x_neigh


class EGNNConvRef(nn.Module):
    """Module form with DGL's parameter names; forwards to :func:`egnn_conv`."""

    def __init__(self, in_size, hidden_size, out_size, edge_feat_size=0):
        super().__init__()
        self.in_size, self.hidden_size = in_size, hidden_size
        self.out_size, self.edge_feat_size = out_size, edge_feat_size
        act = nn.SiLU()
        self.edge_mlp = nn.Sequential(
            nn.Linear(2 * in_size + edge_feat_size + 1, hidden_size), act,
            nn.Linear(hidden_size, hidden_size), act)
        self.node_mlp = nn.Sequential(
            nn.Linear(in_size + hidden_size, hidden_size), act,
            nn.Linear(hidden_size, out_size))
        self.coord_mlp = nn.Sequential(
            nn.Linear(hidden_size, hidden_size), act,
            nn.Linear(hidden_size, 1, bias=False))

    def forward(self, graph, node_feat, coord_feat, edge_feat=None):
        src, dst = graph.edges()
        if self.edge_feat_size == 0:
            edge_feat = None
        params = dict(self.named_parameters())
        return egnn_conv(params, "", src, dst, graph.num_nodes(), node_feat, coord_feat, edge_feat)


def _num_segments(index):
    return int(index.max()) + 1 if index.numel() else 0


def global_mean_pool(x, batch_index, size=None):
    """PyG 2.5.3 ``scatter(x, batch, dim=0, reduce='mean')``: sum / clamp(count, 1)."""
    n = _num_segments(batch_index) if size is None else size
    total = torch.zeros(n, x.shape[1], dtype=x.dtype, device=x.device).index_add(0, batch_index, x)
    count = torch.bincount(batch_index, minlength=n).clamp(min=1).to(x.dtype)
    return total / count.unsqueeze(1)


def global_max_pool(x, batch_index, size=None):
    """PyG 2.5.3 ``scatter(x, batch, dim=0, reduce='max')``."""
    n = _num_segments(batch_index) if size is None else size
    out = torch.full((n, x.shape[1]), float("-inf"), dtype=x.dtype, device=x.device)
    idx = batch_index.unsqueeze(1).expand_as(x)
    out = out.scatter_reduce(0, idx, x, reduce="amax", include_self=True)
    return torch.where(torch.isinf(out), torch.zeros_like(out), out)
