"""Batched residue-graph container with the DGLGraph surface the reference uses.

The reference hands the models a batched ``DGLGraph`` (``dgl.batch`` in
``data/utils.py:160-176``) and only touches ``ndata['x']``,
``edata['edge_attr']``, ``batch_num_nodes()``, ``.device`` and ``.to()``
(``models/hybrid_models.py:316-321``).  ``PackedGraphBatch`` offers exactly that
surface plus what the HIP kernels need:

* CSR by destination: ``rowptr_dst`` (N+1), ``src_sorted`` (E), ``eperm`` (E,
  original edge id of each CSR slot; stable sort, so a node's in-edges keep
  their original relative order and the segment sums have a fixed order);
* CSR by source over the *CSR-by-destination slots*: ``rowptr_src`` (N+1),
  ``pos_by_src`` (E) -- used by the backward gather that replaces the
  scatter-add to source rows;
* ``seg_ptr`` (B+1): node offsets of the graphs (block-diagonal batch).

All indices are int32 (N, E < 2^31).  Index construction is plain torch
(sort / bincount / cumsum) and runs on whichever device the graph lives on, so
it can be done once per graph in DataLoader workers or on the GPU.
"""
from __future__ import annotations

import torch


class CSRIndex:
    __slots__ = ("rowptr_dst", "src_sorted", "dst_sorted", "eperm", "rowptr_src", "pos_by_src", "num_nodes", "num_edges",
                 "_chunks")

    def __init__(self, src, dst, num_nodes):
        src = src.long()
        dst = dst.long()
        n, e = int(num_nodes), int(src.numel())
        order = torch.argsort(dst, stable=True)
        src_sorted = src[order]
        self.eperm = order
        self.src_sorted = src_sorted.to(torch.int32)
        self.dst_sorted = dst[order].to(torch.int32)
        self.rowptr_dst = _rowptr(dst, n)
        order2 = torch.argsort(src_sorted, stable=True)
        self.pos_by_src = order2.to(torch.int32)
        self.rowptr_src = _rowptr(src, n)
        self.num_nodes, self.num_edges = n, e
        self._chunks = {}

    def to(self, device):
        out = object.__new__(CSRIndex)
        for k in ("rowptr_dst", "src_sorted", "dst_sorted", "eperm", "rowptr_src", "pos_by_src"):
            setattr(out, k, getattr(self, k).to(device))
        out.num_nodes, out.num_edges = self.num_nodes, self.num_edges
        out._chunks = {k: v.to(device) for k, v in self._chunks.items()}
        return out

    def chunks(self, k):
        """``chunk_ptr`` (k+1, 2) int32 = (node boundary b_j, rowptr[b_j]): the destination nodes cut into k contiguous
        ranges with (nearly) equal in-edge counts -- the unit of work of one wave of the v3 edge kernels."""
        k = int(k)
        if k not in self._chunks:
            self._chunks[k] = balanced_node_chunks(self.rowptr_dst, k)
        return self._chunks[k]


# Shares of the edges in the forward layer kernel's chunk partition (chunk c = wave c % 4 of workgroup c // 4).
# A wave walks its chunk in 16-edge tiles, and the kernel is issue-bound with two waves per SIMD: what counts is the number of
# tiles per SIMD.  At full residency (512 workgroups = 2 per CU) the hardware places workgroups i and i + 256 on the same CU
# (measured: profiles/r03_*), so their waves share the SIMDs.  With P = ceil(edges per wave pair / 16) ODD, equal shares make
# both waves walk (P + 1) / 2 tiles -- P + 1 per SIMD -- while shares (P + 1) / 2 : (P - 1) / 2 for the first / second half of the
# workgroups make it P (B = 128, E = 72 k: 35 + 35 edges = 3 + 3 tiles -> 42 + 28 edges = 3 + 2 tiles; forward layer launch 48 ->
# 43.5 us).  Used only when the chunks are node-aligned with some slack (>= 6 edges per pair) and the grid is the full one.
FULL_GRID_CHUNKS = 2048
CHUNK_SHARES = "auto"      # "flat": equal shares everywhere (module constant: the chunk-partition tests compare both)


NODE_PASS_ROWS = 64      # rows of one pass of the forward kernel's node half (csrc/egnn_layer_fwd.hip ROWS)


def chunk_shares(e, k, n=None):
    """(wa, wb): weight of a chunk of the first / second half of the workgroups; ``e`` a 0-d / 1-element integer tensor (no sync),
    ``n`` the number of nodes.  Uneven shares (P + 1) / 2 : (P - 1) / 2 when P = ceil(edges per wave pair / 16) is odd and the
    node-aligned chunks have slack -- and only while the LARGER workgroups' nodes do not need one more 64-row pass of the node
    half than equal shares would (average nodes per workgroup + a margin of 4: a 2 : 1 split of 47.5 nodes per workgroup puts
    63 on the first half of the workgroups, and half of those then run a second node pass).
    The device kernel (csrc/segment_ops.hip ``chunk_partition_kernel``) evaluates the same integer rule."""
    one = torch.ones_like(e)
    if CHUNK_SHARES != "auto" or k != FULL_GRID_CHUNKS:
        return one, one
    p = (2 * e + 16 * k - 1) // (16 * k)
    use = (p % 2 == 1) & (p >= 3) & (p * 16 * k - 2 * e >= 6 * k)
    if n is not None:
        big = (8 * int(n) * ((p + 1) // 2) + p * k - 1) // (p * k)          # ceil of the larger workgroups' average node count
        flat = (4 * int(n) + k - 1) // k
        use = use & ((big + 4 + NODE_PASS_ROWS - 1) // NODE_PASS_ROWS == (flat + 4 + NODE_PASS_ROWS - 1) // NODE_PASS_ROWS)
    return torch.where(use, (p + 1) // 2, one), torch.where(use, (p - 1) // 2, one)


def balanced_node_chunks(rowptr, k):
    """Rows (b_j, rowptr[b_j]) of the boundaries b_0 = 0 <= b_1 <= ... <= b_k = N with b_j = first node whose first in-edge index is
    >= E * W(j) / W(k), W = running weight of the chunks (equal shares: j * E / k; see ``chunk_shares``): node-aligned chunks of
    about E/k edges (a node of very high degree leaves its neighbours' chunks short or empty; nodes without in-edges ride
    along with the following node)."""
    n = rowptr.numel() - 1
    rp = rowptr.long()
    e = rp[-1:]                                                   # stays on the device: no sync
    wa, wb = chunk_shares(e, k, n)
    half = k // 2
    j = torch.arange(k + 1, device=rowptr.device, dtype=torch.int64)
    wj = wa * torch.clamp(j, max=half) + wb * torch.clamp(j - half, min=0)
    targets = (wj * e) // (wa * half + wb * (k - half))
    b = torch.searchsorted(rp, targets, right=False)
    b[0] = 0
    b[-1] = n
    b = torch.clamp(b, max=n)
    # (node boundary, first edge of that node) pairs: the kernel gets a chunk's node AND edge range with one load level
    return torch.stack([b, rp[b]], dim=1).to(torch.int32).contiguous()


def _rowptr(index, n):
    counts = torch.bincount(index, minlength=n)
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=index.device)
    ptr[1:] = torch.cumsum(counts, 0)
    return ptr.to(torch.int32)


class PackedGraphBatch:
    """A (batched) directed graph in COO form with feature frames and cached CSR indices."""

    def __init__(self, src, dst, num_nodes, batch_num_nodes=None):
        self._src = torch.as_tensor(src).long()
        self._dst = torch.as_tensor(dst).long()
        self._num_nodes = int(num_nodes)
        if batch_num_nodes is None:
            batch_num_nodes = [self._num_nodes]
        self._counts = [int(c) for c in (batch_num_nodes.tolist() if torch.is_tensor(batch_num_nodes)
                                         else list(batch_num_nodes))]
        if sum(self._counts) != self._num_nodes:
            raise ValueError("batch_num_nodes does not sum to num_nodes")
        self.ndata = {}
        self.edata = {}
        self._csr = None
        self._seg_ptr = None
        self._sorted_cache = {}

    # ---- DGLGraph surface ------------------------------------------------
    def edges(self):
        return self._src, self._dst

    def num_nodes(self):
        return self._num_nodes

    def num_edges(self):
        return int(self._src.numel())

    def batch_num_nodes(self):
        return torch.tensor(self._counts, dtype=torch.int64, device=self.device)

    @property
    def batch_size(self):
        return len(self._counts)

    @property
    def device(self):
        return self._src.device

    def to(self, device, non_blocking=False):
        device = torch.device(device)
        if device == self.device:
            return self
        g = PackedGraphBatch(self._src.to(device, non_blocking=non_blocking),
                             self._dst.to(device, non_blocking=non_blocking), self._num_nodes, self._counts)
        g.ndata = {k: v.to(device, non_blocking=non_blocking) for k, v in self.ndata.items()}
        g.edata = {k: v.to(device, non_blocking=non_blocking) for k, v in self.edata.items()}
        if self._csr is not None:
            g._csr = self._csr.to(device)
        return g

    # ---- kernel-side indices ----------------------------------------------
    def uniform_nodes_per_graph(self):
        """n if every graph has the same (padded) node count, else None (F5 in SURVEY.md)."""
        first = self._counts[0] if self._counts else 0
        return first if all(c == first for c in self._counts) else None

    def csr(self) -> CSRIndex:
        if self._csr is None:
            self._csr = CSRIndex(self._src, self._dst, self._num_nodes)
        return self._csr

    def seg_ptr(self):
        if self._seg_ptr is None or self._seg_ptr.device != self.device:
            ptr = [0]
            for c in self._counts:
                ptr.append(ptr[-1] + c)
            self._seg_ptr = torch.tensor(ptr, dtype=torch.int32, device=self.device)
        return self._seg_ptr

    def edge_feat_csr(self, edge_feat):
        """``edge_feat`` (E, Fe) permuted into CSR-by-destination slot order (cached per tensor)."""
        key = (edge_feat.data_ptr(), tuple(edge_feat.shape), edge_feat._version)
        hit = self._sorted_cache.get("ea")
        if hit is not None and hit[0] == key:
            return hit[1]
        out = edge_feat.detach().to(torch.float32).index_select(0, self.csr().eperm).contiguous()
        self._sorted_cache["ea"] = (key, out)
        return out

    @classmethod
    def from_raw(cls, raw, device=None):
        """Build from a ``synthetic.RawBatch`` (numpy arrays)."""
        g = cls(torch.from_numpy(raw.src), torch.from_numpy(raw.dst), raw.num_nodes, raw.batch_num_nodes.tolist())
        g.ndata["x"] = torch.from_numpy(raw.x)
        g.edata["edge_attr"] = torch.from_numpy(raw.edge_attr)
        g.csr()
        return g.to(device) if device is not None else g


def graph(edges, num_nodes=None):
    """``dgl.graph((src, dst), num_nodes=n)`` (reference ``data/utils.py:64``)."""
    src, dst = edges
    src = torch.as_tensor(src).long()
    dst = torch.as_tensor(dst).long()
    if num_nodes is None:
        num_nodes = int(max(int(src.max()), int(dst.max()))) + 1 if src.numel() else 0
    return PackedGraphBatch(src, dst, num_nodes)


def batch(graphs):
    """``dgl.batch``: block-diagonal union of graphs (reference ``data/utils.py:163``)."""
    srcs, dsts, counts = [], [], []
    offset = 0
    for g in graphs:
        s, d = g.edges()
        srcs.append(s + offset)
        dsts.append(d + offset)
        counts.extend(g._counts)
        offset += g.num_nodes()
    dev = graphs[0].device
    empty = torch.zeros(0, dtype=torch.int64, device=dev)
    out = PackedGraphBatch(torch.cat(srcs) if srcs else empty, torch.cat(dsts) if dsts else empty, offset, counts)
    for key in graphs[0].ndata:
        out.ndata[key] = torch.cat([g.ndata[key] for g in graphs], dim=0)
    for key in graphs[0].edata:
        out.edata[key] = torch.cat([g.edata[key] for g in graphs], dim=0)
    if len(graphs) > 1 and all(g._csr is not None for g in graphs):
        out._csr = _concat_csr([g._csr for g in graphs])     # block-diagonal: the union's indices are the pieces' + offsets
    return out


def _concat_csr(parts):
    """CSR indices of the block-diagonal union of graphs whose own indices exist: every piece is already sorted by
    destination (stable), so the union's arrays are the pieces' arrays with node / slot offsets added -- no sort.
    Equal to ``CSRIndex(src, dst, n)`` of the union (``tests/test_graph_index.py``)."""
    out = object.__new__(CSRIndex)
    n_off, e_off = 0, 0
    cols = {k: [] for k in ("src_sorted", "dst_sorted", "eperm", "pos_by_src")}
    rp_d, rp_s = [], []
    for i, c in enumerate(parts):
        cols["src_sorted"].append(c.src_sorted + n_off)
        cols["dst_sorted"].append(c.dst_sorted + n_off)
        cols["eperm"].append(c.eperm + e_off)
        cols["pos_by_src"].append(c.pos_by_src + e_off)
        last = i == len(parts) - 1
        rp_d.append((c.rowptr_dst if last else c.rowptr_dst[:-1]) + e_off)
        rp_s.append((c.rowptr_src if last else c.rowptr_src[:-1]) + e_off)
        n_off += c.num_nodes
        e_off += c.num_edges
    for k, v in cols.items():
        setattr(out, k, torch.cat(v))
    out.rowptr_dst, out.rowptr_src = torch.cat(rp_d).to(torch.int32), torch.cat(rp_s).to(torch.int32)
    out.num_nodes, out.num_edges = n_off, e_off
    out._chunks = {}
    return out
