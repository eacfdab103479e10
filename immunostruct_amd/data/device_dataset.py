"""Device-resident dataset and on-GPU batcher.

The reference assembles every batch on the host: ``SplitDataset.__getitem__`` -> ``collate`` -> ``dgl.batch`` ->
``.to(device)`` (``data/util_dataloader.py:20-86``, ``data/utils.py:160-176``, ``procedures/train.py:20-21``).  At a
~1.6 ms train step that Python / H2D path would be the bottleneck, so here the whole dataset lives in HBM
(27 k graphs x 190 x 23 fp32 = 472 MB, sequences 642 MB -- trivial on a 288 GB part) as per-graph CSR pieces, and a
batch is assembled by ONE kernel launch (``csrc/segment_ops.hip``, ``is_batch_gather``) from a device tensor of graph
ids: every graph's edges are already in destination order, so the batch CSR is the concatenation of the pieces with
node / edge offsets added -- no sort, no host work.  The result is bit-identical to ``collate`` + ``PackedGraphBatch``
index construction on the same graphs (``tests/test_gpu_models.py::test_device_batcher_matches_collate``).
"""
from __future__ import annotations

import ctypes

import torch

from .. import _lib
from ..engine import StaticGraphBatch
from ..graph import PackedGraphBatch

__all__ = ["DeviceResidentDataset"]


class DeviceResidentDataset:
    """All items of a map-style dataset of ``(graph, sequence, target, property)`` tuples, packed on ``device``.

    ``graph`` is a single-graph :class:`PackedGraphBatch` (every graph padded to the same node count, as the reference
    does in ``data/preprocess.py:343-349``)."""

    def __init__(self, dataset, device, indices=None):
        device = torch.device(device)
        items = [dataset[i] for i in (range(len(dataset)) if indices is None else indices)]
        if not items:
            raise ValueError("empty dataset")
        graphs = [it[0] for it in items]
        n = graphs[0].num_nodes()
        if any(g.num_nodes() != n or g.batch_size != 1 for g in graphs):
            raise ValueError("all graphs must be single graphs padded to the same node count")
        fe = int(graphs[0].edata["edge_attr"].shape[1])
        csrs = [g.csr() for g in graphs]
        counts = torch.tensor([c.num_edges for c in csrs], dtype=torch.int64)
        eoff = torch.zeros(len(graphs) + 1, dtype=torch.int64)
        eoff[1:] = torch.cumsum(counts, 0)
        if int(eoff[-1]) >= 2 ** 31:
            raise ValueError("edge count exceeds int32 indexing")
        self.device = device
        self.num_graphs, self.nodes_per_graph, self.edge_feats = len(graphs), n, fe
        self.node_feats = int(graphs[0].ndata["x"].shape[1])
        self.max_edges = int(counts.max())
        cat = lambda ts: torch.cat([t.cpu() for t in ts]).contiguous().to(device)
        self.x = torch.stack([g.ndata["x"].cpu().float() for g in graphs]).contiguous().to(device)              # [G][n][F]
        self.eoff = eoff.to(torch.int32).to(device)
        self.rowptr_dst = torch.stack([c.rowptr_dst.cpu() for c in csrs]).contiguous().to(device)              # [G][n+1]
        self.rowptr_src = torch.stack([c.rowptr_src.cpu() for c in csrs]).contiguous().to(device)
        self.src = cat([c.src_sorted for c in csrs])
        self.dst = cat([c.dst_sorted for c in csrs])
        self.pos = cat([c.pos_by_src for c in csrs])
        self.ea = cat([g.edge_feat_csr(g.edata["edge_attr"]).reshape(-1, fe) for g in graphs]) if fe else \
            torch.zeros(0, 0, dtype=torch.float32, device=device)
        self.seq = torch.stack([torch.as_tensor(it[1]).float() for it in items]).contiguous().to(device)
        self.y = torch.stack([torch.as_tensor(it[2]).float().reshape(()) for it in items]).contiguous().to(device)
        self.prop = torch.stack([torch.as_tensor(it[3]).float() for it in items]).contiguous().to(device)
        self.device = self.x.device          # canonical form ("cuda" -> "cuda:0")

    @classmethod
    def from_packed(cls, packed, device, binary=None):
        """straight from a :class:`~immunostruct_amd.data.PackedDataset` (``.npz`` of ``data.convert_pyg_directory``): the file
        already holds this class' arrays, so loading is a handful of H2D copies -- no per-graph Python objects, no sorting"""
        if packed.seq is None:
            raise ValueError("the packed dataset has no labels attached")
        device = torch.device(device)
        self = object.__new__(cls)
        self.num_graphs, self.nodes_per_graph, self.node_feats = (int(v) for v in packed.x.shape)
        self.edge_feats = int(packed.ea.shape[1])
        self.max_edges = int((packed.eoff[1:] - packed.eoff[:-1]).max())
        for name, value in (("x", packed.x), ("eoff", packed.eoff), ("rowptr_dst", packed.rowptr_dst), ("rowptr_src", packed.rowptr_src),
                            ("src", packed.src), ("dst", packed.dst), ("pos", packed.pos), ("ea", packed.ea), ("prop", packed.prop)):
            setattr(self, name, value.contiguous().to(device))
        # tokens -> one-hot on the device (the file stores one byte per position)
        self.seq = torch.nn.functional.one_hot(packed.sequence_tokens().to(device).long(), 21).float()
        binary = packed.binary if binary is None else binary
        self.y = (packed.y_bin if binary else packed.y_reg).contiguous().to(device)
        self.device = self.x.device
        return self

    @classmethod
    def concat(cls, first, second):
        """the graphs of ``first`` followed by those of ``second`` (same node count / feature widths): ids of ``second``
        are shifted by ``len(first)``.  Used for (cancer, wild-type) pairs: ONE gather with ids [idx, idx + len(first)]
        assembles the merged batch the paired models encode in a single pass."""
        for k in ("nodes_per_graph", "node_feats", "edge_feats", "device"):
            if getattr(first, k) != getattr(second, k):
                raise ValueError(f"datasets differ in {k}")
        if first.seq.shape[1:] != second.seq.shape[1:] or first.prop.shape[1:] != second.prop.shape[1:]:
            raise ValueError("datasets differ in sequence / property shape")
        self = object.__new__(cls)
        self.device = first.device
        self.num_graphs = first.num_graphs + second.num_graphs
        self.nodes_per_graph, self.node_feats, self.edge_feats = first.nodes_per_graph, first.node_feats, first.edge_feats
        self.max_edges = max(first.max_edges, second.max_edges)
        for k in ("x", "rowptr_dst", "rowptr_src", "src", "dst", "pos", "ea", "seq", "prop", "y"):
            setattr(self, k, torch.cat([getattr(first, k), getattr(second, k)], dim=0))
        self.eoff = torch.cat([first.eoff, second.eoff[1:] + first.eoff[-1]])      # slot offsets; the ids inside a graph stay local
        return self

    def __len__(self):
        return self.num_graphs

    # ---- batches ----------------------------------------------------------------
    def new_batch(self, batch_size):
        """Fixed-capacity buffers for batches of ``batch_size`` graphs: (StaticGraphBatch, seq, prop, y)."""
        n, b = self.nodes_per_graph, int(batch_size)
        template = PackedGraphBatch(torch.zeros(0, dtype=torch.int64, device=self.device),
                                    torch.zeros(0, dtype=torch.int64, device=self.device), b * n, [n] * b)
        template.ndata["x"] = torch.zeros(b * n, self.node_feats, dtype=torch.float32, device=self.device)
        template.edata["edge_attr"] = torch.zeros(0, self.edge_feats, dtype=torch.float32, device=self.device)
        sg = StaticGraphBatch(template, edge_capacity=b * self.max_edges)
        seq = torch.zeros((b,) + tuple(self.seq.shape[1:]), dtype=torch.float32, device=self.device)
        prop = torch.zeros((b,) + tuple(self.prop.shape[1:]), dtype=torch.float32, device=self.device)
        y = torch.zeros(b, dtype=torch.float32, device=self.device)
        return sg, seq, prop, y

    def gather_into(self, idx, sgraph, seq, prop, y):
        """Assemble the batch of the graphs ``idx`` (int64 tensor on the device, len == the buffers' batch size) in place:
        ONE HIP launch for the graph and the sequence / property / target rows, then the work partitions of the layer
        kernels are refreshed from the new rowptr (one more launch; device-side, no sync)."""
        b = int(idx.numel())
        if b != sgraph.batch_size or idx.device != self.device or idx.dtype != torch.int64:
            raise ValueError("idx must be an int64 device tensor with one entry per graph slot of the batch buffers")
        csr = sgraph._csr
        lib = _lib.load()
        jobs = []
        for src, dst in ((self.seq, seq), (self.prop, prop), (self.y, y)):
            if (src.dtype != torch.float32 or dst.dtype != torch.float32 or not src.is_contiguous() or not dst.is_contiguous()
                    or dst.shape[0] != b or dst.shape[1:] != src.shape[1:] or dst.device != self.device):
                raise ValueError("batch buffers must be contiguous float32 device tensors shaped like the dataset's rows")
            jobs.append(_lib.RowGather(src.data_ptr(), dst.data_ptr(), int(src[0].numel()) if src.dim() > 1 else 1, 0))
        rows = (_lib.RowGather * len(jobs))(*jobs)
        _lib.check(lib.is_batch_gather(
            _lib.ptr(idx), b, self.nodes_per_graph, self.node_feats, self.edge_feats, _lib.ptr(self.x), _lib.ptr(self.eoff),
            _lib.ptr(self.rowptr_dst), _lib.ptr(self.rowptr_src), _lib.ptr(self.src), _lib.ptr(self.dst), _lib.ptr(self.pos),
            _lib.ptr(self.ea) if self.edge_feats else None, _lib.ptr(sgraph.ndata["x"]), _lib.ptr(csr.rowptr_dst),
            _lib.ptr(csr.rowptr_src), _lib.ptr(csr.src_sorted), _lib.ptr(csr.dst_sorted), _lib.ptr(csr.pos_by_src),
            _lib.ptr(sgraph._ea_csr) if self.edge_feats else None, ctypes.cast(rows, ctypes.c_void_p), len(jobs),
            _lib.stream_ptr()), "is_batch_gather")
        sgraph.refresh_partitions()
        return sgraph, seq, prop, y
