"""Packed on-disk / in-memory dataset format and the one-time converter from the reference's inputs.

The reference reads one pickled PyG ``Data`` per peptide-MHC structure (``x`` (n,22), ``coords`` (n,3), ``edge_index``
(2,E), ``name``; written by ``preprocessing/cancer_graph_construction_new_KBG.py:137-143``) and, on EVERY start, filters,
de-duplicates, strips the two hydrogen-bond features, appends the coordinates, pads to the dataset-wide node count and
converts to DGL (``data/preprocess.py:15-43``, ``:147-186``, ``:343-349``; ``data/utils.py:13-33,54-67``).  Here that
is done once: :func:`convert_pyg_directory` writes ONE ``.npz`` with the whole dataset already in the layout the GPU
batcher reads (``DeviceResidentDataset.from_packed``):

===============  =====================  ==========================================================================
``x``            (G, n, F) float32      node features ‖ coordinates, zero rows for padded nodes
``eoff``         (G+1,) int32           edge offsets of the graphs in the three edge arrays
``rowptr_dst``   (G, n+1) int32         per-graph CSR by destination (local node ids)
``rowptr_src``   (G, n+1) int32         per-graph CSR by source over the destination-sorted slots
``src``, ``dst`` (E,) int32             endpoints of the destination-sorted edges (stable: original order inside a node)
``pos``          (E,) int32             for every source-sorted position the destination-sorted slot (local to the graph)
``ea``           (E, Fe) float32        edge features in destination-sorted order (the reference: ones, Fe = 1)
``names``        (G,) str               graph names (the part after "Immuno", the reference's join key)
``seq``          (G, L) uint8           optional: tokens of the padded full sequence, 0..19 = ACDEFGHIKLMNPQRSTVWY, 20 = J
``pep``          (G, Lp) uint8          optional: tokens of the padded peptide alone (the model input WITHOUT ``--full-sequence``)
``prop``         (G, 2) float32         optional: Mprop1, Mprop2
``y_reg/y_bin``  (G,) float32           optional: (normalised) foreignness / immunogenicity
===============  =====================  ==========================================================================

The index arrays are exactly what ``graph.CSRIndex`` builds per graph (stable sorts), computed for the whole dataset
with a handful of vectorised torch calls.  Joining graphs with the property / HLA tables (pandas, ``preprocess.py:45-145``)
stays with the caller, who passes ``labels = {name: (full_sequence, mprop1, mprop2, immunogenicity, foreignness)}``.

The pickles are read WITHOUT torch_geometric: an allow-list unpickler maps every ``torch_geometric.*`` class (and unneeded pandas / networkx / graphein attachments) to an inert
attribute bag, loads tensors / storages / plain containers and refuses every other global; the four fields are looked up in
the bag (PyG 2.x keeps them in ``_store._mapping``, 1.x in ``__dict__``).
That layout follows the PyG sources; no real file was available to this build, so it is unverified against one
(``tests/test_packed_format.py`` fabricates files with stand-in classes under the same module paths).
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from ..graph import PackedGraphBatch
from .utils import AMINO_ACIDS, PADDING_CHAR

__all__ = ["PackedDataset", "convert_pyg_directory", "load_pyg_pickle", "list_structure_names"]

_TOKEN = {ch: i for i, ch in enumerate(AMINO_ACIDS + PADDING_CHAR)}
_ARRAYS = ("x", "eoff", "rowptr_dst", "rowptr_src", "src", "dst", "pos", "ea")
_OPTIONAL = ("seq", "pep", "prop", "y_reg", "y_bin")


class PackedDataset:
    """the arrays of the table above, as torch tensors on the host; ``ds[i]`` -> ``(graph, sequence one-hot, target, property)``"""

    def __init__(self, arrays, names=None, binary=False):
        for k in _ARRAYS:
            setattr(self, k, torch.as_tensor(arrays[k]))
        for k in _OPTIONAL:
            setattr(self, k, torch.as_tensor(arrays[k]) if arrays.get(k) is not None else None)
        self.names = list(names) if names is not None else [str(i) for i in range(self.x.shape[0])]
        self.binary = binary
        # which sequence an item carries: the padded HLA + peptide (``--full-sequence``, 283 x 21) or the padded peptide alone
        # (11 x 21, the reference's default: ``SplitDataset(..., full=config.full_sequence)``, data/util_dataloader.py:52-66)
        self.full_sequence = True

    # ---- construction -----------------------------------------------------------------------------
    @classmethod
    def from_graphs(cls, graphs, names=None, pad_to=None, labels=None, binary=False):
        """``graphs``: sequence of ``(x (n_i, F), src (E_i,), dst (E_i,)[, edge_attr (E_i, Fe)])`` with local node ids;
        every graph is zero-padded to ``pad_to`` (default: the largest) nodes, edges keep their endpoints."""
        g_count = len(graphs)
        if not g_count:
            raise ValueError("no graphs")
        n = int(pad_to if pad_to is not None else max(int(g[0].shape[0]) for g in graphs))
        feats = int(graphs[0][0].shape[1])
        x = torch.zeros(g_count, n, feats, dtype=torch.float32)
        counts = torch.tensor([int(g[1].numel()) for g in graphs], dtype=torch.int64)
        for i, g in enumerate(graphs):
            if g[0].shape[0] > n or g[0].shape[1] != feats:
                raise ValueError(f"graph {i}: {tuple(g[0].shape)} does not fit ({n}, {feats})")
            if g[1].numel() and (int(torch.as_tensor(g[1]).max()) >= n or int(torch.as_tensor(g[2]).max()) >= n):
                raise ValueError(f"graph {i}: edge endpoint outside the node range")
            x[i, :g[0].shape[0]] = torch.as_tensor(g[0], dtype=torch.float32)
        gid = torch.repeat_interleave(torch.arange(g_count), counts)
        src = torch.cat([torch.as_tensor(g[1]).long().reshape(-1) for g in graphs])
        dst = torch.cat([torch.as_tensor(g[2]).long().reshape(-1) for g in graphs])
        has_ea = len(graphs[0]) > 3 and graphs[0][3] is not None
        if has_ea:
            fe = max(int(torch.as_tensor(g[3]).reshape(int(g[1].numel()), -1).shape[1]) for g in graphs if int(g[1].numel()))
            ea = torch.cat([torch.as_tensor(g[3], dtype=torch.float32).reshape(int(g[1].numel()), fe) for g in graphs])
        else:
            ea = torch.ones(int(counts.sum()), 1)
        eoff = torch.zeros(g_count + 1, dtype=torch.int64)
        eoff[1:] = torch.cumsum(counts, 0)
        if int(eoff[-1]) >= 2 ** 31:
            raise ValueError("edge count exceeds int32 indexing")
        # graph ids ascend along the concatenation, so ONE stable sort by the global node id is every graph's own stable sort
        order = torch.argsort(gid * n + dst, stable=True)
        src_s, dst_s = src[order], dst[order]
        order2 = torch.argsort(gid * n + src_s, stable=True)
        pos = order2 - eoff[gid]                                        # slot, local to the graph
        rowptr = lambda idx: torch.cat([torch.zeros(g_count, 1, dtype=torch.int64),
                                        torch.cumsum(torch.bincount(gid * n + idx, minlength=g_count * n).view(g_count, n), 1)], 1)
        arrays = dict(x=x, eoff=eoff.to(torch.int32), rowptr_dst=rowptr(dst).to(torch.int32), rowptr_src=rowptr(src).to(torch.int32),
                      src=src_s.to(torch.int32), dst=dst_s.to(torch.int32), pos=pos.to(torch.int32), ea=ea[order].contiguous())
        out = cls(arrays, names, binary)
        if labels is not None:
            out.attach_labels(labels)
        return out

    def padded_to(self, n):
        """the same dataset with every graph zero-padded to ``n`` >= the current node count (padded nodes have no edges, so
        the per-graph CSR rows just repeat their last entry; edges, slots and labels are untouched)"""
        cur = int(self.x.shape[1])
        if n == cur:
            return self
        if n < cur:
            raise ValueError(f"cannot pad {cur}-node graphs down to {n}")
        grow = lambda rp: torch.cat([rp, rp[:, -1:].expand(-1, n - cur)], 1).contiguous()
        arrays = {k: getattr(self, k) for k in _ARRAYS + _OPTIONAL}
        arrays["x"] = torch.nn.functional.pad(self.x, (0, 0, 0, n - cur))
        arrays["rowptr_dst"], arrays["rowptr_src"] = grow(self.rowptr_dst), grow(self.rowptr_src)
        out = PackedDataset(arrays, self.names, self.binary)
        out.full_sequence = self.full_sequence
        return out

    def attach_labels(self, labels):
        """``labels[name] = (full_sequence, mprop1, mprop2, immunogenicity, foreignness[, peptide])``, or a LIST of such rows, one
        per item in dataset order (items that share a structure keep their own values); sequences (and, when given, peptides)
        are right-padded with the padding symbol to the longest one (``data/preprocess.py:351-362``)"""
        if isinstance(labels, dict):
            rows = [labels[name] for name in self.names]
        else:
            rows = list(labels)
            if len(rows) != len(self.names):
                raise ValueError(f"{len(rows)} label rows for {len(self.names)} items")

        def tokens(strings):
            out = np.full((len(strings), max(len(t) for t in strings)), _TOKEN[PADDING_CHAR], dtype=np.uint8)
            for i, t in enumerate(strings):
                try:
                    out[i, :len(t)] = [_TOKEN[ch] for ch in t]
                except KeyError as exc:
                    raise ValueError(f"{self.names[i]}: unknown residue {exc} in its sequence") from None
            return torch.from_numpy(out)
        self.seq = tokens([r[0] for r in rows])
        self.pep = tokens([r[5] for r in rows]) if all(len(r) > 5 for r in rows) else None
        self.prop = torch.tensor([[r[1], r[2]] for r in rows], dtype=torch.float32)
        self.y_bin = torch.tensor([r[3] for r in rows], dtype=torch.float32)
        self.y_reg = torch.tensor([r[4] for r in rows], dtype=torch.float32)

    def normalize(self):
        """foreignness -> [-1, 1] (``data/immmunopred_dataloader.py:67-70``); returns (min, max) for ``denormalize``"""
        lo, hi = self.y_reg.min(), self.y_reg.max()
        self.y_reg = 2 * (self.y_reg - (hi + lo) / 2) / (hi - lo)
        return float(lo), float(hi)

    # ---- file ----------------------------------------------------------------------------------------
    def save(self, path):
        arrays = {k: getattr(self, k).numpy() for k in _ARRAYS}
        arrays.update({k: getattr(self, k).numpy() for k in _OPTIONAL if getattr(self, k) is not None})
        np.savez_compressed(path, names=np.array(self.names, dtype=np.str_), **arrays)

    @classmethod
    def load(cls, path, binary=False):
        with np.load(path, allow_pickle=False) as f:
            missing = [k for k in _ARRAYS + ("names",) if k not in f.files]
            if missing:
                raise ValueError(f"{path}: not a packed dataset (missing {missing})")
            arrays = {k: f[k] for k in f.files if k != "names"}
            names = [str(s) for s in f["names"]]
        return cls(arrays, names, binary)

    # ---- map-style access (host loaders, tests) --------------------------------------------------------
    def __len__(self):
        return int(self.x.shape[0])

    @property
    def class_weights(self):
        pos = float(self.y_bin.sum())
        return {0: float(len(self) - pos), 1: max(pos, 1.0)}

    def sequence_tokens(self):
        """(G, L) tokens of the sequence the items carry (``full_sequence``: HLA + peptide, otherwise the peptide alone)"""
        if self.full_sequence:
            return self.seq
        if self.pep is None:
            raise ValueError("this packed dataset holds no peptide tokens (written before they were stored, or labels without "
                             "the peptide): convert it again, or train with --full-sequence")
        return self.pep

    def one_hot_sequence(self, i):
        return torch.nn.functional.one_hot(self.sequence_tokens()[i].long(), len(_TOKEN)).float()

    def graph(self, i):
        lo, hi = int(self.eoff[i]), int(self.eoff[i + 1])
        g = PackedGraphBatch(self.src[lo:hi].long(), self.dst[lo:hi].long(), int(self.x.shape[1]))
        g.ndata["x"] = self.x[i]
        g.edata["edge_attr"] = self.ea[lo:hi]
        g.csr()
        return g

    def __getitem__(self, i):
        if self.seq is None:
            raise ValueError("no labels attached to this packed dataset")
        return self.graph(i), self.one_hot_sequence(i), (self.y_bin if self.binary else self.y_reg)[i], self.prop[i]


# ---- the reference's pickles ------------------------------------------------------------------------------
class _Bag:
    """stand-in for any torch_geometric class (and for the attachments of the packages in ``_INERT_ROOTS``) inside a pickle: keeps
    the state, runs no code -- as a class (NEWOBJ / BUILD), as a callable (REDUCE) and as a plain value (a class object stored in an
    attribute, e.g. ``Data._edge_attr_cls``)"""

    def __new__(cls, *args, **kwargs):
        return object.__new__(cls)

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"_state": state})


# globals a tensor-holding PyG ``Data`` pickle needs besides the torch_geometric classes themselves; anything else is refused
_ALLOWED_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "set"),
    ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"), ("builtins", "str"), ("builtins", "bool"),
    ("builtins", "slice"), ("builtins", "range"), ("builtins", "frozenset"), ("builtins", "object"), ("builtins", "bytearray"),
    ("builtins", "bytes"), ("builtins", "complex"),
    ("_codecs", "encode"),                 # how protocol 2 spells a bytes object (str -> bytes, latin-1): pure
    ("copyreg", "_reconstructor"),         # cls.__new__ of a class that went through this same allow-list
    ("copy_reg", "_reconstructor"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch", "Size"), ("torch", "device"), ("torch", "dtype"), ("torch.serialization", "_get_layout"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
}


# Packages whose objects may ride on a graph file WITHOUT being needed: graphein's nx -> PyG conversion (the reference's writer,
# preprocessing/cancer_graph_construction_new_KBG.py:92-143) copies graph-level attributes into the ``Data`` object -- residue ids and
# names (lists of str), edge kinds (lists of sets), and, depending on its version / ``columns``, a distance matrix or the PDB table as
# pandas / numpy objects.  The reader needs ``x``, ``coords``, ``edge_index``, ``name``; everything of these packages becomes an inert
# bag (no import, no code), so such a file loads instead of being refused.  Any OTHER global (os, subprocess, builtins.eval,
# torch.storage._load_from_bytes ...) still raises.
_INERT_ROOTS = ("torch_geometric", "pandas", "networkx", "graphein", "biopandas", "scipy", "Bio")


class _Unpickler(pickle.Unpickler):
    """allow-list unpickler: torch_geometric classes become inert bags, tensors / storages / plain containers load, every other
    global (``os.system``, ``builtins.eval``, ...) raises -- a crafted ``.pt`` file cannot run code through this reader.
    ``torch.storage._load_from_bytes`` is NOT on the list: it is ``torch.load(BytesIO(b), weights_only=False)``, i.e. the
    unrestricted unpickler over attacker-chosen bytes; ``torch.save`` files carry their storages through ``persistent_load``
    and never need it."""

    def find_class(self, module, name):
        if module == "__builtin__":      # protocol-2 streams (torch.save's default) name the builtins the Python-2 way
            module = "builtins"
        if (module, name) in _ALLOWED_GLOBALS or (module == "torch" and name.endswith("Storage")) \
                or (module == "torch.storage" and name in ("UntypedStorage", "TypedStorage")):
            return super().find_class(module, name)
        if module.split(".")[0] in _INERT_ROOTS or module.split(".")[0] == "numpy":      # (numpy beyond the array reconstruction above)
            return _Bag
        raise pickle.UnpicklingError(f"global {module}.{name} is not allowed in a graph file")


class _PickleModule:
    __name__ = "immunostruct_amd_pyg_reader"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())


def _field(obj, key):
    d = getattr(obj, "__dict__", {})
    if key in d:
        return d[key]
    store = d.get("_store")
    mapping = getattr(store, "__dict__", {}).get("_mapping") if store is not None else None
    if mapping is not None and key in mapping:
        return mapping[key]
    raise KeyError(f"field {key!r} not found in the pickled graph")


def load_pyg_pickle(path):
    """``(x, coords, edge_index, name)`` of one of the reference's per-structure files"""
    obj = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleModule)
    vals = []
    for k in ("x", "coords", "edge_index"):
        v = _field(obj, k)
        if not torch.is_tensor(v) and isinstance(v, (np.ndarray, list, tuple)):
            v = torch.as_tensor(np.asarray(v))      # (an array- or list-valued field: the CONVERTED value is what is returned)
        if not torch.is_tensor(v):
            raise ValueError(f"{path}: field {k!r} is a {type(v).__name__}, not a tensor")
        vals.append(v)
    name = _field(obj, "name")
    if not isinstance(name, str):
        raise ValueError(f"{path}: field 'name' is a {type(name).__name__}, not a string")
    return (*vals, name)


def list_structure_names(directory):
    """the structure names (the part of ``graph.name`` after "Immuno") of a directory of the reference's ``*.pt`` files, with
    the reference's filter (no ``X``), first occurrence per name, directory order -- the input of the table joins
    (``data.tables.labels_from_tables``)"""
    names, seen = [], set()
    for fname in [f for f in os.listdir(directory) if f.endswith(".pt")]:
        name = load_pyg_pickle(os.path.join(directory, fname))[3]
        if "X" in name:
            continue
        key = name.split("Immuno")[1]
        if key not in seen:
            seen.add(key)
            names.append(key)
    return names


def convert_pyg_directory(directory, out_path=None, feature_size=23, coord_size=3, labels=None, drop_features=2, order=None,
                          pad_to=None):
    """The reference's ``preprocess_graphs`` + ``graph.x = cat(x, coords)`` + ``preprocess_graph`` for a directory of
    ``*.pt`` files, as a :class:`PackedDataset` (saved to ``out_path`` when given).  Rules kept: names containing ``X``
    are skipped, the first graph of every name (the part after ``Immuno``) wins, the last ``drop_features`` node
    features (hydrogen bonding) are cut, every graph is padded to the largest node count.  ``order``: structure names in
    the order the dataset should have (the reference's datasets follow the TABLE's row order, ``data.tables``; a name may
    occur several times: items that share a structure -- ``labels`` is then the list of their rows, aligned with ``order``); names
    without a file are an error, files outside ``order`` are left out.  ``pad_to``: node count to pad to instead of the
    directory's largest graph (the two directories of a cancer / wild-type pair share one count)."""
    graphs, names, seen = [], [], set()
    if labels is not None and not isinstance(labels, dict):
        if order is None or len(order) != len(labels):
            raise ValueError("a list of label rows needs `order` (one structure name per row)")
        wanted = set(order)
    else:
        wanted = set(labels) if labels is not None else None
    for fname in [f for f in os.listdir(directory) if f.endswith(".pt")]:     # directory order, as the reference
        x, coords, edge_index, name = load_pyg_pickle(os.path.join(directory, fname))
        if "X" in name:
            continue
        key = name.split("Immuno")[1]
        if key in seen or (wanted is not None and key not in wanted):
            continue
        seen.add(key)
        x = torch.cat([x[:, :x.shape[1] - drop_features].float(), coords.float()], dim=-1)
        if x.shape[1] != feature_size or coords.shape[1] != coord_size:
            raise ValueError("`convert_pyg_directory`: graph.x shape mismatch.")
        graphs.append((x, edge_index[0], edge_index[1]))
        names.append(key)
    if order is not None:
        at = {k: i for i, k in enumerate(names)}
        missing = [k for k in order if k not in at]
        if missing:
            raise KeyError(f"no graph file for {len(missing)} structure(s), e.g. {missing[0]}")
        graphs, names = [graphs[at[k]] for k in order], list(order)
    packed = PackedDataset.from_graphs(graphs, names, pad_to=pad_to, labels=labels)
    if out_path is not None:
        packed.save(out_path)
    return packed
