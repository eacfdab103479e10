"""From the reference's on-disk inputs (a directory of per-structure ``*.pt`` files + the property / HLA tables named by the
entry scripts' path flags) to packed datasets -- what ``ImmunoPredDataset`` / ``ImmunoPredDatasetComparative`` do on every
start (``data/immmunopred_dataloader.py:20-36,128-155``), done once here: ``data.tables`` joins the tables,
``data.convert_pyg_directory`` reads the graphs in the table's row order and attaches the labels."""
from __future__ import annotations

import os

from torch.utils.data import Dataset

from . import tables
from .packed import PackedDataset, convert_pyg_directory, list_structure_names

__all__ = ["packed_from_reference_inputs", "paired_from_reference_inputs", "PairedDataset", "require_paths"]


def require_paths(**paths):
    missing = {k: v for k, v in paths.items() if not os.path.exists(v)}
    if missing:
        raise SystemExit("input(s) not found: " + ", ".join(f"--{k.replace('_', '-')} {v}" for k, v in missing.items()) +
                         " (the reference never shipped its graph directories; pass --packed FILE or --synthetic N instead)")


def packed_from_reference_inputs(graph_dir, property_path, hla_path, cancer=None, feature_size=23, coord_size=3, binary=False):
    """single-graph dataset (``ImmunoPredDataset``): ``cancer`` defaults to the reference's rule ("Cancer" in the directory name).
    One item per table key that has a structure, in the table's row order, each with ITS key's values -- keys that map to the
    same structure (alleles with identical sequences) are separate items that share a graph, as in the reference
    (``data/immmunopred_dataloader.py:38-60``); the packed dataset then holds that graph once per item."""
    cancer = ("Cancer" in graph_dir) if cancer is None else cancer
    rows = tables.item_rows_from_tables(property_path, hla_path, list_structure_names(graph_dir), cancer=cancer)
    packed = convert_pyg_directory(graph_dir, feature_size=feature_size, coord_size=coord_size, labels=[r[1] for r in rows],
                                   order=[r[0] for r in rows])
    packed.binary = binary
    return packed


class PairedDataset(Dataset):
    """items ``((graph_c, graph_w), (seq_c, seq_w), target, (prop_c, prop_w))`` over two single-graph datasets of equal
    length (``.c`` / ``.w``: what ``procedures.train_model_comparative_device`` keeps on the device)"""

    def __init__(self, cancer, wildtype):
        if len(cancer) != len(wildtype):
            raise ValueError("cancer / wild-type datasets differ in length")
        self.c, self.w = cancer, wildtype
        self.class_weights = cancer.class_weights

    def __len__(self):
        return len(self.c)

    def __getitem__(self, i):
        gc, sc, y, pc = self.c[i]
        gw, sw, _, pw = self.w[i]
        return (gc, gw), (sc, sw), y, (pc, pw)


def paired_from_reference_inputs(graph_dir_cancer, graph_dir_wt, property_path_cancer, property_path_wt, hla_path,
                                 feature_size=23, coord_size=3, binary=True):
    """(cancer, wild-type) pair dataset (``ImmunoPredDatasetComparative``): one item per row of the joined table, in its row
    order, each with THAT row's values -- pairs that share a cancer or a wild-type structure are separate items
    (``data/immmunopred_dataloader.py:156-190``).  Both members are padded to the same node count: the larger of the two
    directories' maxima over the selected graphs (the reference pads each side on its own, ``:146-147``; one common count
    lets a pair share a batch layout -- ``DeviceResidentDataset.concat`` and the merged 2B-graph encoder pass require it)."""
    names_c, names_w = list_structure_names(graph_dir_cancer), list_structure_names(graph_dir_wt)
    rows = tables.paired_item_rows_from_tables(property_path_cancer, property_path_wt, hla_path, names_c, names_w)
    pc = convert_pyg_directory(graph_dir_cancer, feature_size=feature_size, coord_size=coord_size, labels=[r[1] for r in rows],
                               order=[r[0] for r in rows])
    pw = convert_pyg_directory(graph_dir_wt, feature_size=feature_size, coord_size=coord_size, labels=[r[3] for r in rows],
                               order=[r[2] for r in rows])
    n = max(int(pc.x.shape[1]), int(pw.x.shape[1]))
    pc, pw = pc.padded_to(n), pw.padded_to(n)
    pc.binary = pw.binary = binary
    return PairedDataset(_View(pc, range(len(pc))), _View(pw, range(len(pw))))


def _unique(seq):
    seen, out = set(), []
    for s in seq:
        if s not in seen:
            seen.add(s)
            out.append(s)
    return out


class _View(Dataset):
    """rows of a packed dataset by index (several pairs may share a wild-type structure)"""

    def __init__(self, packed, index):
        self.packed, self.index = packed, list(index)

    @property
    def class_weights(self):
        from collections import Counter
        counts = Counter(float(self.packed.y_bin[i]) for i in self.index)
        return {0: float(counts.get(0.0, 1)), 1: float(max(counts.get(1.0, 1), 1))}

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        return self.packed[self.index[i]]
