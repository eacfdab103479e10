"""Packed dataset format + converter from the reference's per-structure pickles."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from immunostruct_amd import data as D
from immunostruct_amd.graph import CSRIndex

AA = "ACDEFGHIKLMNPQRSTVWY"


def random_graphs(count, seed, nmax=12):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(count):
        n = int(rs.randint(5, nmax + 1))
        e = 0 if i == 1 else int(rs.randint(n, 4 * n))
        x = torch.from_numpy(rs.normal(size=(n, 6)).astype(np.float32))
        out.append((x, torch.from_numpy(rs.randint(0, n, size=e)), torch.from_numpy(rs.randint(0, n, size=e)),
                    torch.from_numpy(rs.rand(e, 2).astype(np.float32))))
    return out


def test_packed_index_arrays_equal_per_graph_csr():
    graphs = random_graphs(7, seed=3)
    p = D.PackedDataset.from_graphs(graphs)
    n = p.x.shape[1]
    assert n == max(g[0].shape[0] for g in graphs) and p.eoff[-1] == sum(g[1].numel() for g in graphs)
    for i, (x, src, dst, ea) in enumerate(graphs):
        c = CSRIndex(src, dst, n)
        lo, hi = int(p.eoff[i]), int(p.eoff[i + 1])
        assert hi - lo == src.numel()
        for name, want in (("src", c.src_sorted), ("dst", c.dst_sorted), ("pos", c.pos_by_src)):
            assert torch.equal(getattr(p, name)[lo:hi], want), name
        assert torch.equal(p.rowptr_dst[i], c.rowptr_dst) and torch.equal(p.rowptr_src[i], c.rowptr_src)
        assert torch.equal(p.ea[lo:hi], ea[c.eperm])
        assert torch.equal(p.x[i, :x.shape[0]], x) and float(p.x[i, x.shape[0]:].abs().sum()) == 0
    with pytest.raises(ValueError):
        D.PackedDataset.from_graphs(graphs, pad_to=4)
    bad = list(graphs)
    bad[0] = (bad[0][0], torch.tensor([99]), torch.tensor([0]), torch.zeros(1, 2))
    with pytest.raises(ValueError):
        D.PackedDataset.from_graphs(bad)


def labels_for(names, seed=0):
    rs = np.random.RandomState(seed)
    return {nm: ("".join(rs.choice(list(AA), size=int(rs.randint(20, 30)))), float(rs.rand()), float(rs.rand()),
                 float(rs.rand() < 0.3), float(rs.normal())) for nm in names}


def test_save_load_round_trip_and_items(tmp_path):
    graphs = random_graphs(5, seed=4)
    names = [f"g{i}" for i in range(5)]
    lab = labels_for(names)
    p = D.PackedDataset.from_graphs(graphs, names, labels=lab)
    path = os.path.join(tmp_path, "packed.npz")
    p.save(path)
    q = D.PackedDataset.load(path, binary=True)
    assert q.names == names and q.binary
    for k in ("x", "eoff", "rowptr_dst", "rowptr_src", "src", "dst", "pos", "ea", "seq", "prop", "y_reg", "y_bin"):
        assert torch.equal(getattr(p, k), getattr(q, k)), k
    length = max(len(v[0]) for v in lab.values())
    assert q.seq.shape == (5, length) and q.seq.dtype == torch.uint8
    g, seq, y, prop = q[2]
    want = torch.from_numpy(D.one_hot_encode_sequence(D.pad_peptide_sequence(lab["g2"][0], length))).float()
    assert torch.equal(seq, want) and float(y) == lab["g2"][3] and prop.tolist() == pytest.approx([lab["g2"][1], lab["g2"][2]])
    assert g.num_nodes() == p.x.shape[1] and g.num_edges() == graphs[2][1].numel()
    assert torch.equal(g.csr().src_sorted, p.src[int(p.eoff[2]):int(p.eoff[3])])         # already destination-sorted: a fixed point
    lo, hi = q.normalize()
    assert float(q.y_reg.min()) == pytest.approx(-1) and float(q.y_reg.max()) == pytest.approx(1) and lo < hi
    assert q.class_weights[0] + q.class_weights[1] >= 5
    np.savez(os.path.join(tmp_path, "other.npz"), a=np.zeros(3))
    with pytest.raises(ValueError):
        D.PackedDataset.load(os.path.join(tmp_path, "other.npz"))
    with pytest.raises(ValueError):
        D.PackedDataset.from_graphs(graphs, names, labels={**lab, "g0": ("AB?", 0, 0, 0, 0)})


def fabricate_pickles(directory, layout):
    """files shaped like the reference's inputs, written with stand-in classes under torch_geometric's module paths"""
    mods = {name: types.ModuleType(name) for name in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.data",
                                                       "torch_geometric.data.storage")}

    class GlobalStorage:
        def __init__(self, mapping):
            self._mapping = mapping

    class Data:
        def __init__(self, **fields):
            if layout == "2.x":
                self._store = GlobalStorage(fields)
                self._edge_attr_cls = None
            else:
                self.__dict__.update(fields)

    Data.__module__, Data.__qualname__ = "torch_geometric.data.data", "Data"
    GlobalStorage.__module__, GlobalStorage.__qualname__ = "torch_geometric.data.storage", "GlobalStorage"
    mods["torch_geometric.data.data"].Data = Data
    mods["torch_geometric.data.storage"].GlobalStorage = GlobalStorage
    sys.modules.update(mods)
    rs = np.random.RandomState(5)
    truth = {}
    try:
        for i, key in enumerate(["AAA_1", "CCC_2", "AAA_1", "DXD_3", "EEE_4"]):
            n = 6 + i
            x = torch.from_numpy(rs.rand(n, 22).astype(np.float32))
            coords = torch.from_numpy(rs.normal(size=(n, 3)).astype(np.float32))
            ei = torch.from_numpy(rs.randint(0, n, size=(2, 3 * n)))
            torch.save(Data(x=x, coords=coords, edge_index=ei, name=f"file{i}Immuno{key}"), os.path.join(directory, f"s{i}.pt"))
            truth.setdefault(key, []).append((x, coords, ei))
    finally:
        for name in mods:
            sys.modules.pop(name, None)
    return truth


@pytest.mark.parametrize("layout", ["2.x", "1.x"])
def test_convert_pyg_directory(tmp_path, layout):
    truth = fabricate_pickles(str(tmp_path), layout)
    assert "torch_geometric" not in sys.modules                             # the converter must not need it
    out = os.path.join(tmp_path, "iedb.npz")
    packed = D.convert_pyg_directory(str(tmp_path), out)
    assert sorted(packed.names) == ["AAA_1", "CCC_2", "EEE_4"]              # 'X' names dropped, duplicates collapsed
    assert packed.x.shape[1:] == (10, 23)                                   # padded to the largest kept graph; 22 - 2 + 3 features
    for i, key in enumerate(packed.names):
        cands = truth[key]
        x_kept = packed.x[i]
        match = [c for c in cands if torch.equal(x_kept[:c[0].shape[0]], torch.cat([c[0][:, :-2], c[1]], 1))]
        assert len(match) == 1
        x, coords, ei = match[0]
        c = CSRIndex(ei[0], ei[1], 10)
        lo, hi = int(packed.eoff[i]), int(packed.eoff[i + 1])
        assert torch.equal(packed.src[lo:hi], c.src_sorted) and torch.equal(packed.dst[lo:hi], c.dst_sorted)
        assert bool((packed.ea[lo:hi] == 1).all()) and packed.ea.shape[1] == 1
    again = D.PackedDataset.load(out)
    assert again.names == packed.names and torch.equal(again.x, packed.x)
    lab = labels_for(["AAA_1", "EEE_4"])
    sub = D.convert_pyg_directory(str(tmp_path), labels=lab)
    assert sorted(sub.names) == ["AAA_1", "EEE_4"] and sub.seq is not None
    with pytest.raises(ValueError):
        D.convert_pyg_directory(str(tmp_path), feature_size=24)


def test_device_dataset_from_packed_equals_item_construction():
    """DeviceResidentDataset.from_packed (H2D copies of the file's arrays) == construction from map-style items"""
    graphs = [(g[0], g[1], g[2], g[3][:, :1]) for g in random_graphs(6, seed=8)]
    names = [f"g{i}" for i in range(6)]
    packed = D.PackedDataset.from_graphs(graphs, names, labels=labels_for(names, 2))
    a = D.DeviceResidentDataset.from_packed(packed, "cpu")
    b = D.DeviceResidentDataset(packed, "cpu")
    for k in ("x", "eoff", "rowptr_dst", "rowptr_src", "src", "dst", "pos", "ea", "seq", "prop", "y"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert (a.num_graphs, a.nodes_per_graph, a.node_feats, a.edge_feats, a.max_edges) == \
           (b.num_graphs, b.nodes_per_graph, b.node_feats, b.edge_feats, b.max_edges)
