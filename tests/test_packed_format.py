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


def fabricate_named(directory, names, seed=11):
    """one 2.x-layout graph file per structure name (name = the part after "Immuno")"""
    os.makedirs(directory, exist_ok=True)
    mods = {name: types.ModuleType(name) for name in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.data",
                                                       "torch_geometric.data.storage")}

    class GlobalStorage:
        def __init__(self, mapping):
            self._mapping = mapping

    class Data:
        def __init__(self, **fields):
            self._store = GlobalStorage(fields)

    Data.__module__, Data.__qualname__ = "torch_geometric.data.data", "Data"
    GlobalStorage.__module__, GlobalStorage.__qualname__ = "torch_geometric.data.storage", "GlobalStorage"
    mods["torch_geometric.data.data"].Data = Data
    mods["torch_geometric.data.storage"].GlobalStorage = GlobalStorage
    sys.modules.update(mods)
    rs = np.random.RandomState(seed)
    try:
        for i, key in enumerate(names):
            n = 7 + i % 4
            torch.save(Data(x=torch.from_numpy(rs.rand(n, 22).astype(np.float32)), coords=torch.from_numpy(rs.normal(size=(n, 3)).astype(np.float32)),
                            edge_index=torch.from_numpy(rs.randint(0, n, size=(2, 2 * n))), name=f"AF_{i}_Immuno{key}"),
                       os.path.join(directory, f"z{i:03d}.pt"))
    finally:
        for name in mods:
            sys.modules.pop(name, None)


def test_reader_returns_the_converted_value_of_array_and_list_fields(tmp_path):
    """a file whose ``x`` / ``coords`` / ``edge_index`` is a numpy array or a list passed the reader's check on a CONVERTED
    copy while the caller received the unconverted object; the tensors are what comes back, and the converter packs the file"""
    mods = {name: types.ModuleType(name) for name in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.data")}

    class Data:
        def __init__(self, **fields):
            self.__dict__.update(fields)      # (the 1.x layout: fields in the object's own dict)

    Data.__module__, Data.__qualname__ = "torch_geometric.data.data", "Data"
    mods["torch_geometric.data.data"].Data = Data
    sys.modules.update(mods)
    rs = np.random.RandomState(4)
    n = 9
    x, coords = rs.rand(n, 22).astype(np.float32), rs.normal(size=(n, 3)).astype(np.float32)
    edges = [[int(v) for v in rs.randint(0, n, size=14)], [int(v) for v in rs.randint(0, n, size=14)]]
    path = os.path.join(tmp_path, "a000.pt")
    try:
        torch.save(Data(x=x, coords=coords, edge_index=edges, name="AF_0_ImmunoGILGFVFTL_A0201"), path)
    finally:
        for name in mods:
            sys.modules.pop(name, None)
    gx, gc, ge, gn = D.load_pyg_pickle(path)
    assert all(torch.is_tensor(t) for t in (gx, gc, ge)) and gn == "AF_0_ImmunoGILGFVFTL_A0201"
    assert torch.equal(gx, torch.from_numpy(x)) and torch.equal(gc, torch.from_numpy(coords)) and torch.equal(ge, torch.tensor(edges))
    packed = D.convert_pyg_directory(str(tmp_path))
    assert packed.names == ["GILGFVFTL_A0201"] and packed.x.shape == (1, n, 23) and int(packed.eoff[-1]) == 14
    with pytest.raises(ValueError, match="not a tensor"):
        sys.modules.update(mods)
        try:
            torch.save(Data(x="nope", coords=coords, edge_index=edges, name="AF_1_ImmunoX_Y"), os.path.join(tmp_path, "b.pt"))
        finally:
            for name in mods:
                sys.modules.pop(name, None)
        D.load_pyg_pickle(os.path.join(tmp_path, "b.pt"))


def test_reference_path_flags_build_the_datasets(tmp_path):
    """``--graph-dir-* / --property-path-* / --hla-path`` (train_IEDB_wFT.py:26-29, train_Cancer_wFT.py:30-39): graph files +
    tables -> datasets in the TABLE's row order with the joined labels (``data.reference_inputs`` over ``data.tables``)"""
    import pandas as pd
    from immunostruct_amd.data import tables as T
    rs = np.random.RandomState(2)
    hla = {"HLA-A*02:01": "".join(rs.choice(list(AA), size=272)), "HLA-B*07:02": "".join(rs.choice(list(AA), size=272))}
    hla_csv = os.path.join(tmp_path, "hla.csv")
    pd.DataFrame({"allele": list(hla), "seqs": list(hla.values())}).to_csv(hla_csv, index=False)
    peps = ["".join(rs.choice(list(AA), size=k)) for k in (9, 10, 11, 9, 9, 10)]
    alle = ["HLA-A*02:01", "HLA-B*07:02", "HLA-A*02:01", "HLA-A*02:01", "HLA-B*07:02", "HLA-A*02:01"]
    iedb = pd.DataFrame({"peptide": peps, "allele": alle, "Foreignness_Score": [0.1, 0.2, np.nan, 0.4, 0.5, 0.6],
                         "smoothed_foreign": [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], "Mprop1": rs.rand(6), "Mprop2": rs.rand(6),
                         "immunogenicity": [0, 1, 0, 1, 0, 0]})
    table = os.path.join(tmp_path, "iedb.txt")
    iedb.to_csv(table, sep="\t", index=False)
    names = [T.structure_name(hla[a] + p) for p, a in zip(peps, alle)]
    gdir = os.path.join(tmp_path, "graph_pyg_IEDB")
    fabricate_named(gdir, [names[5], names[0], names[2], names[1], "Q" * 99 + "_abcde", names[3]])      # row 4 has no structure
    ds = D.packed_from_reference_inputs(gdir, table, hla_csv)
    assert ds.names == [names[0], names[1], names[3], names[5]]          # table order; NaN-score row 2 and structure-less row 4 dropped
    assert ds.y_bin.tolist() == [0.0, 1.0, 1.0, 0.0] and ds.y_reg.tolist() == [1.0, 2.0, 4.0, 6.0]
    assert ds.seq.shape == (4, 282) and ds.x.shape[1:] == (10, 23)      # padded to the longest kept sequence (272 + a 10-mer)
    assert ds.prop[2].tolist() == pytest.approx([iedb["Mprop1"][3], iedb["Mprop2"][3]])
    g, seq, y, prop = ds[0]
    assert seq.shape == (282, 21) and float(seq[272 + 9:, 20].sum()) == 1.0       # a 9-mer: one trailing pad symbol
    # pairs: two cancer peptides share one wild-type peptide
    muts = ["".join(rs.choice(list(AA), size=9)) for _ in range(3)]
    wts = ["".join(rs.choice(list(AA), size=9)) for _ in range(2)]
    cancer = pd.DataFrame({"mut_pep": muts, "wt_pep": [wts[0], wts[0], wts[1]], "allele": ["HLA-A0201"] * 3, "immunogenicity": [1, 0, 0],
                           "foreign": [0.3, 0.2, 0.1], "smoothed_foreign": [3.0, 2.0, 1.0], "Mprop1": [0.1, 0.2, 0.3], "Mprop2": [0.4, 0.5, 0.6]})
    wild = cancer[["mut_pep", "wt_pep", "allele", "immunogenicity", "foreign"]].assign(Mprop1_wt=[0.7, 0.7, 0.8], Mprop2_wt=[0.9, 0.9, 1.0])
    tc, tw = os.path.join(tmp_path, "c.txt"), os.path.join(tmp_path, "w.txt")
    cancer.to_csv(tc, sep="\t", index=False)
    wild.to_csv(tw, sep="\t", index=False)
    dc, dw = os.path.join(tmp_path, "graph_pyg_Cancer"), os.path.join(tmp_path, "graph_pyg_Cancer_WT")
    fabricate_named(dc, [T.structure_name(hla["HLA-A*02:01"] + m) for m in muts], seed=3)
    fabricate_named(dw, [T.structure_name(hla["HLA-A*02:01"] + w) for w in wts], seed=4)
    pairs = D.paired_from_reference_inputs(dc, dw, tc, tw, hla_csv)
    assert len(pairs) == 3 and pairs.class_weights == {0: 2.0, 1: 1.0}
    (gc, gw), (sc, sw), y, (pc, pw) = pairs[1]
    assert float(y) == 0.0 and pc.tolist() == pytest.approx([0.2, 0.5]) and pw.tolist() == pytest.approx([0.7, 0.9])
    # the directories' largest graphs differ (9 vs 8 nodes): both members are padded to the common maximum, so that the pair
    # can share one batch layout on the device (`concat` used to raise on the reference's real inputs)
    assert gc.num_nodes() == gw.num_nodes() == 9
    assert pairs.c.packed.x.shape[1] == pairs.w.packed.x.shape[1] == 9
    both = D.DeviceResidentDataset.concat(D.DeviceResidentDataset(pairs.c, "cpu"), D.DeviceResidentDataset(pairs.w, "cpu"))
    assert both.nodes_per_graph == 9 and both.num_graphs == 6
    # re-padding leaves the graph itself alone
    small = D.convert_pyg_directory(dw, labels=None)
    wide = small.padded_to(12)
    assert wide.x.shape[1] == 12 and torch.equal(wide.x[:, :8], small.x) and float(wide.x[:, 8:].abs().sum()) == 0.0
    assert torch.equal(wide.rowptr_dst[:, :9], small.rowptr_dst) and bool((wide.rowptr_dst[:, 9:] == small.rowptr_dst[:, -1:]).all())
    assert torch.equal(D.convert_pyg_directory(dw, pad_to=12).x, wide.x)
    assert torch.equal(pairs[0][1][1], pairs[1][1][1])    # pairs 0 and 1 share the wild-type member


def test_graph_file_reader_refuses_foreign_globals(tmp_path):
    """the reader's unpickler is an allow-list: a crafted file cannot reach os.system & co."""
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))
    path = os.path.join(tmp_path, "evil.pt")
    torch.save({"x": Evil()}, path)
    with pytest.raises(pickle.UnpicklingError):
        D.load_pyg_pickle(path)

    # torch.storage._load_from_bytes is torch.load(weights_only=False) over embedded bytes -- the unrestricted unpickler
    # by another name: refused as well
    import io
    inner = io.BytesIO()
    torch.save({"x": Evil()}, inner)

    class Smuggle:
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (inner.getvalue(),))
    path2 = os.path.join(tmp_path, "smuggle.pt")
    torch.save({"x": Smuggle()}, path2)
    with pytest.raises(pickle.UnpicklingError, match="_load_from_bytes"):
        D.load_pyg_pickle(path2)


def test_reader_on_a_file_shaped_like_pyg_2_5_with_graphein_attachments(tmp_path):
    """What ``torch.save(g_pyg, ...)`` of the reference's writer leaves on disk (preprocessing/cancer_graph_construction_new_KBG.py:92-143),
    reproduced from the sources of torch_geometric 2.5.3 and graphein's nx -> PyG convertor without either package: ``Data.__dict__`` =
    ``{_edge_attr_cls, _tensor_attr_cls, _store}`` with the two attribute classes stored AS CLASS OBJECTS, ``_store`` a
    ``GlobalStorage`` whose ``__getstate__`` turns its weak ``_parent`` reference into the ``Data`` object itself (a cycle in the
    pickle), and the mapping holding, beside the four fields the reader needs, what graphein copies over: residue ids / names (lists
    of str), edge kinds (a list of sets), b-factors, and a distance matrix as a pandas DataFrame.  The reader must return the four
    fields, bit-exact, without importing pandas' unpickling machinery for the attachment (it becomes an inert bag)."""
    import weakref
    import pandas as pd
    names = ("torch_geometric", "torch_geometric.data", "torch_geometric.data.data", "torch_geometric.data.storage")
    mods = {n: types.ModuleType(n) for n in names}

    class BaseStorage:
        def __init__(self, _parent=None, **kwargs):
            self.__dict__["_mapping"] = dict(kwargs)
            self.__dict__["_parent"] = weakref.ref(_parent) if _parent is not None else None

        def __getstate__(self):               # (torch_geometric/data/storage.py: the weak reference becomes the object)
            out = self.__dict__.copy()
            parent = out.get("_parent")
            if parent is not None:
                out["_parent"] = parent()
            return out

        def __setstate__(self, mapping):
            for k, v in mapping.items():
                self.__dict__[k] = v

    class GlobalStorage(BaseStorage):
        pass

    class DataEdgeAttr:
        pass

    class DataTensorAttr:
        pass

    class Data:
        def __init__(self, **fields):
            self.__dict__["_tensor_attr_cls"] = DataTensorAttr
            self.__dict__["_edge_attr_cls"] = DataEdgeAttr
            self.__dict__["_store"] = GlobalStorage(_parent=self, **fields)

        def __getstate__(self):
            return self.__dict__.copy()

        def __setstate__(self, mapping):
            for k, v in mapping.items():
                self.__dict__[k] = v

    for cls, mod in ((Data, "torch_geometric.data.data"), (DataEdgeAttr, "torch_geometric.data.data"),
                     (DataTensorAttr, "torch_geometric.data.data"), (BaseStorage, "torch_geometric.data.storage"),
                     (GlobalStorage, "torch_geometric.data.storage")):
        cls.__module__, cls.__qualname__ = mod, cls.__name__
        setattr(mods[mod], cls.__name__, cls)
    sys.modules.update(mods)
    rs = np.random.RandomState(3)
    n = 9
    x = torch.from_numpy(rs.rand(n, 22).astype(np.float32))
    coords = torch.from_numpy(rs.normal(size=(n, 3)).astype(np.float32))
    ei = torch.from_numpy(rs.randint(0, n, size=(2, 20)))
    path = os.path.join(tmp_path, "full.pt")
    try:
        g = Data(edge_index=ei, coords=coords, name="7abcImmunoGILGFVFTL_A0201", node_id=[f"A:GLY:{i}" for i in range(n)],
                 residue_name=["GLY"] * n, chain_id=["A"] * n, b_factor=torch.from_numpy(rs.rand(n).astype(np.float32)),
                 kind=[{"peptide_bond"}, {"hbond", "ionic"}] * 10, num_nodes=n,
                 dist_mat=pd.DataFrame(rs.rand(n, n), index=[f"r{i}" for i in range(n)]))
        g._store._mapping["x"] = x            # (the writer assigns ``g_pyg.x`` last)
        torch.save(g, path)
    finally:
        for name in mods:
            sys.modules.pop(name, None)
    assert "torch_geometric" not in sys.modules
    gx, gc, ge, gn = D.load_pyg_pickle(path)
    assert torch.equal(gx, x) and torch.equal(gc, coords) and torch.equal(ge, ei) and gn == "7abcImmunoGILGFVFTL_A0201"
    packed = D.convert_pyg_directory(str(tmp_path))
    assert packed.names == ["GILGFVFTL_A0201"] and packed.x.shape == (1, n, 23)
