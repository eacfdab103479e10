"""Batch construction pinned outside the product: ``data.utils.to_dgl`` / ``pad_graph`` / ``collate`` /
``collate_amino_acid`` and ``graph.batch`` against

(a) the oracle's ``graph_ref.graph`` / ``graph_ref.batch`` (the restatement of ``dgl.graph`` / ``dgl.batch``),
(b) the reference's OWN ``data/utils.py:13-33,54-67,160-196`` imported unchanged under ``oracle/shims.py`` (skipped where
    /root/reference is absent), and
(c) ``tests/golden/batch.npz`` -- the outputs of (b) written by ``oracle/make_golden_batch.py`` -- which carries the pin to the GPU
    box, where the on-GPU batcher (``DeviceResidentDataset.gather_into``) is checked against the ORACLE's batch of the same ids and
    an index construction written here with numpy (not against the product's own ``collate`` / ``CSRIndex``).

Integer / index work: every comparison is ``torch.equal`` / ``np.array_equal``."""
import importlib
import os

import numpy as np
import pytest
import torch

from immunostruct_amd.data import augment as A
from immunostruct_amd.data import utils as U
from oracle import graph_ref, shims
from oracle import make_golden_batch as M

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch.npz")


@pytest.fixture(scope="module")
def golden():
    with np.load(GOLDEN) as f:
        return {k: f[k] for k in f.files}


def assert_graph_is(g, prefix, want):
    """the DGLGraph surface the models read (``models/hybrid_models.py:316-321``) against stored arrays"""
    src, dst = g.edges()
    assert src.dtype == torch.int64 and dst.dtype == torch.int64
    assert np.array_equal(src.numpy(), want[f"{prefix}/src"]) and np.array_equal(dst.numpy(), want[f"{prefix}/dst"])
    assert g.ndata["x"].dtype == torch.float32 and np.array_equal(g.ndata["x"].numpy(), want[f"{prefix}/x"])
    assert np.array_equal(g.edata["edge_attr"].numpy(), want[f"{prefix}/edge_attr"])
    assert g.batch_num_nodes().dtype == torch.int64
    assert np.array_equal(g.batch_num_nodes().numpy(), want[f"{prefix}/batch_num_nodes"])
    assert g.num_nodes() == int(want[f"{prefix}/batch_num_nodes"].sum()) and g.num_edges() == len(want[f"{prefix}/src"])


def assert_same_graph(a, b):
    (sa, da), (sb, db) = a.edges(), b.edges()
    assert torch.equal(sa, sb) and torch.equal(da, db)
    assert torch.equal(a.ndata["x"], b.ndata["x"]) and torch.equal(a.edata["edge_attr"], b.edata["edge_attr"])
    assert torch.equal(a.batch_num_nodes(), b.batch_num_nodes())
    assert a.num_nodes() == b.num_nodes() and a.num_edges() == b.num_edges()


def product_pairs(seed_c, seed_w):
    cancer, wt = M.build_samples(U, seed_c, 23), M.build_samples(U, seed_w, 23)
    return [((c[0], w[0]), (c[1], w[1]), c[2], (c[3], w[3])) for c, w in zip(cancer, wt)]


def test_golden_holds_what_the_test_thinks_it_holds(golden):
    assert int(golden["count"]) == M.COUNT >= 8 and int(golden["single_seed"]) == M.SINGLE_SEED
    counts = golden["single/batch_num_nodes"]
    items = M.ragged_items(M.SINGLE_SEED)
    assert len(set(s.num_nodes for s, _, _, _ in items)) > 4               # ragged before padding
    assert (counts == 23).all() and len(counts) == M.COUNT                 # padded to the largest (pad_graph)
    assert items[3][0].edge_index.shape[1] == 0                            # a graph without edges is in the batch
    assert (golden["single/src"] == golden["single/dst"]).any()            # self loops
    e = np.stack([golden["single/src"], golden["single/dst"]])
    assert len(np.unique(e, axis=1).T) < e.shape[1]                        # parallel edges


def test_collate_single_batch_equals_the_reference_golden(golden):
    g, seq, y, prop = U.collate(M.build_samples(U, M.SINGLE_SEED))
    assert_graph_is(g, "single", golden)
    assert np.array_equal(seq.numpy(), golden["single/seq"]) and np.array_equal(y.numpy(), golden["single/y"])
    assert np.array_equal(prop.numpy(), golden["single/prop"])
    # padded rows are zero rows and no edge touches them (data/utils.py:13-33)
    x = g.ndata["x"].view(M.COUNT, 23, -1)
    for i, (s, _, _, _) in enumerate(M.ragged_items(M.SINGLE_SEED)):
        assert not x[i, s.num_nodes:].any()
        assert torch.equal(x[i, :s.num_nodes, :M.FEATURES], s.x) and torch.equal(x[i, :s.num_nodes, M.FEATURES:], s.coords)


def test_collate_paired_batch_equals_the_reference_golden(golden):
    (gc, gw), (sc, sw), y, (pc, pw) = U.collate(product_pairs(M.SINGLE_SEED, M.PAIR_SEED))
    assert_graph_is(gc, "pair/cancer", golden)
    assert_graph_is(gw, "pair/wt", golden)
    for got, key in ((sc, "pair/seq_c"), (sw, "pair/seq_w"), (y, "pair/y"), (pc, "pair/prop_c"), (pw, "pair/prop_w")):
        assert np.array_equal(got.numpy(), golden[key]), key


def test_collate_amino_acid_equals_the_reference_golden(golden):
    samples = [s + (M.amino_of(i),) for i, s in enumerate(M.build_samples(U, M.SINGLE_SEED))]
    g, seq, y, prop, amino = A.collate_amino_acid(samples)
    assert_graph_is(g, "amino", golden)
    assert amino.dtype == torch.int64 and np.array_equal(amino.numpy(), golden["amino/amino"])
    assert np.array_equal(seq.numpy(), golden["single/seq"])


def test_product_chain_equals_the_oracle_restatement():
    """(a): ``graph_ref.graph`` + frames + ``graph_ref.batch`` on the same padded structures, for the whole batch, a sub-batch in
    another order, a batch of ONE graph and a batch of batches (``dgl.batch`` keeps per-graph node counts)."""
    ours = [s[0] for s in M.build_samples(U, M.SINGLE_SEED)]
    ref = []
    for s, _, _, _ in M.ragged_items(M.SINGLE_SEED):
        x = torch.cat([torch.cat([s.x, s.coords], dim=-1), torch.zeros(23 - s.num_nodes, M.FEATURES + M.COORDS)])
        g = graph_ref.graph((s.edge_index[0], s.edge_index[1]), num_nodes=23)
        g.ndata["x"], g.edata["edge_attr"] = x, torch.ones(s.edge_index.shape[1], 1)
        ref.append(g)
    for a, b in zip(ours, ref):
        assert_same_graph(a, b)
    from immunostruct_amd.graph import batch
    for ids in (list(range(M.COUNT)), [7, 3, 3, 0], [5], [3]):
        assert_same_graph(batch([ours[i] for i in ids]), graph_ref.batch([ref[i] for i in ids]))
    nested = batch([batch(ours[:3]), batch(ours[3:4]), batch(ours[4:])])
    assert_same_graph(nested, graph_ref.batch([graph_ref.batch(ref[:3]), graph_ref.batch(ref[3:4]), graph_ref.batch(ref[4:])]))
    assert_same_graph(nested, batch(ours))


@pytest.mark.skipif(not shims.reference_available(), reason="needs /root/reference")
def test_product_chain_equals_the_reference_functions():
    """(b): the reference's own ``pad_graph`` / ``to_dgl`` / ``collate`` / ``collate_amino_acid`` run here, item by item and batched"""
    shims.install()
    R = importlib.import_module("data.utils")
    assert R.__file__.startswith(shims.REFERENCE_PKG)
    for a, b in zip(M.build_samples(U, M.SINGLE_SEED), M.build_samples(R, M.SINGLE_SEED)):
        assert_same_graph(a[0], b[0])
    ga, sa, ya, pa = U.collate(M.build_samples(U, M.PAIR_SEED))
    gb, sb, yb, pb = R.collate(M.build_samples(R, M.PAIR_SEED))
    assert_same_graph(ga, gb)
    assert torch.equal(sa, sb) and torch.equal(ya, yb) and torch.equal(pa, pb)
    ref_pairs = [((c[0], w[0]), (c[1], w[1]), c[2], (c[3], w[3]))
                 for c, w in zip(M.build_samples(R, M.PAIR_SEED, 23), M.build_samples(R, M.SINGLE_SEED, 23))]
    (ga0, ga1), (sa0, sa1), ya, (pa0, pa1) = U.collate(product_pairs(M.PAIR_SEED, M.SINGLE_SEED))
    (gb0, gb1), (sb0, sb1), yb, (pb0, pb1) = R.collate(ref_pairs)
    assert_same_graph(ga0, gb0)
    assert_same_graph(ga1, gb1)
    for x, y in ((sa0, sb0), (sa1, sb1), (ya, yb), (pa0, pb0), (pa1, pb1)):
        assert torch.equal(x, y)
    mine = A.collate_amino_acid([s + (M.amino_of(i),) for i, s in enumerate(M.build_samples(U, M.PAIR_SEED))])
    theirs = R.collate_amino_acid([s + (M.amino_of(i),) for i, s in enumerate(M.build_samples(R, M.PAIR_SEED))])
    assert_same_graph(mine[0], theirs[0])
    assert all(torch.equal(x, y) for x, y in zip(mine[1:], theirs[1:]))
    # the error behaviour of pad_graph (data/utils.py:16-18)
    for mod in (U, R):
        s = M.ragged_items(1)[0][0]
        with pytest.raises(ValueError, match="graph.x shape mismatch"):
            mod.pad_graph(s, 30, M.FEATURES + 1, M.COORDS)


def numpy_index(src, dst, n):
    """destination-ordered CSR of a COO edge list as the kernels read it, written with numpy for this test (stable sorts: a node's
    in-edges keep their list order -- the fixed summation order of the segment sums)"""
    order = np.argsort(dst, kind="stable")
    src_sorted, dst_sorted = src[order], dst[order]
    rowptr = lambda idx: np.concatenate([[0], np.cumsum(np.bincount(idx, minlength=n))]).astype(np.int32)
    return dict(eperm=order, src_sorted=src_sorted.astype(np.int32), dst_sorted=dst_sorted.astype(np.int32), rowptr_dst=rowptr(dst),
                rowptr_src=rowptr(src), pos_by_src=np.argsort(src_sorted, kind="stable").astype(np.int32))


def test_host_index_of_the_golden_batch(golden):
    """``CSRIndex`` of the collated batch (built per graph in ``to_dgl`` and concatenated by ``graph.batch``) against the numpy
    construction on the REFERENCE's batched edge list"""
    g, _, _, _ = U.collate(M.build_samples(U, M.SINGLE_SEED))
    want = numpy_index(golden["single/src"], golden["single/dst"], int(golden["single/batch_num_nodes"].sum()))
    c = g.csr()
    for k, v in want.items():
        assert np.array_equal(getattr(c, k).numpy(), v), k


@pytest.mark.gpu
def test_device_batcher_equals_the_oracle_batch(cuda_device, golden):
    """``DeviceResidentDataset.gather_into`` (one HIP launch from graph ids) against ``graph_ref.batch`` of the same ids -- the
    oracle's restatement of ``collate`` -> ``dgl.batch`` (``data/utils.py:160-176``) -- with the kernel-side index arrays derived
    from the oracle's edge list by ``numpy_index``; the full batch in dataset order is the reference's golden batch."""
    from immunostruct_amd.data import DeviceResidentDataset
    samples = M.build_samples(U, M.SINGLE_SEED)
    ref_graphs = []
    for s, _, _, _ in M.ragged_items(M.SINGLE_SEED):
        g = graph_ref.graph((s.edge_index[0], s.edge_index[1]), num_nodes=23)
        g.ndata["x"] = torch.cat([torch.cat([s.x, s.coords], dim=-1), torch.zeros(23 - s.num_nodes, M.FEATURES + M.COORDS)])
        g.edata["edge_attr"] = torch.ones(s.edge_index.shape[1], 1)
        ref_graphs.append(g)
    dds = DeviceResidentDataset(samples, cuda_device)
    for ids in (list(range(M.COUNT)), [9, 3, 3, 0, 5, 1, 3, 8, 2, 2], [3] * M.COUNT):      # the last: a batch without any edge
        buf = dds.new_batch(len(ids))
        sg, seq, prop, y = dds.gather_into(torch.tensor(ids, dtype=torch.int64, device=cuda_device), *buf)
        want_g = graph_ref.batch([ref_graphs[i] for i in ids])
        src, dst = (t.numpy() for t in want_g.edges())
        if ids == list(range(M.COUNT)):
            assert np.array_equal(src, golden["single/src"]) and np.array_equal(dst, golden["single/dst"])
            assert np.array_equal(sg.ndata["x"].cpu().numpy(), golden["single/x"])
        e = len(src)
        want = numpy_index(src, dst, want_g.num_nodes())
        c = sg.csr()
        assert torch.equal(sg.ndata["x"].cpu(), want_g.ndata["x"])
        assert torch.equal(sg.batch_num_nodes().cpu(), want_g.batch_num_nodes())
        assert np.array_equal(c.rowptr_dst.cpu().numpy(), want["rowptr_dst"]) and np.array_equal(c.rowptr_src.cpu().numpy(), want["rowptr_src"])
        for k in ("src_sorted", "dst_sorted", "pos_by_src"):
            assert np.array_equal(getattr(c, k)[:e].cpu().numpy(), want[k]), k
        assert np.array_equal(sg.edge_feat_csr(None)[:e].cpu().numpy(), want_g.edata["edge_attr"].numpy()[want["eperm"]])
        assert torch.equal(seq.cpu(), torch.stack([samples[i][1] for i in ids]))
        assert torch.equal(prop.cpu(), torch.stack([samples[i][3] for i in ids]))
        assert torch.equal(y.cpu(), torch.stack([samples[i][2] for i in ids]))
