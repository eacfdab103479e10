"""Host-side logic: CSR index construction, batching, synthetic generator invariants."""
import numpy as np
import torch

from immunostruct_amd import synthetic
from immunostruct_amd.graph import PackedGraphBatch, batch


def test_csr_by_destination_and_by_source():
    raw = synthetic.make_batch(3, seed=9, n_pad=25, n_real_choices=(20, 23, 25))
    g = PackedGraphBatch.from_raw(raw)
    csr = g.csr()
    src, dst = raw.src, raw.dst
    rp = csr.rowptr_dst.numpy()
    assert rp[0] == 0 and rp[-1] == raw.num_edges
    for v in range(raw.num_nodes):
        slots = np.arange(rp[v], rp[v + 1])
        orig = csr.eperm.numpy()[slots]
        assert (dst[orig] == v).all()
        assert (np.diff(orig) > 0).all()           # stable: original relative order kept
        assert (csr.src_sorted.numpy()[slots] == src[orig]).all()
    rps, pos = csr.rowptr_src.numpy(), csr.pos_by_src.numpy()
    seen = np.zeros(raw.num_edges, dtype=bool)
    for u in range(raw.num_nodes):
        for p in pos[rps[u]:rps[u + 1]]:
            assert csr.src_sorted.numpy()[p] == u
            seen[p] = True
    assert seen.all()
    assert g.seg_ptr().tolist() == [0, 25, 50, 75]
    assert g.uniform_nodes_per_graph() == 25


def test_batch_offsets_and_frames():
    raws = [synthetic.make_batch(1, seed=s, n_pad=12, n_real_choices=(10, 11, 12)) for s in (1, 2)]
    gs = [PackedGraphBatch.from_raw(r) for r in raws]
    big = batch(gs)
    assert big.num_nodes() == 24 and big.batch_size == 2
    s, d = big.edges()
    e0 = raws[0].num_edges
    assert (s[e0:] >= 12).all() and (d[e0:] >= 12).all() and (s[:e0] < 12).all()
    assert big.ndata["x"].shape == (24, 23) and big.edata["edge_attr"].shape[0] == s.numel()


def test_synthetic_batch_invariants():
    raw = synthetic.make_batch(8, seed=3)
    assert raw.x.shape == (8 * 190, 23)
    assert (raw.src != raw.dst).all()                       # no self loops (sqrt'(0) would be NaN)
    onehot = raw.x[:, :20].sum(1)
    real = onehot == 1
    assert set(np.unique(onehot)) <= {0.0, 1.0}
    assert real[raw.src].all() and real[raw.dst].all()      # padded nodes carry no edges
    assert (raw.x[~real] == 0).all()
    seq = raw.one_hot_sequence()
    assert seq.shape == (8, 283, 21) and (seq.sum(-1) == 1).all()
    again = synthetic.make_batch(8, seed=3)
    assert np.array_equal(raw.x, again.x) and np.array_equal(raw.src, again.src)


def test_balanced_node_chunks_cover_all_nodes():
    from immunostruct_amd.graph import balanced_node_chunks
    rowptr = torch.tensor([0, 3, 3, 10, 11, 11, 11, 40, 41], dtype=torch.int32)
    for k in (1, 2, 3, 8, 64):
        t = balanced_node_chunks(rowptr, k)
        b = t[:, 0].tolist()
        assert b[0] == 0 and b[-1] == 8 and len(b) == k + 1 and all(x <= y for x, y in zip(b, b[1:]))
        assert t[:, 1].tolist() == [int(rowptr[v]) for v in b]
    # two-level shares (graph.chunk_shares): only for the full grid of 2048 chunks, when the tiles per wave pair are odd
    from immunostruct_amd.graph import chunk_shares
    for e in (72313, 144700, 410000, 40000, 110000):
        wa, wb = chunk_shares(torch.tensor([e]), 2048)
        p = -(-2 * e // (16 * 2048))
        exp = ((p + 1) // 2, (p - 1) // 2) if (p % 2 == 1 and p >= 3 and p * 16 * 2048 - 2 * e >= 6 * 2048) else (1, 1)
        assert (int(wa), int(wb)) == exp, (e, p, int(wa), int(wb))
        assert tuple(int(v) for v in chunk_shares(torch.tensor([e]), 1024)) == (1, 1)
    assert tuple(int(v) for v in chunk_shares(torch.tensor([72313]), 2048)) == (3, 2)
    # ... and only while the larger workgroups' nodes fit one pass of the forward kernel's node half: B = 128 x 190 nodes keeps 3 : 2
    # at E / N = 3, but not 2 : 1 at E / N = 2 (63 nodes per workgroup of the first half on average: a second node pass)
    assert tuple(int(v) for v in chunk_shares(torch.tensor([72313]), 2048, 24320)) == (3, 2)      # 57 nodes: still one pass
    assert tuple(int(v) for v in chunk_shares(torch.tensor([43000]), 2048)) == (2, 1)
    assert tuple(int(v) for v in chunk_shares(torch.tensor([43000]), 2048, 24320)) == (1, 1)      # 64 + 4 nodes: a second pass
    assert tuple(int(v) for v in chunk_shares(torch.tensor([43000]), 2048, 40000)) == (2, 1)      # 105 + 4 against 79 + 4 nodes: two passes either way
    reg = torch.arange(0, 35 * 2049, 35, dtype=torch.int32)        # 2048 nodes of in-degree 35: one node per flat chunk
    t = balanced_node_chunks(reg, 2048)
    edges = (t[1:, 1] - t[:-1, 1]).tolist()
    assert sum(edges) == 35 * 2048 and t[0, 0] == 0 and t[-1, 0] == 2048
    assert abs(sum(edges[:1024]) / sum(edges[1024:]) - 1.5) < 0.02      # 3 : 2 between the halves


def test_batch_concatenates_existing_csr_pieces():
    """graph.batch of graphs whose CSR indices exist (DataLoader workers build them per graph; a (cancer, wild-type) pair
    of batches) = the indices of the union built from scratch, without a sort."""
    import numpy as np
    from immunostruct_amd.graph import CSRIndex, batch, graph
    rs = np.random.RandomState(2)
    parts = []
    for n, e in ((7, 20), (5, 0), (9, 31), (4, 6)):
        g = graph((torch.from_numpy(rs.randint(0, n, size=e)), torch.from_numpy(rs.randint(0, n, size=e))), num_nodes=n)
        g.ndata["x"] = torch.from_numpy(rs.normal(size=(n, 3)).astype(np.float32))
        g.edata["edge_attr"] = torch.from_numpy(rs.rand(e, 2).astype(np.float32))
        g.csr()
        parts.append(g)
    left, right = batch(parts[:2]), batch(parts[2:])
    assert left._csr is not None and right._csr is not None
    union = batch([left, right])                      # batches of batches: the paired forward's merge
    assert union._csr is not None and union.batch_num_nodes().tolist() == [7, 5, 9, 4]
    s, d = union.edges()
    want = CSRIndex(s, d, union.num_nodes())
    for k in ("rowptr_dst", "src_sorted", "dst_sorted", "eperm", "rowptr_src", "pos_by_src"):
        assert torch.equal(getattr(union._csr, k), getattr(want, k)), k
        assert getattr(union._csr, k).dtype == getattr(want, k).dtype, k
    assert torch.equal(union.edge_feat_csr(union.edata["edge_attr"]), union.edata["edge_attr"][want.eperm])


def test_sequence_branch_fork_point_follows_the_batch_size(monkeypatch):
    """models/_core.fork_after_layer: the forked sequence branch starts one EGNN layer earlier for batches whose layer launches are
    long (>= FORK_AUTO_EDGES edges); IMMUNOSTRUCT_FORK_AFTER_LAYER pins it"""
    from immunostruct_amd.models import _core
    monkeypatch.setattr(_core, "FORK_AFTER_LAYER", None)
    assert _core.fork_after_layer(72_313) == 3
    assert _core.fork_after_layer(_core.FORK_AUTO_EDGES) == 2
    assert _core.fork_after_layer(144_700) == 2
    monkeypatch.setattr(_core, "FORK_AFTER_LAYER", 1)
    assert _core.fork_after_layer(72_313) == 1 and _core.fork_after_layer(500_000) == 1


def test_backward_kernel_choice_follows_the_measured_rule(monkeypatch):
    """``functional.use_paired_bwd``: the paired 512-thread backward kernel for Fe <= 1 batches of up to 5.5 node tiles of 16 per
    workgroup slot (round 6's step-level sweep: B <= 224 at 190 nodes per graph), the 256-thread kernel beyond, for 8 edge features,
    with z3 recomputed, and under IMMUNOSTRUCT_BWD_PAIRED=0; fewer slots (reserved CUs of the data-parallel step) move the cut with them"""
    from immunostruct_amd import functional as HF
    assert HF.use_paired_bwd(128 * 190, 1) and HF.use_paired_bwd(224 * 190, 1) and HF.use_paired_bwd(150 * 190, 0)
    assert not HF.use_paired_bwd(256 * 190, 1) and not HF.use_paired_bwd(512 * 190, 1)
    assert not HF.use_paired_bwd(128 * 190, 8)
    assert HF.use_paired_bwd(16 * HF.PAIRED_BWD_MAX_TILES, 1) and not HF.use_paired_bwd(16 * HF.PAIRED_BWD_MAX_TILES + 1, 1)
    monkeypatch.setattr(HF, "RESERVED_CUS", 128)      # half the grid: half the tiles
    assert HF.use_paired_bwd(112 * 190, 1) and not HF.use_paired_bwd(128 * 190, 1)
    monkeypatch.setattr(HF, "RESERVED_CUS", 0)
    monkeypatch.setattr(HF, "SAVE_Z3", False)
    assert not HF.use_paired_bwd(128 * 190, 1)
    monkeypatch.setattr(HF, "SAVE_Z3", True)
    monkeypatch.setattr(HF, "BWD_PAIRED", False)
    assert not HF.use_paired_bwd(128 * 190, 1)

