"""Loader-side augmentations: per-item host functions with the reference's rules
(``data/util_dataloader.py:10-86``, ``data/immmunopred_dataloader.py:83-115,232-271``) and the batched on-device form."""
import random

import numpy as np
import torch

from immunostruct_amd import data as D
from immunostruct_amd.data import augment as A


def dataset(n=6, seed=11):
    return D.SyntheticImmunoDataset(n, seed=seed)


def real_rows(g):
    return (g.ndata["x"][:, :-3].sum(1) > 0).nonzero().flatten()


def test_mask_single_structure_marks_one_real_residue():
    random.seed(0)
    g0 = dataset()[0][0]
    for _ in range(5):
        g = A.SplitDataset._rotated(g0)
        before = g.ndata["x"].clone()
        g, amino = A.mask_single_structure(g)
        changed = (g.ndata["x"] != before).any(1).nonzero().flatten()
        assert changed.numel() == 1 and int(changed) in set(real_rows(g0).tolist())
        node = int(changed)
        assert bool((g.ndata["x"][node, :-3] == 1).all())
        assert amino.shape == (1,) and int(before[node, :-3].argmax()) == int(amino)
        assert torch.equal(g.ndata["x"][node, -3:], before[node, -3:])
    empty = A.SplitDataset._rotated(g0)
    empty.ndata["x"][:, :-3] = 0
    _, amino = A.mask_single_structure(empty)
    assert amino.tolist() == [0] and float(empty.ndata["x"][:, :-3].sum()) == 0      # the reference's fallback


def test_pair_mask_picks_the_same_residue_type():
    random.seed(1)
    ds = D.SyntheticPairedDataset(4, seed=5)
    (ga, gb), _, _, _ = ds[1]
    a, b = A.SplitDataset._rotated(ga), A.SplitDataset._rotated(gb)
    a, b, amino = A.mask_single_structure_pair(a, b)
    for g, orig in ((a, ga), (b, gb)):
        node = (g.ndata["x"][:, :-3].sum(1) > 1).nonzero().flatten()
        assert node.numel() == 1 and int(orig.ndata["x"][int(node), :-3].argmax()) == int(amino)


def test_mask_structure_and_sequence():
    random.seed(2)
    g0, seq, _, _ = dataset()[2]
    g = A.SplitDataset._rotated(g0)
    g, _ = A.mask_single_structure(g)
    g = A.mask_structure(g, 7)
    rows = g.ndata["x"][:, :-3].sum(1)
    assert int((rows > 1).sum()) == 1                                     # the self-supervision node survives
    assert int((rows == 0).sum()) >= int((g0.ndata["x"][:, :-3].sum(1) == 0).sum())
    assert int((rows == 1).sum()) >= real_rows(g0).numel() - 8
    s = A.mask_sequence(seq.clone(), seq[-11:], 9)
    diff = (s != seq).any(1).nonzero().flatten()
    assert diff.numel() <= 9 and (diff.numel() == 0 or int(diff.max()) < len(seq) - 11)
    assert bool((s[diff].argmax(1) == A.PAD_INDEX).all())


def test_split_dataset_rules():
    random.seed(3)
    np.random.seed(3)
    ds = dataset()
    plain = A.SplitDataset(ds, "train")
    assert plain[0][0] is ds[0][0] and len(plain[0]) == 4                # the reference's quirk: the ORIGINAL graph
    ssl = A.SplitDataset(ds, "train", return_amino_acid=True)
    g, seq, y, prop, amino = ssl[0]
    orig = ds[0][0]
    assert g is not orig and amino.shape == (1,)
    rows = real_rows(orig)
    d0 = torch.cdist(orig.ndata["x"][rows, -3:], orig.ndata["x"][rows, -3:], compute_mode="donot_use_mm_for_euclid_dist")
    d1 = torch.cdist(g.ndata["x"][rows, -3:], g.ndata["x"][rows, -3:], compute_mode="donot_use_mm_for_euclid_dist")
    assert torch.allclose(d0, d1, atol=1e-3) and not torch.allclose(orig.ndata["x"][rows, -3:], g.ndata["x"][rows, -3:])
    assert float(orig.ndata["x"][:, :-3].max()) == 1 and int((orig.ndata["x"][:, :-3].sum(1) > 1).sum()) == 0   # untouched
    val = A.SplitDataset(ds, "val", return_amino_acid=True)[0]
    assert val[0] is orig and val[4].tolist() == [0]
    batch = D.collate_amino_acid([ssl[i] for i in range(4)])
    assert len(batch) == 5 and batch[4].shape == (4,) and batch[0].batch_size == 4 and batch[1].shape[0] == 4
    pairs = D.SyntheticPairedDataset(4, seed=5)
    item = A.SplitDataset(pairs, "train", comparative=True, return_amino_acid=True, structure_pad_count=2, sequence_pad_count=3)[0]
    assert len(item[0]) == 2 and len(item[1]) == 2 and item[4].shape == (1,)
    pb = D.collate_amino_acid([item, item])
    assert pb[0][0].batch_size == 2 and pb[4].shape == (2,)
    ext = D.ExtendedDataset(ds, 15)
    assert len(ext) == 15 and ext[13][0] is ds[13 % len(ds)][0]


def test_augment_batch_on_device_form():
    ds = dataset(8, seed=4)
    x0 = torch.cat([ds[i][0].ndata["x"] for i in range(8)]).clone()
    n = ds[0][0].num_nodes()
    x0.view(8, n, -1)[5, :, :-3] = 0                                          # a graph with no real residue
    x = x0.clone()
    gen = torch.Generator().manual_seed(9)
    amino = A.augment_batch_on_device(x, 8, gen, structure_pad_count=3)
    f0, f1 = x0.view(8, n, -1), x.view(8, n, -1)
    for b in range(8):
        marked = (f1[b, :, :-3].sum(1) > 1).nonzero().flatten()
        if b == 5:
            assert marked.numel() == 0 and int(amino[b]) == 0
            continue
        assert marked.numel() == 1
        node = int(marked)
        assert float(f0[b, node, :-3].sum()) == 1 and int(f0[b, node, :-3].argmax()) == int(amino[b])
        blanked = ((f0[b, :, :-3].sum(1) == 1) & (f1[b, :, :-3].sum(1) == 0)).sum()
        assert int(blanked) <= 3
        rows = (f0[b, :, :-3].sum(1) > 0).nonzero().flatten()
        d0, d1 = torch.cdist(f0[b, rows, -3:], f0[b, rows, -3:], compute_mode="donot_use_mm_for_euclid_dist"), torch.cdist(f1[b, rows, -3:], f1[b, rows, -3:], compute_mode="donot_use_mm_for_euclid_dist")
        assert torch.allclose(d0, d1, atol=2e-3)
        assert not torch.allclose(f0[b, rows, -3:], f1[b, rows, -3:])
    q = A._random_orthogonal(16, torch.device("cpu"), gen)
    assert torch.allclose(q.transpose(1, 2) @ q, torch.eye(3).expand(16, 3, 3), atol=1e-5)
    seq = torch.stack([ds[i][1] for i in range(8)]).clone()
    s = A.mask_sequence_on_device(seq.clone(), 5, generator=gen)
    changed = (s != seq).any(2)
    assert int(changed[:, -11:].sum()) == 0 and bool((changed.sum(1) <= 5).all())
    assert bool((s[changed].argmax(1) == A.PAD_INDEX).all())
    # merged pair batch: rows i and b + i are padded at the same positions (data/immmunopred_dataloader.py:216-231)
    base = torch.nn.functional.one_hot(torch.randint(0, 20, (8, seq.shape[1]), generator=torch.Generator().manual_seed(3)), 21).float()
    sp = A.mask_sequence_on_device(base.clone(), 5, generator=gen, pairs=True)
    ch = (sp != base).any(2)
    assert torch.equal(ch[:4], ch[4:]) and bool((ch.sum(1) == 5).all()) and int(ch[:, -11:].sum()) == 0
    assert not torch.equal(ch[0], ch[1])


def test_augment_pair_on_device_rules():
    """merged (cancer; wild-type) batch: one masked residue per member, the SAME type in both, taken among the types the two
    graphs share; a pair without a common type stays unmasked (type 0 reported, the reference's fallback); rotations are
    isometries and differ between the members"""
    pairs = 6
    ca, wt = dataset(pairs, seed=5), dataset(pairs, seed=6)
    n = ca[0][0].num_nodes()
    x0 = torch.cat([ca[i][0].ndata["x"] for i in range(pairs)] + [wt[i][0].ndata["x"] for i in range(pairs)]).clone()
    f0 = x0.view(2 * pairs, n, -1)
    f0[2, :, :-3] = 0
    f0[2, :, 3] = 1                      # pair 2: the cancer graph holds only type 3 ...
    f0[pairs + 2, :, :-3] = 0
    f0[pairs + 2, :, 7] = 1              # ... the wild-type graph only type 7: nothing in common
    x = x0.clone()
    gen = torch.Generator().manual_seed(13)
    amino = A.augment_pair_on_device(x, pairs, gen, structure_pad_count=2)
    f1 = x.view(2 * pairs, n, -1)
    for b in range(pairs):
        mc = (f1[b, :, :-3].sum(1) > 1).nonzero().flatten()
        mw = (f1[pairs + b, :, :-3].sum(1) > 1).nonzero().flatten()
        if b == 2:
            assert mc.numel() == 0 and mw.numel() == 0 and int(amino[b]) == 0
            continue
        assert mc.numel() == 1 and mw.numel() == 1
        tc, tw = int(f0[b, int(mc), :-3].argmax()), int(f0[pairs + b, int(mw), :-3].argmax())
        assert tc == tw == int(amino[b])
        assert float(f0[b, int(mc), :-3].sum()) == 1 and float(f0[pairs + b, int(mw), :-3].sum()) == 1
        for g in (b, pairs + b):
            rows = (f0[g, :, :-3].sum(1) > 0).nonzero().flatten()
            d0 = torch.cdist(f0[g, rows, -3:], f0[g, rows, -3:], compute_mode="donot_use_mm_for_euclid_dist")
            d1 = torch.cdist(f1[g, rows, -3:], f1[g, rows, -3:], compute_mode="donot_use_mm_for_euclid_dist")
            assert torch.allclose(d0, d1, atol=2e-3)
            assert int(((f0[g, :, :-3].sum(1) == 1) & (f1[g, :, :-3].sum(1) == 0)).sum()) <= 2


# ---- the reference's own loader outputs (tests/golden/augment.npz, oracle/make_golden_augment.py): the pin that travels ----
def _golden_augment():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz"))


def _item_seed(kind, idx):
    return 1000 * (1 if kind == "single" else 2) + idx


def test_host_forms_reproduce_the_reference_loader_golden():
    """``SplitDataset`` (train split, ``return_amino_acid``, 7 blanked residues, 9 padded positions) seeded like the reference run
    that wrote the golden: identical graphs, sequences and amino-acid labels, single items and pairs"""
    gold = _golden_augment()
    spc, qpc, pep = int(gold["structure_pads"]), int(gold["sequence_pads"]), int(gold["peptide"])
    ds = D.SyntheticImmunoDataset(int(gold["single_n"]), seed=int(gold["single_seed"]))
    split = A.SplitDataset(ds, "train", return_amino_acid=True, structure_pad_count=spc, sequence_pad_count=qpc, peptide_length=pep)
    for idx in range(len(ds)):
        random.seed(_item_seed("single", idx)); np.random.seed(_item_seed("single", idx))
        g, seq, _, _, amino = split[idx]
        assert np.array_equal(g.ndata["x"].numpy(), gold[f"single/{idx}/x"]) and np.array_equal(seq.numpy(), gold[f"single/{idx}/seq"])
        assert np.array_equal(amino.numpy(), gold[f"single/{idx}/amino"])
    dp = D.SyntheticPairedDataset(int(gold["paired_n"]), seed=int(gold["paired_seed"]))
    split = A.SplitDataset(dp, "train", comparative=True, return_amino_acid=True, structure_pad_count=spc, sequence_pad_count=qpc,
                           peptide_length=pep)
    for idx in range(len(dp)):
        random.seed(_item_seed("paired", idx)); np.random.seed(_item_seed("paired", idx))
        graphs, seqs, _, _, amino = split[idx]
        for k in (0, 1):
            assert np.array_equal(graphs[k].ndata["x"].numpy(), gold[f"paired/{idx}/{k}/x"]), (idx, k)
            assert np.array_equal(seqs[k].numpy(), gold[f"paired/{idx}/{k}/seq"]), (idx, k)
        assert np.array_equal(amino.numpy(), gold[f"paired/{idx}/amino"])


def _device_forms_against_golden(dev):
    gold = _golden_augment()
    spc, qpc, pep = int(gold["structure_pads"]), int(gold["sequence_pads"]), int(gold["peptide"])
    ds = D.SyntheticImmunoDataset(int(gold["single_n"]), seed=int(gold["single_seed"]))
    b = len(ds)
    x = torch.cat([ds[i][0].ndata["x"] for i in range(b)]).to(dev)
    seq = torch.stack([ds[i][1] for i in range(b)]).to(dev)
    T = lambda key, dt: torch.as_tensor(np.stack([gold[f"single/{i}/{key}"] for i in range(b)])).to(dt)
    amino = A.augment_batch_on_device(x, b, structure_pad_count=spc,
                                      picks=dict(rotation=T("rotation", torch.float32)[:, 0], node=T("node", torch.int64), pad_nodes=T("pad_nodes", torch.int64)))
    A.mask_sequence_on_device(seq, qpc, peptide_length=pep, positions=T("positions", torch.int64))
    want = T("x", torch.float32).reshape(-1, x.shape[1])
    assert torch.equal(x[:, :-3].cpu(), want[:, :-3]), "one-hot part: masked / blanked residues"
    assert float((x[:, -3:].cpu() - want[:, -3:]).abs().max()) <= 2e-5 * float(want[:, -3:].abs().max()), "rotated coordinates (fp32 vs the reference's fp64 product)"
    assert torch.equal(amino.cpu(), T("amino", torch.int64).flatten())
    assert torch.equal(seq.cpu(), T("seq", torch.float32))
    # pairs: the merged [cancer; wild-type] batch
    dp = D.SyntheticPairedDataset(int(gold["paired_n"]), seed=int(gold["paired_seed"]))
    b = len(dp)
    P = lambda key, k, dt: torch.as_tensor(np.stack([gold[f"paired/{i}/{k}/{key}"] for i in range(b)])).to(dt)
    x2 = torch.cat([dp[i][0][k].ndata["x"] for k in (0, 1) for i in range(b)]).to(dev)
    seq2 = torch.cat([torch.stack([dp[i][1][k] for i in range(b)]) for k in (0, 1)]).to(dev)
    picks = dict(rotation=torch.cat([P("rotation", 0, torch.float32), P("rotation", 1, torch.float32)]),
                 node_c=P("node", 0, torch.int64), node_w=P("node", 1, torch.int64),
                 pad_nodes=torch.cat([P("pad_nodes", 0, torch.int64), P("pad_nodes", 1, torch.int64)]))
    amino = A.augment_pair_on_device(x2, b, structure_pad_count=spc, picks=picks)
    pos = torch.as_tensor(np.stack([gold[f"paired/{i}/positions"] for i in range(b)]))
    A.mask_sequence_on_device(seq2, qpc, peptide_length=pep, pairs=True, positions=pos)
    want = torch.cat([P("x", 0, torch.float32), P("x", 1, torch.float32)]).reshape(-1, x2.shape[1])
    assert torch.equal(x2[:, :-3].cpu(), want[:, :-3]), "pairs: one-hot part"
    assert float((x2[:, -3:].cpu() - want[:, -3:]).abs().max()) <= 2e-5 * float(want[:, -3:].abs().max())
    assert torch.equal(amino.cpu(), torch.as_tensor(np.stack([gold[f"paired/{i}/amino"] for i in range(b)])).flatten())
    assert torch.equal(seq2.cpu(), torch.cat([P("seq", 0, torch.float32), P("seq", 1, torch.float32)]))


def test_device_forms_apply_the_reference_rules_to_the_reference_picks():
    """the whole-batch forms draw differently by construction; given the picks the reference's loader made (rotation, self-supervision
    node, blanked nodes, padded positions) they must produce what it produced"""
    _device_forms_against_golden(torch.device("cpu"))


import pytest  # noqa: E402


@pytest.mark.gpu
def test_device_forms_on_the_gpu_box_against_the_reference_golden(cuda_device):
    _device_forms_against_golden(cuda_device)
