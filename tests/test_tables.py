"""``immunostruct_amd.data.tables`` against the reference's own table joins (``data/preprocess.py:45-145,188-300``) run in
this container under ``oracle/shims.py`` on the tables the reference ships (``data/cedar_data_final_with_mprop1_mprop2_v2.txt``,
``data/HLA_27_seqs_csv.csv``).  The wild-type table and the graph directories were never shipped
(``.MISSING_LARGE_BLOBS``): the wild-type table is derived here from the cancer table, the structures are name-only stubs.
Skipped where /root/reference is absent (the GPU box)."""
import importlib
import os
from types import SimpleNamespace

import numpy as np
import pandas as pd
import pytest
import torch

from immunostruct_amd.data import tables as T
from oracle import shims

pytestmark = pytest.mark.skipif(not shims.reference_available(), reason="needs /root/reference")
CANCER = os.path.join(shims.REFERENCE_ROOT, "data", "cedar_data_final_with_mprop1_mprop2_v2.txt")
HLA = os.path.join(shims.REFERENCE_ROOT, "data", "HLA_27_seqs_csv.csv")


@pytest.fixture(scope="module")
def ref():
    shims.install()
    return importlib.import_module("data.preprocess")


def test_cancer_property_table_matches_reference(ref):
    f_r, fp_r, imm_r, keys_r = ref.preprocess_properties(CANCER, cancer=True)
    f, fp, imm, keys = T.preprocess_properties(CANCER, cancer=True)
    assert keys == keys_r and len(keys) > 2000
    assert f == f_r and fp == fp_r and imm == imm_r


def test_iedb_style_table_matches_reference(ref, tmp_path):
    """the IEDB table was never shipped: same columns fabricated from the cancer rows (some without a foreignness score)"""
    df = pd.read_table(CANCER).head(300)
    a2 = df["allele"].str.split("-", expand=True)[1]
    table = pd.DataFrame({"peptide": df["mut_pep"], "allele": "HLA-" + a2.str[0] + "*" + a2.str[1:3] + ":" + a2.str[3:],
                          "Foreignness_Score": df["foreign"].where(np.arange(300) % 7 != 0), "smoothed_foreign": df["smoothed_foreign"],
                          "Mprop1": df["Mprop1"], "Mprop2": df["Mprop2"], "immunogenicity": df["immunogenicity"]})
    path = tmp_path / "iedb_like.txt"
    table.to_csv(path, sep="\t", index=False)
    got, want = T.preprocess_properties(str(path)), ref.preprocess_properties(str(path))
    assert got[3] == want[3] and got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
    assert T.preprocess_hla(got[3], HLA) == ref.preprocess_hla(want[3], HLA)


def test_hla_names_match_reference(ref):
    keys = T.preprocess_properties(CANCER, cancer=True)[3]
    got, want = T.preprocess_hla(keys, HLA), ref.preprocess_hla(keys, HLA)
    assert got == want
    full, name, pep = next(iter(got.values()))
    assert name == T.structure_name(full) and full.endswith(pep) and len(name) == 99 + 6


def _wildtype_table(tmp_path):
    """a wild-type table that pairs with the shipped cancer table: same triplets / immunogenicity, its own scores; a few
    duplicated triplets with different foreignness (the de-duplication rule) come with the cancer table itself"""
    df = pd.read_table(CANCER)
    rng = np.random.RandomState(3)
    wt = pd.DataFrame({"wt_pep": df["wt_pep"], "mut_pep": df["mut_pep"], "allele": df["allele"], "immunogenicity": df["immunogenicity"],
                       "foreign": df["foreign"], "Mprop1_wt": df["Mprop1"] * 0.5 + rng.uniform(size=len(df)) * 0.1,
                       "Mprop2_wt": df["Mprop2"] * 0.25})
    path = tmp_path / "wt.txt"
    wt.to_csv(path, sep="\t", index=False)
    return str(path)


def test_pair_table_matches_reference(ref, tmp_path):
    wt = _wildtype_table(tmp_path)
    want = ref.preprocess_properties_cancer_wt(CANCER, wt)
    got = T.preprocess_properties_cancer_wt(CANCER, wt)
    pd.testing.assert_frame_equal(got.reset_index(drop=True), want.reset_index(drop=True))
    assert len(got) < len(pd.read_table(CANCER).dropna(subset="foreign"))      # the shipped table does hold duplicated triplets


def _stub(name, width=20):
    return SimpleNamespace(name="prefixImmuno" + name, x=torch.zeros(3, width), coords=torch.zeros(3, 3), num_nodes=3)


def test_structure_matching_matches_reference(ref):
    f_d, fp_d, imm_d, keys = T.preprocess_properties(CANCER, cancer=True)
    mapper = T.preprocess_hla(keys, HLA)
    names = sorted(set(v[1] for v in mapper.values()))
    present = names[::3] + ["A" * 99 + "_00000"]                 # a third of the structures exist, plus one nobody refers to
    kept, used = T.match_structures(dict(mapper), present)
    ref_mapper, ref_graphs = ref.preprocess_sequence_graph(dict(mapper), [_stub(n) for n in present], imm_d, f_d)
    assert kept == ref_mapper and set(used) == set(ref_graphs.keys())
    labels, order = T.labels_from_tables(CANCER, HLA, present, cancer=True)
    assert order == list(ref_mapper.keys()) and set(labels) == set(ref_graphs.keys())
    for key, (full, name, _pep) in ref_mapper.items():
        g = ref_graphs[name]
        assert labels[name][0] == full
        assert labels[name][3] == pytest.approx(float(g.y[0])) and labels[name][4] == pytest.approx(float(g.y[1]))
        assert labels[name][1:3] == tuple(float(v) for v in fp_d[key])


def test_pair_matching_matches_reference(ref, tmp_path):
    wt = _wildtype_table(tmp_path)
    combined = T.preprocess_properties_cancer_wt(CANCER, wt)
    mc, mw = T.preprocess_hla(combined["pep_pair_cancer"], HLA), T.preprocess_hla(combined["pep_pair_wt"], HLA)
    names_c = sorted(set(v[1] for v in mc.values()))
    names_w = sorted(set(v[1] for v in mw.values()))
    # every cancer structure, two thirds of the wild-type ones: the cross check drops the pairs whose wild-type member is gone.
    # (Subsets that remove ONE of several cancer partners of a shared wild-type peptide make the reference itself fail with a
    #  KeyError at data/preprocess.py:276 -- its wild-type -> cancer map keeps one partner per wild-type key.)
    have_c, have_w = names_c, [n for i, n in enumerate(names_w) if i % 3]
    got_df, got_c, got_w = T.match_pairs(combined, dict(mc), dict(mw), have_c, have_w)
    want_df, want_c, want_w, gm_c, gm_w = ref.preprocess_sequence_graph_cancer_wt(
        combined.copy(), dict(mc), dict(mw), [_stub(n) for n in have_c], [_stub(n) for n in have_w])
    assert got_c == want_c and got_w == want_w and len(got_c) > 50
    pd.testing.assert_frame_equal(got_df.reset_index(drop=True), want_df.reset_index(drop=True))
    lab_c, lab_w, pairs = T.paired_labels_from_tables(CANCER, wt, HLA, have_c, have_w)
    assert len(pairs) == len(want_df)
    fmin = float(want_df["smoothed_foreign"].min())
    for (nc, nw), row in zip(pairs, want_df.itertuples(index=False)):
        assert nc == want_c[row.pep_pair_cancer][1] and nw == want_w[row.pep_pair_wt][1]
        assert lab_c[nc][3] == float(row.immunogenicity) and lab_w[nw][3] == 0.0 and lab_w[nw][4] == fmin
        assert float(gm_c[nc].y[1]) == pytest.approx(lab_c[nc][4])
