"""The table side of the reference's dataset construction: property tables + HLA table -> the
per-structure labels that ``data.convert_pyg_directory`` attaches to the packed graphs.

What the reference does, per start of every run (``data/preprocess.py:45-145,188-300``, called from
``data/immmunopred_dataloader.py:28-35,137-154``):

* ``preprocess_properties(table, cancer)``    rows without a foreignness score are dropped; the key of a row is
  ``peptide + allele`` (IEDB) or ``mut_pep + "HLA-A*02:01"``-style allele rebuilt from ``HLA-A0201`` (cancer); three
  dictionaries key -> smoothed foreignness / (Mprop1, Mprop2) / immunogenicity; LAST row wins for a repeated key;
* ``preprocess_properties_cancer_wt(cancer, wt)``    the same keys for both tables, one row per
  ``(mut_pep, wt_pep, allele)`` (duplicates: keep the highest foreignness if immunogenic, the lowest otherwise), inner
  join of the two tables on ``(mut_pep, wt_pep, allele, immunogenicity)``;
* ``preprocess_hla(keys, hla_csv)``    key -> (HLA sequence + peptide, structure name = last 99 residues + "_" + first 5 hex
  digits of the SHA-1 of the full sequence, peptide);
* table <-> structure matching by that structure name (both directions), and for pairs the cancer <-> wild-type cross check.

Here these are plain functions over pandas frames and NAME lists (no graph objects: the graphs stay in the packed file),
written from the rules above and pinned to the reference's functions on the shipped tables
(``tests/test_tables.py``, run where /root/reference exists).  One quirk is kept on purpose: the reference builds the
wild-type key's last allele field from the CANCER frame's column (index-aligned), ``data/preprocess.py:78``.
"""
from __future__ import annotations

import hashlib

import numpy as np
import pandas as pd

__all__ = ["structure_name", "preprocess_properties", "preprocess_properties_cancer_wt", "preprocess_hla", "match_structures",
           "match_pairs", "labels_from_tables", "paired_labels_from_tables", "item_rows_from_tables", "paired_item_rows_from_tables"]


def structure_name(full_sequence):
    """name of the AlphaFold structure of a (HLA sequence + peptide) string: its last 99 residues + "_" + 5 hex digits of
    its SHA-1 (``data/utils.py:157-158``, ``data/preprocess.py:137-139``) -- the part of ``graph.name`` after "Immuno"."""
    return full_sequence[-99:] + "_" + hashlib.sha1(full_sequence.encode()).hexdigest()[:5]


def _allele_field(frame):
    """"HLA-A0201" -> "HLA-A*02:01" (first / second part around the dash, as the reference splits it)"""
    parts = frame["allele"].str.split("-", expand=True)
    a1, a2 = parts[0], parts[1]
    return a1, a2


def _cancer_key(frame, peptide_column, tail_from=None):
    a1, a2 = _allele_field(frame)
    tail = (a2 if tail_from is None else tail_from).str[3:]
    return frame[peptide_column] + (a1 + "-" + a2.str[0] + "*" + a2.str[1:3] + ":" + tail)


def preprocess_properties(table, cancer=False):
    """-> (foreignness dict, (Mprop1, Mprop2) dict, immunogenicity dict, list of keys in table order)"""
    df = table if isinstance(table, pd.DataFrame) else pd.read_table(table)
    if cancer:
        df = df.dropna(subset="foreign").copy()
        df["pep_pair"] = _cancer_key(df, "mut_pep")
    else:
        df = df.dropna(subset="Foreignness_Score").copy()
        df["pep_pair"] = df["peptide"] + df["allele"]
    keys = df["pep_pair"].tolist()
    f_dict = dict(zip(keys, df["smoothed_foreign"]))
    fp2_dict = dict(zip(keys, zip(df["Mprop1"], df["Mprop2"])))
    imm_dict = dict(zip(keys, df["immunogenicity"]))
    return f_dict, fp2_dict, imm_dict, keys


def _one_row_per_triplet(df):
    """duplicates of (mut_pep, wt_pep, allele): keep the row with the highest foreignness when immunogenic, the lowest
    otherwise (first such row on ties, as ``argmax`` / ``argmin``); contradictory immunogenicity is an error.
    INTENTIONAL difference from ``__dedup_property_df`` (``data/preprocess.py:92-130``): the reference collects the duplicates'
    POSITIONS and then looks them up as index LABELS (``df.loc[duplicate_rows]``), which is the same thing only while the frame's
    index is 0..n-1; after ``dropna`` removed a row it reads (or drops) other rows, or raises a KeyError.  Here the rows are
    addressed by label throughout -- the rule in the reference's own docstring; identical on the shipped tables
    (``tests/test_tables.py``)."""
    key = ["mut_pep", "wt_pep", "allele"]
    if df.groupby(key, sort=False)["immunogenicity"].nunique().max() > 1:
        raise AssertionError("same ('mut_pep', 'wt_pep', 'allele') but different immunogenicity")
    fcol = "smoothed_foreign" if "smoothed_foreign" in df else "foreign"
    drop = []
    for _, rows in df.groupby(key, sort=False).indices.items():
        if len(rows) < 2:
            continue
        labels = df.index[rows]
        imm = df.loc[labels[0], "immunogenicity"]
        if imm not in (0, 1):
            raise AssertionError("immunogenicity must be 0 or 1")
        vals = df.loc[labels, fcol].to_numpy()
        keep = labels[int(vals.argmax() if imm == 1 else vals.argmin())]
        drop.extend(lab for lab in labels if lab != keep)
    return df.drop(index=drop) if drop else df


def preprocess_properties_cancer_wt(table_cancer, table_wt):
    """-> one frame, a row per (cancer, wild-type) pair: mut_pep, wt_pep, allele, immunogenicity, pep_pair_cancer,
    pep_pair_wt, smoothed_foreign, Mprop1, Mprop1_wt, Mprop2, Mprop2_wt"""
    c = table_cancer if isinstance(table_cancer, pd.DataFrame) else pd.read_table(table_cancer)
    w = table_wt if isinstance(table_wt, pd.DataFrame) else pd.read_table(table_wt)
    c = c.dropna(subset="foreign").copy()
    w = w.dropna(subset="foreign").copy()
    c["pep_pair_cancer"] = _cancer_key(c, "mut_pep")
    # (reference quirk, kept: the last allele field of the wild-type key comes from the CANCER frame, aligned by row label)
    _, a2_cancer = _allele_field(c)
    w["pep_pair_wt"] = _cancer_key(w, "wt_pep", tail_from=a2_cancer.reindex(w.index))
    short_c = _one_row_per_triplet(c[["mut_pep", "wt_pep", "allele", "immunogenicity", "pep_pair_cancer", "smoothed_foreign", "Mprop1", "Mprop2"]])
    short_w = _one_row_per_triplet(w[["mut_pep", "wt_pep", "allele", "immunogenicity", "foreign", "pep_pair_wt", "Mprop1_wt", "Mprop2_wt"]])
    both = pd.merge(short_c, short_w, on=["mut_pep", "wt_pep", "allele", "immunogenicity"])
    both = both[["mut_pep", "wt_pep", "allele", "immunogenicity", "pep_pair_cancer", "pep_pair_wt", "smoothed_foreign",
                 "Mprop1", "Mprop1_wt", "Mprop2", "Mprop2_wt"]]
    if not (len(short_c) == len(short_w) == len(both)):
        raise AssertionError("cancer and wild-type tables do not pair up one to one")
    return both


def preprocess_hla(keys, hla_path):
    """key ("<peptide>HLA-A*02:01") -> (HLA sequence + peptide, structure name, peptide)"""
    hla = hla_path if isinstance(hla_path, pd.DataFrame) else pd.read_csv(hla_path)
    seq_of = dict(zip(hla["allele"], hla["seqs"]))
    out = {}
    for key in keys:
        pep, allele = key.split("HLA-")
        full = seq_of["HLA-" + allele] + pep
        out[key] = (full, structure_name(full), pep)
    return out


def match_structures(name_mapper, structure_names):
    """the two-way filter of ``preprocess_sequence_graph`` (``data/preprocess.py:147-171``) on names only:
    -> (name_mapper restricted to keys whose structure exists, the structure names that some key refers to, in input order)"""
    have = set(structure_names)
    kept = {k: v for k, v in name_mapper.items() if v[1] in have}
    wanted = set(v[1] for v in kept.values())
    return kept, [s for s in structure_names if s in wanted]


def match_pairs(combined, mapper_cancer, mapper_wt, names_cancer, names_wt):
    """``preprocess_sequence_graph_cancer_wt`` (``data/preprocess.py:188-262``) on names only: drop keys without a structure,
    then pairs of which one member is gone; -> (combined frame restricted to the surviving pairs, both mappers)"""
    mapper_cancer, _ = match_structures(mapper_cancer, names_cancer)
    mapper_wt, _ = match_structures(mapper_wt, names_wt)
    c2w = dict(zip(combined["pep_pair_cancer"], combined["pep_pair_wt"]))
    w2c = dict(zip(combined["pep_pair_wt"], combined["pep_pair_cancer"]))
    mapper_cancer = {k: v for k, v in mapper_cancer.items() if c2w[k] in mapper_wt}
    mapper_wt = {k: v for k, v in mapper_wt.items() if w2c[k] in mapper_cancer}
    keep = combined["pep_pair_cancer"].isin(mapper_cancer.keys()) & combined["pep_pair_wt"].isin(mapper_wt.keys())
    return combined[keep], mapper_cancer, mapper_wt


def item_rows_from_tables(property_path, hla_path, structure_names, cancer=False):
    """One row PER DATASET ITEM, in the reference's order (``ImmunoPredDataset.organize``, ``data/immmunopred_dataloader.py:38-60``:
    one item per table key that has a structure; two keys that map to the same structure -- alleles with identical sequences --
    are two items sharing a graph, each with its own key's values):
    ``[(structure name, (full sequence, Mprop1, Mprop2, immunogenicity, smoothed foreignness, peptide)), ...]``"""
    f_dict, fp2_dict, imm_dict, keys = preprocess_properties(property_path, cancer)
    mapper, _ = match_structures(preprocess_hla(keys, hla_path), structure_names)
    rows = []
    for key, (full, name, pep) in mapper.items():
        m1, m2 = fp2_dict[key]
        rows.append((name, (full, float(m1), float(m2), float(imm_dict[key]), float(f_dict[key]), pep)))
    return rows


def paired_item_rows_from_tables(path_cancer, path_wt, hla_path, names_cancer, names_wt):
    """One row per (cancer, wild-type) pair of the joined table, in its row order (``ImmunoPredDatasetComparative.organize``,
    ``data/immmunopred_dataloader.py:156-190``): ``[(name_c, row_c, name_w, row_w), ...]`` with the label tuples of
    :func:`item_rows_from_tables`; pairs may share a structure, every pair keeps ITS row's values (the wild-type member is
    labelled non-immunogenic with the table's minimal foreignness)."""
    combined = preprocess_properties_cancer_wt(path_cancer, path_wt)
    mc = preprocess_hla(combined["pep_pair_cancer"], hla_path)
    mw = preprocess_hla(combined["pep_pair_wt"], hla_path)
    combined, mc, mw = match_pairs(combined, mc, mw, names_cancer, names_wt)
    fmin = float(combined["smoothed_foreign"].min()) if len(combined) else float("nan")
    out = []
    for row in combined.itertuples(index=False):
        fc, nc, pep_c = mc[row.pep_pair_cancer]
        fw, nw, pep_w = mw[row.pep_pair_wt]
        out.append((nc, (fc, float(row.Mprop1), float(row.Mprop2), float(row.immunogenicity), float(row.smoothed_foreign), pep_c),
                    nw, (fw, float(row.Mprop1_wt), float(row.Mprop2_wt), 0.0, fmin, pep_w)))
    return out


def labels_from_tables(property_path, hla_path, structure_names, cancer=False):
    """``labels[structure name] = (full sequence, Mprop1, Mprop2, immunogenicity, smoothed foreignness, peptide)`` for
    ``data.convert_pyg_directory(..., labels=labels)`` -- what ``ImmunoPredDataset.__init__`` + ``organize`` assemble
    (``data/immmunopred_dataloader.py:28-60``); also returns the keys in dataset order"""
    f_dict, fp2_dict, imm_dict, keys = preprocess_properties(property_path, cancer)
    mapper, _ = match_structures(preprocess_hla(keys, hla_path), structure_names)
    labels = {}
    for key, (full, name, pep) in mapper.items():
        m1, m2 = fp2_dict[key]
        labels[name] = (full, float(m1), float(m2), float(imm_dict[key]), float(f_dict[key]), pep)
    return labels, list(mapper.keys())


def paired_labels_from_tables(path_cancer, path_wt, hla_path, names_cancer, names_wt):
    """labels of both members of every pair, in the row order of the joined table (``ImmunoPredDatasetComparative``,
    ``data/immmunopred_dataloader.py:137-190``): the wild-type member is labelled non-immunogenic with the table's minimal
    foreignness.  -> (labels_cancer, labels_wt, list of (structure name cancer, structure name wild-type))"""
    combined = preprocess_properties_cancer_wt(path_cancer, path_wt)
    mc = preprocess_hla(combined["pep_pair_cancer"], hla_path)
    mw = preprocess_hla(combined["pep_pair_wt"], hla_path)
    combined, mc, mw = match_pairs(combined, mc, mw, names_cancer, names_wt)
    fmin = float(combined["smoothed_foreign"].min()) if len(combined) else float("nan")
    lab_c, lab_w, pairs = {}, {}, []
    for row in combined.itertuples(index=False):
        fc, nc, pep_c = mc[row.pep_pair_cancer]
        fw, nw, pep_w = mw[row.pep_pair_wt]
        lab_c[nc] = (fc, float(row.Mprop1), float(row.Mprop2), float(row.immunogenicity), float(row.smoothed_foreign), pep_c)
        lab_w[nw] = (fw, float(row.Mprop1_wt), float(row.Mprop2_wt), 0.0, fmin, pep_w)
        pairs.append((nc, nw))
    return lab_c, lab_w, pairs
