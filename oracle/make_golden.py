"""Generate the golden vectors under ``tests/golden/`` FROM THE REFERENCE ITSELF.

Run in the build container only (needs ``/root/reference``):

    python -m oracle.make_golden

The reference's own ``models/*.py``, ``utils/loss.py`` and
``utils/contrastive.py`` are imported unchanged (``oracle/shims.py``) and run
on CPU fp32 on seeded synthetic inputs (``immunostruct_amd/synthetic.py``) with
deterministic weights (``synthetic.det_state_dict``), fixed ``eps`` for the
reparameterisation (``torch.randn_like`` is patched for the call) and eval-mode
dropout.  Only data is written: inputs are regenerated from the recorded seeds
(their checksums are stored), expected outputs are stored as float32 arrays.

The EGNNConv / dgl.batch / global pooling arithmetic inside those runs comes
from ``oracle/graph_ref.py`` (third-party code absent from the reference:
parity unpinned for those operators, see ``oracle/__init__.py``).
"""
from __future__ import annotations

import hashlib
import os
import sys
import unittest.mock as mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from immunostruct_amd import synthetic  # noqa: E402
from oracle import functional_ref as FR  # noqa: E402
from oracle import graph_ref, shims  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
BATCH = 16
VAE_IN = synthetic.SEQ_LEN * synthetic.SEQ_ALPHABET
RECON_COLS = slice(0, None, 37)
ATTN_ROWS = [0, 57, 189]

FORWARD_VARIANTS = ["HybridModel", "HybridModelv2", "HybridModelv2_SSL", "HybridModelv2_Comparative",
                    "HybridModel_Comparative", "StructureModel", "StructureModelv2", "DualModel",
                    "SequenceFpModel"]


def checksum(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def ref_graph(raw):
    g = graph_ref.RefGraph(raw.src, raw.dst, raw.num_nodes, raw.batch_num_nodes)
    g.ndata["x"] = torch.from_numpy(raw.x)
    g.edata["edge_attr"] = torch.from_numpy(raw.edge_attr)
    return g


def make_eps(seed, batch=BATCH, latent=32):
    return np.random.RandomState(seed).normal(size=(batch, latent)).astype(np.float32)


def build_reference_model(model_map, name, seed, vae_in=VAE_IN, **kw):
    model = model_map[name](vae_input_dim=vae_in, device="cpu", **kw)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synthetic.det_state_dict(shapes, seed=seed).items()}
    model.load_state_dict(sd)
    model.eval()
    return model, sd, shapes


def run_with_eps(fn, eps_list):
    """Call ``fn`` with torch.randn_like returning the given eps tensors in order."""
    it = iter(eps_list)
    with mock.patch("torch.randn_like", lambda t: next(it).to(t.dtype)):
        return fn()


def golden_forward(model_map, out):
    raw = synthetic.make_batch(BATCH, seed=1)
    g = ref_graph(raw)
    seq = torch.from_numpy(raw.one_hot_sequence())
    prop = torch.from_numpy(raw.prop)
    eps = torch.from_numpy(make_eps(11))
    out["fwd/input_checksum"] = np.frombuffer(bytes.fromhex(checksum(raw.x, raw.src, raw.dst, raw.edge_attr, raw.seq_tokens, raw.prop)), dtype=np.uint8)
    for name in FORWARD_VARIANTS:
        kws = [dict()]
        if "Comparative" in name:
            kws = [dict(use_wt_for_downstream=True), dict(use_wt_for_downstream=False)]
        for kw in kws:
            tag = name + ("" if not kw else f"/wt{int(kw['use_wt_for_downstream'])}")
            model, sd, _ = build_reference_model(model_map, name, seed=3, **kw)
            with torch.no_grad():
                res = run_with_eps(lambda: model(g, seq, prop), [eps])
                emb = run_with_eps(lambda: model(g, seq, prop, return_embedding=True), [eps])
                att = run_with_eps(lambda: model(g, seq, prop, return_attention=True), [eps])
            if torch.is_tensor(res[0]):
                out[f"fwd/{tag}/recon_x_cols"] = res[0][:, RECON_COLS].numpy()
                out[f"fwd/{tag}/mu"] = res[1].numpy()
                out[f"fwd/{tag}/logvar"] = res[2].numpy()
            out[f"fwd/{tag}/final_output"] = res[3].numpy()
            if len(res) == 5:
                out[f"fwd/{tag}/node_prediction"] = res[4].numpy()
            if torch.is_tensor(emb[0]) and emb[0].shape[1] == 64:
                out[f"fwd/{tag}/x_gat_node"] = emb[0].numpy()
            if torch.is_tensor(att[0]) and att[0].dim() >= 3:
                out[f"fwd/{tag}/attention_rows"] = att[0][..., ATTN_ROWS, :].numpy()
            # self-check: the functional oracle must agree exactly with the reference classes
            kwf = dict(use_wt_for_downstream=kw.get("use_wt_for_downstream", True))
            mine = FR.as_reference_tuple(name, FR.forward(name, sd, g, seq, prop, eps=eps, **kwf))
            for a, b in zip(res, mine):
                if torch.is_tensor(a):
                    assert torch.equal(a, b.detach()), f"oracle != reference for {tag}"


PEP_IN = (synthetic.SEQ_LEN - synthetic.HLA_LEN) * synthetic.SEQ_ALPHABET      # 11 x 21 = 231
PEPTIDE_VARIANTS = ["HybridModelv2", "SequenceFpModel", "SequenceModel", "HybridModelv2_Comparative"]


def golden_forward_peptide(model_map, Losses, out):
    """The reference's DEFAULT input width (no ``--full-sequence``): ``vae_input_dim = 11 * 21`` and the sequence tensor is
    the padded peptide's one-hot (``train_IEDB_wFT.py:59-60``, ``data/util_dataloader.py:52-66``).  Forward outputs, the
    regression loss and a few parameter gradients of one train step."""
    raw = synthetic.make_batch(BATCH, seed=1)
    g = ref_graph(raw)
    seq = torch.from_numpy(raw.one_hot_sequence()[:, synthetic.HLA_LEN:]).contiguous()
    prop = torch.from_numpy(raw.prop)
    eps = torch.from_numpy(make_eps(11))
    y = torch.from_numpy(raw.y_reg)
    assert seq.shape[1:] == (11, 21)
    for name in PEPTIDE_VARIANTS:
        tag = f"pep/{name}"
        model, sd, _ = build_reference_model(model_map, name, seed=3, vae_in=PEP_IN)
        for p in model.parameters():
            p.requires_grad_(True)
        res = run_with_eps(lambda: model(g, seq, prop), [eps])
        out[f"{tag}/recon_x"] = res[0].detach().numpy()
        out[f"{tag}/mu"] = res[1].detach().numpy()
        out[f"{tag}/logvar"] = res[2].detach().numpy()
        out[f"{tag}/final_output"] = res[3].detach().numpy()
        loss = Losses(PEP_IN, {0: 81.0, 1: 19.0}, sequence=True).regression_loss(res[0], seq, res[1], res[2], res[3], y)
        loss.backward()
        out[f"{tag}/loss"] = np.float32(loss.item())
        named = dict(model.named_parameters())
        for key in ("vae_fc1.weight", "vae_fc4.bias", "vae_fc21.weight"):
            out[f"{tag}/grad/{key}"] = named[key].grad.numpy().copy()
        mine = FR.as_reference_tuple(name, FR.forward(name, sd, g, seq, prop, eps=eps))
        for a, b in zip(res, mine):
            if torch.is_tensor(a):
                assert torch.equal(a.detach(), b.detach()), f"oracle != reference for {tag}"


def golden_comparative(model_map, Losses, PCL, out):
    raw_c = synthetic.make_batch(BATCH, seed=21)
    raw_w = synthetic.make_batch(BATCH, seed=22)
    gc, gw = ref_graph(raw_c), ref_graph(raw_w)
    sc, sw = torch.from_numpy(raw_c.one_hot_sequence()), torch.from_numpy(raw_w.one_hot_sequence())
    pc, pw = torch.from_numpy(raw_c.prop), torch.from_numpy(raw_w.prop)
    eps_c, eps_w = torch.from_numpy(make_eps(31)), torch.from_numpy(make_eps(32))
    y_bin = torch.from_numpy(raw_c.y_bin)
    assert y_bin.unique().numel() == 2
    y_reg = torch.from_numpy(raw_c.y_reg)
    for name in ["HybridModelv2_Comparative", "HybridModel_Comparative"]:
        for wt in (True, False):
            tag = f"cmp/{name}/wt{int(wt)}"
            model, sd, _ = build_reference_model(model_map, name, seed=5, use_wt_for_downstream=wt)
            for p in model.parameters():
                p.requires_grad_(True)
            losses = Losses(VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
            pcl = PCL(embedding_dim=104)
            pshapes = {k: tuple(v.shape) for k, v in pcl.state_dict().items()}
            psd = {k: torch.from_numpy(v) for k, v in synthetic.det_state_dict(pshapes, seed=9).items()}
            pcl.load_state_dict(psd)
            res = run_with_eps(lambda: model.forward_comparative((gc, gw), (sc, sw), (pc, pw)), [eps_c, eps_w])
            emb, rec, mu, lv, final = res
            l_c = losses.BCE_loss(rec[0], sc, mu[0], lv[0], final, y_bin)
            l_w = losses.BCE_loss(rec[1], sw, mu[1], lv[1], final, y_bin)
            l_con = pcl(emb[0], emb[1], y_bin)
            loss = (l_c + l_w) / 2 + 0.01 * l_con  # procedures/train.py:107-118
            loss.backward()
            out[f"{tag}/emb_cancer"] = emb[0].detach().numpy()
            out[f"{tag}/emb_wt"] = emb[1].detach().numpy()
            out[f"{tag}/final_output"] = final.detach().numpy()
            out[f"{tag}/bce_cancer"] = np.float32(l_c.item())
            out[f"{tag}/bce_wt"] = np.float32(l_w.item())
            out[f"{tag}/contrastive"] = np.float32(l_con.item())
            out[f"{tag}/loss"] = np.float32(loss.item())
            named = dict(model.named_parameters())
            for key in ["GCN_layers.0.edge_mlp.0.weight", "GCN_layers.3.coord_mlp.2.weight",
                        "GCN_layers.5.node_mlp.2.bias", "vae_fc21.weight", "classifier.1.weight",
                        "property_embedding.0.weight"]:
                out[f"{tag}/grad/{key}"] = named[key].grad.numpy().copy()
            # self-check of the functional oracle (value path)
            o = FR.forward_comparative(name, sd, (gc, gw), (sc, sw), (pc, pw), (eps_c, eps_w), use_wt_for_downstream=wt)
            assert torch.equal(o["final_output"], final.detach())
            lc2 = FR.paired_contrastive_loss(psd, o["embeddings"][0], o["embeddings"][1], y_bin)
            assert abs(float(lc2) - l_con.item()) <= 1e-5 * abs(l_con.item()), (float(lc2), l_con.item())
            if name == "HybridModelv2_Comparative" and wt:
                # the degenerate-target early-outs (utils/contrastive.py:38-43)
                out["cmp/contrastive_all_equal"] = np.float32(pcl(emb[0].detach(), emb[1].detach(), torch.ones(BATCH)))
                out["cmp/contrastive_continuous"] = np.float32(pcl(emb[0].detach(), emb[1].detach(), y_reg))


def golden_losses(Losses, PCL, out):
    rng = np.random.RandomState(77)
    b = BATCH
    recon = torch.from_numpy(rng.normal(size=(b, VAE_IN)).astype(np.float32) * 0.3).requires_grad_(True)
    x = torch.from_numpy(synthetic.make_batch(b, seed=4).one_hot_sequence())
    mu = torch.from_numpy(rng.normal(size=(b, 32)).astype(np.float32)).requires_grad_(True)
    lv = torch.from_numpy(rng.normal(size=(b, 32)).astype(np.float32) * 0.5).requires_grad_(True)
    logit = torch.from_numpy(rng.normal(size=(b, 1)).astype(np.float32)).requires_grad_(True)
    y_reg = torch.from_numpy(rng.uniform(-1, 1, size=(b,)).astype(np.float32))
    y_bin = torch.from_numpy((rng.uniform(size=(b,)) < 0.3).astype(np.float32))
    for seq_flag in (True, False):
        losses = Losses(VAE_IN, {0: 81.0, 1: 19.0}, sequence=seq_flag)
        for kind, fn, y in (("regression", losses.regression_loss, y_reg), ("bce", losses.BCE_loss, y_bin)):
            for t in (recon, mu, lv, logit):
                t.grad = None
            val = fn(recon, x, mu, lv, logit, y)
            val.backward()
            tag = f"loss/{kind}/seq{int(seq_flag)}"
            out[f"{tag}/value"] = np.float32(val.item())
            out[f"{tag}/grad_logit"] = logit.grad.numpy().copy()
            if seq_flag:
                out[f"{tag}/grad_recon_cols"] = recon.grad[:, RECON_COLS].numpy().copy()
                out[f"{tag}/grad_mu"] = mu.grad.numpy().copy()
                out[f"{tag}/grad_logvar"] = lv.grad.numpy().copy()
    # paired contrastive loss on its own, with gradients
    pcl = PCL(embedding_dim=104)
    pshapes = {k: tuple(v.shape) for k, v in pcl.state_dict().items()}
    psd = {k: torch.from_numpy(v) for k, v in synthetic.det_state_dict(pshapes, seed=9).items()}
    pcl.load_state_dict(psd)
    ec = torch.from_numpy(rng.normal(size=(b, 104)).astype(np.float32)).requires_grad_(True)
    ew = torch.from_numpy(rng.normal(size=(b, 104)).astype(np.float32)).requires_grad_(True)
    val = pcl(ec, ew, y_bin)
    val.backward()
    out["contrastive/value"] = np.float32(val.item())
    out["contrastive/grad_cancer"] = ec.grad.numpy().copy()
    out["contrastive/grad_wt"] = ew.grad.numpy().copy()
    # self-supervised variants (utils/loss.py:33-61): + cross-entropy of the masked residue's predicted type; an EMPTY
    # prediction tensor drops the term (the validation passes of procedures/train_SSL.py)
    pred_aa = torch.from_numpy(rng.normal(size=(b, 20)).astype(np.float32)).requires_grad_(True)
    aa = torch.from_numpy(rng.randint(0, 20, size=(b,)).astype(np.int64))
    for seq_flag in (True, False):
        losses = Losses(VAE_IN, {0: 81.0, 1: 19.0}, sequence=seq_flag)
        for kind, fn, y in (("regression", losses.regression_loss_SSL, y_reg), ("bce", losses.BCE_loss_SSL, y_bin)):
            for t in (recon, mu, lv, logit, pred_aa):
                t.grad = None
            val = fn(recon, x, mu, lv, logit, y, pred_aa, aa)
            val.backward()
            tag = f"loss_ssl/{kind}/seq{int(seq_flag)}"
            out[f"{tag}/value"] = np.float32(val.item())
            out[f"{tag}/grad_logit"] = logit.grad.numpy().copy()
            out[f"{tag}/grad_pred_aa"] = pred_aa.grad.numpy().copy()
            if seq_flag:
                out[f"{tag}/grad_mu"] = mu.grad.numpy().copy()
            empty = fn(recon, x, mu, lv, logit, y, torch.zeros(0, 20), torch.zeros(0, dtype=torch.int64))
            out[f"{tag}/value_no_residue"] = np.float32(empty.item())


def golden_egnn(out):
    """Per-layer EGNN trajectory from the (unpinned) restatement, fp32 and fp64, Fe = 1 and 8."""
    for fe, seed in ((1, 1), (8, 41)):
        raw = synthetic.make_batch(2, seed=seed, deg_extra=2 if fe == 1 else 7, edge_feats=fe)
        src, dst = torch.from_numpy(raw.src), torch.from_numpy(raw.dst)
        shapes = {}
        for i, din in enumerate([20, 64, 64]):
            p = f"GCN_layers.{i}."
            shapes.update({p + "edge_mlp.0.weight": (64, 2 * din + 1 + fe), p + "edge_mlp.0.bias": (64,),
                           p + "edge_mlp.2.weight": (64, 64), p + "edge_mlp.2.bias": (64,),
                           p + "node_mlp.0.weight": (64, din + 64), p + "node_mlp.0.bias": (64,),
                           p + "node_mlp.2.weight": (64, 64), p + "node_mlp.2.bias": (64,),
                           p + "coord_mlp.0.weight": (64, 64), p + "coord_mlp.0.bias": (64,),
                           p + "coord_mlp.2.weight": (1, 64)})
        sd32 = {k: torch.from_numpy(v) for k, v in synthetic.det_state_dict(shapes, seed=13).items()}
        for dt, tagdt in ((torch.float32, "f32"), (torch.float64, "f64")):
            sd = {k: v.to(dt) for k, v in sd32.items()}
            h = torch.from_numpy(raw.x[:, :20]).to(dt)
            x = torch.from_numpy(raw.x[:, 20:]).to(dt)
            a = torch.from_numpy(raw.edge_attr).to(dt)
            for i in range(3):
                h, x = graph_ref.egnn_conv(sd, f"GCN_layers.{i}.", src, dst, raw.num_nodes, h, x, a)
                out[f"egnn/fe{fe}/{tagdt}/layer{i}/h"] = h.numpy().astype(np.float64 if dt == torch.float64 else np.float32)
                out[f"egnn/fe{fe}/{tagdt}/layer{i}/x"] = x.numpy().astype(np.float64 if dt == torch.float64 else np.float32)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)  # single-threaded reductions: stable summation order
    model_map, Losses, PCL = shims.load_reference()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    groups = {}
    for fname, fn in (("forward.npz", lambda o: golden_forward(model_map, o)),
                      ("forward_peptide.npz", lambda o: golden_forward_peptide(model_map, Losses, o)),
                      ("comparative.npz", lambda o: golden_comparative(model_map, Losses, PCL, o)),
                      ("losses.npz", lambda o: golden_losses(Losses, PCL, o)),
                      ("egnn.npz", golden_egnn)):
        out = {}
        fn(out)
        np.savez_compressed(os.path.join(GOLDEN_DIR, fname), **out)
        groups[fname] = len(out)
        print(f"{fname}: {len(out)} arrays, {os.path.getsize(os.path.join(GOLDEN_DIR, fname)) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()
