"""Oracle restatement of the reference's model forward passes and losses.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Everything here is *functional*: a model is a ``variant`` name plus a plain
``state_dict`` (name -> tensor) whose keys and shapes are exactly the
reference's (``SURVEY.md`` section 8b item 4), so the same dictionary can be
loaded into the reference classes (under ``oracle/shims.py``), fed to these
functions, and loaded into the HIP-backed ``immunostruct_amd.models`` classes.

Reference lines followed
------------------------
* encoder body: ``models/hybrid_models.py:315-340`` (identical bodies at
  ``:81-106``, ``:196-221``, ``:441-466``; ``models/comparative_models.py:
  87-115,433-461``; ``models/ablation_models.py:160-176,280-299,363-386``).
* node attention: ``models/layers.py:13-22`` (single head, no output
  projection) and ``models/layers.py:67-106`` (multi-head + ``w_concat``).
* fusion: ``models/hybrid_models.py:108-111`` (v1), ``:344-351`` (v2:
  combined attention over the fused scalars then mean over features),
  ``models/comparative_models.py:463-496,498-527`` (paired / duplicated).
* losses: ``utils/loss.py:13-61``; paired contrastive loss
  ``utils/contrastive.py:18-83``.

The forward is stochastic in the reference (``reparameterize`` samples
``randn_like`` even in eval, dropout in train): here ``eps`` and the dropout
keep-masks are explicit inputs.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .graph_ref import egnn_conv, global_max_pool, global_mean_pool

# name -> structural description (mirrors the 14 entries of models/mapping.py:7-22)
VARIANTS = {
    #                    graph  attn    vae    prop     comb  ssl    cmp    pool
    "SequenceModel":                 dict(graph=False, attn=None,   vae=True,  prop=None,  comb=None, ssl=False, cmp=False, pool=None),
    "SequenceFpModel":               dict(graph=False, attn=None,   vae=True,  prop="raw", comb=None, ssl=False, cmp=False, pool=None),
    "StructureModel":                dict(graph=True,  attn="mha8", vae=False, prop=None,  comb=None, ssl=False, cmp=False, pool="mean"),
    "StructureModel_SSL":            dict(graph=True,  attn="mha8", vae=False, prop=None,  comb=None, ssl=True,  cmp=False, pool="mean"),
    "StructureModelv2":              dict(graph=True,  attn="mha8", vae=False, prop=None,  comb=None, ssl=True,  cmp=False, pool="meanmax"),
    "HybridModel":                   dict(graph=True,  attn="v1",   vae=True,  prop="emb", comb=None, ssl=False, cmp=False, pool="mean"),
    "HybridModel_SSL":               dict(graph=True,  attn="v1",   vae=True,  prop="emb", comb=None, ssl=True,  cmp=False, pool="mean"),
    "HybridModelv2":                 dict(graph=True,  attn="mha",  vae=True,  prop="emb", comb=16,   ssl=False, cmp=False, pool="mean"),
    "HybridModelv2_SSL":             dict(graph=True,  attn="mha",  vae=True,  prop="emb", comb=32,   ssl=True,  cmp=False, pool="mean"),
    "HybridModel_Comparative":       dict(graph=True,  attn="v1",   vae=True,  prop="emb", comb=None, ssl=False, cmp=True,  pool="mean"),
    "HybridModel_Comparative_SSL":   dict(graph=True,  attn="v1",   vae=True,  prop="emb", comb=None, ssl=True,  cmp=True,  pool="mean"),
    "HybridModelv2_Comparative":     dict(graph=True,  attn="mha",  vae=True,  prop="emb", comb=32,   ssl=False, cmp=True,  pool="mean"),
    "HybridModelv2_Comparative_SSL": dict(graph=True,  attn="mha",  vae=True,  prop="emb", comb=32,   ssl=True,  cmp=True,  pool="mean"),
    "DualModel":                     dict(graph=True,  attn="v1",   vae=True,  prop=None,  comb=None, ssl=False, cmp=False, pool="mean"),
}


def _lin(sd, name, t, bias=True):
    return F.linear(t, sd[name + ".weight"], sd[name + ".bias"] if bias else None)


def _drop(t, keep_mask, p=0.1):
    if keep_mask is None:
        return t
    return t * keep_mask / (1.0 - p)


def single_head_attention(sd, prefix, x):
    """``SelfAttention`` (``models/layers.py:6-22``): no output projection."""
    q, k, v = _lin(sd, prefix + "query", x), _lin(sd, prefix + "key", x), _lin(sd, prefix + "value", x)
    w = torch.softmax(q @ k.transpose(-2, -1) / (k.shape[-1] ** 0.5), dim=-1)
    return w @ v, w


def multi_head_attention(sd, prefix, x, heads):
    """``MultiHeadAttention`` (``models/layers.py:51-106``), mask never passed."""
    q, k, v = _lin(sd, prefix + "w_q", x), _lin(sd, prefix + "w_k", x), _lin(sd, prefix + "w_v", x)
    b, n, d = q.shape
    dh = d // heads

    def split(t):
        return t.view(b, n, heads, dh).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    w = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(dh), dim=-1)
    o = (w @ v).transpose(1, 2).contiguous().view(b, n, d)
    return _lin(sd, prefix + "w_concat", o), w


def encode_graph(spec, sd, g, heads=1):
    """EGNN stack -> node attention -> pooling.  Returns (pooled, attn weights, node h, coords)."""
    feats = g.ndata["x"]
    h, x, a = feats[:, :20], feats[:, 20:], g.edata["edge_attr"]
    src, dst = g.edges()
    n_layers = 0
    while f"GCN_layers.{n_layers}.edge_mlp.0.weight" in sd:
        n_layers += 1
    for i in range(n_layers):
        h, x = egnn_conv(sd, f"GCN_layers.{i}.", src, dst, g.num_nodes(), h, x, a)
    counts = g.batch_num_nodes()
    bsz = int(counts.numel())
    hidden = h.shape[1]
    hb = h.view(bsz, -1, hidden)  # requires equal node counts (padding), hybrid_models.py:326
    if spec["attn"] == "v1":
        out, w = single_head_attention(sd, "self_attention.", hb)
    else:
        nh = 8 if spec["attn"] == "mha8" else heads
        out, w = multi_head_attention(sd, "self_attention.", hb, nh)
    out = out.reshape(-1, hidden)
    batch_index = torch.repeat_interleave(torch.arange(bsz, device=h.device), counts.to(h.device))
    pooled = global_mean_pool(out, batch_index, bsz)
    if spec["pool"] == "meanmax":
        pooled = torch.cat([pooled, global_max_pool(out, batch_index, bsz)], dim=-1)
    return pooled, w, h, x


def encode_item(variant, sd, g, seq, prop, eps=None, drop=None, heads=1):
    """Shared encoder of every variant; returns a dict of named intermediates."""
    spec = VARIANTS[variant]
    drop = drop or {}
    out = {}
    if spec["graph"]:
        out["x_gat_node"], out["attention_weights"], out["node_h"], out["node_x"] = encode_graph(spec, sd, g, heads)
    if spec["prop"] == "emb":
        p = F.relu(_lin(sd, "property_embedding.0", prop))
        p = _drop(p, drop.get("prop"))
        p = F.relu(_lin(sd, "property_embedding.3", p))
    elif spec["prop"] == "raw":
        p = prop
    else:
        p = None
    if spec["vae"]:
        flat = seq.reshape(-1, sd["vae_fc1.weight"].shape[1])
        h1 = F.relu(_lin(sd, "vae_fc1", flat))
        mu, logvar = _lin(sd, "vae_fc21", h1), _lin(sd, "vae_fc22", h1)
        if eps is None:
            eps = torch.randn_like(mu)
        z = mu + eps * torch.exp(0.5 * logvar)
        if p is not None:
            z = torch.cat([z, p], dim=1)
        out.update(mu=mu, logvar=logvar, z_vae=z,
                   recon_x=_lin(sd, "vae_fc4", F.relu(_lin(sd, "vae_fc3", z))))
    return out


def _head(variant, sd, combined, drop):
    """combined attention (v2) -> classifier (+ SSL heads)."""
    spec = VARIANTS[variant]
    drop = drop or {}
    if spec["comb"]:
        c, _ = multi_head_attention(sd, "combined_attention.", combined.unsqueeze(2), 8)
        combined = c.mean(dim=2)
    hid = F.relu(_lin(sd, "classifier.1", combined.flatten(1)))
    hid = _drop(hid, drop.get("cls"))
    if spec["ssl"]:
        return _lin(sd, "classifier_head", hid), _lin(sd, "node_predictor_head", hid)
    return _lin(sd, "classifier.4", hid), None


def forward(variant, sd, g, seq, prop, eps=None, drop=None, heads=1, use_wt_for_downstream=True):
    """``Model.forward``; returns a dict (see :func:`as_reference_tuple`)."""
    spec = VARIANTS[variant]
    it = encode_item(variant, sd, g, seq, prop, eps, drop, heads)
    if spec["graph"] and spec["vae"]:
        parts = [it["x_gat_node"], it["z_vae"]]
        if spec["cmp"] and use_wt_for_downstream:
            parts = parts + parts  # "hot fix" duplication, comparative_models.py:511
        combined = torch.cat(parts, dim=1)
    elif spec["graph"]:
        combined = it["x_gat_node"]
    else:
        combined = it["z_vae"]
    it["combined_in"] = combined
    it["final_output"], it["node_prediction"] = _head(variant, sd, combined, drop)
    return it


def forward_comparative(variant, sd, g_pair, seq_pair, prop_pair, eps_pair=(None, None),
                        drop_pair=(None, None), heads=1, use_wt_for_downstream=True):
    """``Model.forward_comparative`` (``models/comparative_models.py:463-496``).

    ``drop_pair[0]['cls']`` is the classifier keep-mask (there is one classifier call).
    """
    c = encode_item(variant, sd, g_pair[0], seq_pair[0], prop_pair[0], eps_pair[0], drop_pair[0], heads)
    w = encode_item(variant, sd, g_pair[1], seq_pair[1], prop_pair[1], eps_pair[1], drop_pair[1], heads)
    emb_c = torch.cat([c["x_gat_node"], c["z_vae"]], dim=1)
    emb_w = torch.cat([w["x_gat_node"], w["z_vae"]], dim=1)
    combined = torch.cat([emb_c, emb_w], dim=1) if use_wt_for_downstream else emb_c
    final, node_pred = _head(variant, sd, combined, drop_pair[0])
    return dict(cancer=c, wt=w, embeddings=[emb_c, emb_w], final_output=final, node_prediction=node_pred)


def as_reference_tuple(variant, it, return_embedding=False, return_attention=False):
    """Order the outputs the way the reference ``forward`` returns them."""
    spec = VARIANTS[variant]
    if not spec["vae"]:
        first, mu, logvar = 0, 0, 0
    else:
        mu, logvar = it["mu"], it["logvar"]
        first = it["recon_x"]
        if spec["graph"] and return_embedding:
            first = it["x_gat_node"]
        elif spec["graph"] and return_attention:
            first = it["attention_weights"]
    if spec["ssl"]:
        return first, mu, logvar, it["final_output"], it["node_prediction"]
    return first, mu, logvar, it["final_output"]


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------

def kld_mean(mu, logvar):
    return -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())


def regression_loss(recon_x, x, mu, logvar, final_output, y, vae_input_dim, sequence=True):
    """``Losses.regression_loss`` (``utils/loss.py:13-21``)."""
    reg = F.mse_loss(final_output.squeeze(), y.squeeze(), reduction="mean")
    if not sequence:
        return reg
    mse = F.mse_loss(recon_x, x.reshape(-1, vae_input_dim), reduction="mean")
    return 2.0 * reg + 0.5 * mse + 0.5 * kld_mean(mu, logvar)


def bce_loss(recon_x, x, mu, logvar, final_output, y, vae_input_dim, pos_weight, sequence=True):
    """``Losses.BCE_loss`` (``utils/loss.py:23-31``); ``pos_weight = n0/n1`` (``:11``)."""
    pw = torch.as_tensor(pos_weight, dtype=final_output.dtype)
    bce = F.binary_cross_entropy_with_logits(final_output.view(-1), y.view(-1), pos_weight=pw, reduction="mean")
    if not sequence:
        return bce
    mse = F.mse_loss(recon_x, x.reshape(-1, vae_input_dim), reduction="mean")
    return 5.0 * bce + 0.1 * mse + 0.1 * kld_mean(mu, logvar)


def amino_cross_entropy(pred_amino_acid, amino_acid):
    """SSL term (``utils/loss.py:33-37``)."""
    if pred_amino_acid.numel():
        return F.cross_entropy(pred_amino_acid, amino_acid)
    return 0


def paired_contrastive_loss(sd, emb_c, emb_w, is_immunogenic, z_dim=128, lambda_off_diag=1e-2, bn_eps=1e-5):
    """``PairedContrastiveLoss.forward`` (``utils/contrastive.py:37-83``).

    ``sd`` holds the projector: ``projector.0.weight`` (z,104), ``projector.1.
    {weight,bias}`` (BatchNorm1d, always batch statistics: the module is never
    put in eval, ``procedures/train.py:76``), ``projector.3.weight`` (z,z).
    Returns python ``0`` unless the target has exactly two distinct values.
    """
    if is_immunogenic.unique().numel() != 2:
        return 0
    pos = is_immunogenic > is_immunogenic.mean()

    def project(e):
        t = F.linear(e, sd["projector.0.weight"])
        mean = t.mean(0)
        var = t.var(0, unbiased=False)
        t = (t - mean) / torch.sqrt(var + bn_eps) * sd["projector.1.weight"] + sd["projector.1.bias"]
        return F.linear(F.relu(t), sd["projector.3.weight"])

    zc, zw = project(emb_c), project(emb_w)
    bsz = zc.shape[0]
    zc = zc - zc.mean(0)
    zw = zw - zw.mean(0)
    std_c = torch.sqrt(zc.var(dim=0) + 1e-4)
    std_w = torch.sqrt(zw.var(dim=0) + 1e-4)
    std_loss = F.relu(1 - std_c).mean() / 2 + F.relu(1 - std_w).mean() / 2
    pair = zc @ zw.T / z_dim
    corr = zc.T @ zw / bsz
    eye_b = torch.eye(bsz, dtype=zc.dtype)
    ideal = eye_b * pos.to(zc.dtype).unsqueeze(1)
    wgt_b = torch.where(eye_b.bool(), torch.ones_like(eye_b), torch.full_like(eye_b, lambda_off_diag))
    eye_z = torch.eye(z_dim, dtype=zc.dtype)
    wgt_z = torch.where(eye_z.bool(), torch.ones_like(eye_z), torch.full_like(eye_z, lambda_off_diag))
    return ((pair - ideal).pow(2) * wgt_b).sum() + ((corr - eye_z).pow(2) * wgt_z).sum() + std_loss
