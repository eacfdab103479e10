"""GPU parity of the drop-in model classes: HIP path vs golden vectors from the reference and vs the oracle."""
import os
import unittest.mock as mock

import numpy as np
import pytest
import torch

from immunostruct_amd import synthetic
from immunostruct_amd.models import model_map
from immunostruct_amd.utils import Losses, PairedContrastiveLoss
from oracle import functional_ref as FR
from tests import helpers as H

pytestmark = pytest.mark.gpu
BATCH = 16
OUT_TOL, GRAD_TOL = 1e-5, 1e-4      # element-wise (tests/helpers.py): SURVEY 8c / BASELINE.md: 1e-5 rel on outputs, 1e-4 on gradients
# full-size gradients / outputs: allowed multiple of the fp32 oracle's own distance from fp64 (see the [190-128] test).  Round 5: 1.0 --
# wherever the element-wise bound against fp64 is missed, the HIP path must be at least as close to fp64 as the reference arithmetic in
# fp32 (the worse of two summation orders) is; rounds 3 / 4 needed 8 / 2.  Measured with the allowance off (factor 0): ONE tensor of
# all oracle tests misses the bound, GCN_layers.3.coord_mlp.2.weight at B = 128: 1.2 x where the fp32 oracle has 3.0 x
FULL_SIZE_FACTOR = float(os.environ.get("IMMUNOSTRUCT_TEST_FULL_SIZE_FACTOR", "1.0"))
TRAJECTORY_FACTOR = max(2.0, FULL_SIZE_FACTOR)      # weights after 10 AdamW steps: the fp32 oracle's own drift depends on the host's BLAS (HISTORY.md 7.13)


def _with_eps(fn, eps_list, device):
    it = iter(eps_list)
    with mock.patch("torch.randn_like", lambda t: next(it).to(device=device, dtype=t.dtype)):
        return fn()


def _golden_cases():
    gold = H.golden("forward.npz")
    names = sorted({k.split("/")[1] for k in gold.files if k.startswith("fwd/") and k.count("/") >= 2})
    out = []
    for n in names:
        if "Comparative" in n:
            out += [(n, dict(use_wt_for_downstream=True)), (n, dict(use_wt_for_downstream=False))]
        else:
            out.append((n, {}))
    return out


@pytest.mark.parametrize("name,kw", _golden_cases())
def test_forward_matches_reference_golden(cuda_device, name, kw):
    gold = H.golden("forward.npz")
    dev = cuda_device
    raw = synthetic.make_batch(BATCH, seed=1)
    g = H.product_graph(raw, dev)
    seq, prop = torch.from_numpy(raw.one_hot_sequence()).to(dev), torch.from_numpy(raw.prop).to(dev)
    model = model_map[name](vae_input_dim=H.VAE_IN, device=dev, **kw).to(dev)
    model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=3))
    model.eval()
    eps = H.make_eps(11, BATCH)
    with torch.no_grad():
        res = _with_eps(lambda: model(g, seq, prop), [eps], dev)
        emb = _with_eps(lambda: model(g, seq, prop, return_embedding=True), [eps], dev)
        att = _with_eps(lambda: model(g, seq, prop, return_attention=True), [eps], dev)
    tag = "fwd/" + name + ("" if not kw else f"/wt{int(kw['use_wt_for_downstream'])}")
    errs = {"final": H.assert_close(res[3].cpu(), gold[f"{tag}/final_output"], OUT_TOL, "final_output")}
    if f"{tag}/mu" in gold.files:
        H.assert_close(res[1].cpu(), gold[f"{tag}/mu"], OUT_TOL, "mu")
        H.assert_close(res[2].cpu(), gold[f"{tag}/logvar"], OUT_TOL, "logvar")
        H.assert_close(res[0].cpu()[:, H.RECON_COLS], gold[f"{tag}/recon_x_cols"], OUT_TOL, "recon_x")
    else:
        assert res[0] == 0 and res[1] == 0 and res[2] == 0
    if f"{tag}/x_gat_node" in gold.files:
        errs["emb"] = H.assert_close(emb[0].cpu(), gold[f"{tag}/x_gat_node"], OUT_TOL, "x_gat_node")
    if f"{tag}/attention_rows" in gold.files:
        H.assert_close(att[0].cpu()[..., H.ATTN_ROWS, :], gold[f"{tag}/attention_rows"], OUT_TOL, "attention weights")
    if f"{tag}/node_prediction" in gold.files:
        H.assert_close(res[4].cpu(), gold[f"{tag}/node_prediction"], OUT_TOL, "node_prediction")
    print(tag, {k: f"{v:.1e}" for k, v in errs.items()})


@pytest.mark.parametrize("name", ["HybridModelv2", "SequenceFpModel", "SequenceModel", "HybridModelv2_Comparative"])
def test_peptide_width_train_step_matches_reference_golden(cuda_device, name):
    """The reference's DEFAULT input width (no --full-sequence): vae_input_dim = 11 * 21 = 231, sequence = the padded peptide's
    one-hot (train_IEDB_wFT.py:59-60, data/util_dataloader.py:52-66).  Forward outputs, the regression loss and parameter
    gradients of the two wide VAE layers against vectors produced by the reference's own classes."""
    gold = H.golden("forward_peptide.npz")
    dev = cuda_device
    raw = synthetic.make_batch(BATCH, seed=1)
    g = H.product_graph(raw, dev)
    seq, prop = H.peptide_one_hot(raw).to(dev), torch.from_numpy(raw.prop).to(dev)
    assert seq.shape == (BATCH, 11, 21)
    model = model_map[name](vae_input_dim=H.PEP_IN, device=dev).to(dev)
    model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=3))
    model.eval()
    res = _with_eps(lambda: model(g, seq, prop), [H.make_eps(11, BATCH)], dev)
    tag = f"pep/{name}"
    for i, k in enumerate(("recon_x", "mu", "logvar", "final_output")):
        H.assert_close(res[i].detach().cpu(), gold[f"{tag}/{k}"], OUT_TOL, k)
    loss = Losses(H.PEP_IN, {0: 81.0, 1: 19.0}, sequence=True).regression_loss(res[0], seq, res[1], res[2], res[3],
                                                                              torch.from_numpy(raw.y_reg).to(dev))
    assert abs(float(loss.detach()) - float(gold[f"{tag}/loss"])) <= 1e-5 * abs(float(gold[f"{tag}/loss"]))
    loss.backward()
    named = dict(model.named_parameters())
    for key in ("vae_fc1.weight", "vae_fc4.bias", "vae_fc21.weight"):
        H.assert_close(named[key].grad.cpu(), gold[f"{tag}/grad/{key}"], GRAD_TOL, f"grad {key}")


@pytest.mark.parametrize("name", ["HybridModelv2_Comparative", "HybridModel_Comparative"])
@pytest.mark.parametrize("wt", [True, False])
def test_comparative_train_step_matches_reference_golden(cuda_device, name, wt):
    """forward_comparative + BCE losses + paired contrastive loss + backward, as procedures/train.py:84-123."""
    gold = H.golden("comparative.npz")
    dev = cuda_device
    raw_c, raw_w = synthetic.make_batch(BATCH, seed=21), synthetic.make_batch(BATCH, seed=22)
    gc, gw = H.product_graph(raw_c, dev), H.product_graph(raw_w, dev)
    sc, sw = torch.from_numpy(raw_c.one_hot_sequence()).to(dev), torch.from_numpy(raw_w.one_hot_sequence()).to(dev)
    pc, pw = torch.from_numpy(raw_c.prop).to(dev), torch.from_numpy(raw_w.prop).to(dev)
    y = torch.from_numpy(raw_c.y_bin).to(dev)
    model = model_map[name](vae_input_dim=H.VAE_IN, device=dev, use_wt_for_downstream=wt).to(dev)
    model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=5))
    model.eval()  # dropout off (golden was captured in eval mode); gradients still flow
    pcl = PairedContrastiveLoss(embedding_dim=104, device=dev)
    pcl.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in pcl.state_dict().items()}, seed=9))
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    emb, rec, mu, lv, final = _with_eps(lambda: model.forward_comparative((gc, gw), (sc, sw), (pc, pw)),
                                        [H.make_eps(31, BATCH), H.make_eps(32, BATCH)], dev)
    l_c = losses.BCE_loss(rec[0], sc, mu[0], lv[0], final, y)
    l_w = losses.BCE_loss(rec[1], sw, mu[1], lv[1], final, y)
    l_con = pcl(emb[0], emb[1], y)
    loss = (l_c + l_w) / 2 + 0.01 * l_con
    loss.backward()
    tag = f"cmp/{name}/wt{int(wt)}"
    H.assert_close(final.detach().cpu(), gold[f"{tag}/final_output"], OUT_TOL, "final_output")
    H.assert_close(emb[0].detach().cpu(), gold[f"{tag}/emb_cancer"], OUT_TOL, "emb_cancer")
    H.assert_close(emb[1].detach().cpu(), gold[f"{tag}/emb_wt"], OUT_TOL, "emb_wt")
    rel = lambda a, b: abs(float(a) - float(b)) / abs(float(b))
    assert rel(l_c, gold[f"{tag}/bce_cancer"]) <= 1e-5
    assert rel(l_con, gold[f"{tag}/contrastive"]) <= 1e-4, "contrastive loss must match the reference to 1e-4 rel"
    assert rel(loss, gold[f"{tag}/loss"]) <= 1e-5
    named = dict(model.named_parameters())
    errs = {}
    for key in [k for k in gold.files if k.startswith(f"{tag}/grad/")]:
        pname = key.split("/grad/")[1]
        errs[pname] = H.assert_close(named[pname].grad.cpu(), gold[key], GRAD_TOL, f"grad {pname}")
    print(tag, {k: f"{v:.1e}" for k, v in errs.items()})


@pytest.mark.parametrize("seed_kind", ["unit", "scaled", "plain"])
def test_speculative_reconstruction_backward_is_the_plain_backward(cuda_device, seed_kind):
    """functional.SpeculativeBackward (the engine's steps): the reconstruction term's backward launched from the loss gives,
    bit for bit, the gradients of the ordinary backward when the backward is seeded with ``unit_gradient()``; with any other
    seed (a scaled loss, a plain ``backward()``) the speculated results are dropped and the ordinary path runs."""
    from immunostruct_amd import functional as HF
    dev = cuda_device
    raw = synthetic.make_batch(6, seed=71, deg_extra=2)
    sd = H.det_sd(H.model_shapes("HybridModelv2"), seed=3)
    eps = H.make_eps(4, 6)
    seq, prop, y = torch.from_numpy(raw.one_hot_sequence()).to(dev), torch.from_numpy(raw.prop).to(dev), torch.from_numpy(raw.y_reg).to(dev)
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)

    def run(speculate):
        model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
        model.load_state_dict(sd)
        model.eval()
        g = H.product_graph(raw, dev)
        ctx = HF.SpeculativeBackward() if speculate else mock.patch.object(HF.SpeculativeBackward, "enabled", False)
        with ctx:
            res = _with_eps(lambda: model(g, seq, prop), [eps], dev)
            loss = losses.regression_loss(res[0], seq, res[1], res[2], res[3], y)
        node = res[0].grad_fn
        assert (getattr(node, "spec", None) is not None) == speculate      # launched ahead only when asked to
        if seed_kind == "unit":
            loss.backward(HF.unit_gradient(dev))
        elif seed_kind == "scaled":
            (0.5 * loss).backward()
        else:
            loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}

    l0, g0 = run(False)
    l1, g1 = run(True)
    assert l0 == l1
    assert g0.keys() == g1.keys()
    for k in g0:
        assert torch.equal(g0[k], g1[k]), f"{k}: speculative and plain backward differ ({seed_kind} seed)"


@pytest.mark.parametrize("n_pad,b", [(190, 8), (40, 5), (100, 3), (150, 4), (250, 2), (300, 2), (190, 128)])
def test_full_train_step_gradients_vs_oracle(cuda_device, n_pad, b):
    """HybridModelv2: loss and every parameter gradient vs oracle autograd -- at BASELINE config 2's full size (B = 128 x 190
    nodes: what bench.py times), at the reference's padded node count (190) and
    at other dataset-wide node counts (the node-attention kernels have one instantiation per 64 nodes, the edge / node
    kernels tile by 16 / 32 rows: 40, 100, 150 and 250 nodes hit every variant and ragged last tiles).  300 nodes is past the
    node-attention kernels' 256-row limit: the attention is then composed from device-side torch ops (models/layers.py
    ``attend_pooled_mean``), the EGNN stack and everything else stay on the HIP kernels -- same bound."""
    dev = cuda_device
    reals = (n_pad - 2, n_pad - 1, n_pad)
    raw = synthetic.make_batch(b, seed=33, deg_extra=5, n_pad=n_pad, n_real_choices=reals)
    sd = H.det_sd(H.model_shapes("HybridModelv2"), seed=14)
    eps = H.make_eps(5, b)
    y = torch.from_numpy(raw.y_reg)
    # oracle
    sd_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    go = H.oracle_graph(raw)
    seq, prop = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop)
    it = FR.forward("HybridModelv2", sd_o, go, seq, prop, eps=eps)
    lo = FR.regression_loss(it["recon_x"], seq, it["mu"], it["logvar"], it["final_output"], y, H.VAE_IN)
    lo.backward()
    sd_64 = sd_p = None
    if b >= 64:
        # full size: sums over ~150 k edges in fp32 differ between two correct implementations by more than the small-batch
        # tolerance (dw_r sums products with squared distances up to 1e4); the yardstick is then the fp64 oracle -- the HIP
        # gradient must meet the element-wise bound against it, or be within FULL_SIZE_FACTOR x of the fp32 reference
        # arithmetic's own distance from it.  That distance is itself one draw of a round-off realisation, so it is taken as the
        # larger of TWO realisations of the fp32 oracle: the batch as given, and the same batch with its edge list permuted
        # (every scatter-add then runs in another order; the two differ by up to 1.6 x on the worst tensors).
        # Rounds 3 - 4 needed 8 x / 6 x here: eight tensors of the coordinate path sat at 2 - 4 x the fp32 oracle's distance (worst
        # 8.9 x the element-wise bound where the oracle has 3.0 x).  The cause was not in the EGNN stack (stack alone, random
        # upstream gradient: as accurate as the oracle, tests/tools/grad_error_probe.py) but in the FUSION HEAD: its classifier's
        # 104-term fp32 dot-product chain left the logit 3.8 x, and the softmax moments its backward multiplies with left the
        # gradient at the head's inputs 2.7 x, further from fp64 than torch's blocked fp32 sums -- and that backward amplifies what
        # it is given ~ 100 x, into every gradient below it (tests/tools/grad_error_probe_model.py).  With those sums in fp64
        # (blocks of 8 in fp32, block sums in fp64: csrc/combined_attention.hip, csrc/mlp_head.hip) ONE tensor exceeds the
        # element-wise bound, GCN_layers.3.coord_mlp.2.weight at 1.2 x where the fp32 oracle has 3.0 x, and the HIP gradients are
        # closer to fp64 than the oracle's on the tensors that used to fail.  Factor 2 since then (HISTORY.md 7.10), 1 since round 5.
        sd_64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
        it64 = FR.forward("HybridModelv2", sd_64, H.oracle_graph(raw, torch.float64), seq.double(), prop.double(), eps=eps.double())
        FR.regression_loss(it64["recon_x"], seq.double(), it64["mu"], it64["logvar"], it64["final_output"], y.double(), H.VAE_IN).backward()
        sd_p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        itp = FR.forward("HybridModelv2", sd_p, H.oracle_graph_permuted(raw, 7), seq, prop, eps=eps)
        FR.regression_loss(itp["recon_x"], seq, itp["mu"], itp["logvar"], itp["final_output"], y, H.VAE_IN).backward()
    # HIP
    model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
    model.load_state_dict(sd)
    model.eval()
    g = H.product_graph(raw, dev)
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    res = _with_eps(lambda: model(g, seq.to(dev), prop.to(dev)), [eps], dev)
    lh = losses.regression_loss(res[0], seq.to(dev), res[1], res[2], res[3], y.to(dev))
    lh.backward()
    assert abs(float(lh) - float(lo)) <= 1e-5 * abs(float(lo))
    # the outputs the north star names ("logits / embeddings within a stated fp32 tolerance"), element-wise at OUT_TOL, at EVERY size
    # of this test incl. the benchmark's: the small batches against the fp32 oracle; B = 128 with the fp64 oracle as the yardstick
    # like the gradients below (within the bound of it, or within FULL_SIZE_FACTOR x the fp32 oracle's own distance from it)
    with torch.no_grad():
        emb = _with_eps(lambda: model(g, seq.to(dev), prop.to(dev), return_embedding=True), [eps], dev)[0]
    outs = {"final_output": res[3], "mu": res[1], "logvar": res[2], "recon_x": res[0], "x_gat_node": emb}
    for key, hip in outs.items():
        cut = (lambda t: t[:, H.RECON_COLS]) if key == "recon_x" else (lambda t: t)
        hip_v = cut(hip.detach().cpu())
        if sd_64 is None:
            H.assert_close(hip_v, cut(it[key].detach()), OUT_TOL, key)
        else:
            ref64 = cut(it64[key].detach())
            r_hip = H.worst_ratio(hip_v, ref64, OUT_TOL)
            r_ref = max(H.worst_ratio(cut(it[key].detach()), ref64, OUT_TOL), H.worst_ratio(cut(itp[key].detach()), ref64, OUT_TOL))
            print("  output %-14s HIP %.3f x the element-wise bound against fp64, fp32 oracle %.3f x" % (key, r_hip, r_ref))
            assert r_hip <= max(1.0, FULL_SIZE_FACTOR * r_ref), (f"{key}: HIP is {r_hip:.2f} x the element-wise bound away from the fp64 "
                                                              f"oracle, the fp32 oracle {r_ref:.2f} x")
    worst = ("", 0.0)
    worst_ratio = ("", 0.0, 0.0)
    gmax = max(float(v.grad.abs().max()) for v in sd_o.values() if v.grad is not None)
    for name, p in model.named_parameters():
        ref_grad = sd_o[name].grad
        if ref_grad is None:   # parameter unused by the loss (last layer's coord MLP: its x output is dropped)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{name} should have zero gradient"
            continue
        if float(ref_grad.abs().max()) < 1e-6 * gmax:
            # analytically zero gradient (softmax is invariant to the key bias): both sides are pure round-off
            assert float(p.grad.abs().max()) < 1e-5 * gmax, f"{name} should be ~0"
            continue
        if sd_64 is not None:
            r_hip = H.worst_ratio(p.grad.cpu(), sd_64[name].grad, GRAD_TOL)
            r_ref = max(H.worst_ratio(ref_grad, sd_64[name].grad, GRAD_TOL), H.worst_ratio(sd_p[name].grad, sd_64[name].grad, GRAD_TOL))
            if r_hip > worst_ratio[1]:
                worst_ratio = (name, r_hip, r_ref)
            if r_hip > 1.0:
                print("  above the element-wise bound: %-40s HIP %.3f x, fp32 oracle %.3f x (other edge order %.3f x)" % (
                    name, r_hip, H.worst_ratio(ref_grad, sd_64[name].grad, GRAD_TOL), H.worst_ratio(sd_p[name].grad, sd_64[name].grad, GRAD_TOL)))
            assert r_hip <= max(1.0, FULL_SIZE_FACTOR * r_ref), (f"grad {name}: HIP is {r_hip:.2f} x the element-wise bound away from the fp64 "
                                                    f"gradient, the fp32 oracle (worse of two summation orders) {r_ref:.2f} x")
            err = H.rel_err(p.grad.cpu(), sd_64[name].grad)
        else:
            err = H.assert_close(p.grad.cpu(), ref_grad, GRAD_TOL, f"grad {name}")
        if err > worst[1]:
            worst = (name, err)
    print("worst parameter-gradient error", worst)
    if sd_64 is not None:
        print("worst ratio to the element-wise bound against the fp64 gradient: %s HIP %.3f x, fp32 oracle %.3f x" % worst_ratio)


@pytest.mark.parametrize("name,heads", [("HybridModelv2", 4), ("HybridModel", 1), ("StructureModel", 8), ("HybridModelv2", 2)])
def test_other_attention_head_counts_vs_oracle(cuda_device, name, heads):
    """``self_attention_heads`` is a constructor argument of the reference's models (hybrid_models.py:241-251).  The fused node
    attention covers 1 and 8 heads; other counts run the scores / softmax / column mean as device-side torch ops
    (models/layers.py) around the same HIP value / output projection.  Loss and every parameter gradient vs the oracle (fp64
    yardstick: within the element-wise bound of the fp64 gradient, or within FULL_SIZE_FACTOR x the fp32 oracle's own distance from it)."""
    dev = cuda_device
    b = 8          # (tiny batches make the bias gradients of vae_fc22 -- b-term sums of cancelling values that also carry the fusion
    raw = synthetic.make_batch(b, seed=33, deg_extra=3)      # head's closed-form gradient -- sit at 0.5 - 1.5 x the bound: b = 5 is a coin toss)
    kw = {} if name == "StructureModel" else dict(self_attention_heads=heads)
    model = model_map[name](vae_input_dim=H.VAE_IN, device=dev, **kw).to(dev)
    sd = H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=23)
    model.load_state_dict(sd)
    model.eval()
    eps, y = H.make_eps(6, b), torch.from_numpy(raw.y_reg)
    seq, prop = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop)
    seq_loss = name != "StructureModel"

    def oracle(dtype):
        sd_o = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
        it = FR.forward(name, sd_o, H.oracle_graph(raw, dtype), seq.to(dtype), prop.to(dtype), eps=eps.to(dtype), heads=heads)
        if seq_loss:
            lo = FR.regression_loss(it["recon_x"], seq.to(dtype), it["mu"], it["logvar"], it["final_output"], y.to(dtype), H.VAE_IN)
        else:
            lo = FR.regression_loss(None, seq, None, None, it["final_output"], y.to(dtype), H.VAE_IN, sequence=False)
        lo.backward()
        return float(lo.detach()), sd_o
    lo, sd_o = oracle(torch.float32)
    lo64, sd_64 = oracle(torch.float64)
    res = _with_eps(lambda: model(H.product_graph(raw, dev), seq.to(dev), prop.to(dev)), [eps], dev)
    lh = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=seq_loss).regression_loss(res[0], seq.to(dev), res[1], res[2], res[3], y.to(dev))
    lh.backward()
    assert abs(float(lh.detach()) - lo64) <= 1e-5 * abs(lo64)
    gmax = max(float(v.grad.abs().max()) for v in sd_o.values() if v.grad is not None)
    for pname, p in model.named_parameters():
        ref_grad = sd_o[pname].grad
        if ref_grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{pname} should have zero gradient"
        elif float(ref_grad.abs().max()) < 1e-6 * gmax:
            assert float(p.grad.abs().max()) < 1e-5 * gmax, f"{pname} should be ~0"
        else:
            r_hip = H.worst_ratio(p.grad.cpu(), sd_64[pname].grad, GRAD_TOL)
            r_ref = H.worst_ratio(ref_grad, sd_64[pname].grad, GRAD_TOL)
            assert r_hip <= max(1.0, FULL_SIZE_FACTOR * r_ref), (f"grad {pname}: HIP is {r_hip:.2f} x the element-wise bound away from the fp64 "
                                                    f"gradient, the fp32 oracle {r_ref:.2f} x")


@pytest.mark.parametrize("name", ["StructureModel", "HybridModelv2"])
def test_reference_default_command_line_at_full_size_vs_oracle(cuda_device, name):
    """The reference entry point's DEFAULTS (train_IEDB_wFT.py:17,21): ``--model StructureModel`` -- graph only, 8-head node attention,
    ablation_models.py:127-180 -- at ``--batch-size 150``, and the multimodal model at the same batch size (the benchmark's B = 128 is
    BASELINE.json's, not the script's).  B = 150 x 190 nodes: 1782 backward node tiles (a fourth, ragged round of the persistent grid),
    150 workgroups of the attention and head kernels.  Loss, outputs and every parameter gradient with the fp64 oracle as the
    yardstick, like ``test_full_train_step_gradients_vs_oracle[190-128]``."""
    dev = cuda_device
    b = 150
    raw = synthetic.make_batch(b, seed=35, deg_extra=2)
    model = model_map[name](vae_input_dim=H.VAE_IN, device=dev).to(dev)
    sd = H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=15)
    model.load_state_dict(sd)
    model.eval()
    eps, y = H.make_eps(7, b), torch.from_numpy(raw.y_reg)
    seq, prop = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop)
    seq_loss = name != "StructureModel"
    heads = 8 if name == "StructureModel" else 1

    def oracle(dtype, graph):
        sd_o = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
        it = FR.forward(name, sd_o, graph, seq.to(dtype), prop.to(dtype), eps=eps.to(dtype), heads=heads)
        if seq_loss:
            lo = FR.regression_loss(it["recon_x"], seq.to(dtype), it["mu"], it["logvar"], it["final_output"], y.to(dtype), H.VAE_IN)
        else:
            lo = FR.regression_loss(None, seq, None, None, it["final_output"], y.to(dtype), H.VAE_IN, sequence=False)
        lo.backward()
        return float(lo.detach()), sd_o, it
    lo, sd_o, it = oracle(torch.float32, H.oracle_graph(raw))
    _, sd_p, itp = oracle(torch.float32, H.oracle_graph_permuted(raw, 7))
    lo64, sd_64, it64 = oracle(torch.float64, H.oracle_graph(raw, torch.float64))
    g = H.product_graph(raw, dev)
    res = _with_eps(lambda: model(g, seq.to(dev), prop.to(dev)), [eps], dev)
    lh = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=seq_loss).regression_loss(res[0], seq.to(dev), res[1], res[2], res[3], y.to(dev))
    lh.backward()
    assert abs(float(lh.detach()) - lo64) <= 1e-5 * abs(lo64)
    with torch.no_grad():      # (the reference's StructureModel ignores return_embedding, ablation_models.py:160-180: the pooled graph embedding
        emb = _with_eps(lambda: model._encode(g, seq.to(dev), prop.to(dev), need_attention=False), [eps], dev)["x_gat_node"]      # is read where it is formed)
    outs = {"final_output": res[3], "x_gat_node": emb}
    if seq_loss:
        outs.update({"mu": res[1], "logvar": res[2], "recon_x": res[0]})
    for key, hip in outs.items():
        cut = (lambda t: t[:, H.RECON_COLS]) if key == "recon_x" else (lambda t: t)
        ref64 = cut(it64[key].detach())
        r_hip = H.worst_ratio(cut(hip.detach().cpu()), ref64, OUT_TOL)
        r_ref = max(H.worst_ratio(cut(it[key].detach()), ref64, OUT_TOL), H.worst_ratio(cut(itp[key].detach()), ref64, OUT_TOL))
        print("  output %-14s HIP %.3f x the element-wise bound against fp64, fp32 oracle %.3f x" % (key, r_hip, r_ref))
        assert r_hip <= max(1.0, FULL_SIZE_FACTOR * r_ref), f"{key}: HIP {r_hip:.2f} x the bound from fp64, the fp32 oracle {r_ref:.2f} x"
    gmax = max(float(v.grad.abs().max()) for v in sd_o.values() if v.grad is not None)
    worst = ("", 0.0, 0.0)
    for pname, p in model.named_parameters():
        ref_grad = sd_o[pname].grad
        if ref_grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{pname} should have zero gradient"
        elif float(ref_grad.abs().max()) < 1e-6 * gmax:
            assert float(p.grad.abs().max()) < 1e-5 * gmax, f"{pname} should be ~0"
        else:
            r_hip = H.worst_ratio(p.grad.cpu(), sd_64[pname].grad, GRAD_TOL)
            r_ref = max(H.worst_ratio(ref_grad, sd_64[pname].grad, GRAD_TOL), H.worst_ratio(sd_p[pname].grad, sd_64[pname].grad, GRAD_TOL))
            if r_hip > worst[1]:
                worst = (pname, r_hip, r_ref)
            assert r_hip <= max(1.0, FULL_SIZE_FACTOR * r_ref), (f"grad {pname}: HIP is {r_hip:.2f} x the element-wise bound away from the fp64 "
                                                                  f"gradient, the fp32 oracle (worse of two summation orders) {r_ref:.2f} x")
    print("worst ratio to the element-wise bound against the fp64 gradient: %s HIP %.3f x, fp32 oracle %.3f x" % worst)


def test_shapes_outside_the_kernels_build_are_reported(cuda_device, monkeypatch):
    """A constructor argument the HIP kernels are not built for (here: 4 attention heads, an EGNN width of 32) runs as a device-side
    torch composition -- tested against the oracle above, but not the measured path: the first use warns, ``functional.COMPOSED_PATHS``
    counts, ``bench.py`` prints the counts (``config.composed_paths``); the reference's defaults leave the registry empty."""
    import warnings
    from immunostruct_amd import functional as HF
    dev = cuda_device
    raw = synthetic.make_batch(4, seed=3)
    seq, prop = torch.from_numpy(raw.one_hot_sequence()).to(dev), torch.from_numpy(raw.prop).to(dev)
    monkeypatch.setattr(HF, "COMPOSED_PATHS", {})
    with warnings.catch_warnings():
        warnings.simplefilter("error")      # the default model must not warn
        for name in ("HybridModelv2", "StructureModel", "HybridModelv2_Comparative"):
            model = model_map[name](vae_input_dim=H.VAE_IN, device=dev).to(dev)
            g = H.product_graph(raw, dev)
            (model.forward_comparative((g, g), (seq, seq), (prop, prop)) if name.endswith("Comparative") else model(g, seq, prop))
    assert HF.COMPOSED_PATHS == {}
    model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev, self_attention_heads=4).to(dev)
    with pytest.warns(RuntimeWarning, match="node attention with 4 head.*outside the HIP kernels' build"):
        model(H.product_graph(raw, dev), seq, prop)
    with warnings.catch_warnings():
        warnings.simplefilter("error")      # ... once per reason
        model(H.product_graph(raw, dev), seq, prop)
    assert list(HF.COMPOSED_PATHS.values()) == [2]
    model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev, gat_hidden_channels=32).to(dev)
    with pytest.warns(RuntimeWarning, match="EGNN layers of sizes"):
        model(H.product_graph(raw, dev), seq, prop)
    assert len(HF.COMPOSED_PATHS) == 4 and all(v >= 1 for v in HF.COMPOSED_PATHS.values())      # (+ the width-32 attention, in full)
    # the max-pooling ablation needs the full attention output: torch ops by design, and said so
    monkeypatch.setattr(HF, "COMPOSED_PATHS", {})
    model = model_map["StructureModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
    with pytest.warns(RuntimeWarning, match="full \\(n x d\\) node attention output"):
        model(H.product_graph(raw, dev), seq, prop)


@pytest.mark.parametrize("name,hidden,seed", [("HybridModelv2", 32, 41), ("HybridModelv2", 128, 41), ("StructureModelv2", 48, 42)])
def test_other_hidden_sizes_vs_oracle(cuda_device, name, hidden, seed):
    """``gat_hidden_channels`` is a constructor argument of the reference's models (hybrid_models.py:247).  The HIP layer kernels are
    built for its default, 64; any other width runs the EGNN stack and the node attention as device-side torch ops with the same
    fixed summation order (nn.egnn_conv_composed, models/layers.py), everything else -- sequence VAE, fusion head, losses -- on the
    HIP kernels.  Loss and every parameter gradient vs the oracle, and the refusal of a HIP-graph capture.
    (The max-pooling variant runs on batch seed 42: in the seed-41 batch two residues of one graph come out of the attention
    within one fp32 ulp of each other in a channel they lead, so WHICH of them the max selects -- and with it 1 % of the gradient
    of every upstream parameter -- depends on the last bit of the forward: the fp64 oracle and any two fp32 implementations
    disagree there, whatever the width.)"""
    dev = cuda_device
    b = 8
    raw = synthetic.make_batch(b, seed=seed, deg_extra=3)
    model = model_map[name](vae_input_dim=H.VAE_IN, device=dev, gat_hidden_channels=hidden).to(dev)
    assert not model.GCN_layers[0].native
    sd = H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=29)
    model.load_state_dict(sd)
    model.eval()
    eps, y = H.make_eps(7, b), torch.from_numpy(raw.y_reg)
    seq, prop = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop)
    seq_loss = not name.startswith("StructureModel")

    def oracle(dtype):
        sd_o = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
        it = FR.forward(name, sd_o, H.oracle_graph(raw, dtype), seq.to(dtype), prop.to(dtype), eps=eps.to(dtype))
        if seq_loss:
            lo = FR.regression_loss(it["recon_x"], seq.to(dtype), it["mu"], it["logvar"], it["final_output"], y.to(dtype), H.VAE_IN)
        else:
            lo = FR.regression_loss(None, seq, None, None, it["final_output"], y.to(dtype), H.VAE_IN, sequence=False)
        lo.backward()
        return float(lo.detach()), sd_o
    lo, sd_o = oracle(torch.float32)
    lo64, sd_64 = oracle(torch.float64)
    g = H.product_graph(raw, dev)
    res = _with_eps(lambda: model(g, seq.to(dev), prop.to(dev)), [eps], dev)
    lh = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=seq_loss).regression_loss(res[0], seq.to(dev), res[1], res[2], res[3], y.to(dev))
    lh.backward()
    assert abs(float(lh.detach()) - lo64) <= 1e-5 * abs(lo64)
    gmax = max(float(v.grad.abs().max()) for v in sd_o.values() if v.grad is not None)
    for pname, p in model.named_parameters():
        ref_grad = sd_o[pname].grad
        if ref_grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{pname} should have zero gradient"
        elif float(ref_grad.abs().max()) < 1e-6 * gmax:
            assert p.grad is None or float(p.grad.abs().max()) < 1e-5 * gmax, f"{pname} should be ~0"
        else:
            r_hip = H.worst_ratio(p.grad.cpu(), sd_64[pname].grad, GRAD_TOL)
            r_ref = H.worst_ratio(ref_grad, sd_64[pname].grad, GRAD_TOL)
            assert r_hip <= max(1.0, FULL_SIZE_FACTOR * r_ref), (f"grad {pname}: HIP is {r_hip:.2f} x the element-wise bound away from the fp64 "
                                                    f"gradient, the fp32 oracle {r_ref:.2f} x")
    from immunostruct_amd.engine import StaticGraphBatch
    with pytest.raises(NotImplementedError):
        model(StaticGraphBatch(g, g.num_edges() + 8), seq.to(dev), prop.to(dev))


@pytest.mark.parametrize("optimizer", ["torch", "hip"])
@pytest.mark.parametrize("always_pack", [False, True, "serial", "auto"])
def test_captured_hip_graph_step_matches_eager(cuda_device, always_pack, optimizer, monkeypatch):
    """engine.CapturedTrainStep (HIP-graph replay on static, fixed-capacity buffers) reproduces eager training.

    Batches with MORE and with FEWER edges than the captured one are replayed.  (Tolerance, not bit
    equality: hipBLASLt may pick different GEMM algorithms for the captured VAE / attention linears.)"""
    from immunostruct_amd.distributed import FlatGradReducer
    from immunostruct_amd.engine import CapturedTrainStep
    dev = cuda_device
    # True: the data-parallel path (two-stage backward, two gradient buckets, three graphs); "serial": the same
    # with IMMUNOSTRUCT_DP_OVERLAP=0 (one backward graph, one bucket)
    # "auto": both forms captured, the faster one (timed at construction, no trace left in the model) is used
    monkeypatch.setenv("IMMUNOSTRUCT_DP_OVERLAP", {"serial": "0", "auto": "auto"}.get(always_pack, "1"))
    raws = [synthetic.make_batch(6, seed=s, deg_extra=d) for s, d in ((51, 2), (52, 4), (53, 1))]
    batches = [(H.product_graph(r, dev), torch.from_numpy(r.one_hot_sequence()).to(dev),
                torch.from_numpy(r.prop).to(dev), torch.from_numpy(r.y_reg).to(dev)) for r in raws]
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    eps = H.make_eps(9, 6).to(dev)

    def forward_loss(m, g, seq, prop, y):
        with mock.patch("torch.randn_like", lambda t: eps.to(t.dtype)):
            recon, mu, logvar, final = m(g, seq, prop)
        return losses.regression_loss(recon, seq, mu, logvar, final, y)

    def run(captured):
        model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
        model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=6))
        model.eval()
        red = FlatGradReducer(model.parameters(), world=1, always_pack=bool(always_pack))
        if optimizer == "torch":
            opt = torch.optim.Adam(model.parameters(), lr=1e-5, fused=True, capturable=True)
        else:
            from immunostruct_amd import optim
            opt = optim.Adam(model.parameters(), lr=1e-5)      # csrc/optimizer.hip, capturable by construction
        out = []
        if captured:
            # construction performs one eager step on batches[0] (optimizer state must exist before capture)
            eng = CapturedTrainStep(model, opt, red, forward_loss, batches[0], edge_capacity=max(r.num_edges for r in raws), warmup=1)
            if always_pack == "auto":
                # both forms were captured and timed; the one that lost was released with its graphs (round 6)
                assert eng.dp_times["serial_ms"] > 0 and eng.dp_times["two_stage_ms"] > 0 and set(eng._forms) == {eng.two_stage}
            else:
                assert eng.two_stage == (always_pack is True)
            if eng._late is not None:
                late = set(id(p) for p in eng._late)
                names = sorted(k for k, p in model.named_parameters() if id(p) in late)
                assert names and all(k.startswith("GCN_layers.") or k.startswith("self_attention.w_q") or k.startswith("self_attention.w_k")
                                     for k in names), names
                assert len(red.buckets) == 2
            for b in batches:
                out.append(float(eng(*b)))
        else:
            for b in [batches[0]] + batches:
                red.zero()
                loss = forward_loss(model, *b)
                loss.backward()
                red.all_reduce_mean()
                opt.step()
                out.append(float(loss.detach()))
        return out, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    l_e, sd_e = run(False)
    l_c, sd_c = run(True)
    l_e = l_e[1:]   # drop the warm-up step's loss
    for a, b in zip(l_e, l_c):
        assert abs(a - b) <= 1e-5 * abs(a), (l_e, l_c)
    for k in sd_e:
        H.assert_close(sd_c[k], sd_e[k], 1e-5, f"{k} after 3 captured steps")


@pytest.mark.parametrize("promise", [False, True])
def test_loss_that_reads_the_reconstruction_with_torch_ops_gets_the_full_join(cuda_device, promise):
    """the models let the main stream join the sequence branch at the LATENT (``models/_core.py`` EARLY_JOIN) -- safe only
    when the loss reads ``recon_x`` through ``functional.vae_loss``.  A ``forward_loss`` WITHOUT the ``fused_loss`` promise that uses
    ``F.mse_loss`` on the reconstruction must get the full join (every forward counted), a promised one the early join; the captured
    step of the custom loss equals the eager one."""
    import torch.nn.functional as F
    from immunostruct_amd import optim
    from immunostruct_amd.distributed import FlatGradReducer
    from immunostruct_amd.engine import CapturedTrainStep
    from immunostruct_amd.models import _core
    dev = cuda_device
    raws = [synthetic.make_batch(6, seed=s, deg_extra=2) for s in (61, 62)]
    batches = [(H.product_graph(r, dev), torch.from_numpy(r.one_hot_sequence()).to(dev), torch.from_numpy(r.prop).to(dev),
                torch.from_numpy(r.y_reg).to(dev)) for r in raws]
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    eps = H.make_eps(9, 6).to(dev)

    def forward_loss(m, g, seq, prop, y):
        with mock.patch("torch.randn_like", lambda t: eps.to(t.dtype)):
            recon, mu, logvar, final = m(g, seq, prop)
        if promise:
            return losses.regression_loss(recon, seq, mu, logvar, final, y)
        return 2.0 * F.mse_loss(final.flatten(), y) + 0.5 * F.mse_loss(recon, seq.reshape(recon.shape)) - 0.25 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())
    forward_loss.fused_loss = promise

    def run(captured):
        model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
        model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=6))
        model.eval()
        red = FlatGradReducer(model.parameters(), world=1)
        opt = optim.Adam(model.parameters(), lr=1e-5)
        out = []
        if captured:
            before = dict(_core.JOIN_COUNTS)
            eng = CapturedTrainStep(model, opt, red, forward_loss, batches[0], edge_capacity=max(r.num_edges for r in raws), warmup=1)
            took = {k: _core.JOIN_COUNTS[k] - before[k] for k in before}
            assert took["early" if promise else "full"] >= 2 and took["full" if promise else "early"] == 0, took
            for b in batches:
                out.append(float(eng(*b)))
        else:
            for b in [batches[0]] + batches:
                red.zero()
                loss = forward_loss(model, *b)
                loss.backward()
                opt.step()
                out.append(float(loss.detach()))
        torch.cuda.synchronize()
        return out, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    l_e, sd_e = run(False)
    l_c, sd_c = run(True)
    for a, b in zip(l_e[1:], l_c):
        assert abs(a - b) <= 1e-5 * abs(a), (l_e, l_c)
    for k in sd_e:
        H.assert_close(sd_c[k], sd_e[k], 1e-5, f"{k} after 2 captured steps")


def test_full_step_is_deterministic_at_full_residency(cuda_device):
    """utils/seed.py:18 (deterministic algorithms) is part of the reference's contract: the whole captured step of the benchmark's
    size -- B = 128 x 190 nodes, every workgroup slot of the chip taken by the persistent layer kernels -- replayed twice from the
    same weights gives bit-identical loss and gradients.  The regression test of round 4's `buffer_store_dwordx4` + SGPR-offset
    hazard (HISTORY 7.4): that fault left B = 32 bit-stable and changed the gradients from run to run only at full residency.
    (lr = 0, no weight decay: the update leaves the weights where they are, the moments do not feed back into the gradients.)"""
    from immunostruct_amd import optim
    from immunostruct_amd.distributed import FlatGradReducer
    from immunostruct_amd.engine import CapturedTrainStep
    dev = cuda_device
    b = 128
    raw = synthetic.make_batch(b, seed=1, deg_extra=2)
    batch = (H.product_graph(raw, dev), torch.from_numpy(raw.one_hot_sequence()).to(dev), torch.from_numpy(raw.prop).to(dev),
             torch.from_numpy(raw.y_reg).to(dev))
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    eps = H.make_eps(9, b).to(dev)

    def forward_loss(m, g, seq, prop, y):
        with mock.patch("torch.randn_like", lambda t: eps.to(t.dtype)):
            recon, mu, logvar, final = m(g, seq, prop)
        return losses.regression_loss(recon, seq, mu, logvar, final, y)

    model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
    model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=6))
    model.eval()
    red = FlatGradReducer(model.parameters(), world=1)
    opt = optim.Adam(model.parameters(), lr=0.0)
    eng = CapturedTrainStep(model, opt, red, forward_loss, batch, edge_capacity=raw.num_edges, warmup=1)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    runs = []
    for _ in range(3):
        loss = eng(*batch)
        torch.cuda.synchronize()
        runs.append((float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd0[k]), f"{k} moved under lr = 0"
    assert len(runs[0][1]) > 60
    for li, gi in runs[1:]:
        assert li == runs[0][0], (li, runs[0][0])
        assert gi.keys() == runs[0][1].keys()
        for k in gi:
            assert torch.equal(gi[k], runs[0][1][k]), f"{k}: the gradient of two replays of the same step differs in its bits"


def test_device_side_step_random_tensors_train_reproducibly(cuda_device):
    """``CapturedTrainStep(step_random="device")``: a model in TRAINING mode draws its dropout masks and noise from the library's
    generator inside the captured step (one launch per step, no torch generator in the graph): the same seed gives the same
    trajectory, another seed another one (the distributions themselves: ``test_step_random_launch_...`` in test_gpu_kernels.py)."""
    from immunostruct_amd.distributed import FlatGradReducer
    from immunostruct_amd.engine import CapturedTrainStep
    from immunostruct_amd import optim
    dev = cuda_device
    raws = [synthetic.make_batch(6, seed=s, deg_extra=d) for s, d in ((61, 2), (62, 3), (63, 1), (64, 2))]
    batches = [(H.product_graph(r, dev), torch.from_numpy(r.one_hot_sequence()).to(dev),
                torch.from_numpy(r.prop).to(dev), torch.from_numpy(r.y_reg).to(dev)) for r in raws]
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)

    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final = m(g, seq, prop)
        return losses.regression_loss(recon, seq, mu, logvar, final, y)

    from immunostruct_amd import functional as HF

    def run(seed, mode, stream=0):
        model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
        model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=8))
        model.train()
        torch.manual_seed(seed)
        eng = CapturedTrainStep(model, optim.Adam(model.parameters(), lr=1e-4), FlatGradReducer(model.parameters(), world=1), forward_loss,
                                batches[0], edge_capacity=max(r.num_edges for r in raws), warmup=1, step_random=mode, random_stream=stream)
        if mode == "device":
            assert [s["kind"] for s in eng._rand.slots] == ["dropout", "randn", "dropout"]
        out = [float(eng(*b)) for b in batches + batches]
        if mode == "device":
            assert int(eng._rand.state[1]) == 3 + 8      # three single launches of the eager step, then one per step
        return out

    a, b, c = run(5, "device"), run(5, "device"), run(6, "device")
    assert a == b and a != c and len(set(a)) == len(a) and all(np.isfinite(a))
    # the engine of a run's SECOND stage (finetune: same device seed, stream 1) draws another sequence, not the first stage's again --
    # and the same one whatever the process built before it (the key is a property of the stage, not a process counter)
    d, e = run(5, "device", stream=1), run(5, "device", stream=1)
    assert d != a and d == e and all(np.isfinite(d))


@pytest.mark.parametrize("form", ["two_pass", "merged"])
def test_captured_paired_step_matches_eager(cuda_device, form, monkeypatch):
    """The paired (cancer, wild-type) train step -- encoder on both members, fused head, BCE + reconstruction terms, paired
    contrastive loss -- as one captured HIP graph reproduces eager training, including a single-class batch, where the
    contrastive term must vanish (reference utils/contrastive.py:38-43: host early-out; here a device-side gate,
    PairedContrastiveLoss.capturable).  ``two_pass``: pairs of static buffers, one encoder pass per member;
    ``merged``: one batch of 2B graphs [cancer; wild-type], one encoder pass (what the on-GPU batcher delivers)."""
    from immunostruct_amd.distributed import FlatGradReducer
    from immunostruct_amd.engine import CapturedTrainStep
    from immunostruct_amd import optim
    from immunostruct_amd.graph import batch as graph_batch
    from immunostruct_amd.models import _core
    from immunostruct_amd.procedures.train import _paired_loss
    dev = cuda_device
    nb = 6
    monkeypatch.setattr(_core, "MERGE_PAIRS", form == "merged")
    raws = [(synthetic.make_batch(nb, seed=s, deg_extra=d), synthetic.make_batch(nb, seed=s + 100, deg_extra=d2))
            for s, d, d2 in ((61, 2, 3), (62, 4, 1), (63, 1, 2), (64, 3, 3))]
    targets = [torch.tensor(t, dtype=torch.float32, device=dev) for t in
               ([0, 1, 0, 0, 1, 0], [1, 1, 0, 0, 0, 1], [0, 0, 0, 0, 0, 0], [1, 0, 1, 1, 0, 0])]      # third: single class

    def tens(r):
        return H.product_graph(r, dev), torch.from_numpy(r.one_hot_sequence()).to(dev), torch.from_numpy(r.prop).to(dev)

    batches = []
    for (rc, rw), y in zip(raws, targets):
        (gc, sc, pc), (gw, sw, pw) = tens(rc), tens(rw)
        if form == "merged":
            gc.csr(), gw.csr()
            batches.append((graph_batch([gc, gw]), torch.cat([sc, sw]), torch.cat([pc, pw]), torch.cat([y, y])))
        else:
            batches.append(((gc, gw), (sc, sw), (pc, pw), y))
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    eps = [H.make_eps(19, nb).to(dev), H.make_eps(20, nb).to(dev)]
    if form == "merged":
        caps = max(rc.num_edges + rw.num_edges for rc, rw in raws)
    else:
        caps = (max(rc.num_edges for rc, _ in raws), max(rw.num_edges for _, rw in raws))

    def run(captured):
        model = model_map["HybridModelv2_Comparative"](vae_input_dim=H.VAE_IN, device=dev, use_wt_for_downstream=True).to(dev)
        model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=8))
        model.eval()        # dropout off: the two runs draw no random masks
        torch.manual_seed(0)
        contrastive = PairedContrastiveLoss(device=dev, embedding_dim=104)
        contrastive.capturable = captured
        opt = optim.AdamW(model.parameters(), lr=1e-5, weight_decay=1e-6)

        def forward_loss(m, graphs, seqs, props, y):
            it = iter(eps * 8)
            if form == "merged":
                y = y[:nb]
            with mock.patch("torch.randn_like", lambda t: next(it).to(t.dtype)):
                return _paired_loss(m, losses.BCE_loss, (graphs, seqs, y, props), dev, contrastive, 0.05)

        out = []
        if captured:
            red = FlatGradReducer(model.parameters(), world=1)
            eng = CapturedTrainStep(model, opt, red, forward_loss, batches[0], edge_capacity=caps, warmup=1)
            assert eng.paired == (form == "two_pass")
            for b in batches:
                out.append(float(eng(*b)))
        else:
            for b in [batches[0]] + batches:
                opt.zero_grad(set_to_none=True)
                loss = forward_loss(model, *b)
                loss.backward()
                opt.step()
                out.append(float(loss.detach()))
        return out, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    l_e, sd_e = run(False)
    l_c, sd_c = run(True)
    for a, b in zip(l_e[1:], l_c):
        assert abs(a - b) <= 1e-5 * abs(a), (l_e, l_c)
    for k in sd_e:
        H.assert_close(sd_c[k], sd_e[k], 1e-5, f"{k} after 4 captured paired steps ({form})")


def test_merged_pair_encoding_equals_two_passes(cuda_device, monkeypatch):
    """forward_comparative on the reference's 2-tuples: encoding [cancer; wild-type] as one batch (default) gives the
    outputs and parameter gradients of two encoder passes with shared weights (reference comparative_models.py:463-496)."""
    from immunostruct_amd.models import _core
    dev = cuda_device
    nb = 5
    rc, rw = synthetic.make_batch(nb, seed=71, deg_extra=2), synthetic.make_batch(nb, seed=72, deg_extra=4)
    args = ((H.product_graph(rc, dev), H.product_graph(rw, dev)),
            (torch.from_numpy(rc.one_hot_sequence()).to(dev), torch.from_numpy(rw.one_hot_sequence()).to(dev)),
            (torch.from_numpy(rc.prop).to(dev), torch.from_numpy(rw.prop).to(dev)))
    eps = [H.make_eps(3, nb).to(dev), H.make_eps(4, nb).to(dev)]
    res = {}
    for merge in (False, True):
        monkeypatch.setattr(_core, "MERGE_PAIRS", merge)
        model = model_map["HybridModelv2_Comparative"](vae_input_dim=H.VAE_IN, device=dev, use_wt_for_downstream=True).to(dev)
        model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=4))
        model.eval()
        it = iter(eps)
        with mock.patch("torch.randn_like", lambda t: next(it).to(t.dtype)):
            emb, recon, mu, logvar, final = model.forward_comparative(*args)
        outs = [emb[0], emb[1], recon[0], recon[1], mu[0], mu[1], logvar[0], logvar[1], final]
        w = [torch.linspace(-1, 1, t.numel(), device=dev).view_as(t) for t in outs]
        sum((t * wi).sum() for t, wi in zip(outs, w)).backward()
        res[merge] = ([t.detach().cpu() for t in outs], {k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None})
    for i, (a, b) in enumerate(zip(*[res[m][0] for m in (False, True)])):
        H.assert_close(b, a, 2e-6, f"output {i}")
    # the paired loss: one launch over the merged rows == 0.5 * (loss(cancer) + loss(wild-type)) of two passes
    from immunostruct_amd.procedures.train import _paired_loss
    y = torch.tensor([0.0, 1.0, 1.0, 0.0, 1.0], device=dev)
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    vals = {}
    for merge in (False, True):
        monkeypatch.setattr(_core, "MERGE_PAIRS", merge)
        model = model_map["HybridModelv2_Comparative"](vae_input_dim=H.VAE_IN, device=dev, use_wt_for_downstream=True).to(dev)
        model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=4))
        model.eval()
        it = iter(eps)
        with mock.patch("torch.randn_like", lambda t: next(it).to(t.dtype)):
            loss = _paired_loss(model, losses.BCE_loss, (args[0], args[1], y, args[2]), dev, None, 0.0)
        loss.backward()
        vals[merge] = (float(loss.detach()), {k: p.grad.cpu() for k, p in model.named_parameters() if p.grad is not None})
    assert abs(vals[True][0] - vals[False][0]) <= 2e-6 * abs(vals[False][0]), vals
    gmax = max(float(g.abs().max()) for g in vals[False][1].values())
    for k, g in vals[False][1].items():
        if float(g.abs().max()) >= 1e-6 * gmax:
            H.assert_close(vals[True][1][k], g, GRAD_TOL, f"paired-loss grad {k}")
    gmax = max(float(g.abs().max()) for g in res[False][1].values())
    for k, g in res[False][1].items():
        if float(g.abs().max()) < 1e-6 * gmax:
            continue                      # analytically zero (key bias): round-off on both sides
        H.assert_close(res[True][1][k], g, GRAD_TOL, f"grad {k}")


def _paired_inputs(nb, seed, dev):
    """(raw cancer, raw wild-type, merged device batch (g2, seq2, y, prop2) as bench.py's PairedWorkload / the on-GPU batcher build it)"""
    from immunostruct_amd.graph import batch as graph_batch
    rc, rw = synthetic.make_batch(nb, seed=seed, deg_extra=2), synthetic.make_batch(nb, seed=seed + 40, deg_extra=2)
    y = rc.y_bin.copy()
    y[0], y[1] = 0.0, 1.0            # both classes present: the contrastive term is live
    gc, gw = H.product_graph(rc, dev), H.product_graph(rw, dev)
    gc.csr(), gw.csr()
    g2 = graph_batch([gc, gw])
    seq2 = torch.cat([torch.from_numpy(rc.one_hot_sequence()), torch.from_numpy(rw.one_hot_sequence())]).to(dev)
    prop2 = torch.cat([torch.from_numpy(rc.prop), torch.from_numpy(rw.prop)]).to(dev)
    return rc, rw, torch.from_numpy(y), (g2, seq2, torch.from_numpy(y).to(dev), prop2)


def _oracle_paired_loss(sd, psd, rc, rw, y, eps, dtype=torch.float32, permute=None):
    """procedures/train.py:97-118 on the oracle: forward_comparative, (BCE(cancer) + BCE(wild-type)) / 2 + 0.01 * contrastive
    (``permute``: seed of another order of the edge lists -- another realisation of the fp32 round-off)"""
    cast = lambda a: torch.from_numpy(a).to(dtype)
    if permute is None:
        gs = (H.oracle_graph(rc, dtype), H.oracle_graph(rw, dtype))
    else:
        gs = (H.oracle_graph_permuted(rc, permute, dtype), H.oracle_graph_permuted(rw, permute + 1, dtype))
    seqs, props = (cast(rc.one_hot_sequence()), cast(rw.one_hot_sequence())), (cast(rc.prop), cast(rw.prop))
    o = FR.forward_comparative("HybridModelv2_Comparative", sd, gs, seqs, props, (eps[0].to(dtype), eps[1].to(dtype)),
                               use_wt_for_downstream=True)
    c, w, yy = o["cancer"], o["wt"], y.to(dtype)
    lc = FR.bce_loss(c["recon_x"], seqs[0], c["mu"], c["logvar"], o["final_output"], yy, H.VAE_IN, 81.0 / 19.0)
    lw = FR.bce_loss(w["recon_x"], seqs[1], w["mu"], w["logvar"], o["final_output"], yy, H.VAE_IN, 81.0 / 19.0)
    con = FR.paired_contrastive_loss({k: v.to(dtype) for k, v in psd.items() if v.is_floating_point()},
                                     o["embeddings"][0], o["embeddings"][1], yy)
    return (lc + lw) / 2 + 0.01 * con, con


@pytest.mark.parametrize("capturable", [True, False])
def test_paired_product_route_at_full_size_vs_oracle(cuda_device, capturable):
    """BASELINE config 4 at full size (B = 128 pairs = 256 graphs) through the route the loops and ``bench.py --workload paired``
    take -- ``procedures.train._paired_loss``: merged 2B-graph encoder pass, stacked-pair fusion head, ONE loss launch over the
    merged rows, paired contrastive loss on the side stream with the coefficient (and, capturable, the two-class gate) inside its
    launches -- against the oracle's literal ``forward_comparative`` + two ``BCE_loss`` + ``PairedContrastiveLoss``
    (procedures/train.py:97-118, utils/contrastive.py:37-83): loss 1e-5, contrastive term 1e-4 (north star), every parameter
    gradient with the fp64 yardstick of ``test_full_train_step_gradients_vs_oracle`` (see there for the factor)."""
    from immunostruct_amd.procedures.train import _paired_loss
    dev = cuda_device
    nb = 128
    torch.set_num_threads(min(16, torch.get_num_threads()))
    rc, rw, y, batch = _paired_inputs(nb, 71, dev)
    model = model_map["HybridModelv2_Comparative"](vae_input_dim=H.VAE_IN, device=dev, use_wt_for_downstream=True).to(dev)
    sd = H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=17)
    model.load_state_dict(sd)
    model.eval()                      # dropout off; the reparameterisation noise is injected on both sides
    pcl = PairedContrastiveLoss(embedding_dim=104, device=dev)
    psd = H.det_sd({k: tuple(v.shape) for k, v in pcl.state_dict().items()}, seed=9)
    pcl.load_state_dict(psd)
    pcl.capturable = capturable
    eps = (H.make_eps(31, nb), H.make_eps(32, nb))
    seen = []
    inner = pcl.forward
    pcl.forward = lambda *a, **k: (seen.append(inner(*a, **k)), seen[-1])[1]      # the contrastive term as the route computed it
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    lh = _with_eps(lambda: _paired_loss(model, losses.BCE_loss, batch, dev, pcl, 0.01), list(eps), dev)
    lh.backward()
    torch.cuda.synchronize()
    # oracle, fp32 and fp64
    sd32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lo, con = _oracle_paired_loss(sd32, psd, rc, rw, y, eps)
    lo.backward()
    sd64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    lo64, con64 = _oracle_paired_loss(sd64, psd, rc, rw, y, eps, torch.float64)
    lo64.backward()
    sdp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}      # second fp32 realisation: permuted edge lists
    _oracle_paired_loss(sdp, psd, rc, rw, y, eps, permute=11)[0].backward()
    assert abs(float(lh.detach()) - float(lo64.detach())) <= 1e-5 * abs(float(lo64.detach())), (float(lh.detach()), float(lo64.detach()))
    assert len(seen) == 1 and torch.is_tensor(seen[0])
    got = float(seen[0].detach()) / 0.01          # the launches carry the coefficient
    assert abs(got - float(con64.detach())) <= 1e-4 * abs(float(con64.detach())), (got, float(con64.detach()), float(con.detach()))
    gmax = max(float(v.grad.abs().max()) for v in sd32.values() if v.grad is not None)
    worst = ("", 0.0)
    for name, p in model.named_parameters():
        ref = sd32[name].grad
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{name} should have zero gradient"
            continue
        if float(ref.abs().max()) < 1e-6 * gmax:
            assert float(p.grad.abs().max()) < 1e-5 * gmax, f"{name} should be ~0"
            continue
        r_hip = H.worst_ratio(p.grad.cpu(), sd64[name].grad, GRAD_TOL)
        r_ref = max(H.worst_ratio(ref, sd64[name].grad, GRAD_TOL), H.worst_ratio(sdp[name].grad, sd64[name].grad, GRAD_TOL))
        assert r_hip <= max(1.0, FULL_SIZE_FACTOR * r_ref), (f"grad {name}: HIP is {r_hip:.2f} x the element-wise bound away from the fp64 gradient, "
                                                f"the fp32 oracle (worse of two summation orders) {r_ref:.2f} x")
        if r_hip > worst[1]:
            worst = (name, r_hip, r_ref)
    print("paired route, worst gradient (x bound vs fp64; fp32 oracle):", worst)


def test_paired_training_trajectory_matches_oracle(cuda_device):
    """10 AdamW steps (lr 1e-4, weight decay 1e-6: train_Cancer_wFT.py:143-147) of the paired step through
    ``procedures.train._paired_loss`` on fresh pair batches vs the CPU oracle: forward, merged loss, side-stream contrastive
    loss, backward and optimizer, compounded."""
    from immunostruct_amd import optim
    from immunostruct_amd.procedures.train import _paired_loss
    dev = cuda_device
    nb, steps = 32, 10
    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(0)
    model = model_map["HybridModelv2_Comparative"](vae_input_dim=H.VAE_IN, device=dev, use_wt_for_downstream=True).to(dev)
    model.eval()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    pcl = PairedContrastiveLoss(embedding_dim=104, device=dev)
    psd = {k: v.detach().cpu().clone() for k, v in pcl.state_dict().items()}
    pcl.capturable = True
    # the weights after the steps are judged like the full-size gradients: against the oracle in fp64, with the fp32 oracle's own
    # distance from it as the yardstick -- AdamW turns the sign of a near-zero gradient into a full step, so two correct fp32
    # trajectories drift apart by more than the element-wise bound on a few elements, and how far the CPU oracle's fp32
    # realisation drifts depends on the host (thread count, BLAS blocking): against the fp32 oracle alone this test passed on
    # some boxes of the pool and failed on others (vae_fc1.weight at 1.3 x the bound)
    sd64 = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items()}
    opt_h = optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-6)
    opt_o = torch.optim.AdamW(list(sd.values()), lr=1e-4, weight_decay=1e-6)
    opt_64 = torch.optim.AdamW(list(sd64.values()), lr=1e-4, weight_decay=1e-6)
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    worst = 0.0
    for s in range(steps):
        rc, rw, y, batch = _paired_inputs(nb, 300 + s, dev)
        eps = (H.make_eps(700 + s, nb), H.make_eps(800 + s, nb))
        opt_h.zero_grad(set_to_none=True)
        lh = _with_eps(lambda: _paired_loss(model, losses.BCE_loss, batch, dev, pcl, 0.01), list(eps), dev)
        lh.backward()
        opt_h.step()
        opt_o.zero_grad()
        lo, _ = _oracle_paired_loss(sd, psd, rc, rw, y, eps)
        lo.backward()
        opt_o.step()
        opt_64.zero_grad()
        l64, _ = _oracle_paired_loss(sd64, psd, rc, rw, y, eps, dtype=torch.float64)
        l64.backward()
        opt_64.step()
        rel = abs(float(lh.detach()) - float(l64.detach())) / abs(float(l64.detach()))
        worst = max(worst, rel)
        assert rel <= 2e-5, f"step {s}: HIP {float(lh.detach()):.6f} vs fp64 oracle {float(l64.detach()):.6f} (fp32 oracle {float(lo.detach()):.6f})"
    worst_w = ("", 0.0, 0.0)
    for k, v in model.state_dict().items():
        if not v.is_floating_point():
            continue
        # (2e-4: AdamW normalises every gradient element by its own running magnitude, so an element whose gradient is a
        #  cancelling sum carries its relative round-off, 1e-3 - 1e-2 on a handful of vae_fc1.weight's 3 M elements, into the weight
        #  at full step size; measured worst over the boxes of the pool: 1.3 x the 1e-4 bound, the fp32 oracle 0.45 - 0.9 x)
        r_hip = H.worst_ratio(v.detach().cpu(), sd64[k].detach(), 2e-4)
        r_ref = H.worst_ratio(sd[k].detach(), sd64[k].detach(), 2e-4)
        if r_hip > worst_w[1]:
            worst_w = (k, r_hip, r_ref)
        assert r_hip <= max(1.0, TRAJECTORY_FACTOR * r_ref), (f"{k} after {steps} paired steps: HIP is {r_hip:.2f} x the element-wise bound away from "
                                                             f"the fp64 trajectory, the fp32 oracle {r_ref:.2f} x")
    print(f"worst relative loss difference over {steps} paired steps of B = {nb} pairs: {worst:.2e}; worst weights {worst_w}")


@pytest.mark.parametrize("nb,steps", [(12, 10), (128, 20)])
def test_training_trajectory_matches_oracle(cuda_device, nb, steps):
    """Adam steps at the reference's learning rate (1e-3, train_IEDB_wFT.py:19) on fresh batches: the HIP path's loss
    follows the CPU oracle's step by step (same initial weights, same reparameterisation noise) -- the whole chain
    forward / loss / backward / optimizer, compounded; 10 steps of B = 12 and 20 steps at BASELINE config 2's B = 128.
    (The un-normalised EGNN is unstable at this learning rate on synthetic data: a few steps after the B = 12 sequence
    ends BOTH implementations blow up at the same step -- loss 6160 vs 6127 at its step 13 -- which is where a
    trajectory comparison stops being meaningful.)"""
    from immunostruct_amd import optim
    from oracle import graph_ref
    dev = cuda_device
    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(0)
    model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
    model.eval()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt_h, opt_o = optim.Adam(model.parameters(), lr=1e-3), torch.optim.Adam(list(sd.values()), lr=1e-3)
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    worst = 0.0
    for s in range(steps):
        raw = synthetic.make_batch(nb, seed=500 + s, deg_extra=2)
        eps = H.make_eps(900 + s, nb)
        seq, prop, y = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop), torch.from_numpy(raw.y_reg)
        opt_h.zero_grad(set_to_none=True)
        with mock.patch("torch.randn_like", lambda t: eps.to(t.device, t.dtype)):
            recon, mu, logvar, final = model(H.product_graph(raw, dev), seq.to(dev), prop.to(dev))
        lh = losses.regression_loss(recon, seq.to(dev), mu, logvar, final, y.to(dev))
        lh.backward()
        opt_h.step()
        go = graph_ref.RefGraph(raw.src, raw.dst, raw.num_nodes, raw.batch_num_nodes)
        go.ndata["x"], go.edata["edge_attr"] = torch.from_numpy(raw.x), torch.from_numpy(raw.edge_attr)
        opt_o.zero_grad()
        with mock.patch("torch.randn_like", lambda t: eps.to(t.dtype)):
            it = FR.forward("HybridModelv2", sd, go, seq, prop)
        lo = FR.regression_loss(it["recon_x"], seq, it["mu"], it["logvar"], it["final_output"], y, H.VAE_IN)
        lo.backward()
        opt_o.step()
        rel = abs(float(lh.detach()) - float(lo.detach())) / abs(float(lo.detach()))
        worst = max(worst, rel)
        assert rel <= 1e-4, f"step {s}: hip {float(lh.detach())} oracle {float(lo.detach())}"
    print(f"worst relative loss difference over {steps} steps of B = {nb}: {worst:.2e}")


def test_entry_scripts_run_end_to_end(cuda_device, tmp_path):
    """both entry points: pretrain -> new head -> finetune -> inference on a small synthetic set (1 epoch)."""
    from immunostruct_amd import train_Cancer_wFT, train_IEDB_wFT
    common = ["--full-sequence", "--sequence-loss", "--num-epochs", "1", "--learning-rate-pretrain", "1e-4", "--batch-size", "16", "--synthetic", "160",
              "--model-save-dir", str(tmp_path)]
    train_IEDB_wFT.main(["--model", "HybridModelv2"] + common)
    train_IEDB_wFT.main(["--model", "HybridModelv2", "--device-dataset", "--seed", "3"] + common)      # on-GPU batcher + captured step
    train_Cancer_wFT.main(["--use-wt-for-downstream", "--coeff-contrastive", "0.01", "--min-finetuning-batches", "2"] + common)
    train_Cancer_wFT.main(["--use-wt-for-downstream", "--coeff-contrastive", "0.01", "--min-finetuning-batches", "2",
                           "--device-dataset", "--seed", "3"] + common)
    assert len(list(tmp_path.glob("*_finetune.pt"))) == 4


def test_entry_scripts_default_peptide_inputs(cuda_device, tmp_path):
    """the reference's DEFAULT command line has no --full-sequence: models are built 231 wide and fed the padded peptide
    (train_IEDB_wFT.py:22,59-60; train_Cancer_wFT.py:71-72).  Host-loader and on-device paths of both scripts; the
    sequence is never masked in this mode (--sequence-pad-count is ignored, data/util_dataloader.py:52-66)."""
    from immunostruct_amd import train_Cancer_wFT, train_IEDB_wFT
    common = ["--sequence-loss", "--num-epochs", "1", "--learning-rate-pretrain", "1e-4", "--batch-size", "16", "--synthetic", "160",
              "--model-save-dir", str(tmp_path)]
    for extra in ([], ["--device-dataset", "--seed", "3"]):
        tr, te = train_IEDB_wFT.main(["--model", "HybridModelv2"] + common + extra)
        assert np.isfinite(te["roc_auc"])
    ckpt = torch.load(next(iter(tmp_path.glob("HybridModelv2-*fseq_False*_finetune.pt"))), map_location="cpu")
    assert tuple(ckpt["vae_fc1.weight"].shape) == (512, 231) and tuple(ckpt["vae_fc4.weight"].shape) == (231, 512)
    tr, te = train_IEDB_wFT.main(["--model", "HybridModelv2_SSL", "--self-supervision", "--sequence-pad-count", "3",
                                  "--device-dataset", "--seed", "4"] + common)
    assert np.isfinite(te["roc_auc"])
    for extra in ([], ["--device-dataset", "--seed", "3"]):
        tr, te = train_Cancer_wFT.main(["--use-wt-for-downstream", "--coeff-contrastive", "0.01", "--min-finetuning-batches", "2"]
                                       + common + extra)
        assert np.isfinite(te["roc_auc"])


def test_entry_scripts_self_supervision(cuda_device, tmp_path):
    """--self-supervision: masked-residue augmentation -> 5-field batches -> *_SSL models and losses ->
    inference with the train-set Youden threshold; loader path, on-device (captured) path, and the paired script."""
    from immunostruct_amd import train_Cancer_wFT, train_IEDB_wFT
    common = ["--full-sequence", "--sequence-loss", "--num-epochs", "1", "--learning-rate-pretrain", "1e-4", "--batch-size", "16", "--synthetic", "160",
              "--self-supervision", "--structure-pad-count", "2", "--sequence-pad-count", "3", "--model-save-dir", str(tmp_path)]
    keys = {"optimal_threshold", "accuracy", "accuracy_op", "f1", "f1_op", "precision", "precision_op", "recall", "recall_op",
            "roc_auc", "pr_auc", "ppvn", "ppvn_op", "ppv30", "ppv30_op"}
    for extra in ([], ["--device-dataset", "--seed", "3"]):
        train_stats, test_stats = train_IEDB_wFT.main(["--model", "HybridModelv2_SSL"] + common + extra)
        assert set(train_stats) == keys and set(test_stats) == keys
        assert test_stats["optimal_threshold"] == train_stats["optimal_threshold"]
        assert all(np.isfinite(float(v)) for v in train_stats.values())
    for extra in ([], ["--device-dataset", "--seed", "3"]):      # host loaders; merged pair batches gathered + augmented on the GPU
        train_stats, test_stats = train_Cancer_wFT.main(["--model", "HybridModelv2_Comparative_SSL", "--use-wt-for-downstream",
                                                         "--coeff-contrastive", "0.01", "--min-finetuning-batches", "2"] + common + extra)
        assert set(test_stats) == keys and 0.0 <= test_stats["roc_auc"] <= 1.0


def test_entry_script_on_packed_file(cuda_device, tmp_path):
    """--packed: the .npz of data.PackedDataset feeds the host loader path and, array for array, the
    on-GPU batcher; the device path assembles the same batches as collate on the items of the same file."""
    from immunostruct_amd import train_IEDB_wFT
    from immunostruct_amd.data import DeviceResidentDataset, PackedDataset, collate
    raw = synthetic.make_batch(96, seed=21)
    n = int(raw.batch_num_nodes[0])
    gid = raw.dst // n
    graphs = [(torch.from_numpy(raw.x[i * n:(i + 1) * n]), torch.from_numpy(raw.src[gid == i] - i * n),
               torch.from_numpy(raw.dst[gid == i] - i * n)) for i in range(96)]
    tokens = raw.one_hot_sequence().argmax(-1)
    alphabet = "ACDEFGHIKLMNPQRSTVWYJ"
    names = [f"s{i}" for i in range(96)]
    full_of = lambda i: "".join(alphabet[t] for t in tokens[i]).rstrip("J")      # HLA (272) + peptide (9..11), unpadded
    labels = {nm: (full_of(i), float(raw.prop[i, 0]), float(raw.prop[i, 1]), float(raw.y_bin[i]),
                   float(raw.y_reg[i]), full_of(i)[272:]) for i, nm in enumerate(names)}
    path = str(tmp_path / "iedb_packed.npz")
    PackedDataset.from_graphs(graphs, names, labels=labels).save(path)
    packed = PackedDataset.load(path)
    dds = DeviceResidentDataset.from_packed(packed, cuda_device)
    idx = torch.tensor([5, 90, 17, 3], device=cuda_device)
    g, seq, prop, y = dds.gather_into(idx, *dds.new_batch(4))
    hg, hseq, hy, hprop = collate([packed[int(i)] for i in idx])
    assert torch.equal(g.ndata["x"].cpu(), hg.ndata["x"]) and torch.equal(seq.cpu(), hseq) and torch.equal(y.cpu(), hy)
    hc = hg.csr()
    e = hc.num_edges
    assert torch.equal(g._csr.src_sorted[:e].cpu(), hc.src_sorted) and torch.equal(g._csr.rowptr_dst.cpu(), hc.rowptr_dst)
    common = ["--model", "HybridModelv2", "--full-sequence", "--sequence-loss", "--num-epochs", "1", "--learning-rate-pretrain", "1e-4", "--batch-size", "16",
              "--packed", path, "--model-save-dir", str(tmp_path)]
    for extra in ([], ["--device-dataset", "--seed", "3"]):
        train_stats, test_stats = train_IEDB_wFT.main(common + extra)
        assert np.isfinite(test_stats["roc_auc"])
    # without --full-sequence the items carry the file's peptide tokens: the padded peptide = the tail of the padded full sequence
    packed.full_sequence = False
    assert torch.equal(packed[7][1], torch.from_numpy(raw.one_hot_sequence()[7, 272:]))
    pep = DeviceResidentDataset.from_packed(packed, cuda_device)
    assert tuple(pep.seq.shape[1:]) == (11, 21) and torch.equal(pep.seq[7].cpu(), packed[7][1])
    short = [a for a in common if a != "--full-sequence"]
    train_stats, test_stats = train_IEDB_wFT.main(short + ["--device-dataset", "--seed", "5"])
    assert np.isfinite(test_stats["roc_auc"])


def test_inference_matches_host_loop(cuda_device):
    """procedures.inference: probabilities collected on the device == per-batch sigmoid of the model's logits; metric
    dictionary == metric functions on those arrays (reference procedures/infer.py:9-50)."""
    from torch.utils.data import DataLoader
    from immunostruct_amd.data import SyntheticImmunoDataset, SyntheticPairedDataset, collate
    from immunostruct_amd.procedures import evaluate_metrics, find_optimal_threshold, inference, inference_comparative
    dev = cuda_device
    ds = SyntheticImmunoDataset(40, seed=8, binary=True)
    model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
    loader = DataLoader(ds, batch_size=16, collate_fn=collate)
    torch.manual_seed(77)          # the VAE samples eps in eval mode too (reference hybrid_models.py:297-300)
    out = inference(None, model, loader, dev, return_raw_preds=True, optimal_threshold=0.5322)
    probs = []
    model.eval()
    torch.manual_seed(77)
    with torch.no_grad():
        for g, seq, y, prop in loader:
            probs.append(torch.sigmoid(model(g.to(dev), seq.to(dev), prop.to(dev))[3]).reshape(-1).cpu())
    probs = torch.cat(probs).numpy().astype(np.float64)
    np.testing.assert_allclose(out["predicted_probs"], probs, rtol=1e-6)
    np.testing.assert_array_equal(out["true_targets"], ds.y.numpy())
    want = evaluate_metrics(out["true_targets"], out["predicted_probs"], 0.5322)
    assert all(out[k] == want[k] for k in want) and out["optimal_threshold"] == 0.5322
    # without a given threshold: Youden's J on this very set (an untrained model can sit below chance, where the
    # reference's assertion on the threshold fires -- same behaviour here)
    flipped = 1.0 - out["true_targets"] if out["roc_auc"] < 0.5 else out["true_targets"]
    thr = find_optimal_threshold(flipped, out["predicted_probs"])
    assert thr in out["predicted_probs"]
    with pytest.raises(NotImplementedError):
        inference(None, model, loader, dev, clinical_loader=loader)
    pairs = SyntheticPairedDataset(24, seed=9, binary=True)
    pm = model_map["HybridModelv2_Comparative"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
    pout = inference_comparative(None, pm, DataLoader(pairs, batch_size=8, collate_fn=collate), dev, return_raw_preds=True,
                                 optimal_threshold=0.5)
    assert pout["predicted_probs"].shape == (24,) and 0 <= pout["roc_auc"] <= 1


def test_device_batcher_matches_collate(cuda_device):
    """data.DeviceResidentDataset.gather_into (one HIP launch from graph ids) builds exactly the batch that the reference
    pipeline builds on the host: collate -> graph.batch -> CSR index construction (data/utils.py:160-176)."""
    from immunostruct_amd.data import DeviceResidentDataset, SyntheticImmunoDataset, collate
    ds = SyntheticImmunoDataset(14, seed=3, deg_extra=3)
    dds = DeviceResidentDataset(ds, cuda_device)
    buf = dds.new_batch(5)
    for ids in ([5, 2, 9, 0, 13], [1, 1, 7, 3, 12]):        # a second load overwrites the first (fewer / more edges)
        idx = torch.tensor(ids, dtype=torch.int64, device=cuda_device)
        sg, seq, prop, y = dds.gather_into(idx, *buf)
        g_ref, seq_ref, y_ref, prop_ref = collate([ds[i] for i in ids])
        c_ref, c = g_ref.csr(), sg.csr()
        e = g_ref.num_edges()
        assert torch.equal(sg.ndata["x"].cpu(), g_ref.ndata["x"])
        assert torch.equal(c.rowptr_dst.cpu(), c_ref.rowptr_dst) and torch.equal(c.rowptr_src.cpu(), c_ref.rowptr_src)
        assert torch.equal(c.src_sorted[:e].cpu(), c_ref.src_sorted) and torch.equal(c.dst_sorted[:e].cpu(), c_ref.dst_sorted)
        assert torch.equal(c.pos_by_src[:e].cpu(), c_ref.pos_by_src)
        assert torch.equal(sg.edge_feat_csr(None)[:e].cpu(), g_ref.edge_feat_csr(g_ref.edata["edge_attr"]))
        assert torch.equal(seq.cpu(), seq_ref) and torch.equal(prop.cpu(), prop_ref) and torch.equal(y.cpu(), y_ref.float())
        # the work partitions follow the new rowptr
        for k in (8, 64):
            assert torch.equal(c.chunks(k).cpu(), c_ref.chunks(k))
        sg.refresh_partitions()
        assert torch.equal(c.chunks(8).cpu(), c_ref.chunks(8))


def test_train_model_device_runs_and_preserves_the_start_state(cuda_device, tmp_path):
    """procedures.train_model_device: device-resident dataset + on-GPU batcher + captured step.  With zero epochs the
    engine's warm-up must leave model and optimizer untouched; with two epochs it trains (finite, improving loss)."""
    from types import SimpleNamespace
    from immunostruct_amd import optim
    from immunostruct_amd.data import DeviceResidentDataset, SyntheticImmunoDataset
    from immunostruct_amd.procedures import train_model_device
    dev = cuda_device
    ds = SyntheticImmunoDataset(44, seed=5)
    dds = DeviceResidentDataset(ds, dev)
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    torch.manual_seed(0)
    model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
    opt = optim.Adam(model.parameters(), lr=1e-3)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = SimpleNamespace(batch_size=8, num_epochs=0, model_save_path_pretrain=str(tmp_path / "m.pt"), model_save_path_finetune=str(tmp_path / "f.pt"))
    train_model_device(cfg, dev, model, dds, range(36), range(36, 44), opt, losses.regression_loss)
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    cfg.num_epochs = 3
    tl, vl = train_model_device(cfg, dev, model, dds, range(36), range(36, 44), opt, losses.regression_loss)
    assert len(tl) == 3 and all(np.isfinite(tl)) and all(np.isfinite(vl)) and tl[-1] < tl[0]
    assert (tmp_path / "m.pt").exists()
