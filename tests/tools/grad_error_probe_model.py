"""Second half of grad_error_probe.py (test infrastructure: uses the oracle): the FULL model at BASELINE config 2's size, as
test_full_train_step_gradients_vs_oracle runs it.  The stack alone, under a random dense upstream gradient, is as accurate as
the fp32 oracle (grad_error_probe.py) -- so what reaches the stack from above?  Captures dL/dh_L (the gradient at the EGNN stack's
output, direct part + the part through the fused query / key projection) on both sides and prints RMS-relative errors against
the fp64 oracle, then the parameter gradients the same way.

    python tests/tools/grad_error_probe_model.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from immunostruct_amd import synthetic  # noqa: E402
from immunostruct_amd.models import _core, model_map  # noqa: E402
from immunostruct_amd.utils import Losses  # noqa: E402
from oracle import functional_ref as FR  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.tools.grad_error_probe import rms_rel  # noqa: E402


def main():
    import unittest.mock as mock
    dev = torch.device("cuda:0")
    b, n_pad = 128, 190
    raw = synthetic.make_batch(b, seed=33, deg_extra=5, n_pad=n_pad, n_real_choices=(n_pad - 2, n_pad - 1, n_pad))
    sd = H.det_sd(H.model_shapes("HybridModelv2"), seed=14)
    eps = H.make_eps(5, b)
    y = torch.from_numpy(raw.y_reg)
    seq, prop = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop)
    ref = {}
    for dt in (torch.float32, torch.float64):
        sdd = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
        outs = []
        real = FR.egnn_conv

        def conv(*a, **k):
            h, x = real(*a, **k)
            h.retain_grad()
            outs.append(h)
            return h, x
        with mock.patch.object(FR, "egnn_conv", conv):
            it = FR.forward("HybridModelv2", sdd, H.oracle_graph(raw, dt), seq.to(dt), prop.to(dt), eps=eps.to(dt))
        for k in ("x_gat_node", "z_vae", "final_output"):
            it[k].retain_grad()
        FR.regression_loss(it["recon_x"], seq.to(dt), it["mu"], it["logvar"], it["final_output"], y.to(dt), H.VAE_IN).backward()
        mid = {k: it[k].grad for k in ("x_gat_node", "z_vae", "final_output")}
        ref[dt] = dict(h=outs[-1].detach(), dh=outs[-1].grad, grads={k: v.grad for k, v in sdd.items() if v.grad is not None},
                       it={k: v.detach() for k, v in it.items() if torch.is_tensor(v)}, mid=mid)
    model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
    model.load_state_dict(sd)
    model.eval()
    cap = {}
    real_fwd = _core.egnn_stack_forward

    def fwd(*a, **k):
        res = real_fwd(*a, **k)
        cap["h"] = res[0]
        res[0].register_hook(lambda g: cap.__setitem__("g_h", g.detach().clone()))
        if len(res) > 2 and res[2] is not None:
            res[2].register_hook(lambda g: cap.__setitem__("g_qk", g.detach().clone()))
        return res
    real_enc = type(model)._encode

    def enc(self, *a, **k):
        o = real_enc(self, *a, **k)
        cap["pooled"], cap["z_vae"] = o.get("x_gat_node"), o.get("z_vae")
        for k in ("x_gat_node", "z_vae"):
            if torch.is_tensor(o.get(k)) and o[k].requires_grad:
                o[k].register_hook(lambda g, k=k: cap.__setitem__("g_" + k, g.detach().clone()))
        return o
    g = H.product_graph(raw, dev)
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    with mock.patch.object(_core, "egnn_stack_forward", fwd), mock.patch("torch.randn_like", lambda t: eps.to(t.device, t.dtype)), \
            mock.patch.object(type(model), "_encode", enc):
        res = model(g, seq.to(dev), prop.to(dev))
    losses.regression_loss(res[0], seq.to(dev), res[1], res[2], res[3], y.to(dev)).backward()
    f64, f32 = ref[torch.float64], ref[torch.float32]
    print("forward: RMS-relative error against fp64 (HIP, fp32 oracle):")
    hip_fw = dict(recon_x=res[0], mu=res[1], logvar=res[2], final_output=res[3], x_gat_node=cap.get("pooled"), z_vae=cap.get("z_vae"))
    for k in ("x_gat_node", "mu", "logvar", "z_vae", "recon_x", "final_output"):
        if hip_fw.get(k) is not None and k in f64["it"]:
            print(f"  {k:14s} {rms_rel(hip_fw[k].detach().reshape(f64['it'][k].shape), f64['it'][k]):.2e} {rms_rel(f32['it'][k], f64['it'][k]):.2e}")
    z64, z32, zh = f64["it"]["final_output"].reshape(-1), f32["it"]["final_output"].reshape(-1), res[3].detach().double().cpu().reshape(-1)
    t = y.double().reshape(-1)
    print("  d loss / d logit = 2 (z - t) / B: HIP %.2e, fp32 oracle %.2e   (rms z %.3f, rms z - t %.3f)" % (
        rms_rel(zh - t, z64 - t), rms_rel(z32.double() - t, z64 - t), float(z64.pow(2).mean().sqrt()), float((z64 - t).pow(2).mean().sqrt())))
    res[3].register_hook(lambda g: cap.__setitem__("g_final_output", g.detach().clone())) if False else None
    # the fusion head's two stages in isolation, on the HIP path's own inputs: closed-form combined attention -> z, classifier -> logit
    from immunostruct_amd import functional as HF
    with torch.no_grad():
        pieces = [cap["pooled"].detach(), cap["z_vae"].detach()]
        z_hip = HF.combined_attention_mean(pieces, model.combined_attention)
        comb = torch.cat(pieces, dim=1).cpu()
        zs = {}
        for dt in (torch.float32, torch.float64):
            sdd = {k: v.to(dt) for k, v in sd.items()}
            cc, _ = FR.multi_head_attention(sdd, "combined_attention.", comb.to(dt).unsqueeze(2), 8)
            zs[dt] = cc.mean(dim=2)
        print("  combined attention z on the HIP path's inputs: HIP %.2e, fp32 oracle %.2e   (rms z %.3e, rms of the inputs %.3e)" % (
            rms_rel(z_hip, zs[torch.float64]), rms_rel(zs[torch.float32], zs[torch.float64]), float(zs[torch.float64].pow(2).mean().sqrt()),
            float(comb.double().pow(2).mean().sqrt())))
        for dt, zin in ((torch.float32, z_hip.cpu()), (torch.float64, z_hip.cpu().double())):
            sdd = {k: v.to(dt) for k, v in sd.items()}
            hid = torch.relu(torch.nn.functional.linear(zin, sdd["classifier.1.weight"], sdd["classifier.1.bias"]))
            zs[("cls", dt)] = torch.nn.functional.linear(hid, sdd["classifier.4.weight"], sdd["classifier.4.bias"])
        print("  classifier on HIP's z: fp32 torch %.2e (against fp64 torch on the same z); HIP's logit against that fp64: %.2e" % (
            rms_rel(zs[("cls", torch.float32)], zs[("cls", torch.float64)]), rms_rel(res[3].detach().reshape(-1, 1), zs[("cls", torch.float64)].reshape(-1, 1))))
    print("backward, gradients at the head's inputs: RMS-relative error against fp64 (HIP, fp32 oracle):")
    for k in ("x_gat_node", "z_vae"):
        if "g_" + k in cap:
            print(f"  dL/d {k:12s} {rms_rel(cap['g_' + k].reshape(f64['mid'][k].shape), f64['mid'][k]):.2e} {rms_rel(f32['mid'][k], f64['mid'][k]):.2e}")
    print("stack output h_L: RMS-relative error against fp64: HIP %.2e, fp32 oracle %.2e" % (rms_rel(cap["h"], f64["h"]), rms_rel(f32["h"], f64["h"])))
    dh = cap["g_h"].double().cpu()
    if "g_qk" in cap:
        wq, wk = sd["self_attention.w_q.weight"].double(), sd["self_attention.w_k.weight"].double()
        gq = cap["g_qk"].double().cpu()
        dh_direct = dh.clone()
        dh = dh + gq[:, :64] @ wq + gq[:, 64:] @ wk
        print("   (direct part rms %.3e, through q / k rms %.3e)" % (float(dh_direct.pow(2).mean().sqrt()), float((dh - dh_direct).pow(2).mean().sqrt())))
    print("dL/dh_L (what the stack's backward starts from): HIP %.2e, fp32 oracle %.2e   (rms of the gradient %.3e, max %.3e)" % (
        rms_rel(dh, f64["dh"]), rms_rel(f32["dh"], f64["dh"]), float(f64["dh"].pow(2).mean().sqrt()), float(f64["dh"].abs().max())))
    print("parameter gradients: RMS-relative error against fp64: HIP, fp32 oracle, ratio | worst element / (1e-4 max|g|): HIP, fp32 oracle")
    for k, p in model.named_parameters():
        if k not in f64["grads"] or p.grad is None or float(f64["grads"][k].abs().max()) == 0.0:
            continue
        g64, g32 = f64["grads"][k], f32["grads"][k]
        e, o = rms_rel(p.grad, g64), rms_rel(g32, g64)
        flag = " <--" if H.worst_ratio(p.grad.cpu(), g64, 1e-4) > 1.0 else ""
        print(f"  {k:42s} {e:.2e} {o:.2e} {e / max(o, 1e-300):6.2f} | {H.worst_ratio(p.grad.cpu(), g64, 1e-4):6.2f} {H.worst_ratio(g32, g64, 1e-4):6.2f}{flag}")


if __name__ == "__main__":
    main()
