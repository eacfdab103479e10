import sys; sys.path.insert(0,'/root/repo')
import unittest.mock as mock
import torch
from immunostruct_amd import synthetic
from immunostruct_amd.models import model_map
from immunostruct_amd.utils import Losses
from oracle import functional_ref as FR
from tests import helpers as H
dev = torch.device("cuda:0")
for name, heads, b in (("HybridModelv2", 2, 5), ("HybridModelv2", 1, 5), ("HybridModelv2", 2, 16)):
    raw = synthetic.make_batch(b, seed=61, deg_extra=3)
    model = model_map[name](vae_input_dim=H.VAE_IN, device=dev, self_attention_heads=heads).to(dev)
    sd = H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=23)
    model.load_state_dict(sd); model.eval()
    eps, y = H.make_eps(6, b), torch.from_numpy(raw.y_reg)
    seq, prop = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop)
    def oracle(dtype):
        sd_o = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
        it = FR.forward(name, sd_o, H.oracle_graph(raw, dtype), seq.to(dtype), prop.to(dtype), eps=eps.to(dtype), heads=heads)
        it["z_vae"].retain_grad(); it["x_gat_node"].retain_grad(); it["mu"].retain_grad(); it["logvar"].retain_grad()
        lo = FR.regression_loss(it["recon_x"], seq.to(dtype), it["mu"], it["logvar"], it["final_output"], y.to(dtype), H.VAE_IN)
        lo.backward()
        return sd_o, it
    s32, i32 = oracle(torch.float32); s64, i64 = oracle(torch.float64)
    it = iter([eps])
    with mock.patch("torch.randn_like", lambda t: next(it).to(device=dev, dtype=t.dtype)):
        res = model(H.product_graph(raw, dev), seq.to(dev), prop.to(dev))
    for t in res[1:3]: t.retain_grad()
    lh = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True).regression_loss(res[0], seq.to(dev), res[1], res[2], res[3], y.to(dev))
    lh.backward()
    print(name, heads, b)
    for k in ("vae_fc22.bias", "vae_fc22.weight", "vae_fc21.bias", "vae_fc3.weight", "vae_fc3.bias", "classifier.1.weight", "combined_attention.w_q.weight", "vae_fc1.bias"):
        p = dict(model.named_parameters())[k]
        print(f"   {k:34s} hip {H.worst_ratio(p.grad.cpu(), s64[k].grad, 1e-4):6.3f}  oracle32 {H.worst_ratio(s32[k].grad, s64[k].grad, 1e-4):6.3f}")
    print("   d logvar: hip", H.worst_ratio(res[2].grad.cpu(), i64["logvar"].grad, 1e-4), "oracle32", H.worst_ratio(i32["logvar"].grad, i64["logvar"].grad, 1e-4))
    print("   d mu    : hip", H.worst_ratio(res[1].grad.cpu(), i64["mu"].grad, 1e-4), "oracle32", H.worst_ratio(i32["mu"].grad, i64["mu"].grad, 1e-4))
