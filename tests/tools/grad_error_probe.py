"""Where does the full-size distance of the coordinate-path gradients come from?  (test infrastructure: uses the oracle)

The EGNN stack alone at BASELINE config 2's size (B = 128 graphs x 190 nodes, ~146 k edges with deg_extra = 5 as in
test_full_train_step_gradients_vs_oracle), loss = sum(h_L * G): forward outputs after every depth and every parameter gradient,
HIP (fp32) and the oracle in fp32, each against the oracle in fp64 -- as RMS-relative errors (a systematic factor shows there,
an outlier does not) and as the ratio of the two.

    python tests/tools/grad_error_probe.py [--batch 128] [--deg-extra 5] [--layers 6]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from immunostruct_amd import synthetic  # noqa: E402
from immunostruct_amd.nn import EGNNConv, egnn_stack_forward  # noqa: E402
from oracle import graph_ref  # noqa: E402
from tests import helpers as H  # noqa: E402


def rms_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--deg-extra", type=int, default=5)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--gx", type=float, default=0.0, help="weight of a gradient at the final coordinates (the models: none)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = args.layers
    raw = synthetic.make_batch(args.batch, seed=33, deg_extra=args.deg_extra, n_pad=190, n_real_choices=(188, 189, 190))
    n = raw.num_nodes
    sd = H.det_sd(H.egnn_shapes([20] + [64] * (L - 1), 1), seed=14)
    rng = np.random.RandomState(5)
    gh = (rng.normal(size=(n, 64)) / n).astype(np.float32)
    gx = (rng.normal(size=(n, 3)) / n).astype(np.float32) * args.gx
    h0, x0 = raw.x[:, :20].copy(), raw.x[:, 20:].copy()
    src, dst = torch.from_numpy(raw.src), torch.from_numpy(raw.dst)
    print(f"nodes {n} edges {raw.num_edges} layers {L}")
    ref = {}
    for dt in (torch.float32, torch.float64):
        sdd = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
        hh, xx = torch.from_numpy(h0).to(dt), torch.from_numpy(x0).to(dt)
        a = torch.from_numpy(raw.edge_attr).to(dt)
        fw = []
        for i in range(L):
            hh, xx = graph_ref.egnn_conv(sdd, f"GCN_layers.{i}.", src, dst, n, hh, xx, a)
            fw.append((hh.detach(), xx.detach()))
        ((hh * torch.from_numpy(gh).to(dt)).sum() + (xx * torch.from_numpy(gx).to(dt)).sum()).backward()
        ref[dt] = dict(fw=fw, grads={k: v.grad for k, v in sdd.items() if v.grad is not None})
    g = H.product_graph(raw, dev)
    layers = [EGNNConv(20 if i == 0 else 64, 64, 64, 1).to(dev) for i in range(L)]
    for i, layer in enumerate(layers):
        layer.load_state_dict({k.split(".", 2)[2]: v for k, v in sd.items() if k.startswith(f"GCN_layers.{i}.")})
    hd, xd = torch.from_numpy(h0).to(dev), torch.from_numpy(x0).to(dev)
    print("forward, RMS-relative error against fp64:   h: HIP, fp32 oracle, ratio | x: HIP, fp32 oracle, ratio")
    for depth in range(1, L + 1):
        with torch.no_grad():
            hh, xx = egnn_stack_forward(layers[:depth], g, hd, xd, g.edata["edge_attr"])
        f64, f32 = ref[torch.float64]["fw"][depth - 1], ref[torch.float32]["fw"][depth - 1]
        eh, oh, ex, ox = rms_rel(hh, f64[0]), rms_rel(f32[0], f64[0]), rms_rel(xx, f64[1]), rms_rel(f32[1], f64[1])
        print(f"  depth {depth}: h {eh:.2e} {oh:.2e} {eh / oh:5.2f} | x {ex:.2e} {ox:.2e} {ex / ox:5.2f}")
    hh, xx = egnn_stack_forward(layers, g, hd, xd, g.edata["edge_attr"])
    ((hh * torch.from_numpy(gh).to(dev)).sum() + (xx * torch.from_numpy(gx).to(dev)).sum()).backward()
    print("parameter gradients: RMS-relative error against fp64: HIP, fp32 oracle, ratio | worst element / (1e-4 max|g|): HIP, fp32 oracle")
    for i, layer in enumerate(layers):
        for k, p in layer.named_parameters():
            key = f"GCN_layers.{i}.{k}"
            if key not in ref[torch.float64]["grads"] or p.grad is None:
                continue
            g64, g32 = ref[torch.float64]["grads"][key], ref[torch.float32]["grads"][key]
            e, o = rms_rel(p.grad, g64), rms_rel(g32, g64)
            print(f"  {key:38s} {e:.2e} {o:.2e} {e / max(o, 1e-300):6.2f} | {H.worst_ratio(p.grad.cpu(), g64, 1e-4):6.2f} {H.worst_ratio(g32, g64, 1e-4):6.2f}")


if __name__ == "__main__":
    main()
