"""Three captured train steps of HybridModelv2 (and one 8-head StructureModel step) under whatever ``IMMUNOSTRUCT_*`` switches the
environment carries; prints ONE JSON line with the losses, a digest of every parameter after the steps, and what the switches left
behind (stamps).  ``tests/test_gpu_switches.py`` runs it once per value of every switch the product path reads and
compares with the default run: a switch selects another kernel / schedule for the SAME arithmetic.

    IMMUNOSTRUCT_SAVE_Z3=0 python tools/switch_probe.py
"""
import json
import os
import sys
import unittest.mock as mock

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from immunostruct_amd import distributed as D, functional as HF, optim  # noqa: E402
from immunostruct_amd.data import DeviceResidentDataset, SyntheticImmunoDataset  # noqa: E402
from immunostruct_amd.engine import CapturedTrainStep  # noqa: E402
from immunostruct_amd.models import model_map  # noqa: E402
from immunostruct_amd.utils import Losses  # noqa: E402

VAE_IN, LR, BSZ, STEPS = 283 * 21, 1e-3, 24, 3


def run(name, dev, reducer_kw):
    ds = SyntheticImmunoDataset(BSZ * STEPS, seed=11)
    dds = DeviceResidentDataset(ds, dev)
    losses = Losses(VAE_IN, ds.class_weights, sequence=name != "StructureModel")      # (graph only: no reconstruction terms)
    torch.manual_seed(5)
    model = model_map[name](vae_input_dim=VAE_IN, device=dev).to(dev)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()

    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final = m(g, seq, prop)
        return losses.regression_loss(recon, seq, mu, logvar, final, y)
    forward_loss.fused_loss = True
    opt = optim.Adam(model.parameters(), lr=LR)
    red = D.FlatGradReducer(model.parameters(), world=1, **reducer_kw)
    buf = dds.new_batch(BSZ)
    ids = lambda s: torch.arange(s * BSZ, (s + 1) * BSZ, device=dev)
    dds.gather_into(ids(0), *buf)
    with mock.patch("torch.randn_like", torch.zeros_like):
        cap = CapturedTrainStep(model, opt, red, forward_loss, buf, edge_capacity=BSZ * dds.max_edges, warmup=1, preserve_state=True)
        out = []
        for s in range(STEPS):
            dds.gather_into(ids(s), cap.sgraph, cap.seq, cap.prop, cap.y)
            out.append(float(cap.replay()))
    torch.cuda.synchronize()
    digest = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in model.state_dict().items()}
    form = cap.one_graph or ("two-stage" if cap.two_stage else "serial")
    info = dict(losses=out, digest=digest, form=form, reserved=cap.reserved, tuned=cap.dp_times)
    cap.close()
    return info


def main():
    rank, local_rank, world = D.init_from_env()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    collective = D.collectives_forced()
    kw = dict(always_pack=True) if collective else {}
    res = dict(hybrid=run("HybridModelv2", dev, kw), structure=run("StructureModel", dev, kw))
    res["nccl_max_nchannels"] = os.environ.get("NCCL_MAX_NCHANNELS")
    res["stamps"] = [n for _, n in HF.Stamps.report()] if HF.Stamps.enabled else None
    print("SWITCH_PROBE " + json.dumps(res), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
