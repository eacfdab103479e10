"""End-to-end epoch throughput (batching included): DataLoader + collate + H2D + eager step  vs  device-resident
dataset + on-GPU batcher + captured step.  python tools/e2e_throughput.py [num_graphs] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from torch.utils.data import DataLoader
from immunostruct_amd import optim
from immunostruct_amd.data import DeviceResidentDataset, SyntheticImmunoDataset, collate
from immunostruct_amd.models import model_map
from immunostruct_amd.procedures import train_model, train_model_device
from immunostruct_amd.utils import Losses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
ds = SyntheticImmunoDataset(n + bsz, seed=1)
losses = Losses(283 * 21, ds.class_weights, sequence=True)
cfg = SimpleNamespace(batch_size=bsz, num_epochs=2, model_save_path_pretrain="/tmp/e2e_m.pt", model_save_path_finetune="/tmp/e2e_f.pt")
tr_idx, va_idx = list(range(n)), list(range(n, n + bsz))
for mode in ("dataloader", "device"):
    torch.manual_seed(0)
    model = model_map["HybridModelv2"](vae_input_dim=283 * 21, device=dev).to(dev)
    opt = optim.Adam(model.parameters(), lr=1e-3)
    if mode == "dataloader":
        tr = DataLoader(torch.utils.data.Subset(ds, tr_idx), batch_size=bsz, collate_fn=collate, shuffle=True)
        va = DataLoader(torch.utils.data.Subset(ds, va_idx), batch_size=bsz, collate_fn=collate)
        run = lambda: train_model(cfg, dev, model, tr, va, opt, losses.regression_loss)
    else:
        dds = DeviceResidentDataset(ds, dev)
        run = lambda: train_model_device(cfg, dev, model, dds, tr_idx, va_idx, opt, losses.regression_loss)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"[e2e] {mode:10s}: {cfg.num_epochs} epochs x {n} graphs (+ validation) in {dt:.2f} s = {cfg.num_epochs * n / dt:.0f} train graphs/s end to end")
