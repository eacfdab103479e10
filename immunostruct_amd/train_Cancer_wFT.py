"""Entry point 2 -- IEDB pretrain, cancer/wild-type comparative pretrain, comparative finetune with the
optional paired contrastive loss (reference ``train_Cancer_wFT.py:14-229``), on the HIP-backed models.

    python -m immunostruct_amd.train_Cancer_wFT --full-sequence --sequence-loss --use-wt-for-downstream \\
        --coeff-contrastive 0.01 --synthetic 512
"""
from __future__ import annotations

import argparse
import os

import torch
from torch.utils.data import DataLoader, random_split

from . import optim
from .data import DeviceResidentDataset, SyntheticImmunoDataset, SyntheticPairedDataset, collate
from .models.mapping import model_map
from .procedures import (binary_metrics, predict_proba, train_model, train_model_comparative, train_model_comparative_device,
                         train_model_device)
from .utils import Losses, seed_everything


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Entry point.")
    p.add_argument("--model", default="HybridModelv2_Comparative", type=str)
    p.add_argument("--use-wt-for-downstream", action="store_true")
    p.add_argument("--learning-rate-pretrain", default=1e-3, type=float)
    p.add_argument("--learning-rate-finetune", default=1e-4, type=float)
    p.add_argument("--num-epochs", default=40, type=int)
    p.add_argument("--batch-size", default=128, type=int)
    p.add_argument("--num-workers", default=0, type=int)
    p.add_argument("--full-sequence", action="store_true")
    p.add_argument("--sequence-loss", action="store_true")
    p.add_argument("--min-finetuning-batches", default=64, type=int)
    p.add_argument("--model-save-dir", default="./checkpoints/comparative/", type=str)
    p.add_argument("--seed", default=1, type=int)
    p.add_argument("--coeff-contrastive", default=0, type=float)
    p.add_argument("--synthetic", default=0, type=int)
    p.add_argument("--device-dataset", action="store_true",
                   help="keep the datasets in HBM and assemble batches on the GPU (data.DeviceResidentDataset)")
    return p.parse_args(argv)


class _Extended(torch.utils.data.Dataset):
    """modulo oversampling to at least ``desired_len`` items (reference ``data/util_dataloader.py:91-102``)."""

    def __init__(self, dataset, desired_len):
        self.dataset, self.desired_len = dataset, desired_len

    def __len__(self):
        return self.desired_len

    def __getitem__(self, idx):
        return self.dataset[idx % len(self.dataset)]


def main(argv=None):
    config = parse_args(argv)
    if not config.full_sequence or config.synthetic <= 0:
        raise SystemExit("pass --full-sequence --synthetic N (real-data loading is outside this package)")
    if not torch.cuda.is_available():
        raise SystemExit("immunostruct_amd needs a ROCm GPU (no CPU fallback)")
    tag = (f"{config.model}-wtds_{config.use_wt_for_downstream}-lr_pt_{config.learning_rate_pretrain}"
           f"-lr_ft_{config.learning_rate_finetune}-cc_{config.coeff_contrastive}-ep_{config.num_epochs}"
           f"-bs_{config.batch_size}-seed_{config.seed}")
    config.model_save_path_pretrain = os.path.join(config.model_save_dir, tag + "_pretrain.pt")
    config.model_save_path_finetune = os.path.join(config.model_save_dir, tag + "_finetune.pt")
    device = torch.device("cuda")
    seed_everything(config.seed)
    gen = torch.Generator().manual_seed(config.seed)
    input_dim = 283 * 21
    model = model_map[config.model](vae_input_dim=input_dim, device=device,
                                    use_wt_for_downstream=config.use_wt_for_downstream).to(device)
    mk = lambda d, sh: DataLoader(d, batch_size=config.batch_size, collate_fn=collate, shuffle=sh, num_workers=config.num_workers)

    # stage 1: IEDB-style single-graph pretraining through the plain forward
    ds1 = SyntheticImmunoDataset(config.synthetic, seed=config.seed, binary=False)
    tr, va, _ = random_split(ds1, [0.8, 0.1, 0.1], gen)
    losses = Losses(input_dim, ds1.class_weights, sequence=config.sequence_loss)
    opt = optim.AdamW(model.parameters(), lr=config.learning_rate_pretrain, weight_decay=1e-6)
    if config.device_dataset:
        train_model_device(config, device, model, DeviceResidentDataset(ds1, device), tr.indices, va.indices, opt,
                           losses.regression_loss, seed=config.seed)
    else:
        train_model(config, device, model, mk(tr, True), mk(va, False), opt, losses.regression_loss)
    model.load_trained(config.model_save_path_pretrain, new_head=True)

    # stage 2: comparative pretraining on (cancer, wild-type) pairs, continuous target
    ds2 = SyntheticPairedDataset(config.synthetic, seed=config.seed + 1, binary=False)
    tr2, va2, te2 = random_split(ds2, [0.8, 0.1, 0.1], gen)
    opt = optim.AdamW(model.parameters(), lr=config.learning_rate_pretrain, weight_decay=1e-6)
    def fit_pairs(ds, tr_, va_, opt_, loss_fn, sched_=None, stage="pretrain"):
        if config.device_dataset:
            # graph ids of a random_split subset, or of its modulo-oversampled extension (reference ExtendedDataset)
            idx = lambda sub: ([sub.dataset.indices[i % len(sub.dataset)] for i in range(len(sub))]
                               if isinstance(sub, _Extended) else sub.indices)
            return train_model_comparative_device(config, device, model, DeviceResidentDataset(ds.c, device),
                                                  DeviceResidentDataset(ds.w, device), idx(tr_), idx(va_), opt_, loss_fn,
                                                  sched_, stage=stage, seed=config.seed)
        return train_model_comparative(config, device, model, mk(tr_, True), mk(va_, False), opt_, loss_fn, sched_, stage=stage)

    fit_pairs(ds2, tr2, va2, opt, losses.regression_loss)
    model.load_trained(config.model_save_path_pretrain, new_head=True)

    # stage 3: comparative finetuning, BCE (+ coeff * paired contrastive loss)
    ds3 = SyntheticPairedDataset(config.synthetic, seed=config.seed + 1, binary=True)
    tr3, va3, te3 = random_split(ds3, [0.8, 0.1, 0.1], gen)
    want = config.min_finetuning_batches * config.batch_size
    tr3 = _Extended(tr3, want) if len(tr3) < want else tr3
    opt = optim.AdamW(model.parameters(), lr=config.learning_rate_finetune, weight_decay=1e-6)
    sched = torch.optim.lr_scheduler.SequentialLR(opt, [
        torch.optim.lr_scheduler.LinearLR(opt, 0.01, 1.0, total_iters=max(config.num_epochs // 4, 1)),
        torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max(config.num_epochs - config.num_epochs // 4, 1))],
        milestones=[max(config.num_epochs // 4, 1)])
    fit_pairs(ds3, tr3, va3, opt, losses.BCE_loss, sched, stage="finetune")
    model.load_trained(config.model_save_path_finetune, new_head=False)
    prob, y = predict_proba(model, mk(te3, False), device, comparative=True)
    print("test metrics:", binary_metrics(y, prob))


if __name__ == "__main__":
    main()
