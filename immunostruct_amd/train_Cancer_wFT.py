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
from .data import (DeviceResidentDataset, ExtendedDataset, PackedDataset, SplitDataset, SyntheticImmunoDataset,
                   SyntheticPairedDataset, collate, collate_amino_acid, packed_from_reference_inputs,
                   paired_from_reference_inputs, require_paths)
from .models.mapping import model_map
from .procedures import (inference_comparative, inference_comparative_SSL, train_model, train_model_comparative,
                         train_model_comparative_device, train_model_comparative_SSL, train_model_comparative_SSL_device,
                         train_model_device, train_model_SSL, train_model_SSL_device)
from .utils import LinearWarmupCosineAnnealingLR, Losses, seed_everything, update_paths


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
    p.add_argument("--feature-size", default=23, type=int)
    p.add_argument("--coord-size", default=3, type=int)
    p.add_argument("--min-finetuning-batches", default=64, type=int)
    p.add_argument("--model-save-dir", default="$ROOT/checkpoints/comparative_PropIEDB_PropCancer_ImmunoCancer/", type=str)
    p.add_argument("--figure-save-dir", default="$ROOT/figures/comparative_PropIEDB_PropCancer_ImmunoCancer/", type=str)
    p.add_argument("--graph-dir-IEDB", default="$ROOT/data/graph_pyg_IEDB/", type=str)
    p.add_argument("--graph-dir-cancer", default="$ROOT/data/graph_pyg_Cancer/", type=str)
    p.add_argument("--graph-dir-wildtype", default="$ROOT/data/graph_pyg_Cancer_WT/", type=str)
    p.add_argument("--graph-dir-clinical", default="$ROOT/data/graph_pyg_Clinical/", type=str)
    p.add_argument("--property-path-IEDB", default="$ROOT/data/complete_score_Mprops_1_2_smoothed_sasa_v2.txt", type=str)
    p.add_argument("--property-path-cancer", default="$ROOT/data/cedar_data_final_with_mprop1_mprop2_v2.txt", type=str)
    p.add_argument("--property-path-wildtype", default="$ROOT/data/cedar_data_final_WILD_TYPE_with_mprop1_mprop2_v2.txt", type=str)
    p.add_argument("--seq-path-clinical", default="$ROOT/data/hadrup_cancer_df_29K.txt", type=str)
    p.add_argument("--hla-path", default="$ROOT/data/HLA_27_seqs_csv.csv", type=str)
    p.add_argument("--wandb-username", default=None, type=str)
    p.add_argument("--seed", default=1, type=int)
    p.add_argument("--coeff-contrastive", default=0, type=float)
    p.add_argument("--sequence-pad-count", default=0, type=int)
    p.add_argument("--structure-pad-count", default=0, type=int)
    p.add_argument("--self-supervision", action="store_true")   # for the *_SSL models: masked-residue prediction
    p.add_argument("--synthetic", default=0, type=int)
    p.add_argument("--device-dataset", action="store_true",
                   help="keep the datasets in HBM and assemble batches on the GPU (data.DeviceResidentDataset)")
    return p.parse_args(argv)


_Extended = ExtendedDataset      # modulo oversampling (reference data/util_dataloader.py:88-102)


def main(argv=None):
    config = parse_args(argv)
    update_paths(config)
    from_reference = config.synthetic <= 0
    if from_reference:
        # the reference's own inputs (train_Cancer_wFT.py:77-91): three graph directories, three property tables, the HLA table
        require_paths(graph_dir_IEDB=config.graph_dir_IEDB, graph_dir_cancer=config.graph_dir_cancer,
                      graph_dir_wildtype=config.graph_dir_wildtype, property_path_IEDB=config.property_path_IEDB,
                      property_path_cancer=config.property_path_cancer, property_path_wildtype=config.property_path_wildtype,
                      hla_path=config.hla_path)
    if config.wandb_username is not None:
        try:
            import wandb
            wandb.init(project="ImmunoPred-Cancer-Paper-2", entity=config.wandb_username, name=config.model, config=vars(config))
        except ImportError:
            print("wandb is not installed: --wandb-username ignored")
    # (--figure-save-dir, --graph-dir-clinical, --seq-path-clinical feed the reference's Kaplan-Meier clinical validation,
    #  train_Cancer_wFT.py:192-229: outside the hot path -- the flags are accepted so that an existing command line runs)
    if not torch.cuda.is_available():
        raise SystemExit("immunostruct_amd needs a ROCm GPU (no CPU fallback)")
    tag = (f"{config.model}-wtds_{config.use_wt_for_downstream}-lr_pt_{config.learning_rate_pretrain}"
           f"-lr_ft_{config.learning_rate_finetune}-cc_{config.coeff_contrastive}-ep_{config.num_epochs}"
           f"-bs_{config.batch_size}-seed_{config.seed}")
    config.model_save_path_pretrain = os.path.join(config.model_save_dir, tag + "_pretrain.pt")
    config.model_save_path_finetune = os.path.join(config.model_save_dir, tag + "_finetune.pt")
    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N -m immunostruct_amd.train_Cancer_wFT ...`:
    # the device-resident loops shard every epoch over the ranks (procedures.train._device_fit); single process otherwise
    from .distributed import init_from_env
    rank, local_rank, world = init_from_env()
    if world > 1 and not config.device_dataset:
        raise SystemExit("data-parallel runs need --device-dataset (the host-loader loop is single-process, as the reference)")
    if world > 1:
        local_rank = int(os.environ.get("IMMUNOSTRUCT_FORCE_DEVICE", local_rank))      # debugging aid: ranks sharing a GPU (gloo)
        torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank) if world > 1 else torch.device("cuda")

    def sync():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()      # rank 0 has written the checkpoint every rank loads next
    seed_everything(config.seed)
    gen = torch.Generator().manual_seed(config.seed)
    # 283 x 21 (HLA + peptide) with --full-sequence, otherwise the peptide alone, 11 x 21 (train_Cancer_wFT.py:71-72)
    full = config.full_sequence
    input_dim = 283 * 21 if full else 11 * 21
    model = model_map[config.model](vae_input_dim=input_dim, device=device,
                                    use_wt_for_downstream=config.use_wt_for_downstream).to(device)
    ssl = config.self_supervision
    if config.device_dataset and not ssl and (config.sequence_pad_count or config.structure_pad_count):
        raise SystemExit("--sequence-pad-count / --structure-pad-count with --device-dataset need --self-supervision "
                         "(the plain on-device loop applies no train-time masking); drop --device-dataset or the pad counts")
    pads = dict(structure_pad_count=config.structure_pad_count, sequence_pad_count=config.sequence_pad_count)

    def mk(d, split, comparative=False):
        # the reference's split wrapper: train-time augmentation; with --self-supervision a fifth field (masked residue)
        wrapped = SplitDataset(d, split, comparative=comparative, return_amino_acid=ssl, full=full, **pads)
        return DataLoader(wrapped, batch_size=config.batch_size, collate_fn=collate_amino_acid if ssl else collate,
                          shuffle=split == "train", num_workers=config.num_workers)

    # stage 1: IEDB-style single-graph pretraining through the plain forward
    sizes = dict(feature_size=config.feature_size, coord_size=config.coord_size)
    if from_reference:
        ds1 = packed_from_reference_inputs(config.graph_dir_IEDB, config.property_path_IEDB, config.hla_path, **sizes)
        ds1.normalize()                  # foreignness -> [-1, 1] (data/immmunopred_dataloader.py:67-70)
        ds1.full_sequence = full
    else:
        ds1 = SyntheticImmunoDataset(config.synthetic, seed=config.seed, binary=False, full_sequence=full)
    tr, va, _ = random_split(ds1, [0.8, 0.1, 0.1], gen)
    losses = Losses(input_dim, ds1.class_weights, sequence=config.sequence_loss)
    opt = optim.AdamW(model.parameters(), lr=config.learning_rate_pretrain, weight_decay=1e-6)
    if config.device_dataset:
        dds1 = DeviceResidentDataset.from_packed(ds1, device) if isinstance(ds1, PackedDataset) else DeviceResidentDataset(ds1, device)
        if ssl:
            train_model_SSL_device(config, device, model, dds1, tr.indices, va.indices, opt, losses.regression_loss_SSL, seed=config.seed)
        else:
            train_model_device(config, device, model, dds1, tr.indices, va.indices, opt, losses.regression_loss, seed=config.seed)
    else:
        (train_model_SSL if ssl else train_model)(config, device, model, mk(tr, "train"), mk(va, "val"), opt,
                                                  losses.regression_loss_SSL if ssl else losses.regression_loss)
    sync()
    model.load_trained(config.model_save_path_pretrain, new_head=True)

    # stage 2: comparative pretraining on (cancer, wild-type) pairs, continuous target
    if from_reference:
        pairs = paired_from_reference_inputs(config.graph_dir_cancer, config.graph_dir_wildtype, config.property_path_cancer,
                                             config.property_path_wildtype, config.hla_path, binary=False, **sizes)
        lo, hi = pairs.c.packed.normalize()
        pairs.w.packed.y_reg = 2 * (pairs.w.packed.y_reg - (hi + lo) / 2) / (hi - lo)      # one scale for both members
        pairs.c.packed.full_sequence = pairs.w.packed.full_sequence = full
        ds2 = pairs
    else:
        ds2 = SyntheticPairedDataset(config.synthetic, seed=config.seed + 1, binary=False, full_sequence=full)
    tr2, va2, te2 = random_split(ds2, [0.8, 0.1, 0.1], gen)
    opt = optim.AdamW(model.parameters(), lr=config.learning_rate_pretrain, weight_decay=1e-6)
    def fit_pairs(ds, tr_, va_, opt_, loss_fn, sched_=None, stage="pretrain"):
        if config.device_dataset:
            # graph ids of a random_split subset, or of its modulo-oversampled extension (reference ExtendedDataset)
            idx = lambda sub: ([sub.dataset.indices[i % len(sub.dataset)] for i in range(len(sub))]
                               if isinstance(sub, _Extended) else sub.indices)
            run_dev = train_model_comparative_SSL_device if ssl else train_model_comparative_device
            return run_dev(config, device, model, DeviceResidentDataset(ds.c, device), DeviceResidentDataset(ds.w, device),
                           idx(tr_), idx(va_), opt_, loss_fn, sched_, stage=stage, seed=config.seed)
        run = train_model_comparative_SSL if ssl else train_model_comparative
        return run(config, device, model, mk(tr_, "train", True), mk(va_, "val", True), opt_, loss_fn, sched_, stage=stage)

    fit_pairs(ds2, tr2, va2, opt, losses.regression_loss_SSL if ssl else losses.regression_loss)
    sync()
    model.load_trained(config.model_save_path_pretrain, new_head=True)

    # stage 3: comparative finetuning, BCE (+ coeff * paired contrastive loss)
    if from_reference:
        pairs.c.packed.binary = pairs.w.packed.binary = True      # same pairs, binary target (train_Cancer_wFT.py:160-162)
        ds3 = pairs
    else:
        ds3 = SyntheticPairedDataset(config.synthetic, seed=config.seed + 1, binary=True, full_sequence=full)
    tr3, va3, te3 = random_split(ds3, [0.8, 0.1, 0.1], gen)
    want = config.min_finetuning_batches * config.batch_size
    tr3 = _Extended(tr3, want) if len(tr3) < want else tr3
    opt = optim.AdamW(model.parameters(), lr=config.learning_rate_finetune, weight_decay=1e-6)
    sched = LinearWarmupCosineAnnealingLR(opt, warmup_epochs=config.num_epochs // 4,
                                          warmup_start_lr=config.learning_rate_finetune / 100, max_epochs=config.num_epochs)
    fit_pairs(ds3, tr3, va3, opt, losses.BCE_loss_SSL if ssl else losses.BCE_loss, sched, stage="finetune")
    sync()
    model.load_trained(config.model_save_path_finetune, new_head=False)
    # metrics as the reference reports them (train_Cancer_wFT.py:178-190): threshold from the train pairs, applied to the test pairs
    infer = inference_comparative_SSL if ssl else inference_comparative
    train_stats = infer(config, model, mk(tr3, "train", True), device)
    test_stats = infer(config, model, mk(te3, "test", True), device, optimal_threshold=train_stats["optimal_threshold"])
    return train_stats, test_stats


if __name__ == "__main__":
    main()
