"""Entry point 1 -- IEDB pretrain (regress normalised foreignness) then finetune (BCE immunogenicity).

Same stages, flags and checkpoint naming as the reference's ``train_IEDB_wFT.py:15-163``, running the
HIP-backed ``immunostruct_amd.models``.  The reference's on-disk inputs (pickled PyG graphs + tables) were
never shipped: with the reference's path flags (``--graph-dir-IEDB``, ``--property-path-IEDB``, ``--hla-path``) the graph
directory is read and joined with the tables (``data.reference_inputs``); ``--packed FILE`` trains on a packed dataset
written once by ``data.convert_pyg_directory``; ``--synthetic N`` draws N synthetic peptide-MHC graphs of
the same shape instead.

    python -m immunostruct_amd.train_IEDB_wFT --model HybridModelv2 --full-sequence --sequence-loss --synthetic 512
"""
from __future__ import annotations

import argparse
import os

import torch
from torch.utils.data import DataLoader, random_split

from . import optim
from .data import (DeviceResidentDataset, PackedDataset, SplitDataset, SyntheticImmunoDataset, collate, collate_amino_acid,
                   packed_from_reference_inputs, require_paths)
from .models.mapping import model_map
from .procedures import (inference, inference_SSL, train_model, train_model_device, train_model_SSL,
                         train_model_SSL_device)
from .utils import Losses, seed_everything, update_paths


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Entry point.")
    p.add_argument("--model", default="StructureModel", type=str)
    p.add_argument("--learning-rate-pretrain", default=1e-3, type=float)
    p.add_argument("--learning-rate-finetune", default=1e-4, type=float)
    p.add_argument("--num-epochs", default=40, type=int)
    p.add_argument("--batch-size", default=150, type=int)
    p.add_argument("--num-workers", default=0, type=int)
    p.add_argument("--full-sequence", action="store_true")
    p.add_argument("--sequence-loss", action="store_true")
    p.add_argument("--feature-size", default=23, type=int)
    p.add_argument("--coord-size", default=3, type=int)
    p.add_argument("--model-save-dir", default="$ROOT/results/PropIEDB_ImmunoIEDB/", type=str)
    p.add_argument("--graph-dir-IEDB", default="$ROOT/data/graph_pyg_IEDB/", type=str)
    p.add_argument("--property-path-IEDB", default="$ROOT/data/complete_score_Mprops_1_2_smoothed_sasa_v2.txt", type=str)
    p.add_argument("--hla-path", default="$ROOT/data/HLA_27_seqs_csv.csv", type=str)
    p.add_argument("--wandb-username", default=None, type=str)
    p.add_argument("--seed", default=1, type=int)
    p.add_argument("--sequence-pad-count", default=0, type=int)
    p.add_argument("--structure-pad-count", default=0, type=int)
    p.add_argument("--self-supervision", action="store_true")   # for the *_SSL models: masked-residue prediction
    p.add_argument("--synthetic", default=0, type=int, help="number of synthetic graphs to train on")
    p.add_argument("--packed", default=None, type=str,
                   help="packed dataset (.npz written by data.convert_pyg_directory / PackedDataset.save) to train on")
    p.add_argument("--device-dataset", action="store_true",
                   help="keep the dataset in HBM, assemble batches on the GPU and replay the captured train step "
                        "(data.DeviceResidentDataset + procedures.train_model_device) instead of DataLoader + collate")
    return p.parse_args(argv)


def main(argv=None):
    config = parse_args(argv)
    update_paths(config)
    from_reference = config.synthetic <= 0 and not config.packed
    if from_reference:
        # the reference's own inputs (train_IEDB_wFT.py:55-57): graph directory + property table + HLA table
        require_paths(graph_dir_IEDB=config.graph_dir_IEDB, property_path_IEDB=config.property_path_IEDB, hla_path=config.hla_path)
    if config.wandb_username is not None:
        try:
            import wandb
            wandb.init(project="ImmunoStruct", entity=config.wandb_username, name=config.model, config=vars(config))
        except ImportError:
            print("wandb is not installed: --wandb-username ignored")
    tag = (f"{config.model}-lr_pt_{config.learning_rate_pretrain}-lr_ft_{config.learning_rate_finetune}"
           f"-ep_{config.num_epochs}-bs_{config.batch_size}-fseq_{config.full_sequence}-seql_{config.sequence_loss}"
           f"-fs_{config.feature_size}-cs_{config.coord_size}-seed_{config.seed}")
    config.model_save_path_pretrain = os.path.join(config.model_save_dir, tag + "_pretrain.pt")
    config.model_save_path_finetune = os.path.join(config.model_save_dir, tag + "_finetune.pt")
    if not torch.cuda.is_available():
        raise SystemExit("immunostruct_amd needs a ROCm GPU (no CPU fallback)")
    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N -m immunostruct_amd.train_IEDB_wFT ...`:
    # the device-resident loops shard every epoch over the ranks (procedures.train_model_device); single process otherwise
    from .distributed import init_from_env
    rank, local_rank, world = init_from_env()
    if world > 1 and not config.device_dataset:
        raise SystemExit("data-parallel runs need --device-dataset (the host-loader loop is single-process, as the reference)")
    if config.device_dataset and not config.self_supervision and (config.sequence_pad_count or config.structure_pad_count):
        # the reference's SplitDataset masks the sequence / pads the structure at train time also WITHOUT self-supervision
        # (data/util_dataloader.py:60-86); the on-device loop applies those transforms in its self-supervised form only
        raise SystemExit("--sequence-pad-count / --structure-pad-count with --device-dataset need --self-supervision "
                         "(the plain on-device loop applies no train-time masking); drop --device-dataset or the pad counts")
    if world > 1:
        local_rank = int(os.environ.get("IMMUNOSTRUCT_FORCE_DEVICE", local_rank))      # debugging aid: ranks sharing a GPU (gloo)
        torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank) if world > 1 else torch.device("cuda")
    seed_everything(config.seed)
    # sequence length (283: HLA + peptide with --full-sequence, 11: the peptide alone, the reference's default) x 21 symbols
    # (train_IEDB_wFT.py:59-60)
    full = config.full_sequence
    input_dim = 283 * 21 if full else 11 * 21
    model = model_map[config.model](vae_input_dim=input_dim, device=device).to(device)

    ssl = config.self_supervision
    pads = dict(structure_pad_count=config.structure_pad_count, sequence_pad_count=config.sequence_pad_count)

    reference_packed = {}

    def loaders(binary):
        if from_reference:
            if not reference_packed:         # read the graph files and join the tables once for both stages
                reference_packed["ds"] = packed_from_reference_inputs(config.graph_dir_IEDB, config.property_path_IEDB, config.hla_path,
                                                                      feature_size=config.feature_size, coord_size=config.coord_size)
                reference_packed["ds"].normalize()
            ds = reference_packed["ds"]
            ds.binary = binary
        elif config.packed:
            ds = PackedDataset.load(config.packed, binary=binary)
            ds.normalize()                   # foreignness -> [-1, 1] (reference data/immmunopred_dataloader.py:67-70)
        else:
            ds = SyntheticImmunoDataset(config.synthetic, seed=config.seed, binary=binary, full_sequence=full)
        if isinstance(ds, PackedDataset):
            ds.full_sequence = full
        tr, va, te = random_split(ds, [0.8, 0.1, 0.1], torch.Generator().manual_seed(config.seed))
        # the reference's split wrapper (train-time augmentation; with --self-supervision a fifth field, the masked residue)
        wrap = lambda d, split: SplitDataset(d, split, comparative=False, return_amino_acid=ssl, full=full, **pads)
        mk = lambda d, split: DataLoader(wrap(d, split), batch_size=config.batch_size, shuffle=split == "train",
                                         collate_fn=collate_amino_acid if ssl else collate, num_workers=config.num_workers)
        return ds, (tr, va, te), mk(tr, "train"), mk(va, "val"), mk(te, "test")

    def fit(splits, tr, va, opt, loss_fn, stage):
        if config.device_dataset:
            # same split, same batch size; batches are assembled (and, with --self-supervision, augmented) on the device
            full = splits[0].dataset
            dds = DeviceResidentDataset.from_packed(full, device) if isinstance(full, PackedDataset) else DeviceResidentDataset(full, device)
            run = train_model_SSL_device if ssl else train_model_device
            return run(config, device, model, dds, splits[0].indices, splits[1].indices, opt, loss_fn, stage=stage, seed=config.seed)
        return (train_model_SSL if ssl else train_model)(config, device, model, tr, va, opt, loss_fn, stage=stage)

    ds, splits, tr, va, _ = loaders(binary=False)
    losses = Losses(input_dim, ds.class_weights, sequence=config.sequence_loss)
    opt = optim.Adam(model.parameters(), lr=config.learning_rate_pretrain)
    fit(splits, tr, va, opt, losses.regression_loss_SSL if ssl else losses.regression_loss, "pretrain")
    print("DONE PRE-TRAINING")
    if torch.distributed.is_initialized():
        torch.distributed.barrier()      # rank 0 has written the checkpoint every rank loads next
    model.load_trained(config.model_save_path_pretrain, new_head=True)
    ds, splits, tr, va, te = loaders(binary=True)
    opt = optim.Adam(model.parameters(), lr=config.learning_rate_finetune, weight_decay=1e-6)
    fit(splits, tr, va, opt, losses.BCE_loss_SSL if ssl else losses.BCE_loss, "finetune")
    print("DONE FINE TUNING")
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    model.load_trained(config.model_save_path_finetune, new_head=False)
    # metrics as the reference reports them (train_IEDB_wFT.py:117-129): the Youden threshold of the TRAIN set is applied to the test set
    infer = inference_SSL if ssl else inference
    train_stats = infer(config, model, tr, device)
    test_stats = infer(config, model, te, device, optimal_threshold=train_stats["optimal_threshold"])
    return train_stats, test_stats


if __name__ == "__main__":
    main()
