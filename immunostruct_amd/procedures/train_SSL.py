"""Self-supervised training loops with the reference's signatures (``procedures/train_SSL.py:10-180``).

Loaders yield a fifth field, the type of the masked residue (``data.collate_amino_acid``); the ``*_SSL`` models return a
fifth output, the residue logits of ``node_predictor_head``; the loss gets both (``Losses.regression_loss_SSL`` /
``BCE_loss_SSL``: + cross-entropy).  Validation passes empty tensors for the pair, which drops the term
(reference ``:45-46``, ``:151-152``).  ``train_model_SSL_device`` is the same loop on a
:class:`~immunostruct_amd.data.DeviceResidentDataset`: batches are assembled AND augmented on the GPU
(``data.augment_batch_on_device``) and every full batch is one replay of the captured HIP graph.
"""
from __future__ import annotations

import torch

from ..utils import PairedContrastiveLoss
from .train import _add_contrastive, _device_fit, _fit, _reads_recon_through_losses, _to

__all__ = ["train_model_SSL", "train_model_comparative_SSL", "train_model_SSL_device", "train_model_comparative_SSL_device"]


def _ssl_targets(model, pred_amino_acid, amino_acid, device):
    if model.training:
        return pred_amino_acid, amino_acid.to(device)
    return torch.tensor([]), torch.tensor([])


def _single_loss_ssl(model, loss_function, batch, device):
    graph, seq, target, prop = _to(device, tuple(batch[:4]))
    recon, mu, logvar, final, pred = model(graph, seq, prop)
    pred, amino = _ssl_targets(model, pred, batch[4], device)
    return loss_function(recon, seq, mu, logvar, final, target, pred, amino)


def _paired_loss_ssl(model, loss_function, batch, device, contrastive, coeff):
    graphs, seqs, target, props = _to(device, tuple(batch[:4]))
    emb, recon, mu, logvar, final, pred = model.forward_comparative(graphs, seqs, props)
    pred, amino = _ssl_targets(model, pred, batch[4], device)
    loss = 0.5 * (loss_function(recon[0], seqs[0], mu[0], logvar[0], final, target, pred, amino)
                  + loss_function(recon[1], seqs[1], mu[1], logvar[1], final, target, pred, amino))
    return _add_contrastive(loss, contrastive, emb, target, coeff)


def train_model_SSL(config, device, model, train_loader, val_loader, optimizer, loss_function, scheduler=None, stage="pretrain"):
    return _fit(config, model, train_loader, val_loader, optimizer, scheduler, stage,
                lambda batch: _single_loss_ssl(model, loss_function, batch, device))


def train_model_comparative_SSL(config, device, model, train_loader, val_loader, optimizer, loss_function, scheduler=None,
                                stage="pretrain"):
    coeff = float(getattr(config, "coeff_contrastive", 0) or 0)
    contrastive = PairedContrastiveLoss(device=device, embedding_dim=104) if coeff > 0 else None
    return _fit(config, model, train_loader, val_loader, optimizer, scheduler, stage,
                lambda batch: _paired_loss_ssl(model, loss_function, batch, device, contrastive, coeff))


def train_model_SSL_device(config, device, model, dataset, train_index, val_index, optimizer, loss_function, scheduler=None,
                           stage="pretrain", seed=0):
    """``train_model_SSL`` on a device-resident dataset: gather -> augment (rotation, masked residue, optional
    ``config.structure_pad_count`` / ``config.sequence_pad_count``) -> captured step, all on the GPU; data parallel under an
    initialised process group (``procedures.train._device_fit``: every rank augments its own shard with its own stream)."""
    from ..data import augment_batch_on_device, mask_sequence_on_device
    device = dataset.device
    pad_structure = int(getattr(config, "structure_pad_count", 0) or 0)
    pad_sequence = int(getattr(config, "sequence_pad_count", 0) or 0) if getattr(config, "full_sequence", True) else 0     # the peptide alone is never masked (data/util_dataloader.py:52-66)
    dgen = torch.Generator(device=device).manual_seed(int(seed) + 1 + 104729 * _rank())
    amino = {}

    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final, pred = m(g, seq, prop)
        if m.training:
            return loss_function(recon, seq, mu, logvar, final, y, pred, amino[int(y.numel())])
        return loss_function(recon, seq, mu, logvar, final, y, torch.tensor([]), torch.tensor([]))
    forward_loss.fused_loss = _reads_recon_through_losses(loss_function)

    def assemble(idx, buf, train):
        g, seq, prop, y = dataset.gather_into(idx, *buf)
        if train:
            b = int(idx.numel())
            if b not in amino:
                amino[b] = torch.zeros(b, dtype=torch.int64, device=device)
            amino[b].copy_(augment_batch_on_device(g.ndata["x"], b, dgen, structure_pad_count=pad_structure))
            mask_sequence_on_device(seq, pad_sequence, generator=dgen)
        return g, seq, prop, y

    return _device_fit(config, model, optimizer, scheduler, stage, seed, device, train_index, val_index, dataset.new_batch,
                       assemble, forward_loss, lambda b: b * dataset.max_edges)


def train_model_comparative_SSL_device(config, device, model, dataset_cancer, dataset_wt, train_index, val_index, optimizer,
                                       loss_function, scheduler=None, stage="pretrain", seed=0):
    """``train_model_comparative_SSL`` (``procedures/train_SSL.py:71-180``) on device-resident (cancer, wild-type) datasets:
    a batch of B pairs is gathered as ONE merged batch of 2B graphs, augmented on the GPU (independent rotations, one masked
    residue of the SAME type in both members of a pair -- ``data.augment_pair_on_device`` -- optional structure / sequence
    padding) and run as one replay of the captured step; the loss is 0.5 * (L(cancer) + L(wild-type)) with a shared
    prediction term plus the residue cross-entropy (counted once, as the two halves add up to) plus
    ``coeff_contrastive`` x the paired contrastive loss.  Data parallel under an initialised process group."""
    from ..data import DeviceResidentDataset, augment_pair_on_device, mask_sequence_on_device
    device = dataset_cancer.device
    coeff = float(getattr(config, "coeff_contrastive", 0) or 0)
    contrastive = PairedContrastiveLoss(device=device, embedding_dim=104) if coeff > 0 else None
    if contrastive is not None:
        contrastive.capturable = True
    pad_structure = int(getattr(config, "structure_pad_count", 0) or 0)
    pad_sequence = int(getattr(config, "sequence_pad_count", 0) or 0) if getattr(config, "full_sequence", True) else 0     # the peptide alone is never masked (data/util_dataloader.py:52-66)
    dgen = torch.Generator(device=device).manual_seed(int(seed) + 1 + 104729 * _rank())
    both = DeviceResidentDataset.concat(dataset_cancer, dataset_wt)
    shift = len(dataset_cancer)
    amino = {}

    def forward_loss(m, g2, seq2, prop2, y2):
        b = int(y2.numel()) // 2
        target = y2[:b]
        emb, recon, mu, logvar, final, pred = m.forward_comparative(g2, seq2, prop2)
        if m.training:
            pa, aa = pred, amino[b]
        else:
            pa, aa = torch.tensor([]), torch.tensor([])
        merged = [getattr(t, "merged", None) for t in (recon, mu, logvar)]
        if all(t is not None for t in merged):
            loss = loss_function(merged[0], seq2, merged[1], merged[2], final, target, pa, aa)
        else:
            loss = 0.5 * (loss_function(recon[0], seq2[:b], mu[0], logvar[0], final, target, pa, aa)
                          + loss_function(recon[1], seq2[b:], mu[1], logvar[1], final, target, pa, aa))
        return _add_contrastive(loss, contrastive, emb, target, coeff)
    forward_loss.fused_loss = _reads_recon_through_losses(loss_function)

    def assemble(idx, buf, train):
        g2, seq2, prop2, y2 = both.gather_into(torch.cat([idx, idx + shift]), *buf)
        if train:
            b = int(idx.numel())
            if b not in amino:
                amino[b] = torch.zeros(b, dtype=torch.int64, device=device)
            amino[b].copy_(augment_pair_on_device(g2.ndata["x"], b, dgen, structure_pad_count=pad_structure))
            mask_sequence_on_device(seq2, pad_sequence, generator=dgen, pairs=True)
        return g2, seq2, prop2, y2

    return _device_fit(config, model, optimizer, scheduler, stage, seed, device, train_index, val_index,
                       lambda b: both.new_batch(2 * b), assemble, forward_loss, lambda b: 2 * b * both.max_edges)


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_initialized() else 0
