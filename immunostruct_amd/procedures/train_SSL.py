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
from .train import _fit, _save_best, _to, wandb

__all__ = ["train_model_SSL", "train_model_comparative_SSL", "train_model_SSL_device"]


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
    if coeff > 0:
        loss = loss + coeff * contrastive(emb[0], emb[1], target)
    return loss


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
    ``config.structure_pad_count`` / ``config.sequence_pad_count``) -> captured step, all on the GPU."""
    from ..data import augment_batch_on_device, mask_sequence_on_device
    from ..distributed import FlatGradReducer
    from ..engine import CapturedTrainStep
    device = dataset.device
    bsz = int(config.batch_size)
    pad_structure = int(getattr(config, "structure_pad_count", 0) or 0)
    pad_sequence = int(getattr(config, "sequence_pad_count", 0) or 0)
    train_index = torch.as_tensor(train_index, dtype=torch.int64, device=device)
    val_index = torch.as_tensor(val_index, dtype=torch.int64, device=device)
    gen = torch.Generator(device="cpu").manual_seed(int(seed))
    dgen = torch.Generator(device=device).manual_seed(int(seed) + 1)
    amino = {}

    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final, pred = m(g, seq, prop)
        if m.training:
            return loss_function(recon, seq, mu, logvar, final, y, pred, amino[int(y.numel())])
        return loss_function(recon, seq, mu, logvar, final, y, torch.tensor([]), torch.tensor([]))

    def assemble(idx, buf, train):
        g, seq, prop, y = dataset.gather_into(idx, *buf)
        if train:
            b = int(idx.numel())
            if b not in amino:
                amino[b] = torch.zeros(b, dtype=torch.int64, device=device)
            amino[b].copy_(augment_batch_on_device(g.ndata["x"], b, dgen, structure_pad_count=pad_structure))
            mask_sequence_on_device(seq, pad_sequence, generator=dgen)
        return g, seq, prop, y

    captured = None
    if train_index.numel() >= bsz:
        buf = dataset.new_batch(bsz)
        model.train()
        assemble(train_index[:bsz], buf, True)
        reducer = FlatGradReducer(model.parameters(), world=1)
        captured = CapturedTrainStep(model, optimizer, reducer, forward_loss, buf, edge_capacity=bsz * dataset.max_edges,
                                     warmup=1, preserve_state=True)
    tails = {}

    def eager_batch(idx, train):
        b = int(idx.numel())
        if b not in tails:
            tails[b] = dataset.new_batch(b)
        return assemble(idx, tails[b], train)

    train_losses, val_losses = [], []
    best = float("inf")
    for epoch in range(config.num_epochs):
        model.train()
        perm = train_index[torch.randperm(train_index.numel(), generator=gen).to(device)]
        running, steps = None, 0
        for at in range(0, perm.numel(), bsz):
            idx = perm[at:at + bsz]
            if idx.numel() == bsz and captured is not None:
                assemble(idx, (captured.sgraph, captured.seq, captured.prop, captured.y), True)
                loss = captured.replay().clone()
            else:
                g, seq, prop, y = eager_batch(idx, True)
                optimizer.zero_grad(set_to_none=True)
                loss = forward_loss(model, g, seq, prop, y)
                loss.backward()
                optimizer.step()
                loss = loss.detach()
            running = loss if running is None else running + loss
            steps += 1
        train_loss = float(running) / max(steps, 1)
        train_losses.append(train_loss)
        if scheduler is not None:
            scheduler.step()
        model.eval()
        running, vsteps = None, 0
        with torch.no_grad():
            for at in range(0, val_index.numel(), bsz):
                g, seq, prop, y = eager_batch(val_index[at:at + bsz], False)
                loss = forward_loss(model, g, seq, prop, y).detach()
                running = loss if running is None else running + loss
                vsteps += 1
        val_total = float(running) if running is not None else 0.0
        if val_total < best:
            _save_best(config, model, stage)
            best = val_total
        val_loss = val_total / max(vsteps, 1)
        val_losses.append(val_loss)
        if wandb is not None and getattr(wandb, "run", None) is not None:
            wandb.log({stage + "_train_loss": train_loss, stage + "_val_loss": val_loss})
        print(f"Epoch {epoch + 1}, Train Loss: {train_loss:.4f}, Val Loss: {val_loss:.4f}")
    return train_losses, val_losses
