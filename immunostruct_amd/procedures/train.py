"""Training loops with the reference's signatures (``procedures/train.py:10-67`` and ``:70-185``).

``train_model(config, device, model, train_loader, val_loader, optimizer, loss_function, scheduler=None,
stage="pretrain")`` and ``train_model_comparative(...)``: epoch loop, forward, loss, backward, optimizer
step, validation pass, best-validation checkpoint (``config.model_save_path_{pretrain,finetune}``),
optional wandb logging.  Differences that do not change results: the running loss is accumulated on
the device and read back once per epoch (the reference calls ``loss.item()`` every step, a host sync
per step), and ``wandb`` is optional.
"""
from __future__ import annotations

import os

import torch

from ..utils import PairedContrastiveLoss

try:  # optional, exactly as inert as a disabled wandb run when missing
    import wandb
except Exception:  # pragma: no cover
    wandb = None

__all__ = ["train_model", "train_model_comparative", "train_model_device", "train_model_comparative_device"]


def _to(device, obj):
    if isinstance(obj, (tuple, list)):
        return type(obj)(_to(device, o) for o in obj)
    return obj.to(device)


def _save_best(config, model, stage):
    path = config.model_save_path_pretrain if stage == "pretrain" else config.model_save_path_finetune
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save(model.state_dict(), path)


def _single_loss(model, loss_function, batch, device, extra=None):
    graph, seq, target, prop = _to(device, batch)
    recon, mu, logvar, final = model(graph, seq, prop)
    return loss_function(recon, seq, mu, logvar, final, target)


def _paired_loss(model, loss_function, batch, device, contrastive, coeff):
    graphs, seqs, target, props = _to(device, batch)
    emb, recon, mu, logvar, final = model.forward_comparative(graphs, seqs, props)
    merged = [getattr(t, "merged", None) for t in (recon, mu, logvar)]
    if all(m is not None for m in merged) and merged[0].shape[0] == 2 * recon[0].shape[0]:
        # one encoder pass produced both members (equal sizes): 0.5 * (loss(cancer) + loss(wild-type)) -- shared
        # prediction term, averaged reconstruction / KLD means (reference :107-114) -- is the loss over the merged rows
        seq2 = seqs if torch.is_tensor(seqs) else torch.cat([seqs[0], seqs[1]], dim=0)
        loss = loss_function(merged[0], seq2, merged[1], merged[2], final, target)
    else:
        if torch.is_tensor(seqs):       # merged batch [cancer; wild-type] (the on-GPU batcher's form)
            half = seqs.shape[0] // 2
            seqs = (seqs[:half], seqs[half:])
        # the prediction term is shared, the reconstruction terms are averaged (reference :107-114)
        loss = 0.5 * (loss_function(recon[0], seqs[0], mu[0], logvar[0], final, target)
                      + loss_function(recon[1], seqs[1], mu[1], logvar[1], final, target))
    return _add_contrastive(loss, contrastive, emb, target, coeff)


def _add_contrastive(loss, contrastive, emb, target, coeff):
    """loss + coeff * contrastive(cancer embedding, wild-type embedding, target) (reference ``procedures/train.py:116-117``)"""
    if coeff > 0:
        if isinstance(contrastive, PairedContrastiveLoss):      # the coefficient rides inside the loss launches
            loss = loss + _contrastive_beside(contrastive, emb, target, coeff, loss)
        else:
            loss = loss + coeff * contrastive(emb[0], emb[1], target)
    return loss


def _contrastive_beside(contrastive, emb, target, coeff, loss):
    """The paired contrastive loss depends on the two embeddings only -- not on the fusion head or the other loss terms -- so it is
    evaluated on the models' side stream (idle between the sequence branch's forward and backward): its ~ 100 us forward and,
    through autograd's stream bookkeeping, its ~ 95 us backward then run BESIDE the head's forward / loss / backward instead of
    between them (also inside a captured step: fork / join).  ``target`` must be ready on the side stream: the loops' targets
    are step inputs, complete before the forward starts."""
    from ..models import _core
    if not (_core.OVERLAP_BRANCHES and torch.is_tensor(loss) and loss.is_cuda and emb[0].is_cuda):
        return contrastive(emb[0], emb[1], target, scale=coeff)
    main = torch.cuda.current_stream()
    side = _core._side_stream(loss.device)
    ready = getattr(emb[0], "_ready_event", None)      # recorded by the model right behind the embeddings
    pre = None
    if ready is not None and hasattr(contrastive, "prepare_targets"):
        with torch.cuda.stream(side):      # positive mask + two-class gate: they need the targets only, not the embeddings
            pre = contrastive.prepare_targets(target)
    if ready is not None:
        side.wait_event(ready)       # not the fusion head / the other loss terms enqueued since: they run beside this loss
    else:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        c = contrastive(emb[0], emb[1], target, scale=coeff, **({"targets": pre} if pre is not None else {}))
    main.wait_stream(side)
    if torch.is_tensor(c):
        c.record_stream(main)
        for e in emb:
            e.record_stream(side)
    return c


def _fit(config, model, train_loader, val_loader, optimizer, scheduler, stage, step_loss):
    train_losses, val_losses = [], []
    best = float("inf")
    for epoch in range(config.num_epochs):
        model.train()
        running = None
        for batch in train_loader:
            optimizer.zero_grad(set_to_none=True)
            loss = step_loss(batch)
            loss.backward()
            optimizer.step()
            running = loss.detach() if running is None else running + loss.detach()
        train_loss = float(running) / max(len(train_loader), 1)
        train_losses.append(train_loss)
        if scheduler is not None:
            scheduler.step()
        model.eval()
        running = None
        with torch.no_grad():
            for batch in val_loader:
                loss = step_loss(batch)
                running = loss.detach() if running is None else running + loss.detach()
        val_total = float(running) if running is not None else 0.0
        if val_total < best:
            _save_best(config, model, stage)
            best = val_total
        val_loss = val_total / max(len(val_loader), 1)
        val_losses.append(val_loss)
        if wandb is not None and getattr(wandb, "run", None) is not None:
            wandb.log({stage + "_train_loss": train_loss, stage + "_val_loss": val_loss})
        print(f"Epoch {epoch + 1}, Train Loss: {train_loss:.4f}, Val Loss: {val_loss:.4f}")
    return train_losses, val_losses


def train_model(config, device, model, train_loader, val_loader, optimizer, loss_function, scheduler=None, stage="pretrain"):
    return _fit(config, model, train_loader, val_loader, optimizer, scheduler, stage,
                lambda batch: _single_loss(model, loss_function, batch, device))


def train_model_comparative(config, device, model, train_loader, val_loader, optimizer, loss_function, scheduler=None,
                            stage="pretrain"):
    coeff = float(getattr(config, "coeff_contrastive", 0) or 0)
    # as in the reference (:74-78) the projector is created here, is NOT handed to the optimizer and stays in
    # train mode (batch statistics) -- gradients still flow through it into the embeddings
    contrastive = PairedContrastiveLoss(device=device, embedding_dim=104) if coeff > 0 else None
    return _fit(config, model, train_loader, val_loader, optimizer, scheduler, stage,
                lambda batch: _paired_loss(model, loss_function, batch, device, contrastive, coeff))


def _device_fit(config, model, optimizer, scheduler, stage, seed, device, train_index, val_index, new_batch, assemble,
                forward_loss, edge_capacity):
    """The epoch loop shared by the four device-resident training loops.

    ``new_batch(b)`` -> static buffers ``(graph, seq, prop, y)`` for b items; ``assemble(idx, buf, train)`` fills them with the
    items ``idx`` (device int64 ids) -- gather, and for the self-supervised loops the train-time augmentation -- and returns
    the tuple ``forward_loss(model, *tuple)`` consumes.  Every full batch is one replay of the captured HIP graph
    (``engine.CapturedTrainStep``); a trailing partial batch runs eagerly.

    Data parallel (SURVEY.md section 8 e): when ``torch.distributed`` is initialised (one process per GPU, every rank
    holding the dataset) rank r trains on ``perm[r::world]`` of each epoch's permutation (the same seeded permutation on
    all ranks, cut to a multiple of the world size), ``config.batch_size`` items per rank and step; gradients are all-reduced
    through ``distributed.FlatGradReducer`` (inside the captured step: overlapped with the backward when that measures
    faster), rank 0's initial weights are broadcast, the device random streams (reparameterisation noise, dropout,
    augmentation) are offset per rank, every rank validates on the full validation set and rank 0 writes the checkpoint."""
    import torch.distributed as dist
    from ..distributed import FlatGradReducer, broadcast_parameters
    from ..engine import CapturedTrainStep
    bsz = int(config.batch_size)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    train_index = torch.as_tensor(train_index, dtype=torch.int64, device=device)
    val_index = torch.as_tensor(val_index, dtype=torch.int64, device=device)
    gen = torch.Generator(device="cpu").manual_seed(int(seed))
    if dist.is_initialized():      # (world 1 only under IMMUNOSTRUCT_FORCE_COLLECTIVE=1: the collectives are issued all the same)
        broadcast_parameters(model)
        # identical seeds on every rank would make the ranks draw the same noise / dropout masks for their (different) samples
        torch.cuda.manual_seed(int(seed) + 7919 * (rank + 1))

    captured = None
    reducer = FlatGradReducer(model.parameters(), world=world)
    if train_index.numel() // world >= bsz:
        buf = new_batch(bsz)
        model.train()
        first = assemble(train_index[rank:rank + bsz * world:world][:bsz], buf, True)
        # the engine's construction runs one eager warm-up step on this batch and then restores model + optimizer
        # (step_random: the captured step's dropout masks / noise from the library's generator, seeded from the device seed set
        #  above and the STAGE -- pretrain 0, finetune 1: a property of the run's stage, not of what the process built before;
        #  config.step_random = None keeps torch's generator inside the step)
        captured = CapturedTrainStep(model, optimizer, reducer, forward_loss, first, edge_capacity=edge_capacity(bsz),
                                     warmup=1, preserve_state=True, step_random=getattr(config, "step_random", "device"),
                                     random_stream=0 if stage == "pretrain" else 1)
    tails = {}

    def eager_batch(idx, train):
        b = int(idx.numel())
        if b not in tails:
            tails[b] = new_batch(b)
        return assemble(idx, tails[b], train)

    train_losses, val_losses = [], []
    best = float("inf")
    for epoch in range(config.num_epochs):
        model.train()
        perm = train_index[torch.randperm(train_index.numel(), generator=gen).to(device)]
        if world > 1:
            perm = perm[: (perm.numel() // world) * world][rank::world]      # this rank's shard: equal length on every rank
        running, steps = None, 0
        for at in range(0, perm.numel(), bsz):
            idx = perm[at:at + bsz]
            if idx.numel() == bsz and captured is not None:
                assemble(idx, (captured.sgraph, captured.seq, captured.prop, captured.y), True)
                loss = captured.replay().clone()
            else:
                args = eager_batch(idx, True)
                with reducer.live_gradients():      # not the captured graph's (stale) gradient buffers
                    reducer.zero()
                    loss = forward_loss(model, *args)
                    loss.backward()
                    reducer.all_reduce_mean()
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
                loss = forward_loss(model, *eager_batch(val_index[at:at + bsz], False)).detach()
                running = loss if running is None else running + loss
                vsteps += 1
        val_total = float(running) if running is not None else 0.0
        if val_total < best:
            if rank == 0:
                _save_best(config, model, stage)
            best = val_total
        val_loss = val_total / max(vsteps, 1)
        val_losses.append(val_loss)
        if rank == 0:
            if wandb is not None and getattr(wandb, "run", None) is not None:
                wandb.log({stage + "_train_loss": train_loss, stage + "_val_loss": val_loss})
            print(f"Epoch {epoch + 1}, Train Loss: {train_loss:.4f}, Val Loss: {val_loss:.4f}")
    if captured is not None:
        captured.close()      # the captured graphs (they may hold RCCL's nodes) go before the caller can destroy the process group
    return train_losses, val_losses


def _reads_recon_through_losses(loss_function):
    """the promise ``engine.CapturedTrainStep`` asks for (``forward_loss.fused_loss``): the loss is a method of ``utils.Losses``, which
    hands the reconstruction to ``functional.vae_loss`` -- any other callable gets the full stream join in the models"""
    from ..utils import Losses
    return isinstance(getattr(loss_function, "__self__", None), Losses)


def train_model_device(config, device, model, dataset, train_index, val_index, optimizer, loss_function, scheduler=None,
                       stage="pretrain", seed=0):
    """``train_model`` on a :class:`~immunostruct_amd.data.DeviceResidentDataset`: same epoch
    loop, loss, best-validation checkpoint and printed / returned values as :func:`train_model` with a shuffling
    ``DataLoader(batch_size=config.batch_size)``, but batches are assembled on the GPU from graph ids
    (``dataset.gather_into``) and every full batch runs as one replay of the captured HIP graph; data parallel under an
    initialised process group (:func:`_device_fit`).  ``train_index`` / ``val_index``: graph ids (any integer sequence)."""
    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final = m(g, seq, prop)
        return loss_function(recon, seq, mu, logvar, final, y)
    forward_loss.fused_loss = _reads_recon_through_losses(loss_function)

    return _device_fit(config, model, optimizer, scheduler, stage, seed, dataset.device, train_index, val_index,
                       dataset.new_batch, lambda idx, buf, train: dataset.gather_into(idx, *buf), forward_loss,
                       lambda b: b * dataset.max_edges)


def train_model_comparative_device(config, device, model, dataset_cancer, dataset_wt, train_index, val_index, optimizer,
                                   loss_function, scheduler=None, stage="pretrain", seed=0):
    """``train_model_comparative`` with both members of every (cancer, wild-type) pair in device-resident datasets
    (same graph ids in both; targets are the cancer dataset's).  A batch of B pairs is assembled on the GPU as ONE merged
    batch of 2B graphs [cancer; wild-type] (one gather from the concatenated dataset) which the paired model encodes in a
    single pass (``MultimodalNet._encode_pair``); the loss is the reference's (``procedures/train.py:97-114``: shared
    prediction term, averaged reconstruction terms, ``coeff_contrastive`` x paired contrastive loss).  Every full batch is
    one replay of the captured HIP graph of that step; the contrastive loss' class-count early-out
    (``utils/contrastive.py:38-43``) is evaluated on the device there (``PairedContrastiveLoss.capturable``).  Data parallel
    under an initialised process group (:func:`_device_fit`; the contrastive loss is computed per rank on its local pairs)."""
    from ..data import DeviceResidentDataset
    device = dataset_cancer.device
    coeff = float(getattr(config, "coeff_contrastive", 0) or 0)
    contrastive = PairedContrastiveLoss(device=device, embedding_dim=104) if coeff > 0 else None
    if contrastive is not None:
        contrastive.capturable = True
    both = DeviceResidentDataset.concat(dataset_cancer, dataset_wt)
    shift = len(dataset_cancer)

    def forward_loss(m, g2, seq2, prop2, y2):
        # y2 holds the targets of both members; the pair's label is the cancer member's
        return _paired_loss(m, loss_function, (g2, seq2, y2[:y2.numel() // 2], prop2), device, contrastive, coeff)
    forward_loss.fused_loss = _reads_recon_through_losses(loss_function)

    return _device_fit(config, model, optimizer, scheduler, stage, seed, device, train_index, val_index,
                       lambda b: both.new_batch(2 * b), lambda idx, buf, train: both.gather_into(torch.cat([idx, idx + shift]), *buf),
                       forward_loss, lambda b: 2 * b * both.max_edges)
