"""``PairedContrastiveLoss`` with the reference's interface (``utils/contrastive.py:6-83``).

Pulls the cancer / wild-type embedding pair together for non-immunogenic
peptides, pushes it apart for immunogenic ones, decorrelates non-pairs and
feature dimensions.  Same sub-module layout (``projector.{0,1,3}``), same
early-out (python ``0`` unless the target holds exactly two distinct values),
BatchNorm always on batch statistics (the reference never puts the module in
eval mode, ``procedures/train.py:76``).

Evaluated in closed form with diagonal/off-diagonal weights instead of the
reference's boolean-mask in-place scaling:
    loss = sum_ij w_ij (zc zw^T / D - diag(pos))_ij^2 + sum_kl w_kl (zc^T zw / B - I)_kl^2 + std hinge
with w = 1 on the diagonal and ``lambda_off_diag`` elsewhere.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["PairedContrastiveLoss"]


def _weighted_sq(m, ideal_diag, off_weight):
    """sum of squared deviation from diag(ideal_diag), off-diagonal entries weighted by ``off_weight``."""
    sq = m.pow(2)
    diag = torch.diagonal(m)
    on = (diag - ideal_diag).pow(2).sum()
    off = sq.sum() - diag.pow(2).sum()
    return on + off_weight * off


class PairedContrastiveLoss(nn.Module):
    def __init__(self, embedding_dim: int = 104, z_dim: int = 128, lambda_off_diag: float = 1e-2,
                 device: torch.device = torch.device("cpu")):
        super().__init__()
        self.z_dim = z_dim
        self.lambda_off_diag = lambda_off_diag
        self.projector = nn.Sequential(
            nn.Linear(embedding_dim, z_dim, bias=False),
            nn.BatchNorm1d(z_dim),
            nn.ReLU(inplace=True),
            nn.Linear(z_dim, z_dim, bias=False))
        self.device = device
        self.projector.to(device)
        self.capturable = False      # True: no host-side early-out (engine.CapturedTrainStep on paired batches)

    def prepare_targets(self, is_immunogenic):
        """capturable mode: (positive mask, two-class gate) of a target batch -- they depend on the targets alone, so a loop can
        issue this launch long before the embeddings exist (``procedures.train._contrastive_beside``) and hand the result to
        :meth:`forward` as ``targets``; None when not applicable"""
        if self.capturable and is_immunogenic.is_cuda:
            from .. import functional as HF
            return HF.contrastive_targets(is_immunogenic)
        return None

    def forward(self, embedding_cancer, embedding_wt, is_immunogenic, scale=1.0, targets=None):
        """the reference's signature plus ``scale``: returns scale * loss with the factor applied inside the HIP launches (the
        train loops pass their ``coeff_contrastive`` here instead of multiplying the result); ``targets``: the result of
        :meth:`prepare_targets` for the same ``is_immunogenic``"""
        from .. import _lib
        _lib.require_device(embedding_cancer, embedding_wt)      # no CPU path, as everywhere in this package
        gate = pos = None
        if targets is not None:
            pos, gate = targets
        elif self.capturable and is_immunogenic.is_cuda:
            # same rule without a host decision (a captured HIP graph cannot branch on data): the loss is always
            # evaluated -- it is finite for any target -- and multiplied by [the target holds exactly two distinct
            # values], computed on the device (one launch, csrc/contrastive.hip is_contrastive_targets, together with the
            # positive mask); value and gradients are then exactly the reference's 0 otherwise
            from .. import functional as HF
            pos, gate = HF.contrastive_targets(is_immunogenic)
        elif is_immunogenic.unique().numel() != 2:
            return 0  # nothing to contrast (continuous target, or a single-class batch)
        return self._loss(embedding_cancer, embedding_wt, is_immunogenic, pos, gate=gate, scale=scale)

    def _loss(self, embedding_cancer, embedding_wt, is_immunogenic, pos=None, gate=None, scale=1.0):
        if pos is None:
            pos = (is_immunogenic > is_immunogenic.mean()).to(embedding_cancer.dtype)
        if embedding_cancer.shape != embedding_wt.shape:
            raise AssertionError("cancer / wild-type embeddings must have equal shapes")
        if (embedding_cancer.is_cuda and self.z_dim == 128 and 2 <= embedding_cancer.shape[0] <= 256
                and embedding_cancer.dim() == 2 and embedding_cancer.shape[1] <= 256):
            # fused HIP path (csrc/contrastive.hip): the projector is frozen in the reference, so only the embedding
            # gradients are produced; BatchNorm running statistics are not maintained (never read: train mode only)
            from .. import functional as HF
            bn = self.projector[1]
            return HF.paired_contrastive(embedding_cancer, embedding_wt, pos, self.projector[0].weight.detach(),
                                         bn.weight.detach(), bn.bias.detach(), self.projector[3].weight.detach(),
                                         self.lambda_off_diag, gate=gate, scale=scale)
        if embedding_cancer.is_cuda:
            from .. import functional as HF
            HF.composed_path(f"paired contrastive loss over {tuple(embedding_cancer.shape)} embeddings, projector width {self.z_dim} "
                             "(kernels: 2 .. 256 pairs, embedding <= 256, width 128)")
        zc = self.projector(embedding_cancer)
        zw = self.projector(embedding_wt)
        b = zc.shape[0]
        zc = zc - zc.mean(0)
        zw = zw - zw.mean(0)
        hinge = 0.5 * (F.relu(1 - torch.sqrt(zc.var(dim=0) + 1e-4)).mean()
                       + F.relu(1 - torch.sqrt(zw.var(dim=0) + 1e-4)).mean())
        pair = zc @ zw.T / self.z_dim
        corr = zc.T @ zw / b
        ones = torch.ones(self.z_dim, dtype=zc.dtype, device=zc.device)
        loss = (_weighted_sq(pair, pos, self.lambda_off_diag)
                + _weighted_sq(corr, ones, self.lambda_off_diag) + hinge)
        if gate is not None:
            loss = loss * gate
        return loss if scale == 1.0 else loss * scale
