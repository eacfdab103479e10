"""``Losses`` with the reference's call signatures (``utils/loss.py:5-61``), HIP-backed.

``regression_loss`` = 2*MSE(pred, y) + 0.5*MSE(recon, x) + 0.5*KLD_mean
``BCE_loss``        = 5*BCEWithLogits(pred, y, pos_weight=n0/n1) + 0.1*MSE + 0.1*KLD_mean
``sequence=False`` drops the VAE terms (and their coefficients: the prediction
term is then un-weighted, as in the reference).  The ``*_SSL`` variants add the
amino-acid cross-entropy of the masked residue.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import functional as HF

_REGRESSION, _BCE = 0, 1
_COEFFS = {_REGRESSION: (2.0, 0.5, 0.5), _BCE: (5.0, 0.1, 0.1)}


class Losses:
    def __init__(self, vae_input_dim, class_weights, sequence=True):
        self.vae_input_dim = vae_input_dim
        self.sequence = sequence
        self.pos_weight = torch.tensor(float(class_weights[0]) / float(class_weights[1])).float()
        self._pos_weight_f = float(self.pos_weight)

    def _fused(self, mode, recon_x, x, mu, logvar, final_output, y):
        if self.sequence:
            c_pred, c_mse, c_kld = _COEFFS[mode]
            recon, target = recon_x, x.reshape(-1, self.vae_input_dim)
        else:
            c_pred, c_mse, c_kld = 1.0, 0.0, 0.0
            recon = target = mu = logvar = None
        total, _terms = HF.vae_loss(recon, target, mu, logvar, final_output, y, mode, self._pos_weight_f,
                                    c_pred, c_mse, c_kld)
        return total

    def regression_loss(self, recon_x, x, mu, logvar, final_output, y):
        return self._fused(_REGRESSION, recon_x, x, mu, logvar, final_output, y)

    def BCE_loss(self, recon_x, x, mu, logvar, final_output, y):
        return self._fused(_BCE, recon_x, x, mu, logvar, final_output, y)

    @staticmethod
    def _amino(pred_amino_acid, amino_acid):
        return F.cross_entropy(pred_amino_acid, amino_acid) if pred_amino_acid.numel() else 0

    def regression_loss_SSL(self, recon_x, x, mu, logvar, final_output, y, pred_amino_acid, amino_acid):
        return self.regression_loss(recon_x, x, mu, logvar, final_output, y) + self._amino(pred_amino_acid, amino_acid)

    def BCE_loss_SSL(self, recon_x, x, mu, logvar, final_output, y, pred_amino_acid, amino_acid):
        return self.BCE_loss(recon_x, x, mu, logvar, final_output, y) + self._amino(pred_amino_acid, amino_acid)
