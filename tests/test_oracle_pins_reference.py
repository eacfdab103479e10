"""Pin the oracle to the reference's OWN code (build container only: needs /root/reference).

The reference's models / losses / contrastive loss are imported unchanged under
``oracle/shims.py`` and must agree bit-for-bit (values) with the functional oracle
for all 14 ``model_map`` entries, and to rounding for gradients.
"""
import unittest.mock as mock

import pytest
import torch

from immunostruct_amd import synthetic
from oracle import functional_ref as FR
from oracle import shims
from tests import helpers as H

pytestmark = pytest.mark.skipif(not shims.reference_available(), reason="reference sources not present on this box")
torch.set_num_threads(1)


@pytest.fixture(scope="module")
def ref():
    return shims.load_reference()


def _with_eps(fn, eps_list):
    it = iter(eps_list)
    with mock.patch("torch.randn_like", lambda t: next(it).to(t.dtype)):
        return fn()


@pytest.mark.parametrize("name", sorted(FR.VARIANTS))
def test_every_variant_forward_is_bit_identical(ref, name):
    model_map, _, _ = ref
    raw = synthetic.make_batch(3, seed=5)
    g = H.oracle_graph(raw)
    seq, prop = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop)
    model = model_map[name](vae_input_dim=H.VAE_IN, device="cpu")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert shapes == H.model_shapes(name), "product state_dict layout differs from the reference"
    sd = H.det_sd(shapes, seed=7)
    model.load_state_dict(sd)
    model.eval()
    eps = H.make_eps(3, 3)
    with torch.no_grad():
        res = _with_eps(lambda: model(g, seq, prop), [eps])
    mine = FR.as_reference_tuple(name, FR.forward(name, sd, g, seq, prop, eps=eps))
    assert len(res) == len(mine)
    for a, b in zip(res, mine):
        if torch.is_tensor(a):
            assert torch.equal(a, b)
        else:
            assert a == b == 0


def test_training_mode_dropout_masks_are_honoured(ref):
    """oracle's explicit keep-masks == reference nn.Dropout given the same mask draw."""
    model_map, _, _ = ref
    raw = synthetic.make_batch(3, seed=6)
    g = H.oracle_graph(raw)
    seq, prop = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop)
    model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device="cpu")
    sd = H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=8)
    model.load_state_dict(sd)
    model.train()
    eps = H.make_eps(4, 3)
    torch.manual_seed(123)
    res = _with_eps(lambda: model(g, seq, prop), [eps])
    torch.manual_seed(123)  # replay the two bernoulli draws nn.Dropout made (property MLP, classifier)
    m_prop = torch.nn.functional.dropout(torch.ones(3, 32), 0.1, True) * 0.9
    m_cls = torch.nn.functional.dropout(torch.ones(3, 32), 0.1, True) * 0.9
    it = FR.forward("HybridModelv2", sd, g, seq, prop, eps=eps, drop={"prop": m_prop, "cls": m_cls})
    assert torch.allclose(res[3], it["final_output"], rtol=1e-6, atol=1e-7)


def test_losses_and_contrastive_gradients(ref):
    _, Losses, PCL = ref
    torch.manual_seed(0)
    b = 12
    recon = torch.randn(b, H.VAE_IN, requires_grad=True)
    x = torch.from_numpy(synthetic.make_batch(b, seed=2).one_hot_sequence())
    mu, lv = torch.randn(b, 32, requires_grad=True), torch.randn(b, 32, requires_grad=True)
    logit = torch.randn(b, 1, requires_grad=True)
    y = (torch.rand(b) < 0.4).float()
    losses = Losses(H.VAE_IN, {0: 70.0, 1: 30.0}, sequence=True)
    for ref_fn, mine_fn, target in ((losses.regression_loss, lambda *a: FR.regression_loss(*a, H.VAE_IN), torch.randn(b)),
                                    (losses.BCE_loss, lambda *a: FR.bce_loss(*a, H.VAE_IN, 70.0 / 30.0), y)):
        ga = torch.autograd.grad(ref_fn(recon, x, mu, lv, logit, target), [recon, mu, lv, logit])
        gb = torch.autograd.grad(mine_fn(recon, x, mu, lv, logit, target), [recon, mu, lv, logit])
        for a, c in zip(ga, gb):
            assert torch.allclose(a, c, rtol=1e-6, atol=1e-9)
    pcl = PCL(embedding_dim=104)
    psd = {k: v.detach().clone() for k, v in pcl.state_dict().items()}
    ec, ew = torch.randn(b, 104, requires_grad=True), torch.randn(b, 104, requires_grad=True)
    va = pcl(ec, ew, y)
    vb = FR.paired_contrastive_loss(psd, ec, ew, y)
    assert abs(float(va) - float(vb)) <= 1e-5 * abs(float(va))
    ga = torch.autograd.grad(va, [ec, ew])
    gb = torch.autograd.grad(vb, [ec, ew])
    for a, c in zip(ga, gb):
        assert torch.allclose(a, c, rtol=1e-4, atol=1e-7)
