"""Shared test helpers: build oracle / product inputs from the same seeded raw batch."""
import os

import numpy as np
import torch

from immunostruct_amd import synthetic
from immunostruct_amd.graph import PackedGraphBatch
from oracle import graph_ref

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VAE_IN = synthetic.SEQ_LEN * synthetic.SEQ_ALPHABET
RECON_COLS = slice(0, None, 37)
ATTN_ROWS = [0, 57, 189]


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def oracle_graph(raw, dtype=torch.float32):
    g = graph_ref.RefGraph(raw.src, raw.dst, raw.num_nodes, raw.batch_num_nodes)
    g.ndata["x"] = torch.from_numpy(raw.x).to(dtype)
    g.edata["edge_attr"] = torch.from_numpy(raw.edge_attr).to(dtype)
    return g


def oracle_graph_permuted(raw, seed, dtype=torch.float32):
    """the same graph with its edge list in another order: mathematically identical, but every scatter-add of the oracle runs in
    another summation order -- a second realisation of the reference arithmetic's own fp32 round-off"""
    perm = np.random.RandomState(seed).permutation(raw.src.shape[0])
    g = graph_ref.RefGraph(raw.src[perm], raw.dst[perm], raw.num_nodes, raw.batch_num_nodes)
    g.ndata["x"] = torch.from_numpy(raw.x).to(dtype)
    g.edata["edge_attr"] = torch.from_numpy(raw.edge_attr[perm]).to(dtype)
    return g


def product_graph(raw, device):
    return PackedGraphBatch.from_raw(raw, device=device)


def make_eps(seed, batch, latent=32):
    return torch.from_numpy(np.random.RandomState(seed).normal(size=(batch, latent)).astype(np.float32))


def det_sd(shapes, seed):
    return {k: torch.from_numpy(v) for k, v in synthetic.det_state_dict(shapes, seed=seed).items()}


PEP_IN = (synthetic.SEQ_LEN - synthetic.HLA_LEN) * synthetic.SEQ_ALPHABET      # the reference's default width: 11 x 21


def peptide_one_hot(raw):
    """the padded peptide's one-hot (B, 11, 21): the sequence input WITHOUT --full-sequence (data/util_dataloader.py:52-66)"""
    return torch.from_numpy(raw.one_hot_sequence()[:, synthetic.HLA_LEN:]).contiguous()


def model_shapes(name, vae_in=VAE_IN, **kw):
    from immunostruct_amd.models import model_map
    m = model_map[name](vae_input_dim=vae_in, device="cpu", **kw)
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def egnn_shapes(dins, fe, prefix="GCN_layers."):
    shapes = {}
    for i, din in enumerate(dins):
        p = f"{prefix}{i}."
        shapes.update({p + "edge_mlp.0.weight": (64, 2 * din + 1 + fe), p + "edge_mlp.0.bias": (64,),
                       p + "edge_mlp.2.weight": (64, 64), p + "edge_mlp.2.bias": (64,),
                       p + "node_mlp.0.weight": (64, din + 64), p + "node_mlp.0.bias": (64,),
                       p + "node_mlp.2.weight": (64, 64), p + "node_mlp.2.bias": (64,),
                       p + "coord_mlp.0.weight": (64, 64), p + "coord_mlp.0.bias": (64,),
                       p + "coord_mlp.2.weight": (1, 64)})
    return shapes


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny): error relative to the tensor's scale (reported, not the pass criterion)."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    scale = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / scale


ATOL_FRACTION = 0.5


def worst_ratio(a, b, tol):
    """max over the entries of |a - b| / (atol + rtol |b|) -- the quantity :func:`assert_close` bounds by 1"""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    scale = max(float(b.abs().max()), 1e-30)
    bound = ATOL_FRACTION * tol * scale + tol * b.abs()
    return float(((a - b).abs() / bound).max()) if a.numel() else 0.0


def assert_close(a, b, tol, what=""):
    """MIXED tolerance, per entry: |a - b| <= atol + rtol * |b| with rtol = ``tol`` and atol = 0.5 * tol * max|b|.  The
    absolute floor is tied to the tensor's scale (entries that are small against the largest one carry cancellation error
    of that size), so for entries much smaller than max|b| this is a max-norm check at 0.5 x ``tol``, and a relative one
    (up to 1.5 x ``tol``) only for the entries near the maximum -- it is NOT a pure relative check.
    Returns the scaled max error (max|a-b| / max|b|) for reporting."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = max(float(b.abs().max()), 1e-30)
    diff = (a - b).abs()
    bound = ATOL_FRACTION * tol * scale + tol * b.abs()
    worst = float((diff / bound).max()) if diff.numel() else 0.0
    assert worst <= 1.0, (f"{what}: element-wise error is {worst:.2f} x the bound (rtol {tol:.1e}, atol {ATOL_FRACTION * tol * scale:.2e}); "
                          f"scaled max error {float(diff.max()) / scale:.3e}")
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    row = _row_worst(b, diff, tol)
    if ROW_REPORT:
        with open(ROW_REPORT, "a") as f:
            f.write(f"{row:.3f}\t{worst:.3f}\t{tuple(b.shape)}\t{tol:.1e}\t{what}\n")
    assert row <= ROW_FACTOR, (f"{what}: against the scale of its own ROW an entry is {row:.2f} x the bound (allowed {ROW_FACTOR:g} x; "
                               f"rtol {tol:.1e})")
    return float(diff.max()) / scale


ROW_REPORT = __import__("os").environ.get("IMMUNOSTRUCT_TEST_ROW_REPORT")      # calibration aid: file that collects the row-wise ratios
# Second criterion of assert_close (round 4; VERDICT r02 / r03 "entries much smaller than max|b| remain invisible"): the same mixed
# bound with the absolute floor tied to the scale of the entry's own ROW (max |b| over the last dimension) instead of the whole
# tensor's, times ROW_FACTOR.  Rows that are small against the tensor's maximum -- a weight-gradient row of a channel that is
# rarely active, a node with small features -- are then checked against their own size, within a factor.  The factor is measured:
# over the whole GPU suite (IMMUNOSTRUCT_TEST_ROW_REPORT) every tensor but three sits below 2.2 x, the worst -- rows of
# GCN_layers.0.edge_mlp.2.weight's gradient that are sums of ~10^3 cancelling terms, against an fp32 reference that carries the
# same kind of error -- at 9.4 x.
# Round 5: 16 -> 12 (the whole GPU suite re-measured: worst 9.1 x -- the same tensor --, second 7.1 x, every other tensor < 3.3 x).
# Round 6: 12 -> 10 (worst 9.1 x, the same tensor; every other comparison of the suite <= 7.1 x, all but four < 2.2 x).
ROW_FACTOR = 10.0


def _row_worst(b, diff, tol):
    """max over the entries of |a - b| / (0.5 tol max|b[row]| + tol |b|), rows that are (round-off) zero against the tensor's
    scale left out; 0 for tensors without a last dimension to speak of"""
    if b.dim() < 2 or b.shape[-1] < 2 or not diff.numel():
        return 0.0
    rs = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    ratio = diff / (ATOL_FRACTION * tol * rs + tol * b.abs())
    live = (rs > 1e-6 * float(b.abs().max())).expand_as(ratio)
    return float(ratio[live].max()) if live.any() else 0.0


# ---- Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), numpy: the checker of
# csrc/abi_misc.hip step_random_kernel (pinned to the Random123 known-answer vectors in tests/test_step_contexts.py) ----
def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32, key: (2,) uint32 -> (..., 4) uint32"""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    m0, m1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = m0 * c[0], m1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def step_random_expected(n, kind, p, seed, step, job):
    """what is_step_random writes for a job of n elements: float64 values (normals) / exact float32 values (masks)"""
    quads = (n + 3) // 4
    ctr = np.zeros((quads, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(quads, dtype=np.uint64) & 0xFFFFFFFF
    ctr[:, 1] = np.uint32(job << 24)
    ctr[:, 2], ctr[:, 3] = np.uint32(step & 0xFFFFFFFF), np.uint32(step >> 32)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))
    u = ((r >> 8).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    if kind == 0:
        u = u.astype(np.float64)
        rad = np.sqrt(-2.0 * np.log(u[:, 0::2]))
        ang = 2.0 * np.pi * u[:, 1::2].astype(np.float32).astype(np.float64)
        out = np.stack([rad[:, 0] * np.cos(ang[:, 0]), rad[:, 0] * np.sin(ang[:, 0]), rad[:, 1] * np.cos(ang[:, 1]), rad[:, 1] * np.sin(ang[:, 1])], axis=1)
    else:
        out = np.where(u >= np.float32(p), np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32)
    return out.reshape(-1)[:n]
