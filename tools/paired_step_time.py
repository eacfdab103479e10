"""Paired (cancer / wild-type) train step at BASELINE config 4 shape: eager vs captured HIP graph.  B pairs per step."""
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from immunostruct_amd import optim, synthetic  # noqa: E402
from immunostruct_amd.distributed import FlatGradReducer  # noqa: E402
from immunostruct_amd.engine import CapturedTrainStep  # noqa: E402
from immunostruct_amd.graph import PackedGraphBatch  # noqa: E402
from immunostruct_amd.models import model_map  # noqa: E402
from immunostruct_amd.procedures.train import _paired_loss  # noqa: E402
from immunostruct_amd.utils import Losses, PairedContrastiveLoss  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ONLY_MERGED = len(sys.argv) > 2 and sys.argv[2] == "merged"
dev = torch.device("cuda:0")
VAE_IN = synthetic.SEQ_LEN * synthetic.SEQ_ALPHABET


def member(seed):
    r = synthetic.make_batch(B, seed=seed, deg_extra=2)
    return r, PackedGraphBatch.from_raw(r, device=dev), torch.from_numpy(r.one_hot_sequence()).to(dev), torch.from_numpy(r.prop).to(dev)


batches = []
for s in range(3):
    (rc, gc, sc, pc), (rw, gw, sw, pw) = member(10 + s), member(50 + s)
    batches.append(((gc, gw), (sc, sw), (pc, pw), torch.from_numpy(rc.y_bin).to(dev), (rc.num_edges, rw.num_edges)))
caps = tuple(max(b[4][i] for b in batches) for i in range(2))
model = model_map["HybridModelv2_Comparative"](vae_input_dim=VAE_IN, device=dev, use_wt_for_downstream=True).to(dev)
model.train()
losses = Losses(VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
contrastive = PairedContrastiveLoss(device=dev, embedding_dim=104)
opt = optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-6)


def forward_loss(m, graphs, seqs, props, y):
    return _paired_loss(m, losses.BCE_loss, (graphs, seqs, y, props), dev, contrastive, 0.01)


def timed(step, n=30):
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def eager(i):
    g, s, p, y, _ = batches[i % 3]
    opt.zero_grad(set_to_none=True)
    forward_loss(model, g, s, p, y).backward()
    opt.step()


contrastive.capturable = True
t_c = float("nan")
if not ONLY_MERGED:
    eng = CapturedTrainStep(model, opt, FlatGradReducer(model.parameters(), world=1), forward_loss, batches[0][:4], edge_capacity=caps)
    t_c = timed(lambda i: eng(*batches[i % 3][:4]))
# merged form: one batch of 2B graphs [cancer; wild-type], one encoder pass
from immunostruct_amd.graph import batch as graph_batch  # noqa: E402
merged = []
for (gc, gw), (sc, sw), (pc, pw), y, (ec, ew) in batches:
    gc.csr(), gw.csr()
    merged.append((graph_batch([gc, gw]), torch.cat([sc, sw]), torch.cat([pc, pw]), torch.cat([y, y]), ec + ew))


def forward_loss_merged(m, g2, seq2, prop2, y2):
    return _paired_loss(m, losses.BCE_loss, (g2, seq2, y2[:y2.numel() // 2], prop2), dev, contrastive, 0.01)


eng2 = CapturedTrainStep(model, opt, FlatGradReducer(model.parameters(), world=1), forward_loss_merged, merged[0][:4],
                         edge_capacity=max(m[4] for m in merged))
t_m = timed(lambda i: eng2(*merged[i % 3][:4]))
print(f"captured, merged pair batch: {t_m:.3f} ms ({2 * B / t_m:.1f} k graphs/s)")
contrastive.capturable = False
t_e = timed(eager) if not ONLY_MERGED else float("nan")
print(f"paired step, B = {B} pairs ({2 * B} graphs): eager {t_e:.3f} ms ({2 * B / t_e:.1f} k graphs/s), "
      f"captured {t_c:.3f} ms ({2 * B / t_c:.1f} k graphs/s)")
