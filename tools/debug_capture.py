import sys, os, unittest.mock as mock
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from immunostruct_amd import synthetic
from immunostruct_amd.models import model_map
from immunostruct_amd.utils import Losses
from immunostruct_amd.distributed import FlatGradReducer
from immunostruct_amd.engine import CapturedTrainStep
from tests import helpers as H
dev = torch.device("cuda:0")
raws = [synthetic.make_batch(6, seed=s, deg_extra=d) for s, d in ((51, 2), (52, 4), (53, 1), (54, 3))]
batches = [(H.product_graph(r, dev), torch.from_numpy(r.one_hot_sequence()).to(dev), torch.from_numpy(r.prop).to(dev), torch.from_numpy(r.y_reg).to(dev)) for r in raws]
losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
eps = H.make_eps(9, 6).to(dev)
def forward_loss(m, g, seq, prop, y):
    with mock.patch("torch.randn_like", lambda t: eps.to(t.dtype)):
        recon, mu, logvar, final = m(g, seq, prop)
    return losses.regression_loss(recon, seq, mu, logvar, final, y)
def mk():
    model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
    model.load_state_dict(H.det_sd({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=6)); model.eval()
    red = FlatGradReducer(model.parameters(), world=1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-5, fused=True, capturable=True)
    return model, red, opt
ms, rs, os_ = mk(); rs.zero(); forward_loss(ms, *batches[0]).backward(); os_.step(); torch.cuda.synchronize()  # initialise hipBLASLt etc.
me, re_, oe = mk(); mc, rc, oc = mk()
eng = CapturedTrainStep(mc, oc, rc, forward_loss, batches[0], edge_capacity=max(r.num_edges for r in raws), warmup=0)
names = [n for n, p in me.named_parameters() if p.requires_grad]
sizes = [p.numel() for n, p in me.named_parameters() if p.requires_grad]
for i, b in enumerate(batches):
    re_.zero(); le = forward_loss(me, *b); le.backward()
    lc = eng(*b)
    torch.cuda.synchronize()
    d = (re_.flat - rc.flat).abs()
    print(f"step {i}: E={raws[i].num_edges} loss eager {float(le.detach()):.7f} captured {float(lc):.7f} max|dgrad| {float(d.max()):.3e} gradmax {float(re_.flat.abs().max()):.3e}")
    off = 0
    worst = []
    for n, s in zip(names, sizes):
        m = float(d[off:off+s].max()); ref = float(re_.flat[off:off+s].abs().max())
        if m > 1e-6 * max(ref, 1e-12): worst.append((m / max(ref, 1e-30), n))
        off += s
    print("   params with grad mismatch:", sorted(worst, reverse=True)[:8])
    oe.step()
    pd = max(float((a - b2).abs().max()) for a, b2 in zip(me.parameters(), mc.parameters()))
    print("   max param diff after step", pd)
