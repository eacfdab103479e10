"""Stand-alone timing of the fusion head's two launches (is_comb_attn_cls_fwd / _bwd) at the bench shape (B = 128, T = 104, F = 16):
HIP events over 300 calls, for A/B builds (IMMUNOSTRUCT_LIB=...).   python tools/head_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from immunostruct_amd import functional as HF  # noqa: E402
from immunostruct_amd.models.layers import MultiHeadAttention  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    b, t = 128, 104
    mha = MultiHeadAttention(16, 8, input_dim=1).to(dev)
    cls = torch.nn.Sequential(torch.nn.Flatten(1), torch.nn.Linear(t, 32), torch.nn.ReLU(True), torch.nn.Dropout(0.1), torch.nn.Linear(32, 1)).to(dev)
    cls.train()
    mask = HF.dropout_mask(b, 32, 0.1, dev)
    pieces = [torch.randn(b, 64, device=dev, requires_grad=True), torch.randn(b, 40, device=dev, requires_grad=True)]
    g = torch.randn(b, 1, device=dev)

    def fwd():
        with torch.no_grad():
            return HF.combined_attention_classifier(pieces, mha, cls, mask=mask)

    def both():
        y = HF.combined_attention_classifier(pieces, mha, cls, mask=mask)
        y.backward(g)

    for fn in (fwd, both):
        for _ in range(20):
            fn()
    res = {}
    for name, fn in (("forward", fwd), ("forward + backward", both)):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(50):
            fn()
        e0.record()
        for _ in range(300):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 300 * 1e3
    print(f"fusion head: forward {res['forward']:.1f} us, forward + backward {res['forward + backward']:.1f} us "
          f"(host-launch bound if the two are far above the kernels' own ~20 / ~30 us)")


if __name__ == "__main__":
    main()
