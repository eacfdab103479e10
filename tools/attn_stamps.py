"""Per-stage timeline of one workgroup of the attention backward (debug build with -DIS_STAGE_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from immunostruct_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_dbg", "libimmunostruct_hip_dbg.so")
from immunostruct_amd import functional as HF
dev = torch.device("cuda:0")
b, n = 128, 190
lib = _lib.load()
lib.is_debug_stamps_attn.argtypes = [ctypes.c_void_p]; lib.is_debug_stamps_attn.restype = ctypes.c_int
qk = torch.randn(b * n, 128, device=dev, requires_grad=True)
x = torch.randn(b * n, 64, device=dev, requires_grad=True)
names = ["start", "stage K issued", "dab+direct done", "barrier", "t_i done", "pass A done", "barrier", "Q staged", "pass B done"]
for rep in range(3):
    out = HF.attn_colmean(qk, x, b, n, 1)
    out.sum().backward()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 16)()
    assert lib.is_debug_stamps_attn(ctypes.cast(buf, ctypes.c_void_p)) == 0
    t = list(buf)[:9]
    print("attn bwd rep", rep, " ".join(f"{names[i]}:+{t[i]-t[i-1]}" for i in range(1, 9)), " total", t[8] - t[0])
