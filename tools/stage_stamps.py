"""Stage time-stamps (shader clocks) of the forward layer kernel: one wave of workgroup 300 through its chunk
(debug build with -DIS_STAGE_STAMPS in gpurun_dbg/; the backward has tools/bwd_stamps.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from immunostruct_amd import _lib
_lib.LIB_PATH = os.environ.get("IMMUNOSTRUCT_DBG_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_dbg", "libimmunostruct_hip_dbg.so")
from immunostruct_amd import synthetic
from immunostruct_amd.graph import PackedGraphBatch
from immunostruct_amd.nn import EGNNConv, egnn_stack_forward
dev = torch.device("cuda:0")
raw = synthetic.make_batch(int(os.environ.get("B", 128)), seed=1)
g = PackedGraphBatch.from_raw(raw, device=dev)
layers = [EGNNConv(20 if i == 0 else 64, 64, 64, 1).to(dev) for i in range(2)]
lib = _lib.load()
lib.is_debug_stamps3.argtypes = [ctypes.c_void_p]; lib.is_debug_stamps3.restype = ctypes.c_int
lab = ["S0", "SA", "MM1", "MM2", "SEG"]
for rep in range(3):
    with torch.no_grad():
        egnn_stack_forward(layers, g, g.ndata["x"][:, :20], g.ndata["x"][:, 20:], g.edata["edge_attr"])
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    assert lib.is_debug_stamps3(ctypes.cast(buf, ctypes.c_void_p)) == 0
    t = list(buf)
    node = t[56:62]
    t = t[:56]
    n = max(i for i in range(56) if t[i] > 0) + 1
    print("fwd rep", rep, f"prologue (ids, rows, weights staged):+{t[1] - t[0]}",
          " ".join(f"{lab[(i - 2) % 5]}:+{t[i] - t[i - 1]}" for i in range(2, n)), " edge total", t[n - 1] - t[0],
          "| node half: wait for the workgroup's waves:+%d stage X:+%d MM zn1:+%d MM h':+%d MM psd':+%d | kernel %d" % (
              node[1] - node[0], node[2] - node[1], node[3] - node[2], node[4] - node[3], node[5] - node[4], node[5] - t[0]))
