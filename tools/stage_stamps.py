"""Per-stage timeline of one edge-forward workgroup (debug build with -DIS_STAGE_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from immunostruct_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_dbg", "libimmunostruct_hip_dbg.so")
from immunostruct_amd import synthetic
from immunostruct_amd.graph import PackedGraphBatch
from immunostruct_amd.nn import EGNNConv, egnn_stack_forward
dev = torch.device("cuda:0")
raw = synthetic.make_batch(128, seed=1)
g = PackedGraphBatch.from_raw(raw, device=dev)
layers = [EGNNConv(20 if i == 0 else 64, 64, 64, 1).to(dev) for i in range(2)]
lib = _lib.load()
lib.is_debug_stamps.argtypes = [ctypes.c_void_p]; lib.is_debug_stamps.restype = ctypes.c_int
names = ["entry", "consts+rp+barrier", "S0 done", "SA done", "weights staged", "MM1 done", "MM2 done", "barrier", "SEG done(+barrier)"]
for rep in range(3):
    with torch.no_grad():
        egnn_stack_forward(layers, g, g.ndata["x"][:, :20], g.ndata["x"][:, 20:], g.edata["edge_attr"])
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 16)()
    assert lib.is_debug_stamps(ctypes.cast(buf, ctypes.c_void_p)) == 0
    t = list(buf)[:9]
    print("rep", rep, " ".join(f"{names[i]}:+{t[i]-t[i-1]}" for i in range(1, 9)), " total", t[8] - t[0], "(shader-clock cycles)")

if hasattr(lib, "is_debug_stamps3"):
    lib.is_debug_stamps3.argtypes = [ctypes.c_void_p]; lib.is_debug_stamps3.restype = ctypes.c_int
    for rep in range(3):
        with torch.no_grad():
            egnn_stack_forward(layers, g, g.ndata["x"][:, :20], g.ndata["x"][:, 20:], g.edata["edge_attr"])
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 64)()
        assert lib.is_debug_stamps3(ctypes.cast(buf, ctypes.c_void_p)) == 0
        t = list(buf)
        n = max(i for i in range(64) if t[i] > 0) + 1
        lab = ["S0", "SA", "MM1", "MM2", "SEG"]
        print("fwd3 rep", rep, f"prologue:+{t[1]-t[0]}", " ".join(f"{lab[(i-2)%5]}:+{t[i]-t[i-1]}" for i in range(2, n)), " total", t[n-1]-t[0])

if hasattr(lib, "is_debug_stamps_node"):
    lib.is_debug_stamps_node.argtypes = [ctypes.c_void_p]; lib.is_debug_stamps_node.restype = ctypes.c_int
    for rep in range(3):
        with torch.no_grad():
            egnn_stack_forward(layers, g, g.ndata["x"][:, :20], g.ndata["x"][:, 20:], g.edata["edge_attr"])
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16)()
        assert lib.is_debug_stamps_node(ctypes.cast(buf, ctypes.c_void_p)) == 0
        t = list(buf)[:8]
        nn_ = ["start", "weights+X issued", "barrier", "MM_a", "barrier", "MM_b", "barrier", "MM_c"]
        print("node_fwd rep", rep, " ".join(f"{nn_[i]}:+{t[i]-t[i-1]}" for i in range(1, 8)), " total", t[7] - t[0])

# ---- backward timeline (first tile of workgroup 300) ----
lib.is_debug_stamps_bwd.argtypes = [ctypes.c_void_p]; lib.is_debug_stamps_bwd.restype = ctypes.c_int
bn = ["tile start", "rp+barriers+z loads issued", "S0", "E3", "barrier1", "WG1+MM3", "barrier2", "dz2,SA,E1", "barrier3", "WG2+MM4", "barrier4", "SB+GEO(+barrier5)", "SEG(+barrier6)"]
for rep in range(3):
    for l in layers:
        l.zero_grad()
    h0 = g.ndata["x"][:, :20]; x0 = g.ndata["x"][:, 20:]
    hh, xx = egnn_stack_forward(layers, g, h0, x0, g.edata["edge_attr"])
    (hh.sum() + xx.sum()).backward()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 24)()
    assert lib.is_debug_stamps_bwd(ctypes.cast(buf, ctypes.c_void_p)) == 0
    t = list(buf)[:13]
    print("bwd rep", rep, " ".join(f"{bn[i]}:+{t[i]-t[i-1]}" for i in range(1, 13)), " total", t[12] - t[0])

if hasattr(lib, "is_debug_stamps_node"):
    for rep in range(2):
        for l in layers:
            l.zero_grad()
        hh, xx = egnn_stack_forward(layers, g, g.ndata["x"][:, :20], g.ndata["x"][:, 20:], g.edata["edge_attr"])
        (hh.sum() + xx.sum()).backward()
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16)()
        assert lib.is_debug_stamps_node(ctypes.cast(buf, ctypes.c_void_p)) == 0
        t = list(buf)
        print("node_bwd_data rep", rep, f"operands+stage issued:+{t[9]-t[8]} barrier:+{t[10]-t[9]} dh+da1 stages:+{t[11]-t[10]} dX:+{t[12]-t[11]}  total {t[12]-t[8]}")
