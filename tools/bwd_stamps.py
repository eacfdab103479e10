"""Stage time-stamps (s_memtime, 100 MHz constant clock x ... see scale below) of workgroup 300 of the LAST egnn_layer_bwd launch of a
6-layer stack backward at B = 128 (debug build with -DIS_STAGE_STAMPS in gpurun_dbg/)."""
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
layers = [EGNNConv(20 if i == 0 else 64, 64, 64, 1).to(dev) for i in range(6)]
lib = _lib.load()
from immunostruct_amd import functional as HF
read_stamps = lib.is_debug_stamps_bwd8 if HF.BWD_PAIRED else lib.is_debug_stamps_bwd      # IMMUNOSTRUCT_BWD_PAIRED=0: the 256-thread kernel's stamps
read_stamps.argtypes = [ctypes.c_void_p]; read_stamps.restype = ctypes.c_int
h0 = g.ndata["x"][:, :20].contiguous(); x0 = g.ndata["x"][:, 20:].contiguous(); ea = g.edata["edge_attr"]
bn = ["tile start", "rp+pdt staged, z loads issued", "S0", "E3", "barrier1", "WG1+MM3", "barrier2", "dz2,SA,E1", "barrier3", "WG2+MM4", "barrier4",
      "GEO(+barrier5)", "WG3+SEG(+barrier6)"]
for rep in range(4):
    for l in layers:
        l.zero_grad()
    h, x = egnn_stack_forward(layers, g, h0, x0, ea)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    (h.sum() + x.sum()).backward()
    ev1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 24)()
    assert read_stamps(ctypes.cast(buf, ctypes.c_void_p)) == 0
    t = list(buf)
    print(f"rep {rep}: backward of the stack {ev0.elapsed_time(ev1) * 1e3:.0f} us")
    print("  first tile:", " ".join(f"{bn[i]}:+{t[i] - t[i - 1]}" for i in range(1, 13)), " total", t[12] - t[0])
    if t[20]:
        print(f"  node phase front (debug waits): operand pack back +{t[20] - t[13]}, zn1 / g_h rows back +{t[21] - t[20]}, rowptr_src / g_psd / dx back"
              f" +{t[22] - t[21]}, gather rounds +{t[23] - t[22]}, stores + LDS + barrier +{t[14] - t[23]}")
    print(f"  kernel: entry->gather+stage barrier +{t[14] - t[13]}, dh MFMA +{t[15] - t[14]}, da1/dzn1 +{t[16] - t[15]}, dX + end of node phase +{t[17] - t[16]},"
          f" weights staged -> first tile +{t[0] - t[17]}, all tiles +{t[18] - t[0]}, record +{t[19] - t[18]}; whole {t[19] - t[13]} ticks")
