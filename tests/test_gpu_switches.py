"""Every ``IMMUNOSTRUCT_*`` switch the product path still reads (round 6: 13, from 35), once per value, on the GPU: a switch selects
another kernel or another schedule for the SAME arithmetic, so three captured train steps under it must end where the default ends
(``tools/switch_probe.py``: HybridModelv2 -- one-head node attention, fused head, sequence branch -- and StructureModel -- the
8-head attention of the reference's default command line).  The switches read at import time, so every value is a fresh child.

  LIB (the default library by its explicit path) . BWD_PAIRED=0 (the 256-thread backward layer kernel) . SAVE_Z3=0 (z3 recomputed by
  the backward) . ATTN_TILES=32 (32-row attention blocks) . FORK_AFTER_LAYER=1 / 4 (where the sequence branch forks) . STAMPS=1
  (device time stamps inside the captured step) . under a ONE-RANK RCCL group
  (FORCE_COLLECTIVE=1): DP_OVERLAP=0 / 1 / auto, DP_ONE_GRAPH=1 / auto, DP_RESERVED_CUS=0,8, DIST_MAX_NCHANNELS=4 .
  DIST_BACKEND / FORCE_DEVICE: tests/test_gpu_distributed.py (gloo ranks sharing the GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP_OUT = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")


def probe(**env_extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("IMMUNOSTRUCT_") and k not in KEEP_OUT}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, **env_extra)
    res = subprocess.run([sys.executable, os.path.join("tools", "switch_probe.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("SWITCH_PROBE ")]
    assert len(line) == 1, res.stdout[-2000:]
    return json.loads(line[0][len("SWITCH_PROBE "):])


@pytest.fixture(scope="module")
def default(cuda_device):
    return probe()


def assert_same_run(got, want, what, rtol=2e-5):
    for model in ("hybrid", "structure"):
        for a, b in zip(got[model]["losses"], want[model]["losses"]):
            assert abs(a - b) <= 1e-5 * abs(b), (what, model, got[model]["losses"], want[model]["losses"])
        for k, (s, sa) in want[model]["digest"].items():
            if k.endswith("w_k.bias"):
                # a bias of the attention KEYS shifts every score of a query alike: its gradient is zero in exact arithmetic, pure
                # round-off in fp32, and Adam turns round-off of either sign into a step of the full learning rate
                continue
            gs, gsa = got[model]["digest"][k]
            # sum and absolute sum of every tensor after three Adam steps (lr 1e-3: a differing update would show at 1e-3 of |.|)
            assert abs(gsa - sa) <= rtol * sa + 1e-12 and abs(gs - s) <= rtol * sa + 1e-12, (what, model, k, (gs, gsa), (s, sa))


def test_default_run_is_reproducible(default):
    again = probe()
    assert again["hybrid"]["losses"] == default["hybrid"]["losses"] and again["hybrid"]["digest"] == default["hybrid"]["digest"]
    assert default["hybrid"]["form"] == "serial" and default["stamps"] is None
    assert len(set(default["hybrid"]["losses"])) == 3


@pytest.mark.parametrize("switch", ["IMMUNOSTRUCT_BWD_PAIRED=0", "IMMUNOSTRUCT_SAVE_Z3=0", "IMMUNOSTRUCT_ATTN_TILES=32",
                                    "IMMUNOSTRUCT_FORK_AFTER_LAYER=1", "IMMUNOSTRUCT_FORK_AFTER_LAYER=4", "IMMUNOSTRUCT_LIB=default"])
def test_kernel_and_schedule_switches_keep_the_arithmetic(default, switch):
    k, v = switch.split("=")
    if k == "IMMUNOSTRUCT_LIB":
        v = os.path.join(ROOT, "immunostruct_amd", "csrc", "libimmunostruct_hip.so")
    assert_same_run(probe(**{k: v}), default, switch)


def test_device_time_stamps_leave_their_trace_and_nothing_else(default):
    got = probe(IMMUNOSTRUCT_STAMPS="1")
    assert_same_run(got, default, "STAMPS=1")
    assert got["stamps"] and "step start" in got["stamps"] and "optimizer done" in got["stamps"]


@pytest.mark.parametrize("switches", [dict(IMMUNOSTRUCT_DP_OVERLAP="0"), dict(IMMUNOSTRUCT_DP_OVERLAP="1"),
                                      dict(IMMUNOSTRUCT_DP_OVERLAP="auto", IMMUNOSTRUCT_DP_RESERVED_CUS="0,8", IMMUNOSTRUCT_DIST_MAX_NCHANNELS="4"),
                                      dict(IMMUNOSTRUCT_DP_OVERLAP="1", IMMUNOSTRUCT_DP_ONE_GRAPH="1"),
                                      dict(IMMUNOSTRUCT_DP_OVERLAP="auto", IMMUNOSTRUCT_DP_ONE_GRAPH="auto")])
def test_data_parallel_switches_under_a_one_rank_rccl_group(default, switches):
    """with one rank the reduced gradient is the local one: every form of the data-parallel step ends where the single-GPU step ends"""
    port = 29600 + sum(map(ord, json.dumps(switches, sort_keys=True))) % 200
    got = probe(IMMUNOSTRUCT_FORCE_COLLECTIVE="1", MASTER_PORT=str(port), **switches)
    assert_same_run(got, default, str(switches))
    form = got["hybrid"]["form"]
    if switches.get("IMMUNOSTRUCT_DP_ONE_GRAPH") == "1":
        assert form == "graph2"
    elif switches["IMMUNOSTRUCT_DP_OVERLAP"] == "0":
        assert form == "serial"
    elif switches["IMMUNOSTRUCT_DP_OVERLAP"] == "1":
        assert form == "two-stage"
    if "IMMUNOSTRUCT_DP_RESERVED_CUS" in switches:
        assert set(got["hybrid"]["tuned"]["two_stage_ms_by_reserved_cus"]) == {"0", "8"} and got["nccl_max_nchannels"] == "4"
    if switches.get("IMMUNOSTRUCT_DP_ONE_GRAPH") == "auto":
        assert set(got["hybrid"]["tuned"]["one_graph_ms"]) == {"graph", "graph2"}
