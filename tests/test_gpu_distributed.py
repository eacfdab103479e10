"""Multi-rank paths on the one GPU of the test box: two processes (torch.distributed.run, gloo backend) share cuda:0.
RCCL needs one GPU per rank, so the ``nccl`` backend itself is exercised as a ONE-RANK group (the ``*_rccl_*`` tests below:
init, asynchronous all-reduce between captured graphs, wait, timing, form selection); more ranks are the driver's 8-GPU run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script_args, port, ranks=2):
    env = dict(os.environ, IMMUNOSTRUCT_DIST_BACKEND="gloo", IMMUNOSTRUCT_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.gpu
def test_data_parallel_training_loop_two_ranks(cuda_device):
    """procedures.train_model_device under torch.distributed (SURVEY.md 8 e): rank 0's weights are broadcast, every rank
    trains on its shard of each epoch's permutation through the captured step with the gradient all-reduce, the ranks'
    parameters stay bit-identical, only rank 0 writes the checkpoint (tools/dp_train_check.py asserts all of it)."""
    res = _run([os.path.join("tools", "dp_train_check.py")], 29551)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "DP TRAIN CHECK OK" in res.stdout


@pytest.mark.gpu
def test_bench_two_ranks_prints_one_json_line(cuda_device):
    """bench.py as the driver launches it for N = 2: rank 0 prints exactly one JSON line, last, with the whole-job value."""
    import json
    res = _run(["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-kernel-timers"], 29552)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 256 and line["scaling"] == "weak"
    assert line["config"]["grad_allreduce"]["form"] in ("serial", "two-stage backward, bucket 0 overlapped")
    assert sum(1 for ln in lines if ln.lstrip().startswith("{")) == 1


@pytest.mark.gpu
def test_entry_script_two_ranks(cuda_device, tmp_path):
    """train_IEDB_wFT under torch.distributed.run with --device-dataset: pretrain -> new head -> finetune -> inference on
    two ranks; rank 0 writes both checkpoints, every rank loads them after the barrier."""
    res = _run(["-m", "immunostruct_amd.train_IEDB_wFT", "--model", "HybridModelv2", "--full-sequence", "--sequence-loss",
                "--num-epochs", "1", "--learning-rate-pretrain", "1e-4", "--batch-size", "16", "--synthetic", "160", "--device-dataset", "--seed", "3",
                "--model-save-dir", str(tmp_path)], 29553)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("DONE FINE TUNING") == 2 and len(list(tmp_path.glob("*.pt"))) == 2


@pytest.mark.gpu
def test_entry_scripts_two_ranks_self_supervision_and_pairs(cuda_device, tmp_path):
    """the data-parallel form of the remaining device loops: train_IEDB_wFT --self-supervision and
    train_Cancer_wFT (paired, self-supervised) under torch.distributed.run on two ranks"""
    common = ["--full-sequence", "--sequence-loss", "--num-epochs", "1", "--learning-rate-pretrain", "1e-4", "--batch-size", "16", "--synthetic", "160", "--device-dataset",
              "--seed", "3", "--self-supervision", "--model-save-dir", str(tmp_path)]
    res = _run(["-m", "immunostruct_amd.train_IEDB_wFT", "--model", "HybridModelv2_SSL"] + common, 29556)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("DONE FINE TUNING") == 2
    res = _run(["-m", "immunostruct_amd.train_Cancer_wFT", "--model", "HybridModelv2_Comparative_SSL", "--use-wt-for-downstream",
                "--coeff-contrastive", "0.01", "--min-finetuning-batches", "2"] + common, 29557)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert len(list(tmp_path.glob("*_finetune.pt"))) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["split", "tail"])
def test_data_parallel_parity_on_the_hip_model(cuda_device, mode):
    """SURVEY.md section 4 (iv) on the real model (tools/dp_parity_check.py): ``split`` -- 2 ranks x B/2 through the captured
    two-stage step (all-reduce of bucket 0 under the stack backward, 1/world inside Adam) == 1 rank x B eager, same
    weights after 3 Adam steps; ``tail`` -- train_model_device with a shard that is not a multiple of the batch size
    (replays + one eager trailing step that must pack its OWN gradients) == an all-eager data-parallel loop."""
    res = _run([os.path.join("tools", "dp_parity_check.py"), mode], 29554 + (mode == "tail"))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "DP PARITY OK" in res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["split", "split-auto"])
def test_data_parallel_parity_with_four_ranks(cuda_device, mode):
    """Beyond world = 2 without new hardware: FOUR gloo ranks sharing cuda:0, B / 4 graphs each, through the
    captured data-parallel step -- forced two-stage (``split``) and chosen by timing (``split-auto``: every candidate replayed with
    the collectives, the maximum over the ranks decides).  Asserted inside tools/dp_parity_check.py: the chosen form gathered from
    all ranks is ONE form, the ranks' parameters after 3 Adam steps are bit-identical, and they equal 1 rank x B eager."""
    res = _run([os.path.join("tools", "dp_parity_check.py"), mode], 29558 + (mode == "split-auto"), ranks=4)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "DP PARITY OK" in res.stdout and res.stdout.count("split parity (4 ranks") == 4


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(cuda_device):
    """``python bench.py --gpus 2`` WITHOUT torchrun (how the driver starts N = 1): the script becomes the launcher, one
    child per rank, rank 0's single JSON line on stdout, n_gpus == the number asked for and == the process group's size."""
    import json
    env = dict(os.environ, IMMUNOSTRUCT_DIST_BACKEND="gloo", IMMUNOSTRUCT_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-kernel-timers"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["global_batch"] == 256


def _rccl_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "IMMUNOSTRUCT_DIST_BACKEND",
                                                             "IMMUNOSTRUCT_FORCE_DEVICE", "IMMUNOSTRUCT_DP_OVERLAP")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


def _bench_line(env, *flags):
    import json
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-e2e",
                          "--no-kernel-timers", "--no-copy-ceiling"] + list(flags), cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_under_a_one_rank_rccl_group(cuda_device):
    """``IMMUNOSTRUCT_FORCE_COLLECTIVE=1 python bench.py --gpus 1 --force-pack``: the data-parallel step (gradient buckets,
    RCCL all-reduce per bucket between the captured graphs, 1/world inside Adam) in a fresh child under a real ``nccl`` process
    group; the same seeded steps without the group end at the same loss."""
    # the default (round 6): the multi-graph forms only -- the one-graph forms are opt-in until a run with >= 2 ranks has seen them
    line = _bench_line(_rccl_env(IMMUNOSTRUCT_FORCE_COLLECTIVE="1", MASTER_PORT="29570"), "--force-pack")
    ar = line["config"]["grad_allreduce"]
    assert ar["form"] in ("serial", "two-stage backward, bucket 0 overlapped") and not ar["tuned_ms"]["one_graph_ms"]
    assert line["config"]["composed_paths"] == {}      # every operator of the timed step ran on the hand-written kernels
    # IMMUNOSTRUCT_DP_ONE_GRAPH=auto: every form captured, timed on all ranks, the fastest kept
    line = _bench_line(_rccl_env(IMMUNOSTRUCT_FORCE_COLLECTIVE="1", MASTER_PORT="29571", IMMUNOSTRUCT_DP_ONE_GRAPH="auto"), "--force-pack")
    cfg = line["config"]
    assert cfg["dist_backend"] == "nccl" and cfg["rccl_ranks"] == 1 and line["n_gpus"] == 1
    ar = cfg["grad_allreduce"]
    assert ar["form"] in ("serial", "two-stage backward, bucket 0 overlapped", "one graph: serial, collectives captured",
                          "one graph: two-stage backward, bucket 0 overlapped, collectives captured")
    assert ar["one_graph_error"] is None, ar["one_graph_error"]      # RCCL's collectives were captured into the step's graph
    assert ar["tuned_ms"] is not None and ar["tuned_ms"]["serial_ms"] > 0            # auto: every candidate was replayed with the collectives
    assert set(ar["tuned_ms"]["one_graph_ms"]) == {"graph", "graph2"} and all(v > 0 for v in ar["tuned_ms"]["one_graph_ms"].values())
    assert ar["standalone_allreduce"] and all(b["ms"] > 0 for b in ar["standalone_allreduce"])      # time_all_reduce under nccl
    assert sum(ar["buckets"]) == sum(b["floats"] for b in ar["standalone_allreduce"])
    # same loss as without the group: a fixed number of steps on both sides (no adaptive settle blocks, no form timing -- its
    # replays advance the random streams), so that both runs draw the same dropout masks and noise
    fixed = dict(IMMUNOSTRUCT_BENCH_SETTLE_BLOCKS="2", IMMUNOSTRUCT_DP_OVERLAP="0", IMMUNOSTRUCT_DP_ONE_GRAPH="1")
    packed = _bench_line(_rccl_env(IMMUNOSTRUCT_FORCE_COLLECTIVE="1", MASTER_PORT="29574", **fixed), "--force-pack")
    assert packed["config"]["dist_backend"] == "nccl" and packed["config"]["grad_allreduce"]["form"] == "one graph: serial, collectives captured"
    plain = _bench_line(_rccl_env(**fixed))
    assert plain["config"]["dist_backend"] is None and plain["config"]["grad_allreduce"] is None
    a, b = packed["config"]["final_loss"], plain["config"]["final_loss"]
    assert abs(a - b) <= 1e-6 * abs(b), (a, b)


@pytest.mark.gpu
def test_captured_data_parallel_forms_under_a_one_rank_rccl_group(cuda_device):
    """tools/dp_rccl1_check.py: the two-stage captured step (asynchronous all-reduce of bucket 0 under the stack backward,
    ``work.wait()`` honoured in front of the per-bucket update), the serial one and the auto-selected one under a 1-rank
    ``nccl`` group == the eager unpacked step after 3 Adam steps; collectives and waits per step counted; ``time_all_reduce``
    runs under nccl."""
    res = subprocess.run([sys.executable, os.path.join("tools", "dp_rccl1_check.py")], cwd=ROOT,
                         env=_rccl_env(IMMUNOSTRUCT_FORCE_COLLECTIVE="1", MASTER_PORT="29572"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-4000:]
    assert "RCCL 1-RANK CHECK OK" in res.stdout
    assert res.stdout.count("form two-stage") >= 1 and res.stdout.count("form serial") >= 1


@pytest.mark.gpu
def test_training_loop_under_a_one_rank_rccl_group(cuda_device, tmp_path):
    """the entry script's device loop (procedures.train_model_device: broadcast of the weights, captured data-parallel step,
    eager trailing batch, rank-0 checkpoint, barrier) with every collective going through RCCL"""
    res = subprocess.run([sys.executable, "-m", "immunostruct_amd.train_IEDB_wFT", "--model", "HybridModelv2", "--full-sequence",
                          "--sequence-loss", "--num-epochs", "1", "--learning-rate-pretrain", "1e-4", "--batch-size", "16", "--synthetic", "100",
                          "--device-dataset", "--seed", "3", "--model-save-dir", str(tmp_path)], cwd=ROOT,
                         env=_rccl_env(IMMUNOSTRUCT_FORCE_COLLECTIVE="1", MASTER_PORT="29573"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("DONE FINE TUNING") == 1 and len(list(tmp_path.glob("*.pt"))) == 2


def test_bench_refuses_more_gpus_than_visible():
    """never a line with n_gpus != the number requested: with no (or too few) GPUs the launcher exits non-zero"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "IMMUNOSTRUCT_FORCE_DEVICE")}
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "64", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "{" not in res.stdout


@pytest.mark.gpu
def test_gradients_land_in_the_flat_bucket_without_copies(cuda_device):
    """Data-parallel steps pack the gradients into flat buckets (``distributed.FlatGradReducer``).  The producers of 98.7 % of the
    bytes write there directly: the two VAE matrices (``LinearSmallBatchFn`` / ``VaeLatentFn``) and the EGNN stack's 66 tensors (the
    reduction of the partial records, whose per-layer layout IS the bucket's).  After a backward every such ``.grad`` must ALIAS
    its slice -- a gradient handed to autograd through the reducer's own view object would be cloned by AccumulateGrad (12 MB
    per matrix) and copied back by the pack -- and the packed bucket must equal the gradients of a run without a reducer."""
    import numpy as np
    import torch
    from immunostruct_amd import synthetic
    from immunostruct_amd.distributed import FlatGradReducer
    from immunostruct_amd.models import model_map
    from immunostruct_amd.utils import Losses
    from . import helpers as H
    dev = cuda_device
    raw = synthetic.make_batch(6, seed=91, deg_extra=2)
    sd = H.det_sd(H.model_shapes("HybridModelv2"), seed=4)
    seq, prop, y = (torch.from_numpy(a).to(dev) for a in (raw.one_hot_sequence(), raw.prop, raw.y_reg))
    eps = H.make_eps(3, 6).to(dev)
    losses = Losses(H.VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)

    def run(with_reducer):
        import unittest.mock as mock
        model = model_map["HybridModelv2"](vae_input_dim=H.VAE_IN, device=dev).to(dev)
        model.load_state_dict(sd)
        model.eval()
        red = FlatGradReducer(model.parameters(), world=1, always_pack=True) if with_reducer else None
        if red is not None:
            red.zero()
        with mock.patch("torch.randn_like", lambda t: eps.to(t.dtype)):
            recon, mu, logvar, final = model(H.product_graph(raw, dev), seq, prop)
        losses.regression_loss(recon, seq, mu, logvar, final, y).backward()
        named = dict(model.named_parameters())
        if red is not None:
            aliased = [k for k, p in named.items() if p.grad is not None and p.grad.data_ptr() == p._grad_dest.data_ptr()]
            assert "vae_fc1.weight" in aliased and "vae_fc4.weight" in aliased
            stack = [k for k in named if k.startswith("GCN_layers.") and named[k].grad is not None]
            assert stack and all(k in aliased for k in stack), sorted(set(stack) - set(aliased))
            in_place = sum(named[k].numel() for k in aliased)
            assert in_place >= 0.98 * sum(p.numel() for p in named.values() if p.grad is not None)
            red.all_reduce_mean()
            torch.cuda.synchronize()
            return {k: v.clone() for k, v in zip([k for k, p in named.items() if p.requires_grad], red.buckets[0]["views"])}
        return {k: p.grad.clone() if p.grad is not None else torch.zeros_like(p) for k, p in named.items() if p.requires_grad}

    plain, packed = run(False), run(True)
    assert plain.keys() == packed.keys()
    for k in plain:
        assert torch.equal(plain[k], packed[k]), k
