#!/usr/bin/env python
"""Throughput of the ImmunoStruct train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic peptide-MHC residue graphs already
resident in HBM: HybridModelv2 forward (6 EGNN layers -> node attention -> mean-pool, sequence VAE,
property MLP, fusion head) + regression loss + backward + Adam, i.e. the inner loop of
``procedures/train.py:18-29`` on BASELINE config 2 (B = 128 graphs of 190 padded nodes per GPU,
``--full-sequence --sequence-loss``).  N > 1: one process per GPU (launched by torch.distributed.run),
graphs sharded over ranks (weak scaling), one RCCL all-reduce of the flat gradient per step.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline"     -- the dominant HIP kernel (fused EGNN edge backward): algorithmic FLOP / launch divided
                    by its mean launch duration measured with HIP events inside the timed region,
                    against the fp32 MFMA peak; the HBM view of the same launches is included.
  "cpu_baseline" -- the CPU oracle (a port of the reference's un-fused PyTorch path) timed on this
                    box's host cores on a bounded sample of the same workload (N = 1, rank 0 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from immunostruct_amd import distributed as D  # noqa: E402
from immunostruct_amd import functional as HF  # noqa: E402
from immunostruct_amd import optim  # noqa: E402
from immunostruct_amd import synthetic  # noqa: E402
from immunostruct_amd.graph import PackedGraphBatch  # noqa: E402
from immunostruct_amd.models import model_map  # noqa: E402
from immunostruct_amd.utils import Losses  # noqa: E402

VAE_IN = synthetic.SEQ_LEN * synthetic.SEQ_ALPHABET
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak BW (spec)
H = 64


def edge_pass_algorithmic(n_nodes, n_edges, din, fe):
    """Algorithmic bytes / FLOP of one fused EGNN edge launch (DESIGN.md section "Roofline accounting").

    bytes follow SURVEY.md section 8(d); FLOP count only the per-edge dense layers the fused kernel
    actually needs after hoisting the first edge-MLP layer to node level (2 x 64x64 + 64 forward;
    backward = 2 data-grad + 2 weight-grad 64x64 products).
    """
    bytes_fwd = n_edges * (2 * din * 4 + 24 + fe * 4 + 4) + (n_nodes + 1) * 4 + n_nodes * (din * 4 + 12) + n_nodes * (H * 4 + 12)
    bytes_bwd = n_edges * (2 * din * 4 + 24 + fe * 4 + 4) + n_nodes * H * 4 + 2 * n_nodes * (din + 3) * 4
    flop_fwd = 2.0 * n_edges * (H * H + H * H + H)
    flop_bwd = 2.0 * n_edges * (4 * H * H + 2 * H)
    return bytes_fwd, bytes_bwd, flop_fwd, flop_bwd


def build_batches(n_batches, batch, deg_extra, device, seed0):
    out = []
    for i in range(n_batches):
        raw = synthetic.make_batch(batch, seed=seed0 + i, deg_extra=deg_extra)
        g = PackedGraphBatch.from_raw(raw, device=device)
        out.append(dict(g=g, seq=torch.from_numpy(raw.one_hot_sequence()).to(device),
                        prop=torch.from_numpy(raw.prop).to(device), y=torch.from_numpy(raw.y_reg).to(device),
                        raw=raw))
    return out


def cpu_baseline(batch, deg_extra, budget_s=24.0):
    """Oracle train step (forward + loss + backward + Adam) on the host cores; returns graphs/s.

    Two bounded samples (all hardware threads, and 16 threads: these small un-fused ops do not scale to
    hundreds of threads) -- the faster one is reported together with the thread count it used.
    """
    from oracle import functional_ref as FR
    from oracle import graph_ref
    raw = synthetic.make_batch(batch, seed=1, deg_extra=deg_extra)
    shapes = {k: tuple(v.shape) for k, v in model_map["HybridModelv2"](vae_input_dim=VAE_IN, device="cpu").state_dict().items()}
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synthetic.det_state_dict(shapes, seed=3).items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-3)
    g = graph_ref.RefGraph(raw.src, raw.dst, raw.num_nodes, raw.batch_num_nodes)
    g.ndata["x"], g.edata["edge_attr"] = torch.from_numpy(raw.x), torch.from_numpy(raw.edge_attr)
    seq, prop, y = torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop), torch.from_numpy(raw.y_reg)

    def step():
        opt.zero_grad()
        it = FR.forward("HybridModelv2", sd, g, seq, prop)
        loss = FR.regression_loss(it["recon_x"], seq, it["mu"], it["logvar"], it["final_output"], y, VAE_IN)
        loss.backward()
        opt.step()

    all_threads = torch.get_num_threads()
    best = None
    for threads in sorted({all_threads, min(16, all_threads)}, reverse=True):
        torch.set_num_threads(threads)
        step()  # warm-up
        t0, n = time.perf_counter(), 0
        while True:
            step()
            n += 1
            if time.perf_counter() - t0 > budget_s / 2 or n >= 40:
                break
        dt = time.perf_counter() - t0
        rate = batch * n / dt
        if best is None or rate > best[0]:
            best = (rate, threads, n, dt)
    torch.set_num_threads(all_threads)
    rate, threads, n, dt = best
    return dict(value=round(rate, 2), unit="graphs/s", cores=threads, kind="port",
                sample=f"{n} train steps of B={batch} (oracle/functional_ref.py HybridModelv2, fwd+loss+bwd+Adam, "
                       f"{threads} torch threads of {all_threads} available, {dt:.1f} s; best of the all-threads and 16-thread samples)")


def flush_c_stdio():
    """libraries (RCCL) print through libc's buffered stdout, which would otherwise be flushed at exit -- AFTER the one
    JSON line this script promises as its last output"""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def self_launch(n):
    """``python bench.py --gpus N`` without torchrun: start one child per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    in its environment, the contract torch.distributed.run would provide) BEFORE this process touches a GPU, let rank 0
    write the single JSON line to our stdout (the other ranks' stdout goes to stderr), and return non-zero when any
    rank fails -- the survivors are then stopped by PID, never by pattern."""
    import socket
    import subprocess
    have = torch.cuda.device_count()       # counting devices does not initialise the GPU runtime
    if have < n and os.environ.get("IMMUNOSTRUCT_FORCE_DEVICE") is None:
        print(f"bench.py: --gpus {n} requested but {have} GPU(s) visible", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(so.getsockname()[1])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else sys.stderr))
    rc, pending = 0, set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in sorted(pending):
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="graphs per GPU per step")
    ap.add_argument("--deg-extra", type=int, default=2, help="random contact edges per residue (E/N - 1)")
    ap.add_argument("--model", default="HybridModelv2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--force-pack", action="store_true", help="exercise the multi-rank gradient-bucket path on one GPU")
    ap.add_argument("--eager", action="store_true", help="launch every kernel eagerly instead of replaying the captured HIP graph")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain ``python bench.py --gpus N``: become the launcher (no GPU call has happened in this process)
        raise SystemExit(self_launch(args.gpus))
    rank, local_rank, world = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line printed must describe the run asked for")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    if world == 1 and os.environ.get("IMMUNOSTRUCT_FORCE_COLLECTIVE") == "1":
        # debugging aid: a one-rank RCCL group, so that --force-pack issues real (trivial) RCCL collectives on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(0)
        torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    if os.environ.get("IMMUNOSTRUCT_FORCE_DEVICE") is not None:   # debugging aid: several ranks on one GPU (gloo)
        local_rank = int(os.environ["IMMUNOSTRUCT_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    torch.manual_seed(1)
    model = model_map[args.model](vae_input_dim=VAE_IN, device=dev).to(dev)
    D.broadcast_parameters(model)
    model.train()
    reducer = D.FlatGradReducer(model.parameters(), world=world, always_pack=args.force_pack)
    if os.environ.get("IMMUNOSTRUCT_TORCH_ADAM", "0") == "1":
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=not args.eager)
    else:
        opt = optim.Adam(model.parameters(), lr=1e-3)      # csrc/optimizer.hip
    losses = Losses(VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    pool = build_batches(4, args.batch, args.deg_extra, dev, seed0=1000 * (rank + 1))

    def forward_loss(m, g, seq, prop, y):
        recon, mu, logvar, final = m(g, seq, prop)
        return losses.regression_loss(recon, seq, mu, logvar, final, y)

    def eager_step(i):
        b = pool[i % len(pool)]
        reducer.zero()
        loss = forward_loss(model, b["g"], b["seq"], b["prop"], b["y"])
        loss.backward()
        reducer.all_reduce_mean()
        opt.step()
        return loss.detach()

    if args.eager:
        step = eager_step
    else:
        from immunostruct_amd.engine import CapturedTrainStep
        b0 = pool[0]
        captured = CapturedTrainStep(model, opt, reducer, forward_loss, (b0["g"], b0["seq"], b0["prop"], b0["y"]),
                                     edge_capacity=max(b["raw"].num_edges for b in pool))

        def step(i):
            b = pool[i % len(pool)]
            return captured(b["g"], b["seq"], b["prop"], b["y"])

    for i in range(args.warmup):
        step(i)
    flush_c_stdio()       # RCCL's version banner sits in libc's stdout buffer: push it out now, not after the JSON line
    HF.KernelTimer.reset()
    HF.KernelTimer.enabled = args.eager and not args.no_kernel_timers

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    trace = os.environ.get("IMMUNOSTRUCT_STEP_TRACE")     # debugging aid: device time of every timed step (events)
    marks = []
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if trace:
            e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
        last = step(args.warmup + i)
    if trace:
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
    fence()
    dt = time.perf_counter() - t0
    if trace:
        print("[step trace] ms per step: " + " ".join(f"{a.elapsed_time(b):.3f}" for a, b in zip(marks, marks[1:])), file=sys.stderr)
        for blk in range(int(trace)):
            fence()
            b0 = time.perf_counter()
            for i in range(args.steps):
                step(args.warmup + i)
            fence()
            print(f"[step trace] extra block {blk}: {1e3 * (time.perf_counter() - b0) / args.steps:.3f} ms/step", file=sys.stderr)
    HF.KernelTimer.enabled = False
    if os.environ.get("IMMUNOSTRUCT_HOST_TIMES"):     # debugging aid: host-side cost of one isolated step vs its GPU time
        for i in range(5):
            fence()
            h0 = time.perf_counter()
            step(args.warmup + args.steps + i)
            h1 = time.perf_counter()
            fence()
            h2 = time.perf_counter()
            print(f"[host times] step call returned after {1e6 * (h1 - h0):.0f} us, GPU done after {1e6 * (h2 - h0):.0f} us", file=sys.stderr)
    if HF.Stamps.enabled and HF.Stamps.buf is not None:
        fence()
        for t_us, name in HF.Stamps.report():
            print(f"[stamps] {t_us:9.1f} us  {name}", file=sys.stderr)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(last.detach())
    timers_mode = "HIP events around each launch inside the timed region (eager launches)"
    if not args.eager and not args.no_kernel_timers:
        # the timed region replays a captured HIP graph (individual launches cannot be bracketed there):
        # measure per-kernel durations with HIP events on an eager re-run of the same steps
        # (sequence branch on the SAME stream for this re-run: a kernel's duration is then its own, not the co-run slow-down
        #  by whatever the forked branch happens to execute next to it; the in-situ averages are in profiles/)
        from immunostruct_amd.models import _core as model_core
        overlap_saved, model_core.OVERLAP_BRANCHES = model_core.OVERLAP_BRANCHES, False
        for bucket in reducer.buckets:      # the captured graphs are done: pack from the live .grad tensors again
            bucket["sources"] = None
        for i in range(2):      # untimed: the first eager steps after the replays grow the allocator's pool (hipMalloc stalls)
            eager_step(args.warmup + i)
        torch.cuda.synchronize()
        HF.KernelTimer.reset()
        HF.KernelTimer.enabled = True
        for i in range(min(args.steps, 10)):
            eager_step(args.warmup + i)
        torch.cuda.synchronize()
        HF.KernelTimer.enabled = False
        model_core.OVERLAP_BRANCHES = overlap_saved
        timers_mode = ("HIP events around each launch, eager single-stream re-run of the timed steps "
                       "(the timed region replays a HIP graph)")

    if rank == 0:
        graphs = args.batch * world * args.steps
        n_nodes = pool[0]["raw"].num_nodes
        n_edges = float(np.mean([b["raw"].num_edges for b in pool]))
        timers = HF.KernelTimer.summary()
        roof = None
        if "egnn_edge_bwd" in timers:
            # the timed launches are the five full ones per step (layer 0: Din = 20, layers 1-4: Din = 64); the last layer's
            # launch skips the coordinate MLP (its output is unused by the model) and is timed under its own name
            per = [edge_pass_algorithmic(n_nodes, n_edges, din, 1) for din in (20, 64, 64, 64, 64)]
            b_fwd, b_bwd = np.mean([p[0] for p in per]), np.mean([p[1] for p in per])
            f_fwd, f_bwd = np.mean([p[2] for p in per]), np.mean([p[3] for p in per])
            n_b, ms_b = timers["egnn_edge_bwd"]
            n_f, ms_f = timers["egnn_edge_fwd"]
            tf_b, tf_f = f_bwd / (ms_b * 1e-3) / 1e12, f_fwd / (ms_f * 1e-3) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if os.path.isfile(tpath) and args.batch == 128 and args.deg_extra == 2:
                # PMC-measured HBM bytes per launch of this kernel on this workload (profiles/README.md)
                traffic = json.load(open(tpath)).get("egnn_edge_bwd16_kernel", {}).get("bytes")
            roof = dict(kernel="egnn_edge_bwd16_kernel", bound="mfma", achieved=round(tf_b, 2),
                        peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(tf_b / PEAK_FP32_MFMA_TFLOPS, 4),
                        traffic=traffic, measured=timers_mode, launches=n_b, mean_launch_us=round(ms_b * 1e3, 2),
                        algorithmic_flop_per_launch=f_bwd, algorithmic_bytes_per_launch=b_bwd,
                        hbm_view=dict(achieved=round(b_bwd / (ms_b * 1e-3) / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                                      frac=round(b_bwd / (ms_b * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)),
                        forward_kernel=dict(kernel="egnn_edge_fwd3_kernel", mean_launch_us=round(ms_f * 1e3, 2), launches=n_f,
                                            tflops=round(tf_f, 2), frac_mfma=round(tf_f / PEAK_FP32_MFMA_TFLOPS, 4),
                                            hbm_gbs=round(b_fwd / (ms_f * 1e-3) / 1e9, 1),
                                            frac_hbm=round(b_fwd / (ms_f * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)))
            if "gather_segment_sum" in timers:
                n_g, ms_g = timers["gather_segment_sum"]
                gbytes = n_edges * (H * 4 + 12 + 4) + n_nodes * (H * 4 + 12 + 4)
                roof["gather_kernel"] = dict(kernel="gather_segment_sum_kernel", bound="hbm", mean_launch_us=round(ms_g * 1e3, 2),
                                             launches=n_g, achieved=round(gbytes / (ms_g * 1e-3) / 1e9, 1), unit="GB/s",
                                             frac=round(gbytes / (ms_g * 1e-3) / 1e9 / PEAK_HBM_GBS, 4))
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.batch, args.deg_extra)
        line = dict(metric="peptide-MHC graphs/sec (train step)", value=round(graphs / dt, 1), unit="graphs/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload=f"IEDB pretrain step (BASELINE config 2): {args.model}, full-sequence + sequence-loss, "
                                         f"regression loss, Adam; B={args.batch} graphs/GPU x 190 padded nodes, "
                                         f"E~{int(n_edges)} edges/batch (deg_extra={args.deg_extra}), Fe=1",
                                global_batch=args.batch * world, nodes_per_batch=n_nodes, edges_per_batch=int(n_edges),
                                parallelism=f"dp{world}", final_loss=round(final_loss, 5),
                                rccl_ranks=(torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1),
                                dist_backend=(torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
                                launch="eager" if args.eager else "hipGraph replay",
                                grad_allreduce=(None if (args.eager or not reducer.packing) else
                                                dict(form="two-stage backward, bucket 0 overlapped" if captured.two_stage else "serial",
                                                     buckets=[int(b["flat"].numel()) for b in reducer.buckets],
                                                     tuned_ms=captured.dp_times))),
                    roofline=roof, cpu_baseline=cpu,
                    kernel_timers_us={k: [v[0], round(v[1] * 1e3, 2)] for k, v in timers.items()})
        flush_c_stdio()
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
