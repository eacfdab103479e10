"""the reference's OWN ``HybridModelv2`` / ``HybridModelv2_Comparative`` (imported unchanged from
/root/reference under ``oracle/shims.py``; only ``dgl.nn.EGNNConv`` / ``dgl.batch`` / PyG pooling are the oracle's
restatement) timed in the build container: forward + loss + backward + optimizer step, median of ``--steps`` steps after
``--warmup`` warm-ups, for torch thread counts {all cores, 1}, B in {16, 128}, ``deg_extra`` in {1, 2, 5, 8}.

    python tests/tools/cpu_reference_timing.py [--steps 20] [--warmup 5] [--quick] [--write]

``--write`` replaces the table between the ``<!-- cpu_reference_timing:begin/end -->`` markers of BASELINE.md.
Test infrastructure: needs /root/reference, never runs on the GPU box, never imported by the product.
"""
import argparse
import os
import platform
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from immunostruct_amd import synthetic  # noqa: E402
from oracle import graph_ref, shims  # noqa: E402

VAE_IN = synthetic.SEQ_LEN * synthetic.SEQ_ALPHABET


def ref_graph(raw):
    g = graph_ref.RefGraph(raw.src, raw.dst, raw.num_nodes, raw.batch_num_nodes)
    g.ndata["x"], g.edata["edge_attr"] = torch.from_numpy(raw.x), torch.from_numpy(raw.edge_attr)
    return g


def time_steps(step, steps, warmup):
    for _ in range(warmup):
        step()
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def single(model_map, Losses, b, deg):
    raw = synthetic.make_batch(b, seed=1, deg_extra=deg)
    torch.manual_seed(1)
    model = model_map["HybridModelv2"](vae_input_dim=VAE_IN, device="cpu")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)          # train_IEDB_wFT.py:69-74
    losses = Losses(VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    g, seq, prop, y = ref_graph(raw), torch.from_numpy(raw.one_hot_sequence()), torch.from_numpy(raw.prop), torch.from_numpy(raw.y_reg)

    def step():          # procedures/train.py:18-29
        opt.zero_grad()
        recon, mu, logvar, final = model(g, seq, prop)
        loss = losses.regression_loss(recon, seq, mu, logvar, final, y)
        loss.backward()
        opt.step()
        loss.item()
    return step, raw.num_edges, b


def paired(model_map, Losses, PCL, b, deg):
    rc, rw = synthetic.make_batch(b, seed=11, deg_extra=deg), synthetic.make_batch(b, seed=51, deg_extra=deg)
    torch.manual_seed(1)
    model = model_map["HybridModelv2_Comparative"](vae_input_dim=VAE_IN, device="cpu", use_wt_for_downstream=True)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-6)      # train_Cancer_wFT.py:76-92
    losses = Losses(VAE_IN, {0: 81.0, 1: 19.0}, sequence=True)
    pcl = PCL(embedding_dim=104)
    gs = (ref_graph(rc), ref_graph(rw))
    seqs = (torch.from_numpy(rc.one_hot_sequence()), torch.from_numpy(rw.one_hot_sequence()))
    props = (torch.from_numpy(rc.prop), torch.from_numpy(rw.prop))
    y = torch.from_numpy(rc.y_bin)

    def step():          # procedures/train.py:84-123
        opt.zero_grad()
        emb, rec, mu, lv, final = model.forward_comparative(gs, seqs, props)
        lc = losses.BCE_loss(rec[0], seqs[0], mu[0], lv[0], final, y)
        lw = losses.BCE_loss(rec[1], seqs[1], mu[1], lv[1], final, y)
        loss = (lc + lw) / 2 + 0.01 * pcl(emb[0], emb[1], y)
        loss.backward()
        opt.step()
        loss.item()
    return step, rc.num_edges + rw.num_edges, 2 * b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--quick", action="store_true", help="B = 16 only, deg_extra = 2 only (smoke run)")
    ap.add_argument("--write", action="store_true", help="write the table into BASELINE.md")
    args = ap.parse_args()
    model_map, Losses, PCL = shims.load_reference()
    cores = os.cpu_count()
    rows = []
    batches = (16,) if args.quick else (16, 128)
    degs = (2,) if args.quick else (1, 2, 5, 8)
    for kind in ("HybridModelv2", "HybridModelv2_Comparative"):
        for b in batches:
            for deg in degs:
                if kind == "HybridModelv2":
                    step, edges, graphs = single(model_map, Losses, b, deg)
                else:
                    step, edges, graphs = paired(model_map, Losses, PCL, b, deg)
                cells = []
                for threads in (cores, 1):
                    torch.set_num_threads(threads)
                    # the one-thread column of the big batches takes seconds per step: fewer samples there
                    n = args.steps if (threads > 1 or b <= 16) else max(3, args.steps // 5)
                    med = time_steps(step, n, args.warmup if threads > 1 else 1)
                    cells.append((med, graphs / med, n))
                rows.append((kind, b, deg, edges, cells))
                print(f"{kind} B={b} deg_extra={deg} E={edges}: " +
                      ", ".join(f"{t} thr {c[0]:.3f} s/step = {c[1]:.0f} graphs/s (median of {c[2]})" for t, c in zip((cores, 1), cells)), flush=True)
    torch.set_num_threads(cores)
    cpu = platform.processor() or ""
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        pass
    lines = [f"Measured by `tests/tools/cpu_reference_timing.py` (steps {args.steps}, warm-up {args.warmup}) in the build container: "
             f"{cores} x {cpu}, torch {torch.__version__} CPU; the reference's own model / loss / contrastive classes under "
             f"`oracle/shims.py`, forward + loss + backward + optimizer + `.item()`; graphs/s counts both members of a pair.", "",
             f"| model | B | deg_extra | edges / step | s/step ({cores} threads) | graphs/s ({cores} threads) | s/step (1 thread) | graphs/s (1 thread) |",
             "|---|---|---|---|---|---|---|---|"]
    for kind, b, deg, edges, cells in rows:
        lines.append(f"| {kind} | {b} | {deg} | {edges} | {cells[0][0]:.3f} | {cells[0][1]:.0f} | {cells[1][0]:.3f} | {cells[1][1]:.0f} |")
    table = "\n".join(lines)
    print("\n" + table)
    if args.write:
        path = os.path.join(ROOT, "BASELINE.md")
        text = open(path).read()
        begin, end = "<!-- cpu_reference_timing:begin -->", "<!-- cpu_reference_timing:end -->"
        if begin not in text:
            raise SystemExit("BASELINE.md has no cpu_reference_timing markers")
        text = text[:text.index(begin) + len(begin)] + "\n" + table + "\n" + text[text.index(end):]
        open(path, "w").write(text)


if __name__ == "__main__":
    main()
