"""Golden vectors for the inference metrics and the learning-rate schedule, FROM THE REFERENCE ITSELF.

Run in the build container only (needs ``/root/reference``):   python -m oracle.make_golden_metrics

The reference's ``procedures/metric.py`` and ``utils/scheduler.py`` are loaded by file path, unchanged, and run on
seeded inputs; inputs and outputs are stored in ``tests/golden/metrics.npz`` (data only).  Test infrastructure.
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/immunostruct"
KEYS = ["optimal_threshold", "accuracy", "accuracy_op", "f1", "f1_op", "precision", "precision_op", "recall", "recall_op",
        "roc_auc", "pr_auc", "ppvn", "ppvn_op", "ppv30", "ppv30_op"]
SCHEDULES = [(10, 40, 1e-4, 1e-6), (2, 8, 3e-4, 3e-6), (25, 100, 1e-3, 0.0)]     # warmup, max, base lr, warmup start lr


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def metric_cases():
    rs = np.random.RandomState(7)
    cases = []
    for n, pos, sharp, ties in ((200, 0.19, 2.0, False), (64, 0.5, 0.5, False), (500, 0.1, 4.0, True), (40, 0.3, 1.0, True)):
        y = (rs.rand(n) < pos).astype(np.float32)
        logit = sharp * (y - 0.5) + rs.normal(size=n)
        p = 1.0 / (1.0 + np.exp(-logit))
        if ties:
            p = np.round(p, 1 if n < 100 else 2)
        cases.append((y, p.astype(np.float64)))
    return cases


def main():
    metric = load(os.path.join(REF, "procedures", "metric.py"), "ref_metric")
    sched = load(os.path.join(REF, "utils", "scheduler.py"), "ref_scheduler")
    out = {}
    for i, (y, p) in enumerate(metric_cases()):
        thr = metric.find_optimal_threshold(y, p)
        with contextlib.redirect_stdout(io.StringIO()):
            res = metric.evaluate_metrics(y, p, thr)
        out[f"m{i}_y"], out[f"m{i}_p"] = y, p
        out[f"m{i}_out"] = np.array([float(res[k]) for k in KEYS], dtype=np.float64)
        out[f"m{i}_ppvn_scores"] = np.array([metric.mean_PPVn(y, p), metric.mean_PPVn(y, p, topk=30), metric.mean_PPVn(y, p, topk=5)])
    for i, (warm, total, lr, start) in enumerate(SCHEDULES):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=lr)
        s = sched.LinearWarmupCosineAnnealingLR(opt, warmup_epochs=warm, max_epochs=total, warmup_start_lr=start)
        lrs = []
        for _ in range(total):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            s.step()
        out[f"s{i}_cfg"] = np.array([warm, total, lr, start], dtype=np.float64)
        out[f"s{i}_lr"] = np.array(lrs, dtype=np.float64)
    path = os.path.join(ROOT, "tests", "golden", "metrics.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {len(out)} arrays, {os.path.getsize(path) / 1e3:.1f} kB")


if __name__ == "__main__":
    main()
