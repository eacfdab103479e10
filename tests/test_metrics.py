"""Inference metrics and the learning-rate schedule against the reference's outputs (``tests/golden/metrics.npz``,
generated from the reference's ``procedures/metric.py`` / ``utils/scheduler.py`` by ``oracle/make_golden_metrics.py``)
and against scikit-learn on more inputs."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from immunostruct_amd.procedures import metric as M
from immunostruct_amd.utils.scheduler import LinearWarmupCosineAnnealingLR

GOLDEN = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))
KEYS = ["optimal_threshold", "accuracy", "accuracy_op", "f1", "f1_op", "precision", "precision_op", "recall", "recall_op",
        "roc_auc", "pr_auc", "ppvn", "ppvn_op", "ppv30", "ppv30_op"]


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize("case", range(4))
def test_metrics_match_reference_golden(case):
    y, p, want = GOLDEN[f"m{case}_y"], GOLDEN[f"m{case}_p"], GOLDEN[f"m{case}_out"]
    thr = M.find_optimal_threshold(y, p)
    assert thr == want[0]                                  # an element of the score array: exact
    got = quiet(M.evaluate_metrics, y, p, thr)
    assert list(got.keys()) == KEYS                        # same dictionary, same order
    np.testing.assert_allclose([float(got[k]) for k in KEYS], want, rtol=1e-12, atol=0)
    scores = [quiet(M.mean_PPVn, y, p), quiet(M.mean_PPVn, y, p, topk=30), quiet(M.mean_PPVn, y, p, topk=5)]
    np.testing.assert_allclose(scores, GOLDEN[f"m{case}_ppvn_scores"], rtol=1e-14)


@pytest.mark.parametrize("seed", range(6))
def test_curves_match_sklearn(seed):
    from sklearn import metrics as SK
    rs = np.random.RandomState(100 + seed)
    n = int(rs.randint(20, 400))
    y = (rs.rand(n) < 0.3).astype(np.float32)
    y[:2] = (0, 1)
    p = 1 / (1 + np.exp(-(1.5 * (y - 0.5) + rs.normal(size=n))))
    if seed % 2:
        p = np.round(p, 1)                                  # heavy ties
    fpr, tpr, thr = M._roc(y, p)
    f2, t2, th2 = SK.roc_curve(y, p)
    np.testing.assert_allclose(fpr, f2, rtol=0, atol=1e-15)
    np.testing.assert_allclose(tpr, t2, rtol=0, atol=1e-15)
    np.testing.assert_array_equal(thr, th2)
    assert abs(M._trapezoid(fpr, tpr) - SK.roc_auc_score(y, p)) < 1e-12
    pr, rc, _ = SK.precision_recall_curve(y, p)
    assert abs(M._pr_auc(y, p) - SK.auc(rc, pr)) < 1e-12
    pred = p >= 0.5
    tp, fp, fn, tn = M._confusion(y, pred)
    assert abs(M._ratio(2 * tp, 2 * tp + fp + fn) - SK.f1_score(y, pred)) < 1e-12


def test_metric_error_behaviour():
    with pytest.raises(AssertionError):
        M.mean_PPVn(np.ones(3), np.ones(4))
    with pytest.raises(ValueError):                         # a single class: the ROC curve is undefined (sklearn raises too)
        M.find_optimal_threshold(np.ones(5), np.linspace(0, 1, 5))


@pytest.mark.parametrize("case", range(3))
def test_scheduler_matches_reference_golden(case):
    warm, total, lr, start = GOLDEN[f"s{case}_cfg"]
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=float(lr))
    s = LinearWarmupCosineAnnealingLR(opt, warmup_epochs=int(warm), max_epochs=int(total), warmup_start_lr=float(start))
    lrs = []
    for _ in range(int(total)):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        s.step()
    np.testing.assert_allclose(lrs, GOLDEN[f"s{case}_lr"], rtol=1e-12, atol=1e-20)
