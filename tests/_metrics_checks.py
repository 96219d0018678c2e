"""Shared body of the accuracy-metric tests."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check_iou_golden(device):
    """IoU vs tests/golden/iou.npz (the reference's IoU.evaluate, oracle/make_golden.py iou): multi-class, binary,
    non-default thresholds (several channels above the threshold -> clipped class sums), a class that never occurs."""
    from atomai_amd.losses_metrics import IoU
    g = np.load(os.path.join(GOLD, "iou.npz"))
    names = sorted({k.split("|")[0] for k in g.files})
    assert len(names) == 5
    for name in names:
        K, thr = g[f"{name}|cfg"]
        logits = torch.from_numpy(g[f"{name}|logits"]).to(device)
        true = torch.from_numpy(g[f"{name}|true"]).to(device)
        got = IoU(true, logits, True, float(thr)).evaluate()
        assert abs(got - float(g[f"{name}|iou"])) < 1e-6, (name, got, float(g[f"{name}|iou"]))


def check_iou_wide_golden(device):
    """IoU vs tests/golden/iou_wide.npz (the reference's IoU.evaluate): more than 8 classes (the scores of a pixel are
    re-read instead of held in registers) and activation=False (probabilities in)."""
    from atomai_amd.losses_metrics import IoU
    g = np.load(os.path.join(GOLD, "iou_wide.npz"))
    names = sorted({k.split("|")[0] for k in g.files})
    assert len(names) == 5
    for name in names:
        K, thr, act = g[f"{name}|cfg"]
        pred = torch.from_numpy(g[f"{name}|pred"]).to(device)
        true = torch.from_numpy(g[f"{name}|true"]).to(device)
        got = IoU(true, pred, bool(act), float(thr)).evaluate()
        assert abs(got - float(g[f"{name}|iou"])) < 1e-6, (name, got, float(g[f"{name}|iou"]))


def check_fit_with_accuracy(device_is_gpu, tmp_path):
    """Segmentor.fit(compute_accuracy=True): the IoU of every train / test mini-batch is recorded as the reference does
    (trainer.py:163-172) and equals a host evaluation of the reference's formula on the same logits."""
    import atomai_amd as aoi
    rs = np.random.RandomState(3)
    X = rs.rand(6, 32, 32).astype(np.float32)
    y = rs.randint(0, 3, (6, 32, 32))
    m = aoi.models.Segmentor("Unet", nb_classes=3, nb_filters=4)
    m.fit(X, y, X[:2], y[:2], training_cycles=3, batch_size=2, compute_accuracy=True, plot_training_history=False,
          filename=str(tmp_path / "m"))
    assert len(m.loss_acc["train_accuracy"]) == len(m.loss_acc["test_accuracy"]) == 3
    assert all(0.0 <= v <= 1.0 for v in m.loss_acc["train_accuracy"] + m.loss_acc["test_accuracy"])
    # independent evaluation: the reference's arithmetic in numpy on the model's own logits
    m.net.eval()
    xb, yb = m.X_test[0], m.y_test[0]
    with torch.no_grad():
        logits = m.net(xb.to(m.device))
    p = torch.softmax(logits.float().cpu(), 1).numpy()
    pred = sum(c * (p[:, c] > 0.5) for c in range(3))
    pred[pred > 2] = 0
    t = yb.cpu().numpy()
    hist = np.zeros((3, 3), np.float32)
    for a, b in zip(t.reshape(-1), pred.reshape(-1)):
        hist[int(a), int(b)] += 1
    d = np.diag(hist)
    want = float(np.mean(d / (hist.sum(1) + hist.sum(0) - d + 1e-10)))
    got = m.accuracy_fn(yb.to(m.device), logits)
    assert abs(got - want) < 1e-5, (got, want)
