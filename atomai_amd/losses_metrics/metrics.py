"""Accuracy metrics on the device (reference: atomai/losses_metrics/metrics.py:16-95).

``IoU(true, pred, activation=True, thresh=.5).evaluate()`` has the reference's signature and result.  The reference
moves logits and labels to the host, thresholds every frame with ``cv2.threshold`` (THRESH_BINARY: p > thresh -> 1),
squeezes the channels into a class map (``squeeze_channels(clip=True)``, transforms/imaug.py:361-393) and builds one
K x K confusion histogram per frame with ``torch.bincount``; here ONE kernel pass over the logits (``amx_iou_hist``,
csrc/head.hip) produces the same per-frame integer histograms, the host reads N*K*K integers and finishes the Jaccard
arithmetic in float32 in the reference's order (per-frame histograms accumulated, A, B, diag, 1e-10, mean).
"""
import torch

from .. import _lib as L


class IoU:
    """Mean intersection over union of (labels, thresholded predictions)."""

    def __init__(self, true: torch.Tensor, pred: torch.Tensor, activation: bool = True, thresh: float = 0.5):
        if pred.ndim != 4:
            raise AssertionError("expected predictions of shape (N, K, H, W)")
        self.thresh = thresh
        N, K = pred.shape[0], pred.shape[1]
        HW = pred.shape[2] * pred.shape[3]
        self.nb_classes = max(K, 2)
        x = pred.detach().float().contiguous()
        true = true.detach().to(x.device)
        if true.numel() != N * HW:
            raise AssertionError("labels and predictions must describe the same pixels")
        ti = tf = None
        if true.dtype.is_floating_point:
            tf = true.float().contiguous()
        else:
            ti = true.long().contiguous()
        self.hist = torch.zeros((N, self.nb_classes, self.nb_classes), dtype=torch.int32, device=x.device)
        # activation=False: `pred` already holds probabilities (metrics.py:37-41 skipped); any number of classes
        L.call("amx_iou_hist", L.ptr(x), L.ptr(ti), L.ptr(tf), N, K, HW, float(thresh), int(bool(activation)),
               L.ptr(self.hist), L.stream_ptr(x))

    def evaluate(self) -> float:
        """Mean Jaccard index over the classes (metrics.py:80-95), in float32 like the reference: the per-frame count
        matrices are added frame by frame, then inter / (row + col - inter + 1e-10) per class."""
        import numpy as np
        counts = self.hist.cpu().numpy().astype(np.float32)          # the only host transfer: N * K * K integers
        total = np.zeros((self.nb_classes, self.nb_classes), dtype=np.float32)
        for frame in counts:
            total += frame
        inter = np.diagonal(total)
        union = total.sum(axis=1, dtype=np.float32) + total.sum(axis=0, dtype=np.float32) - inter + np.float32(1e-10)
        jcd = inter / union
        return float(jcd[~np.isnan(jcd)].mean(dtype=np.float32))
