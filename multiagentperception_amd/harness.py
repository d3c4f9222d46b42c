"""Harness counterpart of the reference's evaluator.

``evaluate_batches`` replays the call sequence of Trainer_MIMOcom.evaluate
(trainer.py:774-840): eval(), images concatenated on dim 1, labels on dim 0,
``model(images, training=False, MO_flag=True, inference=mode)``, class argmax, confusion
matrix, bandwidth meter.  ``RunningScore`` restates runningScore._fast_hist / get_scores
(metrics.py:99-108, 168-193).  Host-side numpy; the argmax runs on the device.
"""
import numpy as np
import torch


class RunningScore:
    def __init__(self, n_classes):
        self.n_classes = n_classes
        self.hist = np.zeros((n_classes, n_classes), dtype=np.int64)
        self.total_bandw = 0.0
        self.count = 0

    def update(self, label_true, label_pred):
        n = self.n_classes
        lt = np.asarray(label_true).reshape(-1)
        lp = np.asarray(label_pred).reshape(-1)
        keep = (lt >= 0) & (lt < n)
        self.hist += np.bincount(n * lt[keep].astype(np.int64) + lp[keep], minlength=n * n).reshape(n, n)

    def update_hist(self, hist):
        """add a confusion matrix computed elsewhere (the device-side histogram of forward_confusion)."""
        self.hist += np.asarray(hist, dtype=np.int64).reshape(self.n_classes, self.n_classes)

    def update_bandw(self, bandw):
        self.total_bandw += float(bandw)
        self.count += 1

    def scores(self):
        h = self.hist.astype(np.float64)
        diag = np.diag(h)
        with np.errstate(divide="ignore", invalid="ignore"):
            acc = diag.sum() / h.sum()
            acc_cls = np.nanmean(diag / h.sum(axis=1))
            iu = diag / (h.sum(axis=1) + h.sum(axis=0) - diag)
        freq = h.sum(axis=1) / h.sum()
        return {"Overall Acc": float(acc), "Mean Acc": float(acc_cls),
                "FreqW Acc": float((freq[freq > 0] * iu[freq > 0]).sum()), "Mean IoU": float(np.nanmean(iu)),
                "class_iou": iu, "bandwidth": (self.total_bandw / self.count) if self.count else 0.0}


def evaluate_batches(model, batches, device, inference_mode="activated", n_classes=11, mo_flag=True, fused_labels=False,
                     device_hist=False):
    """batches: iterable of (images_list[N] of [B,3,H,W], labels_list[N] of [B,H,W]).
    fused_labels=True uses model.forward_labels (class argmax fused into the upsample: 5 MB of u8 labels cross PCIe
    instead of 231 MB of f32 logits); `images_list` may then also be ONE u8 RGB frame tensor [B,N,H,W,3].
    device_hist=True goes one step further (model.forward_confusion): the confusion matrix itself is accumulated on
    the device, the label map is never written and the host bincount over 5 M pixels per step disappears; the n^2
    counters are read once after the last batch."""
    score = RunningScore(n_classes)
    model.eval()
    model.to(device)
    hist_dev = torch.zeros(n_classes * n_classes, dtype=torch.int64, device=device) if device_hist else None
    for images_list, labels_list in batches:
        labels = torch.cat(tuple(labels_list), dim=0) if mo_flag else labels_list[0]
        if device_hist:
            images = images_list if torch.is_tensor(images_list) else torch.cat(tuple(images_list), dim=1)
            gt = labels if labels.dtype in (torch.uint8, torch.int64) else labels.long()
            _, _, _, band_w = model.forward_confusion(images.to(device), gt.to(device), hist_dev, MO_flag=mo_flag,
                                                      inference=inference_mode)
            score.update_bandw(band_w)
            continue
        if fused_labels:
            images = images_list if torch.is_tensor(images_list) else torch.cat(tuple(images_list), dim=1)
            pred_dev, _, _, band_w = model.forward_labels(images.to(device), MO_flag=mo_flag, inference=inference_mode)
            pred = pred_dev.cpu().numpy().astype(np.int64)
            score.update(labels.numpy(), pred)
            score.update_bandw(band_w)
            continue
        images = torch.cat(tuple(images_list), dim=1)                       # trainer.py:793
        outputs, _, _, band_w = model(images.to(device), training=False, MO_flag=mo_flag, inference=inference_mode)
        pred = outputs.max(1)[1].cpu().numpy()                              # trainer.py:804
        score.update(labels.numpy(), pred)
        score.update_bandw(band_w)
    if device_hist:
        score.update_hist(hist_dev.cpu().numpy())
    return score.scores()
