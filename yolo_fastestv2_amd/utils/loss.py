"""Drop-in for the reference's ``utils/loss.py`` entry point ``compute_loss`` (:130-208): same signature, same 4-tuple
``(lbox, lobj, lcls, loss)`` of float32 tensors of shape (1,) on the logits' device, computed by libyfv2's HIP kernels
(``yfv2_loss``: build_target, CIoU in float64, objectness BCE, class cross-entropy).  ``loss.backward()`` works as in
``train.py:108``: the kernels also produce the gradient of the total loss w.r.t. the six logit maps, which this
autograd.Function hands on to whatever produced the logits - this package's ``Detector`` in ``train()`` mode carries them
down to every parameter (``yfv2_train_backward``, model/detector.py).  SURVEY.md 8(f) row 3."""
import torch

from ..engine import get_engine


class _DetectorLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, targets, *preds):
        need = any(p.requires_grad for p in preds)
        losses, grads = eng.loss(preds, targets, want_grad=need)
        ctx.grads = grads
        lbox, lobj, lcls, total = (losses[i:i + 1].clone() for i in range(4))
        ctx.mark_non_differentiable(lbox, lobj, lcls)        # train.py only ever calls total_loss.backward()
        return lbox, lobj, lcls, total

    @staticmethod
    def backward(ctx, g_lbox, g_lobj, g_lcls, g_total):
        if ctx.grads is None:
            return (None, None) + (None,) * 6
        return (None, None) + tuple(g * g_total for g in ctx.grads)


def compute_loss(preds, targets, cfg, device):
    preds = list(preds)
    if len(preds) != 6:
        raise ValueError("expected the 6-tuple returned by Detector.forward, got %d tensors" % len(preds))
    p0 = preds[0]
    if p0.device.type != "cuda":
        raise RuntimeError("compute_loss: logits must live on the MI355X (no CPU path)")
    eng = getattr(p0, "_yfv2_engine", None)
    if eng is None or eng.device != p0.device or eng.classes != preds[2].shape[1]:
        eng = get_engine(p0.device, cfg["height"], cfg["width"], preds[2].shape[1], cfg["anchor_num"])
    eng.set_anchors(cfg["anchors"])
    return _DetectorLoss.apply(eng, targets, *[p.float() for p in preds])
