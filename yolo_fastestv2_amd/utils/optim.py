"""torch.optim.SGD as train.py:81-85 builds it (momentum 0.949, weight_decay 0.0005, dampening 0, no Nesterov), stepping the
parameters with libyfv2's ``yfv2_sgd_step`` kernel: d = g + wd p; buf = d (first step) | momentum buf + d; p -= lr buf.
A ``torch.optim.Optimizer`` subclass, so ``param_groups`` (train.py:113-117 rewrites ``lr`` during the warm-up) and
``torch.optim.lr_scheduler.MultiStepLR`` (train.py:88-90) work on it unchanged."""
import torch

from ..engine import get_engine


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("lr, momentum and weight_decay must be >= 0")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32:
                    raise RuntimeError("yolo_fastestv2_amd.SGD steps fp32 parameters on the MI355X only")
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                eng = get_engine(p.device, 32, 32, 1, 3)          # any handle of the device: the kernel only needs its error slot
                eng.sgd_step(p.data, p.grad.contiguous(), st["momentum_buffer"], group["lr"], group["momentum"], group["weight_decay"], first)
                # the kernel wrote through the raw pointer: move the autograd version counter as an in-place torch op would,
                # so that Detector.engine_for() sees the change and re-packs the inference weights on the next eval forward
                torch._C._increment_version(p)
        return loss
