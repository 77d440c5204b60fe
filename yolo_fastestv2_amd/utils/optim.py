"""torch.optim.SGD as train.py:81-85 builds it (momentum 0.949, weight_decay 0.0005, dampening 0, no Nesterov), stepping the
parameters with libyfv2's SGD kernel: d = g + wd p; buf = d (first step) | momentum buf + d; p -= lr buf.
A ``torch.optim.Optimizer`` subclass, so ``param_groups`` (train.py:113-117 rewrites ``lr`` during the warm-up) and
``torch.optim.lr_scheduler.MultiStepLR`` (train.py:88-90) work on it unchanged.  All tensors of a parameter group travel in one
table (``yfv2_sgd_step_multi``): three launches for the network's 225 tensors instead of 225."""
import torch

from .. import _lib
from ..engine import get_engine


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("lr, momentum and weight_decay must be >= 0")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        # kernel-argument tables, one per parameter group, keyed by the group's position.  Kept OFF param_groups: state_dict()
        # copies every non-'params' key of a group, and a ctypes array of pointers can be neither pickled (torch.save of the
        # optimizer state) nor deep-copied
        self._yfv2_tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32:
                    raise RuntimeError("yolo_fastestv2_amd.SGD steps fp32 parameters on the MI355X only")
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                todo.append((p, g, st["momentum_buffer"], first))
            if not todo:
                continue
            # the table is rebuilt only when a pointer in it changed (gradients that are views of the Detector's persistent
            # bucket keep their addresses from step to step)
            key = tuple((p.data_ptr(), g.data_ptr(), b.data_ptr(), first) for p, g, b, first in todo)
            cache = self.__dict__.setdefault("_yfv2_tables", {}).setdefault(gi, {})
            if cache.get("key") != key:
                items = (_lib.SgdItem * len(todo))()
                for i, (p, g, b, first) in enumerate(todo):
                    items[i].param, items[i].grad, items[i].momentum_buf = p.data_ptr(), g.data_ptr(), b.data_ptr()
                    items[i].n, items[i].first_step = p.numel(), 1 if first else 0
                cache["key"], cache["items"] = key, items
            dev = todo[0][0].device
            eng = get_engine(dev, 32, 32, 1, 3)          # any handle of the device: the kernel only needs its error slot
            eng.sgd_step_multi(cache["items"], group["lr"], group["momentum"], group["weight_decay"])
            for p, g, _, _ in todo:
                # the kernel wrote through the raw pointer: move the autograd version counter as an in-place torch op would,
                # so that Detector.engine_for() sees the change and re-packs the inference weights on the next eval forward
                torch._C._increment_version(p)
                del g
        return loss
