"""Drop-in for the reference's ``model.detector.Detector`` (model/detector.py:8-47).

Same constructor signature, same ``state_dict`` key set (so the reference's
checkpoints load with ``<All keys matched successfully>``), same forward
contract: ``(B,3,H,W)`` fp32 in [0,1] -> 6-tuple of raw NCHW logits on the
input's device.  The arithmetic is NOT torch: ``forward`` hands the input
pointer to libyfv2's HIP kernels through the C ABI (include/yfv2.h); this module
only owns the parameters (so ``.to()``, ``.parameters()``, ``load_state_dict``
behave like the reference module) and the glue.

Training: in ``.train()`` mode ``forward`` runs the library's train-mode forward (batch-statistics
BatchNorm, running statistics updated in this module's buffers) as a ``torch.autograd.Function`` whose backward is the
library's backward pass - ``total_loss.backward()`` (train.py:110) fills ``.grad`` of every parameter, ``yolo_fastestv2_amd.SGD``
(or any torch optimizer) steps them.  Correctness-first kernels: see yfv2_train.hip.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..engine import Engine
from ..sharded import average_gradients_

STAGE_REPEATS = (4, 8, 4)        # shufflenetv2.py:69
STAGE_CHANNELS = (24, 48, 96, 192)  # detector.py:11
OUT_DEPTH = 72                   # detector.py:10


def state_spec(classes, anchor_num):
    """Yield (key, shape, kind) for every entry of the reference state_dict;
    kind in {"conv", "bn_weight", "bn_bias", "bn_mean", "bn_var", "bn_count", "bias"}.
    Layer structure per SURVEY.md App. A."""
    def conv(name, co, ci, k):
        yield name + ".weight", (co, ci, k, k), "conv"

    def bn(name, c):
        yield name + ".weight", (c,), "bn_weight"
        yield name + ".bias", (c,), "bn_bias"
        yield name + ".running_mean", (c,), "bn_mean"
        yield name + ".running_var", (c,), "bn_var"
        yield name + ".num_batches_tracked", (), "bn_count"

    yield from conv("backbone.first_conv.0", STAGE_CHANNELS[0], 3, 3)
    yield from bn("backbone.first_conv.1", STAGE_CHANNELS[0])
    cin = STAGE_CHANNELS[0]
    for si, rep in enumerate(STAGE_REPEATS):
        cout = STAGE_CHANNELS[si + 1]
        mid = cout // 2
        for i in range(rep):
            p = "backbone.stage%d.%d" % (si + 2, i)
            inp = cin if i == 0 else cin // 2
            yield from conv(p + ".branch_main.0", mid, inp, 1)
            yield from bn(p + ".branch_main.1", mid)
            yield from conv(p + ".branch_main.3", mid, 1, 3)
            yield from bn(p + ".branch_main.4", mid)
            yield from conv(p + ".branch_main.5", cout - inp, mid, 1)
            yield from bn(p + ".branch_main.6", cout - inp)
            if i == 0:
                yield from conv(p + ".branch_proj.0", inp, 1, 3)
                yield from bn(p + ".branch_proj.1", inp)
                yield from conv(p + ".branch_proj.2", inp, inp, 1)
                yield from bn(p + ".branch_proj.3", inp)
            cin = cout
    yield from conv("fpn.conv1x1_2.0", OUT_DEPTH, STAGE_CHANNELS[2] + STAGE_CHANNELS[3], 1)
    yield from bn("fpn.conv1x1_2.1", OUT_DEPTH)
    yield from conv("fpn.conv1x1_3.0", OUT_DEPTH, STAGE_CHANNELS[3], 1)
    yield from bn("fpn.conv1x1_3.1", OUT_DEPTH)
    for head in ("cls_head_2", "reg_head_2", "reg_head_3", "cls_head_3"):
        p = "fpn.%s.block" % head
        yield from conv(p + ".0", OUT_DEPTH, 1, 5)
        yield from bn(p + ".1", OUT_DEPTH)
        yield from conv(p + ".3", OUT_DEPTH, OUT_DEPTH, 1)
        yield from bn(p + ".4", OUT_DEPTH)
        yield from conv(p + ".5", OUT_DEPTH, 1, 5)
        yield from bn(p + ".6", OUT_DEPTH)
        yield from conv(p + ".8", OUT_DEPTH, OUT_DEPTH, 1)
        yield from bn(p + ".9", OUT_DEPTH)
    for name, co in (("output_reg_layers", 4 * anchor_num), ("output_obj_layers", anchor_num),
                     ("output_cls_layers", classes)):
        yield name + ".weight", (co, OUT_DEPTH, 1, 1), "conv"
        yield name + ".bias", (co,), "bias"


class _TrainForward(torch.autograd.Function):
    """Detector.forward in train() mode: forward = yfv2_train_forward, backward = yfv2_train_backward (train.py:105-110).

    Gradients without copies: the library ADDS every parameter's gradient into a bound buffer; those buffers are views of one
    flat work bucket (zeroed per backward, all-reduced once under data_parallel()), which is then copied / added into a second,
    persistent bucket that the parameters' ``.grad`` tensors are views of - two kernels per backward where 225 ``clone()`` calls
    and autograd's per-parameter accumulation used to be.  ``.grad`` set to None (``optimizer.zero_grad()``) or replaced by the
    caller is handled: a missing one becomes a view again, a foreign tensor is added into."""

    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module.engine_for(x, sync=False)
        names = module._train_names()
        state = {k: v for k, v in module.state_dict(keep_vars=True).items() if v.is_floating_point()}
        # the binding lives in the NATIVE handle: Engine.ensure_batch() re-creates that handle (and drops its training state) when
        # any call sees a larger batch than max_batch, while the Python Engine object and the parameter pointers stay what they
        # were - so the handle's generation (Engine._create counts) is part of the key
        key = (id(eng), getattr(eng, "_generation", 0), str(x.device)) + tuple(v.data_ptr() for v in state.values())
        bound = module.__dict__.get("_train_bound")
        if bound is None or bound["key"] != key:
            for k, v in state.items():
                if v.device != x.device or v.dtype != torch.float32 or not v.is_contiguous():
                    raise RuntimeError("training needs every parameter / buffer as a contiguous fp32 tensor on %s ('%s' is not): model.to(device).float()" % (x.device, k))
            total = sum(state[k].numel() for k in names)
            work = torch.zeros(total, device=x.device, dtype=torch.float32)
            keep = torch.zeros(total, device=x.device, dtype=torch.float32)

            def views_of(flat):
                out, off = {}, 0
                for k in names:
                    n = state[k].numel()
                    out[k] = flat[off:off + n].view_as(state[k])
                    off += n
                return out
            bound = module.__dict__["_train_bound"] = {"key": key, "work": work, "keep": keep, "work_views": views_of(work), "keep_views": views_of(keep)}
            eng.train_bind({k: v.data for k, v in state.items()}, bound["work_views"])
        outs = eng.train_forward(x)
        ctx.x = getattr(eng, "_train_x", x)         # the tensor the library really reads again in backward (see Engine.train_forward)
        # the BatchNorm running statistics were just moved through raw pointers (no autograd version bump): every packed
        # inference copy of the weights is stale from here on
        module._synced.clear()
        with torch.no_grad():   # one multi-tensor launch for the 73 counters
            nbs = [mod._buffers["num_batches_tracked"] for mod in module.modules() if mod._buffers.get("num_batches_tracked") is not None]
            if nbs:
                torch._foreach_add_(nbs, 1)
        ctx.eng, ctx.seq, ctx.module, ctx.bound, ctx.names, ctx.dp = eng, eng._train_seq, module, bound, names, module._dp
        return outs

    @staticmethod
    def backward(ctx, *g6):
        if ctx.eng._train_seq != ctx.seq:
            raise RuntimeError("Detector (train mode): backward must follow the forward that produced these logits - another "
                               "train-mode forward ran on this engine in between")
        b = ctx.bound
        work, keep = b["work"], b["keep"]
        work.zero_()
        ctx.eng.train_backward([g if g is not None else torch.zeros(s, device=work.device) for g, s in zip(g6, ctx.eng.logit_shapes(g6[0].shape[0]))])
        if ctx.dp is not None:       # data parallel: one all-reduce over the whole gradient bucket (sharded.average_gradients_)
            average_gradients_(work, group=ctx.dp[0], force=ctx.dp[1])
        # NOTE: this function assigns / accumulates the parameters' .grad ITSELF (views of one persistent bucket) and returns None
        # for them: autograd's own accumulation - and with it torch.autograd.grad(), backward(inputs=...), tensor and
        # post-accumulate-grad hooks on the parameters - is bypassed.  train.py's loop (loss.backward(); optimizer.step()) is
        # what this path serves.  Parameters with requires_grad == False are left alone, as torch leaves them.
        params = dict(ctx.module.named_parameters())
        live = [k for k in ctx.names if params[k].requires_grad]
        frozen = len(live) != len(ctx.names)
        mine = {k: params[k].grad is not None and params[k].grad.data_ptr() == b["keep_views"][k].data_ptr() and params[k].grad.shape == b["keep_views"][k].shape
                for k in live}
        if not frozen and all(mine.values()):
            keep.add_(work)                          # accumulation over `subdivisions` batches (train.py:122), one kernel
        elif not frozen and all(params[k].grad is None for k in live):
            # the state optimizer.zero_grad() leaves behind on this torch (set_to_none=True is the default: train.py:118 runs into
            # it every iteration): ONE copy of the bucket, then every .grad becomes its view again - the parameter-by-parameter
            # branch below spent 225 copy launches per iteration here (rocprofv3: 350 __amd_rocclr_copyBuffer per iteration,
            # 1.2 ms of the 12.3 ms at batch 64)
            keep.copy_(work)
            for k in live:
                params[k].grad = b["keep_views"][k]
        else:
            foreign = {k: params[k].grad for k in live if params[k].grad is not None and not mine[k]}
            for k in live:                           # a mixed state: keep what is there, parameter by parameter
                if mine[k]:
                    b["keep_views"][k].add_(b["work_views"][k])
                else:
                    b["keep_views"][k].copy_(b["work_views"][k])
            for k in live:
                if not mine[k]:
                    if k in foreign:                 # somebody else's gradient tensor: add ours, as autograd would
                        foreign[k].add_(b["keep_views"][k])
                    else:
                        params[k].grad = b["keep_views"][k]
        return (None, None) + (None,) * len(ctx.names)


class _Node(nn.Module):
    """Bare container: gives the parameters their reference dotted names."""


class Detector(nn.Module):
    def __init__(self, classes, anchor_num, load_param, export_onnx=False):
        super().__init__()
        self._engines = {}      # (device, H, W) -> Engine
        self._synced = {}       # engine key -> weight version token
        self._dp = None         # (process group, force) once data_parallel() was called
        self.classes, self.anchor_num = int(classes), int(anchor_num)
        self.export_onnx = export_onnx
        for key, shape, kind in state_spec(self.classes, self.anchor_num):
            *path, leaf = key.split(".")
            node = self
            for comp in path:
                nxt = node._modules.get(comp)
                if nxt is None:
                    nxt = _Node()
                    node.add_module(comp, nxt)
                node = nxt
            if kind == "conv":
                t = torch.empty(shape)
                nn.init.kaiming_uniform_(t, a=math.sqrt(5))  # nn.Conv2d default
                node.register_parameter(leaf, nn.Parameter(t))
            elif kind == "bias":
                bound = 1.0 / math.sqrt(OUT_DEPTH)
                node.register_parameter(leaf, nn.Parameter(torch.empty(shape).uniform_(-bound, bound)))
            elif kind == "bn_weight":
                node.register_parameter(leaf, nn.Parameter(torch.ones(shape)))
            elif kind == "bn_bias":
                node.register_parameter(leaf, nn.Parameter(torch.zeros(shape)))
            elif kind == "bn_mean":
                node.register_buffer(leaf, torch.zeros(shape))
            elif kind == "bn_var":
                node.register_buffer(leaf, torch.ones(shape))
            else:
                node.register_buffer(leaf, torch.zeros(shape, dtype=torch.long))
        if load_param is False:
            # shufflenetv2.py:111-114: ImageNet backbone from a CWD-relative path
            path = "./model/backbone/backbone.pth"
            print("initialize_weights...")
            if os.path.exists(path):
                sd = torch.load(path, map_location="cpu")
                self.load_state_dict({"backbone." + k: v for k, v in sd.items()}, strict=False)
            else:
                print("  (%s not found: backbone keeps its random init)" % path)
        else:
            print("load param...")

    # -- weights -> engine -----------------------------------------------------------------
    # The engine holds a packed copy of the weights.  It is refreshed when (a) load_state_dict / .to() / .float() ... ran
    # (the overrides below mark every engine stale), or (b) a parameter or buffer was modified in place through the
    # tensor itself (optimizer steps, `p.mul_()`: the autograd version counter moves; checked per call on a cached
    # tensor list, ~20 us).  Writes through `p.data` bypass that counter: call sync_weights() after such surgery.
    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.__dict__.pop("_tensors", None)      # assign=True replaces the parameter objects: the cached list would go stale
        self._synced.clear()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.__dict__.pop("_tensors", None)
        if "_synced" in self.__dict__:
            self._synced.clear()
        return out

    def sync_weights(self):
        """Force a re-upload on the next forward (needed only after edits through ``param.data``)."""
        self.__dict__.pop("_tensors", None)
        self._synced.clear()

    def _version_token(self):
        # (object identity, autograd version) of every float tensor.  The cached list is rebuilt whenever the set of
        # parameter / buffer OBJECTS changed (`m.w = nn.Parameter(..)`, load_state_dict(assign=True)): their ids are
        # compared per call (a dict walk over ~330 entries, no tensor work).
        ids = tuple(id(t) for mod in self.modules() for d in (mod._parameters, mod._buffers) for t in d.values() if t is not None)
        cached = self.__dict__.get("_tensors")
        if cached is None or cached[0] != ids:
            cached = self.__dict__["_tensors"] = (ids, [t for t in self.state_dict(keep_vars=True).values() if t.is_floating_point()])
        return ids, tuple(t._version for t in cached[1])

    def data_parallel(self, group=None, enabled=True, force=False):
        """Extension over the reference surface (train.py is single-GPU): after this call every train-mode backward averages
        the gradients over the ranks of `group` (default: the world) with one all-reduce of the flat gradient bucket, so
        train.py's loop, unchanged, trains data-parallel when each rank feeds its own shard of the batch (one process per
        GPU, backend "nccl" = RCCL).  BatchNorm statistics stay per rank.  Returns self."""
        self._dp = (group, bool(force)) if enabled else None
        return self

    def _train_names(self):
        return [k for k, _ in self.named_parameters()]

    def engine_for(self, x, sync=True):
        hh, ww = (int(x.shape[1]), int(x.shape[2])) if (x.dtype == torch.uint8 and x.shape[-1] == 3) else (int(x.shape[2]), int(x.shape[3]))
        key = (str(x.device), hh, ww)
        eng = self._engines.get(key)
        if eng is None:
            eng = Engine(x.device, hh, ww, self.classes, self.anchor_num, max_batch=int(x.shape[0]))
            self._engines[key] = eng
        if not sync:
            return eng           # the training path reads the parameters in place
        tok = self._version_token()
        if self._synced.get(key) != tok:
            eng.load_state_dict(self.state_dict())
            self._synced[key] = tok
        return eng

    # -- forward ---------------------------------------------------------------------------
    def forward(self, x):
        if x.device.type != "cuda":
            raise RuntimeError("yolo_fastestv2_amd.Detector has no CPU path: move the input to the MI355X (.to('cuda'))")
        if self.training:
            if self.export_onnx:
                raise NotImplementedError("export_onnx=True is an inference layout (detector.py:33-44): call .eval()")
            if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
                raise ValueError("train mode takes the fp32 (B,3,H,W) tensor train.py:101 builds")
            params = [p for _, p in self.named_parameters()]
            out = _TrainForward.apply(self, x.contiguous(), *params)
            out[0]._yfv2_engine = self.engine_for(x, sync=False)
            return out
        eng = self.engine_for(x)
        # extension over the reference surface: a uint8 (B,H,W,3) tensor is the image before test.py:34-38's
        # reshape/permute/float()/255 - the stem kernel does that pre-process in its loads
        u8_hwc = x.dtype == torch.uint8 and x.dim() == 4 and x.shape[-1] == 3
        out = eng.forward(x if (u8_hwc or x.dtype == torch.float32) else x.float())
        out[0]._yfv2_engine = eng  # lets handel_preds reuse this handle (utils/utils.py)
        if self.export_onnx:
            # detector.py:33-44 export layout: post-sigmoid/softmax, NHWC, 12 reg + 3 obj + classes
            r2, o2, c2, r3, o3, c3 = out
            print("export onnx ...")
            return (torch.cat((r2.sigmoid(), o2.sigmoid(), F.softmax(c2, dim=1)), 1).permute(0, 2, 3, 1),
                    torch.cat((r3.sigmoid(), o3.sigmoid(), F.softmax(c3, dim=1)), 1).permute(0, 2, 3, 1))
        return out
