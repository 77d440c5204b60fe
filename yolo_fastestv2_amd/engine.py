"""Engine: one libyfv2 handle on one GPU.  PyTorch is used for device memory,
streams and host<->device copies only; every arithmetic op runs in the HIP
kernels behind the C ABI (include/yfv2.h)."""
import ctypes as C

import torch

from . import _lib
from ._lib import MAX_DET, Config, TensorDesc, check

LOGIT_ORDER = ("reg_2", "obj_2", "cls_2", "reg_3", "obj_3", "cls_3")  # detector.py:47


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Engine:
    """Owns a yfv2 handle.  ``max_batch`` grows on demand (the handle is re-created
    and the weights re-uploaded)."""

    def __init__(self, device, height=352, width=352, classes=80, anchor_num=3, anchors=None, max_batch=1, plan=None):
        """plan: dict of yfv2_plan switches ({"fp32_matrix": 1}, {"layer_by_layer": 1}, {"lanes": 2} ...; include/yfv2.h), None = what the
        process environment asks for (YFV2_BF6=0 etc., _lib.plan_from_env: how the tools and the fallback-plan tests pick a plan) - with
        nothing set that is the default plan.  The native library itself reads no environment variable."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("yolo_fastestv2_amd runs on an MI355X only (got device %s); there is no CPU path" % device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.height, self.width = int(height), int(width)
        self.classes, self.anchor_num = int(classes), int(anchor_num)
        self.anchors = [float(a) for a in (anchors if anchors is not None else [0.0] * 12)]
        self._anchors_set = anchors is not None   # decode()/detect() refuse to run on the all-zero placeholder
        if len(self.anchors) != 12:
            raise ValueError("expected 12 anchor values (6 pairs), got %d" % len(self.anchors))
        self.max_batch = 0
        self._h = None
        self._weights = None  # host copies (name -> contiguous fp32 cpu tensor), kept for re-creation
        self.plan = dict(_lib.plan_from_env() if plan is None else plan)
        self._create(max_batch)

    # ---- lifetime -------------------------------------------------------------------------
    def _create(self, max_batch):
        L = _lib.lib()
        cfg = Config()
        cfg.classes, cfg.anchor_num = self.classes, self.anchor_num
        cfg.height, cfg.width = self.height, self.width
        for i, a in enumerate(self.anchors):
            cfg.anchors[i] = a
        cfg.max_batch = int(max_batch)
        cfg.device = self.device.index
        h = C.c_void_p()
        check(L.yfv2_create_ex(C.byref(h), C.byref(cfg), C.byref(_lib.make_plan(self.plan))))
        self.close()
        self._h, self.max_batch = h, int(max_batch)
        self._generation = getattr(self, "_generation", 0) + 1   # a NEW native handle: whatever was bound to the old one (yfv2_train_bind) is gone
        self.rows = int(L.yfv2_num_rows(self._h))
        if self._weights is not None:
            self._upload()

    def close(self):
        if self._h is not None:
            _lib.lib().yfv2_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ensure_batch(self, B):
        if B > self.max_batch:
            torch.cuda.synchronize(self.device)
            self._create(max(B, 2 * self.max_batch if self.max_batch < 64 else B))

    # ---- weights --------------------------------------------------------------------------
    def load_state_dict(self, state):
        """state: mapping reference-key -> tensor (any device/dtype); integer
        buffers (num_batches_tracked) are ignored."""
        host = {}
        for k, v in state.items():
            v = torch.as_tensor(v)
            if not v.is_floating_point():
                continue
            host[k] = v.detach().to("cpu", torch.float32).contiguous()
        self._weights = host
        self._upload()

    def _upload(self):
        names = list(self._weights)
        arr = (TensorDesc * len(names))()
        for i, k in enumerate(names):
            t = self._weights[k]
            arr[i].name = k.encode()
            arr[i].data = t.data_ptr()
            arr[i].numel = t.numel()
        check(_lib.lib().yfv2_load_weights(self._h, arr, len(names)), self._h)

    def set_anchors(self, anchors):
        anchors = [float(a) for a in anchors]
        if anchors != self.anchors:
            if len(anchors) != 12:
                raise ValueError("expected 12 anchor values (6 pairs), got %d" % len(anchors))
            self.anchors = anchors
        self._anchors_set = True
        check(_lib.lib().yfv2_set_anchors(self._h, (C.c_double * 12)(*self.anchors)), self._h)

    # ---- shapes ---------------------------------------------------------------------------
    def logit_shapes(self, B):
        A, nc = self.anchor_num, self.classes
        h2, w2, h3, w3 = self.height // 16, self.width // 16, self.height // 32, self.width // 32
        return [(B, 4 * A, h2, w2), (B, A, h2, w2), (B, nc, h2, w2), (B, 4 * A, h3, w3), (B, A, h3, w3), (B, nc, h3, w3)]

    def _check_x(self, x):
        """fp32 (B,3,H,W) in [0,1] (the reference's Detector input), or uint8 (B,H,W,3) in 0..255: the decoded,
        resized image BEFORE test.py:34-38's reshape/permute/float()/255, which the stem kernel then does itself."""
        if x.device != self.device:
            raise ValueError("input on %s, engine on %s" % (x.device, self.device))
        if x.dtype == torch.uint8:
            if x.dim() != 4 or tuple(x.shape[1:]) != (self.height, self.width, 3):
                raise ValueError("expected uint8 (B,%d,%d,3), got %s" % (self.height, self.width, tuple(x.shape)))
        elif x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[1:]) != (3, self.height, self.width):
            raise ValueError("expected fp32 (B,3,%d,%d) or uint8 (B,%d,%d,3), got %s %s" % (self.height, self.width, self.height, self.width, x.dtype, tuple(x.shape)))
        x = x.contiguous()
        if x.data_ptr() % 16:        # a view into the middle of a storage: the stem kernels load 16-byte (fp32) / 4-byte-aligned 12-byte (uint8) records
            x = x.clone()
        return x

    # ---- the hot path ---------------------------------------------------------------------
    def forward(self, x, out=None):
        x = self._check_x(x)
        B = x.shape[0]
        self.ensure_batch(B)
        if out is None:
            out = [torch.empty(s, device=self.device, dtype=torch.float32) for s in self.logit_shapes(B)]
        ptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in out])
        fn = _lib.lib().yfv2_forward_u8 if x.dtype == torch.uint8 else _lib.lib().yfv2_forward
        check(fn(self._h, _ptr(x), B, ptrs, _stream(self.device)), self._h)
        return tuple(out)

    def _need_anchors(self, what):
        if not self._anchors_set:
            raise RuntimeError("Engine.%s: anchors were never given (constructor `anchors=` or set_anchors(cfg['anchors'])); "
                               "the all-zero placeholder would decode every box to zero size" % what)

    def decode(self, preds, out=None):
        self._need_anchors("decode")
        preds = [p.contiguous() for p in preds]
        B = preds[0].shape[0]
        for p, s in zip(preds, self.logit_shapes(B)):
            if tuple(p.shape) != s or p.dtype != torch.float32 or p.device != self.device:
                raise ValueError("logit tensor %s %s on %s, expected fp32 %s on %s" % (p.dtype, tuple(p.shape), p.device, s, self.device))
        self.ensure_batch(B)
        if out is None:
            out = torch.empty((B, self.rows, 5 + self.classes), device=self.device, dtype=torch.float32)
        ptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in preds])
        check(_lib.lib().yfv2_decode(self._h, ptrs, B, _ptr(out), _stream(self.device)), self._h)
        return out

    def new_det_buffers(self, B):
        """(dets, idx, cnt) for `detect` / `nms`: three views of one flat buffer, so that the sharded path's all-gather
        moves a rank's result in one collective (sharded.gather_detections)."""
        from .sharded import packed_det_buffers
        return packed_det_buffers(B, self.device)

    def nms(self, boxes, conf_thres, iou_thres, classes=None, out=None):
        boxes = boxes.contiguous()
        B = boxes.shape[0]
        if tuple(boxes.shape[1:]) != (self.rows, 5 + self.classes) or boxes.dtype != torch.float32 or boxes.device != self.device:
            raise ValueError("decoded tensor %s %s on %s, expected fp32 (B,%d,%d) on %s" % (
                boxes.dtype, tuple(boxes.shape), boxes.device, self.rows, 5 + self.classes, self.device))
        self.ensure_batch(B)
        dets, idx, cnt = out if out is not None else self.new_det_buffers(B)
        if classes is not None:
            cl = [int(c) for c in classes]
            carr, ncl = (C.c_int32 * len(cl))(*cl), len(cl)
        else:
            carr, ncl = None, 0
        check(_lib.lib().yfv2_nms(self._h, _ptr(boxes), B, float(conf_thres), float(iou_thres), carr, ncl, _ptr(dets),
                                  _ptr(idx), _ptr(cnt), _stream(self.device)), self._h)
        return dets, idx, cnt

    def detect(self, x, conf_thres, iou_thres, out=None, check=True):
        """forward + decode + NMS, enqueue only: the results are device tensors and nothing waits for the device.  check=True
        (default) LOOKS at the range-guard word first (yfv2_nonfinite_peek: a host memory read, no synchronisation) and raises if a
        call that has already completed on this handle tripped it - a loop that never synchronises with the host learns of
        invalid results one call late instead of never; `check_finite()` after a synchronisation is the exact query."""
        self._need_anchors("detect")
        if check and self.peek_nonfinite():
            self.check_finite("detect (an earlier call on this handle)")
        x = self._check_x(x)
        B = x.shape[0]
        self.ensure_batch(B)
        dets, idx, cnt = out if out is not None else self.new_det_buffers(B)
        fn = _lib.lib().yfv2_detect_u8 if x.dtype == torch.uint8 else _lib.lib().yfv2_detect
        _lib.check(fn(self._h, _ptr(x), B, float(conf_thres), float(iou_thres), _ptr(dets), _ptr(idx), _ptr(cnt), _stream(self.device)), self._h)
        return dets, idx, cnt

    def resize(self, frames, out=None):
        """uint8 (B, h, w, 3) frames on the GPU -> uint8 (B, height, width, 3): cv2.resize(..., INTER_LINEAR) of test.py:35 /
        utils/datasets.py:107 on the device; the result feeds forward()/detect() directly."""
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3 or frames.device != self.device:
            raise ValueError("frames must be uint8 (B,h,w,3) on %s" % self.device)
        frames = frames.contiguous()
        B, sh, sw = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        if out is None:
            out = torch.empty((B, self.height, self.width, 3), device=self.device, dtype=torch.uint8)
        check(_lib.lib().yfv2_resize_u8(self._h, _ptr(frames), B, sh, sw, _ptr(out), _stream(self.device)), self._h)
        return out

    def batch_statistics(self, dets, cnt, targets, iou_threshold, sync=True):
        """True-positive flags (B, 300) int32 for the padded detections of nms()/detect() against targets (T,6)
        [image index, label, x1, y1, x2, y2] - utils/utils.py:194-230 get_batch_statistics on the device.
        ``sync=False`` only enqueues (no host wait); call ``stats_overflowed()`` once after the last batch."""
        B = dets.shape[0]
        if tuple(dets.shape) != (B, MAX_DET, 6) or dets.dtype != torch.float32 or dets.device != self.device:
            raise ValueError("dets must be fp32 (B,%d,6) on %s" % (MAX_DET, self.device))
        targets = targets.to(self.device, torch.float32).reshape(-1, 6).contiguous()
        tp = torch.empty((B, MAX_DET), device=self.device, dtype=torch.int32)
        fn = _lib.lib().yfv2_batch_statistics if sync else _lib.lib().yfv2_batch_statistics_async
        check(fn(self._h, _ptr(dets.contiguous()), _ptr(cnt.contiguous()), B, _ptr(targets) if targets.numel() else None,
                 int(targets.shape[0]), float(iou_threshold), _ptr(tp), _stream(self.device)), self._h)
        return tp

    def loss(self, preds, targets, want_grad=False):
        """utils/loss.py:130-208 compute_loss on the device: (losses, grads) with losses a float32 (4,) device tensor
        [lbox, lobj, lcls, total] and grads the gradients of `total` w.r.t. the six logit maps (None unless want_grad)."""
        self._need_anchors("loss")
        preds = [p.detach().contiguous() for p in preds]
        B = preds[0].shape[0]
        for p, s in zip(preds, self.logit_shapes(B)):
            if tuple(p.shape) != s or p.dtype != torch.float32 or p.device != self.device:
                raise ValueError("logit tensor %s %s on %s, expected fp32 %s on %s" % (p.dtype, tuple(p.shape), p.device, s, self.device))
        self.ensure_batch(B)
        targets = torch.as_tensor(targets).to(self.device, torch.float32).reshape(-1, 6).contiguous()
        losses = torch.empty(4, device=self.device, dtype=torch.float32)
        grads = [torch.empty_like(p) for p in preds] if want_grad else None
        ptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in preds])
        gptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in grads]) if want_grad else None
        check(_lib.lib().yfv2_loss(self._h, ptrs, B, _ptr(targets) if targets.numel() else None, int(targets.shape[0]), _ptr(losses),
                                   gptrs, _stream(self.device)), self._h)
        return losses, grads

    # ---- training path: train.py:96-123 on the device -----------------------------------------
    def train_bind(self, tensors, grads):
        """tensors: name -> fp32 device tensor for every floating-point state_dict entry (weights, biases, BatchNorm running
        statistics); grads: name -> fp32 device buffer for every trainable parameter.  The library keeps the POINTERS."""
        def table(d):
            arr = (TensorDesc * len(d))()
            for i, (k, t) in enumerate(d.items()):
                if t.dtype != torch.float32 or t.device != self.device or not t.is_contiguous():
                    raise ValueError("train_bind: '%s' must be a contiguous fp32 tensor on %s" % (k, self.device))
                arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
            return arr
        self._train_keep = (tensors, grads)          # the tensors must outlive the binding
        ta, ga = table(tensors), table(grads)
        check(_lib.lib().yfv2_train_bind(self._h, ta, len(tensors), ga, len(grads)), self._h)

    def train_forward(self, x, out=None):
        """Detector.forward in train() mode (batch-statistics BatchNorm; running statistics updated in the bound buffers)."""
        x = self._check_x(x)
        if x.dtype != torch.float32:
            raise ValueError("train_forward takes the fp32 (B,3,H,W) tensor train.py:101 builds")
        B = x.shape[0]
        if out is None:
            out = [torch.empty(s, device=self.device, dtype=torch.float32) for s in self.logit_shapes(B)]
        ptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in out])
        check(_lib.lib().yfv2_train_forward(self._h, _ptr(x), B, ptrs, _stream(self.device)), self._h)
        self._train_seq = getattr(self, "_train_seq", 0) + 1
        # the library does NOT copy the input: the backward's first-conv weight gradient re-reads it through the raw pointer
        # (yfv2_train.hip).  `x` may be a temporary made here (contiguous() / clone()) or by the caller: hold it until the next
        # train-mode forward replaces the tape, or the caching allocator hands the block to somebody else before backward runs
        self._train_x = x
        return tuple(out)

    def train_backward(self, grads6):
        """From the gradient of the loss w.r.t. the six logit maps down to every parameter: ADDS into the bound gradient buffers."""
        gs = [g.to(self.device, torch.float32).contiguous() for g in grads6]
        ptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in gs])
        check(_lib.lib().yfv2_train_backward(self._h, ptrs, _stream(self.device)), self._h)

    def debug_train_relu_output(self, conv_name):
        """flat host tensor (B*C*H*W, NCHW order): what the ReLU after conv `conv_name` wrote in the last train_forward"""
        L = _lib.lib()
        n = L.yfv2_debug_train_relu_output(self._h, conv_name.encode(), None, 0)
        if n < 0:
            check(-1, self._h)
        host = torch.empty(n, dtype=torch.float32)
        if L.yfv2_debug_train_relu_output(self._h, conv_name.encode(), C.c_void_p(host.data_ptr()), n) != n:
            check(-1, self._h)
        return host

    def sgd_step(self, param, grad, buf, lr, momentum, weight_decay, first):
        check(_lib.lib().yfv2_sgd_step(self._h, _ptr(param), _ptr(grad), _ptr(buf), param.numel(), float(lr), float(momentum), float(weight_decay),
                                       1 if first else 0, _stream(self.device)), self._h)

    def sgd_step_multi(self, items, lr, momentum, weight_decay):
        """items: a prepared (SgdItem * n) table (see utils/optim.py) - one entry per parameter tensor, a few launches in all."""
        check(_lib.lib().yfv2_sgd_step_multi(self._h, items, len(items), float(lr), float(momentum), float(weight_decay), _stream(self.device)), self._h)

    def stats_overflowed(self):
        """Waits for the stream; True if any batch_statistics(sync=False) call since the last query met an image with
        more than 1024 targets (its flags are then invalid).  Clears the flag."""
        over = C.c_int32(0)
        check(_lib.lib().yfv2_batch_statistics_overflow(self._h, C.byref(over), _stream(self.device)), self._h)
        return bool(over.value)

    def nonfinite(self):
        """Waits for the stream; True if any forward / detect on this handle since the last query tripped the range guard of the
        fp16x3 plan (an activation beyond +-4094, an fp32 input beyond +-255.9, or a non-finite value: include/yfv2.h
        yfv2_nonfinite).  Clears the word."""
        flag = C.c_int32(0)
        check(_lib.lib().yfv2_nonfinite(self._h, C.byref(flag), _stream(self.device)), self._h)
        return bool(flag.value)

    def peek_nonfinite(self):
        """True if a kernel that has ALREADY COMPLETED on this handle tripped the range guard; waits for nothing, clears nothing."""
        flag = C.c_int32(0)
        check(_lib.lib().yfv2_nonfinite_peek(self._h, C.byref(flag)), self._h)
        return bool(flag.value)

    def clock_probe_begin(self, workgroups=256, milliseconds=50.0, busy=True):
        """Enqueue the shader-clock probe on the CURRENT stream (include/yfv2.h yfv2_clock_probe_begin)."""
        check(_lib.lib().yfv2_clock_probe_begin(self._h, int(workgroups), float(milliseconds), 1 if busy else 0, _stream(self.device)), self._h)

    def clock_probe_end(self):
        """Waits for the current stream; dict with the effective shader clock (MHz) min / mean / max over the probe's workgroups."""
        out = (C.c_double * 6)()
        check(_lib.lib().yfv2_clock_probe_end(self._h, out, _stream(self.device)), self._h)
        return {"sclk_mhz_min": round(out[0], 1), "sclk_mhz_mean": round(out[1], 1), "sclk_mhz_max": round(out[2], 1),
                "ref_clock_mhz": round(out[3], 3), "interval_ms": round(out[4], 3), "xcds_seen": int(out[5])}

    def check_finite(self, what="forward"):
        """Raise if the range guard tripped (call where the host waits for the device anyway)."""
        if self.nonfinite():
            raise _lib.Yfv2Error(_lib.ERR_RANGE, "%s: an activation left the range of the default (fp16x3) plan - |activation| >= 4094 or a non-finite "
                                     "input; the result is invalid.  Create the handle on the fp32-matrix plan - Engine(..., plan={'fp32_matrix': 1}), "
                                     "yfv2_plan.fp32_matrix = 1, or YFV2_BF6=0 in the environment of the Python layer (every conv on the fp32 matrix "
                                     "instructions, no such bound) - for this model / input" % what)

    # ---- introspection --------------------------------------------------------------------
    def stages(self):
        L = _lib.lib()
        n = L.yfv2_num_stages(self._h)
        out = []
        buf = C.create_string_buffer(256)
        for i in range(n):
            fl, by, ex = C.c_double(), C.c_double(), C.c_double()
            check(L.yfv2_stage_info(self._h, i, buf, 256, C.byref(fl), C.byref(by), C.byref(ex)), self._h)
            st = {"name": buf.value.decode(), "flops_per_image": fl.value, "bytes_per_image": by.value, "external_bytes_per_image": ex.value}
            check(L.yfv2_stage_kernel(self._h, i, buf, 256), self._h)
            st["kernel"] = buf.value.decode()
            out.append(st)
        return out

    def profile_forward(self, x, iters=5):
        """Per-launch mean milliseconds (hipEvent pairs on the current stream)."""
        x = self._check_x(x)
        B = x.shape[0]
        self.ensure_batch(B)
        out = [torch.empty(s, device=self.device, dtype=torch.float32) for s in self.logit_shapes(B)]
        ptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in out])
        n = _lib.lib().yfv2_num_stages(self._h)
        ms = (C.c_float * n)()
        check(_lib.lib().yfv2_profile_forward(self._h, _ptr(x), B, ptrs, int(iters), ms, _stream(self.device)), self._h)
        return list(ms)

    def debug_activation(self, which, B):
        """NHWC activation of the last forward: 0 stem+pool, 1 stage2, 2 C2, 3 C3, 4 S2, 5 S3.
        (0: the default plan runs the stem and stage2.0 as ONE launch that never writes the stem's output - the hook then re-runs the
        stem's own launch on the LAST forward's input, which the caller must still hold.)"""
        L = _lib.lib()
        n = L.yfv2_debug_activation(self._h, which, B, None, 0)
        if n < 0:
            check(int(n), self._h)
        host = torch.empty(n, dtype=torch.float32)
        got = L.yfv2_debug_activation(self._h, which, B, C.c_void_p(host.data_ptr()), n)
        if got < 0:
            check(int(got), self._h)
        return host


def unpack_detections(dets, idx, cnt):
    """(B,300,6),(B,300),(B) device tensors -> (list of (n_i,6) CPU tensors, list of (n_i,) CPU index tensors).
    One D2H copy per tensor (not per image)."""
    cnt_h = cnt.cpu()
    dets_h, idx_h = dets.cpu(), idx.cpu()
    rows, ids = [], []
    for b in range(cnt_h.shape[0]):
        n = int(cnt_h[b])
        rows.append(dets_h[b, :n].clone())
        ids.append(idx_h[b, :n].to(torch.int64))
    return rows, ids


_engines = {}


def get_engine(device, height, width, classes=80, anchor_num=3):
    """Process-wide engine cache: one handle per (device, H, W, classes)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (str(device), int(height), int(width), int(classes), int(anchor_num))
    eng = _engines.get(key)
    if eng is None:
        eng = Engine(device, height, width, classes, anchor_num)
        _engines[key] = eng
    return eng
