"""ctypes binding of libyfv2.so (the C ABI declared in include/yfv2.h).

There is deliberately NO fallback: if the shared library is missing or does not
export the ABI this module raises, and every product entry point fails loudly.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (or
``make -C yolo_fastestv2_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YFV2_LIB: opt-in override used only for same-box A/B of two builds (tools/gpu_quick.sh)
LIB_PATH = os.environ.get("YFV2_LIB") or os.path.join(_HERE, "libyfv2.so")
ABI_VERSION = 6
MAX_DET = 300

OK, ERR_ARG, ERR_CONFIG, ERR_DEVICE, ERR_WEIGHTS, ERR_STATE, ERR_BATCH, ERR_RANGE = 0, -1, -2, -3, -4, -5, -6, -7


class Config(C.Structure):
    _fields_ = [("classes", C.c_int32), ("anchor_num", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
                ("anchors", C.c_double * 12), ("max_batch", C.c_int32), ("device", C.c_int32)]


class Plan(C.Structure):
    """yfv2_plan (include/yfv2.h): WHICH kernels compute the path.  All zero = the default plan."""
    _fields_ = [("struct_size", C.c_int32), ("fp32_matrix", C.c_int32), ("layer_by_layer", C.c_int32), ("post_two_launches", C.c_int32),
                ("front_two_launches", C.c_int32), ("towers_unpaired", C.c_int32), ("lanes", C.c_int32), ("trace", C.c_int32), ("trace_step", C.c_int32)]


PLAN_FIELDS = ("fp32_matrix", "layer_by_layer", "post_two_launches", "front_two_launches", "towers_unpaired", "lanes", "trace", "trace_step")


def make_plan(plan=None):
    """dict (or None) -> Plan.  Unknown keys are an error: a misspelt switch must not silently select the default plan."""
    p = Plan()
    p.struct_size = C.sizeof(Plan)
    p.trace_step = -1
    for k, v in (plan or {}).items():
        if k not in PLAN_FIELDS:
            raise ValueError("unknown plan switch %r (known: %s)" % (k, ", ".join(PLAN_FIELDS)))
        setattr(p, k, int(v))
    return p


def plan_from_env(env=None):
    """The plan the measurement tools and the fallback-plan tests ask for through THEIR environment (the library itself reads none):
    YFV2_BF6=0 -> fp32_matrix, YFV2_FUSED=0 -> layer_by_layer, YFV2_POSTFUSE=0 -> post_two_launches, YFV2_FRONT=0 -> front_two_launches,
    YFV2_TPAIR=0 -> towers_unpaired, YFV2_LANES=N -> lanes, YFV2_TRACE=1 [YFV2_TRACE_STEP=k] -> trace [trace_step]."""
    env = os.environ if env is None else env
    off = lambda k: 1 if env.get(k, "1")[:1] == "0" else 0
    plan = {"fp32_matrix": off("YFV2_BF6"), "layer_by_layer": off("YFV2_FUSED"), "post_two_launches": off("YFV2_POSTFUSE"),
            "front_two_launches": off("YFV2_FRONT"), "towers_unpaired": off("YFV2_TPAIR")}
    if env.get("YFV2_LANES"):
        plan["lanes"] = int(env["YFV2_LANES"])
    if env.get("YFV2_TRACE", "")[:1] == "1":
        plan["trace"] = 1
        if env.get("YFV2_TRACE_STEP"):
            plan["trace_step"] = int(env["YFV2_TRACE_STEP"])
    return plan


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


class SgdItem(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("momentum_buf", C.c_void_p), ("n", C.c_int64), ("first_step", C.c_int32), ("reserved", C.c_int32)]


class Yfv2Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libyfv2 error %d: %s" % (code, msg))
        self.code = code


_PROTOTYPES = {
    # name: (restype, argtypes)
    "yfv2_abi_version": (C.c_int, []),
    "yfv2_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Config)]),
    "yfv2_create_ex": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Config), C.POINTER(Plan)]),
    "yfv2_destroy": (None, [C.c_void_p]),
    "yfv2_last_error": (C.c_char_p, [C.c_void_p]),
    "yfv2_load_weights": (C.c_int, [C.c_void_p, C.POINTER(TensorDesc), C.c_int32]),
    "yfv2_set_anchors": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "yfv2_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_void_p]),
    "yfv2_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_void_p]),
    "yfv2_decode": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p]),
    "yfv2_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_double, C.POINTER(C.c_int32), C.c_int32,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "yfv2_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_double, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p]),
    "yfv2_detect_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_double, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "yfv2_batch_statistics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float,
                                        C.c_void_p, C.c_void_p]),
    "yfv2_batch_statistics_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float,
                                              C.c_void_p, C.c_void_p]),
    "yfv2_batch_statistics_overflow": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "yfv2_nonfinite": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "yfv2_nonfinite_peek": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "yfv2_clock_probe_begin": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "yfv2_clock_probe_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "yfv2_loss": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "yfv2_train_bind": (C.c_int, [C.c_void_p, C.POINTER(TensorDesc), C.c_int32, C.POINTER(TensorDesc), C.c_int32]),
    "yfv2_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_void_p]),
    "yfv2_train_backward": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "yfv2_sgd_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p]),
    "yfv2_sgd_step_multi": (C.c_int, [C.c_void_p, C.POINTER(SgdItem), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "yfv2_resize_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "yfv2_debug_plan_dryrun": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "yfv2_debug_plan_dryrun_ex": (C.c_int, [C.c_void_p, C.POINTER(Plan), C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "yfv2_debug_plan_image": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_int64]),
    "yfv2_debug_plan_image_ex": (C.c_int64, [C.c_void_p, C.POINTER(Plan), C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_int64]),
    "yfv2_debug_plan_c2_label": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "yfv2_num_rows": (C.c_int32, [C.c_void_p]),
    "yfv2_num_stages": (C.c_int32, [C.c_void_p]),
    "yfv2_stage_info": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "yfv2_stage_kernel": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32]),
    "yfv2_profile_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                       C.POINTER(C.c_float), C.c_void_p]),
    "yfv2_debug_repeat_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "yfv2_debug_activation": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "yfv2_debug_train_relu_output": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle of libyfv2.so."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: the HIP extension has not been built (run __graft_entry__.build()); "
                          "yolo_fastestv2_amd has no CPU or PyTorch fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    # the ABI number first: a stale build must fail with THIS message, not with an AttributeError on a newer symbol
    if not hasattr(L, "yfv2_abi_version"):
        raise ImportError("%s does not export yfv2_abi_version: not a libyfv2 build" % LIB_PATH)
    L.yfv2_abi_version.restype, L.yfv2_abi_version.argtypes = C.c_int, []
    if L.yfv2_abi_version() != ABI_VERSION:
        raise ImportError("libyfv2.so ABI %d != binding ABI %d: rebuild (__graft_entry__.build())" % (L.yfv2_abi_version(), ABI_VERSION))
    for name, (res, args) in _PROTOTYPES.items():
        if name.startswith("yfv2_debug_") and os.environ.get("YFV2_LIB") and not hasattr(L, name):
            continue  # an older A/B build may lack a host-only test hook; product entry points are never optional
        fn = getattr(L, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error(handle=None):
    msg = lib().yfv2_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, handle=None):
    if rc != OK:
        raise Yfv2Error(rc, last_error(handle))
