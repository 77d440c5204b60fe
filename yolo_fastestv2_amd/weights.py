"""Seeded random-init weights of the reference architecture (state_dict keys of
model/detector.py's Detector), for benchmarks and tests on boxes without the
COCO checkpoint.  He-normal filters keep activations in the normal fp32 range
through the ~40 sequential layers; BN statistics are mildly randomised so the
scale/shift path is exercised."""
import torch

from .model.detector import STAGE_REPEATS


def random_state_dict(seed=0, classes=80, anchor_num=3):
    """-> dict key -> cpu tensor, same keys/shapes as the reference checkpoint."""
    g = torch.Generator().manual_seed(seed)
    w = {}

    def conv(name, co, ci, k, groups=1):
        fan_in = (ci // groups) * k * k
        w[name + ".weight"] = torch.randn(co, ci // groups, k, k, generator=g) * (2.0 / fan_in) ** 0.5

    def bn(name, c):
        w[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        w[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        w[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        w[name + ".running_var"] = 1.0 + 0.1 * torch.rand(c, generator=g)
        w[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    conv("backbone.first_conv.0", 24, 3, 3)
    bn("backbone.first_conv.1", 24)
    cin = 24
    for si, (rep, cout) in enumerate(zip(STAGE_REPEATS, (48, 96, 192))):
        for i in range(rep):
            p = "backbone.stage%d.%d" % (si + 2, i)
            mid = cout // 2
            inp = cin if i == 0 else cin // 2
            conv(p + ".branch_main.0", mid, inp, 1)
            bn(p + ".branch_main.1", mid)
            conv(p + ".branch_main.3", mid, mid, 3, groups=mid)
            bn(p + ".branch_main.4", mid)
            conv(p + ".branch_main.5", cout - inp, mid, 1)
            bn(p + ".branch_main.6", cout - inp)
            if i == 0:
                conv(p + ".branch_proj.0", inp, inp, 3, groups=inp)
                bn(p + ".branch_proj.1", inp)
                conv(p + ".branch_proj.2", inp, inp, 1)
                bn(p + ".branch_proj.3", inp)
            cin = cout
    conv("fpn.conv1x1_2.0", 72, 288, 1)
    bn("fpn.conv1x1_2.1", 72)
    conv("fpn.conv1x1_3.0", 72, 192, 1)
    bn("fpn.conv1x1_3.1", 72)
    for head in ("cls_head_2", "reg_head_2", "reg_head_3", "cls_head_3"):
        p = "fpn.%s.block" % head
        conv(p + ".0", 72, 72, 5, groups=72)
        bn(p + ".1", 72)
        conv(p + ".3", 72, 72, 1)
        bn(p + ".4", 72)
        conv(p + ".5", 72, 72, 5, groups=72)
        bn(p + ".6", 72)
        conv(p + ".8", 72, 72, 1)
        bn(p + ".9", 72)
    for name, co in (("output_reg_layers", 4 * anchor_num), ("output_obj_layers", anchor_num),
                     ("output_cls_layers", classes)):
        conv(name, co, 72, 1)
        w[name + ".bias"] = 0.1 * torch.randn(co, generator=g)
    return w


def export_weights(state, path):
    """Write a reference state_dict as the flat container include/yfv2.hpp's Detector::loadModel reads
    (the C++ host path has no pickle reader):  b"YFV2W1\\0\\0" | int32 n | n x { int32 name_len | name |
    int64 numel | numel x float32 }, little endian.  Integer buffers (num_batches_tracked) are skipped, exactly
    like Engine.load_state_dict."""
    import struct

    items = []
    for k, v in state.items():
        v = torch.as_tensor(v)
        if v.is_floating_point():
            items.append((k.encode(), v.detach().to("cpu", torch.float32).contiguous().numpy()))
    with open(path, "wb") as f:
        f.write(b"YFV2W1\0\0")
        f.write(struct.pack("<i", len(items)))
        for name, arr in items:
            f.write(struct.pack("<i", len(name)))
            f.write(name)
            f.write(struct.pack("<q", arr.size))
            f.write(arr.astype("<f4", copy=False).tobytes())
    return len(items)
