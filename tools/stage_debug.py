"""Debug helper: stem / stage-2 activations of the current build vs the oracle (two random images): error pattern by channel, column, row."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as orc
w = orc.load_weights("tests/golden/weights_coco.npz")
torch.manual_seed(0)
x = torch.rand(2, 3, 352, 352)
m = yfv2.Detector(80, 3, True).to("cuda")
m.load_state_dict(w); m.eval()
m(x.cuda())
eng = m.engine_for(x.cuda())
ref = orc.forward_stages(w, x)
np.set_printoptions(linewidth=220)
for which, key in enumerate(("stem", "stage2", "c2")):
    r = ref[key].permute(0, 2, 3, 1).contiguous()
    got = eng.debug_activation(which, 2).view(r.shape)
    d = (got - r).abs()
    bad = d > 1e-4
    print("== %s: max err %.4g, bad fraction %.4f" % (key, d.max().item(), bad.float().mean().item()))
    if bad.any():
        print("bad by channel", bad.float().mean(dim=(0, 1, 2)).numpy().round(2))
        print("bad by column ", bad.float().mean(dim=(0, 1, 3)).numpy().round(2))
        print("bad by row    ", bad.float().mean(dim=(0, 2, 3)).numpy().round(2))
        print("got", got[0, 5, 5, :8].numpy().round(4)); print("ref", r[0, 5, 5, :8].numpy().round(4))
        break

# level-0 (stage2.0 output) data that survives in buffer 0 of the pair planes: pairs 0..8 and 12..14
import ctypes as C
from yolo_fastestv2_amd import _lib
def slot_channel(slot):
    p, e = slot >> 1, slot & 1
    b0, b1, b2, h = p // 12, (p % 12) // 6, (p % 6) // 3, 2 * (p % 3) + e
    return b0 + 2 * b1 + 4 * b2 + 8 * h
import torch.nn.functional as F
with torch.no_grad():
    y = orc._conv_bn(w, "backbone.first_conv.0", "backbone.first_conv.1", x, 2, 1, relu=True)
    y = F.max_pool2d(y, 3, 2, 1)
    s20 = orc._shuffle_block(w, "backbone.stage2.0", y, 2)   # (2,48,44,44)
per = 48 * 44 * 44
buf = torch.zeros(2 * 2 * per, dtype=torch.float32)
n = _lib.lib().yfv2_debug_activation(eng._h, 101, 2, C.c_void_p(buf.data_ptr()), buf.numel())
b0 = buf[:2 * per].view(2, 24, 44, 44, 2)
print("raw dump floats:", n)
for p in list(range(0, 9)) + [12, 13, 14]:
    for e in range(2):
        c = slot_channel(2 * p + e)
        d = (b0[:, p, :, :, e] - s20[:, c]).abs()
        print("pair %2d e%d = level-0 channel %2d (%s): max err %.4g  bad %.3f   bad cols %s" % (p, e, c, "proj" if c < 24 else "main", d.max().item(), (d > 1e-4).float().mean().item(),
              (d > 1e-4).float().mean(dim=(0, 1)).numpy().round(1)[:20]))

for p, e in ((0, 0), (0, 1), (12, 0)):
    c = slot_channel(2 * p + e)
    g = b0[0, p, :, :, e]; r = s20[0, c]
    print("channel", c, "got[10,10:16]", g[10, 10:16].numpy().round(4), "ref", r[10, 10:16].numpy().round(4), "mean diff %.4f" % (g - r).mean().item(),
          "corr %.4f" % np.corrcoef(g.flatten().numpy(), r.flatten().numpy())[0, 1])
# is it the main-branch taps / another channel order?  correlate got channel 0 with every reference channel
g = b0[0, 0, :, :, 0].flatten().numpy()
cors = [np.corrcoef(g, s20[0, c].flatten().numpy())[0, 1] for c in range(48)]
print("corr of slot(0,0) with ref channels:", np.array(cors).round(2))
