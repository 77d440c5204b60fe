"""Debug helper: stem output of the current YFV2_STEM variant vs the oracle's stem (two random images)."""
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
got = eng.debug_activation(0, 2).view(2, 88, 88, 24)
ref = orc.forward_stages(w, x)["stem"].permute(0, 2, 3, 1).contiguous()
d = (got - ref).abs()
np.set_printoptions(linewidth=200)
print("max err", d.max().item(), "mean", d.mean().item())
bad = (d > 1e-4)
print("bad fraction", bad.float().mean().item())
print("bad by channel", bad.float().mean(dim=(0, 1, 2)).numpy().round(2))
print("bad by column", bad.float().mean(dim=(0, 1, 3)).numpy().round(2))
print("bad by row", bad.float().mean(dim=(0, 2, 3)).numpy().round(2))
print("got[0,5,5,:8]", got[0, 5, 5, :8].numpy().round(4)); print("ref[0,5,5,:8]", ref[0, 5, 5, :8].numpy().round(4))
