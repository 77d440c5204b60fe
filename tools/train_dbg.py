"""per-parameter table of one training iteration on the device against the float64 oracle replayed on the device's own
ReLU decisions (the comparison tests/test_train_gpu.py asserts); `python tools/train_dbg.py CASE`"""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden"); sys.path.insert(0, "tests")
import make_golden
import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as oracle
dev = torch.device("cuda:0")
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 0
classes, B, T, seed, lr = make_golden.TRAIN_CASES[ci]
w, x, t = make_golden.train_case_inputs(classes, B, T, seed)
anchors = [float(a) for a in np.load("tests/golden/cfg_coco.npz")["anchors"]]
cfg = {"anchor_num": 3, "classes": classes, "width": 352, "height": 352, "anchors": anchors}
model = yfv2.Detector(classes, 3, True).to(dev); model.load_state_dict({k: v.clone() for k, v in w.items()}); model.train()
preds = model(torch.from_numpy(x).to(dev))
losses = yfv2.compute_loss(preds, torch.from_numpy(t).to(dev), cfg, dev)
losses[3].backward()
w64 = {k: (v.double() if v.is_floating_point() else v) for k, v in w.items()}
free = oracle.train_step(w64, torch.from_numpy(x).double(), torch.from_numpy(t), anchors, classes, lr)
eng = preds[0]._yfv2_engine
dec = {}
for name, pre in free["pre_relu"].items():
    d = eng.debug_train_relu_output(name).reshape(pre.shape) > 0
    differ = d != (pre > 0)
    if differ.any():
        print("decisions differ: %-40s %d  max |pre| %.2e" % (name, int(differ.sum()), float(pre[differ].abs().max())))
    dec[name] = d
r64 = oracle.train_step(w64, torch.from_numpy(x).double(), torch.from_numpy(t), anchors, classes, lr, relu_decisions=dec)
r32 = oracle.train_step(w, torch.from_numpy(x), torch.from_numpy(t), anchors, classes, lr, relu_decisions=dec)
grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
G = max(float(v.abs().max()) for v in r64["grads"].values())
for k, p in model.named_parameters():
    t64, ref, d = r64["grads"][k].numpy(), r32["grads"][k].numpy(), grads[k].numpy()
    e_ref, e_dev, sc = np.abs(ref - t64).max(), np.abs(d - t64).max(), np.abs(t64).max()
    flag = "BAD" if e_dev > 3 * e_ref + 1e-6 * G else ""
    print("%-48s scale %.3e  ref %.2e  dev %.2e  ratio %7.2f %s" % (k, sc, e_ref / max(sc, 1e-30), e_dev / max(sc, 1e-30), e_dev / max(e_ref, 1e-30), flag))
