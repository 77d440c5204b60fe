#!/usr/bin/env python3
"""Wall time of one training iteration (train-mode forward, compute_loss, backward, SGD step) through the drop-in surface,
at a few batch sizes, next to the CPU oracle's (= the reference's ATen CPU ops) on this box: python tools/train_probe.py [B ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as oracle

dev = torch.device("cuda:0")
anchors = [float(a) for a in np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cfg_coco.npz"))["anchors"]]
cfg = {"anchor_num": 3, "classes": 80, "width": 352, "height": 352, "anchors": anchors}
for B in [int(b) for b in sys.argv[1:]] or [8, 64]:
    w = yfv2.random_state_dict(1)
    model = yfv2.Detector(80, 3, True).to(dev); model.load_state_dict(w); model.train()
    opt = yfv2.SGD(params=model.parameters(), lr=1e-3, momentum=0.949, weight_decay=0.0005)
    rng = np.random.default_rng(B)
    x = torch.from_numpy(rng.random((B, 3, 352, 352), dtype=np.float32)).to(dev)
    T = 4 * B
    t = np.zeros((T, 6), np.float32); t[:, 0] = rng.integers(0, B, T); t[:, 1] = rng.integers(0, 80, T)
    t[:, 2:4] = rng.random((T, 2)) * 0.9 + 0.05; t[:, 4:6] = rng.random((T, 2)) * 0.5 + 0.03
    tt = torch.from_numpy(t).to(dev)
    def step():
        preds = model(x)
        loss = yfv2.compute_loss(preds, tt, cfg, dev)[3]
        loss.backward(); opt.step(); opt.zero_grad()
        return loss
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.time(); n = 5
    for _ in range(n): l = step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    line = "B=%d: %.1f ms per iteration on the device (%.0f images/s), loss %.3f" % (B, 1e3 * dt, B / dt, float(l.detach()))
    if B <= 8:
        t0 = time.time(); oracle.train_step(w, x.cpu(), torch.from_numpy(t), anchors, 80, 1e-3); line += "; CPU oracle %.0f ms" % (1e3 * (time.time() - t0))
    print(line)
