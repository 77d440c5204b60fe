"""get_batch_statistics: device kernel vs the reference-shaped Python loop (oracle restatement) on a 256-image batch."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as orc
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B = 256
outs, tg = [], []
for b in range(B):
    n = int(rng.integers(20, 300))
    xy = rng.uniform(0, 300, (n, 2)); wh = rng.uniform(8, 120, (n, 2))
    d = np.concatenate((xy, xy + wh, np.sort(rng.uniform(0.01, 1, (n, 1)), 0)[::-1], rng.integers(0, 80, (n, 1))), 1).astype(np.float32)
    outs.append(torch.from_numpy(d))
    for k in rng.choice(n, 7, replace=False):
        tg.append([b, d[k, 5], *(d[k, :4] + rng.normal(0, 3, 4))])
tg = torch.tensor(np.asarray(tg, np.float32))
eng = yfv2.get_engine(dev, 352, 352, 80, 3)
dets = torch.zeros((B, 300, 6)); cnt = torch.zeros((B,), dtype=torch.int32)
for i, o in enumerate(outs):
    dets[i, :o.shape[0]] = o; cnt[i] = o.shape[0]
dets, cnt, tgd = dets.to(dev), cnt.to(dev), tg.to(dev)
eng.batch_statistics(dets, cnt, tgd, 0.5); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tp = eng.batch_statistics(dets, cnt, tgd, 0.5)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 20
t0 = time.perf_counter(); ref = orc.get_batch_statistics([o.numpy() for o in outs], tg.numpy(), 0.5); t_cpu = time.perf_counter() - t0
same = all(np.array_equal(tp[i, :outs[i].shape[0]].cpu().numpy(), ref[i][0].astype(np.int32)) for i in range(B))
print("batch of %d images, %d detections, %d targets: device %.3f ms per call (incl. launch+sync), numpy restatement of the reference loop %.1f ms; identical flags: %s"
      % (B, int(cnt.sum()), tg.shape[0], t_dev * 1e3, t_cpu * 1e3, same))
