"""Data-parallel training path on the one GPU a test box has: a one-rank nccl (= RCCL) process group, Detector.data_parallel(
force=True) so the all-reduce of the gradient bucket is really issued on the device buffer, one iteration of train.py:101-123
- gradients and updated weights must equal the same iteration without the collective (the mean over one rank).  Run by
tests/test_train_gpu.py in its own interpreter (a process group is process-global state)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "golden"))
import make_golden  # noqa: E402  (seeded inputs only)
import yolo_fastestv2_amd as yfv2  # noqa: E402


def iteration(dp, w, x, t, cfg, dev, calls):
    model = yfv2.Detector(cfg["classes"], 3, True).to(dev)
    model.load_state_dict({k: v.clone() for k, v in w.items()})
    model.train()
    if dp:
        model.data_parallel(force=True)
    opt = yfv2.SGD(params=model.parameters(), lr=0.001, momentum=0.949, weight_decay=0.0005)
    n0 = len(calls)
    losses = yfv2.compute_loss(model(x), t, cfg, dev)
    losses[3].backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    return losses, grads, {k: v.detach().clone() for k, v in model.state_dict().items()}, len(calls) - n0


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29651", rank=0, world_size=1)
    calls = []
    real = dist.all_reduce

    def counting(tensor, *a, **k):
        calls.append((tensor.device.type, tensor.numel()))
        return real(tensor, *a, **k)
    dist.all_reduce = counting
    classes, B, T, seed, lr = make_golden.TRAIN_CASES[0]
    w, x, t = make_golden.train_case_inputs(classes, B, T, seed)
    anchors = [float(a) for a in np.load(os.path.join(HERE, "..", "golden", "cfg_coco.npz"))["anchors"]]
    cfg = {"anchor_num": 3, "classes": classes, "width": 352, "height": 352, "anchors": anchors}
    xs, ts = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    l0, g0, a0, n0 = iteration(False, w, xs, ts, cfg, dev, calls)
    l1, g1, a1, n1 = iteration(True, w, xs, ts, cfg, dev, calls)
    assert n0 == 0 and n1 == 1, (n0, n1)                                  # ONE collective per backward ...
    n_par = sum(v.numel() for v in g0.values())
    assert calls[-1] == ("cuda", n_par) and len(g0) == 225, calls[-1]    # ... over the whole bucket, on the device
    # float atomics in the weight-gradient kernel make two runs differ in the last bits: same bound as between two plain runs
    for k in g0:
        sc = float(g0[k].abs().max()) + 1e-12
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-4 * sc, k
    for k in a0:
        assert torch.allclose(a0[k].float(), a1[k].float(), rtol=1e-5, atol=1e-6), k
    assert all(abs(float(a) - float(b)) <= 1e-6 * max(1.0, abs(float(a))) for a, b in zip(l0, l1))
    dist.barrier()
    dist.destroy_process_group()
    print("train_dp ok: one all-reduce of %d floats per backward" % n_par)


if __name__ == "__main__":
    main()
