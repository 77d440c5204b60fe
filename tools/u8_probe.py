"""Stem timing with fp32 NCHW vs uint8 NHWC input at B=256 (per-launch probe is fp32-only: time whole forwards)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
dev = torch.device("cuda:0")
B = 256
eng = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=B)
eng.load_state_dict(yfv2.random_state_dict(0))
xu = torch.randint(0, 256, (B, 352, 352, 3), dtype=torch.uint8, device=dev)
xf = (xu.permute(0, 3, 1, 2).float() / 255.0).contiguous()
for name, x in (("fp32 NCHW", xf), ("uint8 NHWC", xu), ("fp32 NCHW", xf), ("uint8 NHWC", xu)):
    for _ in range(3):
        eng.forward(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.forward(x)
    e1.record(); torch.cuda.synchronize()
    print("%-11s forward %.4f ms  (%.0f img/s)" % (name, e0.elapsed_time(e1) / 20, B / (e0.elapsed_time(e1) / 20) * 1e3))
a = eng.forward(xf); b = eng.forward(xu)
print("max |logit diff| fp32-vs-u8 path:", max(float((p - q).abs().max()) for p, q in zip(a, b)))
