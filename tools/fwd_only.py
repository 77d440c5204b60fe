"""Forward-only driver for counter passes: 3 forwards at B=256 with random-init weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastestv2_amd as yfv2
dev = torch.device("cuda:0")
eng = yfv2.Engine(dev, 352, 352, 80, 3, max_batch=256)
eng.load_state_dict(yfv2.random_state_dict(0))
x = torch.rand(256, 3, 352, 352, device=dev)
for _ in range(3):
    eng.forward(x)
torch.cuda.synchronize()
