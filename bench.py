#!/usr/bin/env python3
"""bench.py - images/sec of the Yolo-FastestV2 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

A "step" is one pass of the whole hot path - forward (backbone + FPN + heads) ->
anchor decode -> class-aware NMS - over one batch of 256 synthetic 352x352x3 fp32
images per GPU that are already resident in HBM, plus (N > 1) the RCCL
all-gather of the padded detections.  BASELINE.json configs[1] (forward only) is
a strict subset of the timed work; its rate is reported as `forward_only_img_s`.
Weights are seeded random-init of the reference architecture (no checkpoint
download is possible); data is torch.rand, so NMS sees few candidates - the
NMS-heavy case is covered by tests, not by this number.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings):
  value      whole-job images/s (all ranks' images / max-over-ranks time)
  roofline   the dominant launch of the forward: algorithmic bytes (or flops) per
             launch / its mean duration, measured with hipEvent pairs on the
             launch stream inside this process (yfv2_profile_forward)
  cpu_baseline  the CPU oracle (same ATen CPU ops as the reference + numpy
             decode/NMS) timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # dense fp32 MFMA peak (= fp32 vector peak)
ANCHORS = [12.64, 19.39, 37.88, 51.48, 55.71, 138.31, 126.91, 78.23, 131.57, 214.55, 279.92, 258.87]  # data/coco.data:17


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE.json: 256)")
    ap.add_argument("--conf", type=float, default=0.3)
    ap.add_argument("--iou", type=float, default=0.4)
    ap.add_argument("--profile-iters", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--weights", choices=("random", "coco"), default="random")
    return ap.parse_args()


def measured_traffic(stage_name):
    """HBM bytes per launch of the kernel behind a forward stage, from the newest committed PMC
    summary (profiles/*_traffic.json, produced by tools/gpu_traffic.sh + tools/traffic_summary.py:
    FETCH_SIZE and WRITE_SIZE in separate --pmc passes, gfx950 half-read correction, calibrated on a
    known-size copy).  PMC counters cannot be collected from inside this process -> None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    kern = json.load(open(files[-1])).get("kernels", {})
    n = stage_name
    if n.startswith("stem"):
        key = "stem_"                      # stem_px_kernel<...> (stem_kernel in rounds before r01p)
    elif "s2 block, lane-per-pixel" in n:
        key = "s2px_kernel"
    elif "s1 block, lane-per-pixel" in n:
        key = "s1px_kernel"
    elif "fused s2 block" in n:
        key = "block_s2_kernel<24" if "stage2" in n else "block_s2_kernel<48"
    elif "fused s1 block" in n:
        key = "block_s1_kernel<24" if "stage2" in n else ("block_s1_kernel<48" if "stage3" in n else "block_s1_kernel<96")
    else:
        return None, os.path.basename(files[-1])
    for k, v in kern.items():
        if key in k:
            return v["total_bytes"], os.path.basename(files[-1])
    return None, os.path.basename(files[-1])


def timed(fn, steps, sync, barrier):
    barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync(); barrier()
    return time.perf_counter() - t0


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world
    import torch.distributed as dist
    import yolo_fastestv2_amd as yfv2

    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    under_launcher = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if world > 1 or under_launcher:  # under torch.distributed.run even N=1 goes through RCCL (exercises the N>1 code path)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    use_dist = dist.is_initialized()

    def barrier():
        if use_dist:
            dist.barrier()

    def sync():
        torch.cuda.synchronize(dev)

    # ---- model + data ---------------------------------------------------------------------
    if a.weights == "coco":
        import numpy as np
        z = np.load(os.path.join(REPO, "tests", "golden", "weights_coco.npz"))
        sd = {k: torch.from_numpy(z[k]) for k in z.files}
    else:
        sd = yfv2.random_state_dict(0)
    eng = yfv2.Engine(dev, 352, 352, 80, 3, anchors=ANCHORS, max_batch=a.batch)
    eng.load_state_dict(sd)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.rand(a.batch, 3, 352, 352, device=dev, generator=g)  # resident in HBM before timing
    det_bufs = eng.new_det_buffers(a.batch)
    logit_bufs = [torch.empty(s, device=dev) for s in eng.logit_shapes(a.batch)]

    def step():
        d, i, c = eng.detect(x, a.conf, a.iou, out=det_bufs)
        if use_dist:
            yfv2.gather_detections(d, i, c, force=True)

    for _ in range(a.warmup):
        step()
    dt = timed(step, a.steps, sync, barrier)
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    value = world * a.batch * a.steps / dt

    # forward only (BASELINE.json configs[1]) - same batch, same buffers
    for _ in range(2):
        eng.forward(x, out=logit_bufs)
    dt_f = timed(lambda: eng.forward(x, out=logit_bufs), a.steps, sync, barrier)
    fwd_img_s = world * a.batch * a.steps / dt_f
    # the same forward fed with uint8 (B,H,W,3) images (yfv2_forward_u8: test.py:34-38's pre-process inside the stem) -
    # an extra, never `value`: the contract's input is the fp32 tensor
    xu = (x.permute(0, 2, 3, 1) * 255.0).round().clamp(0, 255).to(torch.uint8).contiguous()
    for _ in range(2):
        eng.forward(xu, out=logit_bufs)
    dt_u = timed(lambda: eng.forward(xu, out=logit_bufs), a.steps, sync, barrier)
    fwd_u8_img_s = world * a.batch * a.steps / dt_u
    del xu

    cnt_h = det_bufs[2].float().cpu()
    out = None
    if rank == 0:
        # ---- roofline of the dominant launch (hipEvents on the launch stream) -------------
        stages = eng.stages()
        ms = eng.profile_forward(x, iters=a.profile_iters)
        kern = []
        for s, m in zip(stages, ms):
            is_mfma = any(t in s["name"] for t in (" pw", ".pw", "output_", "conv1x1", "stem", "lane-per-pixel"))   # launches whose conv runs on the MFMA
            by, fl = s["bytes_per_image"] * a.batch, s["flops_per_image"] * a.batch
            kern.append({"name": s["name"], "ms": m, "gbs": by / (m * 1e-3) / 1e9 if m > 0 else 0.0,
                         "tflops": fl / (m * 1e-3) / 1e12 if m > 0 else 0.0, "mfma": is_mfma, "bytes": by, "flops": fl})
        dom = max(kern, key=lambda k: k["ms"])
        # every launch of this net is left of the fp32 ridge (19.7 flop/B) unfused, so price
        # the dominant launch against HBM; pointwise launches also carry their MFMA fraction
        traffic, traffic_src = measured_traffic(dom["name"])
        roof = {"kernel": dom["name"], "bound": "hbm", "achieved": round(dom["gbs"], 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(dom["gbs"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(dom["ms"], 4), "algorithmic_bytes_per_launch": dom["bytes"],
                "mfma_tflops": round(dom["tflops"], 2) if dom["mfma"] else None,
                "mfma_frac": round(dom["tflops"] / MFMA_F32_PEAK_TF, 4) if dom["mfma"] else None}
        tot_ms = sum(k["ms"] for k in kern)
        groups = {}
        for k in kern:
            n = k["name"]
            key = ("stem" if n.startswith("stem") else "dw3x3" if "dw3x3" in n else "dw5x5" if "dw5x5" in n else "pw1x1(mfma)")
            gq = groups.setdefault(key, {"ms": 0.0, "bytes": 0.0, "flops": 0.0, "launches": 0})
            gq["ms"] += k["ms"]; gq["bytes"] += k["bytes"]; gq["flops"] += k["flops"]; gq["launches"] += 1
        for gq in groups.values():
            gq["gbs"] = round(gq["bytes"] / (gq["ms"] * 1e-3) / 1e9, 1)
            gq["hbm_frac"] = round(gq["gbs"] / HBM_PEAK_GBS, 4)
            gq["tflops"] = round(gq["flops"] / (gq["ms"] * 1e-3) / 1e12, 2)
            gq["ms"] = round(gq["ms"], 4)
            del gq["bytes"], gq["flops"]

        # ---- CPU baseline: the oracle on this box's host cores, bounded sample ------------
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            from oracle import yfv2_oracle as oracle
            # the reference's CPU path is ATen/oneDNN on all host cores; on a 256-thread
            # box that oversubscribes these tiny convs badly, so probe a few thread counts
            # on a small batch and time the bounded sample with the fastest one
            ncores = os.cpu_count() or 1
            try:
                ncores = min(ncores, len(os.sched_getaffinity(0)))
            except AttributeError:
                pass
            bs = 64
            xc = x[:bs].cpu()
            cands = sorted({c for c in (8, 16, 32, 64, 128, ncores) if c <= ncores})
            best_t, best_rate = cands[0], 0.0
            for c in cands:
                torch.set_num_threads(c)
                oracle.forward(sd, xc[:4])  # warm-up at this thread count
                t0 = time.perf_counter()
                oracle.detect(sd, xc[:16], ANCHORS, 352, a.conf, a.iou)
                el = time.perf_counter() - t0
                if 16 / el > best_rate:
                    best_t, best_rate = c, 16 / el
                if el > 4.0:
                    break  # larger counts only get slower from here
            torch.set_num_threads(best_t)
            n_img, t0 = 0, time.perf_counter()
            while True:
                oracle.detect(sd, xc, ANCHORS, 352, a.conf, a.iou)
                n_img += bs
                el = time.perf_counter() - t0
                if el >= a.cpu_seconds or n_img >= 8192:
                    break
            cpu = {"value": round(n_img / el, 1), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                   "host_threads_available": ncores,
                   "sample": "%d synthetic images in batches of %d through oracle forward(ATen CPU)+decode+NMS, %.1f s, "
                             "thread count chosen by a 16-image probe over %s" % (n_img, bs, el, cands)}

        out = {
            "metric": "images/sec at 352x352 bs=256 per GPU (forward+decode+NMS)", "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batch %d/GPU synthetic 352x352x3 fp32 (torch.rand), seeded random-init weights, "
                                   "Detector forward + anchor decode + class-aware NMS(conf %.2f, iou %.2f)%s; BASELINE.json "
                                   "configs[1] (forward only) is the subset reported in forward_only_img_s"
                                   % (a.batch, a.conf, a.iou, " + RCCL all-gather of padded detections" if use_dist else ""),
                       "global_batch": world * a.batch, "weights": a.weights, "parallelism": "batch-sharded x%d" % world},
            "forward_only_img_s": round(fwd_img_s, 1), "forward_only_ms": round(1e3 * dt_f / a.steps, 4),
            "forward_from_uint8_hwc_img_s": round(fwd_u8_img_s, 1), "forward_from_uint8_hwc_ms": round(1e3 * dt_u / a.steps, 4),
            "roofline": roof, "cpu_baseline": cpu,
            "detections_per_image": {"mean": round(float(cnt_h.mean()), 1), "max": int(cnt_h.max())},
            "forward_launches": len(kern), "forward_sum_of_launch_ms": round(tot_ms, 4), "kernel_groups": groups,
        }
        print(json.dumps(out), flush=True)
    barrier()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
