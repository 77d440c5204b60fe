#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of the default `bench.py` run: how many kernels execute at the same time inside the
timed region of the pipelined `value` leg (the last 20 of the first 80 + 5 + 20 steps that rotate over the three queues).
usage: python tools/overlap_from_trace.py gpurun_out/r03k/prof/r03k_kernel_trace.csv [spinup warmup steps]"""
import collections, csv, statistics, sys
path = sys.argv[1]
spin, warm, steps = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (80, 5, 20)
rows = list(csv.DictReader(open(path)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows)
print("dispatches per queue:", dict(collections.Counter(e[2] for e in ev)))
nms = [e for e in ev if "nms_kernel" in e[3]]              # one per step: the step's last launch
seq = [e[2] for e in nms]
start = next(i for i in range(len(seq) - 6) if len(set(seq[i:i + 3])) == 3 and seq[i:i + 3] == seq[i + 3:i + 6])
blk = nms[start:start + spin + warm + steps]
t_lo, t_hi = blk[spin + warm - 1][1], blk[spin + warm + steps - 1][1]
print("timed window: %.3f ms for %d steps = %.4f ms per step (under the profiler)" % ((t_hi - t_lo) / 1e6, steps, (t_hi - t_lo) / 1e6 / steps))
inside = [e for e in ev if e[1] > t_lo and e[0] < t_hi and "kernel" in e[3]]
pts = sorted([(max(s, t_lo), 1) for s, e, q, n in inside] + [(min(e, t_hi), -1) for s, e, q, n in inside])
act, last, hist = 0, t_lo, collections.Counter()
for t, d in pts:
    hist[act] += t - last; last = t; act += d
hist[act] += t_hi - last
tot = sum(hist.values())
print("share of the window with k kernels executing:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print("sum of kernel durations / window = %.2f" % (sum(min(e, t_hi) - max(s, t_lo) for s, e, q, n in inside) / tot))
for key in ("stem_h3_kernel", "s2h_kernel", "s1h_kernel", "block_s1chain6", "towerh_kernel<6", "nms_kernel"):
    d = [(e - s) / 1e3 for s, e, q, n in inside if key in n]
    if d:
        print("%-18s n = %3d  mean %.1f us while overlapped" % (key, len(d), statistics.mean(d)))
