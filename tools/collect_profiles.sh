#!/bin/bash
# After a tools/gpu_r4.sh TAG call: copy the summaries gpurun merged back under gpurun_out/TAG into profiles/ (tracked),
# optionally marked (e.g. "slowbox").  usage (in the build container, repo root): bash tools/collect_profiles.sh TAG [MARK]
TAG=${1:?tag}; MARK=${2:+_$2}; O=gpurun_out/$TAG; P=profiles/${TAG}${MARK}
[ -d "$O" ] || { echo "no $O"; exit 1; }
tail -1 $O/bench.log > ${P}_bench.json
tail -30 $O/pytest_gpu.log > ${P}_pytest_gpu_tail.txt
for f in kernel_stats.csv parity_counts.json pmc_sq_summary.txt scale_probe.txt src_hash.txt lanes_probe.txt post_phases.txt train_probe.txt stem_pattern.txt power_probe.txt traffic.json pmc.json; do
  [ -f $O/$f ] && cp $O/$f ${P}_$f
done
[ -f $O/smoke.log ] && cp $O/smoke.log ${P}_smoke.txt
ls ${P}_* | wc -l
