#!/bin/bash
# Same-box A/B of the side streams for the 11x11 towers (YFV2_SIDE=1) + the kernel timeline of one forward with them on
for rep in 1 2 3; do
  echo "-- YFV2_SIDE=1 (#$rep)"; YFV2_SIDE=1 timeout 200 python tools/order_probe.py 30 2>&1 | tail -1
  echo "-- default: one stream (#$rep)"; timeout 200 python tools/order_probe.py 30 2>&1 | tail -1
done
YFV2_SIDE=1 bash tools/overlap_probe.sh | tail -12
