#!/bin/bash
# A/B of the two table-walk variants (plain XYZZ accumulation vs batch-affine levels) and of the lane count, interleaved on one box
for rep in 1 2; do for cfg in "ba 131072" "xyzz 131072" "xyzz 98304" "xyzz 65536"; do
  set -- $cfg
  echo -n "mode=$1 lanes=$2 "
  KZG_HIP_FB_MODE=$1 KZG_HIP_FB_LANES=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fk20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
done; done
