#!/bin/bash
# A/B of the two table-walk variants (plain XYZZ accumulation vs batch-affine levels), interleaved on one box
for rep in 1 2; do for m in xyzz ba; do
  echo -n "mode=$m "
  KZG_HIP_FB_MODE=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fk20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
done; done
