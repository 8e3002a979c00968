#!/bin/bash
# sweeps the fixed-base table budget (c = 13, 14, 15) for the commitment bench
for g in 70 115 210; do
  echo -n "budget_gb=$g "
  KZG_HIP_FB_BUDGET_GB=$g python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fk20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
done
