#!/bin/bash
# FK20Multi (scale 16, chunk 16) throughput vs batch per GPU
for b in 128 256; do
  echo -n "fk20_multi_batch=$b "
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --fk20-multi-batch $b 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fk20_multi']['value'], d['fk20_multi']['ms_per_all_proofs'])"
done
