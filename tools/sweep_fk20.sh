#!/bin/bash
# FK20 single (scale 12) throughput vs batch per GPU
for b in 128 256 512; do
  echo -n "fk20_batch=$b "
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --fk20-multi-batch 0 --fk20-batch $b 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fk20']['value'], d['fk20']['ms_per_all_proofs'], d['reference_benchmarks']['fft_g1_scale12_per_s']['value'])"
done
