#!/bin/bash
# A/B of the GLV scalar-multiplication variants (KZG_HIP_G1_MUL: fast = signed 5-bit windows, wnaf = width-5 NAF with product calls,
# inl = + inlined doubling products, anything else = default, all products inlined) in the G1 FFT stages: FK20 (batch 128), FK20Multi (batch 128), FFT_G1 (batch 64)
for mode in fast wnaf inl all default; do
  echo -n "$mode "
  KZG_HIP_G1_MUL=$mode KZG_HIP_FB_BUDGET_GB=10 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fk20']['value'], d['fk20_multi']['value'], d['reference_benchmarks']['fft_g1_scale12_per_s']['value'])"
done
