#!/bin/bash
# per-kernel durations of one DAUsingFK20 batch (scale 12, batch 128) for both scalar-multiplication variants
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for mode in fast wnaf; do
  out=$R/gpurun_out/fkstage_$mode; rm -rf $out
  KZG_HIP_G1_MUL=$mode KZG_HIP_FB_BUDGET_GB=10 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $R/bench.py --steps 2 --warmup 0 --batch 8 --no-cpu-baseline --fk20-multi-batch 1 > /dev/null 2>&1
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  python - "$f" $mode <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Grid_Size"]) if "Grid_Size" in r else 0) for r in rows if "k_g1_fft_stage" in r["Kernel_Name"]]
big = [x for x in d if x[1] >= 128 * 2048 * 1]
print(sys.argv[2], len(d), " ".join("%.2f" % x[0] for x in big[:24]))
PY
done
