#!/bin/bash
# per-launch durations of the stage kernels of one DAUsingFK20 step at several batch sizes (kernel trace): where a small batch loses against a large one
# usage (GPU box): bash tools/fk20_stage_trace.sh "8 32 64 256 1024"
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for b in ${1:-8 32 64 256 1024}; do
  out=$R/gpurun_out/fkstage_b$b; rm -rf $out
  KZG_HIP_FB_BUDGET_GB=10 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $R/bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline --no-extras --fk20-multi-batch 0 --fk20-batch $b > /dev/null 2>&1
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  python - "$f" $b <<'PY'
import csv, sys
b = int(sys.argv[2])
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gk = "Grid_Size" if "Grid_Size" in rows[0] else ("Grid_Size_X" if "Grid_Size_X" in rows[0] else None)
if gk is None:
    print(sorted(rows[0].keys())); sys.exit(0)
st = [r for r in rows if "k_g1_fft_stage" in r["Kernel_Name"] and int(r[gk]) in (b * 2048, 4 * b * 2048)]   # exactly this step's launches (the bench also runs 64 plain transforms)
last = st[-22:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in last]
gaps = [(int(last[i + 1]["Start_Timestamp"]) - int(last[i]["End_Timestamp"])) / 1e6 for i in range(len(last) - 1)]
print("batch %d: %d stage launches, sum %.2f ms (%.3f ms per polynomial), per launch: %s; gaps between them: max %.3f ms, sum %.3f ms" %
      (b, len(d), sum(d), sum(d) / b, " ".join("%.2f" % x for x in d), max(gaps), sum(gaps)))
PY
  rm -rf $out
done
