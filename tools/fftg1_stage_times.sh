#!/bin/bash
# per-stage kernel durations of batched FFT_G1 (scale 12, batch 64) for both scalar-multiplication variants
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for mode in fast wnaf; do
  out=$R/gpurun_out/stage_$mode; rm -rf $out
  KZG_HIP_G1_MUL=$mode rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $R/tools/fftg1_probe.py ${GB:-64} 2 2>/dev/null | tail -1
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  python - "$f" $mode <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_g1_fft_stage" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(sys.argv[2], len(d), " ".join("%.2f" % x for x in d[-12:]))
PY
done
