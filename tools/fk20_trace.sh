#!/bin/bash
# kernel-by-kernel timeline of ONE DAUsingFK20 batch (scale 12) at the bench's batch size
R=$(pwd); cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/fk20trace; rm -rf $out
KZG_HIP_FB_BUDGET_GB=4 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $R/bench.py --steps 2 --warmup 0 --batch 8 --no-cpu-baseline --fk20-multi-batch 0 > /dev/null 2>&1
python - "$(find $out -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last FK20 step = kernels after the last k_toeplitz_coeffs launch
idx = max(i for i, r in enumerate(rows) if "k_toeplitz_coeffs" in r["Kernel_Name"])
step = rows[idx:]
end = next((i for i, r in enumerate(step) if "k_fr_fft_tile" in r["Kernel_Name"] and i > 3), len(step))
agg = collections.OrderedDict()
for r in step[:end if end > 10 else len(step)]:
    k = r["Kernel_Name"].split("(")[0].replace("kzg::", "")[:40]
    agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in agg.values())
for k, v in agg.items(): print("%-42s x%3d %9.3f ms %5.1f %%" % (k, v[0], v[1], 100 * v[1] / tot))
print("total %.1f ms" % tot)
PY
