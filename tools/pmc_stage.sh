#!/bin/bash
# VALU / scratch instruction counters of k_g1_fft_stage for both scalar-multiplication variants (one counter group per pass)
R=$(pwd); cd /tmp && export TMPDIR=/tmp
if [ "$1" = "list" ]; then rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|name)\s*:\s*SQ_[A-Z_0-9]+" | sort -u | head -200; rocprofv3 -L 2>/dev/null | grep -c SQ_; exit; fi
for mode in fast default; do
 for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  out=$R/gpurun_out/pmcs_$mode; rm -rf $out
  KZG_HIP_G1_MUL=${mode/default/} rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o t -- python $R/tools/fftg1_probe.py ${GB:-64} 1 > /dev/null 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" $mode <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_g1_fft_stage" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(sys.argv[2], k, "last-stage launches: %.4g" % (sum(v[-1:]) ), "max %.4g" % max(v))
PY
 done
done
