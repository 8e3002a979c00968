#!/bin/bash
# instruction / stall counters of k_fb_accumulate (commitment step, 512 blobs), one counter group per pass
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  out=$R/gpurun_out/pmcc; rm -rf $out
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fk20 > /dev/null 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_fb_accumulate" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "avg per launch %.5g" % (sum(v) / len(v)), "launches", len(v))
PY
done
