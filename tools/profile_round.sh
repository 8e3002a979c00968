#!/bin/bash
# rocprofv3 passes behind profiles/*: kernel trace + stats of the whole bench, then FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU /
# GRBM_GUI_ACTIVE of the commitment step
# (counter passes are separate and carry no other trace domain).  usage (GPU box): bash tools/profile_round.sh <tag>
tag=${1:-vX}
R=$(pwd); out=$R/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o $tag -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/trace_bench.json 2> $out/trace_err.txt
db=$(find $out/trace -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $db $out/kernel_stats.md > /dev/null
for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -o $tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fk20 > /dev/null 2> $out/pmc_${ctr}_err.txt
  f=$(find $out/pmc_$ctr -name "*counter_collection.csv" | head -1)
  python - "$f" $ctr <<'PY' | tee -a $out/pmc_summary.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_fb_accumulate" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
v = [float(r["Counter_Value"]) for r in rows]
print(sys.argv[2], "k_fb_accumulate launches", len(v), "avg", sum(v) / max(1, len(v)))
PY
done
cd $R && python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $out/bench_line.json
rm -rf $out/trace/*/*.db $out/pmc_*/*/*agent_info.csv 2>/dev/null
du -sh $out
