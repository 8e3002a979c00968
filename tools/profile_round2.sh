#!/bin/bash
# rocprofv3 passes behind profiles/r02_*: (1) kernel trace + stats of the bench, (2) one --pmc pass per counter (the counter
# passes carry no other trace domain than --kernel-trace) for the two dominant kernels: k_fb_accumulate (commitment step, 512
# blobs, c = 16 table) and k_g1_fft_stage (FK20 step, 512 polynomials).  usage (GPU box): bash tools/profile_round2.sh <tag>
tag=${1:-r02}
R=$(pwd); out=$R/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH_FK="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --fk20-multi-batch 0 --fk20-batch 512"   # FK20 step: 512 polynomials
BENCH_FB="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fk20 --batch 512"                               # commitment step only: every launch has 512 blobs
rocprofv3 --kernel-trace --stats -d $out/trace -o $tag -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --batch 512 --fk20-batch 512 --fk20-multi-batch 256 > $out/trace_bench.json 2> $out/trace_err.txt
db=$(find $out/trace -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $db $out/kernel_stats.md > /dev/null
for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES; do
 for which in FB FK; do
  if [ $which = FB ]; then cmd=$BENCH_FB; else cmd=$BENCH_FK; fi
  rm -rf $out/pmc_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -o $tag -- $cmd > /dev/null 2> $out/pmc_${ctr}_err.txt
  f=$(find $out/pmc_$ctr -name "*counter_collection.csv" | head -1)
  python - "$f" $ctr $which <<'PY' >> $out/pmc_rows.jsonl
import csv, json, sys
ctr = sys.argv[2]
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != ctr: continue
    name = r["Kernel_Name"]
    key = None
    if sys.argv[3] == "FB" and "k_fb_accumulate" in name and int(r["Grid_Size"]) == 131072: key = "k_fb_accumulate"
    elif sys.argv[3] == "FK" and "k_g1_fft_stage" in name and int(r["Grid_Size"]) == 1048576: key = "k_g1_fft_stage"
    if key is None: continue
    a = acc.setdefault(key, {"n": 0, "sum": 0.0, "scratch": int(r["Scratch_Size"]), "vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"])})
    a["n"] += 1; a["sum"] += float(r["Counter_Value"])
for k, a in acc.items():
    print(json.dumps({"kernel": k, "counter": ctr, "launches": a["n"], "avg_per_launch": a["sum"] / a["n"], "scratch_bytes_per_lane": a["scratch"], "vgprs": a["vgpr"], "lds": a["lds"]}))
PY
 done
done
# the un-profiled line at the SAME launch shapes as the counter passes (512 blobs / 512 polynomials per step; bench.py's defaults are larger)
cd $R && python bench.py --steps 10 --warmup 3 --batch 512 --fk20-batch 512 --fk20-multi-batch 256 2>/dev/null | tail -1 > $out/bench_line.json
rm -rf $out/trace/*/*.db $out/pmc_*/*agent_info.csv $out/pmc_*/*kernel_trace.csv $out/pmc_*/*counter_collection.csv 2>/dev/null
cat $out/pmc_rows.jsonl
du -sh $out
