#!/bin/bash
# rocprofv3 passes behind profiles/r06_*, at the launch shapes bench.py TIMES by default (4096-blob commitment step, 1024-polynomial FK20
# step, 512-polynomial FK20 step on 4096-element blobs, 1024 F_r transforms per launch):
#   (1) kernel trace of `bench.py --no-cpu-baseline --no-extras --no-in-process` (no table sweep: every k_fb_accumulate launch of 4096 workgroups
#       walks the headline's 16-bit-window table (8 windows walked by both GLV halves, 103 GB)), summarised PER LAUNCH SHAPE (kernel, grid, workgroup) by tools/rocprof_summary.py:
#       kernel_stats.md + kernel_shapes.json -- bench.py reads the latter (roofline.profile_avg_ms);
#   (2) unless SKIP_PMC is set: counter passes (no other trace domain than --kernel-trace) over three commands:
#       FB = commitment step only, FK = FK20 step only, FR = tools/fr_probe.py (k_fr_fft4096_r4, k_das_ext2048_r4);
#       passes: FETCH_SIZE | WRITE_SIZE | SQ group 1 | SQ group 2 (+ GRBM_GUI_ACTIVE)
# usage (GPU box): bash tools/profile_round6.sh <tag>
tag=${1:-r06}
R=$(pwd); out=$R/gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD_FB="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fk20 --no-in-process"
CMD_FK="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-in-process --fk20-multi-batch 0 --fk20-4096-batch 0 --batch 512"
CMD_FR="env KZG_FR_PROBE_NO_AB=1 python $R/tools/fr_probe.py 1024"
if [ -z "$SKIP_TRACE" ]; then
rocprofv3 --kernel-trace --stats -d $out/trace -o $tag -- python $R/bench.py --no-cpu-baseline --no-extras --no-in-process > $out/trace_bench.json 2> $out/trace_err.txt
db=$(find $out/trace -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $db $out/kernel_stats.md $out/kernel_shapes.json > /dev/null
fi
if [ -z "$SKIP_PMC" ]; then
P1="FETCH_SIZE"; P2="WRITE_SIZE"
P3="SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
P4="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM"
i=0
for pass in "$P1" "$P2" "$P3" "$P4"; do
 i=$((i+1))
 for which in FB FK FR; do
  if [ $which = FB ]; then cmd=$CMD_FB; elif [ $which = FK ]; then cmd=$CMD_FK; else cmd=$CMD_FR; fi
  d=$out/pmc_${i}_$which; rm -rf $d
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -o $tag -- $cmd > /dev/null 2> $out/pmc_${i}_${which}_err.txt
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_rows.py "$f" $which >> $out/pmc_rows.jsonl
  rm -rf $d
 done
done
fi
cd $R && python bench.py --no-in-process 2>/dev/null | tail -1 > $out/bench_line.json
rm -rf $out/trace/*/*.db 2>/dev/null
[ -f $out/pmc_rows.jsonl ] && wc -l $out/pmc_rows.jsonl; du -sh $out
