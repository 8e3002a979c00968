#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py at its DEFAULT step sizes (4096 blobs, 1024 / 512 polynomials per step): the per-kernel
# summary whose k_fb_accumulate average must agree with roofline.avg_launch_ms of the un-profiled line.  usage (GPU box): bash tools/profile_default_bench.sh
R=$(pwd); out=$R/gpurun_out/prof_default; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o def -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/trace_bench.json 2> $out/trace_err.txt
db=$(find $out/trace -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $db $out/kernel_stats.md > /dev/null
cd $R && python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench_line.json
rm -rf $out/trace/*/*.db
python - <<'PY'
import json
d = json.load(open("gpurun_out/prof_default/bench_line.json"))
print("un-profiled line of the same box: %.0f commitments/s, %.3f ms per step, roofline.avg_launch_ms %.3f; FK20 %.0f/s" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["fk20"]["value"]))
PY
grep -E "k_fb_accumulate<0>|k_g1_fft_stage<4, 1>|k_g1_fft_stage_dif<1>" $out/kernel_stats.md
