#!/bin/bash
# timeline of ONE host-buffer DAUsingFK20 batch call (kernels + copies, start offset and duration): where a mid-size call spends its time outside the stage launches
# usage (GPU box): bash tools/fk20_call_timeline.sh 64
R=$(pwd); cd /tmp && export TMPDIR=/tmp
b=${1:-64}
out=$R/gpurun_out/fkcall_b$b; rm -rf $out
KZG_HIP_FB_BUDGET_GB=10 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- python $R/tools/fk20_batch_probe.py $b > $R/gpurun_out/fkcall_b$b.log 2>&1
tail -1 $R/gpurun_out/fkcall_b$b.log
python - "$out" $b <<'PY'
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", r.get("Bytes", ""))))
ev.sort()
# the last call = everything after the last gap > 3 ms preceded by a D2H... simpler: take events after the last occurrence of the first kernel name of a call
names = [e[2] for e in ev]
first = None
for i in range(len(ev) - 1, 0, -1):
    if ev[i][0] - ev[i - 1][1] > 2_000_000 or i == 1:
        first = i; break
call = ev[first:]
t0 = call[0][0]
agg = {}
for s, e, n in call:
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6
print("last call: %d events, span %.2f ms, busy %.2f ms" % (len(call), (call[-1][1] - t0) / 1e6, sum(v[1] for v in agg.values())))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-62s x%-4d %8.3f ms" % (n, c, t))
prev = t0
for s, e, n in call:
    if (s - prev) / 1e6 > 0.15: print("  gap %.3f ms before %s at +%.2f ms" % ((s - prev) / 1e6, n, (s - t0) / 1e6))
    prev = max(prev, e)
PY
rm -rf $out
