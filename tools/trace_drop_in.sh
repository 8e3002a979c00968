#!/bin/bash
# GPU timeline of the one-blob-per-call API under T native host threads: busy fraction of the device, kernels per batch, gaps.
# usage (GPU box): bash tools/trace_drop_in.sh [threads]
T=${1:-64}
R=$(pwd); out=$R/gpurun_out/trace_dropin; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ONLY=1 rocprofv3 --kernel-trace -d $out -o t -- python $R/tools/drop_in_probe.py $T > $out/probe.txt 2>&1
cat $out/probe.txt | grep native
db=$(find $out -name "*.db" | head -1)
python - $db <<'PY'
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1]).cursor()
rows = c.execute("select s.kernel_name, d.start, d.end, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
import re
def short(n):
    m = re.match(r"_ZN3kzg(\d+)", n)
    return n[m.end():m.end() + int(m.group(1))] + ("<split>" if "ILb1E" in n else "") if m else n.split("(")[0].replace("kzg::", "")
rows = [(short(n), s, e, g) for n, s, e, g in rows]
# the timed region: the last 60 % of the walk launches
walks = [r for r in rows if r[0].startswith("k_fb_accumulate")]
lo = walks[int(len(walks) * 0.4)][1]; hi = walks[-1][2]
sel = [r for r in rows if r[1] >= lo and r[2] <= hi]
ev = sorted([(s, 1) for _, s, e, _ in sel] + [(e, -1) for _, s, e, _ in sel])
busy = 0; depth = 0; last = None; over = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last; over[depth] += t - last
    depth += d; last = t
print("window %.1f ms, device busy %.1f %% (time with >= 1 kernel running); concurrency histogram (kernels in flight: share of window): %s" % ((hi - lo) / 1e6, 100.0 * busy / (hi - lo), {k: round(v / (hi - lo), 3) for k, v in sorted(over.items())}))
by = collections.defaultdict(list)
for n, s, e, g in sel: by[n].append((e - s) / 1e3)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s launches %5d  avg %8.1f us  total %8.1f ms (%.1f %% of window)" % (n[:28], len(v), sum(v) / len(v), sum(v) / 1e3, 100 * sum(v) / 1e3 / ((hi - lo) / 1e6)))
grids = collections.Counter(g for n, s, e, g in sel if n.startswith("k_fb_accumulate"))
print("walk grid sizes (lanes: launches):", dict(sorted(grids.items())))
PY
rm -rf $out/*/*.db
