#!/bin/bash
# kernel timeline of the LAST call of a single-call operation: tools/trace_ops.sh <op> [rows]   (GPU box)
op=$1; rows=${2:-40}
R=$(pwd); out=$R/gpurun_out/trace_$op; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out -o t -- python $R/tools/trace_ops.py $op > /dev/null 2>&1
db=$(find $out -name "*.db" | head -1)
python - $db $rows <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
rows = c.execute("select s.kernel_name, d.start, d.end, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
rows = rows[-int(sys.argv[2]):]
t0 = rows[0][1]
for name, st, en, grid in rows:
    print("%-28s start %9.1f us  dur %8.1f us  grid %8d" % (name.split("(")[0].replace("kzg::", "")[:28], (st - t0) / 1e3, (en - st) / 1e3, grid))
PY
rm -rf $out
