#!/bin/bash
# tools/coalesce_sweep.sh -- the coalescer's policy knobs against the one-blob API from 64 and 256 native threads (same box, stock library)
R=$(cd "$(dirname "$0")/.." && pwd)
export TABLE_GB=${TABLE_GB:-110} ONLY=1 KZG_HIP_COALESCE_STATS=1
run() { echo "-- $*"; for T in 64 256; do env "$@" python $R/tools/drop_in_probe.py $T 2>&1 | grep -E "native|coalescer" | sed 's/, per batch//'; done; }
run A=0
run KZG_HIP_COALESCE_EXEC=1
run KZG_HIP_COALESCE_EXEC=2
run KZG_HIP_COALESCE_PER_BATCH=32
run KZG_HIP_COALESCE_PER_BATCH=24 KZG_HIP_COALESCE_EXEC=3
run KZG_HIP_COALESCE_PER_BATCH=96
run KZG_HIP_COALESCE_US=50
run KZG_HIP_COALESCE_US=400
run KZG_HIP_COALESCE_SPIN_US=0
run KZG_HIP_COALESCE_SPIN_US=150
run KZG_HIP_COALESCE_PER_BATCH=100 KZG_HIP_COALESCE_US=400
run A=1
