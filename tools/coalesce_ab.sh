#!/bin/bash
# tools/coalesce_ab.sh -- same-box A/B of the request coalescer: the one-blob API from 1 .. 256 native threads on the stock library and on a
# variant (KZG_HIP_LIB), alternating, two rounds each.  `bash tools/coalesce_ab.sh tools/_variants/old_coalesce/libkzg_hip.so`
R=$(cd "$(dirname "$0")/.." && pwd)
VAR=$1
export TABLE_GB=${TABLE_GB:-110}
for round in 1 2; do
  echo "== round $round: stock"; python $R/tools/drop_in_probe.py 64 2>&1 | grep -v amdgpu.ids
  echo "== round $round: variant $VAR"; KZG_HIP_LIB=$VAR KZG_HIP_LIB_ALLOW_MISSING=1 python $R/tools/drop_in_probe.py 64 2>&1 | grep -v amdgpu.ids
done
