#!/bin/bash
# same-box A/B of single-call latencies: stock library against tools/_variants/$1/libkzg_hip.so, interleaved 3 times (medians are printed by
# tools/latency_probe.py per run).  usage (GPU box): bash tools/ab_latency.sh <variant>
R=$(pwd)
for i in 1 2 3; do
  echo "== run $i stock"; python tools/latency_probe.py 2>/dev/null | grep -E "CommitToPoly|ComputeProofSingle|LinCombG1\(4096"
  echo "== run $i variant $1"; KZG_HIP_LIB_ALLOW_MISSING=1 KZG_HIP_LIB=$R/tools/_variants/$1/libkzg_hip.so python tools/latency_probe.py 2>/dev/null | grep -E "CommitToPoly|ComputeProofSingle|LinCombG1\(4096"
done
