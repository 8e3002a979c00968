#!/bin/bash
# same-box A/B of the final build against round 5's library (tools/_variants/r05/libkzg_hip.so = commit 604078d built with the same flags): the 4096-blob walk, a lone
# CommitToPoly / ComputeProofSingle / eth.ComputeKZGProof, alternating, three rounds.  usage (GPU box): bash tools/ab_round5.sh
R=$(cd "$(dirname "$0")/.." && pwd)
for i in 1 2 3; do
  for v in r06 r05; do
    if [ $v = r06 ]; then export -n KZG_HIP_LIB; unset KZG_HIP_LIB KZG_HIP_LIB_ALLOW_MISSING; else export KZG_HIP_LIB=$R/tools/_variants/r05/libkzg_hip.so KZG_HIP_LIB_ALLOW_MISSING=1; fi
    echo "== run $i $v"
    python $R/tools/walk_probe.py 4096 110 12 2>/dev/null | tail -1 | sed 's/.*| batch/batch/'
    python $R/tools/lone_commit_trace.py commit 2>/dev/null | tail -1
    python $R/tools/lone_commit_trace.py proof 2>/dev/null | tail -1
    python $R/tools/lone_eth_proof_trace.py 2>/dev/null | tail -1
  done
done
