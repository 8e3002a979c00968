#!/bin/bash
# round-6 fuzz pass (GPU box) on the final build: the F_p inversion (cooperative == lane == Python), the promotion of repeated point sets in kzg_hip_lincomb_g1, the
# commitment / MSM paths with the cooperative inversion on and off, the multi-device handle incl. an injected hang, F_r and G1 entry points.  One summary line per run.
R=$(cd "$(dirname "$0")/.." && pwd)
run() { echo "== $*"; env "$@" 2>&1 | grep -E "mismatch|Error|Panic|Traceback" | tail -1; }
run KZG_X=0 python $R/tools/fuzz_inv.py 8 71
run KZG_X=0 python $R/tools/fuzz_promote.py 160 72
run KZG_HIP_LINCOMB_PROMOTE_AFTER=1 python $R/tools/fuzz_promote.py 120 73
run KZG_X=0 python $R/tools/fuzz_msm.py 80 74
run KZG_HIP_COOP_INV=0 python $R/tools/fuzz_msm.py 30 75
run KZG_HIP_MSM_REDUCE=chunks python $R/tools/fuzz_msm.py 30 76
run KZG_HIP_FB_GLV=0 python $R/tools/fuzz_msm.py 20 77
run KZG_X=0 python $R/tools/fuzz_fr.py 120 78
run KZG_X=0 python $R/tools/fuzz_g1.py 40 79
run KZG_X=0 python $R/tools/fuzz_multi.py 40 80
run KZG_HIP_MULTI_FAULT=peer-hang KZG_HIP_MULTI_PROBE_TIMEOUT_MS=300 python $R/tools/fuzz_multi.py 16 81
run KZG_HIP_MULTI_FAULT=rccl-hang KZG_HIP_MULTI_TRANSPORT=rccl KZG_HIP_MULTI_PROBE_TIMEOUT_MS=300 python $R/tools/fuzz_multi.py 8 84
run KZG_HIP_MULTI_TRANSPORT=host KZG_HIP_MULTI_FFT=sharded python $R/tools/fuzz_multi.py 24 82
run KZG_HIP_MULTI_FAULT=rccl-block KZG_HIP_MULTI_TRANSPORT=rccl KZG_HIP_MULTI_PROBE_TIMEOUT_MS=300 python $R/tools/fuzz_multi.py 8 85
