#!/bin/bash
# round-5 fuzz pass (GPU box): the multi-device handle on every transport, the bucket MSM with each reduce form forced, the F_r entry points with the
# 256-lane form of the 4096-point kernel forced.  Prints one summary line per run.
R=$(cd "$(dirname "$0")/.." && pwd)
run() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -2; }
run KZG_X=0 python $R/tools/fuzz_multi.py 24 51
run KZG_HIP_MULTI_TRANSPORT=host python $R/tools/fuzz_multi.py 24 52
run KZG_HIP_MULTI_FAULT=peer python $R/tools/fuzz_multi.py 16 53
run KZG_HIP_MSM_REDUCE=chunks python $R/tools/fuzz_msm.py 30 54
run KZG_HIP_MSM_REDUCE=scan KZG_HIP_MSM_SEG=1 python $R/tools/fuzz_msm.py 30 55
run KZG_HIP_FR_FFT=r16 python $R/tools/fuzz_fr.py 60 56
run KZG_HIP_FB_GLV=0 python $R/tools/fuzz_msm.py 20 57
