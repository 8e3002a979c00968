R=$(pwd)
run() { echo "== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
run KZG_X=0 python $R/tools/fuzz_msm.py 80 61
run KZG_X=0 python $R/tools/fuzz_fr.py 120 62
run KZG_X=0 python $R/tools/fuzz_g1.py 40 63
run KZG_X=0 python $R/tools/fuzz_multi.py 40 64
run KZG_HIP_MULTI_TRANSPORT=host KZG_HIP_MULTI_FFT=sharded python $R/tools/fuzz_multi.py 24 65
