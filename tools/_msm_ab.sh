R=$PWD
python tools/msm_probe.py 8 16 32 64 256 512 2>&1 | tail -1
KZG_HIP_MSM_SEG=1 python tools/msm_probe.py 8 16 32 64 2>&1 | tail -1
KZG_HIP_LIB=$R/tools/_variants/seg32/libkzg_hip.so KZG_HIP_MSM_SEG=1 python tools/msm_probe.py 16 64 512 2>&1 | tail -1
KZG_HIP_LIB=$R/tools/_variants/seg128/libkzg_hip.so KZG_HIP_MSM_SEG=1 python tools/msm_probe.py 16 64 512 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/msm_trace -o msm -- python $R/tools/msm_probe.py 512 > /dev/null 2>&1
db=$(find $R/gpurun_out/msm_trace -name "*.db" | head -1); python $R/tools/rocprof_summary.py $db | grep -E "k_msm|kernel" | head -12
rm -rf $R/gpurun_out/msm_trace
