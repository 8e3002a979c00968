#!/bin/bash
# builds the stand-alone A/B harness of the 256-lane F_r transform against the in-tree library (run on a GPU box: tools/ab_fr_r16/r16_ab [batch] [reps])
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value ${KZG_AB_SAVE_TEMPS:+-save-temps=obj} r16_ab.hip -o r16_ab -L../../go-kzg_amd -lkzg_hip -Wl,-rpath,'$ORIGIN/../../go-kzg_amd'
echo built tools/ab_fr_r16/r16_ab
