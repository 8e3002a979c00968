for v in stock quad_inline wave_coop stock quad_inline wave_coop; do
  if [ $v = stock ]; then L=""; else L="tools/_variants/$v/libkzg_hip.so"; fi
  echo "== $v"
  KZG_HIP_LIB=$L python tools/walk_probe.py 4096 110 12 2>/dev/null | tail -1
  KZG_HIP_LIB=$L python tools/walk_probe.py 64 110 30 2>/dev/null | tail -1
  KZG_HIP_LIB=$L python tools/lone_commit_trace.py commit 2>/dev/null | tail -1
done
