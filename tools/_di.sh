for ex in 3 2 1; do for T in 64 256; do echo "== EXEC=$ex T=$T"; KZG_HIP_COALESCE_EXEC=$ex KZG_HIP_COALESCE_STATS=1 ONLY=1 python tools/drop_in_probe.py $T 2>&1 | grep -E "native|coalescer"; done; done
nproc; cat /sys/fs/cgroup/cpu.max
