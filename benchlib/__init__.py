"""Parts of bench.py (the driver's entry point at the repo root): workload synthesis and the timing harness (workload), the CPU baseline (cpu),
the self-launcher for N > 1 ranks (launch), the multi-device handle measured in a child process (in_process) and the roofline arithmetic (roofline)."""
