"""`python bench.py --gpus N` without a launcher: one rank per GPU under torch.distributed.run, ONE JSON line on stdout."""
import subprocess
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1 with a kernel-chosen port), pass their stderr through, and print exactly ONE line on stdout: rank 0's
    JSON line.  Anything else a rank or the launcher wrote to stdout goes to stderr.  Returns the exit code for sys.exit."""
    import signal
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KZG_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    sys.stderr.write("[bench.py] no WORLD_SIZE in the environment: launching %d ranks: %s\n" % (n_ranks, " ".join(cmd)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate()
    except BaseException:
        os.killpg(proc.pid, signal.SIGKILL)                  # the launcher AND its ranks (own session), nothing else
        proc.wait()
        raise
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    for l in out.splitlines():
        if not l.startswith('{"metric"'):
            sys.stderr.write(l + "\n")
    if proc.returncode != 0 or len(lines) != 1:
        sys.stderr.write("[bench.py] the %d-rank run ended with exit code %d and %d JSON lines\n" % (n_ranks, proc.returncode, len(lines)))
        return proc.returncode or 1
    print(lines[0])
    return 0
