"""one way to start N ranks for the multi-process tests: torchrun picks its own rendezvous port (--standalone), so there is
no probe-then-bind race and nothing to retry"""
import os, subprocess, sys


def torchrun(nproc, script, *args, timeout=600, env=None, cwd=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={nproc}", script, *args]
    e = dict(os.environ, OMP_NUM_THREADS="1")
    if env:
        e.update(env)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=cwd)
