"""GPU tests (-m gpu) of bench.py's output contract: one JSON line with the metric, the roofline object of the dominant kernel
and (N = 1) the CPU baseline; and the N > 1 code path (domain decomposition, max-over-ranks timing) rehearsed with two
processes on the one GPU through the host transport (RCCL refuses two ranks on one device)."""
import json, os, subprocess, sys, socket
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def last_json_line(text):
    lines = [l for l in text.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def check_common(d, n_gpus, steps, warmup):
    assert d["metric"] == "fine_wilson_clover_gflops" and d["unit"] == "GFLOP/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (n_gpus, steps, warmup)
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    # value is the whole-job rate: flop per site x sites of all ranks / time per step
    sites = 32 ** 4 * n_gpus
    assert abs(d["value"] - d["config"]["flop_per_site"] * sites / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "50", "--warmup", "10", "--no-solve"],
                       capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = last_json_line(r.stdout)
    check_common(d, 1, 50, 10)
    assert d["roofline"]["frac"] > 0.4                       # north-star target: >= 40 % of the HBM roofline at 32^4
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]


def test_two_processes_through_the_host_transport():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--transport", "host", "--no-cpu-baseline", "--no-solve"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = last_json_line(r.stdout)
    check_common(d, 2, 5, 2)
    assert "process grid 2x1x1x1" in d["config"]["parallelism"]
