"""GPU tests (-m gpu) of bench.py's output contract: one JSON line with the metric, the roofline object of the dominant kernel
and (N = 1) the CPU baseline; the N > 1 paths -- `bench.py --gpus 2` starting its own ranks, and the same under torchrun as the
driver launches it -- rehearsed with two processes on the one GPU through the host transport (RCCL refuses two ranks on one
device); and the failure behaviour: a rank that dies or hangs in the solve leg ends the job with a non-zero code, with the
headline line still printed."""
import json, os, subprocess, sys
import pytest
from launcher import torchrun

pytestmark = pytest.mark.gpu
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
BENCH = os.path.join(REPO, "bench.py")


def last_json_line(text):
    lines = [l for l in text.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def check_common(d, n_gpus, steps, warmup, sites_per_gpu=32 ** 4):
    assert d["metric"] == "fine_wilson_clover_gflops" and d["unit"] == "GFLOP/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (n_gpus, steps, warmup)
    assert d["config"]["ranks"] == n_gpus
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    # value is the whole-job rate: flop per site x sites of all ranks / time per step
    sites = sites_per_gpu * n_gpus
    assert abs(d["value"] - d["config"]["flop_per_site"] * sites / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0


def test_single_gpu_line():
    r = subprocess.run([sys.executable, BENCH, "--steps", "50", "--warmup", "10", "--no-strong"],
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = last_json_line(r.stdout)
    check_common(d, 1, 50, 10)
    assert d["roofline"]["frac"] > 0.4                       # north-star target: >= 40 % of the HBM roofline at 32^4
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    s = d["solve"]
    assert s["true_relres"] < 1e-10 and s["seconds_per_solve"] > 0
    # the same setup once more in the same context (a second rand() stream: the count may move by one)
    assert s["setup_seconds_repeated"] > 0 and abs(s["repeated_setup_solve"]["iterations"] - s["iterations"]) <= 1
    assert s["repeated_setup_solve"]["true_relres"] < 1e-10
    assert len(r.stdout.strip().splitlines()) == 1           # stdout is the one JSON line (RCCL's banner and the like go to stderr)
    if "iterations_reference" in s:                          # the reference's run of the same 32^4 case (committed fixture)
        assert abs(s["iterations"] - s["iterations_reference"]) <= 1
    if c["kind"] == "reference":
        assert c["solve"]["seconds"] > 0 and c["solve"]["iterations"] > 0 and c["solve"]["kind"] == "reference"


SMALL = ["--lattice", "16", "16", "16", "16", "--strong-lattice", "16", "16", "16", "16", "--steps", "5", "--warmup", "2",
         "--transport", "host", "--no-cpu-baseline"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher on the command line: two ranks, the strong-scaling solve on one global lattice"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, capture_output=True, text=True, timeout=900, cwd=REPO,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = last_json_line(r.stdout)
    check_common(d, 2, 5, 2, 16 ** 4)
    assert "process grid 2x1x1x1" in d["config"]["parallelism"]
    s = d["strong_scaling"]
    assert s["n_gpus"] == 2 and s["scaling"] == "strong" and s["true_relres"] < 1e-10 and 8 <= s["iterations"] <= 20
    assert "16x16x16x16" in s["workload"] and "local 8x16x16x16" in s["workload"]
    # the north star's strong-scaling metric at the top level of the N > 1 line, and what travelled during the timed solve
    ns = d["north_star_strong_scaling"]
    assert ns["scaling"] == "strong" and ns["n_gpus"] == 2 and ns["seconds_per_solve"] == s["seconds_per_solve"] and ns["iterations"] == s["iterations"]
    m = s["messages"]
    kinds = {e["bytes_per_face_site"]: e for e in m["halo_exchanges"]}
    assert 48 in kinds and 96 in kinds and kinds[48]["what"].startswith("level 0 fp32") and kinds[96]["exchanges"] >= s["iterations"]
    assert kinds[48]["messages"] == 2 * kinds[48]["exchanges"]             # one split direction: two messages per exchange
    assert kinds[48]["bytes_per_message"] == 48 * 16 ** 3                  # the T face of the local 8 x 16^3 lattice, 6 complex fp32 per site
    assert m["allreduce"]["calls"] >= 2 * s["iterations"] and m["allgather"]["calls"] > 0


def test_two_processes_under_torchrun():
    """the driver's form: torchrun starts the ranks, bench.py checks --gpus against the world size"""
    r = torchrun(2, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--transport", "host", "--no-cpu-baseline", "--no-strong",
                 timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = last_json_line(r.stdout)
    check_common(d, 2, 5, 2)
    r = torchrun(2, BENCH, "--gpus", "4", "--no-strong", "--transport", "host", timeout=300, cwd=REPO)
    assert r.returncode != 0 and "--gpus 4 but the launcher started 2" in (r.stdout + r.stderr)


@pytest.mark.parametrize("fault,extra", [("DDAMG_BENCH_FAIL_RANK", []), ("DDAMG_BENCH_HANG_RANK", ["--leg-timeout", "20"])])
def test_failed_or_hung_rank_gives_nonzero_exit_and_keeps_the_headline(fault, extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env[fault] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL + extra, capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert r.returncode != 0, r.stdout[-2000:]
    d = last_json_line(r.stdout)
    check_common(d, 2, 5, 2, 16 ** 4)
    assert "error" in d["strong_scaling"]


def test_rehearsal_of_the_eight_gpu_point_on_one_gpu():
    """`--rehearse 8` (the default at N = 1): the per-GPU problem of the 8-GPU decomposition -- here of a 32^4 lattice, local
    16x16x16x32 -- with three split directions through the RCCL self-exchange and the coarsest level gathered; the `rehearsal`
    object carries the predicted seconds per solve per GPU next to the N = 1 time of the same run"""
    r = subprocess.run([sys.executable, BENCH, "--steps", "5", "--warmup", "2", "--no-solve", "--no-cpu-baseline", "--rehearse", "8",
                        "--strong-lattice", "32", "32", "32", "32"], capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = last_json_line(r.stdout)
    s, h = d["strong_scaling"], d["rehearsal"]
    assert "error" not in h, h
    assert h["n_gpus_rehearsed"] == 8 and h["process_grid"] == [2, 2, 2, 1] and h["local_lattice"] == [16, 16, 16, 32]
    assert h["self_exchange"] == [-1, -1, -1, 1] and h["true_relres"] < 1e-10
    assert abs(h["iterations"] - h["same_lattice_without_the_machinery"]["iterations"]) <= 1      # the machinery changes no result
    assert h["predicted_seconds_per_solve_per_gpu"] >= h["seconds_per_solve_per_gpu"] > 0
    assert abs(h["predicted_speedup_vs_n1"] - s["seconds_per_solve"] / h["predicted_seconds_per_solve_per_gpu"]) < 1e-9
    assert h["coarsest_level"]["gathered_sites_on_n_gpus"] == 8 * h["coarsest_level"]["rehearsed_sites"]
    # what this "rank" sent during the timed solve, payload by payload (to be read against the message table of docs/design/06a_rehearsal_and_messages.md)
    m = h["messages"]
    kinds = {e["bytes_per_face_site"]: e for e in m["halo_exchanges"]}
    assert m["transport"] == "rccl" and kinds[48]["messages"] == 6 * kinds[48]["exchanges"]        # three split directions
    assert kinds[48]["exchanges_per_outer_iteration"] >= 4 and kinds[96]["exchanges"] >= h["iterations"]   # 4 colour sweeps per smoother call
    assert 8 * 48 in kinds and m["allgather"]["calls"] > 0 and m["allreduce"]["milliseconds_on_the_transport_stream"] > 0
