#!/usr/bin/env python3
"""tools/commit_profiles.py [SRC_DIR] -- copy the artefacts of tools/gpu/profile_r02.sh (default gpurun_out/prof_r02) into
profiles/ and derive the two provenance files bench.py reads:
  profiles/r02_traffic.json            fabric bytes per launch of the fine operator (2*FETCH_SIZE + WRITE_SIZE), with the
                                       commit and the hash of the kernel sources it was measured on
  profiles/r02_strong_scaling_n1.json  the one-GPU point of the 64^4 strong-scaling solve
Run it on the commit the profile was taken from (the working tree's kernel sources are hashed)."""
import json, os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "prof_r02")
dst = os.path.join(REPO, "profiles")
for f in sorted(os.listdir(src)):
    if f.startswith("r02_"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, "r02_bench_line.json"))
commit = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()

pmc = json.load(open(os.path.join(src, "r02_pmc_bench.json")))
key = [k for k in pmc if "dirac_apply_lds_kernel<float" in k][0]
fetch, write = pmc[key]["FETCH_SIZE"]["mean"], pmc[key]["WRITE_SIZE"]["mean"]
V = 32 ** 4
byts = int((2 * fetch + write) * 1024)
traffic = {
    "dirac_apply_lds_kernel<float>": {
        "kernel": key, "fetch_size_kib": fetch, "write_size_kib": write, "bytes_per_launch": byts,
        "algorithmic_bytes_per_launch": 816 * V, "ratio_to_algorithmic": byts / (816 * V),
        "bytes_the_layout_moves_per_launch": 672 * V,
        "workload": "32^4 fp32, bench.py --steps 25 (two-row links, arithmetic neighbours, non-temporal clover loads and result stores)",
        "source": "profiles/r02_pmc_bench.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, %d launches each)" % pmc[key]["FETCH_SIZE"]["launches"],
        "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE tallies the 128 B requests of 16 B/lane streams at 64 B (MI355X_MICROARCH.md, HBM section; calibrated in round 1 with a known-byte copy, r01_pmc_dirac_gather.json). Infinity-Cache hits are counted, so this is fabric traffic and an upper bound on HBM traffic. 672 B/site is what the kernel's own layout has to move (24 in + 24 out + 48 link + 72 clover reals); the 816 B/site of the roofline figure is the reference's storage (SURVEY.md 8d).",
    },
    "commit": commit, "kernel_source_sha16": bench.kernel_source_hash(),
}
json.dump(traffic, open(os.path.join(dst, "r02_traffic.json"), "w"), indent=1)

line = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
ss = line["strong_scaling"]
n1 = {"global_lattice": [64, 64, 64, 64], "seconds_per_solve": ss["seconds_per_solve"], "setup_seconds": ss["setup_seconds"],
      "iterations": ss["iterations"], "coarse_iterations": ss["coarse_iterations"], "true_relres": ss["true_relres"],
      "source": "profiles/r02_bench_line.json: python3 bench.py (default) on one MI355X under rocprofv3 --kernel-trace, commit " + commit}
json.dump(n1, open(os.path.join(dst, "r02_strong_scaling_n1.json"), "w"), indent=1)
print(json.dumps(traffic["dirac_apply_lds_kernel<float>"], indent=1)[:600]); print(n1)
