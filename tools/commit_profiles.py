#!/usr/bin/env python3
"""tools/commit_profiles.py [SRC_DIR [ROUND]] -- copy the artefacts of `tools/gpu/run.sh profile <round>` (default gpurun_out/profile,
r04) into profiles/ and derive the provenance files bench.py reads:
  profiles/<round>_traffic.json            fabric bytes per launch of the fine operator (2*FETCH_SIZE + WRITE_SIZE), with the
                                           commit and the hash of the kernel sources it was measured on
  profiles/<round>_strong_scaling_n1.json  the one-GPU point of the 64^4 strong-scaling solve
  profiles/<round>_mfma_busy.json          matrix-core busy fraction of the many-right-hand-side coarse operator kernels, with
                                           the hash of their sources
Run it on the commit the profile was taken from (the working tree's kernel sources are hashed)."""
import json, os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench

R = sys.argv[2] if len(sys.argv) > 2 else "r04"
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "profile")
dst = os.path.join(REPO, "profiles")
for f in sorted(os.listdir(src)):
    if f.startswith(R + "_"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, R + "_bench_line.json"))
commit = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()

pmc = json.load(open(os.path.join(src, R + "_pmc_bench.json")))
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
        "source": "profiles/" + R + "_pmc_bench.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, %d launches each)" % pmc[key]["FETCH_SIZE"]["launches"],
        "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE tallies the 128 B requests of 16 B/lane streams at 64 B (MI355X_MICROARCH.md, HBM section; calibrated in round 1 with a known-byte copy, r01_pmc_dirac_gather.json). Infinity-Cache hits are counted, so this is fabric traffic and an upper bound on HBM traffic. 672 B/site is what the kernel's own layout has to move (24 in + 24 out + 48 link + 72 clover reals); the 816 B/site of the roofline figure is the reference's storage (SURVEY.md 8d).",
    },
    "commit": commit, "kernel_source_sha16": bench.kernel_source_hash(),
}
json.dump(traffic, open(os.path.join(dst, R + "_traffic.json"), "w"), indent=1)

line = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
ss = line["strong_scaling"]
n1 = {"global_lattice": [64, 64, 64, 64], "seconds_per_solve": ss["seconds_per_solve"], "setup_seconds": ss["setup_seconds"],
      "iterations": ss["iterations"], "coarse_iterations": ss["coarse_iterations"], "true_relres": ss["true_relres"],
      "source": "profiles/" + R + "_bench_line.json: python3 bench.py (default) on one MI355X under rocprofv3 --kernel-trace, commit " + commit}
json.dump(n1, open(os.path.join(dst, R + "_strong_scaling_n1.json"), "w"), indent=1)

# matrix-core busy fraction of the kernels that apply a coarse operator to many right-hand sides
def busy(fname, key):
    try:
        d = json.load(open(os.path.join(src, fname)))
        k = [x for x in d if key in x][0]
        b = d[k]["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"]; cyc = d[k]["GRBM_GUI_ACTIVE"]["mean"] / 8.0    # counted per XCD
        return {"kernel": k, "mfma_busy": b / (cyc * 1024.0), "launches": d[k]["SQ_VALU_MFMA_BUSY_CYCLES"]["launches"], "source": "profiles/" + fname}
    except Exception as e:
        return {"error": str(e)}
mf = {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
      "level1_block_minres": busy(R + "_pmc_mfma64.json", "cm_block_minres_op_kernel"), "level1_apply": busy(R + "_pmc_mfma64.json", "cm_apply_op_kernel"),
      "level1_restrict": busy(R + "_pmc_mfma64.json", "cm_restrict_kernel"), "level1_interpolate": busy(R + "_pmc_mfma64.json", "cm_interpolate_kernel"),
      "lockstep_hop": busy(R + "_pmc_mfma_lockstep32.json", "ls_hop_op_kernel"), "lockstep_self": busy(R + "_pmc_mfma_lockstep32.json", "ls_self_op_kernel"),
      "galerkin_coarse_apply": busy(R + "_pmc_mfma.json", "coarse_batch_apply_kernel"), "galerkin_restrict": busy(R + "_pmc_mfma.json", "restrict_mfma_kernel<2"),
      "galerkin_coarse_restrict": busy(R + "_pmc_mfma.json", "coarse_batch_restrict_store_mfma_kernel"),
      "commit": commit, "kernel_source_sha16": bench.mfma_source_hash()}
json.dump(mf, open(os.path.join(dst, R + "_mfma_busy.json"), "w"), indent=1)
print(json.dumps(traffic["dirac_apply_lds_kernel<float>"], indent=1)[:600]); print(n1); print(json.dumps(mf, indent=1))
