#!/usr/bin/env python3
"""tools/roofline_table.py -- the solve-path kernels of the 32^4 two-level configuration against the HBM roofline, from the
committed kernel statistics (profiles/r02_bench_kernel_stats.csv, profiles/r02_solve32_kernel_stats.csv): average
duration, algorithmic bytes per launch (SURVEY.md 8d figures x units per launch), achieved rate, fraction of 8 TB/s.
Writes profiles/r02_roofline_table.md."""
import csv, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = 32 ** 4
Vc, n = 8 ** 4, 48
def rows(f):
    return list(csv.DictReader(open(os.path.join(REPO, "profiles", f))))
def avg(rs, key):
    r = [x for x in rs if key in x["Name"]]
    return (float(r[0]["AverageUs"]), int(r[0]["Calls"])) if r else (None, 0)
b, s = rows(R + "_bench_kernel_stats.csv"), rows(R + "_solve32_kernel_stats.csv")
table = [
    ("dirac_apply_lds_kernel<float>", "fine operator, fp32 (headline)", b, "dirac_apply_lds_kernel<float", 816 * V, "816 B/site"),
    ("dirac_apply_lds_kernel<double>", "fine operator, fp64 (outer residual)", s, "dirac_apply_lds_kernel<double", 1632 * V, "1632 B/site"),
    ("sap_pair_kernel", "Schwarz colour launch incl. 4 MinRes steps (mean of both launch types)", s, "sap_pair_kernel", 1060 * (V // 2), "1060 B/site of one colour"),
    ("restrict_kernel", "restriction, Nvec 24", s, "restrict_kernel<float", 2400 * V, "2400 B/site"),
    ("interpolate_kernel", "interpolation (+=), Nvec 24", s, "interpolate_kernel<float", 2496 * V, "2400 + 96 B/site"),
    ("coarse_site_kernel<.., HOP>", "coarsest level, half hopping term, 8^4 x 48", s, "coarse_site_kernel<float, 6, 1>", (Vc // 2) * 8 * n * n * 8, "8 n^2 complex per output site"),
    ("coarse_site_kernel<.., SELF>", "coarsest level, self-coupling product on one parity", s, "coarse_site_kernel<float, 6, 2>", (Vc // 2) * n * n * 8, "n^2 complex per site"),
]
out = ["| kernel | what | launches | avg us | algorithmic bytes / launch | TB/s | of 8 TB/s |", "|---|---|---|---|---|---|---|"]
for name, what, rs, key, byts, unit in table:
    us, calls = avg(rs, key)
    if us is None:
        continue
    out.append(f"| `{name}` | {what} | {calls} | {us:.1f} | {byts / 1e6:.1f} MB ({unit}) | {byts / us / 1e6:.2f} | {byts / us / 1e6 / 8.0:.2f} |")
txt = ("Solve-path kernels at 32^4 (two levels, Nvec 24) against the HBM roofline; durations from the committed rocprofv3 kernel\n"
       "statistics (`" + R + "_bench_kernel_stats.csv`, `" + R + "_solve32_kernel_stats.csv`; the solve run includes the setup's launches of the\n"
       "same kernels), bytes = SURVEY.md section 8d per-unit figures x units per launch.  Written by tools/roofline_table.py.\n\n" + "\n".join(out) + "\n")
open(os.path.join(REPO, "profiles", R + "_roofline_table.md"), "w").write(txt)
print(txt)
