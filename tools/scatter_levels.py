"""Profiling aid (GPU box): per-level duration of the grid-gradient scatter. Run after
   RNB_SCATTER_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline
   python tools/scatter_levels.py <dir> [n_levels]"""
import csv
import glob
import sys

d = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "k_grid_scatter" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
per_step = [rows[i:i + L] for i in range(0, len(rows) - L + 1, L)][-20:]
print("steps used:", len(per_step))
tot = 0.0
for l in range(L):
    us = [(s[l][1] - s[l][0]) / 1e3 for s in per_step]
    tot += sum(us) / len(us)
    print("level %2d  %-40s %8.1f us" % (l, per_step[-1][l][2][:40], sum(us) / len(us)))
print("sum %.1f us" % tot)
