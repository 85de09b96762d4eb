"""GPU box: per-kernel average duration over the LAST n_steps training steps of a rocprofv3 --kernel-trace run
(the --stats table also averages the burn-in steps, whose ray counts and active levels differ).
   python tools/steady_stats.py <dir> <n_steps>"""
import csv
import glob
import json
import sys

d, n_steps = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rnb::", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("k_loss_pass2") and not r[2].startswith("k_loss_pass2_samples")]  # one per step: k_loss_pass2_rays (two-launch form) or k_loss_pass2<..>
lo = marks[-n_steps - 1]
hi = marks[-1]
acc = {}
for s, e, k in rows[lo:hi]:
    a = acc.setdefault(k, [0, 0])
    a[0] += e - s
    a[1] += 1
out = {k: {"avg_us": round(v[0] / v[1] / 1e3, 2), "calls_per_step": round(v[1] / n_steps, 2), "us_per_step": round(v[0] / n_steps / 1e3, 2)} for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])}
print(json.dumps({"steps": n_steps, "wall_us_per_step": round((rows[hi][0] - rows[lo][0]) / n_steps / 1e3, 2), "kernels": out}, indent=1))
