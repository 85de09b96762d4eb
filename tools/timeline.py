"""Profiling aid (GPU box): kernel timeline of one steady-state step from a rocprofv3 --kernel-trace CSV.
   python tools/timeline.py <dir> [step_from_end]"""
import csv
import glob
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rnb::", ""), r.get("Queue_Id", "?")))
rows.sort()
# one cycle = from one k_scan_compact (once per step) to the next
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_loss_pass2") and not r[2].startswith("k_loss_pass2_samples")]  # one per step: k_loss_pass2_rays (two-launch form) or k_loss_pass2<..>
i0, i1 = starts[-back - 1], starts[-back]
t0 = rows[i0][0]
print("step length %.1f us" % ((rows[i1][0] - t0) / 1e3))
for r in rows[i0:i1 + 1]:
    print("%9.1f %9.1f  %7.1f us  q%-3s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2][:50]))
