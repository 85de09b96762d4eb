"""Per-kernel average of one PMC counter from a rocprofv3 --pmc run (counter_collection CSV) -> JSON on stdout.
   python tools/pmc_summary.py <dir> <COUNTER> [last_n]     averages over the last_n launches of each kernel (default 100:
   the steady state after the burn-in, where all levels are live and the ray count has settled)."""
import csv
import glob
import json
import sys

d, counter = sys.argv[1], sys.argv[2]
last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
acc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rnb::", "")
            acc.setdefault(k, []).append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
out = {}
for k, v in acc.items():
    v.sort()
    tail = [x[1] for x in v[-last_n:]]
    out[k] = {"avg": sum(tail) / len(tail), "launches": len(tail), "launches_total": len(v)}
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["avg"] * kv[1]["launches"]))
print(json.dumps({"counter": counter, "averaged_over": "last %d launches of each kernel" % last_n, "kernels": out}, indent=1))
