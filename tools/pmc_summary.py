"""Per-kernel average of one PMC counter from a rocprofv3 --pmc run (counter_collection CSV) -> JSON on stdout."""
import csv
import glob
import json
import sys

d, counter = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rnb::", "")
            a = acc.setdefault(k, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
out = {k: {"avg": v[0] / v[1], "launches": v[1]} for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])}
print(json.dumps({"counter": counter, "kernels": out}, indent=1))
