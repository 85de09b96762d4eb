#!/bin/bash
# GPU box: the counter calibration of tools/probe_counters.hip -> gpurun_out/r05_counter_calibration.json (copied to profiles/ by hand).
# One counter per rocprofv3 pass, as the microarchitecture guide prescribes; every probe kernel is launched twice (warm + timed): the LAST launch is read.
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
hipcc --offload-arch=gfx950 -O3 -o tools/probe_counters tools/probe_counters.hip || exit 1
mkdir -p gpurun_out
tools/probe_counters > /tmp/probe_counters_plain.json || exit 1
for C in FETCH_SIZE WRITE_SIZE TCC_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum; do
  rm -rf /tmp/cal_$C
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/cal_$C -- $ROOT/tools/probe_counters > /tmp/cal_$C.log 2>&1) || echo "pass $C failed"
done
python - <<'PY'
import csv, glob, json
plain = json.load(open("/tmp/probe_counters_plain.json"))["kernels"]
out = {"_source": "tools/probe_counters.hip under rocprofv3 --pmc <counter> --kernel-trace, one counter per pass (tools/calibrate_counters.sh); the last launch of each kernel",
       "_units": "FETCH_SIZE / WRITE_SIZE: KiB as reported x 1024 = bytes; the TCC_* counters: requests", "kernels": {}}
for name, t in plain.items():
    out["kernels"][name] = dict(t, gb_per_s_true=round((t["true_read_bytes"] + t["true_write_bytes"]) / (t["ms"] * 1e-3) / 1e9, 1), ops_per_s=round(t["ops"] / (t["ms"] * 1e-3)))
for counter in ("FETCH_SIZE", "WRITE_SIZE", "TCC_ATOMIC_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
    rows = []
    for f in glob.glob("/tmp/cal_%s/**/*counter_collection.csv" % counter, recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter:
                    rows.append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    for name in plain:
        vals = [v for _, k, v in rows if name in k]
        if vals:
            out["kernels"][name][counter] = vals[-1] * (1024 if counter in ("FETCH_SIZE", "WRITE_SIZE") else 1)
for name, k in out["kernels"].items():
    if k.get("FETCH_SIZE") and k["true_read_bytes"]:
        k["FETCH_SIZE_over_true_read"] = round(k["FETCH_SIZE"] / k["true_read_bytes"], 4)
    if k.get("WRITE_SIZE") and k["true_write_bytes"]:
        k["WRITE_SIZE_over_true_write"] = round(k["WRITE_SIZE"] / k["true_write_bytes"], 4)
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_ATOMIC_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"):
        if k.get(c) is not None and k["ops"]:
            k[c + "_per_op"] = round(k[c] / k["ops"], 3)
json.dump(out, open("gpurun_out/r05_counter_calibration.json", "w"), indent=1)
for name, k in out["kernels"].items():
    print(name, {q: k[q] for q in k if q.endswith("_over_true_read") or q.endswith("_over_true_write") or q.endswith("_per_op") or q in ("ms", "gb_per_s_true", "ops_per_s")})
PY
