"""Execution-unit counters per kernel (tools/collect_pmc.sh passes) + steady-state durations (tools/steady_stats.py, serial
mode) -> profiles/r01_pmc_units.json: L2 atomic request rate against the measured device limit, MFMA busy fraction.
   python tools/pmc_report.py [pmc_dir] [steady_serial.json] [out.json]"""
import json
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
steady = sys.argv[2] if len(sys.argv) > 2 else "profiles/r02_steady_serial_step2000.json"
out_path = sys.argv[3] if len(sys.argv) > 3 else "profiles/r02_pmc_units.json"
ATOMIC_PROBE = 21.0e9  # L2 atomic requests/s, device-wide, the best any pattern of tools/probe_atomics*.hip reached (one 64-B line of one instruction = one request)
CLOCK_HZ = 2.4e9       # MI355X peak engine clock
N_SIMD = 256 * 4


def load(name):
    try:
        return json.load(open("%s/%s.json" % (d, name)))["kernels"]
    except OSError:
        return {}


atom, busy, mops = load("TCC_ATOMIC_sum"), load("SQ_VALU_MFMA_BUSY_CYCLES"), load("SQ_INSTS_VALU_MFMA_MOPS_F16")
dur = json.load(open(steady))["kernels"]
rep = {"_source": "rocprofv3 --pmc <one counter per pass>, kernels serialised (cfg.overlap=0), averages over the last 100 launches; "
                  "durations: %s (rocprofv3 --kernel-trace, last 200 steps)" % steady,
       "_atomic_probe_requests_per_s": ATOMIC_PROBE, "_mfma_busy_denominator": "duration x %.1f GHz x %d SIMDs" % (CLOCK_HZ / 1e9, N_SIMD), "kernels": {}}
for k, v in dur.items():
    us = v["avg_us"]
    e = {"avg_us": us}
    if atom.get(k, {}).get("avg", 0) > 1000:
        r = atom[k]["avg"]
        e["l2_atomic_requests"] = round(r)
        e["l2_atomic_requests_per_s"] = round(r / (us * 1e-6))
        e["frac_of_probe_rate"] = round(r / (us * 1e-6) / ATOMIC_PROBE, 3)
    if busy.get(k, {}).get("avg", 0) > 0:
        e["mfma_busy_cycles"] = round(busy[k]["avg"])
        e["mfma_mops_f16"] = round(mops.get(k, {}).get("avg", 0))
        e["mfma_busy_frac"] = round(busy[k]["avg"] / (us * 1e-6 * CLOCK_HZ * N_SIMD), 4)
    if len(e) > 1:
        rep["kernels"][k] = e
json.dump(rep, open(out_path, "w"), indent=1)
for k, e in rep["kernels"].items():
    print(k[:48], e)
