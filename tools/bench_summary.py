"""One line per bench.py record: python tools/bench_summary.py file.json ..."""
import json
import sys

for path in sys.argv[1:]:
    lines = [l for l in open(path) if l.startswith("{")]
    if not lines:
        print(path, "no record")
        continue
    d = json.loads(lines[0])
    b = d["config"].get("burn_in") or {}
    print(path)
    print("  value %.0f rays/s  %.4f ms/step  | window %s  late %s  | roofline.frac %s" % (d["value"], d["ms_per_step"], (d.get("window_1000_2000") or {}).get("ms_per_step"),
          (d.get("late_regime") or {}).get("ms_per_step"), (d.get("roofline") or {}).get("frac")))
    print("  burn-in: %s, state %s | parity_mode %s | deterministic_mode %s end state %s" % (b.get("mode"), (b.get("state_sha256") or "")[:16], (d.get("parity_mode") or {}).get("ms_per_step"),
          (d.get("deterministic_mode") or {}).get("ms_per_step"), ((d.get("deterministic_mode") or {}).get("end_state_sha256") or "")[:16]))
    print("  timed steps ms:", d.get("timed_steps_ms"))
