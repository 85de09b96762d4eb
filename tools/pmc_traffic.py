"""FETCH_SIZE / WRITE_SIZE summaries (tools/collect_pmc.sh) -> bytes per training step per bench kernel group."""
import json
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
out_path = sys.argv[2] if len(sys.argv) > 2 else "profiles/r05_pmc_traffic.json"
F = json.load(open(d + "/FETCH_SIZE.json"))["kernels"]
W = json.load(open(d + "/WRITE_SIZE.json"))["kernels"]
fb = [k for k in F if k.startswith("k_fwd_bwd")]
steps_total = sum(F[k]["launches_total"] for k in fb)  # one k_fwd_bwd* launch per step


def per_step(T, ks):
    # average of the kernel's last launches x its launches per step (k_forward_chained runs twice a step; rounds 2-3 divided the sum over the last 100 launches of
    # every kernel by 100 steps, which halved the two-launch kernels)
    return sum(T[k]["avg"] * T[k]["launches_total"] / steps_total for k in ks if k in T) * 1024


groups = {"k_forward": [k for k in F if k.startswith("k_forward_chained")], "k_fwd_bwd": fb, "k_grid_scatter": [k for k in F if k.startswith("k_grid_scatter")],
          "k_grid_scatter_lds": [k for k in F if k.startswith("k_grid_scatter_lds")], "k_grid_scatter_quad_rl": [k for k in F if k.startswith("k_grid_scatter_quad_rl")],
          "k_grid_scatter_quad": [k for k in F if k.startswith("k_grid_scatter_quad") and not k.startswith("k_grid_scatter_quad_rl")],
          "k_adam_ema": ["k_adam_ema"], "k_dw*7+k_dw_finish": [k for k in F if "k_dw" in k], "k_loss_pass1": ["k_loss_pass1", "k_loss_pass1_heads"],
          "k_loss_pass2+k_rollover": [k for k in F if k.startswith("k_loss_pass2")], "k_march_count": [k for k in F if k.startswith("k_march_count")], "k_march_write": [k for k in F if k.startswith("k_march_write")],
          "k_scan_rays": [k for k in F if k.startswith("k_scan_rays")], "k_scan_compact": [k for k in F if k.startswith("k_scan_compact")]}
out = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernels serialised, cfg.overlap=0), bench.py --steps 10 after 2000 burn-in steps; tools/collect_pmc.sh + tools/pmc_traffic.py",
       "_units": "bytes per training step AS COUNTED (counter value in KiB x 1024). FETCH_SIZE = read requests x 64 B on gfx950: exact for scattered 8-byte gathers (64-byte requests), half of the bytes of "
                 "coalesced streams (128-byte requests) -- profiles/r05_counter_calibration.json; bench.py applies the factor per kernel group (FETCH_CORRECTION) when it reads this file. WRITE_SIZE is exact; an atomic "
                 "without return is one write request counted as 32 B.",
       "per_step": {}}
for g, ks in groups.items():
    f, w = per_step(F, ks), per_step(W, ks)
    out["per_step"][g] = {"fetch_bytes": round(f), "write_bytes": round(w), "total_bytes": round(f + w)}
json.dump(out, open(out_path, "w"), indent=1)
print({g: round(v["total_bytes"] / 1e6, 1) for g, v in out["per_step"].items()})
