#!/usr/bin/env python3
"""How many global atomics would the hash-grid gradient scatter issue if addends were merged ACROSS rays, not only along a ray?

Trains config 4 to a step, reads the compacted samples of the next step and counts, per level: samples, runs of equal cells along the compacted
(ray-major) order -- what k_grid_scatter_quad_rl / _quad issue today, times the corners -- and the distinct cells / distinct table entries over the
whole batch -- what a perfect merge (samples ordered by cell) would issue. Prints one JSON object. Measurement only; nothing in the product reads it.
    python tools/scatter_merge_bound.py [--step 1000]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step", type=int, default=1000)
    args = ap.parse_args()
    from rnb_neus2_amd import api, synthetic, dp

    ctx = api.Context(apply_no_albedo=1, mask_loss_weight=1.0)
    ctx.init_params()
    ctx.set_dataset(*synthetic.make_scene(64, 800))
    tr = dp.DataParallelTrainer(ctx)
    st = None
    for _ in range(args.step):
        st = tr.step()
    n = int(st.measured_batch_size)
    coords = ctx.get("COORDS_COMPACTED").reshape(-1, 7)[:n]
    off, res, scale = ctx.grid_tables()
    xyz = coords[:, :3].astype(np.float32)
    P = (np.uint64(1), np.uint64(2654435761), np.uint64(805459861))
    out = {"step": int(ctx.training_step), "samples": n, "rays": int(st.rays_per_batch), "levels": []}
    for l in range(len(res)):
        pos = xyz * np.float32(scale[l]) + np.float32(0.5)
        cell = np.floor(pos).astype(np.int64)
        r = int(res[l])
        size = int(off[l + 1] - off[l])
        key = (cell[:, 0] * r + cell[:, 1]) * r + cell[:, 2]
        runs = 1 + int(np.count_nonzero(key[1:] != key[:-1]))
        uniq_cells = int(np.unique(key).size)
        dense = r ** 3 <= size
        ent = []
        for c in range(8):
            cc = cell + np.array([c & 1, (c >> 1) & 1, (c >> 2) & 1])
            if dense:
                e = (cc[:, 0] + cc[:, 1] * r + cc[:, 2] * r * r) % size
            else:
                u = cc.astype(np.uint64)
                e = ((u[:, 0] * P[0]) ^ (u[:, 1] * P[1]) ^ (u[:, 2] * P[2])) & np.uint64(0xFFFFFFFF)
                e = e % np.uint64(size)
            ent.append(e.astype(np.int64))
        ent = np.stack(ent, 1)
        uniq_entries = int(np.unique(ent).size)
        # 32-byte lines of the fp32 gradient table (8 bytes per entry): what the memory side counts per atomic instruction lane group
        uniq_lines = int(np.unique(ent >> 2).size)
        change = np.concatenate([[True], key[1:] != key[:-1]])
        # lines per run: distinct 32-byte lines among the 8 corners of a cell, summed over the runs
        el = ent[change] >> 2
        el.sort(axis=1)
        lines_per_run = 1 + np.count_nonzero(el[:, 1:] != el[:, :-1], axis=1)
        out["levels"].append({"level": l, "res": r, "dense": bool(dense), "table": size, "runs_along_rays": runs, "distinct_cells": uniq_cells,
                              "lines_issued_today": int(lines_per_run.sum()), "distinct_entries": uniq_entries, "distinct_lines": uniq_lines,
                              "bound_ratio_lines": round(uniq_lines / max(1, int(lines_per_run.sum())), 4)})
    t_today = sum(x["lines_issued_today"] for x in out["levels"])
    t_bound = sum(x["distinct_lines"] for x in out["levels"])
    out["total_lines_today_all_levels"] = t_today
    out["total_lines_perfect_merge"] = t_bound
    print(json.dumps(out))


if __name__ == "__main__":
    main()
