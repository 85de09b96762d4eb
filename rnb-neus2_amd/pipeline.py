"""Pipeline orchestration around the `testbed` executable (mirror of rnb_neus2/pipeline.py: same entry points, same
arguments, same testbed command lines — pinned against the reference's own by tests/golden/pipeline_argv.json).

Structure: every workflow is first *planned* as a list of `Stage`s (argv + what it must leave behind), then executed.
The plan functions are pure, which is what the tests compare with the recorded reference behaviour."""
import os
import shutil
import subprocess
from dataclasses import dataclass, field
from pathlib import Path
from typing import List


class SimpleLogger:
    def info(self, msg):
        print("[INFO] {}".format(msg))

    def warning(self, msg):
        print("[WARN] {}".format(msg))

    def error(self, msg):
        print("[ERROR] {}".format(msg))


@dataclass
class Stage:
    name: str
    max_iter: int
    flags: List[str] = field(default_factory=list)

    def argv(self, testbed_path, scene_path):
        return [testbed_path, "--scene", str(scene_path) + "/", "--maxiter", str(self.max_iter), "--no-gui"] + list(self.flags)


def stage1_iterations(max_steps):
    return int(max_steps * 2 / 3)  # pipeline.py:63


def warmup_iterations(max_steps, warmup_ratio):
    return max(int(max_steps * warmup_ratio), 1000)  # pipeline.py:116


def snapshot_candidates(data_dir, iters):
    """Where a stage's snapshot is looked for, in order. (pipeline.py:77-85)"""
    name = "snapshot_{}.msgpack".format(iters)
    return [os.path.join(data_dir, "output", name), os.path.join(data_dir, name)]


def common_flags(mask_weight=1.0, super_normal=False, use_l1=False, use_rgb_plus=True):
    """run_full_pipeline's keyword -> flag mapping. (pipeline.py:281-288)"""
    flags = ["--mask-weight", str(mask_weight)]
    if super_normal:
        flags.append("--supernormal")
    if use_l1:
        flags.append("--lone")
    if not use_rgb_plus:
        flags.append("--no-rgbplus")
    return flags


def plan_two_stage(data_dir, max_steps, flags, resolution=1024, no_albedo=False, extra_flags=None):
    """Stage 1 trains 2/3 of the budget and saves a snapshot; stage 2 resumes from it with optimal lights up to
    `max_steps`, then meshes. (pipeline.py:56-103)"""
    tail = (["--no-albedo"] if no_albedo else []) + list(extra_flags or [])
    it1 = stage1_iterations(max_steps)
    snapshot = snapshot_candidates(data_dir, it1)[0]
    return [Stage("Stage 1", it1, list(flags) + ["--save-snapshot"] + tail),
            Stage("Stage 2", max_steps, list(flags) + ["--opti-lights", "--snapshot", snapshot, "--resolution", str(resolution), "--save-mesh", "--save-snapshot", "--free-memory"] + tail)]


def plan_warmup(max_steps, flags, warmup_ratio=0.1):
    """Geometry-only phase before albedo scaling: normals only, 512^3 mesh. (pipeline.py:116-128)"""
    return Stage("Phase 1 (warmup)", warmup_iterations(max_steps, warmup_ratio), list(flags) + ["--no-albedo", "--save-mesh", "--resolution", "512", "--free-memory"])


def run_testbed(testbed_path, scene_path, max_iter, flags, stage_name, logger=None):
    """One testbed process; stdout is relayed line by line, a non-zero exit raises. (pipeline.py:27-53)"""
    logger = logger or SimpleLogger()
    cmd = Stage(stage_name, max_iter, list(flags)).argv(testbed_path, scene_path)
    logger.info("{} command: {}".format(stage_name, " ".join(cmd)))
    result = subprocess.run(cmd, capture_output=True, text=True)
    if result.stdout:
        for line in result.stdout.strip().split("\n"):
            logger.info(line)
    if result.returncode != 0:
        if result.stderr:
            logger.error(result.stderr)
        raise RuntimeError("{} failed with code {}".format(stage_name, result.returncode))
    logger.info("{} completed".format(stage_name))


def run_two_stage(testbed_path, data_dir, max_steps, common_flags, resolution=1024, no_albedo=False, extra_flags=None, logger=None):
    logger = logger or SimpleLogger()
    first, second = plan_two_stage(data_dir, max_steps, common_flags, resolution, no_albedo, extra_flags)
    logger.info("Stage 1: {} iterations".format(first.max_iter))
    run_testbed(testbed_path, data_dir, first.max_iter, first.flags, first.name, logger)
    found = [p for p in snapshot_candidates(data_dir, first.max_iter) if os.path.exists(p)]
    if not found:
        raise RuntimeError("Snapshot not found after {} iterations".format(first.max_iter))
    second.flags[second.flags.index("--snapshot") + 1] = found[0]
    logger.info("Stage 2: {} iterations (opti-lights)".format(max_steps))
    run_testbed(testbed_path, data_dir, second.max_iter, second.flags, second.name, logger)


def find_phase1_mesh(data_dir, warmup_steps):
    """mesh_<steps>.obj in <data_dir>/output, else the newest mesh_*.obj there. (pipeline.py:131-141)"""
    out = os.path.join(data_dir, "output")
    path = os.path.join(out, "mesh_{}.obj".format(warmup_steps))
    if os.path.exists(path):
        return path
    candidates = list(Path(out).glob("mesh_*.obj"))
    if not candidates:
        raise RuntimeError("Phase 1 mesh not found in {}".format(out))
    return str(max(candidates, key=lambda p: p.stat().st_mtime))


def run_with_albedo_scaling(testbed_path, data_dir, max_steps, common_flags, resolution=1024, warmup_ratio=0.1, n_samples=2000, logger=None):
    """Warm-up (geometry) -> per-view albedo gains from the warm-up mesh -> albedos/ replaced by the scaled set and the
    warm-up output removed -> the two-stage run with albedo. (pipeline.py:106-175)"""
    logger = logger or SimpleLogger()
    from .albedo_scaling import compute_albedo_scale_ratios, scale_and_save_albedos

    warm = plan_warmup(max_steps, common_flags, warmup_ratio)
    logger.info("=== Phase 1: Geometry only ({} steps) ===".format(warm.max_iter))
    run_testbed(testbed_path, data_dir, warm.max_iter, warm.flags, warm.name, logger)
    mesh_path = find_phase1_mesh(data_dir, warm.max_iter)
    logger.info("=== Albedo scaling ===")
    albedo_dir, scaled_dir = os.path.join(data_dir, "albedos"), os.path.join(data_dir, "albedos_scaled")
    ratios = compute_albedo_scale_ratios(albedo_path=albedo_dir, camera_source=os.path.join(data_dir, "transform.json"), mesh_path=mesh_path, n_samples=n_samples, logger=logger)
    scale_and_save_albedos(albedo_path=albedo_dir, output_albedo_path=scaled_dir, scale_ratios=ratios, logger=logger)
    shutil.rmtree(albedo_dir)
    os.rename(scaled_dir, albedo_dir)
    logger.info("Albedos scaled and replaced")
    shutil.rmtree(os.path.join(data_dir, "output"), ignore_errors=True)
    logger.info("=== Phase 3: Full training with scaled albedos ===")
    run_two_stage(testbed_path, data_dir, max_steps, common_flags, resolution=resolution, logger=logger)


def find_output_mesh(data_dir):
    """Newest mesh_*.o* in <data_dir>/output, else in <data_dir> (json/txt/msgpack excluded). (pipeline.py:185-198)"""
    out = os.path.join(data_dir, "output")
    files = list(Path(out).glob("mesh_*.o*")) if os.path.isdir(out) else []
    if not files:
        files = list(Path(data_dir).glob("mesh_*.o*"))
    files = [f for f in files if f.suffix not in (".json", ".txt", ".msgpack")]
    if not files:
        raise RuntimeError("No mesh files in {} or {}".format(out, data_dir))
    return max(files, key=lambda p: p.stat().st_mtime)


def postprocess_mesh(data_dir, output_mesh_path, logger=None):
    """Keep the largest connected component (by area), orient normals outward, export OBJ, drop the training output
    directory. (pipeline.py:178-219)"""
    logger = logger or SimpleLogger()
    from .meshproc import load_obj, save_obj

    mesh_file = find_output_mesh(data_dir)
    logger.info("Post-processing: {}".format(mesh_file.name))
    mesh = load_obj(str(mesh_file))
    parts = mesh.split()
    if len(parts) > 1:
        mesh = max(parts, key=lambda m: m.area)
        logger.info("Kept largest component ({} vertices)".format(len(mesh.vertices)))
    mesh.fix_normals()
    os.makedirs(os.path.dirname(output_mesh_path) or ".", exist_ok=True)
    save_obj(output_mesh_path, mesh)
    logger.info("Mesh exported to: {}".format(output_mesh_path))
    shutil.rmtree(os.path.join(data_dir, "output"), ignore_errors=True)


def run_full_pipeline(input_path, testbed_path, output_dir, max_steps=10000, mesh_resolution=1024, scaling_mode="auto", sphere_scale=1.0, margin_px=20,
                      warmup_ratio=0.1, mask_weight=1.0, super_normal=False, use_l1=False, use_rgb_plus=True, has_albedo=False,
                      albedo_sfm_path="", mask_sfm_path="", mask_folder_path="", n_samples=2000, logger=None):
    """load -> prepare (<output_dir>/prepared_data) -> train (two-stage, or warm-up + albedo scaling + two-stage when
    `has_albedo`) -> post-process to <output_dir>/mesh.obj, which is returned. (pipeline.py:222-305)"""
    logger = logger or SimpleLogger()
    from .dataloaders import load_data
    from .prepare import prepare_testbed_data

    logger.info("=== Loading data from {} ===".format(input_path))
    data = load_data(input_path, albedo_sfm_path=albedo_sfm_path, mask_sfm_path=mask_sfm_path, mask_folder_path=mask_folder_path, logger=logger)
    data_dir = os.path.join(output_dir, "prepared_data")
    logger.info("=== Preparing testbed data ===")
    prepare_testbed_data(data, data_dir, logger, scaling_mode=scaling_mode, sphere_scale=sphere_scale, margin_px=margin_px)
    flags = common_flags(mask_weight, super_normal, use_l1, use_rgb_plus)
    if has_albedo:
        run_with_albedo_scaling(testbed_path, data_dir, max_steps, flags, resolution=mesh_resolution, warmup_ratio=warmup_ratio, n_samples=n_samples, logger=logger)
    else:
        run_two_stage(testbed_path, data_dir, max_steps, flags, resolution=mesh_resolution, no_albedo=True, logger=logger)
    output_mesh = os.path.join(output_dir, "mesh.obj")
    postprocess_mesh(data_dir, output_mesh, logger)
    logger.info("=== Pipeline complete ===")
    return output_mesh
