#!/usr/bin/env python3
"""Reconstruction pipeline from the command line. The option set is the reference's (run_pipeline.py:27-92), so existing
invocations keep working:

    python run_pipeline.py --input data/scene/ --testbed build/testbed --output out/scene
    python run_pipeline.py --input normals.sfm --testbed build/testbed --albedo-sfm albedos.sfm --mask-sfm masks.sfm --has-albedo
"""
import argparse

import numpy as np

from rnb_neus2_amd.pipeline import run_full_pipeline

SCALING_MODES = ("auto", "pcd", "silhouettes", "silhouettes_v2", "cameras", "none")

# one line per option:  flags | kind | default | what it does        (kind: str / int / float / flag / mode; default "!" = required)
_SPEC = """
--input -i          | str   | !      | where the scene comes from: a folder holding cameras.npz, an .npz, an .sfm or a .json file
--testbed -t        | str   | !      | the training executable (build/testbed)
--output -o         | str   | output | folder that receives the prepared scene, snapshots and meshes
--max-steps         | int   | 10000  | number of training iterations over both stages
--mesh-resolution   | int   | 1024   | lattice size of the final marching-cubes extraction
--scaling-mode      | mode  | auto   | how the scene is brought into the unit sphere
--sphere-scale      | float | 1.0    | radius the normalised scene should fill
--margin-px         | int   | 20     | silhouettes_v2 only: slack around the masks, in pixels
--warmup-ratio      | float | 0.1    | --has-albedo only: share of the iterations spent before the albedos are rescaled
--mask-weight       | float | 1.0    | weight of the silhouette term of the loss
--has-albedo        | flag  |        | train in two phases and rescale the albedo maps in between
--albedo-sfm        | str   |        | SfMData file listing the albedo maps (SfM input)
--mask-sfm          | str   |        | SfMData file listing the masks (SfM input)
--mask-folder       | str   |        | directory of mask images
--supernormal       | flag  |        | SuperNormal variant of the shading loss
--l1                | flag  |        | L1 instead of L2 colour loss
--no-rgbplus        | flag  |        | switch the RGB+ channel off
--n-samples         | int   | 2000   | pixels per view used to estimate the albedo gains
--seed              | int   | 0      | seed of numpy's generator
"""


def build_parser():
    parser = argparse.ArgumentParser(description="RNb-NeuS2 pipeline on the MI355X training path")
    casts = {"int": int, "float": float}
    for line in _SPEC.strip().splitlines():
        flags, kind, default, text = (field.strip() for field in line.split("|"))
        kw = {"help": text}
        if kind == "flag":
            kw["action"] = "store_true"
        else:
            if kind in casts:
                kw["type"] = casts[kind]
            if kind == "mode":
                kw["choices"] = list(SCALING_MODES)
            if default == "!":
                kw["required"] = True
            else:
                kw["default"] = casts.get(kind, str)(default) if default else ""
        parser.add_argument(*flags.split(), **kw)
    return parser


def pipeline_kwargs(ns):
    """argparse namespace -> keywords of rnb_neus2_amd.pipeline.run_full_pipeline."""
    renamed = {"input": "input_path", "testbed": "testbed_path", "output": "output_dir", "supernormal": "super_normal", "l1": "use_l1",
               "albedo_sfm": "albedo_sfm_path", "mask_sfm": "mask_sfm_path", "mask_folder": "mask_folder_path"}
    kw = {renamed.get(k, k): v for k, v in vars(ns).items() if k not in ("seed", "no_rgbplus")}
    kw["use_rgb_plus"] = not ns.no_rgbplus
    return kw


def main(argv=None):
    ns = build_parser().parse_args(argv)
    np.random.seed(ns.seed)
    return run_full_pipeline(**pipeline_kwargs(ns))


if __name__ == "__main__":
    main()
