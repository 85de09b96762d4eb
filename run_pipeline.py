#!/usr/bin/env python3
"""Command-line entry of the reconstruction pipeline (same options as the reference's run_pipeline.py:27-92):

    python run_pipeline.py --input data/scene/ --testbed build/testbed --output out/scene
    python run_pipeline.py --input normals.sfm --testbed build/testbed --albedo-sfm albedos.sfm --mask-sfm masks.sfm --has-albedo
"""
import argparse

import numpy as np

from rnb_neus2_amd.pipeline import run_full_pipeline

OPTIONS = [
    (("--input", "-i"), dict(required=True, help="Input data: directory (cameras.npz), .npz, .sfm, or .json")),
    (("--testbed", "-t"), dict(required=True, help="Path to the testbed binary")),
    (("--output", "-o"), dict(default="output", help="Output directory (default: output)")),
    (("--max-steps",), dict(type=int, default=10000, help="Max training steps (default: 10000)")),
    (("--mesh-resolution",), dict(type=int, default=1024, help="Marching cubes resolution (default: 1024)")),
    (("--scaling-mode",), dict(default="auto", choices=["auto", "pcd", "silhouettes", "silhouettes_v2", "cameras", "none"], help="Scene normalization mode (default: auto)")),
    (("--sphere-scale",), dict(type=float, default=1.0, help="Target sphere radius (default: 1.0)")),
    (("--margin-px",), dict(type=int, default=20, help="Pixel margin for silhouettes_v2 (default: 20)")),
    (("--warmup-ratio",), dict(type=float, default=0.1, help="Phase 1 ratio for albedo mode (default: 0.1)")),
    (("--mask-weight",), dict(type=float, default=1.0, help="Mask loss weight (default: 1.0)")),
    (("--has-albedo",), dict(action="store_true", help="Enable two-phase training with albedo scaling")),
    (("--albedo-sfm",), dict(default="", help="Path to albedo SfMData (SfM mode)")),
    (("--mask-sfm",), dict(default="", help="Path to mask SfMData (SfM mode)")),
    (("--mask-folder",), dict(default="", help="Folder with mask images")),
    (("--supernormal",), dict(action="store_true", help="Enable SuperNormal mode")),
    (("--l1",), dict(action="store_true", help="Use L1 norm for color loss")),
    (("--no-rgbplus",), dict(action="store_true", help="Disable RGB+ normalization")),
    (("--n-samples",), dict(type=int, default=2000, help="Samples for albedo scaling (default: 2000)")),
    (("--seed",), dict(type=int, default=0, help="Random seed (default: 0)")),
]


def build_parser():
    parser = argparse.ArgumentParser(description="RNb-NeuS2 on MI355X: neural surface reconstruction pipeline")
    for names, kw in OPTIONS:
        parser.add_argument(*names, **kw)
    return parser


def pipeline_kwargs(args):
    return dict(input_path=args.input, testbed_path=args.testbed, output_dir=args.output, max_steps=args.max_steps, mesh_resolution=args.mesh_resolution,
                scaling_mode=args.scaling_mode, sphere_scale=args.sphere_scale, margin_px=args.margin_px, warmup_ratio=args.warmup_ratio,
                mask_weight=args.mask_weight, super_normal=args.supernormal, use_l1=args.l1, use_rgb_plus=not args.no_rgbplus, has_albedo=args.has_albedo,
                albedo_sfm_path=args.albedo_sfm, mask_sfm_path=args.mask_sfm, mask_folder_path=args.mask_folder, n_samples=args.n_samples)


def main(argv=None):
    args = build_parser().parse_args(argv)
    np.random.seed(args.seed)
    return run_full_pipeline(**pipeline_kwargs(args))


if __name__ == "__main__":
    main()
