"""ctypes declarations of include/rnb_neus2.h (structs, enums, prototypes).

``declare(lib, prefix)`` attaches argtypes/restypes to an already loaded CDLL. The product always uses the
prefix ``rnb_`` (librnb_neus2_hip.so); the test suite applies the same declarations to its CPU checker, which exports
the identical signatures under its own prefix.
"""
import ctypes as C

ABI_VERSION = 5

# status codes (rnb_status)
OK, ERR_INVALID, ERR_DEVICE, ERR_NOMEM, ERR_NO_SAMPLES = 0, -1, -2, -3, -4

N_SDF_MLP_PARAMS = 3072
N_RGB_MLP_PARAMS = 8192
N_VARIANCE_PARAMS = 4
GRIDSIZE = 128
CASCADES = 8


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_levels", C.c_uint32),
        ("log2_hashmap_size", C.c_uint32),
        ("base_resolution", C.c_uint32),
        ("per_level_scale", C.c_float),
        ("valid_level_scale", C.c_float),
        ("base_valid_level_scale", C.c_float),
        ("base_training_step", C.c_uint32),
        ("sdf_bias", C.c_float),
        ("target_batch_size", C.c_uint32),
        ("initial_rays_per_batch", C.c_uint32),
        ("max_rays_per_batch", C.c_uint32),
        ("aabb_scale", C.c_uint32),
        ("seed", C.c_uint32),
        ("mask_loss_weight", C.c_float),
        ("ek_loss_weight", C.c_float),
        ("apply_L2", C.c_uint32),
        ("apply_rgbplus", C.c_uint32),
        ("apply_no_albedo", C.c_uint32),
        ("apply_light_opti", C.c_uint32),
        ("apply_supernormal", C.c_uint32),
        ("apply_relu", C.c_uint32),
        ("apply_bce", C.c_uint32),
        ("snap_to_pixel_centers", C.c_uint32),
        ("learning_rate", C.c_float),
        ("beta1", C.c_float),
        ("beta2", C.c_float),
        ("epsilon", C.c_float),
        ("l2_reg", C.c_float),
        ("ema_decay", C.c_float),
        ("lr_decay_start", C.c_uint32),
        ("lr_decay_interval", C.c_uint32),
        ("lr_decay_base", C.c_float),
        ("density_grid_decay", C.c_float),
        ("world_size", C.c_uint32),
        ("rank", C.c_uint32),
        ("only_sdf_training", C.c_uint32),
        ("overlap", C.c_uint32),
        ("accumulate", C.c_uint32),
        ("deterministic", C.c_uint32),
        ("reserved", C.c_uint32 * 4),
    ]


class View(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("focal_length", C.c_float * 2),
        ("principal_point", C.c_float * 2),
        ("xform", C.c_float * 12),
    ]


class StepStats(C.Structure):
    _fields_ = [
        ("training_step", C.c_uint32),
        ("rays_per_batch", C.c_uint32),
        ("next_rays_per_batch", C.c_uint32),
        ("measured_batch_size", C.c_uint32),
        ("measured_batch_size_before_compaction", C.c_uint32),
        ("n_rays_kept", C.c_uint32),
        ("density_grid_updated", C.c_uint32),
        ("loss", C.c_float),
        ("ek_loss", C.c_float),
        ("mask_loss", C.c_float),
        ("prep_ms", C.c_float),
        ("step_ms", C.c_float),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


# rnb_buffer_id
BUF = dict(
    PARAMS_FP32=0, PARAMS_FP16=1, PARAMS_EMA=2, GRADS_FP32=3, ADAM_M=4, ADAM_V=5, ADAM_STEPS=6,
    DENSITY_GRID=7, DENSITY_BITFIELD=8, DENSITY_MEAN=9, RAY_INDICES=10, RAYS=11, NUMSTEPS=12,
    COORDS=13, MLP_OUT=14, DLOSS_DOUT=15, COORDS_COMPACTED=16, LOSS=17, EK_LOSS=18, MASK_LOSS=19,
    COUNTERS=20, DENSITY_GRID_TMP=21, GRID_SAMPLE_POS=22, GRID_SAMPLE_IDX=23, STEP_VECTOR=24, GRID_SAMPLE_POS_EVAL=25, GRID_SAMPLE_IDX_EVAL=26, GRADS_FP16=27,
)
BUF_DTYPE = dict(
    PARAMS_FP32="f4", PARAMS_FP16="f2", PARAMS_EMA="f2", GRADS_FP32="f4", ADAM_M="f4", ADAM_V="f4", ADAM_STEPS="u4",
    DENSITY_GRID="f4", DENSITY_BITFIELD="u1", DENSITY_MEAN="f4", RAY_INDICES="u4", RAYS="f4", NUMSTEPS="u4",
    COORDS="f4", MLP_OUT="f2", DLOSS_DOUT="f2", COORDS_COMPACTED="f4", LOSS="f4", EK_LOSS="f4", MASK_LOSS="f4",
    COUNTERS="u4", DENSITY_GRID_TMP="f4", GRID_SAMPLE_POS="f4", GRID_SAMPLE_IDX="u4", STEP_VECTOR="f8", GRID_SAMPLE_POS_EVAL="f4", GRID_SAMPLE_IDX_EVAL="u4", GRADS_FP16="f2",
)
ACCUM_FP32, ACCUM_HALF = 0, 1  # rnb_accumulate
BUF_READONLY = 0x100  # RNB_BUF_READONLY
GRID_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)  # rnb_grid_exchange_fn(user, grid_tmp, n_elements, stream)
PRIM = dict(PCG32=0, MORTON=1, SRGB=2, RAY_BOX=3, MARCH=4, ACTIVATION=5, WARP=6, LOSS=7, PIXEL=8, GRID=9, READ_RGBA=10, CAMERA_RAY=11, RAY_TARGETS=12, LOSS_SAMPLE=13, RAY_LOSS=14, ENCODE=15, MARCH_RAY=16, SDF_DENSITY=17, PREP_DUE=18, DW_SLICED=19)  # rnb_primitive
PRIM_IN_WORDS, PRIM_OUT_WORDS = (6, 3, 1, 8, 9, 1, 9, 9, 9, 7, 32, 20, 35, 37, 16, 263, 10, 2, 1, 33028), (4, 4, 2, 3, 7, 3, 11, 5, 3, 3, 5, 9, 7, 28, 9, 16, 23, 1, 2, 16)
H2D, D2H, D2D = 0, 1, 2

_ctx = C.c_void_p
_stream = C.c_void_p
_u32, _u64, _i = C.c_uint32, C.c_uint64, C.c_int

# name -> (restype, argtypes); every name here must be exported by the library (tests check this against the header)
PROTOTYPES = {
    "last_error": (C.c_char_p, []),
    "abi_version": (_u32, []),
    "default_config": (_i, [C.POINTER(Config)]),
    "create": (_i, [C.POINTER(Config), C.POINTER(_ctx)]),
    "destroy": (_i, [_ctx]),
    "update_config": (_i, [_ctx, C.POINTER(Config)]),
    "n_params": (_u64, [_ctx]),
    "param_layout": (_i, [_ctx, C.POINTER(_u64)]),
    "grid_tables": (_i, [_ctx, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.c_float)]),
    "init_params": (_i, [_ctx, C.POINTER(C.c_float)]),
    "set_params": (_i, [_ctx, C.POINTER(C.c_float)]),
    "buffer": (_i, [_ctx, _i, C.POINTER(C.c_void_p), C.POINTER(_u64)]),
    "params_changed": (_i, [_ctx]),
    "bitfield_changed": (_i, [_ctx]),
    "memcpy": (_i, [_ctx, C.c_void_p, C.c_void_p, _u64, _i]),
    "device_malloc": (_i, [_ctx, _u64, C.POINTER(C.c_void_p)]),
    "device_free": (_i, [_ctx, C.c_void_p]),
    "set_dataset": (_i, [_ctx, _u32, C.POINTER(View), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "set_training_step": (_i, [_ctx, _u32]),
    "valid_level": (_u32, [_ctx]),
    "update_density_grid": (_i, [_ctx, _stream]),
    "update_density_bitfield": (_i, [_ctx, _stream]),
    "update_density_grid_begin": (_i, [_ctx, _stream]),
    "update_density_grid_end": (_i, [_ctx, _stream]),
    "set_grid_exchange": (_i, [_ctx, C.c_void_p, C.c_void_p]),
    "density": (_i, [_ctx, _stream, C.c_void_p, _u32, C.c_void_p, _i]),
    "sdf": (_i, [_ctx, _stream, C.c_void_p, _u32, C.c_void_p, _i]),
    "forward_infer": (_i, [_ctx, _stream, C.c_void_p, _u32, C.c_void_p, _i]),
    "sdf_lattice": (_i, [_ctx, _stream, C.POINTER(_u32), C.c_float, C.c_float, C.c_void_p, _i]),
    "marching_cubes": (_i, [_ctx, _stream, C.c_void_p, C.POINTER(_u32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(_u32), C.POINTER(_u32)]),
    "generate_training_samples": (_i, [_ctx, _stream, _u32, _u32, _u32]),
    "compute_loss": (_i, [_ctx, _stream, _u32, _u32]),
    "forward_backward": (_i, [_ctx, _stream]),
    "optimizer_step": (_i, [_ctx, _stream]),
    "train_step": (_i, [_ctx, _stream, C.POINTER(StepStats)]),
    "train_step_begin": (_i, [_ctx, _stream]),
    "train_step_end": (_i, [_ctx, _stream, C.POINTER(StepStats)]),
    "profile_enable": (_i, [_ctx, _i]),
    "profile_count": (_i, [_ctx]),
    "profile_get": (_i, [_ctx, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_u64), C.POINTER(C.c_double)]),
    "train_step_apply": (_i, [_ctx, _stream]),
    "train_step_local": (_i, [_ctx, _stream, C.POINTER(_u64), C.POINTER(C.c_double)]),
    "train_step_finish": (_i, [_ctx, C.POINTER(_u64), C.POINTER(C.c_double), C.POINTER(StepStats)]),
    "training_step": (_u32, [_ctx]),
    "rays_per_batch": (_u32, [_ctx]),
    "set_controller": (_i, [_ctx, _u32, _u32, _u32, _u32]),
    "set_optimizer_step": (_i, [_ctx, _u32]),
    "eval_primitives": (_i, [_ctx, _i, C.c_void_p, _u32, C.c_void_p]),
    "gradient_parts": (_i, [_ctx, C.POINTER(_u64 * 2 * 3), C.POINTER(_u32)]),
    "gradient_part_wait": (_i, [_ctx, _u32, _stream]),
    "train_step_apply_early": (_i, [_ctx, _stream]),
    "shard_layout": (_i, [_ctx, C.POINTER(_u64 * 4 * 3), C.POINTER(_u32), C.POINTER(_u64)]),
    "train_step_apply_shard": (_i, [_ctx, _u32, _stream]),
    "train_step_apply_done": (_i, [_ctx, _stream]),
}


class Functions:
    """Bound, typed entry points of one library."""

    def __init__(self, lib, prefix):
        self.lib = lib
        self.prefix = prefix
        missing = []
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(lib, prefix + name)
            except AttributeError:
                missing.append(prefix + name)
                continue
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        if missing:
            raise ImportError("library %s lacks symbols: %s" % (getattr(lib, "_name", lib), ", ".join(missing)))


def declare(lib, prefix="rnb_"):
    return Functions(lib, prefix)
