"""Host-side mirror of the reference's training interface over the C-ABI of include/rnb_neus2.h.

The reference drives this path through ``Testbed`` (src/testbed.cu:2776-2872, src/testbed_nerf.cu:3560-4138);
``Context`` keeps the same verbs (reset_network/init params, load dataset, train step, per-stage calls) and hands
everything to ``librnb_neus2_hip.so``. There is no CPU fallback: if the HIP library is missing, loading fails.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import Config, View, StepStats, BUF, BUF_DTYPE

__all__ = ["Context", "Config", "View", "StepStats", "load_library", "default_config", "library_path",
           "load_sdf_init_weights", "RnbError", "BUF"]

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO_ROOT = os.path.dirname(_PKG_DIR)
_LIB_NAME = "librnb_neus2_hip.so"


class RnbError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rnb error %d: %s" % (code, message))
        self.code = code


def library_path():
    return os.path.join(_PKG_DIR, _LIB_NAME)


_FUNCS = None


def load_library():
    """Load the HIP library. Fails loudly when it has not been built (no fallback path exists)."""
    global _FUNCS
    if _FUNCS is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "%s not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback for the hot path." % path)
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        _FUNCS = _abi.declare(lib, "rnb_")
        if _FUNCS.abi_version() != _abi.ABI_VERSION:
            raise RuntimeError("ABI version mismatch between %s and the Python host side" % path)
    return _FUNCS


def default_config(fns=None, **overrides):
    fns = fns or load_library()
    cfg = Config()
    rc = fns.default_config(C.byref(cfg))
    if rc != 0:
        raise RnbError(rc, fns.last_error().decode())
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError("rnb_config has no field %r" % k)
        setattr(cfg, k, v)
    return cfg


def load_sdf_init_weights(path=None):
    """The sphere-SDF initialisation of the density MLP (nerf_network.h:585-623)."""
    path = path or os.path.join(_REPO_ROOT, "utils", "mlp_weights_hidden_layer_num_1_hidden_size_32.txt")
    w = np.loadtxt(path, dtype=np.float64).astype(np.float32).ravel()
    if w.size != _abi.N_SDF_MLP_PARAMS:
        raise ValueError("expected %d SDF-MLP weights in %s, found %d" % (_abi.N_SDF_MLP_PARAMS, path, w.size))
    return np.ascontiguousarray(w)


def _stream_handle(stream):
    if stream is None:
        return None
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(getattr(stream, "cuda_stream"))  # torch.cuda.Stream


class Context:
    """One training context on the current HIP device (``rnb_ctx``)."""

    def __init__(self, cfg=None, fns=None, **overrides):
        self.f = fns or load_library()
        self.cfg = cfg if cfg is not None else default_config(self.f, **overrides)
        self._h = C.c_void_p()
        self._check(self.f.create(C.byref(self.cfg), C.byref(self._h)))
        self._keep = []

    # -- plumbing -------------------------------------------------------
    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise RnbError(rc, self.f.last_error().decode(errors="replace"))
        return rc

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.f.destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def update_config(self, **overrides):
        """Run-time switches (loss flags, weights, optimizer hyper-parameters, only_sdf_training)."""
        for k, v in overrides.items():
            if not hasattr(self.cfg, k):
                raise AttributeError("rnb_config has no field %r" % k)
            setattr(self.cfg, k, v)
        self._check(self.f.update_config(self._h, C.byref(self.cfg)))

    @property
    def n_params(self):
        return int(self.f.n_params(self._h))

    def param_layout(self):
        off = (C.c_uint64 * 5)()
        self._check(self.f.param_layout(self._h, off))
        return dict(sdf=off[0], rgb=off[1], grid=off[2], variance=off[3], end=off[4])

    def grid_tables(self):
        n = self.cfg.n_levels
        off = (C.c_uint32 * (n + 1))()
        res = (C.c_uint32 * n)()
        sc = (C.c_float * n)()
        self._check(self.f.grid_tables(self._h, off, res, sc))
        return np.array(off, dtype=np.uint32), np.array(res, dtype=np.uint32), np.array(sc, dtype=np.float32)

    def buffer(self, name, read_only=False):
        """(pointer, n_bytes) of a context buffer; a device pointer for the HIP library. Without ``read_only`` the library assumes the
        caller writes through the pointer before its next call (cached forms of weights / occupancy are dropped, the optimizer-state
        views are packed back); see rnb_buffer in include/rnb_neus2.h."""
        ptr = C.c_void_p()
        nb = C.c_uint64()
        self._check(self.f.buffer(self._h, BUF[name] | (_abi.BUF_READONLY if read_only else 0), C.byref(ptr), C.byref(nb)))
        return ptr.value, nb.value

    def params_changed(self):
        """Training weights were written through a pointer kept from buffer("PARAMS_FP16"): drop the cached LDS weight images."""
        self._check(self.f.params_changed(self._h))

    def bitfield_changed(self):
        """The occupancy bitfield was written through a pointer kept from buffer("DENSITY_BITFIELD") (put() calls this itself)."""
        self._check(self.f.bitfield_changed(self._h))

    def get(self, name, count=None, offset=0):
        """Copy (part of) a context buffer to a numpy array; count/offset in elements."""
        ptr, nb = self.buffer(name, read_only=True)
        dt = np.dtype(BUF_DTYPE[name])
        total = nb // dt.itemsize
        if count is None:
            count = total - offset
        if offset + count > total:
            raise ValueError("%s: range [%d, %d) exceeds %d elements" % (name, offset, offset + count, total))
        out = np.empty(count, dtype=dt)
        if count:
            self._check(self.f.memcpy(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr + offset * dt.itemsize),
                                      count * dt.itemsize, _abi.D2H))
        return out

    def put(self, name, array, offset=0):
        ptr, nb = self.buffer(name)
        dt = np.dtype(BUF_DTYPE[name])
        a = np.ascontiguousarray(array, dtype=dt).ravel()
        if (offset + a.size) * dt.itemsize > nb:
            raise ValueError("%s: write of %d elements at %d exceeds buffer" % (name, a.size, offset))
        if a.size:
            self._check(self.f.memcpy(self._h, C.c_void_p(ptr + offset * dt.itemsize), a.ctypes.data_as(C.c_void_p),
                                      a.size * dt.itemsize, _abi.H2D))
            if name == "DENSITY_BITFIELD":
                self.bitfield_changed()
            elif name == "PARAMS_FP16":
                self.params_changed()

    # -- parameters -----------------------------------------------------
    def init_params(self, sdf_weights=None):
        w = load_sdf_init_weights() if sdf_weights is None else np.ascontiguousarray(sdf_weights, dtype=np.float32)
        self._check(self.f.init_params(self._h, w.ctypes.data_as(C.POINTER(C.c_float))))

    def set_params(self, params_fp32):
        p = np.ascontiguousarray(params_fp32, dtype=np.float32).ravel()
        if p.size != self.n_params:
            raise ValueError("expected %d params, got %d" % (self.n_params, p.size))
        self._check(self.f.set_params(self._h, p.ctypes.data_as(C.POINTER(C.c_float))))

    # -- dataset --------------------------------------------------------
    def set_dataset(self, views, normals, albedos):
        """views: sequence of dicts(width, height, focal_length(2), principal_point(2), xform(3x4 c2w));
        normals/albedos: per view uint16 arrays [H, W, 4] (RGBA16, alpha = mask)."""
        n = len(views)
        if n == 0 or len(normals) != n or len(albedos) != n:
            raise ValueError("views / normals / albedos must be non-empty and of equal length")
        arr = (View * n)()
        nptr = (C.c_void_p * n)()
        aptr = (C.c_void_p * n)()
        keep = []
        for i, v in enumerate(views):
            arr[i].width = int(v["width"])
            arr[i].height = int(v["height"])
            arr[i].focal_length[:] = [float(x) for x in v["focal_length"]]
            arr[i].principal_point[:] = [float(x) for x in v["principal_point"]]
            arr[i].xform[:] = [float(x) for x in np.asarray(v["xform"], dtype=np.float32).reshape(12)]
            nm = np.ascontiguousarray(normals[i], dtype=np.uint16)
            al = np.ascontiguousarray(albedos[i], dtype=np.uint16)
            if nm.size != arr[i].width * arr[i].height * 4 or al.size != nm.size:
                raise ValueError("view %d: image size does not match width*height*4" % i)
            keep += [nm, al]
            nptr[i] = nm.ctypes.data
            aptr[i] = al.ctypes.data
        self._check(self.f.set_dataset(self._h, n, arr, nptr, aptr))

    # -- stages ---------------------------------------------------------
    def set_training_step(self, step):
        self._check(self.f.set_training_step(self._h, int(step)))

    @property
    def valid_level(self):
        return int(self.f.valid_level(self._h))

    @property
    def training_step(self):
        return int(self.f.training_step(self._h))

    @property
    def rays_per_batch(self):
        return int(self.f.rays_per_batch(self._h))

    def set_controller(self, training_step, rays_per_batch, measured_before_compaction=0, n_rays_total=0):
        self._check(self.f.set_controller(self._h, int(training_step), int(rays_per_batch), int(measured_before_compaction), int(n_rays_total)))

    def eval_primitives(self, kind, items):
        """rnb_eval_primitives: uint32 items [n, words_in(kind)] -> uint32 [n, words_out(kind)] (floats as bit patterns)."""
        k = _abi.PRIM[kind]
        a = np.ascontiguousarray(items, dtype=np.uint32).reshape(-1, _abi.PRIM_IN_WORDS[k])
        out = np.empty((a.shape[0], _abi.PRIM_OUT_WORDS[k]), dtype=np.uint32)
        self._check(self.f.eval_primitives(self._h, k, a.ctypes.data_as(C.c_void_p), a.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    # -- a training state as data (what a snapshot holds, src/testbed.cu:3333-3390, plus the optimizer's moments) ---------------------------
    def training_state(self, last_stats=None):
        """Everything a fresh context needs to continue this run: master weights, Adam moments and per-parameter step counts, EMA weights, the number of optimizer
        steps taken (= the learning-rate schedule's position), the occupancy grid and the ray controller. `last_stats`: the rnb_step_stats of the last step
        (its un-compacted sample count feeds the controller). The ray generator's position is not part of it (nor of the reference's snapshot)."""
        return dict(params=self.get("PARAMS_FP32").copy(), adam_m=self.get("ADAM_M").copy(), adam_v=self.get("ADAM_V").copy(), adam_steps=self.get("ADAM_STEPS").copy(),
                    ema=self.get("PARAMS_EMA").copy(), grid=self.get("DENSITY_GRID").copy(), step=self.training_step, rays=self.rays_per_batch,
                    before=int(last_stats.measured_batch_size_before_compaction) if last_stats is not None else 0)

    def load_training_state(self, state):
        """The inverse: set_params resets the optimizer (trainer.h:263-275), then the moments, step counts and EMA weights are put back, the schedule's position,
        the occupancy grid (and its bitfield) and the controller. Works across accumulate / deterministic modes: the state is mode-independent data."""
        self.set_params(state["params"])
        self.put("ADAM_M", state["adam_m"])
        self.put("ADAM_V", state["adam_v"])
        self.put("ADAM_STEPS", state["adam_steps"])
        self.put("PARAMS_EMA", state["ema"])
        self.set_optimizer_step(state["step"])
        self.put("DENSITY_GRID", state["grid"])
        self.update_density_bitfield()
        self.set_controller(state["step"], state["rays"], state["before"], 0)

    def set_optimizer_step(self, step):
        """Optimizer steps taken so far (adam.h:486-495, exponential_decay.h:143-147): step counter + learning-rate factor."""
        self._check(self.f.set_optimizer_step(self._h, int(step)))

    def gradient_parts(self):
        """[(first, last+1), ...] blocks of GRADS_FP32 in the order they become final during the queued backward pass."""
        arr = (C.c_uint64 * 2 * 3)()
        n = C.c_uint32()
        self._check(self.f.gradient_parts(self._h, C.byref(arr), C.byref(n)))
        return [(int(arr[k][0]), int(arr[k][1])) for k in range(n.value)]

    def train_step_apply_early(self, stream_handle):
        self._check(self.f.train_step_apply_early(self._h, C.c_void_p(stream_handle)))

    def gradient_part_wait(self, part, stream_handle):
        self._check(self.f.gradient_part_wait(self._h, int(part), C.c_void_p(stream_handle)))

    def shard_layout(self):
        """([(lo, hi, own_lo, own_hi), ...], capacity): blocks of the sharded data-parallel optimizer in completion order
        (rnb_shard_layout); every parameter-shaped buffer is allocated up to `capacity` elements."""
        arr = (C.c_uint64 * 4 * 3)()  # RNB_MAX_SHARD_PARTS
        n = C.c_uint32()
        cap = C.c_uint64()
        self._check(self.f.shard_layout(self._h, C.byref(arr), C.byref(n), C.byref(cap)))
        return [tuple(int(arr[k][j]) for j in range(4)) for k in range(n.value)], int(cap.value)

    def train_step_apply_shard(self, part, stream_handle=None):
        self._check(self.f.train_step_apply_shard(self._h, int(part), C.c_void_p(stream_handle)))

    def train_step_apply_done(self, stream_handle=None):
        self._check(self.f.train_step_apply_done(self._h, C.c_void_p(stream_handle)))

    def profile_enable(self, on=True):
        self._check(self.f.profile_enable(self._h, int(bool(on))))

    def profile(self):
        """Per-kernel-group HIP-event timings accumulated since profile_enable(True)."""
        out = []
        for i in range(self.f.profile_count(self._h)):
            name, ms, n, units = C.c_char_p(), C.c_double(), C.c_uint64(), C.c_double()
            self._check(self.f.profile_get(self._h, i, C.byref(name), C.byref(ms), C.byref(n), C.byref(units)))
            out.append(dict(kernel=name.value.decode(), total_ms=ms.value, launches=n.value, units=units.value))
        return out

    def update_density_grid(self, stream=None):
        self._check(self.f.update_density_grid(self._h, _stream_handle(stream)))

    def update_density_grid_begin(self, stream=None):
        """First half of an occupancy update: samples + network on this rank's share (rnb_update_density_grid_begin)."""
        self._check(self.f.update_density_grid_begin(self._h, _stream_handle(stream)))

    def update_density_grid_end(self, stream=None):
        self._check(self.f.update_density_grid_end(self._h, _stream_handle(stream)))

    def set_grid_exchange(self, fn):
        """fn(grid_tmp_ptr, n_elements, stream_handle) -> None takes the element-wise max of DENSITY_GRID_TMP over the data-parallel ranks (stream-ordered);
        None removes it (replicated occupancy updates). rnb_set_grid_exchange."""
        if fn is None:
            self._grid_cb = None
            self._check(self.f.set_grid_exchange(self._h, None, None))
            return

        def trampoline(_user, ptr, n, stream):
            try:
                fn(ptr, int(n), stream)
                return 0
            except Exception:  # no exception may cross the C boundary
                import traceback
                traceback.print_exc()
                return -1
        self._grid_cb = _abi.GRID_EXCHANGE_FN(trampoline)  # kept alive with the context
        self._check(self.f.set_grid_exchange(self._h, C.cast(self._grid_cb, C.c_void_p), None))

    def update_density_bitfield(self, stream=None):
        self._check(self.f.update_density_bitfield(self._h, _stream_handle(stream)))

    def _point_query(self, fn, xyz, inference):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        n = xyz.shape[0]
        cptr, cb = self.buffer("COORDS")
        optr, ob = self.buffer("MLP_OUT")
        if n * 12 > cb or n * 2 > ob:
            raise ValueError("too many points for the scratch buffers")
        self.put("COORDS", xyz)
        self._check(fn(self._h, None, C.c_void_p(cptr), n, C.c_void_p(optr), int(bool(inference))))
        return self.get("MLP_OUT", n)

    def density(self, xyz, inference=False):
        """NerfNetwork::density on host points [n,3] -> float16[n] (staged through the context's scratch)."""
        return self._point_query(self.f.density, xyz, inference)

    def sdf(self, xyz, inference=True):
        return self._point_query(self.f.sdf, xyz, inference)

    def forward_infer(self, coords, inference=False):
        """NerfNetwork::inference_mixed_precision on host coords [n,7] -> float16[n,16]."""
        coords = np.ascontiguousarray(coords, dtype=np.float32).reshape(-1, 7)
        n = coords.shape[0]
        cptr, cb = self.buffer("COORDS")
        optr, ob = self.buffer("MLP_OUT")
        if n * 28 > cb or n * 32 > ob:
            raise ValueError("too many samples for the scratch buffers")
        self.put("COORDS", coords)
        self._check(self.f.forward_infer(self._h, None, C.c_void_p(cptr), n, C.c_void_p(optr), int(bool(inference))))
        return self.get("MLP_OUT", n * 16).reshape(n, 16)

    # -- mesh extraction (src/testbed_nerf.cu:4218-4269, src/marching_cubes.cu:794-822) ----------------------
    def device_malloc(self, n_bytes):
        ptr = C.c_void_p()
        self._check(self.f.device_malloc(self._h, int(n_bytes), C.byref(ptr)))
        return ptr.value

    def device_free(self, ptr):
        self._check(self.f.device_free(self._h, C.c_void_p(ptr)))

    def sdf_lattice(self, res, lattice_min=0.0, lattice_max=1.0, inference=True):
        """SDF on the lattice res^3 (int or 3 ints, x fastest) of [lattice_min, lattice_max)^3 -> library-side pointer to
        float[res^3] (device memory for the HIP library); release with device_free."""
        r = (C.c_uint32 * 3)(*([int(res)] * 3 if np.isscalar(res) else [int(x) for x in res]))
        ptr = self.device_malloc(int(r[0]) * int(r[1]) * int(r[2]) * 4)
        self._check(self.f.sdf_lattice(self._h, None, r, float(lattice_min), float(lattice_max), C.c_void_p(ptr), int(bool(inference))))
        return ptr

    def marching_cubes(self, density_ptr, res, aabb_min=(0.0, 0.0, 0.0), aabb_max=(1.0, 1.0, 1.0), thresh=0.0):
        """Iso-surface of a library-side lattice -> (verts float32[n,3], indices uint32[m]) on the host."""
        r = (C.c_uint32 * 3)(*([int(res)] * 3 if np.isscalar(res) else [int(x) for x in res]))
        mn, mx = (C.c_float * 3)(*[float(x) for x in aabb_min]), (C.c_float * 3)(*[float(x) for x in aabb_max])
        pv, pi, nv, ni = C.c_void_p(), C.c_void_p(), C.c_uint32(), C.c_uint32()
        self._check(self.f.marching_cubes(self._h, None, C.c_void_p(density_ptr), r, mn, mx, float(thresh), C.byref(pv), C.byref(pi), C.byref(nv), C.byref(ni)))
        verts = np.empty((nv.value, 3), np.float32)
        idx = np.empty(ni.value, np.uint32)
        if nv.value:
            self._check(self.f.memcpy(self._h, verts.ctypes.data_as(C.c_void_p), pv, nv.value * 12, _abi.D2H))
        if ni.value:
            self._check(self.f.memcpy(self._h, idx.ctypes.data_as(C.c_void_p), pi, ni.value * 4, _abi.D2H))
        self.device_free(pv.value)
        self.device_free(pi.value)
        return verts, idx

    def upload(self, array):
        """numpy array -> library-side buffer (device_malloc + copy); release with device_free."""
        a = np.ascontiguousarray(array)
        ptr = self.device_malloc(a.nbytes)
        self._check(self.f.memcpy(self._h, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, _abi.H2D))
        return ptr

    def download(self, ptr, count, dtype):
        out = np.empty(int(count), dtype=dtype)
        self._check(self.f.memcpy(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, _abi.D2H))
        return out

    def forward_infer_staged(self, n, stream=None):
        """Evaluate the first n samples already in COORDS into MLP_OUT (what train_nerf_step does, testbed_nerf.cu:3967)."""
        cptr, _ = self.buffer("COORDS")
        optr, _ = self.buffer("MLP_OUT")
        self._check(self.f.forward_infer(self._h, _stream_handle(stream), C.c_void_p(cptr), int(n), C.c_void_p(optr), 0))

    def generate_training_samples(self, n_rays, n_rays_total=0, max_samples=None, stream=None):
        if max_samples is None:
            max_samples = self.cfg.target_batch_size * 16
        self._check(self.f.generate_training_samples(self._h, _stream_handle(stream), int(n_rays), int(n_rays_total), int(max_samples)))

    def compute_loss(self, n_rays, n_rays_total=0, stream=None):
        self._check(self.f.compute_loss(self._h, _stream_handle(stream), int(n_rays), int(n_rays_total)))

    def forward_backward(self, stream=None):
        self._check(self.f.forward_backward(self._h, _stream_handle(stream)))

    def optimizer_step(self, stream=None):
        self._check(self.f.optimizer_step(self._h, _stream_handle(stream)))

    def train_step(self, stream=None, allow_no_samples=False):
        st = StepStats()
        rc = self.f.train_step(self._h, _stream_handle(stream), C.byref(st))
        self._check(rc, allow=(_abi.ERR_NO_SAMPLES,) if allow_no_samples else ())
        return st

    def train_step_begin(self, stream=None):
        self._check(self.f.train_step_begin(self._h, _stream_handle(stream)))

    def train_step_apply(self, stream=None):
        self._check(self.f.train_step_apply(self._h, _stream_handle(stream)))

    def train_step_local(self, stream=None):
        """(counters uint64[4], loss_sums float64[3]) of this rank for the step just applied."""
        cnt = (C.c_uint64 * 4)()
        sums = (C.c_double * 3)()
        self._check(self.f.train_step_local(self._h, _stream_handle(stream), cnt, sums))
        return np.array(cnt, dtype=np.uint64), np.array(sums, dtype=np.float64)

    def train_step_finish(self, counters, loss_sums, allow_no_samples=False):
        cnt = (C.c_uint64 * 4)(*[int(x) for x in counters])
        sums = (C.c_double * 3)(*[float(x) for x in loss_sums])
        st = StepStats()
        rc = self.f.train_step_finish(self._h, cnt, sums, C.byref(st))
        self._check(rc, allow=(_abi.ERR_NO_SAMPLES,) if allow_no_samples else ())
        return st

    def train_step_end(self, stream=None, allow_no_samples=False):
        st = StepStats()
        rc = self.f.train_step_end(self._h, _stream_handle(stream), C.byref(st))
        self._check(rc, allow=(_abi.ERR_NO_SAMPLES,) if allow_no_samples else ())
        return st
