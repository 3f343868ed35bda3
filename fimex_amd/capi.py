"""ctypes binding of the C ABI in include/fimex_amd.h (libfimex_amd.so).

Harness glue for tests/ and bench.py: numpy arrays in and out for the *_host entry points,
raw device pointers (e.g. torch tensors' data_ptr()) for the *_device ones.  No compute
happens here and there is no fallback: a missing library or a missing GPU raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfimex_amd.so")
# the same sources built with -DFIMEX_AMD_TUNING: the only build that reads the FIMEX_AMD_<NAME> experiment switches
TUNING_LIB_PATH = os.path.join(_HERE, "libfimex_amd_tuning.so")

OK, ERROR = 1, -1

# include/fimex_amd.h (values of mifi_interpol_method)
NEAREST_NEIGHBOR, BILINEAR, BICUBIC, COORD_NN, COORD_NN_KD = 0, 1, 2, 3, 4
FORWARD_SUM, FORWARD_MEAN, FORWARD_MEDIAN, FORWARD_MAX, FORWARD_MIN = 5, 6, 7, 8, 9
FORWARD_UNDEF_SUM, FORWARD_UNDEF_MEAN, FORWARD_UNDEF_MEDIAN, FORWARD_UNDEF_MAX, FORWARD_UNDEF_MIN = 10, 11, 12, 13, 14
PROJ_AXIS, LONGITUDE, LATITUDE = 0, 1, 2
BICUBIC_REFERENCE, BICUBIC_FAST = 0, 1


class FimexAmdError(RuntimeError):
    """FIMEX_AMD_ERROR from the library; the message is fimex_amd_last_error()."""


class PlanInfo(ctypes.Structure):
    _fields_ = [("funcType", ctypes.c_int), ("device", ctypes.c_int),
                ("inX", ctypes.c_size_t), ("inY", ctypes.c_size_t), ("outX", ctypes.c_size_t), ("outY", ctypes.c_size_t),
                ("planBytes", ctypes.c_size_t), ("undefinedCells", ctypes.c_size_t), ("borderCells", ctypes.c_size_t),
                ("maxBucket", ctypes.c_size_t), ("mappedSourceCells", ctypes.c_size_t),
                ("stagedCells", ctypes.c_size_t), ("tileW", ctypes.c_size_t), ("tileH", ctypes.c_size_t)]


BATCH_MAX_POSITIONS = 16


class BatchInfo(ctypes.Structure):
    _fields_ = [("d_data", ctypes.c_void_p), ("bytes", ctypes.c_size_t), ("bytesProbed", ctypes.c_size_t), ("bytesHeld", ctypes.c_size_t),
                ("stepBytes", ctypes.c_size_t), ("positions", ctypes.c_int), ("chosen", ctypes.c_int), ("trimmed", ctypes.c_int),
                ("msAtPosition", ctypes.c_float * BATCH_MAX_POSITIONS), ("probeSeconds", ctypes.c_double)]


class Process2d(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("relaxCrit", ctypes.c_float), ("corrEff", ctypes.c_float),
                ("maxLoop", ctypes.c_size_t), ("repeat", ctypes.c_ushort), ("setWeight", ctypes.c_char),
                ("defaultVal", ctypes.c_float)]


PROCESS_FILL2D, PROCESS_CREEPFILL2D, PROCESS_CREEPFILLVAL2D = 1, 2, 3

_F = ctypes.POINTER(ctypes.c_float)
_D = ctypes.POINTER(ctypes.c_double)
_Z = ctypes.c_size_t
_ZP = ctypes.POINTER(ctypes.c_size_t)
_V = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/fimex_amd.h declares
SYMBOLS = {
    "fimex_amd_last_error": (ctypes.c_char_p, []),
    "fimex_amd_abi_version": (ctypes.c_int, []),
    "fimex_amd_device_count": (ctypes.c_int, []),
    "fimex_amd_set_device": (ctypes.c_int, [ctypes.c_int]),
    "fimex_amd_release_caches": (ctypes.c_int, []),
    "fimex_amd_regrid_plan_create": (ctypes.c_int, [ctypes.c_int, _D, _D, _Z, _Z, _Z, _Z, _Z, ctypes.POINTER(_V)]),
    "fimex_amd_regrid_plan_create_device": (ctypes.c_int, [ctypes.c_int, _V, _V, _Z, _Z, _Z, _Z, _Z, _V, ctypes.POINTER(_V)]),
    "fimex_amd_regrid_plan_create_opt": (ctypes.c_int, [ctypes.c_int, _D, _D, _Z, _Z, _Z, _Z, _Z, ctypes.c_int, ctypes.POINTER(_V)]),
    "fimex_amd_regrid_plan_create_device_opt": (ctypes.c_int, [ctypes.c_int, _V, _V, _Z, _Z, _Z, _Z, _Z, ctypes.c_int, _V, ctypes.POINTER(_V)]),
    "fimex_amd_regrid_plan_destroy": (ctypes.c_int, [_V]),
    "fimex_amd_regrid_plan_info": (ctypes.c_int, [_V, ctypes.POINTER(PlanInfo)]),
    "fimex_amd_regrid_apply_host": (ctypes.c_int, [_V, _F, _Z, _F, _Z, _ZP]),
    "fimex_amd_regrid_apply_device": (ctypes.c_int, [_V, _V, _Z, _V, _V]),
    "fimex_amd_regrid_plan_tune_device": (ctypes.c_int, [_V, _V, _Z, _V, _V, ctypes.POINTER(ctypes.c_int)]),
    "fimex_amd_regrid_apply_gather_device": (ctypes.c_int, [_V, _V, _Z, _V, _V]),
    "fimex_amd_regrid_batch_alloc_device": (ctypes.c_int, [_V, _V, _Z, ctypes.c_int, _V, ctypes.POINTER(_V)]),
    "fimex_amd_regrid_source_batch_alloc_device": (ctypes.c_int, [_V, _Z, ctypes.c_int, _V, ctypes.POINTER(_V)]),
    "fimex_amd_batch_get_info": (ctypes.c_int, [_V, ctypes.POINTER(BatchInfo)]),
    "fimex_amd_batch_free": (ctypes.c_int, [_V]),
    "fimex_amd_regrid_slice_host": (ctypes.c_int, [_V, _F, _Z, ctypes.c_float, ctypes.POINTER(Process2d), _Z, _F, ctypes.c_float,
                                                   _V, ctypes.c_int, ctypes.POINTER(Process2d), _Z, _F, _Z, _ZP]),
    "fimex_amd_vector_plan_create": (ctypes.c_int, [_D, _Z, _Z, ctypes.POINTER(_V)]),
    "fimex_amd_vector_plan_destroy": (ctypes.c_int, [_V]),
    "fimex_amd_vector_reproject_values_host": (ctypes.c_int, [_V, _F, _F, _Z]),
    "fimex_amd_vector_reproject_values_device": (ctypes.c_int, [_V, _V, _V, _Z, _V]),
    "fimex_amd_vector_reproject_direction_host": (ctypes.c_int, [_V, _F, _Z]),
    "fimex_amd_vector_reproject_direction_device": (ctypes.c_int, [_V, _V, _Z, _V]),
    "fimex_amd_fill2d_host": (ctypes.c_int, [_Z, _Z, _Z, _F, ctypes.c_float, ctypes.c_float, _Z, _ZP]),
    "fimex_amd_fill2d_device": (ctypes.c_int, [_Z, _Z, _Z, _V, ctypes.c_float, ctypes.c_float, _Z, _ZP, _V]),
    "fimex_amd_creepfill2d_host": (ctypes.c_int, [_Z, _Z, _Z, _F, ctypes.c_ushort, ctypes.c_char, _ZP]),
    "fimex_amd_creepfill2d_device": (ctypes.c_int, [_Z, _Z, _Z, _V, ctypes.c_ushort, ctypes.c_char, _ZP, _V]),
    "fimex_amd_creepfillval2d_host": (ctypes.c_int, [_Z, _Z, _Z, _F, ctypes.c_float, ctypes.c_ushort, ctypes.c_char, _ZP]),
    "fimex_amd_creepfillval2d_device": (ctypes.c_int, [_Z, _Z, _Z, _V, ctypes.c_float, ctypes.c_ushort, ctypes.c_char, _ZP, _V]),
    "fimex_amd_bad2nan_device": (ctypes.c_int, [_V, _Z, ctypes.c_float, _V]),
    "fimex_amd_nan2bad_device": (ctypes.c_int, [_V, _Z, ctypes.c_float, _V]),
    "fimex_amd_points2position_device": (ctypes.c_int, [_V, _Z, _D, ctypes.c_int, ctypes.c_int, _V]),
    "fimex_amd_points2position_host": (ctypes.c_int, [_D, _Z, _D, ctypes.c_int, ctypes.c_int]),
    "fimex_amd_data2interpolation_device": (ctypes.c_int, [_V, ctypes.c_int, _Z, ctypes.c_double, _V, _V]),
    "fimex_amd_interpolation2data_device": (ctypes.c_int, [_V, _Z, ctypes.c_int, ctypes.c_double, _V, _V]),
    "fimex_amd_regrid_apply_typed_device": (ctypes.c_int, [_V, _V, ctypes.c_int, _Z, ctypes.c_double, _V, _V]),
    "fimex_amd_data2interpolation_host": (ctypes.c_int, [_V, ctypes.c_int, _Z, ctypes.c_double, _F]),
    "fimex_amd_interpolation2data_host": (ctypes.c_int, [_F, _Z, ctypes.c_int, ctypes.c_double, _V]),
    "fimex_amd_regrid_slice_typed_host": (ctypes.c_int, [_V, _V, ctypes.c_int, _Z, ctypes.c_double, ctypes.POINTER(Process2d), _Z,
                                                         _V, ctypes.c_int, ctypes.c_double, _V, ctypes.c_int,
                                                         ctypes.POINTER(Process2d), _Z, _V, _Z, _ZP]),
    "fimex_amd_get_values_1d_f_device": (ctypes.c_int, [ctypes.c_int, _V, _V, _V, _Z, ctypes.c_double, ctypes.c_double, ctypes.c_double, _V]),
    "fimex_amd_get_values_1d_f_host": (ctypes.c_int, [ctypes.c_int, _F, _F, _F, _Z, ctypes.c_double, ctypes.c_double, ctypes.c_double]),
    "fimex_amd_get_values_linear_d_device": (ctypes.c_int, [_V, _V, _V, _Z, ctypes.c_double, ctypes.c_double, ctypes.c_double, _V]),
    "fimex_amd_project_values_host": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, _D, _D, _Z]),
    "fimex_amd_project_values_device": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, _V, _V, _Z, _V]),
    "fimex_amd_project_axes_host": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, _D, _D, _Z, _Z, _D, _D]),
    "fimex_amd_project_axes_device": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, _D, _D, _Z, _Z, _V, _V, _V]),
    "fimex_amd_get_vector_reproject_matrix_host": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, _D, _D, ctypes.c_int, ctypes.c_int, _Z, _Z, _D]),
    "fimex_amd_get_vector_reproject_matrix_device": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, _D, _D, ctypes.c_int, ctypes.c_int, _Z, _Z, _V, _V]),
    "fimex_amd_get_vector_reproject_matrix_field_host": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, _D, _D, _Z, _Z, _D]),
    "fimex_amd_get_vector_reproject_matrix_points_host": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, _D, _D, _Z, _D]),
    "fimex_amd_vector_reproject_direction_scaled_host": (ctypes.c_int, [_V, _F, _Z, ctypes.c_double, ctypes.c_double]),
    "fimex_amd_vector_reproject_direction_scaled_device": (ctypes.c_int, [_V, _V, _Z, ctypes.c_double, ctypes.c_double, _V]),
    "fimex_amd_rotate_vector_typed_host": (ctypes.c_int, [_V, _V, ctypes.c_int, ctypes.c_double, _V, ctypes.c_int, ctypes.c_double, _Z, ctypes.c_int,
                                                          ctypes.c_int, ctypes.c_double, _V]),
    "fimex_amd_projection_is_degree": (ctypes.c_int, [ctypes.c_char_p]),
    "fimex_amd_coord_nearest_host": (ctypes.c_int, [_D, _D, _Z, _D, _D, _Z, _Z]),
    "fimex_amd_coord_nearest_device": (ctypes.c_int, [_V, _V, _Z, _V, _V, _Z, _Z, _V]),
    "fimex_amd_coord_kdtree_host": (ctypes.c_int, [ctypes.c_double, _D, _D, _Z, _D, _D, _Z, _Z]),
    "fimex_amd_coord_kdtree_device": (ctypes.c_int, [ctypes.c_double, _V, _V, _Z, _V, _V, _Z, _Z, _V]),
    "fimex_amd_grid_distance_host": (ctypes.c_int, [_D, _D, _Z, _Z, _D]),
    "fimex_amd_scan_sum_device": (ctypes.c_int, [_V, _Z, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_double), _ZP, _V]),
}

_lib = None
_libs = {}


def _open(path):
    if not os.path.exists(path):
        raise FimexAmdError("%s is missing: build it with `python -m fimex_amd.build` (needs hipcc); "
                            "there is no CPU fallback" % path)
    try:
        # share the HIP runtime torch has already mapped (same SONAME) when torch is in the process
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Load libfimex_amd.so; raises when it has not been built (python -m fimex_amd.build)."""
    global _lib
    if _lib is None:
        use_tuning_build(False)
    return _lib


def use_tuning_build(on=True):
    """Switch this module to libfimex_amd_tuning.so (scripts/ sweeps, tests that force a fallback kernel through a
    FIMEX_AMD_<NAME> switch) or back to the product library.  Plans belong to the library that made them: switch before
    creating them.  Returns the previous setting."""
    global _lib
    path = TUNING_LIB_PATH if on else LIB_PATH
    was = _lib is not None and _lib is _libs.get(TUNING_LIB_PATH)
    if path not in _libs:
        _libs[path] = _open(path)
    _lib = _libs[path]
    return was


def _check(rc):
    if rc != OK:
        raise FimexAmdError(load().fimex_amd_last_error().decode() or "fimex_amd call failed")


def release_caches():
    _check(load().fimex_amd_release_caches())


def device_count():
    return load().fimex_amd_device_count()


def set_device(ordinal):
    _check(load().fimex_amd_set_device(ordinal))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _fp(a):
    return a.ctypes.data_as(_F)


def _dp(a):
    return a.ctypes.data_as(_D)


class RegridPlan:
    """fimex_amd_regrid_plan: backward (per output cell) or forward (per input cell) positions."""

    def __init__(self, funcType, pointsOnXAxis, pointsOnYAxis, inX, inY, outX, outY, bicubic=None):
        px, py = _f64(pointsOnXAxis).ravel(), _f64(pointsOnYAxis).ravel()
        if px.size != py.size:
            raise ValueError("position arrays differ in size")
        self._h = _V()
        self.inX, self.inY, self.outX, self.outY = inX, inY, outX, outY
        if bicubic is None:
            _check(load().fimex_amd_regrid_plan_create(funcType, _dp(px), _dp(py), px.size, inX, inY, outX, outY,
                                                       ctypes.byref(self._h)))
        else:  # BICUBIC_REFERENCE / BICUBIC_FAST
            _check(load().fimex_amd_regrid_plan_create_opt(funcType, _dp(px), _dp(py), px.size, inX, inY, outX, outY, bicubic,
                                                           ctypes.byref(self._h)))

    @classmethod
    def from_device(cls, funcType, d_px, d_py, nPoints, inX, inY, outX, outY, stream=0, bicubic=None):
        self = cls.__new__(cls)
        self._h = _V()
        self.inX, self.inY, self.outX, self.outY = inX, inY, outX, outY
        if bicubic is None:
            _check(load().fimex_amd_regrid_plan_create_device(funcType, d_px, d_py, nPoints, inX, inY, outX, outY,
                                                              stream, ctypes.byref(self._h)))
        else:
            _check(load().fimex_amd_regrid_plan_create_device_opt(funcType, d_px, d_py, nPoints, inX, inY, outX, outY, bicubic,
                                                                  stream, ctypes.byref(self._h)))
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().fimex_amd_regrid_plan_destroy(self._h)
            self._h = _V()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may be gone already
            pass

    def info(self):
        i = PlanInfo()
        _check(load().fimex_amd_regrid_plan_info(self._h, ctypes.byref(i)))
        return {k: getattr(i, k) for k, _ in PlanInfo._fields_}

    def apply_host(self, inData):
        """interpolateValues: [nz][inY][inX] float32 host array -> [nz][outY][outX]."""
        a = _f32(inData).ravel()
        n = _Z(0)
        _check(load().fimex_amd_regrid_apply_host(self._h, _fp(a), a.size, None, 0, ctypes.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        _check(load().fimex_amd_regrid_apply_host(self._h, _fp(a), a.size, _fp(out), out.size, ctypes.byref(n)))
        return out.reshape(-1, self.outY, self.outX)

    def apply_device(self, d_in, nz, d_out, stream=0):
        _check(load().fimex_amd_regrid_apply_device(self._h, d_in, nz, d_out, stream))

    def apply_gather_device(self, d_in, nz, d_out, stream=0):
        """The same regrid through the per-lane gather kernels (cross-check of the staged kernels on whole batches)."""
        _check(load().fimex_amd_regrid_apply_gather_device(self._h, d_in, nz, d_out, stream))

    def alloc_source_batch(self, nz, candidates=4, stream=0):
        """Source batch [nz][inY][inX] placed by the library (fimex_amd_regrid_source_batch_alloc_device), zero-filled."""
        return Batch(self, 0, nz, candidates, stream, source=True)

    def alloc_batch(self, d_in, nz, positions=8, stream=0):
        """Output batch [nz][outY][outX] placed by the library (fimex_amd_regrid_batch_alloc_device)."""
        return Batch(self, d_in, nz, positions, stream)

    def tune_device(self, d_in, nz, d_out, stream=0):
        """Times the plan's workgroup shapes on these device buffers and keeps the faster (0: default shape, 1: the other)."""
        chosen = ctypes.c_int(0)
        _check(load().fimex_amd_regrid_plan_tune_device(self._h, d_in, nz, d_out, stream, ctypes.byref(chosen)))
        return chosen.value


class Batch:
    """fimex_amd_batch: device memory of one output batch, placed where the plan's apply launch runs fastest."""

    def __init__(self, plan, d_in, nz, positions=8, stream=0, source=False):
        self._h = _V()
        if source:  # the SOURCE batch [nz][inY][inX]: `positions` whole allocations tried (d_in is not used)
            _check(load().fimex_amd_regrid_source_batch_alloc_device(plan._h, nz, positions, stream, ctypes.byref(self._h)))
        else:
            _check(load().fimex_amd_regrid_batch_alloc_device(plan._h, d_in, nz, positions, stream, ctypes.byref(self._h)))
        i = BatchInfo()
        _check(load().fimex_amd_batch_get_info(self._h, ctypes.byref(i)))
        self.info = {k: getattr(i, k) for k, _ in BatchInfo._fields_ if k != "msAtPosition"}
        self.info["msAtPosition"] = [float(i.msAtPosition[k]) for k in range(i.positions)] if i.positions > 1 else []
        self.data_ptr = i.d_data
        self.nz, self.outY, self.outX = (nz, plan.inY, plan.inX) if source else (nz, plan.outY, plan.outX)

    def as_tensor(self):
        """The batch as a torch tensor [nz][outY][outX] (no copy; keep this object alive as long as the tensor)."""
        import torch

        class _Cai:  # __cuda_array_interface__ of library-owned device memory
            pass
        c = _Cai()
        c.__cuda_array_interface__ = {"shape": (self.nz, self.outY, self.outX), "typestr": "<f4", "data": (int(self.data_ptr), False),
                                      "version": 2, "strides": None}
        t = torch.as_tensor(c, device="cuda")
        t._fimex_amd_batch = self
        return t

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().fimex_amd_batch_free(self._h)
            self._h = _V()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VectorPlan:
    """fimex_amd_vector_plan from the reference's double[4*ox*oy] rotation matrix."""

    def __init__(self, matrix, ox, oy):
        m = _f64(matrix).ravel()
        if m.size != 4 * ox * oy:
            raise ValueError("matrix must hold 4*ox*oy doubles")
        self._h = _V()
        self.ox, self.oy = ox, oy
        _check(load().fimex_amd_vector_plan_create(_dp(m), ox, oy, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().fimex_amd_vector_plan_destroy(self._h)
            self._h = _V()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def reproject_values_host(self, u, v):
        u, v = _f32(u).copy(), _f32(v).copy()
        _check(load().fimex_amd_vector_reproject_values_host(self._h, _fp(u.reshape(-1)), _fp(v.reshape(-1)), u.size))
        return u, v

    def reproject_values_device(self, d_u, d_v, oz, stream=0):
        _check(load().fimex_amd_vector_reproject_values_device(self._h, d_u, d_v, oz, stream))

    def reproject_direction_host(self, angles):
        a = _f32(angles).copy()
        _check(load().fimex_amd_vector_reproject_direction_host(self._h, _fp(a.reshape(-1)), a.size))
        return a

    def reproject_direction_device(self, d_angles, oz, stream=0):
        _check(load().fimex_amd_vector_reproject_direction_device(self._h, d_angles, oz, stream))

    def reproject_direction_scaled_host(self, angles, scale, offset):
        """packed angles: scale * a + offset, rotate, (a - offset) / scale (src/CDMProcessor.cc:621-636)."""
        a = _f32(angles).copy()
        _check(load().fimex_amd_vector_reproject_direction_scaled_host(self._h, _fp(a.reshape(-1)), a.size, scale, offset))
        return a


def fill2d_process(relaxCrit, corrEff, maxLoop):
    return Process2d(PROCESS_FILL2D, relaxCrit, corrEff, maxLoop, 0, b"\0", 0.0)


def creepfill2d_process(repeat, setWeight):
    return Process2d(PROCESS_CREEPFILL2D, 0.0, 0.0, 0, repeat, bytes([setWeight & 0xFF]), 0.0)


def creepfillval2d_process(repeat, setWeight, defaultVal):
    return Process2d(PROCESS_CREEPFILLVAL2D, 0.0, 0.0, 0, repeat, bytes([setWeight & 0xFF]), defaultVal)


def regrid_slice_host(plan, inData, badValue=float("nan"), pre=(), post=(), counterpart=None,
                      badValueCounterpart=float("nan"), vec=None, isXComponent=True):
    """The whole CDMInterpolator::getDataSlice body on the GPU (fimex_amd_regrid_slice_host)."""
    a = _f32(inData).ravel()
    c = _f32(counterpart).ravel() if counterpart is not None else None
    pre_a = (Process2d * len(pre))(*pre) if pre else None
    post_a = (Process2d * len(post))(*post) if post else None
    n = _Z(0)
    args = (plan._h, _fp(a), a.size, badValue, pre_a, len(pre), _fp(c) if c is not None else None, badValueCounterpart,
            vec._h if vec is not None else None, 1 if isXComponent else 0, post_a, len(post))
    _check(load().fimex_amd_regrid_slice_host(*args, None, 0, ctypes.byref(n)))
    out = np.empty(n.value, dtype=np.float32)
    _check(load().fimex_amd_regrid_slice_host(*args, _fp(out), out.size, ctypes.byref(n)))
    return out.reshape(-1, plan.outY, plan.outX)


# CDMDataType codes (include/fimex/CDMDataType.h:35-49) and their numpy element types
CDM_CHAR, CDM_SHORT, CDM_INT, CDM_FLOAT, CDM_DOUBLE, CDM_UCHAR, CDM_USHORT, CDM_UINT, CDM_INT64, CDM_UINT64 = 1, 2, 3, 4, 5, 7, 8, 9, 10, 11
CDM_DTYPES = {CDM_CHAR: np.int8, CDM_SHORT: np.int16, CDM_INT: np.int32, CDM_FLOAT: np.float32, CDM_DOUBLE: np.float64,
              CDM_UCHAR: np.uint8, CDM_USHORT: np.uint16, CDM_UINT: np.uint32, CDM_INT64: np.int64, CDM_UINT64: np.uint64}


def cdm_type_of(dtype):
    for code, dt in CDM_DTYPES.items():
        if np.dtype(dt) == np.dtype(dtype):
            return code
    raise TypeError("no CDMDataType for %s" % dtype)


def regrid_apply_typed_device(plan, d_in, cdmType, nz, badValue, d_out, stream=0):
    _check(load().fimex_amd_regrid_apply_typed_device(plan._h, d_in, cdmType, nz, badValue, d_out, stream))


def data2interpolation_device(d_in, cdmType, n, badValue, d_out, stream=0):
    _check(load().fimex_amd_data2interpolation_device(d_in, cdmType, n, badValue, d_out, stream))


def interpolation2data_device(d_in, n, cdmType, badValue, d_out, stream=0):
    _check(load().fimex_amd_interpolation2data_device(d_in, n, cdmType, badValue, d_out, stream))


def regrid_slice_typed_host(plan, inData, badValue, pre=(), post=(), counterpart=None, badValueCounterpart=float("nan"),
                            vec=None, isXComponent=True):
    """fimex_amd_regrid_slice_typed_host: the slice in the variable's stored type in, the same type out."""
    a = np.ascontiguousarray(inData).ravel()
    t = cdm_type_of(a.dtype)
    c = np.ascontiguousarray(counterpart).ravel() if counterpart is not None else None
    pre_a = (Process2d * len(pre))(*pre) if pre else None
    post_a = (Process2d * len(post))(*post) if post else None
    n = _Z(0)
    args = (plan._h, a.ctypes.data, t, a.size, badValue, pre_a, len(pre), c.ctypes.data if c is not None else None,
            cdm_type_of(c.dtype) if c is not None else 0, badValueCounterpart, vec._h if vec is not None else None,
            1 if isXComponent else 0, post_a, len(post))
    _check(load().fimex_amd_regrid_slice_typed_host(*args, None, 0, ctypes.byref(n)))
    out = np.empty(n.value, dtype=a.dtype)
    _check(load().fimex_amd_regrid_slice_typed_host(*args, out.ctypes.data, out.size, ctypes.byref(n)))
    return out.reshape(-1, plan.outY, plan.outX)


def _slices(field):
    a = _f32(field).copy()
    if a.ndim == 2:
        a = a[None]
    nz, ny, nx = a.shape
    return a, nx, ny, nz


def fill2d_host(field, relaxCrit, corrEff, maxLoop):
    a, nx, ny, nz = _slices(field)
    n = (ctypes.c_size_t * nz)()
    _check(load().fimex_amd_fill2d_host(nx, ny, nz, _fp(a.reshape(-1)), relaxCrit, corrEff, maxLoop, n))
    return a.reshape(np.shape(field)), list(n)


def creepfill2d_host(field, repeat, setWeight):
    a, nx, ny, nz = _slices(field)
    n = (ctypes.c_size_t * nz)()
    _check(load().fimex_amd_creepfill2d_host(nx, ny, nz, _fp(a.reshape(-1)), repeat, bytes([setWeight & 0xFF]), n))
    return a.reshape(np.shape(field)), list(n)


def creepfillval2d_host(field, defaultVal, repeat, setWeight):
    a, nx, ny, nz = _slices(field)
    n = (ctypes.c_size_t * nz)()
    _check(load().fimex_amd_creepfillval2d_host(nx, ny, nz, _fp(a.reshape(-1)), defaultVal, repeat,
                                                bytes([setWeight & 0xFF]), n))
    return a.reshape(np.shape(field)), list(n)


def fill2d_device(d_field, nx, ny, nz, relaxCrit, corrEff, maxLoop, stream=0):
    n = (ctypes.c_size_t * nz)()
    _check(load().fimex_amd_fill2d_device(nx, ny, nz, d_field, relaxCrit, corrEff, maxLoop, n, stream))
    return list(n)


def creepfill2d_device(d_field, nx, ny, nz, repeat, setWeight, stream=0):
    n = (ctypes.c_size_t * nz)()
    _check(load().fimex_amd_creepfill2d_device(nx, ny, nz, d_field, repeat, bytes([setWeight & 0xFF]), n, stream))
    return list(n)


def bad2nan_device(d_data, n, bad, stream=0):
    _check(load().fimex_amd_bad2nan_device(d_data, n, bad, stream))


def nan2bad_device(d_data, n, bad, stream=0):
    _check(load().fimex_amd_nan2bad_device(d_data, n, bad, stream))


BLEND_NEAREST, BLEND_LINEAR, BLEND_LINEAR_WEAK_EXTRAPOL, BLEND_LINEAR_NO_EXTRAPOL, BLEND_LINEAR_CONST_EXTRAPOL, BLEND_LOG, BLEND_LOG_LOG = range(7)


def get_values_1d_host(kind, fieldA, fieldB, a, b, x):
    """mifi_get_values_*_f between two fields; raises where the reference returns MIFI_ERROR."""
    A, B = _f32(fieldA), _f32(fieldB)
    out = np.empty(A.shape, np.float32)
    _check(load().fimex_amd_get_values_1d_f_host(kind, _fp(A.reshape(-1)), _fp(B.reshape(-1)), _fp(out.reshape(-1)), A.size, a, b, x))
    return out


def get_values_1d_device(kind, d_A, d_B, d_out, n, a, b, x, stream=0):
    _check(load().fimex_amd_get_values_1d_f_device(kind, d_A, d_B, d_out, n, a, b, x, stream))


def get_values_linear_d_device(d_A, d_B, d_out, n, a, b, x, stream=0):
    _check(load().fimex_amd_get_values_linear_d_device(d_A, d_B, d_out, n, a, b, x, stream))


def project_values_host(proj_input, proj_output, x, y):
    """mifi_project_values: returns the transformed copies of x and y."""
    xs, ys = _f64(x).copy(), _f64(y).copy()
    fx, fy = xs.reshape(-1), ys.reshape(-1)
    _check(load().fimex_amd_project_values_host(proj_input.encode(), proj_output.encode(), _dp(fx), _dp(fy), fx.size))
    return xs, ys


def project_axes_host(proj_input, proj_output, xAxis, yAxis):
    """mifi_project_axes: two [iy][ix] fields."""
    ax, ay = _f64(xAxis).ravel(), _f64(yAxis).ravel()
    ox, oy = np.empty(ax.size * ay.size), np.empty(ax.size * ay.size)
    _check(load().fimex_amd_project_axes_host(proj_input.encode(), proj_output.encode(), _dp(ax), _dp(ay), ax.size, ay.size, _dp(ox), _dp(oy)))
    return ox.reshape(ay.size, ax.size), oy.reshape(ay.size, ax.size)


def project_axes_device(proj_input, proj_output, xAxis, yAxis, d_outX, d_outY, stream=0):
    ax, ay = _f64(xAxis).ravel(), _f64(yAxis).ravel()
    _check(load().fimex_amd_project_axes_device(proj_input.encode(), proj_output.encode(), _dp(ax), _dp(ay), ax.size, ay.size, d_outX, d_outY, stream))


def get_vector_reproject_matrix_host(proj_input, proj_output, outXAxis, outYAxis, xAxisType=PROJ_AXIS, yAxisType=PROJ_AXIS):
    """mifi_get_vector_reproject_matrix: float64 [oy*ox*4]."""
    ax, ay = _f64(outXAxis).ravel(), _f64(outYAxis).ravel()
    m = np.empty(4 * ax.size * ay.size)
    _check(load().fimex_amd_get_vector_reproject_matrix_host(proj_input.encode(), proj_output.encode(), _dp(ax), _dp(ay), xAxisType, yAxisType,
                                                             ax.size, ay.size, _dp(m)))
    return m


def get_vector_reproject_matrix_device(proj_input, proj_output, outXAxis, outYAxis, xAxisType, yAxisType, d_matrix, stream=0):
    ax, ay = _f64(outXAxis).ravel(), _f64(outYAxis).ravel()
    _check(load().fimex_amd_get_vector_reproject_matrix_device(proj_input.encode(), proj_output.encode(), _dp(ax), _dp(ay), xAxisType, yAxisType,
                                                               ax.size, ay.size, d_matrix, stream))


def get_vector_reproject_matrix_field_host(proj_input, proj_output, inXField, inYField):
    fx, fy = _f64(inXField), _f64(inYField)
    oy, ox = fx.shape
    m = np.empty(4 * fx.size)
    _check(load().fimex_amd_get_vector_reproject_matrix_field_host(proj_input.encode(), proj_output.encode(), _dp(fx.reshape(-1)), _dp(fy.reshape(-1)), ox, oy, _dp(m)))
    return m


def get_vector_reproject_matrix_points_host(proj_input, proj_output, inputIsMetric, outX, outY):
    px, py = _f64(outX).ravel(), _f64(outY).ravel()
    m = np.empty(4 * px.size)
    _check(load().fimex_amd_get_vector_reproject_matrix_points_host(proj_input.encode(), proj_output.encode(), 1 if inputIsMetric else 0, _dp(px), _dp(py), px.size, _dp(m)))
    return m


def rotate_vector_typed_host(vec, xData, xFill, yData, yFill, returnX=True, outFill=None):
    """CDMProcessor's vector rotation on stored types; returns the requested component in its own type."""
    x, y = np.ascontiguousarray(xData), np.ascontiguousarray(yData)
    keep = x if returnX else y
    out = np.empty(keep.shape, keep.dtype)
    _check(load().fimex_amd_rotate_vector_typed_host(vec._h, x.ctypes.data, cdm_type_of(x.dtype), xFill, y.ctypes.data, cdm_type_of(y.dtype), yFill,
                                                     x.size, 1 if returnX else 0, cdm_type_of(keep.dtype),
                                                     (xFill if returnX else yFill) if outFill is None else outFill, out.ctypes.data))
    return out


def projection_is_degree(proj):
    r = load().fimex_amd_projection_is_degree(proj.encode())
    if r < 0:
        raise FimexAmdError(load().fimex_amd_last_error().decode() or "fimex_amd call failed")
    return bool(r)


def coord_nearest_host(lonPoints, latPoints, lonVals, latVals):
    """MIFI_INTERPOL_COORD_NN plan: (x index, y index) of the closest source cell per target point, -1 where none."""
    px, py = _f64(lonPoints).copy().ravel(), _f64(latPoints).copy().ravel()
    lo, la = _f64(lonVals), _f64(latVals)
    orgY, orgX = lo.shape
    _check(load().fimex_amd_coord_nearest_host(_dp(px), _dp(py), px.size, _dp(lo.reshape(-1)), _dp(la.reshape(-1)), orgX, orgY))
    return px, py


def coord_kdtree_host(maxDist, lonPoints, latPoints, lonVals, latVals):
    """MIFI_INTERPOL_COORD_NN_KD plan: closest source cell within maxDist metres, -1000 where none."""
    px, py = _f64(lonPoints).copy().ravel(), _f64(latPoints).copy().ravel()
    lo, la = _f64(lonVals), _f64(latVals)
    orgY, orgX = lo.shape
    _check(load().fimex_amd_coord_kdtree_host(maxDist, _dp(px), _dp(py), px.size, _dp(lo.reshape(-1)), _dp(la.reshape(-1)), orgX, orgY))
    return px, py


def grid_distance_host(lonVals, latVals):
    lo, la = _f64(lonVals), _f64(latVals)
    orgY, orgX = lo.shape
    out = ctypes.c_double(0)
    _check(load().fimex_amd_grid_distance_host(_dp(lo.reshape(-1)), _dp(la.reshape(-1)), orgX, orgY, ctypes.byref(out)))
    return out.value


def scan_sum_device(d_values, n, mode=0, average=0.0, algo=1, stream=0):
    """(sum, nUndefined) of the fills' scan-order double accumulation over n device floats."""
    out = ctypes.c_double(0.0)
    und = ctypes.c_size_t(0)
    _check(load().fimex_amd_scan_sum_device(d_values, n, mode, average, algo, ctypes.byref(out), ctypes.byref(und), stream))
    return out.value, und.value


def points2position_host(points, axis, axis_type=PROJ_AXIS):
    p = _f64(points).copy()
    ax = _f64(axis).ravel()
    flat = p.reshape(-1)
    _check(load().fimex_amd_points2position_host(_dp(flat), flat.size, _dp(ax), ax.size, axis_type))
    return p


def points2position_device(d_points, n, axis, axis_type=PROJ_AXIS, stream=0):
    ax = _f64(axis).ravel()
    _check(load().fimex_amd_points2position_device(d_points, n, _dp(ax), ax.size, axis_type, stream))
