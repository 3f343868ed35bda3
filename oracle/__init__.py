"""ctypes front-end of the CPU oracle (oracle/fimex_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (fimex_amd/) never imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfimex_oracle.so")

# method codes, include/fimex/mifi_constants.h:52-147
NEAREST, BILINEAR, BICUBIC, COORD_NN, COORD_NN_KD = 0, 1, 2, 3, 4
FWD_SUM, FWD_MEAN, FWD_MEDIAN, FWD_MAX, FWD_MIN = 5, 6, 7, 8, 9
FWD_UNDEF_SUM, FWD_UNDEF_MEAN, FWD_UNDEF_MEDIAN, FWD_UNDEF_MAX, FWD_UNDEF_MIN = 10, 11, 12, 13, 14
OK, ERROR = 1, -1
PROJ_AXIS, LONGITUDE, LATITUDE = 0, 1, 2


def build(force=False):
    """Compile the oracle with gcc (no-op when up to date)."""
    src = os.path.join(_HERE, "fimex_oracle.c")
    hdr = os.path.join(_HERE, "fimex_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libfimex_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _declare(_lib)
    return _lib


_F = ctypes.POINTER(ctypes.c_float)
_D = ctypes.POINTER(ctypes.c_double)
_Z = ctypes.c_size_t


def _declare(L):
    for name in ("orc_get_values_f", "orc_get_values_bilinear_f", "orc_get_values_bicubic_f"):
        getattr(L, name).argtypes = [_F, _F, ctypes.c_double, ctypes.c_double, _Z, _Z, _Z]
        getattr(L, name).restype = ctypes.c_int
    L.orc_interpolate_values.argtypes = [ctypes.c_int, _D, _D, _F, _Z, _Z, _Z, _Z, _Z, _F, ctypes.c_int]
    L.orc_interpolate_values.restype = ctypes.c_int
    L.orc_create_reduced_domain.argtypes = [_D, _D, _Z, _Z, _Z] + [ctypes.POINTER(_Z)] * 4
    L.orc_create_reduced_domain.restype = ctypes.c_int
    L.orc_round_and_clamp.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.orc_round_and_clamp.restype = ctypes.c_int
    L.orc_forward_interpolate_values.argtypes = [ctypes.c_int, _D, _D, _F, _Z, _Z, _Z, _Z, _Z, _F]
    L.orc_forward_interpolate_values.restype = ctypes.c_int
    L.orc_vector_reproject_values_by_matrix_f.argtypes = [_D, _F, _F, _Z, _Z, _Z]
    L.orc_vector_reproject_values_by_matrix_f.restype = ctypes.c_int
    L.orc_vector_reproject_direction_by_matrix_f.argtypes = [_D, _F, _Z, _Z, _Z]
    L.orc_vector_reproject_direction_by_matrix_f.restype = ctypes.c_int
    L.orc_fill2d_f.argtypes = [_Z, _Z, _F, ctypes.c_float, ctypes.c_float, _Z, ctypes.POINTER(_Z)]
    L.orc_fill2d_f.restype = ctypes.c_int
    L.orc_creepfill2d_f.argtypes = [_Z, _Z, _F, ctypes.c_ushort, ctypes.c_char, ctypes.POINTER(_Z)]
    L.orc_creepfill2d_f.restype = ctypes.c_int
    L.orc_creepfillval2d_f.argtypes = [_Z, _Z, _F, ctypes.c_float, ctypes.c_ushort, ctypes.c_char, ctypes.POINTER(_Z)]
    L.orc_creepfillval2d_f.restype = ctypes.c_int
    L.orc_points2position.argtypes = [_D, _Z, _D, ctypes.c_int, ctypes.c_int]
    L.orc_points2position.restype = ctypes.c_int
    L.orc_bad2nanf.argtypes = [_F, _F, ctypes.c_float]
    L.orc_bad2nanf.restype = _Z
    L.orc_nanf2bad.argtypes = [_F, _F, ctypes.c_float]
    L.orc_nanf2bad.restype = _Z
    L.orc_get_grid_distance.argtypes = [_D, _D, _Z, _Z]
    L.orc_get_grid_distance.restype = ctypes.c_double
    L.orc_fast_translate_points.argtypes = [_D, _D, _Z, _D, _D, _Z, _Z]
    L.orc_fast_translate_points.restype = ctypes.c_int
    L.orc_flann_translate_points.argtypes = [ctypes.c_double, _D, _D, _Z, _D, _D, _Z, _Z]
    L.orc_flann_translate_points.restype = ctypes.c_int
    L.orc_get_values_1d_f.argtypes = [ctypes.c_int, _F, _F, _F, _Z, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    L.orc_get_values_1d_f.restype = ctypes.c_int
    L.orc_get_values_linear_d.argtypes = [_D, _D, _D, _Z, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    L.orc_get_values_linear_d.restype = ctypes.c_int
    L.orc_data2interpolation_array.argtypes = [ctypes.c_void_p, ctypes.c_int, _Z, ctypes.c_double, _F]
    L.orc_data2interpolation_array.restype = ctypes.c_int
    L.orc_interpolation_array2data.argtypes = [_F, _Z, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    L.orc_interpolation_array2data.restype = ctypes.c_int
    L.orc_vector_matrix_from_deltas.argtypes = [_D] * 6 + [ctypes.c_double, ctypes.c_double, ctypes.c_int, _Z, _D]
    L.orc_vector_matrix_from_deltas.restype = ctypes.c_int


def _f(a):
    return a.ctypes.data_as(_F)


def _d(a):
    return a.ctypes.data_as(_D)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


_POINT = {NEAREST: "orc_get_values_f", BILINEAR: "orc_get_values_bilinear_f", BICUBIC: "orc_get_values_bicubic_f"}


def get_values(method, infield, x, y, ix, iy, iz=1):
    """One point, all z: the reference's mifi_get_values{,_bilinear,_bicubic}_f."""
    a = _c32(infield).ravel()
    assert a.size == ix * iy * iz
    out = np.empty(iz, dtype=np.float32)
    getattr(lib(), _POINT[method])(_f(a), _f(out), float(x), float(y), ix, iy, iz)
    return out


def interpolate_values(method, px, py, infield, inX, inY, outX, outY, nthreads=1):
    """CachedInterpolation::interpolateValues: infield [nz][inY][inX] -> [nz][outY][outX]."""
    a = _c32(infield).ravel()
    nz = a.size // (inX * inY)
    px, py = _c64(px).ravel(), _c64(py).ravel()
    assert px.size == outX * outY and py.size == outX * outY and a.size == nz * inX * inY
    out = np.empty(nz * outX * outY, dtype=np.float32)
    rc = lib().orc_interpolate_values(method, _d(px), _d(py), _f(a), inX, inY, nz, outX, outY, _f(out), nthreads)
    if rc != OK:
        raise ValueError("unknown interpolation function: %d" % method)
    return out.reshape(nz, outY, outX)


def forward_interpolate_values(method, px, py, infield, inX, inY, outX, outY):
    """CachedForwardInterpolation::interpolateValues; px/py per SOURCE cell."""
    a = _c32(infield).ravel()
    nz = a.size // (inX * inY)
    px, py = _c64(px).ravel(), _c64(py).ravel()
    assert px.size == inX * inY and py.size == inX * inY
    out = np.empty(nz * outX * outY, dtype=np.float32)
    rc = lib().orc_forward_interpolate_values(method, _d(px), _d(py), _f(a), inX, inY, nz, outX, outY, _f(out))
    if rc != OK:
        raise ValueError("unknown forward interpolation method: %d" % method)
    return out.reshape(nz, outY, outX)


def create_reduced_domain(px, py, inX, inY):
    px, py = _c64(px).ravel().copy(), _c64(py).ravel().copy()
    v = [_Z(0) for _ in range(4)]
    made = lib().orc_create_reduced_domain(_d(px), _d(py), px.size, inX, inY, *[ctypes.byref(x) for x in v])
    if not made:
        return None
    return dict(px=px, py=py, xMin=v[0].value, yMin=v[1].value, inX=v[2].value, inY=v[3].value)


def round_and_clamp(d, mini, maxi, invalid=-1):
    return lib().orc_round_and_clamp(float(d), mini, maxi, invalid)


def vector_reproject_values(matrix, u, v, ox, oy):
    m = _c64(matrix).ravel()
    u, v = _c32(u).copy().ravel(), _c32(v).copy().ravel()
    oz = u.size // (ox * oy)
    lib().orc_vector_reproject_values_by_matrix_f(_d(m), _f(u), _f(v), ox, oy, oz)
    return u.reshape(oz, oy, ox), v.reshape(oz, oy, ox)


def vector_reproject_direction(matrix, angles, ox, oy):
    m = _c64(matrix).ravel()
    a = _c32(angles).copy().ravel()
    oz = a.size // (ox * oy)
    lib().orc_vector_reproject_direction_by_matrix_f(_d(m), _f(a), ox, oy, oz)
    return a.reshape(oz, oy, ox)


def vector_matrix_from_deltas(out_x, out_y, xdx, ydy, deltaX, deltaY, out_is_latlon):
    out_x, out_y = _c64(out_x).ravel(), _c64(out_y).ravel()
    ax, ay = _c64(xdx[0]).ravel(), _c64(xdx[1]).ravel()
    bx, by = _c64(ydy[0]).ravel(), _c64(ydy[1]).ravel()
    m = np.empty(4 * out_x.size, dtype=np.float64)
    lib().orc_vector_matrix_from_deltas(_d(out_x), _d(out_y), _d(ax), _d(ay), _d(bx), _d(by),
                                        float(deltaX), float(deltaY), int(bool(out_is_latlon)), out_x.size, _d(m))
    return m


def fill2d(field, relaxCrit, corrEff, maxLoop):
    """In-place semantic of mifi_fill2d_f on one [ny][nx] slice; returns (filled copy, nChanged, rc)."""
    a = _c32(field).copy()
    ny, nx = a.shape
    n = _Z(0)
    rc = lib().orc_fill2d_f(nx, ny, _f(a), relaxCrit, corrEff, maxLoop, ctypes.byref(n))
    return a, n.value, rc


def creepfill2d(field, repeat, setWeight):
    a = _c32(field).copy()
    ny, nx = a.shape
    n = _Z(0)
    rc = lib().orc_creepfill2d_f(nx, ny, _f(a), repeat, bytes([setWeight & 0xFF]), ctypes.byref(n))
    return a, n.value, rc


def creepfillval2d(field, defaultVal, repeat, setWeight):
    a = _c32(field).copy()
    ny, nx = a.shape
    n = _Z(0)
    rc = lib().orc_creepfillval2d_f(nx, ny, _f(a), defaultVal, repeat, bytes([setWeight & 0xFF]), ctypes.byref(n))
    return a, n.value, rc


def points2position(points, axis, axis_type=PROJ_AXIS):
    p = _c64(points).copy().ravel()
    ax = _c64(axis).ravel()
    lib().orc_points2position(_d(p), p.size, _d(ax), ax.size, axis_type)
    return p.reshape(np.shape(points))


def bad2nan(a, bad):
    a = _c32(a).copy()
    flat = a.reshape(-1)
    lib().orc_bad2nanf(_f(flat), ctypes.cast(flat.ctypes.data + flat.nbytes, _F), bad)
    return a


# CDMDataType codes (include/fimex/CDMDataType.h:35-49) <-> numpy dtypes
CDM_CHAR, CDM_SHORT, CDM_INT, CDM_FLOAT, CDM_DOUBLE, CDM_UCHAR, CDM_USHORT, CDM_UINT, CDM_INT64, CDM_UINT64 = 1, 2, 3, 4, 5, 7, 8, 9, 10, 11
CDM_DTYPES = {CDM_CHAR: np.int8, CDM_SHORT: np.int16, CDM_INT: np.int32, CDM_FLOAT: np.float32, CDM_DOUBLE: np.float64,
              CDM_UCHAR: np.uint8, CDM_USHORT: np.uint16, CDM_UINT: np.uint32, CDM_INT64: np.int64, CDM_UINT64: np.uint64}


def cdm_type_of(dtype):
    for code, dt in CDM_DTYPES.items():
        if np.dtype(dt) == np.dtype(dtype):
            return code
    raise TypeError("no CDMDataType for %s" % dtype)


def data2interpolation_array(a, bad):
    """data2InterpolationArray: typed array -> float32 with the fill value as NaN."""
    a = np.ascontiguousarray(a)
    out = np.empty(a.shape, np.float32)
    rc = lib().orc_data2interpolation_array(ctypes.c_void_p(a.ctypes.data), cdm_type_of(a.dtype), ctypes.c_size_t(a.size),
                                            ctypes.c_double(bad), _f(out.reshape(-1)))
    assert rc == OK
    return out


def interpolation_array2data(a, newType, bad):
    """interpolationArray2Data: float32 -> array of the CDMDataType newType with NaN as the fill value."""
    a = _c32(a)
    out = np.empty(a.shape, CDM_DTYPES[newType])
    rc = lib().orc_interpolation_array2data(_f(a.reshape(-1)), ctypes.c_size_t(a.size), newType, ctypes.c_double(bad),
                                            ctypes.c_void_p(out.ctypes.data))
    assert rc == OK
    return out


BLEND_NEAREST, BLEND_LINEAR, BLEND_LINEAR_WEAK_EXTRAPOL, BLEND_LINEAR_NO_EXTRAPOL, BLEND_LINEAR_CONST_EXTRAPOL, BLEND_LOG, BLEND_LOG_LOG = range(7)


def get_values_1d(kind, fieldA, fieldB, a, b, x):
    """mifi_get_values_{nearest,linear*,log,log_log}_f: (out, return code)."""
    A, B = _c32(fieldA), _c32(fieldB)
    out = np.full(A.shape, -12345.0, np.float32)
    rc = lib().orc_get_values_1d_f(kind, _f(A.reshape(-1)), _f(B.reshape(-1)), _f(out.reshape(-1)), A.size, a, b, x)
    return out, rc


def get_values_linear_d(fieldA, fieldB, a, b, x):
    A, B = np.ascontiguousarray(fieldA, np.float64), np.ascontiguousarray(fieldB, np.float64)
    out = np.empty(A.shape)
    assert lib().orc_get_values_linear_d(_d(A.reshape(-1)), _d(B.reshape(-1)), _d(out.reshape(-1)), A.size, a, b, x) == OK
    return out


def grid_distance(lonVals, latVals):
    lo, la = np.ascontiguousarray(lonVals, np.float64), np.ascontiguousarray(latVals, np.float64)
    return lib().orc_get_grid_distance(_d(lo.reshape(-1)), _d(la.reshape(-1)), lo.shape[1], lo.shape[0])


def fast_translate_points(lonPoints, latPoints, lonVals, latVals):
    """COORD_NN: (x, y) index of the closest source cell per point, -1 where none."""
    px, py = np.array(lonPoints, np.float64).ravel(), np.array(latPoints, np.float64).ravel()
    lo, la = np.ascontiguousarray(lonVals, np.float64), np.ascontiguousarray(latVals, np.float64)
    assert lib().orc_fast_translate_points(_d(px), _d(py), px.size, _d(lo.reshape(-1)), _d(la.reshape(-1)), lo.shape[1], lo.shape[0]) == OK
    return px, py


def flann_translate_points(maxDist, lonPoints, latPoints, lonVals, latVals):
    """COORD_NN_KD: closest source cell within maxDist metres, -1000 where none."""
    px, py = np.array(lonPoints, np.float64).ravel(), np.array(latPoints, np.float64).ravel()
    lo, la = np.ascontiguousarray(lonVals, np.float64), np.ascontiguousarray(latVals, np.float64)
    assert lib().orc_flann_translate_points(maxDist, _d(px), _d(py), px.size, _d(lo.reshape(-1)), _d(la.reshape(-1)), lo.shape[1], lo.shape[0]) == OK
    return px, py


def nan2bad(a, bad):
    a = _c32(a).copy()
    flat = a.reshape(-1)
    lib().orc_nanf2bad(_f(flat), ctypes.cast(flat.ctypes.data + flat.nbytes, _F), bad)
    return a
