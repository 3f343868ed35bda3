/*
 * fimex_oracle.h -- CPU restatement of the Fimex regridding hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in fimex_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the CPU baseline.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference tree).  The reference's own src/interpolation.c cannot be built
 * in this image (it includes PROJ.4's proj_api.h, which is absent), so this
 * restatement is pinned by the reference's own known-answer tests and data
 * fixtures instead (tests/test_oracle_kats.py, tests/golden/).
 *
 * Arithmetic contract: built with -ffp-contract=off, no -ffast-math; every
 * float/double operation is performed in the same type and order as in the
 * reference so that results are bit-identical to the reference compiled for
 * x86-64 (SSE2, FLT_EVAL_METHOD 0).
 */
#ifndef FIMEX_ORACLE_H_
#define FIMEX_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* method codes, include/fimex/mifi_constants.h:52-147 */
enum {
    ORC_NEAREST = 0, ORC_BILINEAR = 1, ORC_BICUBIC = 2, ORC_COORD_NN = 3, ORC_COORD_NN_KD = 4,
    ORC_FWD_SUM = 5, ORC_FWD_MEAN = 6, ORC_FWD_MEDIAN = 7, ORC_FWD_MAX = 8, ORC_FWD_MIN = 9,
    ORC_FWD_UNDEF_SUM = 10, ORC_FWD_UNDEF_MEAN = 11, ORC_FWD_UNDEF_MEDIAN = 12,
    ORC_FWD_UNDEF_MAX = 13, ORC_FWD_UNDEF_MIN = 14
};
#define ORC_OK 1      /* mifi_constants.h:261 */
#define ORC_ERROR -1  /* mifi_constants.h:259 */
/* axis types, mifi_constants.h:263-268 */
#define ORC_PROJ_AXIS 0
#define ORC_LONGITUDE 1
#define ORC_LATITUDE 2

/* per-point kernels: src/interpolation.c:862-879, 881-957, 959-1028 */
int orc_get_values_f(const float* in, float* out, double x, double y, size_t ix, size_t iy, size_t iz);
int orc_get_values_bilinear_f(const float* in, float* out, double x, double y, size_t ix, size_t iy, size_t iz);
int orc_get_values_bicubic_f(const float* in, float* out, double x, double y, size_t ix, size_t iy, size_t iz);

/* backward apply loop: src/CachedInterpolation.cc:93-147.  nthreads<=1: serial. */
int orc_interpolate_values(int funcType, const double* px, const double* py,
                           const float* in, size_t inX, size_t inY, size_t inZ,
                           size_t outX, size_t outY, float* out, int nthreads);

/* reduced domain: src/CachedInterpolation.cc:149-200.  px/py modified in place.
 * returns 1 if a reduced domain was created (xMin,yMin,newInX,newInY filled). */
int orc_create_reduced_domain(double* px, double* py, size_t n, size_t inX, size_t inY,
                              size_t* xMin, size_t* yMin, size_t* newInX, size_t* newInY);

/* RoundAndClamp: src/Utils.cc:42-58 */
int orc_round_and_clamp(double d, int mini, int maxi, int invalid);

/* forward apply: src/CachedForwardInterpolation.cc:38-131 */
int orc_forward_interpolate_values(int funcType, const double* px, const double* py,
                                   const float* in, size_t inX, size_t inY, size_t inZ,
                                   size_t outX, size_t outY, float* out);

/* vector rotation: src/interpolation.c:790-812, 814-835 */
int orc_vector_reproject_values_by_matrix_f(const double* matrix, float* u, float* v, size_t ox, size_t oy, size_t oz);
int orc_vector_reproject_direction_by_matrix_f(const double* matrix, float* angles, size_t ox, size_t oy, size_t oz);

/* fills: src/interpolation.c:1246-1376, 1378-1537 */
int orc_fill2d_f(size_t nx, size_t ny, float* field, float relaxCrit, float corrEff, size_t maxLoop, size_t* nChanged);
int orc_creepfill2d_f(size_t nx, size_t ny, float* field, unsigned short repeat, char setWeight, size_t* nChanged);
int orc_creepfillval2d_f(size_t nx, size_t ny, float* field, float defaultVal, unsigned short repeat, char setWeight, size_t* nChanged);

/* axis positions: src/interpolation.c:104-217 */
int orc_points2position(double* points, size_t n, const double* axis, int num, int axis_type);

/* fill value <-> NaN: src/interpolation.c:1775-1793 */
size_t orc_bad2nanf(float* begin, float* end, float badVal);
size_t orc_nanf2bad(float* begin, float* end, float badVal);

/* coordinate-based nearest neighbour plans, src/CDMInterpolator.cc:991-1217 (lon / lat in rad, fields [orgY][orgX]) */
double orc_get_grid_distance(const double* lonVals, const double* latVals, size_t orgX, size_t orgY);
int orc_fast_translate_points(double* pointsX, double* pointsY, size_t n, const double* lonVals, const double* latVals, size_t orgX, size_t orgY);
int orc_flann_translate_points(double maxDist, double* pointsX, double* pointsY, size_t n, const double* lonVals, const double* latVals,
                               size_t orgX, size_t orgY);

/* 1-D blends between two fields (time / vertical interpolation), src/interpolation.c:1030-1156 */
enum { ORC_1D_NEAREST = 0, ORC_1D_LINEAR, ORC_1D_LINEAR_WEAK_EXTRAPOL, ORC_1D_LINEAR_NO_EXTRAPOL, ORC_1D_LINEAR_CONST_EXTRAPOL,
       ORC_1D_LOG, ORC_1D_LOG_LOG };
int orc_get_values_1d_f(int kind, const float* infieldA, const float* infieldB, float* outfield, size_t n, double a, double b, double x);
int orc_get_values_linear_d(const double* infieldA, const double* infieldB, double* outfield, size_t n, double a, double b, double x);

/* CDMDataType, include/fimex/CDMDataType.h:35-49 */
enum {
    ORC_CDM_NAT = 0, ORC_CDM_CHAR, ORC_CDM_SHORT, ORC_CDM_INT, ORC_CDM_FLOAT, ORC_CDM_DOUBLE, ORC_CDM_STRING,
    ORC_CDM_UCHAR, ORC_CDM_USHORT, ORC_CDM_UINT, ORC_CDM_INT64, ORC_CDM_UINT64
};
/* bytes of one element of a CDMDataType, 0 for the types this path cannot carry */
size_t orc_cdm_type_size(int type);
/* data2InterpolationArray, src/CDMInterpolator.cc:115-119: Data::asFloat() then mifi_bad2nanf with the fill value */
int orc_data2interpolation_array(const void* in, int type, size_t n, double badValue, float* out);
/* interpolationArray2Data, src/CDMInterpolator.cc:121-124: convertDataType(NaN, 1, 0, newType, badValue, 1, 0) */
int orc_interpolation_array2data(const float* in, size_t n, int newType, double badValue, void* out);

/* rotation matrix from projected points: src/interpolation.c:330-438.
 * The PROJ.4 calls of the reference are replaced by caller-supplied arrays of
 * already projected points (x+dx,y) and (x,y+dy); see oracle/proj_oracle.py. */
int orc_vector_matrix_from_deltas(const double* out_x, const double* out_y,
                                  const double* xdx_x, const double* xdx_y,
                                  const double* ydy_x, const double* ydy_y,
                                  double deltaX, double deltaY, int outIsLatLon,
                                  size_t n, double* matrix);

#ifdef __cplusplus
}
#endif
#endif
