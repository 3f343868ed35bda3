/*
 * fimex_amd.h -- C ABI of the MI355X regridding engine (libfimex_amd.so).
 *
 * This is the drop-in boundary for the hot path of Fimex's CDMInterpolator
 * (L1 + L2 of SURVEY.md): what a maintainer binds from the reference's C++ in
 * place of the per-point mifi_* calls.  Every entry point cites the reference
 * interface it replaces (paths relative to the reference tree).  Plain pointers
 * and sizes only; no C++ or torch types.
 *
 * Conventions (same as the reference's C kernels, include/fimex/mifi_constants.h:259-261):
 *   return FIMEX_AMD_OK (1) or FIMEX_AMD_ERROR (-1); after an error
 *   fimex_amd_last_error() returns a message for the calling thread.
 *   Undefined values are IEEE NaN (mifi_constants.h:249-256).
 *   Fields are C arrays [nz][ny][nx], x fastest (include/fimex/interpolation.h:423-426).
 *
 * *_host entry points take host pointers (the reference's boost::shared_array
 * buffers), copy to the GPU, run the kernels and copy back; they are re-entrant
 * and may be called concurrently from several threads on one plan, as the
 * reference's writers do (src/NetCDF_CDMWriter.cc:749-753).
 * *_device entry points take device pointers plus a hipStream_t (passed as
 * void*; NULL = the default stream) and only enqueue work on that stream.
 *
 * There is no CPU fallback: every compute entry point fails with
 * FIMEX_AMD_ERROR when no gfx950 device is usable.
 */
#ifndef FIMEX_AMD_H_
#define FIMEX_AMD_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIMEX_AMD_OK 1
#define FIMEX_AMD_ERROR -1

/* interpolation methods: values of enum mifi_interpol_method, include/fimex/mifi_constants.h:52-147 */
#define FIMEX_AMD_INTERPOL_NEAREST_NEIGHBOR 0
#define FIMEX_AMD_INTERPOL_BILINEAR 1
#define FIMEX_AMD_INTERPOL_BICUBIC 2
#define FIMEX_AMD_INTERPOL_COORD_NN 3
#define FIMEX_AMD_INTERPOL_COORD_NN_KD 4
#define FIMEX_AMD_INTERPOL_FORWARD_SUM 5
#define FIMEX_AMD_INTERPOL_FORWARD_MEAN 6
#define FIMEX_AMD_INTERPOL_FORWARD_MEDIAN 7
#define FIMEX_AMD_INTERPOL_FORWARD_MAX 8
#define FIMEX_AMD_INTERPOL_FORWARD_MIN 9
#define FIMEX_AMD_INTERPOL_FORWARD_UNDEF_SUM 10
#define FIMEX_AMD_INTERPOL_FORWARD_UNDEF_MEAN 11
#define FIMEX_AMD_INTERPOL_FORWARD_UNDEF_MEDIAN 12
#define FIMEX_AMD_INTERPOL_FORWARD_UNDEF_MAX 13
#define FIMEX_AMD_INTERPOL_FORWARD_UNDEF_MIN 14

/* axis types of mifi_points2position, include/fimex/mifi_constants.h:263-268 */
#define FIMEX_AMD_PROJ_AXIS 0
#define FIMEX_AMD_LONGITUDE 1
#define FIMEX_AMD_LATITUDE 2

/* ------------------------------------------------------------------ library */
/** Message of the last error raised on the calling thread ("" if none). */
const char* fimex_amd_last_error(void);
/** ABI version of this header (major*100 + minor). */
int fimex_amd_abi_version(void);
/** The *_host entry points stream caller buffers through pinned staging that is kept between calls (up to 4 idle sets of
 *  at most 384 MB pinned host memory and 384 MB - 1.1 GB of device memory each, allocated on first use; at most 8 in use at
 *  once, further concurrent callers wait).  This frees the idle ones. */
int fimex_amd_release_caches(void);
/** Number of usable gfx950 devices (0 when there is none; never an error). */
int fimex_amd_device_count(void);
/** Device the calling thread creates plans on (hipSetDevice). */
int fimex_amd_set_device(int ordinal);

/* ------------------------------------------------------------- regrid plans */
/**
 * Opaque, immutable regrid plan resident in HBM.  Replaces the state of
 * CachedInterpolation (include/fimex/CachedInterpolation.h:105-161: two
 * std::vector<double> + a function pointer) and of CachedForwardInterpolation
 * (src/CachedForwardInterpolation.h:37-59: two std::vector<int> + aggregator).
 */
typedef struct fimex_amd_regrid_plan fimex_amd_regrid_plan;

/**
 * Replaces the constructors CachedInterpolation::CachedInterpolation
 * (src/CachedInterpolation.cc:93-116) and
 * CachedForwardInterpolation::CachedForwardInterpolation
 * (src/CachedForwardInterpolation.cc:62-90) -- same arguments minus the
 * dimension names.
 *
 * funcType NEAREST_NEIGHBOR / BILINEAR / BICUBIC / COORD_NN / COORD_NN_KD:
 *   backward plan; pointsOnXAxis/pointsOnYAxis hold, per OUTPUT cell
 *   (nPoints == outX*outY), the fractional position in the input grid.
 * funcType FORWARD_*: forward plan; the arrays hold, per INPUT cell
 *   (nPoints == inX*inY), the fractional position in the output grid.
 * Any other funcType fails ("unknown interpolation function", as
 * CachedInterpolation.cc:114 / CachedForwardInterpolation.cc:88 throw).
 * The arrays are host memory and are not referenced after the call returns.
 */
int fimex_amd_regrid_plan_create(int funcType,
                                 const double* pointsOnXAxis, const double* pointsOnYAxis, size_t nPoints,
                                 size_t inX, size_t inY, size_t outX, size_t outY,
                                 fimex_amd_regrid_plan** plan);
/** Same, with the two position arrays already in device memory (plan build stays on the GPU). */
int fimex_amd_regrid_plan_create_device(int funcType,
                                        const double* d_pointsOnXAxis, const double* d_pointsOnYAxis, size_t nPoints,
                                        size_t inX, size_t inY, size_t outX, size_t outY,
                                        void* stream, fimex_amd_regrid_plan** plan);
/**
 * The same two constructors with the arithmetic of the bicubic kernel chosen per plan (ignored by every other method):
 *   FIMEX_AMD_BICUBIC_REFERENCE  the reference's operations in the reference's order -- products and row sums in double, four
 *                                accumulations into the float result (src/interpolation.c:1005-1019): bit-identical output;
 *   FIMEX_AMD_BICUBIC_FAST       weights rounded to float, float fused multiply-adds: differs from the reference by less than
 *                                1e-5 of the largest magnitude in the 4x4 stencil (typically 1e-7), and the launch is bound by
 *                                memory instead of by FP64 arithmetic.  NaN and out-of-domain behaviour are unchanged.
 */
#define FIMEX_AMD_BICUBIC_REFERENCE 0
#define FIMEX_AMD_BICUBIC_FAST 1
int fimex_amd_regrid_plan_create_opt(int funcType,
                                     const double* pointsOnXAxis, const double* pointsOnYAxis, size_t nPoints,
                                     size_t inX, size_t inY, size_t outX, size_t outY,
                                     int bicubicArithmetic, fimex_amd_regrid_plan** plan);
int fimex_amd_regrid_plan_create_device_opt(int funcType,
                                            const double* d_pointsOnXAxis, const double* d_pointsOnYAxis, size_t nPoints,
                                            size_t inX, size_t inY, size_t outX, size_t outY,
                                            int bicubicArithmetic, void* stream, fimex_amd_regrid_plan** plan);
int fimex_amd_regrid_plan_destroy(fimex_amd_regrid_plan* plan);

typedef struct fimex_amd_plan_info {
    int funcType;
    int device;                /* HIP ordinal the plan lives on */
    size_t inX, inY, outX, outY;
    size_t planBytes;          /* bytes of plan one apply launch reads (B_plan of DESIGN.md) */
    size_t undefinedCells;     /* backward: output cells that are NaN for every input; forward: empty buckets */
    size_t borderCells;        /* backward bilinear: cells on a border branch (interpolation.c:903-948) */
    size_t maxBucket;          /* forward: largest bucket (source cells per target) */
    size_t mappedSourceCells;  /* forward: source cells that fall inside the target grid */
    size_t stagedCells;        /* LDS-staged kernels (backward; forward plans with long buckets): source cells streamed per slice over all tiles (0: gather / lane kernels) */
    size_t tileW, tileH;       /* LDS-staged kernels: output cells per tile */
} fimex_amd_plan_info;
int fimex_amd_regrid_plan_info(const fimex_amd_regrid_plan* plan, fimex_amd_plan_info* info);

/**
 * Replaces CachedInterpolationInterface::interpolateValues
 * (include/fimex/CachedInterpolation.h:67; src/CachedInterpolation.cc:118-147,
 * src/CachedForwardInterpolation.cc:92-131).
 * inData: host [size/(inX*inY)][inY][inX], not modified.
 * outData: host buffer the caller owns (the reference allocates it itself,
 * CachedInterpolation.cc:123); outCapacity in floats must be >= *newSize.
 * *newSize = outX*outY*(size/(inX*inY)), as CachedInterpolation.cc:120-122.
 * Pass outData == NULL to query *newSize only.
 */
int fimex_amd_regrid_apply_host(const fimex_amd_regrid_plan* plan,
                                const float* inData, size_t size,
                                float* outData, size_t outCapacity, size_t* newSize);
/** Device-resident form: d_in [nz][inY][inX] -> d_out [nz][outY][outX], enqueued on stream. */
int fimex_amd_regrid_apply_device(const fimex_amd_regrid_plan* plan,
                                  const float* d_in, size_t nz, float* d_out, void* stream);

/**
 * Optional, no counterpart in the reference: a bilinear plan (and a bicubic one with FIMEX_AMD_BICUBIC_FAST) holds its LDS-staged form in two workgroup shapes with identical
 * results, and which of them is faster depends on the device at hand and on the batch length.  This call regrids
 * the caller's nz slices a few times with each shape (d_out ends up holding the regridded slices), keeps the faster one for every
 * later apply of this plan and reports it in *chosenShape (0: the default shape, 1: the other; NULL allowed).  Plans without a
 * second shape and batches too short for the staged kernels return 0 at once.  Synchronises the stream; not to be called while
 * other threads apply the same plan.
 */
int fimex_amd_regrid_plan_tune_device(fimex_amd_regrid_plan* plan, const float* d_in, size_t nz, float* d_out, void* stream,
                                      int* chosenShape);

/**
 * Cross-check, no counterpart in the reference: the same regrid through the per-lane gather kernels (one lane per output cell
 * reads its stencil straight from memory) whatever LDS-staged form the plan holds.  Backward plans only.  The result equals
 * fimex_amd_regrid_apply_device's bit for bit (FIMEX_AMD_BICUBIC_FAST plans: the reference's arithmetic, i.e. within the stated
 * 1e-5); bench.py and the tests use it to check EVERY slice of a long batch on the device.
 */
int fimex_amd_regrid_apply_gather_device(const fimex_amd_regrid_plan* plan, const float* d_in, size_t nz, float* d_out, void* stream);

/* ------------------------------ source and output batches placed by the library */
/**
 * The reference allocates the result of every interpolateValues call itself (src/CachedInterpolation.cc:123) and receives its input
 * from the reader.  A caller that keeps its slices in device memory allocates the source batch [nz][inY][inX] and the output batch
 * [nz][outY][outX] once -- and which allocation they lie in moves the apply launch by several per cent (the source's
 * by 4-5 %, the output's by 1-3 %).  These two calls allocate the batches for the caller: `candidates` whole allocations are made and
 * held at once, the plan's apply launch is timed with each in its role (median of three launches), the fastest is kept and the others
 * are freed before the call returns.
 *   fimex_amd_regrid_source_batch_alloc_device  the buffer the caller's reader fills: every candidate is zero-filled and regridded
 *                                               into a scratch output; the kept one holds zeros.  Allocate this one first.
 *   fimex_amd_regrid_batch_alloc_device         the output batch: the caller's source batch d_in is regridded into every candidate
 *                                               (the batch holds the regridded slices of d_in afterwards only by accident).
 * candidates == 1: a plain allocation, nothing is timed (d_in may be NULL).  Fewer candidates are tried when device memory is short.
 * Both synchronise the stream.
 */
#define FIMEX_AMD_BATCH_MAX_POSITIONS 16
typedef struct fimex_amd_batch fimex_amd_batch;
typedef struct fimex_amd_batch_info {
    void* d_data;        /* the batch in device memory (plain hipMalloc memory) */
    size_t bytes;        /* of the batch */
    size_t bytesProbed;  /* device memory allocated while the candidates were tried */
    size_t bytesHeld;    /* device memory behind this batch after the call (== bytes) */
    size_t stepBytes;    /* 0 (candidates are separate allocations) */
    int positions;       /* candidates tried */
    int chosen;          /* the one that was kept */
    int trimmed;         /* 1: the other candidates were freed */
    float msAtPosition[FIMEX_AMD_BATCH_MAX_POSITIONS];  /* median time of the apply launch with each candidate */
    double probeSeconds; /* wall time of the whole call */
} fimex_amd_batch_info;
int fimex_amd_regrid_source_batch_alloc_device(const fimex_amd_regrid_plan* plan, size_t nz, int candidates, void* stream,
                                               fimex_amd_batch** batch);
int fimex_amd_regrid_batch_alloc_device(const fimex_amd_regrid_plan* plan, const float* d_in, size_t nz, int candidates, void* stream,
                                        fimex_amd_batch** batch);
int fimex_amd_batch_get_info(const fimex_amd_batch* batch, fimex_amd_batch_info* info);
int fimex_amd_batch_free(fimex_amd_batch* batch);

/* ---------------------------------------------------------- vector rotation */
/**
 * Opaque rotation plan.  Replaces CachedVectorReprojection
 * (include/fimex/CachedVectorReprojection.h:33-63): built from the same
 * double[4*ox*oy] matrix (cos, sin, -sin, phi per cell, src/interpolation.c:429-432).
 * The matrix is copied; a compact (cos,sin) / phi form is kept in HBM.
 */
typedef struct fimex_amd_vector_plan fimex_amd_vector_plan;
int fimex_amd_vector_plan_create(const double* matrix, size_t ox, size_t oy, fimex_amd_vector_plan** plan);
int fimex_amd_vector_plan_destroy(fimex_amd_vector_plan* plan);
/** Replaces CachedVectorReprojection::reprojectValues (src/CachedVectorReprojection.cc:35-44) ->
 *  mifi_vector_reproject_values_by_matrix_f (src/interpolation.c:790-812); u, v rotated in place. */
int fimex_amd_vector_reproject_values_host(const fimex_amd_vector_plan* plan, float* u, float* v, size_t size);
int fimex_amd_vector_reproject_values_device(const fimex_amd_vector_plan* plan, float* d_u, float* d_v, size_t oz, void* stream);
/** Replaces CachedVectorReprojection::reprojectDirectionValues (src/CachedVectorReprojection.cc:46-55) ->
 *  mifi_vector_reproject_direction_by_matrix_f (src/interpolation.c:814-835); degrees, in place. */
int fimex_amd_vector_reproject_direction_host(const fimex_amd_vector_plan* plan, float* angles, size_t size);
int fimex_amd_vector_reproject_direction_device(const fimex_amd_vector_plan* plan, float* d_angles, size_t oz, void* stream);

/* ------------------------------------------------------ 2-D fill processes */
/*
 * Batch forms of the InterpolatorProcess2d implementations
 * (include/fimex/CDMInterpolator.h:49-88) as driven by processArray_
 * (src/CDMInterpolator.cc:136-159): every [ny][nx] slice of field[nz][ny][nx] is
 * filled in place, independently.  nChanged: NULL or size_t[nz] receiving the
 * per-slice count of undefined cells the reference returns through *nChanged.
 */
/** mifi_fill2d_f, include/fimex/interpolation.h:483, src/interpolation.c:1246-1376 */
int fimex_amd_fill2d_host(size_t nx, size_t ny, size_t nz, float* field,
                          float relaxCrit, float corrEff, size_t maxLoop, size_t* nChanged);
int fimex_amd_fill2d_device(size_t nx, size_t ny, size_t nz, float* d_field,
                            float relaxCrit, float corrEff, size_t maxLoop, size_t* nChanged, void* stream);
/** mifi_creepfill2d_f, include/fimex/interpolation.h:505, src/interpolation.c:1495-1519 */
int fimex_amd_creepfill2d_host(size_t nx, size_t ny, size_t nz, float* field,
                               unsigned short repeat, char setWeight, size_t* nChanged);
int fimex_amd_creepfill2d_device(size_t nx, size_t ny, size_t nz, float* d_field,
                                 unsigned short repeat, char setWeight, size_t* nChanged, void* stream);
/** mifi_creepfillval2d_f, include/fimex/interpolation.h:528, src/interpolation.c:1521-1537 */
int fimex_amd_creepfillval2d_host(size_t nx, size_t ny, size_t nz, float* field, float defaultVal,
                                  unsigned short repeat, char setWeight, size_t* nChanged);
int fimex_amd_creepfillval2d_device(size_t nx, size_t ny, size_t nz, float* d_field, float defaultVal,
                                    unsigned short repeat, char setWeight, size_t* nChanged, void* stream);

/* ------------------------------------------- the whole per-slice sequence */
/** One registered 2-D process: the parameters of InterpolatorFill2d / InterpolatorCreepFill2d /
 *  InterpolatorCreepFillVal2d (include/fimex/CDMInterpolator.h:55-88). */
#define FIMEX_AMD_PROCESS_FILL2D 1
#define FIMEX_AMD_PROCESS_CREEPFILL2D 2
#define FIMEX_AMD_PROCESS_CREEPFILLVAL2D 3
typedef struct fimex_amd_process2d {
    int kind;               /* FIMEX_AMD_PROCESS_* */
    float relaxCrit;        /* fill2d */
    float corrEff;          /* fill2d */
    size_t maxLoop;         /* fill2d */
    unsigned short repeat;  /* creepfill */
    char setWeight;         /* creepfill */
    float defaultVal;       /* creepfillval2d */
} fimex_amd_process2d;

/**
 * Replaces the body of CDMInterpolator::getDataSlice between reading the input and converting the output
 * (src/CDMInterpolator.cc:255-285) with the data staying in HBM between the steps:
 *   fill value -> NaN (:115-119), pre-processes per z slice (:256, :136-159), interpolateValues (:259),
 *   for an x/y vector component: the same on the counterpart, then reprojectValues (:261-283),
 *   post-processes (:284), NaN -> fill value (:285, without the type conversion, which stays with Data).
 * inData / counterpart: host [size/(inX*inY)][inY][inX]; badValue*: the variables' fill values (NaN = none).
 * counterpart == NULL or vec == NULL: scalar variable.  isXComponent != 0: inData is the x component (u) and
 * counterpart the y component (v); otherwise the other way round; outData receives the requested component.
 * pre / post: the registered processes in order (may be NULL when the count is 0).
 * outData == NULL queries *newSize only.
 */
int fimex_amd_regrid_slice_host(const fimex_amd_regrid_plan* plan, const float* inData, size_t size, float badValue,
                                const fimex_amd_process2d* pre, size_t nPre,
                                const float* counterpart, float badValueCounterpart,
                                const fimex_amd_vector_plan* vec, int isXComponent,
                                const fimex_amd_process2d* post, size_t nPost,
                                float* outData, size_t outCapacity, size_t* newSize);

/* ------------------------------------------------- edges of the path (a13) */
/** mifi_bad2nanf / mifi_nanf2bad, src/interpolation.c:1775-1793, on n device floats in place. */
int fimex_amd_bad2nan_device(float* d_data, size_t n, float badVal, void* stream);
int fimex_amd_nan2bad_device(float* d_data, size_t n, float badVal, void* stream);

/** CDMProcessor's direction rotation of packed angles (src/CDMProcessor.cc:621-636): scale * a + offset, rotate
 *  (a9), (a - offset) / scale, in one pass; angles [oz][oy][ox] in place. */
int fimex_amd_vector_reproject_direction_scaled_host(const fimex_amd_vector_plan* plan, float* angles, size_t size, double scale, double offset);
int fimex_amd_vector_reproject_direction_scaled_device(const fimex_amd_vector_plan* plan, float* d_angles, size_t oz, double scale,
                                                       double offset, void* stream);

/* ------------------------------------------------ typed slices */
/** CDMDataType, include/fimex/CDMDataType.h:35-49 (same values). */
typedef enum fimex_amd_datatype {
    FIMEX_AMD_CDM_NAT = 0, FIMEX_AMD_CDM_CHAR, FIMEX_AMD_CDM_SHORT, FIMEX_AMD_CDM_INT, FIMEX_AMD_CDM_FLOAT,
    FIMEX_AMD_CDM_DOUBLE, FIMEX_AMD_CDM_STRING, FIMEX_AMD_CDM_UCHAR, FIMEX_AMD_CDM_USHORT, FIMEX_AMD_CDM_UINT,
    FIMEX_AMD_CDM_INT64, FIMEX_AMD_CDM_UINT64
} fimex_amd_datatype;
/** data2InterpolationArray, src/CDMInterpolator.cc:115-119: n device elements of cdmType -> float (Data::asFloat())
 *  with the variable's fill value as NaN (mifi_bad2nanf), in one pass. */
int fimex_amd_data2interpolation_device(const void* d_in, int cdmType, size_t n, double badValue, float* d_out, void* stream);
/** interpolationArray2Data, src/CDMInterpolator.cc:121-124: float -> cdmType as
 *  DataImpl<float>::convertDataType(MIFI_UNDEFINED_F, 1, 0, cdmType, badValue, 1, 0) does (NaN -> fill value,
 *  integers rounded through MetNoFimex::round), in one pass. */
int fimex_amd_interpolation2data_device(const float* d_in, size_t n, int cdmType, double badValue, void* d_out, void* stream);
/** interpolateValues on device-resident slices of a variable's stored type: d_in [nz][inY][inX] elements of cdmType ->
 *  d_out [nz][outY][outX] elements of the same type, i.e. data2InterpolationArray, the regrid and interpolationArray2Data
 *  of src/CDMInterpolator.cc:251-285 without pre/post-processes.  For backward plans on 1-, 2-byte and 32-bit integer
 *  types this is ONE kernel that reads and writes the stored type (no float copy of the slices exists); other
 *  combinations run the three passes on temporaries. */
int fimex_amd_regrid_apply_typed_device(const fimex_amd_regrid_plan* plan, const void* d_in, int cdmType, size_t nz,
                                        double badValue, void* d_out, void* stream);
/** The same two conversions on host buffers (copied to the GPU and back), for callers that run their own 2-D processes
 *  on the float array in between. */
int fimex_amd_data2interpolation_host(const void* in, int cdmType, size_t n, double badValue, float* out);
int fimex_amd_interpolation2data_host(const float* in, size_t n, int cdmType, double badValue, void* out);
/** fimex_amd_regrid_slice_host on the variable's stored type: everything CDMInterpolator::getDataSlice
 *  (src/CDMInterpolator.cc:251-285) does with a slice, including both conversions; only `size` elements of dataType
 *  cross PCIe in, *newSize elements of dataType come back (half the bytes for packed shorts).  size and outCapacity
 *  count elements.  The counterpart of a vector variable may be stored in another type. */
int fimex_amd_regrid_slice_typed_host(const fimex_amd_regrid_plan* plan, const void* inData, int dataType, size_t size, double badValue,
                                      const fimex_amd_process2d* pre, size_t nPre,
                                      const void* counterpart, int counterpartType, double badValueCounterpart,
                                      const fimex_amd_vector_plan* vec, int isXComponent,
                                      const fimex_amd_process2d* post, size_t nPost,
                                      void* outData, size_t outCapacity, size_t* newSize);

/** CDMProcessor::getDataSlice's vector rotation on stored types (src/CDMProcessor.cc:590-618): both components become
 *  float/NaN (data2InterpolationArray), are rotated in place (a8), and the requested one (returnX != 0: x) comes back in
 *  outType with outFill as interpolationArray2Data does.  size elements per component, [size/(ox*oy)][oy][ox]. */
int fimex_amd_rotate_vector_typed_host(const fimex_amd_vector_plan* plan, const void* xData, int xType, double xFill,
                                       const void* yData, int yType, double yFill, size_t size, int returnX,
                                       int outType, double outFill, void* outData);

/* ------------------------------------------------------- plan build helpers */
/** mifi_points2position, include/fimex/interpolation.h:415, src/interpolation.c:148-217:
 *  n device doubles (radians or metres) -> fractional axis indices, in place. axis: host, num entries. */
int fimex_amd_points2position_device(double* d_points, size_t n, const double* axis, int num, int axis_type, void* stream);
/** Same on n host doubles (copied to the GPU and back). */
int fimex_amd_points2position_host(double* points, size_t n, const double* axis, int num, int axis_type);

/* ------------------------------------------ 1-D blends between two fields (8f n4) */
/** mifi_get_values_{nearest, linear, linear_weak_extrapol, linear_no_extrapol, linear_const_extrapol, log, log_log}_f
 *  (include/fimex/interpolation.h, src/interpolation.c:1030-1156): outfield = blend of infieldA (at coordinate a) and
 *  infieldB (at b) at coordinate x, n values, as the time and vertical interpolators call them.  out may alias A or B.
 *  Returns FIMEX_AMD_ERROR where the reference returns MIFI_ERROR (non-positive a, b or x for the log blends). */
typedef enum fimex_amd_blend1d {
    FIMEX_AMD_1D_NEAREST = 0, FIMEX_AMD_1D_LINEAR, FIMEX_AMD_1D_LINEAR_WEAK_EXTRAPOL, FIMEX_AMD_1D_LINEAR_NO_EXTRAPOL,
    FIMEX_AMD_1D_LINEAR_CONST_EXTRAPOL, FIMEX_AMD_1D_LOG, FIMEX_AMD_1D_LOG_LOG
} fimex_amd_blend1d;
int fimex_amd_get_values_1d_f_device(int kind, const float* d_infieldA, const float* d_infieldB, float* d_outfield, size_t n,
                                     double a, double b, double x, void* stream);
int fimex_amd_get_values_1d_f_host(int kind, const float* infieldA, const float* infieldB, float* outfield, size_t n,
                                   double a, double b, double x);
/** mifi_get_values_linear_d, src/interpolation.c:1065-1083. */
int fimex_amd_get_values_linear_d_device(const double* d_infieldA, const double* d_infieldB, double* d_outfield, size_t n,
                                         double a, double b, double x, void* stream);

/* ----------------------------------- plan building across projections (8f n2) */
/* The reference calls PROJ.4 (pj_init_plus / pj_transform) here; this library carries its own projections:
 * latlong/longlat, stere, lcc, merc, tmerc, etmerc, utm, laea, aea, geos, omerc, sinu, cea, ortho, aeqd, nsper, ob_tran with o_proj=longlat (radians at this boundary for
 * geographic and rotated coordinates, as PROJ.4's legacy API), on a sphere (+R, +a +e=0, +ellps=sphere) or an ellipsoid
 * (+ellps, +datum=WGS84|NAD83, +a with +b/+rf/+f/+e/+es).  Three- and seven-parameter datum shifts (+towgs84, +datum=WGS84|NAD83|GGRS87|potsdam)
 * are applied as pj_transform does; strings that need a grid shift, +units, +pm, +axis or another projection make the call
 * fail with a message. */
/** mifi_project_values, include/fimex/interpolation.h / src/interpolation.c:1158-1197: n points in place. */
int fimex_amd_project_values_host(const char* proj_input, const char* proj_output, double* in_out_x_vals, double* in_out_y_vals, size_t num);
int fimex_amd_project_values_device(const char* proj_input, const char* proj_output, double* d_x_vals, double* d_y_vals, size_t num, void* stream);
/** mifi_project_axes, src/interpolation.c:1199-1244: the [iy][ix] mesh of two host axes, transformed; the device form
 *  leaves the two fields in device memory (feed fimex_amd_points2position_device / fimex_amd_regrid_plan_create_device). */
int fimex_amd_project_axes_host(const char* proj_input, const char* proj_output, const double* in_x_axis, const double* in_y_axis,
                                size_t ix, size_t iy, double* out_xproj_axis, double* out_yproj_axis);
int fimex_amd_project_axes_device(const char* proj_input, const char* proj_output, const double* in_x_axis, const double* in_y_axis,
                                  size_t ix, size_t iy, double* d_out_xproj_axis, double* d_out_yproj_axis, void* stream);
/** mifi_get_vector_reproject_matrix, src/interpolation.c:719-788: matrix[4*ox*oy] = (cos, sin, -sin, phi) of the local
 *  rotation from proj_input to proj_output on the mesh of the output axes (degrees for FIMEX_AMD_LONGITUDE /
 *  FIMEX_AMD_LATITUDE axis types, as the reference).  Host or device destination. */
int fimex_amd_get_vector_reproject_matrix_host(const char* proj_input, const char* proj_output, const double* out_x_axis,
                                               const double* out_y_axis, int out_x_axis_type, int out_y_axis_type,
                                               size_t ox, size_t oy, double* matrix);
int fimex_amd_get_vector_reproject_matrix_device(const char* proj_input, const char* proj_output, const double* out_x_axis,
                                                 const double* out_y_axis, int out_x_axis_type, int out_y_axis_type,
                                                 size_t ox, size_t oy, double* d_matrix, void* stream);
/** mifi_get_vector_reproject_matrix_field, src/interpolation.c:657-717: the [oy][ox] mesh is given as two fields in the
 *  INPUT projection (CDMProcessor's rotation to lat/lon, src/CDMProcessor.cc:123-136). */
int fimex_amd_get_vector_reproject_matrix_field_host(const char* proj_input, const char* proj_output, const double* in_x_field,
                                                     const double* in_y_field, size_t ox, size_t oy, double* matrix);
/** mifi_get_vector_reproject_matrix_points, src/interpolation.c:607-655: on points in the OUTPUT projection (m or rad),
 *  finite difference of 100 m (inputIsMetric) or 1e-5 rad. */
int fimex_amd_get_vector_reproject_matrix_points_host(const char* proj_input, const char* proj_output, int inputIsMetric,
                                                      const double* out_x_points, const double* out_y_points, size_t on, double* matrix);
/** Projection::isDegree (src/coordSys/Projection.cc): 1 for geographic and rotated lat/lon strings, 0 otherwise, -1 on error. */
int fimex_amd_projection_is_degree(const char* proj);

/* ------------------------------- coordinate-based nearest neighbour plans (8f n3) */
/** fastTranslatePointsToClosestInputCell with getGridDistance, src/CDMInterpolator.cc:1069-1217 (MIFI_INTERPOL_COORD_NN):
 *  pointsOnXAxis / pointsOnYAxis hold longitude / latitude (rad) of every target cell on entry and the x / y index of the
 *  closest source cell (as doubles, -1 when none lies within the grid's region of influence) on return; lonVals / latVals:
 *  the source grid's coordinates in rad, [orgYDimSize][orgXDimSize], NaN = undefined.  The result is the plan of
 *  fimex_amd_regrid_plan_create(MIFI_INTERPOL_COORD_NN, ...). */
int fimex_amd_coord_nearest_host(double* pointsOnXAxis, double* pointsOnYAxis, size_t nPoints,
                                 const double* lonVals, const double* latVals, size_t orgXDimSize, size_t orgYDimSize);
int fimex_amd_coord_nearest_device(double* d_pointsOnXAxis, double* d_pointsOnYAxis, size_t nPoints,
                                   const double* d_lonVals, const double* d_latVals, size_t orgXDimSize, size_t orgYDimSize, void* stream);
/** flannTranslatePointsToClosestInputCell, src/CDMInterpolator.cc:991-1067 (MIFI_INTERPOL_COORD_NN_KD): closest source
 *  cell within maxDist metres (chord on a sphere of MIFI_EARTH_RADIUS_M), -1000 when none. */
int fimex_amd_coord_kdtree_host(double maxDist, double* pointsOnXAxis, double* pointsOnYAxis, size_t nPoints,
                                const double* lonVals, const double* latVals, size_t orgXDimSize, size_t orgYDimSize);
int fimex_amd_coord_kdtree_device(double maxDist, double* d_pointsOnXAxis, double* d_pointsOnYAxis, size_t nPoints,
                                  const double* d_lonVals, const double* d_latVals, size_t orgXDimSize, size_t orgYDimSize, void* stream);
/** getGridDistance, src/CDMInterpolator.cc:1069-1141: the region of influence (rad) COORD_NN derives from the source grid. */
int fimex_amd_grid_distance_host(const double* lonVals, const double* latVals, size_t orgXDimSize, size_t orgYDimSize, double* maxGridDistance);

/* ------------------------------------------------------------- diagnostics */
/** The scan-order double sums the fills start with (src/interpolation.c:1256-1264 sum of the defined values, mode 0;
 *  :1288-1299 sum of |v - average|, mode 1; mode 2 only counts), on n device floats: exactly the value the reference's
 *  sequential loop accumulates.  algo 0 walks the additions one by one, algo 1 evaluates them binade-parallel (the
 *  fills' default); both must agree bit for bit -- this entry exists so that tests can check that directly. */
int fimex_amd_scan_sum_device(const float* d_values, size_t n, int mode, double average, int algo,
                              double* sum, size_t* nUndefined, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FIMEX_AMD_H_ */
