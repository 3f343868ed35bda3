// Regrid and rotation plans as they live in HBM.
//
// The reference keeps, per output cell, two doubles (fractional source position,
// include/fimex/CachedInterpolation.h:108-109) and re-derives floor / fraction /
// border class for every cell of every call.  Here that derivation runs once, on the
// GPU, when the plan is created; the apply kernels read a compact structure-of-arrays
// form that reproduces the reference's arithmetic bit for bit:
//
//   nearest   pos[n]            u32  source offset inside a slice, kInvalidPos = undefined
//   bilinear  pos[n], xf[n], yf[n]   u32 + 2 x f32.  xf/yf are the fractions already rounded
//             to float exactly as src/interpolation.c:885,888 does.  A set sign bit in xf
//             (yf) marks "nearest neighbour in x (y)", i.e. the border branches of
//             interpolation.c:903-948; pos then already points at the rounded cell.
//   bicubic   pos[n], xfd[n], yfd[n] u32 + 2 x f64 (double fractions, interpolation.c:971,973);
//             pos is the (x0-1, y0-1) corner of the 4x4 stencil.
//   forward   offsets[nOut+1], src[nMapped]   CSR inverse of the reference's per-source-cell
//             target index (src/CachedForwardInterpolation.cc:72-73), buckets in source scan
//             order so that sums add in the reference's push_back order.
#pragma once

#include "common.hpp"

#include <functional>
#include <mutex>

namespace fimex_amd {

constexpr uint32_t kInvalidPos = 0xFFFFFFFFu;
// source slices are addressed with 32-bit cell offsets (and 32-bit byte offsets in the kernels)
constexpr size_t kMaxSliceCells = (size_t(1) << 30) - 1;

enum class PlanKind { Nearest, Bilinear, Bicubic, Forward };

enum class Aggregate : int { Sum = 0, Mean = 1, Median = 2, Max = 3, Min = 4 };

}  // namespace fimex_amd

namespace fimex_amd {
// LDS-staged form of a bilinear / bicubic plan (staged.hip): per tile the source row segments to stream into
// LDS, per output cell the 16-bit LDS offsets of its stencil rows; the fractions are those of the gather plan.
struct StagedPlan {
    bool valid = false;
    uint32_t tileW = 0, tileH = 0, per = 0, kmax = 0, tilesX = 0, nTiles = 0;
    size_t stagedCells = 0;  // source cells streamed per slice (16-byte granules, all tiles)
    DeviceArray<uint32_t> tileRows;
    DeviceArray<uint2> tileHdr;
    DeviceArray<uint32_t> ldsA, ldsB;  // LDS offsets of stencil rows 0|1 (and 2|3 for bicubic), 16 bits each
};

// Second LDS-staged form (staged2.hip): tiles of one height and varying width (narrower where the source footprint of an
// output cell is larger, so that every tile fits the same LDS budget), workgroups of 256-1024 threads, and per tile the
// list of 16-byte source chunks itself instead of row segments.
struct StagedTile {
    uint32_t x0, y0, w;     // output columns [x0, x0 + w) of rows [y0, y0 + tileH)
    uint32_t nChunks;       // 16-byte chunks streamed per slice
    uint32_t chunkBase;     // first entry in chunkOff
    uint32_t rsv[3];
};
struct Staged2Plan {
    bool valid = false;
    uint32_t nt = 0, per = 0, kmax = 0, tileH = 0, tileWMax = 0, nTiles = 0, gridX = 0, ldsBytes = 0, depth = 2;
    size_t stagedCells = 0, totalChunks = 0;
    DeviceArray<StagedTile> tiles;
    DeviceArray<uint32_t> order;     // workgroup -> tile, ~0u = none (tile rows dealt to the XCDs in stripes)
    DeviceArray<uint32_t> chunkOff;  // source cell offset of every chunk inside a slice
    DeviceArray<uint32_t> ldsA, ldsB;
};

// forward_tiled.hip: the LDS-staged form of a forward plan whose buckets are long (tiles of 64 targets, one wave each)
struct ForwardTile {
    uint32_t chunkBase;  // first entry of the tile in chunkOff
    uint32_t nChunks;    // ~0u: the tile's buckets are spread too far for LDS, it reads from memory
    uint32_t stepBase;   // first entry of the tile in steps
    uint32_t maxLen;     // longest bucket of the tile; 0: every target of the tile is empty
};
struct ForwardTiles {
    bool valid = false;
    uint32_t tw = 0, th = 0, tilesX = 0, nTiles = 0, slotChunks = 0, groups = 0;  // groups: eight-step groups of the longest bucket
    uint32_t cellBytes = 4;  // element size of the slices this form stages: 4 (float), 2 or 1 (stored types)
    size_t stagedTiles = 0, directTiles = 0, stagedChunks = 0;
    DeviceArray<ForwardTile> tiles;
    DeviceArray<uint32_t> chunkOff;      // source cell index of every 16-byte chunk a tile stages, row by row
    DeviceArray<unsigned short> steps;   // LDS position of every cell of every bucket, in the order the lanes of a tile walk them
};
}  // namespace fimex_amd

struct fimex_amd_regrid_plan {
    int funcType = 0;
    int device = 0;
    fimex_amd::PlanKind kind = fimex_amd::PlanKind::Nearest;
    bool bicubicFast = false;  // FIMEX_AMD_BICUBIC_FAST: float fused multiply-adds in the LDS-staged bicubic kernel
    size_t inX = 0, inY = 0, outX = 0, outY = 0;

    // backward plans
    fimex_amd::DeviceArray<uint32_t> pos;
    fimex_amd::DeviceArray<float> xf, yf;
    fimex_amd::DeviceArray<double> xfd, yfd;
    fimex_amd::StagedPlan staged;
    fimex_amd::Staged2Plan staged2;
    // Second workgroup shape of the same plan (bilinear: 512 threads on 256 x 8 tiles next to 1024 threads on 512 x 8; bicubic in float
    // arithmetic: 256 threads on 128 x 8 next to 512 on 256 x 8), same results
    // bit for bit.  Which one is faster depends on the device at hand and on the batch length;
    // fimex_amd_regrid_plan_tune_device times both on the caller's buffers and sets useAlt.
    fimex_amd::Staged2Plan staged2Alt;
    int useAlt = 0;
    size_t planBytesShape[2] = {0, 0};  // info.planBytes with either shape

    // The second staged form for slices of 1- and 2-byte stored types (staged2.hip): built from this plan's own arrays on the
    // first typed apply of that element size, under the mutex (plans are shared by threads; everything else is immutable).
    struct TypedForms {
        std::mutex mtx;
        bool tried[2] = {false, false};  // [0] 2-byte elements, [1] 1-byte elements
        fimex_amd::Staged2Plan form[2];
    };
    mutable TypedForms typed2;

    // forward plans
    fimex_amd::Aggregate aggregate = fimex_amd::Aggregate::Sum;
    bool undefAggr = false;
    fimex_amd::DeviceArray<uint32_t> offsets, src;
    fimex_amd::ForwardTiles fwdTiles;
    // the same for slices of 2- and 1-byte stored types, built on the first typed apply of that element size (see TypedForms)
    struct FwdTypedForms {
        std::mutex mtx;
        bool tried[2] = {false, false};
        fimex_amd::ForwardTiles form[2];
    };
    mutable FwdTypedForms fwdTyped;

    fimex_amd_plan_info info{};
};

struct fimex_amd_vector_plan {
    int device = 0;
    size_t ox = 0, oy = 0;
    fimex_amd::DeviceArray<double2> cossin;  // (m0, m1) of the reference matrix
    fimex_amd::DeviceArray<double> phi;      // m3
};

namespace fimex_amd {

// regrid.hip
void build_backward_plan(fimex_amd_regrid_plan& plan, const double* d_px, const double* d_py, hipStream_t stream);
void launch_backward_apply(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream);
void launch_backward_gather(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream);

// staged.hip
bool build_staged_plan(fimex_amd_regrid_plan& plan, const double* d_px, const double* d_py, hipStream_t stream);
void launch_staged_apply(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream);
bool launch_staged_apply_typed(const fimex_amd_regrid_plan& plan, const void* d_in, int cdmType, size_t nz, double badValue, void* d_out,
                               hipStream_t stream);

// staged2.hip
bool build_staged2_plan(fimex_amd_regrid_plan& plan, const double* d_px, const double* d_py, hipStream_t stream);
void launch_staged2_apply(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream);
bool launch_staged2_apply_typed(const fimex_amd_regrid_plan& plan, const void* d_in, int cdmType, size_t nz, double badValue, void* d_out,
                                hipStream_t stream);

// forward.hip
void build_forward_plan(fimex_amd_regrid_plan& plan, const double* d_px, const double* d_py, hipStream_t stream);
void launch_forward_apply(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream);

// forward_tiled.hip
void build_forward_tiles(fimex_amd_regrid_plan& plan, hipStream_t stream);
bool launch_forward_tiled(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream);
bool launch_forward_tiled_typed(const fimex_amd_regrid_plan& plan, const void* d_in, int cdmType, size_t nz, double badValue, void* d_out,
                                hipStream_t stream);

// vector.hip
void build_vector_plan(fimex_amd_vector_plan& plan, const double* h_matrix);
void launch_vector_values(const fimex_amd_vector_plan& plan, float* d_u, float* d_v, size_t oz, hipStream_t stream);
void launch_vector_direction_scaled(const fimex_amd_vector_plan& plan, float* d_angles, size_t oz, double scale, double offset, hipStream_t stream);
void launch_vector_direction(const fimex_amd_vector_plan& plan, float* d_angles, size_t oz, hipStream_t stream);

// convert.hip
void launch_bad2nan(float* d, size_t n, float bad, hipStream_t stream);
void launch_nan2bad(float* d, size_t n, float bad, hipStream_t stream);
void launch_points2position(double* d_points, size_t n, const double* h_axis, int num, int axisType, hipStream_t stream);
bool launch_get_values_1d_f(int kind, const float* A, const float* B, float* out, size_t n, double a, double b, double x, hipStream_t stream);
void launch_get_values_linear_d(const double* A, const double* B, double* out, size_t n, double a, double b, double x, hipStream_t stream);
size_t cdm_type_size(int cdmType);  // throws for types without a float form
void launch_data2interpolation(const void* d_in, int cdmType, size_t n, double badValue, float* d_out, hipStream_t stream);
void launch_interpolation2data(const float* d_in, size_t n, int cdmType, double badValue, void* d_out, hipStream_t stream);

bool launch_typed_apply(const fimex_amd_regrid_plan& plan, const void* d_in, int cdmType, size_t nz, double badValue, void* d_out,
                        hipStream_t stream);

// projection.hip: pj_transform-level plan building on the device
void launch_project_values(const char* projIn, const char* projOut, double* d_x, double* d_y, size_t n, hipStream_t stream);
void launch_project_axes(const char* projIn, const char* projOut, const double* h_xAxis, const double* h_yAxis, size_t ix, size_t iy,
                         double* d_outX, double* d_outY, hipStream_t stream);
void launch_vector_reproject_matrix(const char* projIn, const char* projOut, const double* h_outXAxis, const double* h_outYAxis,
                                    int xAxisType, int yAxisType, size_t ox, size_t oy, double* d_matrix, hipStream_t stream);
void launch_vector_reproject_matrix_field(const char* projIn, const char* projOut, const double* h_inX, const double* h_inY, size_t ox,
                                          size_t oy, double* d_matrix, hipStream_t stream);
void launch_vector_reproject_matrix_points(const char* projIn, const char* projOut, int inputIsMetric, const double* h_outX,
                                           const double* h_outY, size_t on, double* d_matrix, hipStream_t stream);
int projection_is_degree(const char* proj);

// coordsearch.hip: coordinate-based nearest neighbour plans
double grid_distance(const double* d_lon, const double* d_lat, size_t orgX, size_t orgY, hipStream_t stream);
void launch_coord_nearest(double* d_pointsX, double* d_pointsY, size_t nPoints, const double* d_lon, const double* d_lat, size_t orgX,
                          size_t orgY, hipStream_t stream);
void launch_coord_kdtree(double maxDist, double* d_pointsX, double* d_pointsY, size_t nPoints, const double* d_lon, const double* d_lat,
                         size_t orgX, size_t orgY, hipStream_t stream);

// fill.hip
void run_fill2d(size_t nx, size_t ny, size_t nz, float* d_field, float relaxCrit, float corrEff, size_t maxLoop,
                size_t* h_nChanged, hipStream_t stream);
void run_creepfill(size_t nx, size_t ny, size_t nz, float* d_field, bool useDefault, float defaultVal,
                   unsigned short repeat, char setWeight, size_t* h_nChanged, hipStream_t stream);

void run_scan_sum(const float* d_values, size_t n, int mode, double average, int algo, double* h_sum, size_t* h_nUndefined,
                  hipStream_t stream);

// batch.hip: output batches placed by the library
fimex_amd_batch* batch_alloc(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, int positions, hipStream_t stream);
fimex_amd_batch* batch_alloc_source(const fimex_amd_regrid_plan& plan, size_t nz, int candidates, hipStream_t stream);
void batch_free(fimex_amd_batch* batch);
const fimex_amd_batch_info& batch_info(const fimex_amd_batch& batch);
void apply_plan_device(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream);

// hostpipe.hip: streamed transfers for the *_host entry points
using SliceChunkFn = std::function<void(const void* dRawIn, void* dRawOut, float* dFIn, float* dFOut, size_t nzc, hipStream_t stream)>;
bool pipelined_slices(int device, const void* in, size_t inSliceBytes, void* out, size_t outSliceBytes, size_t inSliceFloats,
                      size_t outSliceFloats, size_t nz, const SliceChunkFn& fn);

void host_to_device(void* d_dst, const void* h_src, size_t bytes, hipStream_t stream);
void device_to_host(void* h_dst, const void* d_src, size_t bytes, hipStream_t stream);
void release_host_pipes();

// tuning knobs read once from the environment (FIMEX_AMD_<NAME>), for bench sweeps
int tuning(const char* name, int fallback);

// batches shorter than this take the gather kernels: the staged kernels pay a per-tile set-up (chunk list, per-output plan)
// that only amortises over a few slices (launch_backward_apply and fimex_amd_regrid_plan_tune_device share the rule)
inline size_t staged_min_nz() { return (size_t)tuning("STAGED_MIN_NZ", 4); }

}  // namespace fimex_amd
