// extern "C" boundary of libfimex_amd.so (include/fimex_amd.h).
#include "plan.hpp"

#include <cstdlib>
#include <cstring>
#include <memory>
#include <algorithm>
#include <vector>

namespace fimex_amd {

namespace {
thread_local std::string g_lastError;
}

void set_last_error(const std::string& msg) { g_lastError = msg; }

// Experiment switches.  The product library does not read the environment: every switch has its
// measured default compiled in.  Only the tuning build (-DFIMEX_AMD_TUNING -> libfimex_amd_tuning.so, loaded by scripts/ and by
// the parity tests that force a fallback kernel) reads FIMEX_AMD_<NAME>, at every call so that one process can sweep settings.
int tuning(const char* name, int fallback)
{
#ifdef FIMEX_AMD_TUNING
    const std::string key = std::string("FIMEX_AMD_") + name;
    const char* v = std::getenv(key.c_str());
    if (!v || !*v) return fallback;
    const int parsed = std::atoi(v);
    return parsed < 0 ? fallback : parsed;
#else
    (void)name;
    return fallback;
#endif
}

namespace {

int usable_device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int usable = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) usable++;
    }
    return usable;
}

}  // namespace

int current_device_checked()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        (void)hipGetLastError();
        throw Error("no HIP device: fimex_amd has no CPU fallback, an MI355X (gfx950) is required");
    }
    int dev = 0;
    FA_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    FA_HIP(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        throw Error(std::string("device is ") + prop.gcnArchName + ", the kernels are built for gfx950 only");
    return dev;
}

void require_current_device(int planDevice)
{
    int dev = 0;
    FA_HIP(hipGetDevice(&dev));
    FA_REQUIRE(dev == planDevice, "plan lives on device " + std::to_string(planDevice) +
                                      " but the calling thread's current device is " + std::to_string(dev));
}

void apply_plan_device(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream)
{
    if (plan.kind == PlanKind::Forward) launch_forward_apply(plan, d_in, nz, d_out, stream);
    else launch_backward_apply(plan, d_in, nz, d_out, stream);
}

namespace {

bool is_backward(int funcType)
{
    return funcType == FIMEX_AMD_INTERPOL_NEAREST_NEIGHBOR || funcType == FIMEX_AMD_INTERPOL_BILINEAR ||
           funcType == FIMEX_AMD_INTERPOL_BICUBIC || funcType == FIMEX_AMD_INTERPOL_COORD_NN ||
           funcType == FIMEX_AMD_INTERPOL_COORD_NN_KD;
}

bool is_forward(int funcType)
{
    return funcType >= FIMEX_AMD_INTERPOL_FORWARD_SUM && funcType <= FIMEX_AMD_INTERPOL_FORWARD_UNDEF_MIN;
}

std::unique_ptr<fimex_amd_regrid_plan> new_plan(int funcType, size_t nPoints, size_t inX, size_t inY, size_t outX, size_t outY)
{
    // same failure as CachedInterpolation.cc:114 / CachedForwardInterpolation.cc:88
    FA_REQUIRE(is_backward(funcType) || is_forward(funcType), "unknown interpolation function: " + std::to_string(funcType));
    auto plan = std::make_unique<fimex_amd_regrid_plan>();
    plan->funcType = funcType;
    plan->inX = inX;
    plan->inY = inY;
    plan->outX = outX;
    plan->outY = outY;
    if (is_backward(funcType)) {
        plan->kind = funcType == FIMEX_AMD_INTERPOL_BILINEAR ? PlanKind::Bilinear
                   : funcType == FIMEX_AMD_INTERPOL_BICUBIC  ? PlanKind::Bicubic
                                                             : PlanKind::Nearest;
        FA_REQUIRE(nPoints == outX * outY, "backward plans need one position per output cell (outX*outY)");
    } else {
        plan->kind = PlanKind::Forward;
        const int k = (funcType - FIMEX_AMD_INTERPOL_FORWARD_SUM) % 5;
        plan->aggregate = static_cast<Aggregate>(k);  // sum, mean, median, max, min
        plan->undefAggr = funcType >= FIMEX_AMD_INTERPOL_FORWARD_UNDEF_SUM;
        FA_REQUIRE(nPoints == inX * inY, "forward plans need one position per input cell (inX*inY)");
    }
    plan->device = current_device_checked();
    plan->info.funcType = funcType;
    plan->info.device = plan->device;
    plan->info.inX = inX;
    plan->info.inY = inY;
    plan->info.outX = outX;
    plan->info.outY = outY;
    return plan;
}

void build_plan(fimex_amd_regrid_plan& plan, const double* d_px, const double* d_py, hipStream_t stream)
{
    if (plan.kind == PlanKind::Forward) build_forward_plan(plan, d_px, d_py, stream);
    else build_backward_plan(plan, d_px, d_py, stream);
}

void apply_device(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream)
{
    apply_plan_device(plan, d_in, nz, d_out, stream);
}

// host <-> device round trip shared by the in-place *_host entry points
template <typename F>
void with_device_copy(float* h, size_t n, hipStream_t stream, F&& body)
{
    DeviceArray<float> d(n);
    host_to_device(d.get(), h, n * sizeof(float), stream);
    body(d.get());
    device_to_host(h, d.get(), n * sizeof(float), stream);
    FA_HIP(hipStreamSynchronize(stream));
}

}  // namespace
}  // namespace fimex_amd

using namespace fimex_amd;

extern "C" {

const char* fimex_amd_last_error(void) { return g_lastError.c_str(); }

int fimex_amd_release_caches(void)
{
    return c_guard([&] { release_host_pipes(); });
}

int fimex_amd_abi_version(void) { return 131; }  // 1.31: source and output batches placed by the library, the gather cross-check

int fimex_amd_device_count(void) { return usable_device_count(); }

int fimex_amd_set_device(int ordinal)
{
    return c_guard([&] {
        FA_HIP(hipSetDevice(ordinal));
        (void)current_device_checked();
    });
}

int fimex_amd_regrid_plan_create(int funcType, const double* px, const double* py, size_t nPoints, size_t inX, size_t inY,
                                 size_t outX, size_t outY, fimex_amd_regrid_plan** out)
{
    return fimex_amd_regrid_plan_create_opt(funcType, px, py, nPoints, inX, inY, outX, outY, FIMEX_AMD_BICUBIC_REFERENCE, out);
}

int fimex_amd_regrid_plan_create_device(int funcType, const double* d_px, const double* d_py, size_t nPoints, size_t inX,
                                        size_t inY, size_t outX, size_t outY, void* stream, fimex_amd_regrid_plan** out)
{
    return fimex_amd_regrid_plan_create_device_opt(funcType, d_px, d_py, nPoints, inX, inY, outX, outY, FIMEX_AMD_BICUBIC_REFERENCE, stream, out);
}

static void set_arithmetic(fimex_amd_regrid_plan& plan, int bicubicArithmetic)
{
    FA_REQUIRE(bicubicArithmetic == FIMEX_AMD_BICUBIC_REFERENCE || bicubicArithmetic == FIMEX_AMD_BICUBIC_FAST,
               "unknown bicubic arithmetic: " + std::to_string(bicubicArithmetic));
    plan.bicubicFast = plan.kind == PlanKind::Bicubic && bicubicArithmetic == FIMEX_AMD_BICUBIC_FAST;
}

int fimex_amd_regrid_plan_create_opt(int funcType, const double* px, const double* py, size_t nPoints, size_t inX, size_t inY,
                                     size_t outX, size_t outY, int bicubicArithmetic, fimex_amd_regrid_plan** out)
{
    return c_guard([&] {
        FA_REQUIRE(out != nullptr, "plan output pointer is NULL");
        *out = nullptr;
        FA_REQUIRE(px != nullptr && py != nullptr, "position arrays are NULL");
        auto plan = new_plan(funcType, nPoints, inX, inY, outX, outY);
        set_arithmetic(*plan, bicubicArithmetic);
        ScopedStream stream;
        DeviceArray<double> d_px(nPoints), d_py(nPoints);
        host_to_device(d_px.get(), px, nPoints * sizeof(double), stream.get());
        host_to_device(d_py.get(), py, nPoints * sizeof(double), stream.get());
        build_plan(*plan, d_px.get(), d_py.get(), stream.get());
        stream.sync();
        *out = plan.release();
    });
}

int fimex_amd_regrid_plan_create_device_opt(int funcType, const double* d_px, const double* d_py, size_t nPoints, size_t inX,
                                            size_t inY, size_t outX, size_t outY, int bicubicArithmetic, void* stream,
                                            fimex_amd_regrid_plan** out)
{
    return c_guard([&] {
        FA_REQUIRE(out != nullptr, "plan output pointer is NULL");
        *out = nullptr;
        FA_REQUIRE(d_px != nullptr && d_py != nullptr, "position arrays are NULL");
        auto plan = new_plan(funcType, nPoints, inX, inY, outX, outY);
        set_arithmetic(*plan, bicubicArithmetic);
        build_plan(*plan, d_px, d_py, as_stream(stream));
        *out = plan.release();
    });
}

int fimex_amd_regrid_plan_destroy(fimex_amd_regrid_plan* plan)
{
    return c_guard([&] {
        if (!plan) return;
        ScopedDevice dev(plan->device);
        delete plan;
    });
}

int fimex_amd_regrid_plan_info(const fimex_amd_regrid_plan* plan, fimex_amd_plan_info* info)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr && info != nullptr, "NULL argument");
        *info = plan->info;
    });
}

int fimex_amd_regrid_apply_host(const fimex_amd_regrid_plan* plan, const float* inData, size_t size, float* outData,
                                size_t outCapacity, size_t* newSize)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr && newSize != nullptr, "NULL argument");
        const size_t inLayer = plan->inX * plan->inY, outLayer = plan->outX * plan->outY;
        const size_t nz = size / inLayer;  // CachedInterpolation.cc:121
        *newSize = outLayer * nz;          // :122
        if (outData == nullptr) return;    // size query
        FA_REQUIRE(inData != nullptr || nz == 0, "inData is NULL");
        FA_REQUIRE(outCapacity >= *newSize, "output buffer too small");
        if (nz == 0) return;
        ScopedDevice dev(plan->device);
        // slices are independent: stream them through pinned staging, transfers overlapping the kernels
        if (pipelined_slices(plan->device, inData, inLayer * sizeof(float), outData, outLayer * sizeof(float), 0, 0, nz,
                             [&](const void* dIn, void* dOut, float*, float*, size_t nzc, hipStream_t st) {
                                 apply_device(*plan, static_cast<const float*>(dIn), nzc, static_cast<float*>(dOut), st);
                             }))
            return;
        ScopedStream stream;
        DeviceArray<float> d_in(nz * inLayer), d_out(nz * outLayer);
        host_to_device(d_in.get(), inData, d_in.bytes(), stream.get());
        apply_device(*plan, d_in.get(), nz, d_out.get(), stream.get());
        device_to_host(outData, d_out.get(), d_out.bytes(), stream.get());
        stream.sync();
    });
}

int fimex_amd_regrid_apply_device(const fimex_amd_regrid_plan* plan, const float* d_in, size_t nz, float* d_out, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr, "NULL plan");
        if (nz == 0) return;
        FA_REQUIRE(d_in != nullptr && d_out != nullptr, "NULL device buffer");
        require_current_device(plan->device);
        apply_device(*plan, d_in, nz, d_out, as_stream(stream));
    });
}

int fimex_amd_regrid_plan_tune_device(fimex_amd_regrid_plan* plan, const float* d_in, size_t nz, float* d_out, void* stream, int* chosenShape)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr, "NULL plan");
        if (chosenShape) *chosenShape = plan->useAlt;
        // nothing to choose between: no second shape, or a batch that takes the gather kernels anyway (the choice made for the
        // long batches stays as it is)
        if (nz < staged_min_nz() || !plan->staged2Alt.valid || !plan->staged2.valid) return;
        FA_REQUIRE(d_in != nullptr && d_out != nullptr, "NULL device buffer");
        require_current_device(plan->device);
        hipStream_t st = as_stream(stream);
        hipEvent_t e0, e1;
        FA_HIP(hipEventCreate(&e0));
        FA_HIP(hipEventCreate(&e1));
        float best[2] = {0.f, 0.f};
        try {
            for (int shape = 0; shape < 2; ++shape) {
                plan->useAlt = shape;
                std::vector<float> ms;
                for (int rep = 0; rep < 7; ++rep) {  // two launches to settle, five timed: the median counts
                    FA_HIP(hipEventRecord(e0, st));
                    apply_device(*plan, d_in, nz, d_out, st);
                    FA_HIP(hipEventRecord(e1, st));
                    FA_HIP(hipEventSynchronize(e1));
                    float t = 0.f;
                    FA_HIP(hipEventElapsedTime(&t, e0, e1));
                    if (rep >= 2) ms.push_back(t);
                }
                std::sort(ms.begin(), ms.end());
                best[shape] = ms[ms.size() / 2];
            }
        } catch (...) {
            plan->useAlt = 0;
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            throw;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        plan->useAlt = best[1] < 0.99f * best[0] ? 1 : 0;  // the default shape unless the other one is clearly faster
        const auto& s = plan->useAlt ? plan->staged2Alt : plan->staged2;
        plan->info.planBytes = plan->planBytesShape[plan->useAlt];
        plan->info.stagedCells = s.stagedCells;
        plan->info.tileW = s.tileWMax;
        plan->info.tileH = s.tileH;
        if (chosenShape) *chosenShape = plan->useAlt;
    });
}

int fimex_amd_regrid_apply_gather_device(const fimex_amd_regrid_plan* plan, const float* d_in, size_t nz, float* d_out, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr, "NULL plan");
        FA_REQUIRE(plan->kind != PlanKind::Forward, "the gather kernels serve backward plans");
        if (nz == 0) return;
        FA_REQUIRE(d_in != nullptr && d_out != nullptr, "NULL device buffer");
        require_current_device(plan->device);
        launch_backward_gather(*plan, d_in, nz, d_out, as_stream(stream));
    });
}

int fimex_amd_regrid_batch_alloc_device(const fimex_amd_regrid_plan* plan, const float* d_in, size_t nz, int positions, void* stream,
                                        fimex_amd_batch** batch)
{
    return c_guard([&] {
        FA_REQUIRE(batch != nullptr, "batch output pointer is NULL");
        *batch = nullptr;
        FA_REQUIRE(plan != nullptr, "NULL plan");
        require_current_device(plan->device);
        *batch = batch_alloc(*plan, d_in, nz, positions, as_stream(stream));
    });
}

int fimex_amd_regrid_source_batch_alloc_device(const fimex_amd_regrid_plan* plan, size_t nz, int candidates, void* stream, fimex_amd_batch** batch)
{
    return c_guard([&] {
        FA_REQUIRE(batch != nullptr, "batch output pointer is NULL");
        *batch = nullptr;
        FA_REQUIRE(plan != nullptr, "NULL plan");
        require_current_device(plan->device);
        *batch = batch_alloc_source(*plan, nz, candidates, as_stream(stream));
    });
}

int fimex_amd_batch_get_info(const fimex_amd_batch* batch, fimex_amd_batch_info* info)
{
    return c_guard([&] {
        FA_REQUIRE(batch != nullptr && info != nullptr, "NULL argument");
        *info = batch_info(*batch);
    });
}

int fimex_amd_batch_free(fimex_amd_batch* batch)
{
    return c_guard([&] { batch_free(batch); });
}

int fimex_amd_vector_plan_create(const double* matrix, size_t ox, size_t oy, fimex_amd_vector_plan** out)
{
    return c_guard([&] {
        FA_REQUIRE(out != nullptr, "plan output pointer is NULL");
        *out = nullptr;
        FA_REQUIRE(matrix != nullptr, "matrix is NULL");
        auto plan = std::make_unique<fimex_amd_vector_plan>();
        plan->device = current_device_checked();
        plan->ox = ox;
        plan->oy = oy;
        build_vector_plan(*plan, matrix);
        *out = plan.release();
    });
}

int fimex_amd_vector_plan_destroy(fimex_amd_vector_plan* plan)
{
    return c_guard([&] {
        if (!plan) return;
        ScopedDevice dev(plan->device);
        delete plan;
    });
}

int fimex_amd_vector_reproject_values_host(const fimex_amd_vector_plan* plan, float* u, float* v, size_t size)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr, "NULL plan");
        const size_t layer = plan->ox * plan->oy;
        const size_t oz = size / layer;  // CachedVectorReprojection.cc:41
        if (oz == 0) return;
        FA_REQUIRE(u != nullptr && v != nullptr, "NULL buffer");
        ScopedDevice dev(plan->device);
        ScopedStream stream;
        const size_t n = oz * layer;
        DeviceArray<float> d_u(n), d_v(n);
        host_to_device(d_u.get(), u, n * sizeof(float), stream.get());
        host_to_device(d_v.get(), v, n * sizeof(float), stream.get());
        launch_vector_values(*plan, d_u.get(), d_v.get(), oz, stream.get());
        device_to_host(u, d_u.get(), n * sizeof(float), stream.get());
        device_to_host(v, d_v.get(), n * sizeof(float), stream.get());
        stream.sync();
    });
}

int fimex_amd_vector_reproject_values_device(const fimex_amd_vector_plan* plan, float* d_u, float* d_v, size_t oz, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr, "NULL plan");
        if (oz == 0) return;
        FA_REQUIRE(d_u != nullptr && d_v != nullptr, "NULL device buffer");
        require_current_device(plan->device);
        launch_vector_values(*plan, d_u, d_v, oz, as_stream(stream));
    });
}

int fimex_amd_vector_reproject_direction_host(const fimex_amd_vector_plan* plan, float* angles, size_t size)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr, "NULL plan");
        const size_t layer = plan->ox * plan->oy;
        const size_t oz = size / layer;  // CachedVectorReprojection.cc:52
        if (oz == 0) return;
        FA_REQUIRE(angles != nullptr, "NULL buffer");
        ScopedDevice dev(plan->device);
        ScopedStream stream;
        with_device_copy(angles, oz * layer, stream.get(),
                         [&](float* d) { launch_vector_direction(*plan, d, oz, stream.get()); });
    });
}

int fimex_amd_vector_reproject_direction_device(const fimex_amd_vector_plan* plan, float* d_angles, size_t oz, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr, "NULL plan");
        if (oz == 0) return;
        FA_REQUIRE(d_angles != nullptr, "NULL device buffer");
        require_current_device(plan->device);
        launch_vector_direction(*plan, d_angles, oz, as_stream(stream));
    });
}

int fimex_amd_fill2d_host(size_t nx, size_t ny, size_t nz, float* field, float relaxCrit, float corrEff, size_t maxLoop,
                          size_t* nChanged)
{
    return c_guard([&] {
        if (nx * ny * nz == 0) return;
        FA_REQUIRE(field != nullptr, "NULL buffer");
        (void)current_device_checked();
        ScopedStream stream;
        with_device_copy(field, nx * ny * nz, stream.get(), [&](float* d) {
            run_fill2d(nx, ny, nz, d, relaxCrit, corrEff, maxLoop, nChanged, stream.get());
        });
    });
}

int fimex_amd_fill2d_device(size_t nx, size_t ny, size_t nz, float* d_field, float relaxCrit, float corrEff, size_t maxLoop,
                            size_t* nChanged, void* stream)
{
    return c_guard([&] {
        if (nx * ny * nz == 0) return;
        FA_REQUIRE(d_field != nullptr, "NULL device buffer");
        (void)current_device_checked();
        run_fill2d(nx, ny, nz, d_field, relaxCrit, corrEff, maxLoop, nChanged, as_stream(stream));
    });
}

int fimex_amd_creepfill2d_host(size_t nx, size_t ny, size_t nz, float* field, unsigned short repeat, char setWeight,
                               size_t* nChanged)
{
    return c_guard([&] {
        if (nx * ny * nz == 0) return;
        FA_REQUIRE(field != nullptr, "NULL buffer");
        (void)current_device_checked();
        ScopedStream stream;
        with_device_copy(field, nx * ny * nz, stream.get(), [&](float* d) {
            run_creepfill(nx, ny, nz, d, false, 0.f, repeat, setWeight, nChanged, stream.get());
        });
    });
}

int fimex_amd_creepfill2d_device(size_t nx, size_t ny, size_t nz, float* d_field, unsigned short repeat, char setWeight,
                                 size_t* nChanged, void* stream)
{
    return c_guard([&] {
        if (nx * ny * nz == 0) return;
        FA_REQUIRE(d_field != nullptr, "NULL device buffer");
        (void)current_device_checked();
        run_creepfill(nx, ny, nz, d_field, false, 0.f, repeat, setWeight, nChanged, as_stream(stream));
    });
}

int fimex_amd_creepfillval2d_host(size_t nx, size_t ny, size_t nz, float* field, float defaultVal, unsigned short repeat,
                                  char setWeight, size_t* nChanged)
{
    return c_guard([&] {
        if (nx * ny * nz == 0) return;
        FA_REQUIRE(field != nullptr, "NULL buffer");
        (void)current_device_checked();
        ScopedStream stream;
        with_device_copy(field, nx * ny * nz, stream.get(), [&](float* d) {
            run_creepfill(nx, ny, nz, d, true, defaultVal, repeat, setWeight, nChanged, stream.get());
        });
    });
}

int fimex_amd_creepfillval2d_device(size_t nx, size_t ny, size_t nz, float* d_field, float defaultVal, unsigned short repeat,
                                    char setWeight, size_t* nChanged, void* stream)
{
    return c_guard([&] {
        if (nx * ny * nz == 0) return;
        FA_REQUIRE(d_field != nullptr, "NULL device buffer");
        (void)current_device_checked();
        run_creepfill(nx, ny, nz, d_field, true, defaultVal, repeat, setWeight, nChanged, as_stream(stream));
    });
}

int fimex_amd_bad2nan_device(float* d_data, size_t n, float badVal, void* stream)
{
    return c_guard([&] {
        if (n == 0) return;
        FA_REQUIRE(d_data != nullptr, "NULL device buffer");
        (void)current_device_checked();
        launch_bad2nan(d_data, n, badVal, as_stream(stream));
    });
}

int fimex_amd_nan2bad_device(float* d_data, size_t n, float badVal, void* stream)
{
    return c_guard([&] {
        if (n == 0) return;
        FA_REQUIRE(d_data != nullptr, "NULL device buffer");
        (void)current_device_checked();
        launch_nan2bad(d_data, n, badVal, as_stream(stream));
    });
}

int fimex_amd_points2position_device(double* d_points, size_t n, const double* axis, int num, int axis_type, void* stream)
{
    return c_guard([&] {
        if (n == 0) return;
        FA_REQUIRE(d_points != nullptr && axis != nullptr, "NULL argument");
        (void)current_device_checked();
        launch_points2position(d_points, n, axis, num, axis_type, as_stream(stream));
    });
}

namespace {

// CDMInterpolator::getDataSlice, src/CDMInterpolator.cc:251-285, on one step of one variable; typed == false is the
// float-in / float-out form (conversions reduced to mifi_bad2nanf / mifi_nanf2bad with a float fill value)
void regrid_slice(const fimex_amd_regrid_plan* plan, bool typed, const void* inData, int dataType, size_t size, double badValue,
                  const fimex_amd_process2d* pre, size_t nPre, const void* counterpart, int counterpartType,
                  double badValueCounterpart, const fimex_amd_vector_plan* vec, int isXComponent,
                  const fimex_amd_process2d* post, size_t nPost, void* outData, size_t outCapacity, size_t* newSize)
{
    FA_REQUIRE(plan != nullptr && newSize != nullptr, "NULL argument");
    FA_REQUIRE((nPre == 0 || pre != nullptr) && (nPost == 0 || post != nullptr), "NULL process list");
    const size_t inLayer = plan->inX * plan->inY, outLayer = plan->outX * plan->outY;
    const size_t nz = size / inLayer;
    *newSize = outLayer * nz;
    if (outData == nullptr) return;
    FA_REQUIRE(outCapacity >= *newSize, "output buffer too small");
    const size_t elem = typed ? cdm_type_size(dataType) : sizeof(float);
    if (nz == 0) return;
    FA_REQUIRE(inData != nullptr, "inData is NULL");
    const bool vector = counterpart != nullptr && vec != nullptr;
    if (vector) FA_REQUIRE(vec->device == plan->device && vec->ox == plan->outX && vec->oy == plan->outY,
                           "vector reprojection does not match the regrid plan");
    const size_t elemOther = (vector && typed) ? cdm_type_size(counterpartType) : sizeof(float);
    ScopedDevice dev(plan->device);
    if (!vector && nPre == 0 && nPost == 0) {
        // conversion, regrid, conversion per chunk of slices, transfers streamed
        const bool convert = typed && dataType != FIMEX_AMD_CDM_FLOAT;
        if (pipelined_slices(plan->device, inData, inLayer * elem, outData, outLayer * elem, convert ? inLayer : 0, typed ? outLayer : 0, nz,
                             [&](const void* dIn, void* dOut, float* fIn, float* fOut, size_t nzc, hipStream_t st) {
                                 if (typed && launch_typed_apply(*plan, dIn, dataType, nzc, badValue, dOut, st)) return;
                                 const float* src = static_cast<const float*>(dIn);
                                 if (convert) {
                                     launch_data2interpolation(dIn, dataType, nzc * inLayer, badValue, fIn, st);
                                     src = fIn;
                                 } else {
                                     launch_bad2nan(const_cast<float*>(src), nzc * inLayer, (float)badValue, st);  // staging copy, not the caller's
                                 }
                                 if (typed) {
                                     apply_device(*plan, src, nzc, fOut, st);
                                     launch_interpolation2data(fOut, nzc * outLayer, dataType, badValue, dOut, st);
                                 } else {
                                     apply_device(*plan, src, nzc, static_cast<float*>(dOut), st);
                                     launch_nan2bad(static_cast<float*>(dOut), nzc * outLayer, (float)badValue, st);
                                 }
                             }))
            return;
    }
    ScopedStream stream;
    hipStream_t st = stream.get();
    auto run = [&](const fimex_amd_process2d* list, size_t n, float* d, size_t nx, size_t ny) {
        for (size_t i = 0; i < n; ++i) {
            const fimex_amd_process2d& p = list[i];
            switch (p.kind) {
            case FIMEX_AMD_PROCESS_FILL2D: run_fill2d(nx, ny, nz, d, p.relaxCrit, p.corrEff, p.maxLoop, nullptr, st); break;
            case FIMEX_AMD_PROCESS_CREEPFILL2D: run_creepfill(nx, ny, nz, d, false, 0.f, p.repeat, p.setWeight, nullptr, st); break;
            case FIMEX_AMD_PROCESS_CREEPFILLVAL2D: run_creepfill(nx, ny, nz, d, true, p.defaultVal, p.repeat, p.setWeight, nullptr, st); break;
            default: throw Error("unknown 2-D process kind " + std::to_string(p.kind));
            }
        }
    };
    // one component: upload in its stored type, -> float with the fill value as NaN, pre-processes, regrid
    auto regrid = [&](const void* h_in, int type, size_t bytesPerElem, double bad, DeviceArray<float>& d_out) {
        DeviceArray<float> d_in(nz * inLayer);
        DeviceArray<unsigned char> d_raw;
        if (typed && type != FIMEX_AMD_CDM_FLOAT) {
            d_raw.allocate(nz * inLayer * bytesPerElem);
            host_to_device(d_raw.get(), h_in, d_raw.bytes(), st);
            launch_data2interpolation(d_raw.get(), type, d_in.size(), bad, d_in.get(), st);
        } else {
            host_to_device(d_in.get(), h_in, d_in.bytes(), st);
            launch_bad2nan(d_in.get(), d_in.size(), (float)bad, st);
        }
        run(pre, nPre, d_in.get(), plan->inX, plan->inY);
        d_out.allocate(nz * outLayer);
        apply_device(*plan, d_in.get(), nz, d_out.get(), st);
        FA_HIP(hipStreamSynchronize(st));  // d_in / d_raw are released on return
    };
    DeviceArray<float> d_main, d_other;
    regrid(inData, dataType, elem, badValue, d_main);
    if (vector) {
        regrid(counterpart, counterpartType, elemOther, badValueCounterpart, d_other);
        if (isXComponent) launch_vector_values(*vec, d_main.get(), d_other.get(), nz, st);
        else launch_vector_values(*vec, d_other.get(), d_main.get(), nz, st);
    }
    run(post, nPost, d_main.get(), plan->outX, plan->outY);
    if (typed) {
        DeviceArray<unsigned char> d_typed(d_main.size() * elem);
        launch_interpolation2data(d_main.get(), d_main.size(), dataType, badValue, d_typed.get(), st);
        device_to_host(outData, d_typed.get(), d_typed.bytes(), st);
        stream.sync();
    } else {
        launch_nan2bad(d_main.get(), d_main.size(), (float)badValue, st);
        device_to_host(outData, d_main.get(), d_main.bytes(), st);
        stream.sync();
    }
}

}  // namespace

int fimex_amd_regrid_slice_host(const fimex_amd_regrid_plan* plan, const float* inData, size_t size, float badValue,
                                const fimex_amd_process2d* pre, size_t nPre, const float* counterpart,
                                float badValueCounterpart, const fimex_amd_vector_plan* vec, int isXComponent,
                                const fimex_amd_process2d* post, size_t nPost, float* outData, size_t outCapacity,
                                size_t* newSize)
{
    return c_guard([&] {
        regrid_slice(plan, false, inData, FIMEX_AMD_CDM_FLOAT, size, badValue, pre, nPre, counterpart, FIMEX_AMD_CDM_FLOAT,
                     badValueCounterpart, vec, isXComponent, post, nPost, outData, outCapacity, newSize);
    });
}

int fimex_amd_regrid_slice_typed_host(const fimex_amd_regrid_plan* plan, const void* inData, int dataType, size_t size, double badValue,
                                      const fimex_amd_process2d* pre, size_t nPre, const void* counterpart, int counterpartType,
                                      double badValueCounterpart, const fimex_amd_vector_plan* vec, int isXComponent,
                                      const fimex_amd_process2d* post, size_t nPost, void* outData, size_t outCapacity,
                                      size_t* newSize)
{
    return c_guard([&] {
        regrid_slice(plan, true, inData, dataType, size, badValue, pre, nPre, counterpart, counterpartType, badValueCounterpart,
                     vec, isXComponent, post, nPost, outData, outCapacity, newSize);
    });
}

int fimex_amd_data2interpolation_device(const void* d_in, int cdmType, size_t n, double badValue, float* d_out, void* stream)
{
    return c_guard([&] {
        (void)cdm_type_size(cdmType);
        if (n == 0) return;
        FA_REQUIRE(d_in != nullptr && d_out != nullptr, "NULL device buffer");
        (void)current_device_checked();
        launch_data2interpolation(d_in, cdmType, n, badValue, d_out, as_stream(stream));
    });
}

int fimex_amd_regrid_apply_typed_device(const fimex_amd_regrid_plan* plan, const void* d_in, int cdmType, size_t nz, double badValue,
                                        void* d_out, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr, "NULL plan");
        (void)cdm_type_size(cdmType);
        if (nz == 0) return;
        FA_REQUIRE(d_in != nullptr && d_out != nullptr, "NULL device buffer");
        ScopedDevice dev(plan->device);
        hipStream_t st = as_stream(stream);
        if (launch_typed_apply(*plan, d_in, cdmType, nz, badValue, d_out, st)) return;
        const size_t inLayer = plan->inX * plan->inY, outLayer = plan->outX * plan->outY;
        DeviceArray<float> fIn(nz * inLayer), fOut(nz * outLayer);
        launch_data2interpolation(d_in, cdmType, nz * inLayer, badValue, fIn.get(), st);
        apply_device(*plan, fIn.get(), nz, fOut.get(), st);
        launch_interpolation2data(fOut.get(), nz * outLayer, cdmType, badValue, d_out, st);
        FA_HIP(hipStreamSynchronize(st));  // the temporaries are released on return
    });
}

int fimex_amd_data2interpolation_host(const void* in, int cdmType, size_t n, double badValue, float* out)
{
    return c_guard([&] {
        const size_t elem = cdm_type_size(cdmType);
        if (n == 0) return;
        FA_REQUIRE(in != nullptr && out != nullptr, "NULL argument");
        (void)current_device_checked();
        ScopedStream stream;
        DeviceArray<unsigned char> d_in(n * elem);
        DeviceArray<float> d_out(n);
        host_to_device(d_in.get(), in, d_in.bytes(), stream.get());
        launch_data2interpolation(d_in.get(), cdmType, n, badValue, d_out.get(), stream.get());
        device_to_host(out, d_out.get(), d_out.bytes(), stream.get());
        stream.sync();
    });
}

int fimex_amd_interpolation2data_host(const float* in, size_t n, int cdmType, double badValue, void* out)
{
    return c_guard([&] {
        const size_t elem = cdm_type_size(cdmType);
        if (n == 0) return;
        FA_REQUIRE(in != nullptr && out != nullptr, "NULL argument");
        (void)current_device_checked();
        ScopedStream stream;
        DeviceArray<float> d_in(n);
        DeviceArray<unsigned char> d_out(n * elem);
        host_to_device(d_in.get(), in, d_in.bytes(), stream.get());
        launch_interpolation2data(d_in.get(), n, cdmType, badValue, d_out.get(), stream.get());
        device_to_host(out, d_out.get(), d_out.bytes(), stream.get());
        stream.sync();
    });
}

int fimex_amd_interpolation2data_device(const float* d_in, size_t n, int cdmType, double badValue, void* d_out, void* stream)
{
    return c_guard([&] {
        (void)cdm_type_size(cdmType);
        if (n == 0) return;
        FA_REQUIRE(d_in != nullptr && d_out != nullptr, "NULL device buffer");
        (void)current_device_checked();
        launch_interpolation2data(d_in, n, cdmType, badValue, d_out, as_stream(stream));
    });
}

int fimex_amd_points2position_host(double* points, size_t n, const double* axis, int num, int axis_type)
{
    return c_guard([&] {
        if (n == 0) return;
        FA_REQUIRE(points != nullptr && axis != nullptr, "NULL argument");
        (void)current_device_checked();
        ScopedStream stream;
        DeviceArray<double> d(n);
        host_to_device(d.get(), points, n * sizeof(double), stream.get());
        launch_points2position(d.get(), n, axis, num, axis_type, stream.get());
        device_to_host(points, d.get(), n * sizeof(double), stream.get());
        stream.sync();
    });
}

int fimex_amd_get_values_1d_f_device(int kind, const float* d_A, const float* d_B, float* d_out, size_t n, double a, double b, double x, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(n == 0 || (d_A != nullptr && d_B != nullptr && d_out != nullptr), "NULL device buffer");
        (void)current_device_checked();
        if (!launch_get_values_1d_f(kind, d_A, d_B, d_out, n, a, b, x, as_stream(stream)))
            throw Error("log blend needs positive coordinates (src/interpolation.c:1137, 1149)");
    });
}

int fimex_amd_get_values_1d_f_host(int kind, const float* A, const float* B, float* out, size_t n, double a, double b, double x)
{
    return c_guard([&] {
        FA_REQUIRE(n == 0 || (A != nullptr && B != nullptr && out != nullptr), "NULL argument");
        (void)current_device_checked();
        ScopedStream stream;
        DeviceArray<float> d(3 * n);
        if (n) {
            host_to_device(d.get(), A, n * sizeof(float), stream.get());
            host_to_device(d.get() + n, B, n * sizeof(float), stream.get());
        }
        if (!launch_get_values_1d_f(kind, d.get(), d.get() + n, d.get() + 2 * n, n, a, b, x, stream.get()))
            throw Error("log blend needs positive coordinates (src/interpolation.c:1137, 1149)");
        if (n) device_to_host(out, d.get() + 2 * n, n * sizeof(float), stream.get());
        stream.sync();
    });
}

int fimex_amd_get_values_linear_d_device(const double* d_A, const double* d_B, double* d_out, size_t n, double a, double b, double x, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(n == 0 || (d_A != nullptr && d_B != nullptr && d_out != nullptr), "NULL device buffer");
        (void)current_device_checked();
        launch_get_values_linear_d(d_A, d_B, d_out, n, a, b, x, as_stream(stream));
    });
}

int fimex_amd_project_values_device(const char* proj_input, const char* proj_output, double* d_x, double* d_y, size_t num, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(num == 0 || (d_x != nullptr && d_y != nullptr), "NULL device buffer");
        (void)current_device_checked();
        launch_project_values(proj_input, proj_output, d_x, d_y, num, as_stream(stream));
    });
}

int fimex_amd_project_values_host(const char* proj_input, const char* proj_output, double* x, double* y, size_t num)
{
    return c_guard([&] {
        FA_REQUIRE(num == 0 || (x != nullptr && y != nullptr), "NULL argument");
        (void)current_device_checked();
        ScopedStream stream;
        DeviceArray<double> d(2 * num);
        if (num) {
            host_to_device(d.get(), x, num * sizeof(double), stream.get());
            host_to_device(d.get() + num, y, num * sizeof(double), stream.get());
        }
        launch_project_values(proj_input, proj_output, d.get(), d.get() + num, num, stream.get());
        if (num) {
            device_to_host(x, d.get(), num * sizeof(double), stream.get());
            device_to_host(y, d.get() + num, num * sizeof(double), stream.get());
        }
        stream.sync();
    });
}

int fimex_amd_project_axes_device(const char* proj_input, const char* proj_output, const double* in_x_axis, const double* in_y_axis,
                                  size_t ix, size_t iy, double* d_outX, double* d_outY, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(ix * iy == 0 || (in_x_axis != nullptr && in_y_axis != nullptr && d_outX != nullptr && d_outY != nullptr), "NULL argument");
        (void)current_device_checked();
        launch_project_axes(proj_input, proj_output, in_x_axis, in_y_axis, ix, iy, d_outX, d_outY, as_stream(stream));
    });
}

int fimex_amd_project_axes_host(const char* proj_input, const char* proj_output, const double* in_x_axis, const double* in_y_axis,
                                size_t ix, size_t iy, double* outX, double* outY)
{
    return c_guard([&] {
        const size_t n = ix * iy;
        FA_REQUIRE(n == 0 || (in_x_axis != nullptr && in_y_axis != nullptr && outX != nullptr && outY != nullptr), "NULL argument");
        (void)current_device_checked();
        ScopedStream stream;
        DeviceArray<double> d(2 * n);
        launch_project_axes(proj_input, proj_output, in_x_axis, in_y_axis, ix, iy, d.get(), d.get() + n, stream.get());
        if (n) {
            device_to_host(outX, d.get(), n * sizeof(double), stream.get());
            device_to_host(outY, d.get() + n, n * sizeof(double), stream.get());
        }
        stream.sync();
    });
}

int fimex_amd_get_vector_reproject_matrix_device(const char* proj_input, const char* proj_output, const double* out_x_axis,
                                                 const double* out_y_axis, int xType, int yType, size_t ox, size_t oy,
                                                 double* d_matrix, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(ox * oy == 0 || (out_x_axis != nullptr && out_y_axis != nullptr && d_matrix != nullptr), "NULL argument");
        (void)current_device_checked();
        launch_vector_reproject_matrix(proj_input, proj_output, out_x_axis, out_y_axis, xType, yType, ox, oy, d_matrix, as_stream(stream));
    });
}

int fimex_amd_get_vector_reproject_matrix_host(const char* proj_input, const char* proj_output, const double* out_x_axis,
                                               const double* out_y_axis, int xType, int yType, size_t ox, size_t oy, double* matrix)
{
    return c_guard([&] {
        const size_t n = ox * oy;
        FA_REQUIRE(n == 0 || (out_x_axis != nullptr && out_y_axis != nullptr && matrix != nullptr), "NULL argument");
        (void)current_device_checked();
        ScopedStream stream;
        DeviceArray<double> d(4 * n);
        launch_vector_reproject_matrix(proj_input, proj_output, out_x_axis, out_y_axis, xType, yType, ox, oy, d.get(), stream.get());
        if (n) device_to_host(matrix, d.get(), d.bytes(), stream.get());
        stream.sync();
    });
}

int fimex_amd_get_vector_reproject_matrix_field_host(const char* proj_input, const char* proj_output, const double* in_x_field,
                                                     const double* in_y_field, size_t ox, size_t oy, double* matrix)
{
    return c_guard([&] {
        const size_t n = ox * oy;
        FA_REQUIRE(n == 0 || (in_x_field != nullptr && in_y_field != nullptr && matrix != nullptr), "NULL argument");
        (void)current_device_checked();
        ScopedStream stream;
        DeviceArray<double> d(4 * n);
        launch_vector_reproject_matrix_field(proj_input, proj_output, in_x_field, in_y_field, ox, oy, d.get(), stream.get());
        if (n) device_to_host(matrix, d.get(), d.bytes(), stream.get());
        stream.sync();
    });
}

int fimex_amd_get_vector_reproject_matrix_points_host(const char* proj_input, const char* proj_output, int inputIsMetric,
                                                      const double* out_x_points, const double* out_y_points, size_t on, double* matrix)
{
    return c_guard([&] {
        FA_REQUIRE(on == 0 || (out_x_points != nullptr && out_y_points != nullptr && matrix != nullptr), "NULL argument");
        (void)current_device_checked();
        ScopedStream stream;
        DeviceArray<double> d(4 * on);
        launch_vector_reproject_matrix_points(proj_input, proj_output, inputIsMetric, out_x_points, out_y_points, on, d.get(), stream.get());
        if (on) device_to_host(matrix, d.get(), d.bytes(), stream.get());
        stream.sync();
    });
}

int fimex_amd_vector_reproject_direction_scaled_device(const fimex_amd_vector_plan* plan, float* d_angles, size_t oz, double scale,
                                                       double offset, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr && (oz == 0 || d_angles != nullptr), "NULL argument");
        ScopedDevice dev(plan->device);
        launch_vector_direction_scaled(*plan, d_angles, oz, scale, offset, as_stream(stream));
    });
}

int fimex_amd_vector_reproject_direction_scaled_host(const fimex_amd_vector_plan* plan, float* angles, size_t size, double scale, double offset)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr, "NULL argument");
        const size_t layer = plan->ox * plan->oy, oz = layer ? size / layer : 0;
        if (oz == 0) return;
        FA_REQUIRE(angles != nullptr, "NULL argument");
        ScopedDevice dev(plan->device);
        ScopedStream stream;
        DeviceArray<float> d(oz * layer);
        host_to_device(d.get(), angles, d.bytes(), stream.get());
        launch_vector_direction_scaled(*plan, d.get(), oz, scale, offset, stream.get());
        device_to_host(angles, d.get(), d.bytes(), stream.get());
        stream.sync();
    });
}

int fimex_amd_rotate_vector_typed_host(const fimex_amd_vector_plan* plan, const void* xData, int xType, double xFill, const void* yData,
                                       int yType, double yFill, size_t size, int returnX, int outType, double outFill, void* outData)
{
    return c_guard([&] {
        FA_REQUIRE(plan != nullptr, "NULL argument");
        const size_t ex = cdm_type_size(xType), ey = cdm_type_size(yType), eo = cdm_type_size(outType);
        const size_t layer = plan->ox * plan->oy, oz = layer ? size / layer : 0;
        if (size == 0) return;
        FA_REQUIRE(xData != nullptr && yData != nullptr && outData != nullptr, "NULL argument");
        ScopedDevice dev(plan->device);
        ScopedStream stream;
        hipStream_t st = stream.get();
        DeviceArray<unsigned char> rawX(size * ex), rawY(size * ey), rawOut(size * eo);
        DeviceArray<float> u(size), v(size);
        host_to_device(rawX.get(), xData, rawX.bytes(), st);
        host_to_device(rawY.get(), yData, rawY.bytes(), st);
        launch_data2interpolation(rawX.get(), xType, size, xFill, u.get(), st);   // CDMProcessor.cc:607-608
        launch_data2interpolation(rawY.get(), yType, size, yFill, v.get(), st);
        launch_vector_values(*plan, u.get(), v.get(), oz, st);                      // :612 (whole slices only, as the reference)
        launch_interpolation2data(returnX ? u.get() : v.get(), size, outType, outFill, rawOut.get(), st);  // :614-618
        device_to_host(outData, rawOut.get(), rawOut.bytes(), st);
        stream.sync();
    });
}

int fimex_amd_projection_is_degree(const char* proj)
{
    int r = -1;
    const int rc = c_guard([&] { r = projection_is_degree(proj); });
    return rc == FIMEX_AMD_OK ? r : -1;
}

}  // extern "C"

namespace {
template <typename F>
void coord_search_host(double* px, double* py, size_t nPoints, const double* lon, const double* lat, size_t orgX, size_t orgY, F&& run)
{
    const size_t n = orgX * orgY;
    FA_REQUIRE(nPoints == 0 || (px != nullptr && py != nullptr), "NULL argument");
    FA_REQUIRE(n == 0 || (lon != nullptr && lat != nullptr), "NULL argument");
    (void)current_device_checked();
    ScopedStream stream;
    DeviceArray<double> d_q(2 * nPoints), d_src(2 * n);
    if (nPoints) {
        host_to_device(d_q.get(), px, nPoints * sizeof(double), stream.get());
        host_to_device(d_q.get() + nPoints, py, nPoints * sizeof(double), stream.get());
    }
    if (n) {
        host_to_device(d_src.get(), lon, n * sizeof(double), stream.get());
        host_to_device(d_src.get() + n, lat, n * sizeof(double), stream.get());
    }
    run(d_q.get(), d_q.get() + nPoints, d_src.get(), d_src.get() + n, stream.get());
    if (nPoints) {
        device_to_host(px, d_q.get(), nPoints * sizeof(double), stream.get());
        device_to_host(py, d_q.get() + nPoints, nPoints * sizeof(double), stream.get());
    }
    stream.sync();
}
}  // namespace

extern "C" {

int fimex_amd_coord_nearest_host(double* px, double* py, size_t nPoints, const double* lon, const double* lat, size_t orgX, size_t orgY)
{
    return c_guard([&] {
        coord_search_host(px, py, nPoints, lon, lat, orgX, orgY, [&](double* qx, double* qy, const double* dlon, const double* dlat, hipStream_t st) {
            launch_coord_nearest(qx, qy, nPoints, dlon, dlat, orgX, orgY, st);
        });
    });
}

int fimex_amd_coord_nearest_device(double* d_px, double* d_py, size_t nPoints, const double* d_lon, const double* d_lat, size_t orgX, size_t orgY,
                                   void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(nPoints == 0 || (d_px != nullptr && d_py != nullptr && d_lon != nullptr && d_lat != nullptr), "NULL device buffer");
        (void)current_device_checked();
        launch_coord_nearest(d_px, d_py, nPoints, d_lon, d_lat, orgX, orgY, as_stream(stream));
    });
}

int fimex_amd_coord_kdtree_host(double maxDist, double* px, double* py, size_t nPoints, const double* lon, const double* lat, size_t orgX, size_t orgY)
{
    return c_guard([&] {
        coord_search_host(px, py, nPoints, lon, lat, orgX, orgY, [&](double* qx, double* qy, const double* dlon, const double* dlat, hipStream_t st) {
            launch_coord_kdtree(maxDist, qx, qy, nPoints, dlon, dlat, orgX, orgY, st);
        });
    });
}

int fimex_amd_coord_kdtree_device(double maxDist, double* d_px, double* d_py, size_t nPoints, const double* d_lon, const double* d_lat, size_t orgX,
                                  size_t orgY, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(nPoints == 0 || (d_px != nullptr && d_py != nullptr && d_lon != nullptr && d_lat != nullptr), "NULL device buffer");
        (void)current_device_checked();
        launch_coord_kdtree(maxDist, d_px, d_py, nPoints, d_lon, d_lat, orgX, orgY, as_stream(stream));
    });
}

int fimex_amd_grid_distance_host(const double* lon, const double* lat, size_t orgX, size_t orgY, double* maxGridDistance)
{
    return c_guard([&] {
        const size_t n = orgX * orgY;
        FA_REQUIRE(n > 0 && lon != nullptr && lat != nullptr && maxGridDistance != nullptr, "NULL or empty argument");
        (void)current_device_checked();
        ScopedStream stream;
        DeviceArray<double> d(2 * n);
        host_to_device(d.get(), lon, n * sizeof(double), stream.get());
        host_to_device(d.get() + n, lat, n * sizeof(double), stream.get());
        *maxGridDistance = grid_distance(d.get(), d.get() + n, orgX, orgY, stream.get());
    });
}

int fimex_amd_scan_sum_device(const float* d_values, size_t n, int mode, double average, int algo, double* sum, size_t* nUndefined, void* stream)
{
    return c_guard([&] {
        FA_REQUIRE(sum != nullptr && (d_values != nullptr || n == 0), "NULL argument");
        (void)current_device_checked();
        run_scan_sum(d_values, n, mode, average, algo, sum, nUndefined, as_stream(stream));
    });
}

}  // extern "C"
