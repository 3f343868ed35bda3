#include "CachedInterpolation.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <iostream>

namespace FimexAmd {

void checkAmd(int rc, const char* what)
{
    if (rc != FIMEX_AMD_OK) throw CDMException(std::string(what) + ": " + fimex_amd_last_error());
}

int mifi_string_to_interpolation_method(const char* mString)
{
    static const struct { const char* name; int method; } table[] = {
        {"bilinear", MIFI_INTERPOL_BILINEAR}, {"nearestneighbor", MIFI_INTERPOL_NEAREST_NEIGHBOR},
        {"bicubic", MIFI_INTERPOL_BICUBIC}, {"coord_nearestneighbor", MIFI_INTERPOL_COORD_NN},
        {"coord_kdtree", MIFI_INTERPOL_COORD_NN_KD}, {"forward_sum", MIFI_INTERPOL_FORWARD_SUM},
        {"forward_mean", MIFI_INTERPOL_FORWARD_MEAN}, {"forward_median", MIFI_INTERPOL_FORWARD_MEDIAN},
        {"forward_max", MIFI_INTERPOL_FORWARD_MAX}, {"forward_min", MIFI_INTERPOL_FORWARD_MIN},
        {"forward_undef_sum", MIFI_INTERPOL_FORWARD_UNDEF_SUM}, {"forward_undef_mean", MIFI_INTERPOL_FORWARD_UNDEF_MEAN},
        {"forward_undef_median", MIFI_INTERPOL_FORWARD_UNDEF_MEDIAN}, {"forward_undef_max", MIFI_INTERPOL_FORWARD_UNDEF_MAX},
        {"forward_undef_min", MIFI_INTERPOL_FORWARD_MIN},  // as the reference: src/interpolation.c:97-98
    };
    for (const auto& e : table)
        if (std::strcmp(e.name, mString) == 0) return e.method;
    return MIFI_INTERPOL_UNKNOWN;
}

// ------------------------------------------------------------------------------------------ PlanHolder
PlanHolder::~PlanHolder() { reset(); }

void PlanHolder::reset()
{
    if (plan_) fimex_amd_regrid_plan_destroy(plan_);
    plan_ = nullptr;
}

void PlanHolder::create(int funcType, const std::vector<double>& px, const std::vector<double>& py, size_t inX, size_t inY,
                        size_t outX, size_t outY)
{
    reset();
    if (px.size() != py.size()) throw CDMException("pointsOnXAxis and pointsOnYAxis differ in size");
    // an unknown funcType fails here with "unknown interpolation function", like CachedInterpolation.cc:114
    checkAmd(fimex_amd_regrid_plan_create(funcType, px.data(), py.data(), px.size(), inX, inY, outX, outY, &plan_),
             "creating regrid plan");
}

shared_array<float> PlanHolder::apply(const float* inData, size_t size, size_t& newSize) const
{
    if (!plan_) throw CDMException("regrid plan not initialised");
    checkAmd(fimex_amd_regrid_apply_host(plan_, inData, size, nullptr, 0, &newSize), "error during interpolation");
    shared_array<float> out(new float[newSize ? newSize : 1]);  // owned by the caller, CachedInterpolation.cc:123
    checkAmd(fimex_amd_regrid_apply_host(plan_, inData, size, out.get(), newSize, &newSize), "error during interpolation");
    return out;
}

// ---------------------------------------------------------------------------------- CachedInterpolation
CachedInterpolation::CachedInterpolation(const std::string& xDimName, const std::string& yDimName, int funcType,
                                         const std::vector<double>& pointsOnXAxis, const std::vector<double>& pointsOnYAxis,
                                         size_t inX, size_t inY, size_t outX, size_t outY)
    : CachedInterpolationInterface(xDimName, yDimName), pointsOnXAxis(pointsOnXAxis), pointsOnYAxis(pointsOnYAxis),
      funcType(funcType), inX(inX), inY(inY), outX(outX), outY(outY)
{
    switch (funcType) {  // src/CachedInterpolation.cc:107-115
    case MIFI_INTERPOL_BILINEAR: case MIFI_INTERPOL_BICUBIC: case MIFI_INTERPOL_NEAREST_NEIGHBOR:
    case MIFI_INTERPOL_COORD_NN: case MIFI_INTERPOL_COORD_NN_KD: break;
    default: throw CDMException("unknown interpolation function: " + std::to_string(funcType));
    }
    plan_.create(funcType, this->pointsOnXAxis, this->pointsOnYAxis, inX, inY, outX, outY);
}

shared_array<float> CachedInterpolation::interpolateValues(shared_array<float> inData, size_t size, size_t& newSize) const
{
    return plan_.apply(inData.get(), size, newSize);
}

namespace {
// src/CachedInterpolation.cc:149-157
long long clampLL(long long low, double dvalue, long long high)
{
    const long long value = static_cast<long long>(dvalue);
    if (value < low) return low;
    if (value < high) return value;
    return high;
}
}  // namespace

void CachedInterpolation::createReducedDomain(std::string xDimName, std::string yDimName)
{
    if (reducedDomain_) return;  // don't set twice
    if (pointsOnXAxis.empty()) return;
    const double pMinX = *std::min_element(pointsOnXAxis.begin(), pointsOnXAxis.end());
    const double pMinY = *std::min_element(pointsOnYAxis.begin(), pointsOnYAxis.end());
    const double pMaxX = *std::max_element(pointsOnXAxis.begin(), pointsOnXAxis.end());
    const double pMaxY = *std::max_element(pointsOnYAxis.begin(), pointsOnYAxis.end());
    const long long EXTEND = 2;  // two cells for bicubic
    const long long minX = clampLL(0, std::floor(pMinX) - EXTEND, (long long)inX - 1);
    const long long minY = clampLL(0, std::floor(pMinY) - EXTEND, (long long)inY - 1);
    const long long maxX = clampLL(0, std::ceil(pMaxX) + EXTEND, (long long)inX - 1);
    const long long maxY = clampLL(0, std::ceil(pMaxY) + EXTEND, (long long)inY - 1);
    if ((maxX - minX) < 1 || (maxY - minY) < 1) return;
    for (size_t xy = 0; xy < pointsOnXAxis.size(); ++xy) {
        pointsOnXAxis[xy] -= minX;
        pointsOnYAxis[xy] -= minY;
    }
    auto rid = std::make_shared<ReducedInterpolationDomain>();
    rid->xDim = xDimName;
    rid->yDim = yDimName;
    rid->xMin = (size_t)minX;
    rid->yMin = (size_t)minY;
    rid->xOrg = inX;
    rid->yOrg = inY;
    reducedDomain_ = rid;
    inX = (size_t)(maxX - minX + 1);
    inY = (size_t)(maxY - minY + 1);
    plan_.create(funcType, pointsOnXAxis, pointsOnYAxis, inX, inY, outX, outY);  // the plan follows the cropped grid
}

// ---------------------------------------------------------------------------- CachedForwardInterpolation
CachedForwardInterpolation::CachedForwardInterpolation(const std::string& xDimName, const std::string& yDimName, int funcType,
                                                       const std::vector<double>& pOnX, const std::vector<double>& pOnY,
                                                       size_t inX, size_t inY, size_t outX, size_t outY)
    : CachedInterpolationInterface(xDimName, yDimName), inX(inX), inY(inY), outX(outX), outY(outY)
{
    if (funcType < MIFI_INTERPOL_FORWARD_SUM || funcType > MIFI_INTERPOL_FORWARD_UNDEF_MIN)  // CachedForwardInterpolation.cc:88
        throw CDMException("unknown forward interpolation method: " + std::to_string(funcType));
    plan_.create(funcType, pOnX, pOnY, inX, inY, outX, outY);
}

shared_array<float> CachedForwardInterpolation::interpolateValues(shared_array<float> inData, size_t size, size_t& newSize) const
{
    return plan_.apply(inData.get(), size, newSize);
}

// ------------------------------------------------------------------------------ CachedVectorReprojection
CachedVectorReprojection::CachedVectorReprojection(int method, shared_array<double> matrix, int ox, int oy)
    : method(method), matrix(matrix), ox((size_t)ox), oy((size_t)oy)
{
    if (this->ox != 0 && this->oy != 0 && matrix)
        checkAmd(fimex_amd_vector_plan_create(matrix.get(), this->ox, this->oy, &plan_), "creating vector reprojection");
}

CachedVectorReprojection::~CachedVectorReprojection()
{
    if (plan_) fimex_amd_vector_plan_destroy(plan_);
}

void CachedVectorReprojection::reprojectValues(shared_array<float>& uValues, shared_array<float>& vValues, size_t size) const
{
    if (ox == 0 || oy == 0 || !matrix) {
        std::cerr << "WARN fimex.CachedVectorReprojection: not initialized, using identity" << std::endl;
        return;
    }
    if (fimex_amd_vector_reproject_values_host(plan_, uValues.get(), vValues.get(), size) != FIMEX_AMD_OK)
        throw CDMException(std::string("Error during reprojection of vector-values: ") + fimex_amd_last_error());
}

void CachedVectorReprojection::reprojectDirectionValues(shared_array<float>& angles, size_t size) const
{
    if (ox == 0 || oy == 0 || !matrix) {
        std::cerr << "WARN fimex.CachedVectorReprojection: not initialized, using identity" << std::endl;
        return;
    }
    if (fimex_amd_vector_reproject_direction_host(plan_, angles.get(), size) != FIMEX_AMD_OK)
        throw CDMException(std::string("Error during reprojection of vector-direction-values: ") + fimex_amd_last_error());
}

// ----------------------------------------------------------------------------------- 2-D fill processes
void InterpolatorFill2d::applyBatch(float* array, size_t nx, size_t ny, size_t nz)
{
    checkAmd(fimex_amd_fill2d_host(nx, ny, nz, array, relaxCrit_, corrEff_, maxLoop_, nullptr), "fill2d");
}

void InterpolatorCreepFill2d::applyBatch(float* array, size_t nx, size_t ny, size_t nz)
{
    checkAmd(fimex_amd_creepfill2d_host(nx, ny, nz, array, repeat_, setWeight_, nullptr), "creepfill2d");
}

void InterpolatorCreepFillVal2d::applyBatch(float* array, size_t nx, size_t ny, size_t nz)
{
    checkAmd(fimex_amd_creepfillval2d_host(nx, ny, nz, array, defVal_, repeat_, setWeight_, nullptr), "creepfillval2d");
}

bool InterpolatorFill2d::describe(fimex_amd_process2d& out) const
{
    out = fimex_amd_process2d{};
    out.kind = FIMEX_AMD_PROCESS_FILL2D;
    out.relaxCrit = relaxCrit_;
    out.corrEff = corrEff_;
    out.maxLoop = maxLoop_;
    return true;
}

bool InterpolatorCreepFill2d::describe(fimex_amd_process2d& out) const
{
    out = fimex_amd_process2d{};
    out.kind = FIMEX_AMD_PROCESS_CREEPFILL2D;
    out.repeat = repeat_;
    out.setWeight = setWeight_;
    return true;
}

bool InterpolatorCreepFillVal2d::describe(fimex_amd_process2d& out) const
{
    out = fimex_amd_process2d{};
    out.kind = FIMEX_AMD_PROCESS_CREEPFILLVAL2D;
    out.repeat = repeat_;
    out.setWeight = setWeight_;
    out.defaultVal = defVal_;
    return true;
}

}  // namespace FimexAmd
