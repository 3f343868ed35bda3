// C++ host mirror of the reference's regrid plan objects, implemented over the C ABI of
// libfimex_amd.so (include/fimex_amd.h).  Same names, argument meaning and error behaviour as
//   include/fimex/CachedInterpolation.h:60-161, src/CachedForwardInterpolation.h:37-59,
//   include/fimex/CachedVectorReprojection.h:33-63, include/fimex/CDMInterpolator.h:49-88,
// with std:: types where the reference uses boost:: (boost is not available here).  A Fimex maintainer
// can lift these bodies into the MetNoFimex classes unchanged.
#pragma once

#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "fimex_amd.h"

namespace FimexAmd {

// MetNoFimex::CDMException stand-in
class CDMException : public std::runtime_error {
public:
    explicit CDMException(const std::string& msg) : std::runtime_error("CDMException: " + msg) {}
};

// boost::shared_array<T> stand-in
template <typename T>
using shared_array = std::shared_ptr<T[]>;

// values of enum mifi_interpol_method (include/fimex/mifi_constants.h:52-147)
enum {
    MIFI_INTERPOL_UNKNOWN = -1,
    MIFI_INTERPOL_NEAREST_NEIGHBOR = 0, MIFI_INTERPOL_BILINEAR, MIFI_INTERPOL_BICUBIC, MIFI_INTERPOL_COORD_NN,
    MIFI_INTERPOL_COORD_NN_KD, MIFI_INTERPOL_FORWARD_SUM, MIFI_INTERPOL_FORWARD_MEAN, MIFI_INTERPOL_FORWARD_MEDIAN,
    MIFI_INTERPOL_FORWARD_MAX, MIFI_INTERPOL_FORWARD_MIN, MIFI_INTERPOL_FORWARD_UNDEF_SUM, MIFI_INTERPOL_FORWARD_UNDEF_MEAN,
    MIFI_INTERPOL_FORWARD_UNDEF_MEDIAN, MIFI_INTERPOL_FORWARD_UNDEF_MAX, MIFI_INTERPOL_FORWARD_UNDEF_MIN
};
enum { MIFI_VECTOR_KEEP_SIZE = 0, MIFI_VECTOR_RESIZE = 1 };
enum { MIFI_PROJ_AXIS = 0, MIFI_LONGITUDE = 1, MIFI_LATITUDE = 2 };

// mifi_string_to_interpolation_method (src/interpolation.c:66-101), including its quirk:
// "forward_undef_min" maps to MIFI_INTERPOL_FORWARD_MIN (:97-98)
int mifi_string_to_interpolation_method(const char* mString);

// include/fimex/CachedInterpolation.h:41-56
struct ReducedInterpolationDomain {
    std::string xDim, yDim;
    size_t xMin = 0, xOrg = 0, yMin = 0, yOrg = 0;
};

// include/fimex/CachedInterpolation.h:60-99 (getInputDataSlice lives in CDMInterpolator here: it needs the reader)
class CachedInterpolationInterface {
public:
    CachedInterpolationInterface(std::string xDimName, std::string yDimName) : xDimName_(std::move(xDimName)), yDimName_(std::move(yDimName)) {}
    virtual ~CachedInterpolationInterface() {}
    virtual shared_array<float> interpolateValues(shared_array<float> inData, size_t size, size_t& newSize) const = 0;
    virtual size_t getInX() const = 0;
    virtual size_t getInY() const = 0;
    virtual size_t getOutX() const = 0;
    virtual size_t getOutY() const = 0;
    virtual std::shared_ptr<ReducedInterpolationDomain> reducedDomain() const { return std::shared_ptr<ReducedInterpolationDomain>(); }
    // the engine's plan behind this object, NULL for implementations that are not backed by libfimex_amd
    virtual const fimex_amd_regrid_plan* amdPlan() const { return nullptr; }
    const std::string& xDimName() const { return xDimName_; }
    const std::string& yDimName() const { return yDimName_; }

private:
    std::string xDimName_, yDimName_;
};

// shared implementation: owns the fimex_amd_regrid_plan
class PlanHolder {
public:
    PlanHolder() = default;
    ~PlanHolder();
    PlanHolder(const PlanHolder&) = delete;
    PlanHolder& operator=(const PlanHolder&) = delete;
    void create(int funcType, const std::vector<double>& px, const std::vector<double>& py, size_t inX, size_t inY, size_t outX, size_t outY);
    void reset();
    shared_array<float> apply(const float* inData, size_t size, size_t& newSize) const;
    fimex_amd_regrid_plan* get() const { return plan_; }

private:
    fimex_amd_regrid_plan* plan_ = nullptr;
};

// include/fimex/CachedInterpolation.h:105-161, src/CachedInterpolation.cc:93-200
class CachedInterpolation : public CachedInterpolationInterface {
public:
    CachedInterpolation(const std::string& xDimName, const std::string& yDimName, int funcType,
                        const std::vector<double>& pointsOnXAxis, const std::vector<double>& pointsOnYAxis,
                        size_t inX, size_t inY, size_t outX, size_t outY);
    shared_array<float> interpolateValues(shared_array<float> inData, size_t size, size_t& newSize) const override;
    size_t getInX() const override { return inX; }
    size_t getInY() const override { return inY; }
    size_t getOutX() const override { return outX; }
    size_t getOutY() const override { return outY; }
    std::shared_ptr<ReducedInterpolationDomain> reducedDomain() const override { return reducedDomain_; }
    // src/CachedInterpolation.cc:159-200; run immediately after construction
    void createReducedDomain(std::string xDimName, std::string yDimName);
    const PlanHolder& plan() const { return plan_; }
    const fimex_amd_regrid_plan* amdPlan() const override { return plan_.get(); }

private:
    std::vector<double> pointsOnXAxis, pointsOnYAxis;
    int funcType;
    size_t inX, inY, outX, outY;
    std::shared_ptr<ReducedInterpolationDomain> reducedDomain_;
    PlanHolder plan_;
};

// src/CachedForwardInterpolation.h:37-59, src/CachedForwardInterpolation.cc:62-131
class CachedForwardInterpolation : public CachedInterpolationInterface {
public:
    CachedForwardInterpolation(const std::string& xDimName, const std::string& yDimName, int funcType,
                               const std::vector<double>& pointsOnXAxis, const std::vector<double>& pointsOnYAxis,
                               size_t inX, size_t inY, size_t outX, size_t outY);
    shared_array<float> interpolateValues(shared_array<float> inData, size_t size, size_t& newSize) const override;
    size_t getInX() const override { return inX; }
    size_t getInY() const override { return inY; }
    size_t getOutX() const override { return outX; }
    size_t getOutY() const override { return outY; }
    const fimex_amd_regrid_plan* amdPlan() const override { return plan_.get(); }

private:
    size_t inX, inY, outX, outY;
    PlanHolder plan_;
};

// include/fimex/CachedVectorReprojection.h:33-63, src/CachedVectorReprojection.cc:35-55
class CachedVectorReprojection {
public:
    CachedVectorReprojection() {}
    CachedVectorReprojection(int method, shared_array<double> matrix, int ox, int oy);
    ~CachedVectorReprojection();
    CachedVectorReprojection(const CachedVectorReprojection&) = delete;
    CachedVectorReprojection& operator=(const CachedVectorReprojection&) = delete;
    // in place; an uninitialised object is the identity (reference: WARN + return, :37-40)
    void reprojectValues(shared_array<float>& uValues, shared_array<float>& vValues, size_t size) const;
    void reprojectDirectionValues(shared_array<float>& angles, size_t size) const;
    size_t getXSize() const { return ox; }
    size_t getYSize() const { return oy; }
    const fimex_amd_vector_plan* handle() const { return plan_; }  // NULL for the uninitialised (identity) object

private:
    int method = MIFI_VECTOR_KEEP_SIZE;
    shared_array<double> matrix;
    size_t ox = 0, oy = 0;
    fimex_amd_vector_plan* plan_ = nullptr;
};

// include/fimex/CDMInterpolator.h:49-88.  operator() keeps the reference's per-slice signature; applyBatch is what
// processArray_ (src/CDMInterpolator.cc:136-159) uses here: all z slices of a call in one launch.
class InterpolatorProcess2d {
public:
    virtual void operator()(float* array, size_t nx, size_t ny) { applyBatch(array, nx, ny, 1); }
    virtual void applyBatch(float* array, size_t nx, size_t ny, size_t nz) = 0;
    // the three built-in processes describe themselves so that getDataSlice can keep the data on the GPU between
    // the steps (fimex_amd_regrid_slice_host); user-defined processes return false and run on the host array
    virtual bool describe(fimex_amd_process2d& out) const { (void)out; return false; }
    virtual ~InterpolatorProcess2d() {}
};

class InterpolatorFill2d : public InterpolatorProcess2d {
public:
    InterpolatorFill2d(float relaxCrit, float corrEff, size_t maxLoop) : relaxCrit_(relaxCrit), corrEff_(corrEff), maxLoop_(maxLoop) {}
    void applyBatch(float* array, size_t nx, size_t ny, size_t nz) override;
    bool describe(fimex_amd_process2d& out) const override;

private:
    float relaxCrit_, corrEff_;
    size_t maxLoop_;
};

class InterpolatorCreepFill2d : public InterpolatorProcess2d {
public:
    InterpolatorCreepFill2d(unsigned short repeat, char setWeight) : repeat_(repeat), setWeight_(setWeight) {}
    void applyBatch(float* array, size_t nx, size_t ny, size_t nz) override;
    bool describe(fimex_amd_process2d& out) const override;

private:
    unsigned short repeat_;
    char setWeight_;
};

class InterpolatorCreepFillVal2d : public InterpolatorProcess2d {
public:
    InterpolatorCreepFillVal2d(unsigned short repeat, char setWeight, float defaultValue) : repeat_(repeat), setWeight_(setWeight), defVal_(defaultValue) {}
    void applyBatch(float* array, size_t nx, size_t ny, size_t nz) override;
    bool describe(fimex_amd_process2d& out) const override;

private:
    unsigned short repeat_;
    char setWeight_;
    float defVal_;
};

// turns FIMEX_AMD_ERROR into a CDMException carrying fimex_amd_last_error()
void checkAmd(int rc, const char* what);

}  // namespace FimexAmd
