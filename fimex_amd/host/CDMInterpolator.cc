#include "CDMInterpolator.h"

#include <cmath>
#include <iostream>
#include <limits>

namespace FimexAmd {

namespace {
const double DEG_TO_RAD = .0174532925199432958;  // proj_api.h
const std::string LAT_LON_PROJSTR = "+proj=latlong +R=6371000";  // sphere stand-in for MIFI_WGS84_LATLON_PROJ4 (no datum shift either way)

bool isDegreeUnit(const std::string& unit) { return unit.find("degree") != std::string::npos; }  // boost::regex(".*degree.*")

void toRad(std::vector<double>& v)
{
    for (double& d : v) d *= DEG_TO_RAD;
}

void points2position(std::vector<double>& points, const std::vector<double>& axis, int axisType)
{
    checkAmd(fimex_amd_points2position_host(points.data(), points.size(), axis.data(), (int)axis.size(), axisType),
             "mifi_points2position");
}
}  // namespace

CDMInterpolator::CDMInterpolator(std::shared_ptr<GridReader> dataReader) : dataReader_(dataReader) {}

void CDMInterpolator::changeProjection(int method, const std::string& proj_input, const std::vector<double>& out_x_axis,
                                       const std::vector<double>& out_y_axis, const std::string& out_x_axis_unit,
                                       const std::string& out_y_axis_unit)
{
    const bool xDeg = isDegreeUnit(out_x_axis_unit), yDeg = isDegreeUnit(out_y_axis_unit);
    switch (method) {  // src/CDMInterpolator.cc:429-449
    case MIFI_INTERPOL_NEAREST_NEIGHBOR:
    case MIFI_INTERPOL_BILINEAR:
    case MIFI_INTERPOL_BICUBIC:
        changeProjectionByProjectionParameters(method, proj_input, out_x_axis, out_y_axis, xDeg, yDeg);
        break;
    case MIFI_INTERPOL_COORD_NN:
    case MIFI_INTERPOL_COORD_NN_KD:
        // the coordinate search that builds these plans (src/CDMInterpolator.cc:992-1220) is not part of this round;
        // plans built elsewhere apply through CachedInterpolation as in the reference
        throw CDMException("coord_nearestneighbor / coord_kdtree plan construction is not implemented");
    case MIFI_INTERPOL_FORWARD_SUM: case MIFI_INTERPOL_FORWARD_MEAN: case MIFI_INTERPOL_FORWARD_MEDIAN:
    case MIFI_INTERPOL_FORWARD_MAX: case MIFI_INTERPOL_FORWARD_MIN: case MIFI_INTERPOL_FORWARD_UNDEF_SUM:
    case MIFI_INTERPOL_FORWARD_UNDEF_MEAN: case MIFI_INTERPOL_FORWARD_UNDEF_MEDIAN: case MIFI_INTERPOL_FORWARD_UNDEF_MAX:
    case MIFI_INTERPOL_FORWARD_UNDEF_MIN:
        changeProjectionByForwardInterpolation(method, proj_input, out_x_axis, out_y_axis, xDeg, yDeg);
        break;
    default:
        throw CDMException("unknown projection method: " + std::to_string(method));
    }
}

// src/CDMInterpolator.cc:1422-1503
void CDMInterpolator::changeProjectionByProjectionParameters(int method, const std::string& proj_input, std::vector<double> outXAxis,
                                                             std::vector<double> outYAxis, bool xDegree, bool yDegree)
{
    const Projection outProj(proj_input), orgProj(dataReader_->projString());
    const std::vector<double> outXAxisOrg = outXAxis, outYAxisOrg = outYAxis;
    int outXAxisType = MIFI_PROJ_AXIS, outYAxisType = MIFI_PROJ_AXIS;
    if (xDegree) { toRad(outXAxis); outXAxisType = MIFI_LONGITUDE; }
    if (yDegree) { toRad(outYAxis); outYAxisType = MIFI_LATITUDE; }
    (void)outXAxisType; (void)outYAxisType;

    // positions of the new grid's cells in the original projection (:1458), then on the original axes (:1475-1476)
    projectAxes(outProj, orgProj, outXAxis, outYAxis, pointsOnXAxis_, pointsOnYAxis_);
    std::vector<double> orgX = dataReader_->xAxis(), orgY = dataReader_->yAxis();
    int miupXAxis = MIFI_PROJ_AXIS, miupYAxis = MIFI_PROJ_AXIS;
    if (orgProj.isDegree()) {
        miupXAxis = MIFI_LONGITUDE;
        miupYAxis = MIFI_LATITUDE;
        toRad(orgX);
        toRad(orgY);
    }
    points2position(pointsOnXAxis_, orgX, miupXAxis);
    points2position(pointsOnYAxis_, orgY, miupYAxis);

    auto ci = std::make_shared<CachedInterpolation>(dataReader_->xDimName(), dataReader_->yDimName(), method, pointsOnXAxis_,
                                                    pointsOnYAxis_, orgX.size(), orgY.size(), outXAxis.size(), outYAxis.size());
    ci->createReducedDomain(dataReader_->xDimName(), dataReader_->yDimName());  // :1484
    cachedInterpolation_ = ci;

    // rotation of x/y vector components (:1491-1500); the reference only builds it when the file holds such vectors
    vectorReprojectMatrix(orgProj, outProj, outXAxis, outYAxis, matrix_);
    shared_array<double> m(new double[matrix_.size()]);
    std::copy(matrix_.begin(), matrix_.end(), m.get());
    cachedVectorReprojection_ =
        std::make_shared<CachedVectorReprojection>(MIFI_VECTOR_KEEP_SIZE, m, (int)outXAxis.size(), (int)outYAxis.size());
}

// src/CDMInterpolator.cc:1242-1333
void CDMInterpolator::changeProjectionByForwardInterpolation(int method, const std::string& proj_input, std::vector<double> outXAxis,
                                                             std::vector<double> outYAxis, bool xDegree, bool yDegree)
{
    std::vector<double> lonVals, latVals;
    std::vector<double> orgX = dataReader_->xAxis(), orgY = dataReader_->yAxis();
    if (!dataReader_->lonLat(lonVals, latVals)) {
        // geographic grid: the matrix of its two axes (lonLatVals2Matrix, :1285-1289); otherwise project the grid
        const Projection orgProj(dataReader_->projString());
        std::vector<double> ax = orgX, ay = orgY;
        if (orgProj.isDegree()) { toRad(ax); toRad(ay); }
        const Projection geo(LAT_LON_PROJSTR);
        projectAxes(orgProj, geo, ax, ay, lonVals, latVals);  // radians
    } else {
        toRad(lonVals);
        toRad(latVals);
    }
    int miupXAxis = MIFI_PROJ_AXIS, miupYAxis = MIFI_PROJ_AXIS;
    if (xDegree) { toRad(outXAxis); miupXAxis = MIFI_LONGITUDE; }
    if (yDegree) { toRad(outYAxis); miupYAxis = MIFI_LATITUDE; }
    // all input points in output coordinates (:1311), then cell positions on the output axes (:1316-1317)
    const Projection geo(LAT_LON_PROJSTR), outProj(proj_input);
    transform(geo, outProj, lonVals.data(), latVals.data(), lonVals.size());
    points2position(lonVals, outXAxis, miupXAxis);
    points2position(latVals, outYAxis, miupYAxis);
    pointsOnXAxis_ = lonVals;
    pointsOnYAxis_ = latVals;
    cachedInterpolation_ = std::make_shared<CachedForwardInterpolation>(dataReader_->xDimName(), dataReader_->yDimName(), method,
                                                                       pointsOnXAxis_, pointsOnYAxis_, orgX.size(), orgY.size(),
                                                                       outXAxis.size(), outYAxis.size());
    cachedVectorReprojection_.reset();  // "vector data found, but not possible to interpolate with forward-interpolation" (:1331)
    matrix_.clear();
}

shared_array<float> CDMInterpolator::readInput(const std::string& varName, size_t unLimDimPos, size_t& size) const
{
    const auto rd = cachedInterpolation_->reducedDomain();
    const size_t x0 = rd ? rd->xMin : 0, y0 = rd ? rd->yMin : 0;
    return dataReader_->getDataSlice(varName, unLimDimPos, x0, cachedInterpolation_->getInX(), y0, cachedInterpolation_->getInY(), size);
}

// src/CDMInterpolator.cc:136-159 -- all z slices of the array in one batch
void CDMInterpolator::processArray(const std::vector<std::shared_ptr<InterpolatorProcess2d>>& processes, float* array, size_t size,
                                   size_t nx, size_t ny) const
{
    if (processes.empty()) return;
    const size_t nz = size / (nx * ny);
    for (const auto& p : processes) p->applyBatch(array, nx, ny, nz);
}

// src/CDMInterpolator.cc:235-287
shared_array<float> CDMInterpolator::getDataSlice(const std::string& varName, size_t unLimDimPos, size_t& size)
{
    if (!dataReader_->hasVariable(varName)) throw CDMException("variable not found: " + varName);
    if (!cachedInterpolation_) throw CDMException("no cached interpolation for " + varName);  // :247-249
    const VariableInfo var = dataReader_->variable(varName);
    size_t inSize = 0;
    shared_array<float> data = readInput(varName, unLimDimPos, inSize);
    size = 0;
    if (inSize == 0) return data;  // :252-253
    const float nan = std::numeric_limits<float>::quiet_NaN();
    const float badValue = var.hasFillValue ? (float)var.fillValue : nan;

    const bool rotate = var.spatialVector &&
                        !(var.direction.find("x") == std::string::npos && var.direction.find("y") == std::string::npos);
    shared_array<float> counterpart;
    float badCounterpart = nan;
    bool isX = true;
    if (rotate) {
        if (cachedVectorReprojection_) {
            size_t cSize = 0;
            counterpart = readInput(var.counterpart, unLimDimPos, cSize);  // :269
            const VariableInfo cv = dataReader_->variable(var.counterpart);
            badCounterpart = cv.hasFillValue ? (float)cv.fillValue : nan;
            if (var.direction.find("x") != std::string::npos) isX = true;
            else if (var.direction.find("y") != std::string::npos) isX = false;
            else throw CDMException("could not find x,y direction for vector: " + varName + ", direction: " + var.direction);
        } else {
            std::cerr << "WARN fimex.CDMInterpolator: Cannot reproject vector " << var.name << std::endl;  // :280
        }
    }

    std::vector<fimex_amd_process2d> pre(preprocesses_.size()), post(postprocesses_.size());
    bool builtin = true;
    for (size_t i = 0; i < pre.size(); ++i) builtin = preprocesses_[i]->describe(pre[i]) && builtin;
    for (size_t i = 0; i < post.size(); ++i) builtin = postprocesses_[i]->describe(post[i]) && builtin;
    const auto* plan = dynamic_cast<const CachedInterpolation*>(cachedInterpolation_.get());
    if (builtin && plan != nullptr) {
        // the whole sequence :255-285 in one call, the slices resident in HBM between the steps
        const fimex_amd_regrid_plan* h = plan->plan().get();
        const fimex_amd_vector_plan* vec = counterpart ? cachedVectorReprojection_->handle() : nullptr;
        const float* cp = vec ? counterpart.get() : nullptr;  // uninitialised reprojection = identity (:37-40)
        checkAmd(fimex_amd_regrid_slice_host(h, data.get(), inSize, badValue, pre.data(), pre.size(), cp, badCounterpart, vec, isX,
                                             post.data(), post.size(), nullptr, 0, &size),
                 "interpolateValues");
        shared_array<float> out(new float[size ? size : 1]);
        checkAmd(fimex_amd_regrid_slice_host(h, data.get(), inSize, badValue, pre.data(), pre.size(), cp, badCounterpart, vec, isX,
                                             post.data(), post.size(), out.get(), size, &size),
                 "interpolateValues");
        return out;
    }

    // general path (forward plans, user-defined processes): the reference's sequence call by call
    if (var.hasFillValue)
        for (size_t i = 0; i < inSize; ++i) if (data[i] == badValue) data[i] = nan;  // mifi_bad2nanf
    processArray(preprocesses_, data.get(), inSize, cachedInterpolation_->getInX(), cachedInterpolation_->getInY());
    shared_array<float> iArray = cachedInterpolation_->interpolateValues(data, inSize, size);
    if (counterpart) {
        if (!std::isnan(badCounterpart))
            for (size_t i = 0; i < inSize; ++i) if (counterpart[i] == badCounterpart) counterpart[i] = nan;
        processArray(preprocesses_, counterpart.get(), inSize, cachedInterpolation_->getInX(), cachedInterpolation_->getInY());
        size_t cs = 0;
        shared_array<float> cArray = cachedInterpolation_->interpolateValues(counterpart, inSize, cs);
        if (isX) cachedVectorReprojection_->reprojectValues(iArray, cArray, size);
        else cachedVectorReprojection_->reprojectValues(cArray, iArray, size);
    }
    processArray(postprocesses_, iArray.get(), size, cachedInterpolation_->getOutX(), cachedInterpolation_->getOutY());
    if (var.hasFillValue)
        for (size_t i = 0; i < size; ++i) if (std::isnan(iArray[i])) iArray[i] = badValue;  // mifi_nanf2bad
    return iArray;
}

}  // namespace FimexAmd
