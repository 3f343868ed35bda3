#include "CDMInterpolator.h"

#include <algorithm>
#include <cmath>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>

namespace FimexAmd {

namespace {
const double DEG_TO_RAD = .0174532925199432958;  // proj_api.h
const std::string LAT_LON_PROJSTR = "+proj=latlong +datum=WGS84 +towgs84=0,0,0 +no_defs";  // MIFI_WGS84_LATLON_PROJ4, mifi_constants.h

bool isDegreeUnit(const std::string& unit) { return unit.find("degree") != std::string::npos; }  // boost::regex(".*degree.*")

void toRad(std::vector<double>& v)
{
    for (double& d : v) d *= DEG_TO_RAD;
}

// mifi_project_axes / mifi_project_values / mifi_get_vector_reproject_matrix on the GPU
void projectAxesAmd(const std::string& from, const std::string& to, const std::vector<double>& xAxis, const std::vector<double>& yAxis,
                    std::vector<double>& outX, std::vector<double>& outY)
{
    outX.resize(xAxis.size() * yAxis.size());
    outY.resize(outX.size());
    checkAmd(fimex_amd_project_axes_host(from.c_str(), to.c_str(), xAxis.data(), yAxis.data(), xAxis.size(), yAxis.size(), outX.data(), outY.data()),
             ("unable to project axes from " + from + " to " + to).c_str());
}

bool isDegreeProjection(const std::string& proj)
{
    const int r = fimex_amd_projection_is_degree(proj.c_str());
    if (r < 0) throw CDMException(std::string("projection: ") + fimex_amd_last_error());
    return r != 0;
}

const double RAD_TO_DEG = 57.29577951308232;  // proj_api.h

// Projection::getProj4EarthString: the figure-of-the-earth parameters of a proj4 string
std::string proj4EarthString(const std::string& proj)
{
    std::istringstream in(proj);
    std::string tok, earth;
    while (in >> tok) {
        const std::string key = tok.substr(0, tok.find('='));
        for (const char* k : {"+a", "+b", "+e", "+es", "+f", "+rf", "+R", "+ellps", "+datum", "+towgs84"})
            if (key == k) earth += (earth.empty() ? "" : " ") + tok;
    }
    return earth;
}

// Projection::convertFromLonLat / convertToLonLat (src/coordSys/Projection.cc:74-140): degrees in, the grid's own unit out
// (degrees for geographic and rotated grids, metres otherwise) and back
void convertFromLonLat(const std::string& proj, std::vector<double>& x, std::vector<double>& y)
{
    if (x.empty()) return;
    toRad(x);
    toRad(y);
    const std::string fromProj = "+proj=latlong " + proj4EarthString(proj);
    if (fromProj == proj) return;  // as the reference (:130): the values stay in radians
    checkAmd(fimex_amd_project_values_host(fromProj.c_str(), proj.c_str(), x.data(), y.data(), x.size()),
             ("convertFromLonLat: unable to convert from '" + fromProj + "' to '" + proj + "'").c_str());
    if (isDegreeProjection(proj)) {
        for (double& v : x) v *= RAD_TO_DEG;
        for (double& v : y) v *= RAD_TO_DEG;
    }
}

void convertToLonLat(const std::string& proj, std::vector<double>& x, std::vector<double>& y)
{
    if (x.empty()) return;
    if (isDegreeProjection(proj)) {
        toRad(x);
        toRad(y);
    }
    const std::string toProj = "+proj=latlong " + proj4EarthString(proj);
    if (proj == toProj) return;
    checkAmd(fimex_amd_project_values_host(proj.c_str(), toProj.c_str(), x.data(), y.data(), x.size()),
             ("convertToLonLat: unable to convert from '" + proj + "' to '" + toProj + "'").c_str());
    for (double& v : x) v *= RAD_TO_DEG;
    for (double& v : y) v *= RAD_TO_DEG;
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
        changeProjectionByCoordinates(method, proj_input, out_x_axis, out_y_axis, xDeg, yDeg);
        break;
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
    const std::string orgProjStr = dataReader_->projString();
    const std::vector<double> outXAxisOrg = outXAxis, outYAxisOrg = outYAxis;
    int outXAxisType = MIFI_PROJ_AXIS, outYAxisType = MIFI_PROJ_AXIS;
    if (xDegree) { toRad(outXAxis); outXAxisType = MIFI_LONGITUDE; }
    if (yDegree) { toRad(outYAxis); outYAxisType = MIFI_LATITUDE; }

    // positions of the new grid's cells in the original projection (:1458), then on the original axes (:1475-1476)
    projectAxesAmd(proj_input, orgProjStr, outXAxis, outYAxis, pointsOnXAxis_, pointsOnYAxis_);
    std::vector<double> orgX = dataReader_->xAxis(), orgY = dataReader_->yAxis();
    int miupXAxis = MIFI_PROJ_AXIS, miupYAxis = MIFI_PROJ_AXIS;
    if (isDegreeProjection(orgProjStr)) {
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
    matrix_.resize(4 * outXAxis.size() * outYAxis.size());
    checkAmd(fimex_amd_get_vector_reproject_matrix_host(orgProjStr.c_str(), proj_input.c_str(), outXAxisOrg.data(), outYAxisOrg.data(),
                                                        outXAxisType, outYAxisType, outXAxis.size(), outYAxis.size(), matrix_.data()),
             "mifi_get_vector_reproject_matrix");  // :1497
    shared_array<double> m(new double[matrix_.size()]);
    std::copy(matrix_.begin(), matrix_.end(), m.get());
    cachedVectorReprojection_ =
        std::make_shared<CachedVectorReprojection>(MIFI_VECTOR_KEEP_SIZE, m, (int)outXAxis.size(), (int)outYAxis.size());
}

// src/CDMInterpolator.cc:460-510
void CDMInterpolator::changeProjection(int method, const std::vector<double>& lonVals, const std::vector<double>& latVals)
{
    if (lonVals.size() != latVals.size()) {  // :465-469: logged, nothing changes
        std::cerr << "changeProjection, number of longitude and latitude values differs: " << lonVals.size() << " != " << latVals.size() << std::endl;
        return;
    }
    switch (method) {
    case MIFI_INTERPOL_NEAREST_NEIGHBOR:
    case MIFI_INTERPOL_BILINEAR:
    case MIFI_INTERPOL_BICUBIC: {
        // the reference passes the values on as float Data (:475-481)
        std::vector<float> tmplLat(latVals.begin(), latVals.end()), tmplLon(lonVals.begin(), lonVals.end());
        changeProjectionByProjectionParametersToLatLonTemplate(method, LAT_LON_PROJSTR, lonVals.size(), 1, tmplLat, tmplLon);
        break;
    }
    case MIFI_INTERPOL_COORD_NN: case MIFI_INTERPOL_COORD_NN_KD: case MIFI_INTERPOL_FORWARD_SUM: case MIFI_INTERPOL_FORWARD_MEAN:
    case MIFI_INTERPOL_FORWARD_MEDIAN: case MIFI_INTERPOL_FORWARD_MAX: case MIFI_INTERPOL_FORWARD_MIN:
        throw CDMException("projection method: " + std::to_string(method) + ", not supported");
    default:
        throw CDMException("unknown projection method: " + std::to_string(method));
    }
}

// src/CDMInterpolator.cc:512-633
void CDMInterpolator::changeProjectionToCrossSections(int method, const std::vector<CrossSectionDefinition>& crossSections)
{
    const std::string proj = dataReader_->projString();
    const std::vector<double> xAxis = dataReader_->xAxis(), yAxis = dataReader_->yAxis();  // degrees or metres, as the grid has them
    if (xAxis.size() < 2 || yAxis.size() < 2) throw CDMException("x- or y-axis sizes < 2 elements, not possible to interpolate");
    const double dx = xAxis[1] - xAxis[0], dy = yAxis[1] - yAxis[0];
    if (dx == 0 || dy == 0) throw CDMException("cross-section calculation: dx or dy derived from first two elements == 0");

    std::vector<double> lonVals, latVals;
    std::vector<size_t> startPositions;
    std::vector<std::string> names;
    for (const CrossSectionDefinition& cs : crossSections) {
        if (cs.lonLatCoordinates.empty()) continue;
        names.push_back(cs.name);
        startPositions.push_back(lonVals.size());
        if (cs.lonLatCoordinates.size() == 1) {
            lonVals.push_back(cs.lonLatCoordinates[0].first);
            latVals.push_back(cs.lonLatCoordinates[0].second);
            continue;
        }
        for (size_t i = 1; i < cs.lonLatCoordinates.size(); ++i) {
            std::vector<double> xLon = {cs.lonLatCoordinates[i - 1].first, cs.lonLatCoordinates[i].first};
            std::vector<double> yLat = {cs.lonLatCoordinates[i - 1].second, cs.lonLatCoordinates[i].second};
            convertFromLonLat(proj, xLon, yLat);
            const double xLonD = xLon[1] - xLon[0], yLatD = yLat[1] - yLat[0];
            // number of grid points between two waypoints (:565)
            const size_t num = static_cast<size_t>(std::floor(std::max(std::fabs(xLonD / dx), std::fabs(yLatD / dy))));
            std::vector<double> xLonPart, yLatPart;
            if (i == 1) {  // the first waypoint belongs to the first leg only
                xLonPart.push_back(xLon[0]);
                yLatPart.push_back(yLat[0]);
            }
            for (size_t j = 1; j < num; ++j) {
                xLonPart.push_back(xLon[0] + j * xLonD / num);
                yLatPart.push_back(yLat[0] + j * yLatD / num);
            }
            xLonPart.push_back(xLon[1]);
            yLatPart.push_back(yLat[1]);
            convertToLonLat(proj, xLonPart, yLatPart);
            lonVals.insert(lonVals.end(), xLonPart.begin(), xLonPart.end());
            latVals.insert(latVals.end(), yLatPart.begin(), yLatPart.end());
        }
    }
    if (names.empty()) throw CDMException("no cross-section with coordinates");
    // vcross_name / vcross_bnds (:587-626)
    csNames_ = names;
    csBounds_.assign(2 * names.size(), 0);
    for (size_t i = 0; i + 1 < names.size(); ++i) {
        csBounds_[2 * i] = (int)startPositions[i];
        csBounds_[2 * i + 1] = (int)startPositions[i + 1] - 1;
    }
    csBounds_[2 * (names.size() - 1)] = (int)startPositions.back();
    csBounds_[2 * (names.size() - 1) + 1] = (int)lonVals.size() - 1;
    targetLon_ = lonVals;
    targetLat_ = latVals;
    changeProjection(method, lonVals, latVals);  // :632
}

// src/CDMInterpolator.cc:651-712: only the three backward methods (:697-708)
void CDMInterpolator::changeProjectionToTemplate(int method, const std::vector<float>& tmplLonVals, const std::vector<float>& tmplLatVals,
                                                 size_t outX, size_t outY)
{
    if (tmplLonVals.size() != outX * outY || tmplLatVals.size() != outX * outY)
        throw CDMException("template longitude / latitude do not have outX * outY values");
    switch (method) {
    case MIFI_INTERPOL_NEAREST_NEIGHBOR:
    case MIFI_INTERPOL_BILINEAR:
    case MIFI_INTERPOL_BICUBIC:
        changeProjectionByProjectionParametersToLatLonTemplate(method, LAT_LON_PROJSTR, outX, outY, tmplLatVals, tmplLonVals);
        break;
    default:
        throw CDMException("unknown projection method: " + std::to_string(method));
    }
}

// src/CDMInterpolator.cc:1706-1820
void CDMInterpolator::changeProjectionByProjectionParametersToLatLonTemplate(int method, const std::string& tmpl_proj_input, size_t outX,
                                                                             size_t outY, const std::vector<float>& tmplLatVals,
                                                                             const std::vector<float>& tmplLonVals)
{
    const std::string orgProjStr = dataReader_->projString();
    const size_t n = tmplLatVals.size();
    // template data is in degrees (:1761-1766)
    std::vector<double> latY(tmplLatVals.begin(), tmplLatVals.end()), lonX(tmplLonVals.begin(), tmplLonVals.end());
    toRad(latY);
    toRad(lonX);
    const std::vector<double> latRad = latY, lonRad = lonX;
    // template lat / lon expressed in the original projection (:1773), then on the original axes (:1792-1793)
    checkAmd(fimex_amd_project_values_host(tmpl_proj_input.c_str(), orgProjStr.c_str(), lonX.data(), latY.data(), n),
             ("unable to project values from " + orgProjStr + " to " + tmpl_proj_input).c_str());
    std::vector<double> orgX = dataReader_->xAxis(), orgY = dataReader_->yAxis();
    int miupXAxis = MIFI_PROJ_AXIS, miupYAxis = MIFI_PROJ_AXIS;
    const bool degree = isDegreeProjection(orgProjStr);
    if (degree) {
        miupXAxis = MIFI_LONGITUDE;
        miupYAxis = MIFI_LATITUDE;
        toRad(orgX);
        toRad(orgY);
    }
    points2position(latY, orgY, miupYAxis);
    points2position(lonX, orgX, miupXAxis);
    pointsOnXAxis_ = lonX;
    pointsOnYAxis_ = latY;
    auto ci = std::make_shared<CachedInterpolation>(dataReader_->xDimName(), dataReader_->yDimName(), method, pointsOnXAxis_,
                                                    pointsOnYAxis_, orgX.size(), orgY.size(), outX, outY);
    ci->createReducedDomain(dataReader_->xDimName(), dataReader_->yDimName());  // :1802
    cachedInterpolation_ = ci;
    // rotation of x/y vectors to east / north at every template point (:1808-1824)
    matrix_.resize(4 * n);
    checkAmd(fimex_amd_get_vector_reproject_matrix_points_host(orgProjStr.c_str(), LAT_LON_PROJSTR.c_str(), degree ? 0 : 1, lonRad.data(),
                                                               latRad.data(), n, matrix_.data()),
             "mifi_get_vector_reproject_matrix_points");
    shared_array<double> m(new double[matrix_.size()]);
    std::copy(matrix_.begin(), matrix_.end(), m.get());
    cachedVectorReprojection_ = std::make_shared<CachedVectorReprojection>(MIFI_VECTOR_KEEP_SIZE, m, (int)n, 1);
}

// src/CDMInterpolator.cc:304-326
double CDMInterpolator::getMaxDistanceOfInterest(const std::vector<double>& out_x_axis, const std::vector<double>& out_y_axis, bool isMetric) const
{
    if (maxDistance_ > 0) return maxDistance_;
    // the largest step of the output axes (as given, the reference does not convert degrees here) is the region of influence
    const double factor = isMetric ? 1. : 6371000.;  // MIFI_EARTH_RADIUS_M
    double maxX = 0, maxY = 0;
    for (size_t i = 0; i + 1 < out_x_axis.size(); ++i) maxX = std::max(factor * std::fabs(out_x_axis[i + 1] - out_x_axis[i]), maxX);
    for (size_t j = 0; j + 1 < out_y_axis.size(); ++j) maxY = std::max(factor * std::fabs(out_y_axis[j + 1] - out_y_axis[j]), maxY);
    return std::max(maxX, maxY);
}

// src/CDMInterpolator.cc:1335-1420
void CDMInterpolator::changeProjectionByCoordinates(int method, const std::string& proj_input, const std::vector<double>& out_x_axis,
                                                    const std::vector<double>& out_y_axis, bool xDegree, bool yDegree)
{
    const std::vector<double> orgX = dataReader_->xAxis(), orgY = dataReader_->yAxis();
    std::vector<double> lonVals, latVals;
    if (dataReader_->lonLat(lonVals, latVals)) {  // 2-D longitude / latitude variables, degrees (:1347-1357)
        if (lonVals.size() != orgX.size() * orgY.size() || latVals.size() != lonVals.size())
            throw CDMException("longitude / latitude fields do not match the grid");
    } else if (isDegreeProjection(dataReader_->projString())) {  // lonLatVals2Matrix (:1376-1380)
        lonVals.resize(orgX.size() * orgY.size());
        latVals.resize(lonVals.size());
        for (size_t j = 0; j < orgY.size(); ++j)
            for (size_t i = 0; i < orgX.size(); ++i) { lonVals[j * orgX.size() + i] = orgX[i]; latVals[j * orgX.size() + i] = orgY[j]; }
    } else {
        throw CDMException("coordinate interpolation needs longitude and latitude of the source grid");
    }
    toRad(lonVals);
    toRad(latVals);
    std::vector<double> outXAxis = out_x_axis, outYAxis = out_y_axis;
    bool isMetric = true;
    if (xDegree) { isMetric = false; toRad(outXAxis); }  // :1386-1393
    if (yDegree) toRad(outYAxis);
    projectAxesAmd(proj_input, LAT_LON_PROJSTR, outXAxis, outYAxis, pointsOnXAxis_, pointsOnYAxis_);  // :1399
    if (method == MIFI_INTERPOL_COORD_NN) {
        checkAmd(fimex_amd_coord_nearest_host(pointsOnXAxis_.data(), pointsOnYAxis_.data(), pointsOnXAxis_.size(), lonVals.data(), latVals.data(),
                                              orgX.size(), orgY.size()),
                 "fastTranslatePointsToClosestInputCell");
    } else {
        const double maxDistance = getMaxDistanceOfInterest(out_x_axis, out_y_axis, isMetric);  // :1407
        checkAmd(fimex_amd_coord_kdtree_host(maxDistance, pointsOnXAxis_.data(), pointsOnYAxis_.data(), pointsOnXAxis_.size(), lonVals.data(),
                                             latVals.data(), orgX.size(), orgY.size()),
                 "flannTranslatePointsToClosestInputCell");
    }
    cachedInterpolation_ = std::make_shared<CachedInterpolation>(dataReader_->xDimName(), dataReader_->yDimName(), method, pointsOnXAxis_,
                                                                 pointsOnYAxis_, orgX.size(), orgY.size(), out_x_axis.size(), out_y_axis.size());
    cachedVectorReprojection_.reset();  // "vector data found, but not possible? to interpolate with coordinate-interpolation" (:1418)
    matrix_.clear();
}

// src/CDMInterpolator.cc:1242-1333
void CDMInterpolator::changeProjectionByForwardInterpolation(int method, const std::string& proj_input, std::vector<double> outXAxis,
                                                             std::vector<double> outYAxis, bool xDegree, bool yDegree)
{
    std::vector<double> lonVals, latVals;
    std::vector<double> orgX = dataReader_->xAxis(), orgY = dataReader_->yAxis();
    if (!dataReader_->lonLat(lonVals, latVals)) {
        // geographic grid: the matrix of its two axes (lonLatVals2Matrix, :1285-1289); otherwise project the grid
        const std::string orgProjStr = dataReader_->projString();
        std::vector<double> ax = orgX, ay = orgY;
        if (isDegreeProjection(orgProjStr)) { toRad(ax); toRad(ay); }
        projectAxesAmd(orgProjStr, LAT_LON_PROJSTR, ax, ay, lonVals, latVals);  // radians
    } else {
        toRad(lonVals);
        toRad(latVals);
    }
    int miupXAxis = MIFI_PROJ_AXIS, miupYAxis = MIFI_PROJ_AXIS;
    if (xDegree) { toRad(outXAxis); miupXAxis = MIFI_LONGITUDE; }
    if (yDegree) { toRad(outYAxis); miupYAxis = MIFI_LATITUDE; }
    // all input points in output coordinates (:1311), then cell positions on the output axes (:1316-1317)
    checkAmd(fimex_amd_project_values_host(LAT_LON_PROJSTR.c_str(), proj_input.c_str(), lonVals.data(), latVals.data(), lonVals.size()),
             ("unable to project values from " + LAT_LON_PROJSTR + " to " + proj_input).c_str());
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

size_t sizeOfDataType(int dataType)
{
    switch (dataType) {
    case FIMEX_AMD_CDM_CHAR: case FIMEX_AMD_CDM_UCHAR: return 1;
    case FIMEX_AMD_CDM_SHORT: case FIMEX_AMD_CDM_USHORT: return 2;
    case FIMEX_AMD_CDM_INT: case FIMEX_AMD_CDM_UINT: case FIMEX_AMD_CDM_FLOAT: return 4;
    case FIMEX_AMD_CDM_DOUBLE: case FIMEX_AMD_CDM_INT64: case FIMEX_AMD_CDM_UINT64: return 8;
    default: throw CDMException("cannot convert datatype " + std::to_string(dataType));  // src/DataImpl.h:343-345
    }
}

// CDM::getFillValue without a _FillValue attribute: defaultFillValue_ (src/CDM.cc:490-505, include/fimex/CDMconstants.h:149-158)
double defaultFillValue(int dataType)
{
    switch (dataType) {
    case FIMEX_AMD_CDM_DOUBLE: return 9.9692099683868690e+36;
    case FIMEX_AMD_CDM_FLOAT: return 9.9692099683868690e+36f;
    case FIMEX_AMD_CDM_INT64: return (double)(-9223372036854775806LL);
    case FIMEX_AMD_CDM_INT: return -2147483647.;
    case FIMEX_AMD_CDM_SHORT: return -32767.;
    case FIMEX_AMD_CDM_CHAR: return -127.;
    case FIMEX_AMD_CDM_UINT64: return (double)18446744073709551614ULL;
    case FIMEX_AMD_CDM_UINT: return 4294967295.;
    case FIMEX_AMD_CDM_USHORT: return 65535.;
    case FIMEX_AMD_CDM_UCHAR: return 255.;
    default: return std::numeric_limits<double>::quiet_NaN();  // MIFI_UNDEFINED_D
    }
}

// default for readers that only hold floats
TypedData GridReader::getTypedDataSlice(const std::string& varName, size_t unLimDimPos, size_t x0, size_t nx, size_t y0, size_t ny)
{
    TypedData d;
    d.dataType = FIMEX_AMD_CDM_FLOAT;
    shared_array<float> f = getDataSlice(varName, unLimDimPos, x0, nx, y0, ny, d.size);
    d.bytes = shared_array<unsigned char>(f, reinterpret_cast<unsigned char*>(f.get()));  // aliasing: shares ownership
    return d;
}

TypedData CDMInterpolator::readTypedInput(const std::string& varName, size_t unLimDimPos, size_t levelStart, size_t levelSize) const
{
    const auto rd = cachedInterpolation_->reducedDomain();
    const size_t x0 = rd ? rd->xMin : 0, y0 = rd ? rd->yMin : 0;
    TypedData d = dataReader_->getTypedDataSlice(varName, unLimDimPos, x0, cachedInterpolation_->getInX(), y0, cachedInterpolation_->getInY());
    if (d.dataType != dataReader_->variable(varName).dataType)
        throw CDMException("reader delivered " + varName + " in type " + std::to_string(d.dataType));
    const size_t layer = cachedInterpolation_->getInX() * cachedInterpolation_->getInY();
    const size_t levels = layer ? d.size / layer : 0;
    if (levelStart == 0 && (levelSize == SliceBuilder::npos || levelSize == levels)) return d;
    // the SliceBuilder's level range (the reference's reader slices while reading, src/CachedInterpolation.cc:67-90)
    if (levelStart > levels || (levelSize != SliceBuilder::npos && levelStart + levelSize > levels))
        throw CDMException("slice of " + varName + " exceeds its " + std::to_string(levels) + " levels");
    const size_t n = (levelSize == SliceBuilder::npos ? levels - levelStart : levelSize) * layer, elem = sizeOfDataType(d.dataType);
    TypedData part;
    part.dataType = d.dataType;
    part.size = n;
    part.bytes = shared_array<unsigned char>(new unsigned char[n ? n * elem : 1]);
    std::copy(d.bytes.get() + levelStart * layer * elem, d.bytes.get() + levelStart * layer * elem + n * elem, part.bytes.get());
    return part;
}

// src/CDMInterpolator.cc:235-287
TypedData CDMInterpolator::getTypedDataSlice(const std::string& varName, size_t unLimDimPos)
{
    return regridLevels(varName, unLimDimPos, 0, SliceBuilder::npos);
}

// src/CDMInterpolator.cc:162-233
TypedData CDMInterpolator::getTypedDataSlice(const std::string& varName, const SliceBuilder& sb)
{
    TypedData full = regridLevels(varName, sb.unLimDimPos, sb.levelStart, sb.levelSize);
    if (full.size == 0) return full;  // :189-190
    const size_t outX = cachedInterpolation_->getOutX(), outY = cachedInterpolation_->getOutY();
    const size_t nx = sb.xSize == SliceBuilder::npos ? outX - std::min(outX, sb.xStart) : sb.xSize;
    const size_t ny = sb.ySize == SliceBuilder::npos ? outY - std::min(outY, sb.yStart) : sb.ySize;
    if (sb.xStart + nx > outX || sb.yStart + ny > outY) throw CDMException("slice of " + varName + " exceeds the output grid");
    if (nx == outX && ny == outY) return full;
    // slice the x and y direction of the data (:222-232)
    const size_t levels = full.size / (outX * outY), elem = sizeOfDataType(full.dataType);
    TypedData out;
    out.dataType = full.dataType;
    out.size = levels * ny * nx;
    out.bytes = shared_array<unsigned char>(new unsigned char[out.size ? out.size * elem : 1]);
    for (size_t z = 0; z < levels; ++z)
        for (size_t y = 0; y < ny; ++y) {
            const unsigned char* src = full.bytes.get() + ((z * outY + sb.yStart + y) * outX + sb.xStart) * elem;
            std::copy(src, src + nx * elem, out.bytes.get() + (z * ny + y) * nx * elem);
        }
    return out;
}

TypedData CDMInterpolator::regridLevels(const std::string& varName, size_t unLimDimPos, size_t levelStart, size_t levelSize)
{
    if (!dataReader_->hasVariable(varName)) throw CDMException("variable not found: " + varName);
    if (!cachedInterpolation_) throw CDMException("no cached interpolation for " + varName);  // :247-249
    const VariableInfo var = dataReader_->variable(varName);
    TypedData data = readTypedInput(varName, unLimDimPos, levelStart, levelSize);
    if (data.size == 0) return data;  // :252-253
    const double badValue = var.hasFillValue ? var.fillValue : defaultFillValue(var.dataType);  // CDM::getFillValue, :254

    const bool rotate = var.spatialVector &&
                        !(var.direction.find("x") == std::string::npos && var.direction.find("y") == std::string::npos);
    TypedData counterpart;
    double badCounterpart = std::numeric_limits<double>::quiet_NaN();
    bool isX = true, haveCounterpart = false;
    if (rotate) {
        if (cachedVectorReprojection_) {
            counterpart = readTypedInput(var.counterpart, unLimDimPos, levelStart, levelSize);  // :269
            haveCounterpart = true;
            const VariableInfo cv = dataReader_->variable(var.counterpart);
            badCounterpart = cv.hasFillValue ? cv.fillValue : defaultFillValue(cv.dataType);
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
    const fimex_amd_regrid_plan* h = cachedInterpolation_->amdPlan();
    TypedData out;
    out.dataType = var.dataType;  // :285 variable.getDataType()
    const size_t elem = sizeOfDataType(out.dataType);
    if (builtin && h != nullptr) {
        // the whole sequence :254-285 in one call, the slices resident in HBM between the steps
        const fimex_amd_vector_plan* vec = haveCounterpart ? cachedVectorReprojection_->handle() : nullptr;
        const void* cp = vec ? counterpart.bytes.get() : nullptr;  // uninitialised reprojection = identity (:37-40)
        checkAmd(fimex_amd_regrid_slice_typed_host(h, data.bytes.get(), data.dataType, data.size, badValue, pre.data(), pre.size(), cp,
                                                   counterpart.dataType, badCounterpart, vec, isX, post.data(), post.size(), nullptr, 0,
                                                   &out.size),
                 "interpolateValues");
        out.bytes = shared_array<unsigned char>(new unsigned char[out.size ? out.size * elem : 1]);
        checkAmd(fimex_amd_regrid_slice_typed_host(h, data.bytes.get(), data.dataType, data.size, badValue, pre.data(), pre.size(), cp,
                                                   counterpart.dataType, badCounterpart, vec, isX, post.data(), post.size(),
                                                   out.bytes.get(), out.size, &out.size),
                 "interpolateValues");
        return out;
    }

    // general path (user-defined processes or interpolation objects): the reference's sequence call by call
    auto toFloat = [&](const TypedData& d, double bad) {  // data2InterpolationArray, :115-119
        shared_array<float> f(new float[d.size]);
        checkAmd(fimex_amd_data2interpolation_host(d.bytes.get(), d.dataType, d.size, bad, f.get()), "data2InterpolationArray");
        return f;
    };
    shared_array<float> array = toFloat(data, badValue);
    processArray(preprocesses_, array.get(), data.size, cachedInterpolation_->getInX(), cachedInterpolation_->getInY());
    size_t newSize = 0;
    shared_array<float> iArray = cachedInterpolation_->interpolateValues(array, data.size, newSize);
    if (haveCounterpart) {
        shared_array<float> cArrayIn = toFloat(counterpart, badCounterpart);
        processArray(preprocesses_, cArrayIn.get(), data.size, cachedInterpolation_->getInX(), cachedInterpolation_->getInY());
        size_t cs = 0;
        shared_array<float> cArray = cachedInterpolation_->interpolateValues(cArrayIn, data.size, cs);
        if (isX) cachedVectorReprojection_->reprojectValues(iArray, cArray, newSize);
        else cachedVectorReprojection_->reprojectValues(cArray, iArray, newSize);
    }
    processArray(postprocesses_, iArray.get(), newSize, cachedInterpolation_->getOutX(), cachedInterpolation_->getOutY());
    out.size = newSize;
    out.bytes = shared_array<unsigned char>(new unsigned char[newSize ? newSize * elem : 1]);
    checkAmd(fimex_amd_interpolation2data_host(iArray.get(), newSize, out.dataType, badValue, out.bytes.get()), "interpolationArray2Data");  // :285
    return out;
}

// float convenience form for float variables
shared_array<float> CDMInterpolator::getDataSlice(const std::string& varName, size_t unLimDimPos, size_t& size)
{
    if (dataReader_->hasVariable(varName) && dataReader_->variable(varName).dataType != FIMEX_AMD_CDM_FLOAT)
        throw CDMException("getDataSlice: " + varName + " is not stored as float, use getTypedDataSlice");
    TypedData d = getTypedDataSlice(varName, unLimDimPos);
    size = d.size;
    return shared_array<float>(d.bytes, reinterpret_cast<float*>(d.bytes.get()));
}

}  // namespace FimexAmd
