// Test driver for the C++ host mirror (tests/test_host_cdminterpolator.py): reads a grid description and float
// slices from simple binary files, runs CDMInterpolator::changeProjection + getDataSlice, writes the results.
//
// usage: host_cli <spec.txt> <out_dir>
// spec.txt (one item per line):
//   proj <proj4 string of the source grid>
//   xaxis <file of doubles>      yaxis <file of doubles>
//   var <name> <levels> <file [steps][levels][ny][nx]> <fill|nan> [vector <counterpart> <x|y>] [type <char|short|int|float|double|uchar|ushort|uint|int64|uint64>]
//       (the file holds elements of that type, float by default)
//   method <name>                 outproj <proj4>
//   lon2d|lat2d <file of doubles [ny][nx], degrees>   (coord_nearestneighbor / coord_kdtree / forward_*)   maxdist <metres>
//   outx <file of doubles> <unit> outy <file of doubles> <unit>
//   points <file of lon doubles> <file of lat doubles>            (changeProjection(method, lonVals, latVals) instead of outx / outy)
//   crosssection <name> <lon> <lat> [<lon> <lat> ...]            (repeatable; changeProjectionToCrossSections; also writes
//       vcross_bnds.i32, target_lon.f64, target_lat.f64 and prints "vcross <names...>")
//   template <file of lon floats> <file of lat floats> <nx> <ny>  (changeProjectionToTemplate)
//   pre|post fill2d <relaxCrit> <corrEff> <maxLoop> | creepfill2d <repeat> <weight> | creepfillval2d <repeat> <weight> <default>
//   slice <var> <step> <levelStart> <levelSize> <xStart> <xSize> <yStart> <ySize>   (getDataSlice with a SliceBuilder; output
//       <out_dir>/<var>_<step>_slice.f32 | .raw)
//   get <var> <step>              (repeatable; output: <out_dir>/<var>_<step>.f32 for float variables, .raw in the stored type otherwise)
// Also writes <out_dir>/points_x.f64, points_y.f64 (plan positions) and matrix.f64 (rotation matrix, if any).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>

#include "CDMInterpolator.h"

using namespace FimexAmd;

template <typename T>
static std::vector<T> readAll(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw CDMException("cannot read " + path);
    const size_t bytes = (size_t)f.tellg();
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), bytes);
    return v;
}

template <typename T>
static void writeAll(const std::string& path, const T* p, size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(p), n * sizeof(T));
}

class FileGridReader : public GridReader {
public:
    std::string proj;
    std::vector<double> x, y;
    std::map<std::string, VariableInfo> vars;
    std::map<std::string, std::vector<unsigned char>> data;  // raw elements of the variable's type
    std::vector<double> lon2d, lat2d;                         // optional 2-D coordinates, degrees, [ny][nx]
    bool lonLat(std::vector<double>& lon, std::vector<double>& lat) const override
    {
        if (lon2d.empty() || lat2d.empty()) return false;
        lon = lon2d;
        lat = lat2d;
        return true;
    }
    std::string projString() const override { return proj; }
    std::vector<double> xAxis() const override { return x; }
    std::vector<double> yAxis() const override { return y; }
    bool hasVariable(const std::string& n) const override { return vars.count(n) != 0; }
    VariableInfo variable(const std::string& n) const override { return vars.at(n); }
    TypedData getTypedDataSlice(const std::string& n, size_t step, size_t x0, size_t nx, size_t y0, size_t ny) override
    {
        const VariableInfo& v = vars.at(n);
        const std::vector<unsigned char>& d = data.at(n);
        const size_t NX = x.size(), NY = y.size(), e = sizeOfDataType(v.dataType);
        TypedData out;
        out.dataType = v.dataType;
        out.size = v.levels * nx * ny;
        out.bytes = shared_array<unsigned char>(new unsigned char[out.size ? out.size * e : 1]);
        for (size_t l = 0; l < v.levels; ++l)
            for (size_t j = 0; j < ny; ++j)
                std::memcpy(out.bytes.get() + ((l * ny + j) * nx) * e, d.data() + (((step * v.levels + l) * NY + y0 + j) * NX + x0) * e, nx * e);
        return out;
    }
    shared_array<float> getDataSlice(const std::string& n, size_t step, size_t x0, size_t nx, size_t y0, size_t ny, size_t& size) override
    {
        if (vars.at(n).dataType != FIMEX_AMD_CDM_FLOAT) throw CDMException(n + " is not stored as float");
        TypedData d = getTypedDataSlice(n, step, x0, nx, y0, ny);
        size = d.size;
        return shared_array<float>(d.bytes, reinterpret_cast<float*>(d.bytes.get()));
    }
};

// host-only mode (no GPU):  host_cli --method <name>  -> prints the method code
static int hostOnly(int argc, char** argv)
{
    const std::string mode = argv[1];
    if (mode == "--method" && argc == 3) {
        std::cout << mifi_string_to_interpolation_method(argv[2]) << std::endl;
        return 0;
    }
    return 2;
}

int main(int argc, char** argv)
{
    if (argc >= 2 && std::string(argv[1]).rfind("--", 0) == 0) {
        try {
            const int rc = hostOnly(argc, argv);
            if (rc == 2) std::cerr << "bad arguments\n";
            return rc;
        } catch (const std::exception& e) {
            std::cerr << e.what() << std::endl;
            return 1;
        }
    }
    if (argc != 3) { std::cerr << "usage: host_cli <spec.txt> <out_dir>\n"; return 2; }
    try {
        auto reader = std::make_shared<FileGridReader>();
        std::ifstream spec(argv[1]);
        const std::string outDir = argv[2];
        std::string line, method, outproj, outxUnit, outyUnit;
        std::vector<double> outx, outy, pointLon, pointLat;
        std::vector<float> tmplLon, tmplLat;
        size_t tmplNx = 0, tmplNy = 0;
        bool usePoints = false, useTemplate = false;
        std::vector<CrossSectionDefinition> crossSections;
        double maxDist = -1;
        std::vector<std::pair<std::string, size_t>> gets;
        std::vector<std::pair<std::string, SliceBuilder>> slices;
        std::vector<std::pair<bool, std::shared_ptr<InterpolatorProcess2d>>> procs;
        while (std::getline(spec, line)) {
            std::istringstream in(line);
            std::string key;
            if (!(in >> key)) continue;
            if (key == "proj") { std::getline(in, reader->proj); }
            else if (key == "outproj") { std::getline(in, outproj); }
            else if (key == "xaxis") { std::string f; in >> f; reader->x = readAll<double>(f); }
            else if (key == "yaxis") { std::string f; in >> f; reader->y = readAll<double>(f); }
            else if (key == "lon2d") { std::string f; in >> f; reader->lon2d = readAll<double>(f); }
            else if (key == "lat2d") { std::string f; in >> f; reader->lat2d = readAll<double>(f); }
            else if (key == "maxdist") { in >> maxDist; }
            else if (key == "outx") { std::string f; in >> f >> outxUnit; outx = readAll<double>(f); }
            else if (key == "outy") { std::string f; in >> f >> outyUnit; outy = readAll<double>(f); }
            else if (key == "points") { std::string a, b; in >> a >> b; pointLon = readAll<double>(a); pointLat = readAll<double>(b); usePoints = true; }
            else if (key == "template") { std::string a, b; in >> a >> b >> tmplNx >> tmplNy; tmplLon = readAll<float>(a); tmplLat = readAll<float>(b); useTemplate = true; }
            else if (key == "crosssection") {
                std::string name;
                in >> name;
                std::vector<std::pair<double, double>> pts;
                double lo, la;
                while (in >> lo >> la) pts.push_back({lo, la});
                crossSections.push_back(CrossSectionDefinition(name, pts));
            }
            else if (key == "method") { in >> method; }
            else if (key == "var") {
                VariableInfo v;
                std::string file, fill, kw;
                in >> v.name >> v.levels >> file >> fill;
                if (fill != "nan") { v.hasFillValue = true; v.fillValue = std::stod(fill); }
                while (in >> kw) {
                    if (kw == "vector") { v.spatialVector = true; in >> v.counterpart >> v.direction; }
                    else if (kw == "type") {
                        std::string t;
                        in >> t;
                        static const std::map<std::string, int> types = {
                            {"char", FIMEX_AMD_CDM_CHAR}, {"short", FIMEX_AMD_CDM_SHORT}, {"int", FIMEX_AMD_CDM_INT}, {"float", FIMEX_AMD_CDM_FLOAT},
                            {"double", FIMEX_AMD_CDM_DOUBLE}, {"uchar", FIMEX_AMD_CDM_UCHAR}, {"ushort", FIMEX_AMD_CDM_USHORT},
                            {"uint", FIMEX_AMD_CDM_UINT}, {"int64", FIMEX_AMD_CDM_INT64}, {"uint64", FIMEX_AMD_CDM_UINT64}};
                        if (!types.count(t)) throw CDMException("unknown type " + t);
                        v.dataType = types.at(t);
                    }
                }
                reader->vars[v.name] = v;
                reader->data[v.name] = readAll<unsigned char>(file);
            } else if (key == "pre" || key == "post") {
                std::string kind;
                in >> kind;
                std::shared_ptr<InterpolatorProcess2d> p;
                if (kind == "fill2d") { float a, b; size_t n; in >> a >> b >> n; p = std::make_shared<InterpolatorFill2d>(a, b, n); }
                else if (kind == "creepfill2d") { int r, w; in >> r >> w; p = std::make_shared<InterpolatorCreepFill2d>((unsigned short)r, (char)w); }
                else if (kind == "creepfillval2d") { int r, w; float d; in >> r >> w >> d; p = std::make_shared<InterpolatorCreepFillVal2d>((unsigned short)r, (char)w, d); }
                else throw CDMException("unknown process " + kind);
                procs.push_back({key == "pre", p});
            } else if (key == "slice") {
                std::string v;
                SliceBuilder sb;
                in >> v >> sb.unLimDimPos >> sb.levelStart >> sb.levelSize >> sb.xStart >> sb.xSize >> sb.yStart >> sb.ySize;
                slices.push_back({v, sb});
            } else if (key == "get") { std::string v; size_t s; in >> v >> s; gets.push_back({v, s}); }
        }
        CDMInterpolator interp(reader);
        for (auto& p : procs) { if (p.first) interp.addPreprocess(p.second); else interp.addPostprocess(p.second); }
        const int m = mifi_string_to_interpolation_method(method.c_str());
        if (m == MIFI_INTERPOL_UNKNOWN) throw CDMException("unknown method " + method);
        interp.setDistanceOfInterest(maxDist);
        if (!crossSections.empty()) {
            interp.changeProjectionToCrossSections(m, crossSections);
            writeAll(outDir + "/vcross_bnds.i32", interp.crossSectionBounds().data(), interp.crossSectionBounds().size());
            writeAll(outDir + "/target_lon.f64", interp.targetLongitudes().data(), interp.targetLongitudes().size());
            writeAll(outDir + "/target_lat.f64", interp.targetLatitudes().data(), interp.targetLatitudes().size());
            std::cout << "vcross";
            for (const std::string& n : interp.crossSectionNames()) std::cout << " " << n;
            std::cout << std::endl;
        } else if (usePoints) interp.changeProjection(m, pointLon, pointLat);
        else if (useTemplate) interp.changeProjectionToTemplate(m, tmplLon, tmplLat, tmplNx, tmplNy);
        else interp.changeProjection(m, outproj, outx, outy, outxUnit, outyUnit);
        writeAll(outDir + "/points_x.f64", interp.pointsOnXAxis().data(), interp.pointsOnXAxis().size());
        writeAll(outDir + "/points_y.f64", interp.pointsOnYAxis().data(), interp.pointsOnYAxis().size());
        writeAll(outDir + "/matrix.f64", interp.rotationMatrix().data(), interp.rotationMatrix().size());
        auto ci = interp.cachedInterpolation();
        std::cout << "inX " << ci->getInX() << " inY " << ci->getInY() << " outX " << ci->getOutX() << " outY " << ci->getOutY();
        if (auto rd = ci->reducedDomain()) std::cout << " reduced xMin " << rd->xMin << " yMin " << rd->yMin;
        if (ci->amdPlan()) {  // which apply kernel the plan holds: LDS-staged (source cells streamed per slice, widest tile) or gather
            fimex_amd_plan_info info;
            if (fimex_amd_regrid_plan_info(ci->amdPlan(), &info) == FIMEX_AMD_OK)
                std::cout << " stagedCells " << info.stagedCells << " tile " << info.tileW << "x" << info.tileH;
        }
        std::cout << std::endl;
        for (auto& g : gets) {
            const TypedData out = interp.getTypedDataSlice(g.first, g.second);
            const bool isFloat = out.dataType == FIMEX_AMD_CDM_FLOAT;
            writeAll(outDir + "/" + g.first + "_" + std::to_string(g.second) + (isFloat ? ".f32" : ".raw"), out.bytes.get(),
                     out.size * sizeOfDataType(out.dataType));
        }
        for (auto& sl : slices) {
            const TypedData out = interp.getTypedDataSlice(sl.first, sl.second);
            const bool isFloat = out.dataType == FIMEX_AMD_CDM_FLOAT;
            writeAll(outDir + "/" + sl.first + "_" + std::to_string(sl.second.unLimDimPos) + "_slice" + (isFloat ? ".f32" : ".raw"),
                     out.bytes.get(), out.size * sizeOfDataType(out.dataType));
        }
        return 0;
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
}
