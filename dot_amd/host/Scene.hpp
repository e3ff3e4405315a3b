// Scene.hpp -- host-side scene layer in C++ (the reference's own language): script parsing, .msh
// reading, normalisation, handle detection, the 3-D Dirichlet scripts, and a seedless partitioner.
// Everything here runs BEFORE the hot path and only produces the plain arrays include/dotmi.h takes.
//
// Restates (paths relative to /root/reference/src)
//   Config::loadFromFile            Config.cpp:43-208
//   IglUtils::readTetMesh           Utils/IglUtils.cpp:680-749 ; findSurfaceTris :558-590
//   main(): rotate / scale / shift  main.cpp:692-712 ; IglUtils::findBorderVerts Utils/IglUtils.cpp:909-927
//   AnimScripter::initAnimScript    AnimScripter.cpp:29-289 ; stepAnimScript :291-470
//   Mesh::setLameParam              Mesh.cpp:741-744
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace dot_amd {

struct Config {  // subset of DOT::Config the DOT path reads; defaults Config.cpp:33-37
    std::string energy = "FCR", timeStepper = "DOT", script = "null", shapePath;
    int partitionAmt = -1, blockSize = -1, warmStart = 2;
    double size = 1.0, duration = 10.0, dt = 0.025, rho = 1.0, YM = 100.0, PR = 0.4;
    bool withGravity = true;
    double rotDeg = 0.0, rotAxis[3] = {0, 0, 0}, handleRatio = 0.01;
    std::vector<double> tol;
    bool restart = false;       // `restart <status file>` (Config.cpp:164-167)
    std::string statusPath;
    // fields the DOT path does not consume; kept so that output/<name>/config.txt echoes the script the way
    // Config::saveToFile does (Config.cpp:209-302)
    std::string timeIntegration = "BE", shapeType = "grid", appendStr;
    int inexactSolve = 0, resolution = 100, maxIterAPD = 1000;
    bool orthographic = false, disableCout = false;
    double zoom = 1.0;
    std::vector<double> tuning;
};

inline Config parse_script(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open script " + path);
    Config c;
    std::string line;
    while (std::getline(in, line)) {
        std::stringstream ss(line);
        std::string tok;
        if (!(ss >> tok)) continue;
        if (tok == "energy") {
            ss >> c.energy;
            if (c.energy != "FCR" && c.energy != "SNH") c.energy = "SNH";  // default of getEnergyTypeByStr, Config.cpp:348-357
        } else if (tok == "timeStepper") {
            ss >> c.timeStepper;
            static const char *known[] = {"Newton", "ADMM", "ADMMDD", "LBFGS", "LBFGSH", "LBFGSHI", "LBFGSJH", "DOT", "GSDD"};
            bool ok = false;
            for (const char *k : known) ok |= c.timeStepper == k;
            if (!ok) c.timeStepper = "Newton";  // default of getTimeStepperTypeByStr, Config.cpp:378-387
            // Config.cpp:62-80: only the domain-decomposed steppers read a partition count; a negative count is
            // followed by the nodes per block ("DOT -1 1024", main.cpp:792-798), 0 / 1 mean the default 4
            if (c.timeStepper == "ADMMDD" || c.timeStepper == "DOT" || c.timeStepper == "LBFGSJH" ||
                c.timeStepper == "GSDD") {
                int n;
                if (ss >> n) {
                    c.partitionAmt = n;
                    if (n < 0) ss >> c.blockSize;
                    else if (n < 2) c.partitionAmt = 4;
                }
            } else if (c.timeStepper == "ADMM") {   // Config.cpp:81-89
                ss >> c.maxIterAPD;
                if (c.maxIterAPD < 1) c.maxIterAPD = 10;
            }
        } else if (tok == "timeIntegration") {
            ss >> c.timeIntegration;
            c.timeIntegration = "BE";   // the only entry of timeIntegrationTypeStrs, and the default of the lookup
        } else if (tok == "inexactSolve") ss >> c.inexactSolve;
        else if (tok == "resolution") ss >> c.resolution;
        else if (tok == "view") {
            std::string v;
            ss >> v;
            c.orthographic = v == "orthographic";
        } else if (tok == "zoom") ss >> c.zoom;
        else if (tok == "appendStr") ss >> c.appendStr;
        else if (tok == "disableCout") c.disableCout = true;
        else if (tok == "tuning") {
            int n = 0;
            ss >> n;
            c.tuning.resize(n > 0 ? n : 0);
            for (auto &t : c.tuning) in >> t;     // `file >> tuneI` (Config.cpp:183-191)
        } else if (tok == "size") ss >> c.size;
        else if (tok == "time") ss >> c.duration >> c.dt;
        else if (tok == "density") ss >> c.rho;
        else if (tok == "stiffness") ss >> c.YM >> c.PR;
        else if (tok == "turnOffGravity") c.withGravity = false;
        else if (tok == "script") ss >> c.script;
        else if (tok == "shape") {
            std::string kind;
            ss >> kind;
            static const char *shapes[] = {"grid", "square", "rectangle", "spikes", "Sharkey", "cylinder", "input"};
            c.shapeType = "grid";   // default of getShapeTypeByStr (Config.cpp:402-411)
            for (const char *k : shapes)
                if (kind == k) c.shapeType = k;
            if (kind == "input") ss >> c.shapePath;
        } else if (tok == "rotateModel") ss >> c.rotAxis[0] >> c.rotAxis[1] >> c.rotAxis[2] >> c.rotDeg;  // axis first, Config.cpp:173-176
        else if (tok == "handleRatio") ss >> c.handleRatio;
        else if (tok == "warmStart") ss >> c.warmStart;
        else if (tok == "restart") {
            c.restart = true;
            ss >> c.statusPath;
        }
        else if (tok == "tol") {
            int n = 0;
            ss >> n;
            for (int i = 0; i < n && std::getline(in, line); ++i) c.tol.push_back(std::stod(line));
        }
    }
    return c;
}

struct TetMesh {
    std::vector<double> V;   // nV*3
    std::vector<int32_t> T;  // nT*4
    std::vector<int32_t> SF; // nSF*3: the `$Surface` section of the .msh when the file has one (IglUtils.cpp:722-736)
    int nV() const { return (int)V.size() / 3; }
    int nT() const { return (int)T.size() / 4; }
};

inline TetMesh read_tet_msh(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open mesh " + path);
    TetMesh m;
    std::string line;
    while (std::getline(in, line) && line.rfind("$Nodes", 0) != 0) {}
    int one, nV;
    in >> one >> nV;
    std::getline(in, line);
    std::getline(in, line);  // ignored header line
    m.V.resize(3 * (size_t)nV);
    for (int i = 0; i < nV; ++i) {
        int id;
        in >> id >> m.V[3 * i] >> m.V[3 * i + 1] >> m.V[3 * i + 2];
    }
    while (std::getline(in, line) && line.rfind("$Elements", 0) != 0) {}
    int nT;
    in >> one >> nT;
    std::getline(in, line);
    std::getline(in, line);
    m.T.resize(4 * (size_t)nT);
    for (int e = 0; e < nT; ++e) {
        int id, a, b, c, d;
        in >> id >> a >> b >> c >> d;
        m.T[4 * e] = a - 1; m.T[4 * e + 1] = b - 1; m.T[4 * e + 2] = c - 1; m.T[4 * e + 3] = d - 1;
    }
    if (!in) throw std::runtime_error("malformed mesh " + path);
    // optional `$Surface` section: "<count>" then "a b c" rows, 1-based (IglUtils.cpp:722-736)
    while (std::getline(in, line))
        if (line.rfind("$Surface", 0) == 0) {
            int nS = 0;
            in >> nS;
            m.SF.resize(3 * (size_t)nS);
            for (int i = 0; i < 3 * nS; ++i) {
                in >> m.SF[i];
                m.SF[i] -= 1;
            }
            if (!in) throw std::runtime_error("malformed $Surface section in " + path);
            break;
        }
    return m;
}

// TetGen pair <prefix>.node / <prefix>.ele as the reference reads it (IglUtils.cpp:751-793): "n 3 0 0" then
// "id x y z" rows; "n 4 0" then "id a b c d" rows, the indices taken as they are (zero-based files)
inline TetMesh read_node_ele(const std::string &prefix)
{
    TetMesh m;
    std::ifstream in(prefix + ".node");
    if (!in) throw std::runtime_error("cannot open mesh " + prefix + ".node");
    int nN = 0, nDim = 0, z0 = 0, z1 = 0;
    in >> nN >> nDim >> z0 >> z1;
    if (!in || nN < 4 || nDim != 3) throw std::runtime_error("malformed " + prefix + ".node");
    m.V.resize(3 * (size_t)nN);
    for (int i = 0; i < nN; ++i) {
        int id;
        in >> id >> m.V[3 * i] >> m.V[3 * i + 1] >> m.V[3 * i + 2];
    }
    if (!in) throw std::runtime_error("malformed " + prefix + ".node");
    std::ifstream ie(prefix + ".ele");
    if (!ie) throw std::runtime_error("cannot open mesh " + prefix + ".ele");
    int nE = 0, nD1 = 0;
    ie >> nE >> nD1 >> z0;
    if (!ie || nE < 0 || nD1 != 4) throw std::runtime_error("malformed " + prefix + ".ele");
    m.T.resize(4 * (size_t)nE);
    for (int e = 0; e < nE; ++e) {
        int id;
        ie >> id >> m.T[4 * e] >> m.T[4 * e + 1] >> m.T[4 * e + 2] >> m.T[4 * e + 3];
    }
    if (!ie) throw std::runtime_error("malformed " + prefix + ".ele");
    return m;
}

// main.cpp:678-691: no suffix -> .node/.ele pair, ".msh" -> the MSH reader
inline TetMesh load_tet_mesh(const std::string &path)
{
    const size_t slash = path.find_last_of('/'), dot = path.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return read_node_ele(path);
    if (path.substr(dot) == ".msh") return read_tet_msh(path);
    throw std::runtime_error("unsupported tet mesh file format: " + path);
}

// status<n> as Optimizer::saveStatus writes it (Optimizer.cpp:1096-1132) and the ctor reads it back (:126-177)
inline void read_status(const std::string &path, int nV, int &timestep, std::vector<double> &x, std::vector<double> &v)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open status file " + path);
    x.clear();
    v.clear();
    timestep = 0;
    std::string line;
    while (std::getline(in, line)) {
        std::stringstream ss(line);
        std::string tok;
        if (!(ss >> tok)) continue;
        if (tok == "timestep") ss >> timestep;
        else if (tok == "position") {
            int rows = 0, cols = 0;
            ss >> rows >> cols;
            if (rows != nV || cols != 3) throw std::runtime_error("status file does not match the mesh: " + path);
            x.resize(3 * (size_t)nV);
            for (auto &c : x) in >> c;
        } else if (tok == "velocity") {
            int n = 0;
            ss >> n;
            if (n != 3 * nV) throw std::runtime_error("status file does not match the mesh: " + path);
            v.resize(3 * (size_t)nV);
            for (auto &c : v) in >> c;
        }
    }
    if (x.empty() || v.empty() || !in.eof()) throw std::runtime_error("malformed status file " + path);
}

// The surface triangles SF the reference works with: the mesh file's `$Surface` section when it has one, else
// IglUtils::findSurfaceTris (IglUtils.cpp:558-590): the four faces of every tet, outward oriented, in a map keyed
// by the ORIENTED triple (lexicographic, Triplet.h:29-45); a face is on the surface when none of the three
// rotations of its reversal is present; output in map order.  tri2tet = IglUtils::buildSTri2Tet (:591-625).
inline std::vector<std::array<int, 3>> find_surface_tris(const TetMesh &m, std::vector<int> *tri2tet = nullptr)
{
    static const int FACE[4][3] = {{0, 2, 1}, {0, 3, 2}, {0, 1, 3}, {1, 2, 3}};
    std::map<std::array<int, 3>, int> tri;
    for (int e = 0; e < m.nT(); ++e)
        for (auto &f : FACE) tri[{m.T[4 * e + f[0]], m.T[4 * e + f[1]], m.T[4 * e + f[2]]}] = e;
    std::vector<std::array<int, 3>> out;
    if (!m.SF.empty()) {
        for (size_t i = 0; i < m.SF.size() / 3; ++i) out.push_back({m.SF[3 * i], m.SF[3 * i + 1], m.SF[3 * i + 2]});
    } else {
        for (auto &kv : tri) {
            const auto &t = kv.first;
            if (tri.count({t[2], t[1], t[0]}) || tri.count({t[1], t[0], t[2]}) || tri.count({t[0], t[2], t[1]})) continue;
            out.push_back(t);
        }
    }
    if (tri2tet) {
        tri2tet->clear();
        for (auto &t : out) {
            auto it = tri.find(t);
            if (it == tri.end()) it = tri.find({t[1], t[2], t[0]});
            if (it == tri.end()) it = tri.find({t[2], t[0], t[1]});
            tri2tet->push_back(it == tri.end() ? -1 : it->second);
        }
    }
    return out;
}

// surface vertices re-indexed in ascending tet-vertex order and the surface triangles in that numbering: the
// V_surf / F_surf of main.cpp:800-830 that Optimizer::saveStatus writes as <n>.obj (Optimizer.cpp:1137-1150)
struct SurfaceMesh {
    std::vector<int> surfIndToTet;             // surface vertex -> tet vertex
    std::vector<std::array<int, 3>> F_surf;    // in surface numbering
};
inline SurfaceMesh build_surface_mesh(const TetMesh &m)
{
    const auto SF = find_surface_tris(m);
    std::vector<int> tetIndToSurf(m.nV(), -1);
    std::vector<char> on(m.nV(), 0);
    for (auto &t : SF)
        for (int k = 0; k < 3; ++k) on[t[k]] = 1;
    SurfaceMesh S;
    for (int v = 0; v < m.nV(); ++v)
        if (on[v]) {
            tetIndToSurf[v] = (int)S.surfIndToTet.size();
            S.surfIndToTet.push_back(v);
        }
    for (auto &t : SF) S.F_surf.push_back({tetIndToSurf[t[0]], tetIndToSurf[t[1]], tetIndToSurf[t[2]]});
    return S;
}

// label.obj (one "v <subdomain> 0 0" line per surface triangle) and wire.poly (surface wire frame), the two
// partition-visualisation files ADMMDDTimeStepper's ctor writes (ADMMDDTimeStepper.cpp:375-442)
inline void write_partition_files(const std::string &dir, const TetMesh &m, const std::vector<double> &x,
                                  const std::vector<int32_t> &epart)
{
    std::vector<int> tri2tet;
    const auto surf = find_surface_tris(m, &tri2tet);
    FILE *out = std::fopen((dir + "/label.obj").c_str(), "w");
    if (!out) throw std::runtime_error("cannot write into " + dir);
    for (size_t i = 0; i < surf.size(); ++i) std::fprintf(out, "v %d 0 0\n", (int)epart[tri2tet[i]]);
    std::fclose(out);
    std::vector<int> toSurf(m.nV(), -1), toTet;
    {
        std::vector<char> on(m.nV(), 0);
        for (auto &t : surf)
            for (int k = 0; k < 3; ++k) on[t[k]] = 1;
        for (int v = 0; v < m.nV(); ++v)
            if (on[v]) {
                toSurf[v] = (int)toTet.size();
                toTet.push_back(v);
            }
    }
    out = std::fopen((dir + "/wire.poly").c_str(), "w");
    if (!out) throw std::runtime_error("cannot write into " + dir);
    std::fprintf(out, "POINTS\n");
    for (size_t i = 0; i < toTet.size(); ++i)
        std::fprintf(out, "%zu: %le %le %le\n", i + 1, x[3 * toTet[i]], x[3 * toTet[i] + 1], x[3 * toTet[i] + 2]);
    std::fprintf(out, "POLYS\n");
    for (size_t f = 0; f < surf.size(); ++f)
        for (int k = 0; k < 3; ++k)
            std::fprintf(out, "%zu: %d %d\n", 3 * f + k + 1, toSurf[surf[f][k]] + 1, toSurf[surf[f][(k + 1) % 3]] + 1);
    std::fprintf(out, "END\n");
    std::fclose(out);
}

// Eigen::AngleAxis::toRotationMatrix
inline void angle_axis_matrix(double angle, const double ax_in[3], double R[3][3])
{
    const double n = std::sqrt(ax_in[0] * ax_in[0] + ax_in[1] * ax_in[1] + ax_in[2] * ax_in[2]);
    const double ax[3] = {ax_in[0] / n, ax_in[1] / n, ax_in[2] / n};
    const double s = std::sin(angle), c = std::cos(angle);
    const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]}, ca[3] = {(1 - c) * ax[0], (1 - c) * ax[1], (1 - c) * ax[2]};
    double t = ca[0] * ax[1];
    R[0][1] = t - sa[2]; R[1][0] = t + sa[2];
    t = ca[0] * ax[2];
    R[0][2] = t + sa[1]; R[2][0] = t - sa[1];
    t = ca[1] * ax[2];
    R[1][2] = t - sa[0]; R[2][1] = t + sa[0];
    R[0][0] = ca[0] * ax[0] + c; R[1][1] = ca[1] * ax[1] + c; R[2][2] = ca[2] * ax[2] + c;
}

inline void normalize(std::vector<double> &V, double size, double rotDeg, const double rotAxis[3])
{
    const int nV = (int)V.size() / 3;
    if (rotDeg != 0.0) {
        double R[3][3];
        angle_axis_matrix(rotDeg / 180.0 * M_PI, rotAxis, R);
        for (int i = 0; i < nV; ++i) {
            const double x = V[3 * i], y = V[3 * i + 1], z = V[3 * i + 2];
            for (int r = 0; r < 3; ++r) V[3 * i + r] = R[r][0] * x + R[r][1] * y + R[r][2] * z;
        }
    }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < nV; ++i)
        for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], V[3 * i + d]); hi[d] = std::max(hi[d], V[3 * i + d]); }
    const double s = size / std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]});
    for (auto &v : V) v *= s;
    for (int d = 0; d < 3; ++d) lo[d] = 1e300;
    for (int i = 0; i < nV; ++i)
        for (int d = 0; d < 3; ++d) lo[d] = std::min(lo[d], V[3 * i + d]);
    for (int i = 0; i < nV; ++i)
        for (int d = 0; d < 3; ++d) V[3 * i + d] -= lo[d];
}

inline void find_border_verts(const std::vector<double> &V, double ratio, std::vector<int> border[2])
{
    const int nV = (int)V.size() / 3;
    double lo = 1e300, hi = -1e300;
    for (int i = 0; i < nV; ++i) { lo = std::min(lo, V[3 * i]); hi = std::max(hi, V[3 * i]); }
    const double rng = hi - lo;
    for (int i = 0; i < nV; ++i) {
        if (V[3 * i] < lo + rng * ratio) border[0].push_back(i);
        else if (V[3 * i] > hi - rng * ratio) border[1].push_back(i);
    }
}

inline void lame(double YM, double PR, double &mu, double &lam)
{
    mu = YM / 2.0 / (1.0 + PR);
    lam = YM * PR / (1.0 + PR) / (1.0 - 2.0 * PR);
}

// 3-D scripts of AnimScripter.cpp, including rubberBandPull (the one script that changes the fixed set)
class AnimScripter {
public:
    std::vector<uint8_t> fixed;
    std::string script;

    AnimScripter(const std::string &scr, const std::vector<double> &Vrest, const std::vector<int> border[2])
        : script(scr)
    {
        const int nV = (int)Vrest.size() / 3;
        fixed.assign(nV, 0);
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int i = 0; i < nV; ++i)
            for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], Vrest[3 * i + d]); hi[d] = std::max(hi[d], Vrest[3 * i + d]); }
        for (int d = 0; d < 3; ++d) center_[d] = 0.5 * (lo[d] + hi[d]);
        auto sgn = [](int b) { return b % 2 ? -1.0 : 1.0; };
        if (scr == "null" || scr == "fall") {
            if (scr == "fall") fallOffset_ = 0.5 * std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
        } else if (scr == "hang") {
            for (int b = 0; b < 2; ++b)
                if (!border[b].empty()) fixed[border[b].back()] = 1;
        } else if (scr == "rubberBandPull") {
            // AnimScripter.cpp:220-258: top/bottom 2% slabs pulled apart in y, waist pulled in -x then released
            for (int i = 0; i < nV; ++i) {
                const double y = Vrest[3 * i + 1], rng = hi[1] - lo[1];
                if (y < lo[1] + rng * 0.02) { fixed[i] = 1; vel_[i] = {0, -0.2, 0}; group1_.push_back(i); }
                else if (y > hi[1] - rng * 0.02) { fixed[i] = 1; vel_[i] = {0, 0.2, 0}; group1_.push_back(i); }
                else if (y < hi[1] - rng * 0.48 && y > lo[1] + rng * 0.48) {
                    fixed[i] = 1; vel_[i] = {-2.5, 0, 0}; group0_.push_back(i);
                    if (turnVert_ < 0) { turnVert_ = i; turnLo_ = Vrest[3 * i] - 5.0; }
                }
            }
        } else {
            const bool known = scr == "stretch" || scr == "squash" || scr == "stretchnsquash" || scr == "twist" ||
                               scr == "twistnstretch" || scr == "twistnsns" || scr == "twistnsns_old";
            if (!known) throw std::runtime_error("unsupported script " + scr);
            for (int b = 0; b < 2; ++b)
                for (int v : border[b]) {
                    fixed[v] = 1;
                    if (scr == "stretch") vel_[v] = {sgn(b) * -0.1, 0, 0};
                    else if (scr == "squash") vel_[v] = {sgn(b) * 0.03, 0, 0};
                    else if (scr == "stretchnsquash") vel_[v] = {sgn(b) * -0.9, 0, 0};
                    else if (scr == "twist") ang_[v] = sgn(b) * -0.1 * M_PI;
                    else if (scr == "twistnstretch") { ang_[v] = sgn(b) * -0.1 * M_PI; vel_[v] = {sgn(b) * -0.1, 0, 0}; }
                    else { ang_[v] = sgn(b) * -0.4 * M_PI; vel_[v] = {sgn(b) * (scr == "twistnsns" ? -1.2 : -0.9), 0, 0}; }
                }
            if ((scr == "twistnsns" || scr == "twistnsns_old" || scr == "stretchnsquash") && !border[0].empty()) {
                turnVert_ = border[0].front();
                const double xv = Vrest[3 * turnVert_];
                turnLo_ = xv - (scr == "twistnsns" ? 1.2 : 0.8);
                turnHi_ = xv + 0.4;
            }
        }
    }

    std::vector<double> initial_positions(const std::vector<double> &Vrest) const
    {
        std::vector<double> x = Vrest;
        if (fallOffset_ != 0.0)
            for (size_t i = 1; i < x.size(); i += 3) x[i] += fallOffset_;
        return x;
    }

    // fills idx / pos for this step from the current positions; returns 1 if the fixed set changed
    int step(const std::vector<double> &x, double dt, std::vector<int32_t> &idx, std::vector<double> &pos)
    {
        idx.clear();
        pos.clear();
        int changed = 0;
        if (script == "null" || script == "fall" || script == "hang") return 0;
        if (script == "rubberBandPull") {
            if (turnVert_ >= 0 && x[3 * turnVert_] <= turnLo_) {  // release the waist, stop the ends
                turnLo_ = -1e300;
                for (int v : group0_) { fixed[v] = 0; vel_[v] = {0, 0, 0}; }
                for (int v : group1_) vel_[v] = {0, 0, 0};
                changed = 1;
            }
        }
        bool flip = false;
        if (script != "rubberBandPull" && turnVert_ >= 0) {
            const double xv = x[3 * turnVert_];
            flip = xv <= turnLo_ || xv >= turnHi_;
        }
        if (flip)
            for (auto &kv : vel_) kv.second[0] *= -1.0;
        const int nV = (int)fixed.size();
        for (int v = 0; v < nV; ++v) {
            const bool moving = fixed[v] || (script == "rubberBandPull" && vel_.count(v));
            if (!moving) continue;
            double d[3] = {0, 0, 0};
            auto a = ang_.find(v);
            if (a != ang_.end()) {
                double R[3][3];
                const double ux[3] = {1, 0, 0};
                angle_axis_matrix(a->second * dt, ux, R);
                double rel[3] = {x[3 * v] - center_[0], x[3 * v + 1] - center_[1], x[3 * v + 2] - center_[2]};
                for (int r = 0; r < 3; ++r)
                    d[r] = (R[r][0] * rel[0] + R[r][1] * rel[1] + R[r][2] * rel[2] + center_[r]) - x[3 * v + r];
            }
            auto w = vel_.find(v);
            if (w != vel_.end())
                for (int r = 0; r < 3; ++r) d[r] += w->second[r] * dt;
            idx.push_back(v);
            for (int r = 0; r < 3; ++r) pos.push_back(x[3 * v + r] + d[r]);
        }
        return changed;
    }

private:
    std::map<int, double> ang_;
    std::map<int, std::array<double, 3>> vel_;
    std::vector<int> group0_, group1_;
    double center_[3] = {0, 0, 0}, fallOffset_ = 0.0, turnLo_ = -1e300, turnHi_ = 1e300;
    int turnVert_ = -1;
};

// seedless recursive coordinate bisection of element centroids (same rule as dot_amd/scene.py)
inline std::vector<int32_t> partition_rcb(const TetMesh &m, int nparts)
{
    const int nT = m.nT();
    std::vector<std::array<double, 3>> cent(nT);
    for (int e = 0; e < nT; ++e)
        for (int d = 0; d < 3; ++d) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += m.V[3 * (size_t)m.T[4 * e + k] + d];
            cent[e][d] = s / 4.0;
        }
    std::vector<int32_t> epart(nT, 0);
    std::vector<int> ids(nT);
    for (int e = 0; e < nT; ++e) ids[e] = e;
    struct Job { int b, e, lo, n; };
    std::vector<Job> stack{{0, nT, 0, nparts}};
    while (!stack.empty()) {
        const Job j = stack.back();
        stack.pop_back();
        if (j.n == 1) {
            for (int i = j.b; i < j.e; ++i) epart[ids[i]] = j.lo;
            continue;
        }
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int i = j.b; i < j.e; ++i)
            for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], cent[ids[i]][d]); hi[d] = std::max(hi[d], cent[ids[i]][d]); }
        int ax = 0;
        for (int d = 1; d < 3; ++d)
            if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
        std::stable_sort(ids.begin() + j.b, ids.begin() + j.e, [&](int a, int b) { return cent[a][ax] < cent[b][ax]; });
        const int nl = j.n / 2;
        const int cut = j.b + (int)std::lround((double)(j.e - j.b) * nl / j.n);
        stack.push_back({cut, j.e, j.lo + nl, j.n - nl});
        stack.push_back({j.b, cut, j.lo, nl});
    }
    return epart;
}

}  // namespace dot_amd
