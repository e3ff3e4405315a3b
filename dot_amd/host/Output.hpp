// Output.hpp -- the on-disk formats of the reference's headless run that carry numbers from the hot path:
//   info.txt   saveInfoForPresent (main.cpp:338-358): two header lines, then Timer::print (Utils/Timer.hpp:58-68) of
//              `timer` (1 activity), `timer_step` (14 activities, names main.cpp:867-880) and `timer_temp3` (7
//              activities, :882-888), then "<distortion> 0"
//   <n>.obj    igl::writeOBJ(V_surf, F_surf) from Optimizer::saveStatus (Optimizer.cpp:1137-1150): surface vertices
//              re-indexed, Eigen FullPrecision (15 significant digits), faces 1-based
//   config.txt Config::saveToFile (Config.cpp:209-302), the echo of the effective script
// The timer_step slots are filled from dotmi_step_stats.ms_phase (HIP events on the library's stream).
#pragma once
#include <array>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Scene.hpp"

namespace dot_amd {

static const char *const TIMER_STEP_NAMES[14] = {
    "matrixComputation", "matrixAssembly", "symbolicFactorization", "numericalFactorization", "backSolve",
    "lineSearch_other", "modifyGrad", "modifySearchDir", "updateHistory", "lineSearch_eVal", "fullyImplicit_eComp",
    "solve_extraComp", "compGrad", "CCD"};
static const char *const TIMER_TEMP3_NAMES[7] = {"init", "initPrimal", "initDual", "initWeights", "initCons", "subdSolve",
                                                 "consSolve"};

struct RunTimers {
    double descent = 0;     // seconds inside the stepper's solve (timer activity 0)
    double step[14] = {0};  // seconds per timer_step activity
    double temp3[7] = {0};  // ADMM-only activities: always 0 for DOT
};

// Timer::print (Utils/Timer.hpp:58-68)
inline void print_timer(std::ostream &os, int n, const double *timings, const char *const *names)
{
    double sum = 0.0;
    os << n << " activities:\n";
    for (int i = 0; i < n; ++i) {
        os.width(10);
        os << std::right << timings[i] << " s: " << names[i] << "\n";
        sum += timings[i];
    }
    os.width(10);
    os << std::right << sum << " s: Total\n";
}

inline void write_info_txt(const std::string &path, int vertAmtInput, int nT, int iterNum, int innerIterAmt,
                           const RunTimers &t)
{
    std::ofstream file(path);
    if (!file) throw std::runtime_error("cannot write " + path);
    file << vertAmtInput << " " << nT << std::endl;
    // energyParams[0] = 1.0 for the DOT run (main.cpp:891), so the last field is 0
    file << iterNum << " " << innerIterAmt << " 0 0 " << 1.0 - 1.0 << std::endl;
    static const char *const descentName[1] = {"descent"};
    print_timer(file, 1, &t.descent, descentName);
    print_timer(file, 14, t.step, TIMER_STEP_NAMES);
    print_timer(file, 7, t.temp3, TIMER_TEMP3_NAMES);
    file << 0.0 << " " << 0.0 << std::endl;
}

// output/<name>/config.txt: the effective script as Config::saveToFile echoes it (Config.cpp:209-302; written by
// main.cpp:786).  Same tokens, order, conditions and default ostream number formatting; pinned byte for byte against
// the reference's own saveToFile on every shipped script (tests/golden/ref_formats.json).
inline void write_config_txt(const std::string &path, const Config &c)
{
    std::ofstream file(path);
    if (!file) throw std::runtime_error("cannot write " + path);
    file << "energy " << c.energy << std::endl;
    file << "timeIntegration " << c.timeIntegration << std::endl;
    file << "timeStepper " << c.timeStepper;
    if (c.timeStepper == "ADMMDD" || c.timeStepper == "DOT" || c.timeStepper == "LBFGSJH" || c.timeStepper == "GSDD") {
        if (c.blockSize > 0) file << " -1 " << c.blockSize;
        else file << " " << c.partitionAmt;
    } else if (c.timeStepper == "ADMM") {
        file << " " << c.maxIterAPD;
    }
    file << std::endl;
    file << "inexactSolve " << c.inexactSolve << std::endl;
    file << "warmStart " << c.warmStart << std::endl;
    file << "resolution " << c.resolution << std::endl;
    file << "size " << c.size << std::endl;
    file << "time " << c.duration << " " << c.dt << std::endl;
    file << "density " << c.rho << std::endl;
    file << "stiffness " << c.YM << " " << c.PR << std::endl;
    if (!c.withGravity) file << "turnOffGravity" << std::endl;
    file << "script " << c.script << std::endl;
    if (c.handleRatio != 0.01) file << "handleRatio " << c.handleRatio << std::endl;
    file << "shape " << c.shapeType;
    if (c.shapeType == "input") file << " " << c.shapePath;
    file << std::endl;
    if (c.rotDeg != 0.0)
        file << "rotateModel " << c.rotAxis[0] << " " << c.rotAxis[1] << " " << c.rotAxis[2] << " " << c.rotDeg << std::endl;
    if (c.restart) file << "restart " << c.statusPath << std::endl;
    if (!c.tuning.empty()) {
        file << "tuning " << c.tuning.size() << std::endl;
        for (double t : c.tuning) file << t << std::endl;
    }
    file << "view " << (c.orthographic ? "orthographic" : "perspective") << std::endl;
    file << "zoom " << c.zoom << std::endl;
    if (!c.appendStr.empty()) file << "appendStr " << c.appendStr << std::endl;
    if (c.disableCout) file << "disableCout" << std::endl;
    if (!c.tol.empty()) {
        file << "tol " << c.tol.size() << std::endl;
        for (double t : c.tol) file << t << std::endl;
    }
}

// igl::writeOBJ(str, V, F) (libigl writeOBJ.cpp:100-120): IOFormat(FullPrecision, DontAlignCols, " ", "\n", "v ", "",
// "", "\n") for V and the same with "f " for F + 1
inline void write_surface_obj(const std::string &path, const std::vector<double> &x, const std::vector<int> &surfIndToTet,
                              const std::vector<std::array<int, 3>> &F_surf)
{
    std::ofstream s(path);
    if (!s) throw std::runtime_error("cannot write " + path);
    s.precision(15);   // Eigen::FullPrecision for double
    for (size_t i = 0; i < surfIndToTet.size(); ++i) {
        const double *p = &x[3 * (size_t)surfIndToTet[i]];
        s << "v " << p[0] << " " << p[1] << " " << p[2] << "\n";
    }
    for (auto &f : F_surf) s << "f " << f[0] + 1 << " " << f[1] + 1 << " " << f[2] + 1 << "\n";
}

}  // namespace dot_amd
