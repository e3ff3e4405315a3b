// ref_formats.cpp -- TEST INFRASTRUCTURE ONLY.  Driver that makes the REFERENCE's own code write the two text formats
// of SURVEY.md section 8(f) rank 2 that it can write without TBB, so that dot_amd/host/Output.hpp is pinned byte for
// byte against reference-produced files instead of against a reader of our own:
//   ref_formats config <script> <out>   DOT::Config::loadFromFile + Config::saveToFile (src/Config.cpp:43-302,
//                                       compiled where it lies) = output/<name>/config.txt (main.cpp:786)
//   ref_formats info <out> nV nT iterNum innerIterAmt t0 .. t21
//                                       Timer::print (src/Utils/Timer.hpp:58-68, included where it lies) inside the
//                                       statements of saveInfoForPresent (main.cpp:338-358, restated here: main.cpp
//                                       itself needs the viewer); activities registered as main.cpp:865-888 does
// Two things here are NOT the reference and are not pinned by this driver:
//   * the `script <name>` token: Config.cpp calls AnimScripter<3>::getStrByAnimScriptType, defined in
//     AnimScripter.cpp, which includes <tbb/tbb.h> (absent).  The definition below returns the marker "@SCRIPT@" so
//     that saveToFile runs to its end; the tests substitute the marker.  getAnimScriptTypeByStr stays undefined
//     (lazy binding) and the driver is only fed scripts without a `script` line, as ref_config.cpp.
//   * the timings: Timer has no setter, so its private vector is reached with `#define private public`.
#include <cassert>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include <sys/time.h>

// every standard header Timer.hpp includes is already in (guards), so the redefinition only touches class Timer
#define private public
#include "Timer.hpp"
#undef private
#include "Config.hpp"

namespace DOT {
template <>
std::string AnimScripter<3>::getStrByAnimScriptType(AnimScriptType)
{
    return "@SCRIPT@";
}
}  // namespace DOT

static DOT::Config config;   // a global like main.cpp:31: the enum members the constructor leaves alone are zero

int main(int argc, char **argv)
{
    if (argc >= 4 && std::string(argv[1]) == "config") {
        if (config.loadFromFile(argv[2]) != 0) return 1;
        config.saveToFile(argv[3]);
        return 0;
    }
    if (argc == 7 + 22 && std::string(argv[1]) == "info") {
        Timer timer, timer_step, timer_temp3;
        timer.new_activity("descent");
        static const char *stepNames[14] = {"matrixComputation", "matrixAssembly", "symbolicFactorization",
                                            "numericalFactorization", "backSolve", "lineSearch_other", "modifyGrad",
                                            "modifySearchDir", "updateHistory", "lineSearch_eVal", "fullyImplicit_eComp",
                                            "solve_extraComp", "compGrad", "CCD"};
        static const char *temp3Names[7] = {"init", "initPrimal", "initDual", "initWeights", "initCons", "subdSolve",
                                            "consSolve"};
        for (const char *n : stepNames) timer_step.new_activity(n);
        for (const char *n : temp3Names) timer_temp3.new_activity(n);
        timer.timings_[0] = std::atof(argv[7]);
        for (int k = 0; k < 14; ++k) timer_step.timings_[k] = std::atof(argv[8 + k]);
        for (int k = 0; k < 7; ++k) timer_temp3.timings_[k] = std::atof(argv[22 + k]);
        const double energyParam0 = 1.0;   // main.cpp:891
        std::ofstream file(argv[2]);
        file << std::atoi(argv[3]) << " " << std::atoi(argv[4]) << std::endl;
        file << std::atoi(argv[5]) << " " << std::atoi(argv[6]) << " 0 0 " << 1.0 - energyParam0 << std::endl;
        timer.print(file);
        timer_step.print(file);
        timer_temp3.print(file);
        double distortion = 0.0;
        file << distortion << " " << 0.0 << std::endl;
        return 0;
    }
    std::cerr << "usage: ref_formats config <script> <out> | info <out> nV nT iterNum innerIterAmt t0..t21\n";
    return 2;
}
