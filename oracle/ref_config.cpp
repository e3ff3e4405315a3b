// ref_config.cpp -- TEST INFRASTRUCTURE ONLY.  Driver around the REFERENCE's own script parser src/Config.cpp,
// compiled where it lies (oracle/Makefile target _ref/librefconfig.so).  Pins the f1 parser (dot_amd/scene.py,
// dot_amd/host/Scene.hpp).  Limitation: the `script <name>` token calls AnimScripter<3>::getAnimScriptTypeByStr,
// defined in AnimScripter.cpp, which needs <tbb/tbb.h> (absent) -- that symbol stays undefined in the library
// (an executable linked with lazy binding), so the driver is only ever fed script files WITHOUT a `script` line; the name table of that token
// is not pinned by this library.
#include "Config.hpp"

#include <cstdio>
#include <sstream>
#include <string>

extern "C" {

// parses `path` with DOT::Config::loadFromFile and prints the fields the DOT path consumes as "key value" lines
int ref_config_parse(const char *path, char *out, int cap)
{
    DOT::Config c;
    const int rc = c.loadFromFile(path);
    std::ostringstream o;
    o.precision(17);
    o << "rc " << rc << "\n";
    o << "energy " << DOT::Config::getStrByEnergyType(c.energyType) << "\n";
    o << "timeStepper " << DOT::Config::getStrByTimeStepperType(c.timeStepperType) << "\n";
    o << "partitionAmt " << c.partitionAmt << "\n";
    o << "blockSize " << c.blockSize << "\n";
    o << "size " << c.size << "\n";
    o << "duration " << c.duration << "\n";
    o << "dt " << c.dt << "\n";
    o << "rho " << c.rho << "\n";
    o << "YM " << c.YM << "\n";
    o << "PR " << c.PR << "\n";
    o << "withGravity " << (c.withGravity ? 1 : 0) << "\n";
    o << "shape " << DOT::Config::getStrByShapeType(c.shapeType) << "\n";
    o << "inputShapePath " << c.inputShapePath << "\n";
    o << "warmStart " << c.warmStart << "\n";
    o << "handleRatio " << c.handleRatio << "\n";
    o << "rotDeg " << c.rotDeg << "\n";
    if (c.rotDeg != 0.0)  // the constructor leaves rotAxis uninitialised (Config.cpp:33-37)
        o << "rotAxis " << c.rotAxis[0] << " " << c.rotAxis[1] << " " << c.rotAxis[2] << "\n";
    o << "restart " << (c.restart ? 1 : 0) << "\n";
    o << "statusPath " << c.statusPath << "\n";
    o << "tol " << c.tol.size();
    for (double t : c.tol) o << " " << t;
    o << "\n";
    const std::string s = o.str();
    if ((int)s.size() + 1 > cap) return -1;
    std::snprintf(out, cap, "%s", s.c_str());
    return 0;
}

}  // extern "C"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    static char buf[1 << 16];
    if (ref_config_parse(argv[1], buf, sizeof(buf))) return 1;
    std::fputs(buf, stdout);
    return 0;
}
