// Compile/link check of the C++ adapter (tests/test_abi.py builds it with g++, no HIP headers, and runs
// it: without a GPU precompute() must throw the ABI's "no CPU fallback" error; with one it steps a tiny bar).
#include <cstdio>
#include <cstring>
#include "DotHipTimeStepper.hpp"

int main()
{
    // 2 x 1 x 1 cubes of 6 Kuhn tets would need the mesh generator; a single positively oriented pair of
    // tets sharing a face is enough to exercise every entry point
    const double V[] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 1};
    const int32_t F[] = {0, 1, 2, 3, 1, 2, 3, 4};
    const double u[] = {35714.2857, 35714.2857}, lam[] = {142857.1428, 142857.1428};
    const uint8_t fixed[] = {1, 0, 0, 0, 0};
    const int32_t epart[] = {0, 1};
    dot_amd::MeshView m;
    m.nV = 5; m.nT = 2; m.V_rest = V; m.F = F; m.u = u; m.lambda = lam; m.density = 1000.0; m.isFixedVert = fixed;
    dot_amd::Options o;
    o.energyType = DOTMI_ENERGY_SNH; o.partitionAmt = 2; o.epart = epart;
    dot_amd::DotHipTimeStepper ts(m, o, V);
    ts.setTime(1.0, 0.025);
    ts.setRelGL2Tol();
    try {
        ts.precompute();
    } catch (const std::exception &e) {
        std::printf("precompute failed: %s\n", e.what());
        return std::strstr(e.what(), "no CPU fallback") ? 3 : 1;
    }
    const int rc = ts.solve(2);
    std::printf("solve -> %d, iter %d, inner %d, tol %.6e\n", rc, ts.getIterNum(), ts.getInnerIterAmt(), ts.getTargetGRes());
    return (rc == 0 && ts.getIterNum() == 2) ? 0 : 2;
}
