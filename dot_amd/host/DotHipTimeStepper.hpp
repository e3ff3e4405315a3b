// DotHipTimeStepper.hpp -- C++ host adapter over the C ABI (include/dotmi.h) with the surface the
// reference's main loop uses on DOT::Optimizer<3> (src/TimeStepper/Optimizer.hpp:83-112, call sites
// src/main.cpp:92-132, :936-942).  Header-only, no HIP headers: a reference-side maintainer adds this
// class to the stepper factory and links libdotmi.so.
//
//   reference member            here
//   ------------------------    -----------------------------------------------------------------
//   Optimizer(mesh, energy,..)  DotHipTimeStepper(MeshView, Options)          (arrays are copied by the ABI)
//   setTime(duration, dt)       setTime                                       (before precompute)
//   precompute()                precompute            -> dotmi_create         (DOTTimeStepper.cpp:150-178)
//   setRelGL2Tol(tol)           setRelGL2Tol                                  (before precompute)
//   solve(maxIter)              solve                 -> script move + dotmi_step, same 0/1/2 codes
//   getResult().V               getResult             -> dotmi_get_state
//   getIterNum/getInnerIterAmt  same
//   updatePrecondMtrAndFactorize same                 -> dotmi_refactor / dotmi_refix
//   saveStatus()                saveStatus(path)      -> status<n> text format (Optimizer.cpp:1096-1132)
#pragma once
#include <cstdint>
#include <cstdio>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dotmi.h"

namespace dot_amd {

struct MeshView {  // Mesh<3> fields the path reads (src/Mesh.hpp:38-60); row-major, caller-owned
    int32_t nV = 0, nT = 0;
    const double *V_rest = nullptr;  // nV*3
    const int32_t *F = nullptr;      // nT*4
    const double *u = nullptr;       // nT  (Mesh::u)
    const double *lambda = nullptr;  // nT
    double density = 1.0;
    const uint8_t *isFixedVert = nullptr;  // nV
};

struct Options {  // the Config fields the DOT stepper reads (src/Config.hpp)
    int energyType = DOTMI_ENERGY_FCR;  // ET_FCR / ET_SNH
    bool withGravity = true;
    int partitionAmt = 4;
    const int32_t *epart = nullptr;  // METIS::partMesh result (nT)
    const int32_t *vpart = nullptr;  // optional METIS::partMesh_nodes result (nV): block-Jacobi subdomains (LBFGS-JH)
    int device = 0, rank = 0, world = 1;
    const void *commId = nullptr;
    double alphaMin = 0.1;  // lower clamp of alpha_0 (Optimizer.cpp:1085); 1.0 = unit first step (LBFGS-H, :1088)
    int flags = 0;  // DOTMI_FLAG_* (e.g. DOTMI_FLAG_TIME_PHASES to fill the reference's timer_step slots)
};

class DotHipTimeStepper {
public:
    // scripted Dirichlet motion for one step: fill idx/pos from the current positions
    // (AnimScripter::stepAnimScript, AnimScripter.cpp:291-470); return 1 if the fixed set changed
    using Script = std::function<int(const std::vector<double> &x, double dt, std::vector<int32_t> &idx,
                                     std::vector<double> &pos, std::vector<uint8_t> &fixed)>;

    DotHipTimeStepper(const MeshView &mesh, const Options &opt, const double *x_init)
        : mesh_(mesh), opt_(opt), x0_(x_init, x_init + 3 * (size_t)mesh.nV),
          fixed_(mesh.isFixedVert, mesh.isFixedVert + mesh.nV)
    {
    }
    ~DotHipTimeStepper() { dotmi_destroy(h_); }
    DotHipTimeStepper(const DotHipTimeStepper &) = delete;
    DotHipTimeStepper &operator=(const DotHipTimeStepper &) = delete;

    void setTime(double duration, double dt)  // Optimizer.cpp:249-257
    {
        require_not_built("setTime");
        dt_ = dt;
        frameAmt_ = (int)(duration / dt);
    }
    void setRelGL2Tol(double relTol = 1.0e-5)  // Optimizer.cpp:222-228 (squares its argument internally)
    {
        if (h_ && relTol != relTol_) throw std::logic_error("setRelGL2Tol after precompute: rebuild the stepper");
        relTol_ = relTol;
    }
    void setAllowEDecRelTol(bool) {}  // main.cpp:942 switches it off; this path never uses it
    void setScript(Script s) { script_ = std::move(s); }

    void precompute()  // DOTTimeStepper.cpp:150-178
    {
        require_not_built("precompute");
        dotmi_mesh m{};
        m.nV = mesh_.nV;
        m.nT = mesh_.nT;
        m.X_rest = mesh_.V_rest;
        m.T = mesh_.F;
        m.mu = mesh_.u;
        m.lambda = mesh_.lambda;
        m.density = mesh_.density;
        m.fixed = fixed_.data();
        m.epart = opt_.epart;
        m.nParts = opt_.partitionAmt;
        m.vpart = opt_.vpart;
        dotmi_params p{};
        p.energy = opt_.energyType;
        p.dt = dt_;
        p.gravity[1] = opt_.withGravity ? -9.80665 : 0.0;  // Optimizer.cpp:107-110
        p.relTol = relTol_;
        p.history = 5;        // DOTTimeStepper.cpp:45
        p.iterCap = 10000;    // DOTTimeStepper.cpp:302
        p.alphaMin = opt_.alphaMin;
        p.device = opt_.device;
        p.rank = opt_.rank;
        p.world = opt_.world;
        p.comm_id = opt_.commId;
        p.flags = opt_.flags;
        if (int rc = dotmi_create(&m, &p, x0_.data(), &h_))
            throw std::runtime_error(std::string("dotmi_create: ") + dotmi_last_error(nullptr) + " (" +
                                     std::to_string(rc) + ")");
    }

    // 0 stepped, 1 all frames done, 2 stepped but iteration cap / line-search failure (Optimizer.cpp:327-368)
    int solve(int maxIter = 1)
    {
        require_built("solve");
        int flag = 0;
        std::vector<double> x(3 * (size_t)mesh_.nV), pos;
        std::vector<int32_t> idx;
        for (int it = 0; it < maxIter; ++it) {
            if (script_) {
                check(dotmi_get_state(h_, x.data(), nullptr, nullptr), "get_state");
                idx.clear();
                pos.clear();
                // move first, then re-pattern at the moved positions (Optimizer.cpp:333-335: stepAnimScript moves
                // result.V and only then updatePrecondMtrAndFactorize runs)
                const bool changed = script_(x, dt_, idx, pos, fixed_);
                check(dotmi_set_dirichlet(h_, (int32_t)idx.size(), idx.data(), pos.data()), "set_dirichlet");
                if (changed) check(dotmi_refix(h_, fixed_.data()), "refix");
            }
            if (globalIterNum_ >= frameAmt_) {
                ++globalIterNum_;
                return 1;
            }
            dotmi_step_stats st;
            const int rc = check(dotmi_step(h_, &st), "step");
            if (rc == 2) flag = 2;
            innerIterAmt_ += st.iters;
            lastStats_ = st;
            ++globalIterNum_;
        }
        return flag;
    }

    std::vector<double> getResult()  // result.V, row-major nV x 3
    {
        require_built("getResult");
        std::vector<double> x(3 * (size_t)mesh_.nV);
        check(dotmi_get_state(h_, x.data(), nullptr, nullptr), "get_state");
        return x;
    }
    int getIterNum() const { return globalIterNum_; }
    int getInnerIterAmt() const { return innerIterAmt_; }
    double getTargetGRes() const { return dotmi_target_gres(h_); }
    const dotmi_step_stats &lastStats() const { return lastStats_; }
    void updatePrecondMtrAndFactorize()
    {
        require_built("updatePrecondMtrAndFactorize");
        check(dotmi_refactor(h_, nullptr), "refactor");
    }

    // status<n>: "timestep n", positions, velocity (xyz interleaved), dx_Elastic (Optimizer.cpp:1096-1132)
    void saveStatus(const std::string &path)
    {
        require_built("saveStatus");
        const size_t n = 3 * (size_t)mesh_.nV;
        std::vector<double> x(n), v(n), xt(n);
        check(dotmi_get_state(h_, x.data(), v.data(), xt.data()), "get_state");
        FILE *f = std::fopen(path.c_str(), "w");
        if (!f) throw std::runtime_error("cannot open " + path);
        std::fprintf(f, "timestep %d\n\nposition %d 3\n", globalIterNum_, mesh_.nV);
        for (int i = 0; i < mesh_.nV; ++i) std::fprintf(f, "%le %le %le\n", x[3 * i], x[3 * i + 1], x[3 * i + 2]);
        std::fprintf(f, "\nvelocity %zu\n", n);
        for (size_t i = 0; i < n; ++i) std::fprintf(f, "%le\n", v[i]);
        std::fprintf(f, "\ndx_Elastic %d 3\n", mesh_.nV);
        for (int i = 0; i < mesh_.nV; ++i)
            std::fprintf(f, "%le %le %le\n", x[3 * i] - xt[3 * i], x[3 * i + 1] - xt[3 * i + 1],
                         x[3 * i + 2] - xt[3 * i + 2]);
        std::fclose(f);
    }
    // restart (Config token `restart <path>`, Optimizer.cpp:126-177): positions + velocity
    void restoreState(const std::vector<double> &x, const std::vector<double> &v, int timestep)
    {
        require_built("restoreState");
        check(dotmi_set_state(h_, x.data(), v.data(), nullptr), "set_state");
        check(dotmi_refactor(h_, nullptr), "refactor");
        globalIterNum_ = timestep;
    }

    dotmi_handle *handle() { return h_; }

private:
    int check(int rc, const char *what)
    {
        if (rc < 0) throw std::runtime_error(std::string("dotmi_") + what + ": " + dotmi_last_error(h_));
        return rc;
    }
    void require_built(const char *w) const
    {
        if (!h_) throw std::logic_error(std::string(w) + " before precompute()");
    }
    void require_not_built(const char *w) const
    {
        if (h_) throw std::logic_error(std::string(w) + " after precompute()");
    }
    MeshView mesh_;
    Options opt_;
    std::vector<double> x0_;
    std::vector<uint8_t> fixed_;
    Script script_;
    dotmi_handle *h_ = nullptr;
    double dt_ = 0.025, relTol_ = 1.0e-5;  // Optimizer.cpp:111 default setTime(10, 0.025)
    int frameAmt_ = 400, globalIterNum_ = 0, innerIterAmt_ = 0;
    dotmi_step_stats lastStats_{};
};

}  // namespace dot_amd
