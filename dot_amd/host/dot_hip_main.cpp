// dot_hip -- headless runner with the reference's CLI shape (`DOT_bin 100 <script.txt>`, README.md:76-93,
// main.cpp:599-989 in mode 100) on top of libdotmi.so.  Host side only: parses the reference's script
// format, builds the scene, steps with DotHipTimeStepper and writes the reference's output files
//   output/<name>/iterStats.txt   per step "n 0 E |g|^2" then per iteration "n alpha E |g|^2"
//                                 (DOTTimeStepper.cpp:299,304,329; Optimizer.cpp:870)
//   output/<name>/log.txt         "<n>th tol: <targetGRes>" and "Timestep<n> innerIterAmt = ..." lines
//                                 (Optimizer.cpp:227, DOTTimeStepper.cpp:338)
//   output/<name>/status<n>       restartable state (Optimizer.cpp:1096-1132)
//   output/<name>/<n>.obj         surface mesh per step (Optimizer.cpp:1137-1150)
//   output/<name>/config.txt      echo of the effective script (Config::saveToFile, Config.cpp:209-302; main.cpp:786)
//   output/<name>/info.txt        nV nT / steps innerIters / wall-clock summary (main.cpp:338-358)
//   output/<name>/label.obj, wire.poly   partition labels of the surface triangles, surface wire frame
//                                 (ADMMDDTimeStepper.cpp:375-442)
// Script token `restart <status file>` resumes from a saved status (Optimizer.cpp:126-177).
//
// usage: dot_hip 100 <script.txt> [--mesh-root DIR] [--parts N] [--energy FCR|SNH] [--epart raw.i32]
//                [--frames K] [--out DIR] [--device D] [--no-files] [--dump-scene K] [--dump-config] [--dump-formats DIR]
//                [--echo-config FILE] [--fast]
//        dot_hip --write-info FILE nV nT steps iters t0..t21
#include <chrono>
#include <cstring>
#include <iostream>
#include <sys/stat.h>

#include "DotHipTimeStepper.hpp"
#include "Output.hpp"
#include "Scene.hpp"

using namespace dot_amd;

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    // pure formatter (no script, no GPU): info.txt from given numbers -- compared byte for byte with what the reference's
    // Timer::print writes for the same numbers (tests/golden/ref_formats.json)
    if (argc == 3 + 4 + 22 && std::string(argv[1]) == "--write-info") {
        RunTimers rt;
        rt.descent = std::atof(argv[7]);
        for (int k = 0; k < 14; ++k) rt.step[k] = std::atof(argv[8 + k]);
        for (int k = 0; k < 7; ++k) rt.temp3[k] = std::atof(argv[22 + k]);
        write_info_txt(argv[2], std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6]), rt);
        return 0;
    }
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s 100 <script.txt> [--mesh-root DIR] [--parts N] [--energy FCR|SNH] "
                             "[--epart raw.i32] [--frames K] [--out DIR] [--device D] [--no-files] [--dump-scene K] [--dump-config] [--dump-formats DIR] [--fast]\n", argv[0]);
        return 2;
    }
    if (std::string(argv[1]) != "100") {
        std::fprintf(stderr, "only the headless mode 100 exists here (viewer / diagnostics / mesh tools are out of scope)\n");
        return 2;
    }
    const std::string scriptPath = argv[2];
    std::string meshRoot = ".", outDir, epartFile, energyOverride;
    int partsOverride = -1, frames = -1, device = 0, dumpScene = -1;
    bool files = true, dumpConfig = false, fast = false;
    std::string dumpFormats, echoConfig;
    for (int i = 3; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) throw std::runtime_error("missing value for " + a); return argv[++i]; };
        if (a == "--mesh-root") meshRoot = next();
        else if (a == "--parts") partsOverride = std::stoi(next());
        else if (a == "--energy") energyOverride = next();
        else if (a == "--epart") epartFile = next();
        else if (a == "--frames") frames = std::stoi(next());
        else if (a == "--out") outDir = next();
        else if (a == "--device") device = std::stoi(next());
        else if (a == "--no-files") files = false;
        else if (a == "--dump-scene") dumpScene = std::stoi(next());
        else if (a == "--dump-config") dumpConfig = true;
        else if (a == "--dump-formats") dumpFormats = next();   // write 0.obj + info.txt of the initial scene into DIR (no GPU)
        else if (a == "--echo-config") echoConfig = next();   // write config.txt (Config::saveToFile's echo) to this path and exit
        else if (a == "--fast") fast = true;   // device-resident loop: the loop slots of info.txt stay 0
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    try {
        Config cfg = parse_script(scriptPath);
        if (!echoConfig.empty()) {
            write_config_txt(echoConfig, cfg);
            return 0;
        }
        if (dumpConfig) {
            // the parsed fields as "key value" lines, the format of oracle/ref_config.cpp (the reference's own
            // Config::loadFromFile): tests/test_oracle_pin.py compares the two on every input script of the reference
            std::printf("energy %s\ntimeStepper %s\npartitionAmt %d\nblockSize %d\n", cfg.energy.c_str(),
                        cfg.timeStepper.c_str(), cfg.partitionAmt, cfg.blockSize);
            std::printf("size %.17g\nduration %.17g\ndt %.17g\nrho %.17g\nYM %.17g\nPR %.17g\n", cfg.size, cfg.duration,
                        cfg.dt, cfg.rho, cfg.YM, cfg.PR);
            std::printf("withGravity %d\ninputShapePath %s\nwarmStart %d\nhandleRatio %.17g\nrotDeg %.17g\n",
                        cfg.withGravity ? 1 : 0, cfg.shapePath.c_str(), cfg.warmStart, cfg.handleRatio, cfg.rotDeg);
            if (cfg.rotDeg != 0.0) std::printf("rotAxis %.17g %.17g %.17g\n", cfg.rotAxis[0], cfg.rotAxis[1], cfg.rotAxis[2]);
            std::printf("restart %d\nstatusPath %s\ntol %zu", cfg.restart ? 1 : 0, cfg.statusPath.c_str(), cfg.tol.size());
            for (double t : cfg.tol) std::printf(" %.17g", t);
            std::printf("\n");
            return 0;
        }
        if (!energyOverride.empty()) cfg.energy = energyOverride;
        if (cfg.shapePath.empty()) throw std::runtime_error("script has no `shape input <mesh>` (primitive shapes are 2-D only)");
        TetMesh mesh = load_tet_mesh(cfg.shapePath[0] == '/' ? cfg.shapePath : meshRoot + "/" + cfg.shapePath);
        normalize(mesh.V, cfg.size, cfg.rotDeg, cfg.rotAxis);
        std::vector<int> border[2];
        find_border_verts(mesh.V, cfg.handleRatio, border);
        AnimScripter scripter(cfg.script, mesh.V, border);
        std::vector<double> x0 = scripter.initial_positions(mesh.V);

        // `timeStepper LBFGSH` (LBFGSTimeStepper with D0T_H, LBFGSTimeStepper.cpp:196-262, :338-420): L-BFGS whose initial
        // inverse Hessian is the factored GLOBAL projected Hessian and whose line search starts from step 1 -- this
        // path with the whole mesh as ONE subdomain (no averaging: dup = 1) and the alpha_0 clamp at 1
        const bool newton = cfg.timeStepper == "Newton";   // projected Newton: one subdomain, DOTMI_FLAG_NEWTON
        const bool lbfgsH = cfg.timeStepper == "LBFGSH" || newton;
        int nParts = partsOverride > 0 ? partsOverride : cfg.partitionAmt;
        if (lbfgsH) nParts = 1;
        if (cfg.blockSize > 0 && partsOverride <= 0) nParts = mesh.nV() / cfg.blockSize + 1;  // main.cpp:792-798
        if (nParts < 2 && !lbfgsH) nParts = 4;
        std::vector<int32_t> epart;
        if (lbfgsH) {
            epart.assign(mesh.nT(), 0);
        } else if (!epartFile.empty()) {
            std::ifstream f(epartFile, std::ios::binary);
            epart.resize(mesh.nT());
            f.read((char *)epart.data(), sizeof(int32_t) * epart.size());
            if (!f) throw std::runtime_error("cannot read " + epartFile);
        } else {
            // METIS is third-party: pass --epart for the reference's partition; otherwise the library's own partitioner
            epart.resize(mesh.nT());
            if (dotmi_partition(mesh.nV(), mesh.nT(), mesh.T.data(), mesh.V.data(), nParts, epart.data()) != 0)
                throw std::runtime_error("dotmi_partition failed");
        }
        std::string name = scriptPath.substr(scriptPath.find_last_of('/') + 1);
        name = name.substr(0, name.find_last_of('.'));
        const bool outGiven = !outDir.empty();
        if (outDir.empty()) outDir = "output/" + name;

        if (!dumpFormats.empty()) {
            mkdir(dumpFormats.c_str(), 0755);
            const SurfaceMesh S = build_surface_mesh(mesh);
            write_surface_obj(dumpFormats + "/0.obj", x0, S.surfIndToTet, S.F_surf);
            RunTimers rt;   // a fixed pattern so a reader can check which value sits under which name
            rt.descent = 12.5;
            for (int k = 0; k < 14; ++k) rt.step[k] = 0.001 * (k + 1);
            write_info_txt(dumpFormats + "/info.txt", mesh.nV(), mesh.nT(), 7, 123, rt);
            write_partition_files(dumpFormats, mesh, x0, epart);
            return 0;
        }
        // --dump-scene K: print the scene the hot path would receive and K scripted moves (no GPU needed);
        // with --out also the partition files
        if (dumpScene >= 0) {
            int nfixed = 0;
            for (auto f : scripter.fixed) nfixed += f;
            std::printf("scene nV %d nT %d nfixed %d energy %s dt %.17g\n", mesh.nV(), mesh.nT(), nfixed, cfg.energy.c_str(), cfg.dt);
            std::vector<double> x = x0, pos;
            std::vector<int32_t> idx;
            for (int k = 0; k < dumpScene; ++k) {
                scripter.step(x, cfg.dt, idx, pos);
                double sum[3] = {0, 0, 0};
                for (size_t i = 0; i < idx.size(); ++i)
                    for (int d = 0; d < 3; ++d) { x[3 * idx[i] + d] = pos[3 * i + d]; sum[d] += pos[3 * i + d]; }
                std::printf("move %d n %zu sum %.17g %.17g %.17g\n", k, idx.size(), sum[0], sum[1], sum[2]);
            }
            if (outGiven) {
                mkdir(outDir.c_str(), 0755);
                write_partition_files(outDir, mesh, x0, epart);
            }
            if (cfg.restart) {
                int t = 0;
                std::vector<double> xs, vs;
                read_status(cfg.statusPath, mesh.nV(), t, xs, vs);
                double sx = 0, sv = 0;
                for (double c : xs) sx += c;
                for (double c : vs) sv += c;
                std::printf("restart timestep %d sumx %.17g sumv %.17g\n", t, sx, sv);
            }
            return 0;
        }

        double mu, lam;
        lame(cfg.YM, cfg.PR, mu, lam);
        std::vector<double> u(mesh.nT(), mu), lambda(mesh.nT(), lam);
        MeshView mv;
        mv.nV = mesh.nV(); mv.nT = mesh.nT(); mv.V_rest = mesh.V.data(); mv.F = mesh.T.data();
        mv.u = u.data(); mv.lambda = lambda.data(); mv.density = cfg.rho; mv.isFixedVert = scripter.fixed.data();
        Options opt;
        opt.energyType = cfg.energy == "SNH" ? DOTMI_ENERGY_SNH : DOTMI_ENERGY_FCR;
        opt.withGravity = cfg.withGravity; opt.partitionAmt = nParts; opt.epart = epart.data(); opt.device = device;
        // the reference keeps its timer_step running all the time; here the per-phase HIP-event brackets cost a
        // host-driven loop, so they are on whenever files are written unless --fast asks for the device-resident loop
        if (files && !fast) opt.flags |= DOTMI_FLAG_TIME_PHASES;
        if (cfg.timeStepper == "GSDD") opt.flags |= DOTMI_FLAG_GSDD;
        if (lbfgsH) opt.alphaMin = 1.0;
        if (newton) opt.flags |= DOTMI_FLAG_NEWTON;
        // `timeStepper LBFGSJH <n>`: block-Jacobi on a vertex partition (the reference takes METIS::partMesh_nodes;
        // without METIS a vertex goes to the lowest-numbered subdomain among its elements) and a unit first step
        std::vector<int32_t> vpart;
        if (cfg.timeStepper == "LBFGSJH") {
            vpart.assign(mesh.nV(), nParts);
            for (int e = 0; e < mesh.nT(); ++e)
                for (int k = 0; k < 4; ++k) vpart[mesh.T[4 * e + k]] = std::min(vpart[mesh.T[4 * e + k]], epart[e]);
            opt.vpart = vpart.data();
            opt.alphaMin = 1.0;
        }   // `timeStepper GSDD <n>`: the Gauss-Seidel sibling

        FILE *fIter = nullptr, *fLog = nullptr;
        if (files) {
            mkdir("output", 0755);
            mkdir(outDir.c_str(), 0755);
            fIter = std::fopen((outDir + "/iterStats.txt").c_str(), "w");
            fLog = std::fopen((outDir + "/log.txt").c_str(), "w");
            if (!fIter || !fLog) throw std::runtime_error("cannot write into " + outDir);
            write_partition_files(outDir, mesh, x0, epart);
            write_config_txt(outDir + "/config.txt", cfg);   // main.cpp:786
        }
        const SurfaceMesh surf = files ? build_surface_mesh(mesh) : SurfaceMesh();
        RunTimers timers;

        const double tSetup = now_s();
        DotHipTimeStepper ts(mv, opt, x0.data());
        ts.setTime(cfg.duration, cfg.dt);
        ts.setRelGL2Tol(cfg.tol.empty() ? 1.0e-5 : cfg.tol[0]);
        ts.setScript([&](const std::vector<double> &x, double dt, std::vector<int32_t> &idx, std::vector<double> &pos,
                         std::vector<uint8_t> &fx) {
            const int changed = scripter.step(x, dt, idx, pos);
            if (changed) fx = scripter.fixed;
            return changed;
        });
        ts.precompute();
        timers.step[DOTMI_T_SYMBOLIC_FACTORIZATION] = now_s() - tSetup;   // pattern + layout work of the setup (analyze_pattern's role)
        int firstFrame = 0;
        if (cfg.restart) {
            std::vector<double> xs, vs;
            read_status(cfg.statusPath, mesh.nV(), firstFrame, xs, vs);
            ts.restoreState(xs, vs, firstFrame);
            std::printf("restarted from %s at time step %d\n", cfg.statusPath.c_str(), firstFrame);
        }
        std::printf("setup %.3f s, nV %d nT %d, %d subdomains, tol %.6e\n", now_s() - tSetup, mesh.nV(), mesh.nT(), nParts, ts.getTargetGRes());

        const int nFrames = frames > 0 ? frames : (int)(cfg.duration / cfg.dt);
        long lineSearch = 0;
        double tStep = 0;
        std::vector<double> al(10001), En(10001), g2(10001);
        for (int n = firstFrame; n < nFrames; ++n) {
            if (files) {
                char buf[512];
                std::snprintf(buf, sizeof(buf), "%s/status%d", outDir.c_str(), n);
                ts.saveStatus(buf);
                std::snprintf(buf, sizeof(buf), "%s/%d.obj", outDir.c_str(), n);
                write_surface_obj(buf, ts.getResult(), surf.surfIndToTet, surf.F_surf);
                std::fprintf(fLog, "%dth tol: %g\n", n, ts.getTargetGRes());
            }
            const double t0 = now_s();
            const int rc = ts.solve(1);
            tStep += now_s() - t0;
            if (rc == 1) break;
            const auto &st = ts.lastStats();
            lineSearch += st.ls_halvings;
            for (int k = 0; k < DOTMI_T_COUNT; ++k) timers.step[k] += 1e-3 * st.ms_phase[k];
            if (files) {
                const int k = dotmi_last_iter_log(ts.handle(), 10001, al.data(), En.data(), g2.data());
                std::fprintf(fIter, "%d 0 %g %g\n", n, st.E0, st.g2_0);
                for (int i = 0; i < k; ++i) std::fprintf(fIter, "%d %g %g %g\n", n, al[i], En[i], g2[i]);
                std::fprintf(fLog, "Timestep%d innerIterAmt = %d, accumulated line search steps %ld\n", n, ts.getInnerIterAmt(), lineSearch);
                if (rc == 2) std::fprintf(fLog, "!!! maxIter reached for timeStep%d\n", n);
            }
            std::printf("FRAME %d ms %.3f iters %d halvings %d E %.17g status %d\n", n, st.ms_total, st.iters, st.ls_halvings, st.E, rc);
        }
        if (files) {
            timers.descent = tStep;
            write_info_txt(outDir + "/info.txt", mesh.nV(), mesh.nT(), ts.getIterNum(), ts.getInnerIterAmt(), timers);
            std::fclose(fIter);
            std::fclose(fLog);
        }
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "dot_hip: %s\n", e.what());
        return 1;
    }
}
