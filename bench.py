#!/usr/bin/env python
"""bench.py -- ms/time-step (and L-BFGS iterations/frame) of the DOT hot path on N MI355X.

A "step" = one backward-Euler time step of the workload: scripted handle move + dotmi_step
(L-BFGS-H solve to the reference's tolerance + Hessian refresh + subdomain refactorisation).
State, mesh, Hessians and factors are resident in HBM before the timed region starts; per step the
host hands over only the scripted handle positions (~24 KB) and reads back a few hundred bytes of
reduction partials per line-search trial.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant hand-written kernel pair (subdomain back-solve), algorithmic bytes /
                measured HIP-event time on the library's stream, against 8 TB/s
  cpu_baseline  the CPU oracle (a port of the reference algorithm, oracle/dot_oracle.c) timed on the
                host cores of this box on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_WORKLOAD = "bar17K_twist"  # BASELINE.json configs[1]
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=12)
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from dot_amd import lib as dl
    from dot_amd.configs import WORKLOADS, load_workload
    from dot_amd.timestepper import DOTTimeStepper, comm_unique_id

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    comm_id = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl")
        buf = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            buf.copy_(torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(buf, 0)
        comm_id = bytes(buf.cpu().numpy().tobytes())

    sc, ep, nparts = load_workload(args.workload)
    cfg = sc.cfg
    ts = DOTTimeStepper(sc, ep, nparts, device=local_rank, rank=rank, world=world, comm_id=comm_id,
                        flags=dl.FLAG_TIME_BACKSOLVE)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # scripted (Dirichlet) vertices are never moved by the solver: the scripter keeps their positions itself, as the
    # reference's host mesh does, instead of reading all positions back every step
    cached = cfg.script != "rubberBandPull"
    if cached:
        sc.scripter.track(sc.x0)

    def one_step():
        x = None if cached else ts.getResult()
        idx, pos = sc.scripter.step(x, cfg.dt)
        if idx.size:
            ts.setDirichlet(idx, pos)
        return ts.step()

    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    stats = [one_step() for _ in range(args.steps)]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = 1e3 * elapsed / args.steps
    iters = [s.iters for s in stats]
    # ---- roofline of the dominant hand-written kernel, measured inside the timed region ------------------
    pre_ms = sum(s.ms_precond for s in stats)
    pre_n = sum(s.precond_launches for s in stats)
    bytes_per_launch = stats[0].precond_bytes            # 8 x structural non-zeros of the inverse factors of THIS rank's parts
    avg_ms = pre_ms / max(pre_n, 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if pre_n else 0.0
    # HBM traffic of one back-solve from the PMC counters (FETCH_SIZE / WRITE_SIZE need their own rocprofv3
    # passes, so they are collected separately on the same kernel + workload and committed under profiles/)
    traffic = None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_backsolve_pmc.json")))
    pmc = pmcs[-1] if pmcs else ""
    if world == 1 and pmc:
        with open(pmc) as f:
            rec = json.load(f)
        if rec.get("workload") == args.workload:
            traffic = rec["hbm_bytes_per_backsolve"]
    roofline = {
        "bound": "hbm", "kernel": "backsolve_kernel: subdomain back-solve p_s = X_s^T (X_s r_s), nested-dissection block-sparse inverse factors",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "algorithmic_bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(avg_ms, 5),
        # every 8th back-solve of the timed region is bracketed with HIP events (an event record costs ~6 us
        # of stream time, so bracketing all of them would inflate the metric by ~3%)
        "launches_timed": int(pre_n), "launches_total": int(sum(iters)),
        "share_of_step_time": round(avg_ms * sum(iters) / (1e3 * elapsed), 3),
    }

    # ---- second roofline: the once-per-step factorisation against the dense FP64 matrix-core peak ----------------
    FP64_MFMA_PEAK = 78.6   # TFLOP/s, MI355X FP64 matrix = vector peak (256 CUs x 4 SIMD x 16 lanes x 2 x 2.4 GHz)
    fact_ms = float(np.mean([s.ms_factor for s in stats]))
    fact_tf = stats[0].factor_flops / (fact_ms * 1e-3) / 1e12 if fact_ms > 0 else 0.0
    roofline_factor = {
        "bound": "mfma", "kernel": "block-sparse inverse-Cholesky of the subdomain blocks (rocBLAS dgemm_strided_batched "
        "+ chol_inv_node128 / chol_inv_base), once per step",
        "achieved": round(fact_tf, 2), "peak": FP64_MFMA_PEAK, "unit": "TFLOP/s", "frac": round(fact_tf / FP64_MFMA_PEAK, 4),
        "flop_per_factorisation": float(stats[0].factor_flops), "avg_ms": round(fact_ms, 4),
        "note": "flop as executed (identity padding included); the phase is bound by chains of small dependent kernels, "
                "see profiles/r01_factor_experiments.txt",
    }

    out = None
    if rank == 0:
        out = {
            "metric": "ms_per_time_step", "value": round(ms_per_step, 3), "unit": "ms", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "mesh fixture tests/golden/meshes (reference input mesh), scripted handles",
            "config": {
                "workload": args.workload, "nV": int(sc.V_rest.shape[0]), "nT": int(sc.T.shape[0]),
                "energy": cfg.energy, "subdomains": int(nparts), "dt": cfg.dt, "script": cfg.script,
                "rel_tol": 1e-5, "target_gres": ts.targetGRes,
                "parallelism": f"{nparts} subdomains sharded over {world} GPU(s), RCCL all-reduce" if world > 1
                               else f"{nparts} subdomains on 1 GPU",
            },
            "iters_per_frame": round(float(np.mean(iters)), 2), "iters": iters,
            "step_breakdown_ms": {
                "lbfgs_loop": round(float(np.mean([s.ms_loop for s in stats])), 3),
                "hessian_assembly": round(float(np.mean([s.ms_hessian for s in stats])), 3),
                "subdomain_factor": round(float(np.mean([s.ms_factor for s in stats])), 3),
                "back_solve_kernels": round(avg_ms * float(np.mean(iters)), 3),
            },
            "roofline": roofline,
            "roofline_factor": roofline_factor,
        }
        # ---- CPU baseline on this box's host cores: bounded sample of the same workload ---------------
        if not args.no_cpu_baseline and world == 1:
            from tests import oracle_py as O
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            O.lib().dor_set_threads(threads)
            sc2, ep2, _ = load_workload(args.workload)
            orc = O.OracleSim(sc2.V_rest, sc2.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc2.fixed, sc2.x0,
                              ep2, nparts, cfg.with_gravity)
            nwarm = min(args.warmup, 2)
            times, cits = [], []
            budget_t0 = time.perf_counter()
            for k in range(nwarm + args.cpu_steps):
                x = orc.state()[0]
                idx, pos = sc2.scripter.step(x, cfg.dt)
                orc.move(idx, pos)
                c0 = time.perf_counter()
                so = orc.step()
                if k >= nwarm:
                    times.append(time.perf_counter() - c0)
                    cits.append(so.iters)
                if time.perf_counter() - budget_t0 > 30.0 and len(times) >= 3:
                    break
            out["cpu_baseline"] = {
                "value": round(1e3 * float(np.mean(times)), 2), "unit": "ms", "cores": threads, "kind": "port",
                "sample": f"steps {nwarm}..{nwarm + len(times) - 1} of {args.workload} (same partition, same tolerance), "
                          f"oracle/dot_oracle.c with OpenMP, iters/step {cits}",
            }
        print(json.dumps(out), flush=True)
    ts.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
