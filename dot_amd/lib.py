"""ctypes binding of libdotmi.so (include/dotmi.h).  No torch types cross this boundary.

The HIP library is the only compute path: if it is missing or no GPU is present every call fails
loudly -- there is no CPU fallback in the product."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DOTMI_LIBRARY: another build of the same ABI (tests load libdotmi_testhooks.so, the build with the fault-injection hook)
LIB_PATH = os.environ.get("DOTMI_LIBRARY") or os.path.join(_HERE, "libdotmi.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_up = C.POINTER(C.c_uint8)

ENERGY_FCR = 0
ENERGY_SNH = 1
FLAG_TIME_BACKSOLVE = 2
FLAG_FORCE_DIST = 4
FLAG_HOST_LOOP = 8
FLAG_TIME_PHASES = 16
FLAG_GSDD = 32
FLAG_NEWTON = 64
FLAG_ASYNC_REFRESH = 128
FLAG_OWNER_EXCHANGE = 256


class Mesh(C.Structure):
    _fields_ = [("nV", C.c_int32), ("nT", C.c_int32), ("X_rest", c_dp), ("T", c_ip), ("mu", c_dp),
                ("lam", c_dp), ("density", C.c_double), ("fixed", c_up), ("epart", c_ip),
                ("nParts", C.c_int32), ("vpart", c_ip)]


class Params(C.Structure):
    _fields_ = [("energy", C.c_int32), ("dt", C.c_double), ("gravity", C.c_double * 3),
                ("relTol", C.c_double), ("history", C.c_int32), ("iterCap", C.c_int32),
                ("alphaMin", C.c_double), ("device", C.c_int32), ("rank", C.c_int32),
                ("world", C.c_int32), ("comm_id", C.c_void_p), ("flags", C.c_int32),
                ("allreduce", C.c_void_p), ("allreduce_ctx", C.c_void_p)]


ALLREDUCE_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int64)


class StepStats(C.Structure):
    _fields_ = [("iters", C.c_int32), ("ls_halvings", C.c_int32), ("energy_evals", C.c_int32),
                ("status", C.c_int32), ("E0", C.c_double), ("g2_0", C.c_double), ("E", C.c_double),
                ("g2", C.c_double), ("ms_total", C.c_double), ("ms_loop", C.c_double),
                ("ms_hessian", C.c_double), ("ms_factor", C.c_double), ("ms_precond", C.c_double),
                ("precond_launches", C.c_int64), ("precond_bytes", C.c_int64),
                ("factor_flops", C.c_double), ("ms_phase", C.c_double * 14),
                ("collective_calls", C.c_int64), ("collective_bytes", C.c_int64), ("ms_collective", C.c_double),
                ("collective_timed", C.c_int64), ("collective_timed_bytes", C.c_int64),
                ("backsolve_launches", C.c_int64), ("backsolve_stopped", C.c_int64),
                ("backsolve_held", C.c_int64), ("backsolve_held_rejected", C.c_int64),
                ("paired_slots", C.c_int64), ("paired_redone", C.c_int64),
                ("spec_slots", C.c_int64), ("spec_redone", C.c_int64)]


# enum dotmi_bench_kind (include/dotmi.h)
BENCH_KERNELS = ["elem_energy_grad", "elem_energy", "vertex_gather", "spmv_dots", "backsolve", "merge", "build_qpad",
                 "build_p", "step_forward", "elem_hessian", "assemble",
                 # the forms the device loop's early order launches (after at least one step)
                 "spmv_zp", "merge_early", "elem_step", "gather_early", "dirstep", "elem_vertex"]

EXPORTS = [
    "dotmi_create", "dotmi_destroy", "dotmi_last_error", "dotmi_comm_unique_id", "dotmi_comm_ranks", "dotmi_set_state",
    "dotmi_get_state", "dotmi_set_dirichlet", "dotmi_refix", "dotmi_step", "dotmi_last_iter_log",
    "dotmi_target_gres", "dotmi_eval_energy", "dotmi_eval_gradient", "dotmi_eval_elem_hessians",
    "dotmi_refactor", "dotmi_apply_precond", "dotmi_spmv", "dotmi_get_features", "dotmi_part_size", "dotmi_padded_size",
    "dotmi_part_matrix", "dotmi_factor_storage_bytes", "dotmi_factor_kind", "dotmi_backsolve_form", "dotmi_plan_backsolve_form", "dotmi_probe_direction", "dotmi_bench_precond", "dotmi_bench_energy", "dotmi_bench_kernel", "dotmi_plan_shards", "dotmi_plan_layout", "dotmi_plan_tile_schedule", "dotmi_plan_tile_schedule_two_level", "dotmi_plan_tile_deps", "dotmi_plan_backsolve_tiles", "dotmi_plan_patches", "dotmi_plan_vpatches", "dotmi_plan_rank", "dotmi_partition",
]

_lib = None


class DotmiError(RuntimeError):
    pass


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DotmiError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(or make -C dot_amd/csrc). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    L.dotmi_create.argtypes = [C.POINTER(Mesh), C.POINTER(Params), c_dp, C.POINTER(H)]
    L.dotmi_create.restype = C.c_int
    L.dotmi_destroy.argtypes = [H]
    L.dotmi_destroy.restype = None
    L.dotmi_last_error.argtypes = [H]
    L.dotmi_last_error.restype = C.c_char_p
    L.dotmi_comm_unique_id.argtypes = [C.c_void_p]
    if hasattr(L, "dotmi_comm_ranks"):     # (a DOTMI_LIBRARY build from before round 5 lacks the two measurement entries)
        L.dotmi_comm_ranks.argtypes = [H]
        L.dotmi_comm_ranks.restype = C.c_int32
    L.dotmi_set_state.argtypes = [H, c_dp, c_dp, c_dp]
    L.dotmi_get_state.argtypes = [H, c_dp, c_dp, c_dp]
    L.dotmi_set_dirichlet.argtypes = [H, C.c_int32, c_ip, c_dp]
    L.dotmi_refix.argtypes = [H, c_up]
    L.dotmi_step.argtypes = [H, C.POINTER(StepStats)]
    L.dotmi_last_iter_log.argtypes = [H, C.c_int32, c_dp, c_dp, c_dp]
    L.dotmi_target_gres.argtypes = [H]
    L.dotmi_target_gres.restype = C.c_double
    L.dotmi_eval_energy.argtypes = [H, c_dp, c_dp]
    L.dotmi_eval_gradient.argtypes = [H, c_dp, c_dp]
    L.dotmi_eval_elem_hessians.argtypes = [H, c_dp, c_dp]
    L.dotmi_refactor.argtypes = [H, c_dp]
    L.dotmi_apply_precond.argtypes = [H, c_dp, c_dp]
    L.dotmi_spmv.argtypes = [H, c_dp, c_dp]
    L.dotmi_get_features.argtypes = [H, c_dp, c_dp, c_dp]
    L.dotmi_part_size.argtypes = [H, C.c_int32]
    L.dotmi_part_size.restype = C.c_int32
    L.dotmi_padded_size.argtypes = [H]
    L.dotmi_padded_size.restype = C.c_int32
    L.dotmi_part_matrix.argtypes = [H, C.c_int32, C.c_int, c_dp, c_ip]
    L.dotmi_probe_direction.argtypes = [H, c_dp, C.c_int32, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.dotmi_factor_storage_bytes.argtypes = [H]
    L.dotmi_factor_storage_bytes.restype = C.c_int64
    if hasattr(L, "dotmi_backsolve_form"):   # (round 6 entry; guarded like the ones below)
        L.dotmi_backsolve_form.argtypes = [H]
        L.dotmi_backsolve_form.restype = C.c_int32
    if hasattr(L, "dotmi_factor_kind"):
        L.dotmi_factor_kind.argtypes = [H]
        L.dotmi_factor_kind.restype = C.c_int32
    L.dotmi_bench_precond.argtypes = [H, C.c_int32, c_dp, C.POINTER(C.c_int64)]
    L.dotmi_bench_energy.argtypes = [H, C.c_int32, c_dp, C.POINTER(C.c_int64)]
    L.dotmi_bench_kernel.argtypes = [H, C.c_int32, C.c_int32, c_dp, C.POINTER(C.c_int64)]
    L.dotmi_plan_shards.argtypes = [C.c_int32, c_ip, C.c_int32, c_ip]
    L.dotmi_plan_rank.argtypes = [C.c_int32, C.c_int32, c_ip, c_ip, C.c_int32, C.c_int32, C.c_int32, c_ip, c_ip, c_ip, c_ip,
                                  c_ip, c_ip, c_ip]
    L.dotmi_partition.argtypes = [C.c_int32, C.c_int32, c_ip, c_dp, C.c_int32, c_ip]
    L.dotmi_plan_layout.argtypes = [C.c_int32, C.c_int32, c_ip, c_dp, c_ip, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, c_ip, c_ip, c_ip, c_ip]
    if hasattr(L, "dotmi_plan_backsolve_tiles"):   # (round 5 entry; guarded like the two above -- ADVICE r05)
        L.dotmi_plan_backsolve_tiles.argtypes = [C.c_int32, C.c_int32, c_ip, c_dp, c_ip, C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_int32, c_ip, c_ip]
    for name in EXPORTS:
        fn = getattr(L, name, None)
        if fn is None:
            continue
        if fn.restype is C.c_int or name in ("dotmi_create",):
            fn.restype = C.c_int
    _lib = L
    return L


def dp(a: np.ndarray):
    return a.ctypes.data_as(c_dp)


def ip(a: np.ndarray):
    return a.ctypes.data_as(c_ip)


def up(a: np.ndarray):
    return a.ctypes.data_as(c_up)
