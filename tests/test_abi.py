"""The C-ABI shared library: builds, loads, exports exactly what include/dotmi.h declares, and fails
loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dot_amd import lib as dl
from tests.workloads import load_workload
from dot_amd.sharding import part_scalar_sizes, plan_shards

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "dotmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dotmi_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(dl.EXPORTS) == names


def test_library_exports_every_declared_symbol():
    L = dl.load()
    for name in declared_functions():
        assert hasattr(L, name), name


def test_struct_layouts_match_header(tmp_path):
    """Compile the header with gcc and compare sizeof/offsetof with the ctypes mirrors."""
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text("""
#include <stddef.h>
#include <stdio.h>
#include "dotmi.h"
int main(void){
  printf("%zu %zu %zu %zu\\n", sizeof(dotmi_mesh), offsetof(dotmi_mesh, density), offsetof(dotmi_mesh, epart), offsetof(dotmi_mesh, nParts));
  printf("%zu %zu %zu %zu %zu\\n", sizeof(dotmi_params), offsetof(dotmi_params, gravity), offsetof(dotmi_params, alphaMin), offsetof(dotmi_params, comm_id), offsetof(dotmi_params, flags));
  printf("%zu %zu %zu %zu\\n", sizeof(dotmi_step_stats), offsetof(dotmi_step_stats, E0), offsetof(dotmi_step_stats, ms_precond), offsetof(dotmi_step_stats, precond_bytes));
  return 0; }
""")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    rows = [[int(t) for t in l.split()] for l in subprocess.check_output([str(exe)]).decode().splitlines()]
    assert rows[0] == [C.sizeof(dl.Mesh), dl.Mesh.density.offset, dl.Mesh.epart.offset, dl.Mesh.nParts.offset]
    assert rows[1] == [C.sizeof(dl.Params), dl.Params.gravity.offset, dl.Params.alphaMin.offset,
                       dl.Params.comm_id.offset, dl.Params.flags.offset]
    assert rows[2] == [C.sizeof(dl.StepStats), dl.StepStats.E0.offset, dl.StepStats.ms_precond.offset,
                       dl.StepStats.precond_bytes.offset]


def test_plan_shards_host_entry_point_matches_python_mirror():
    L = dl.load()
    sc, ep, nparts = load_workload("bar17K_twist")
    ps = part_scalar_sizes(sc.T, ep, nparts)
    for world in (1, 2, 3, 4, 8):
        first = np.zeros(world + 1, dtype=np.int32)
        assert L.dotmi_plan_shards(nparts, dl.ip(ps), world, dl.ip(first)) == 0
        assert list(first) == plan_shards(ps, world)
    assert L.dotmi_plan_shards(nparts, dl.ip(ps), 0, dl.ip(np.zeros(2, dtype=np.int32))) == -1


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a machine without a GPU")
def test_create_fails_loudly_without_gpu():
    from dot_amd.timestepper import DOTTimeStepper
    sc, ep, nparts = load_workload("synbar:4x2x2:2")
    with pytest.raises(dl.DotmiError) as e:
        DOTTimeStepper(sc, ep, nparts)
    assert "-4" in str(e.value) and "no CPU fallback" in str(e.value)


def test_invalid_arguments_are_rejected_before_touching_a_device():
    L = dl.load()
    h = C.c_void_p()
    assert L.dotmi_create(None, None, None, C.byref(h)) in (-1, -4)
    assert L.dotmi_create(None, None, None, None) == -1
    assert L.dotmi_step(None, None) == -1
    assert L.dotmi_target_gres(None) == 0.0


def test_cpp_adapter_builds_links_and_fails_loudly_or_steps(tmp_path):
    """dot_amd/host/DotHipTimeStepper.hpp (the C++ mirror of the Optimizer surface) compiles with plain
    g++ (no HIP headers), links against libdotmi.so, and either steps (GPU box) or reports the ABI's
    no-CPU-fallback error (exit 3)."""
    import subprocess
    exe = tmp_path / "adapter_check"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", os.path.join(ROOT, "dot_amd", "host", "adapter_check.cpp"),
                           "-o", str(exe), "-L" + os.path.join(ROOT, "dot_amd"), "-ldotmi",
                           "-Wl,-rpath," + os.path.join(ROOT, "dot_amd")])
    rc = subprocess.call([str(exe)])
    assert rc == (0 if _has_gpu() else 3)
