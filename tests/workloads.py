"""The workload table lives in the package (dot_amd/workloads.py); the tests keep importing it from here."""
from dot_amd.workloads import *  # noqa: F401,F403
from dot_amd.workloads import MESH_DIR, PART_DIR, WORKLOADS, load_workload  # noqa: F401
