"""How the path shards over the GPUs of one node.

Subdomains are the unit: rank r owns a contiguous group of parts, the elements of those parts
(energy / gradient contributions), the dense factors of those parts (back-solve), and a contiguous
vertex slice (inertia terms, rows of the global SpMV).  Vectors are replicated; the only exchange
steps are sum-all-reduces of p, of [g ; E] and of the two scalars of alpha_0.
This module is the Python mirror of dotmi_plan_shards (dot_amd/csrc/dotmi_create.hip) plus the index sets
that follow from it; tests check the two agree."""
from __future__ import annotations

import bisect
from typing import List, Tuple

import numpy as np


def part_scalar_sizes(T: np.ndarray, epart: np.ndarray, nparts: int) -> np.ndarray:
    """3 * (#vertices of the elements of each part)   (ADMMDDTimeStepper.cpp:161-193)"""
    return np.array([3 * np.unique(T[epart == p]).size for p in range(nparts)], dtype=np.int32)


def plan_shards(psize: np.ndarray, world: int) -> List[int]:
    cost = [0.0]
    for n in psize:
        cost.append(cost[-1] + float(n) * float(n))
    nP = len(psize)
    first = [0] * (world + 1)
    first[world] = nP
    for r in range(1, world):
        target = cost[nP] * r / world
        c = bisect.bisect_left(cost, target)
        if c > 0 and target - cost[c - 1] < cost[c] - target:
            c -= 1
        first[r] = min(max(c, first[r - 1]), nP)
    return first


def owned_elements(epart: np.ndarray, first: List[int], rank: int) -> np.ndarray:
    return np.nonzero((epart >= first[rank]) & (epart < first[rank + 1]))[0].astype(np.int32)


def vertex_slice(nV: int, rank: int, world: int) -> Tuple[int, int]:
    return nV * rank // world, nV * (rank + 1) // world


def plan_layout(V_rest: np.ndarray, T: np.ndarray, epart: np.ndarray, nparts: int, p0: int = 0, p1: int = -1,
                levels: int = -1, min_split: int = -1):
    """Nested-dissection layout of the dense blocks of parts [p0,p1) -- dotmi_plan_layout (host only).
    Returns (nodes[n,6] = off,size,childA,childC,offS,sizeS ; nmax ; [pos per part], [verts per part])."""
    import ctypes as C

    from .lib import load

    L = load()
    p1 = nparts if p1 < 0 else p1
    T = np.ascontiguousarray(T, dtype=np.int32)
    V = np.ascontiguousarray(V_rest, dtype=np.float64)
    ep = np.ascontiguousarray(epart, dtype=np.int32)
    verts = [np.unique(T[ep == p]).astype(np.int32) for p in range(p0, p1)]
    pos = np.zeros(max(1, sum(v.size for v in verts)), dtype=np.int32)
    nodes = np.zeros((256, 6), dtype=np.int32)
    nn, nmax = C.c_int32(0), C.c_int32(0)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    rc = L.dotmi_plan_layout(V.shape[0], T.shape[0], ip(T), V.ctypes.data_as(C.POINTER(C.c_double)), ip(ep), nparts,
                             p0, p1, levels, min_split, 256, ip(nodes), C.byref(nn), C.byref(nmax), ip(pos))
    if rc != 0:
        raise ValueError(f"dotmi_plan_layout failed ({rc})")
    out, base = [], 0
    for v in verts:
        out.append(pos[base:base + v.size].copy())
        base += v.size
    return nodes[:nn.value].copy(), int(nmax.value), out, verts
