"""How the path shards over the GPUs of one node.

Subdomains are the unit: rank r owns a contiguous group of parts, the elements of those parts
(energy / gradient contributions), the dense factors of those parts (back-solve), and a contiguous
vertex slice (inertia terms, rows of the global SpMV).  Vectors are replicated; the only exchange
steps are sum-all-reduces of p, of [g ; E] and of the two scalars of alpha_0.
This module is the Python mirror of dotmi_plan_shards (dot_amd/csrc/dotmi.hip) plus the index sets
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
