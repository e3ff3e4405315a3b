"""how many entries of merge_tiles_early's lists (one per back-solve tile whose column range holds the dof) would
summing inside the tile packs remove?  Counted from the product's own job table (dotmi_plan_backsolve_tiles, host only):

  entries / dof            today: every tile of the subdomain whose columns [cb, r0 + rows) hold the dof
  ... from pack members    the tiles of at most BS_WAVE = 256 columns (four per workgroup, one wavefront each)
  ... if summed            one partial row per (subdomain, region): the pack members of one region share cb, so a planner that packs
                           them together and sums their rows in LDS leaves ONE entry where they left several

    python tools/count_merge_entries.py bar17K_twist monkey18K_stiff ...
"""
import sys

import numpy as np

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from test_host_logic import _plan_bs_tiles

for name in sys.argv[1:] or ["bar17K_twist"]:
    tiles, c, nodes, nmax, pos, n = _plan_bs_tiles(name)
    part, r0, rows, cb, idx, job = tiles.T
    small = (r0 + rows - cb) <= 256
    tot = sm = summed = live = 0
    for p in range(n):
        m = part == p
        cov = np.zeros(nmax, int)
        covs = np.zeros(nmax, int)
        covr = np.zeros(nmax, int)
        ends = {}
        for a, k, b, s in zip(r0[m], rows[m], cb[m], small[m]):
            cov[b:a + k] += 1
            if s:
                covs[b:a + k] += 1
                ends[b] = max(ends.get(b, 0), a + k)
        for b, e in ends.items():
            covr[b:e] += 1
        lv = np.zeros(nmax, bool)
        for q in pos[p]:
            lv[q:q + 3] = True
        tot += cov[lv].sum(); sm += covs[lv].sum(); summed += covr[lv].sum(); live += lv.sum()
    print(f"{name}: {int(small.sum())} pack members + {int((~small).sum())} one-tile jobs; entries per (subdomain, dof) {tot / live:.2f}, "
          f"of them from pack members {sm / live:.2f}; with one summed row per region {summed / live:.2f} "
          f"-> {(tot - sm + summed) / live:.2f} ({100 * (sm - summed) / tot:.1f} % fewer)")
