"""Golden vectors produced by the REFERENCE's own code (oracle/_ref/librefpin.so, compiled from
/root/reference/src by `make -C oracle ref`).  Run HERE:  python tests/golden/make_ref_vectors.py
Stored: inputs and the reference's outputs only (data), in tests/golden/ref_vectors.npz.

  svd_*      IglUtils::computeSVD_SIMD path (Utils/SVD_EFTYCHIOS, AVX2)         -> U, S, V
  psi_*/phat_*  ENERGY_* / PHAT_* macros of Utils/SIMD_DOUBLE_MACROS.hpp          (FCR, SNH)
  pd3_*      IglUtils::makePD<double,3>    (Utils/IglUtils.hpp:253)
  pd2_*      IglUtils::makePD2d<double,2>  (Utils/IglUtils.hpp:271)
  hess_*     two IglUtils::dF_div_dx_mult passes around a symmetric 9x9 (Energy.cpp:767-769)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tests import oracle_py as O  # noqa: E402

R = O.ref()
dp = O._dp
rng = np.random.default_rng(20240929)
out = {}

# ---- SVD: mirrors the reference's micro-benchmark (MeshProcessing.hpp:317-374: F = I + Rand/2),
# plus rest state, inverted, rank-deficient and near-equal singular values
n = 512
F = np.eye(3)[None] + (rng.random((n, 3, 3)) - 0.5)
F[0] = np.eye(3)
F[1] = np.diag([1.0, 1.0, -1.0])                       # inverted
F[2] = np.diag([2.0, 0.5, 0.0])                        # rank 2
F[3] = np.eye(3) + 1e-9 * rng.standard_normal((3, 3))  # near rest
F[4] = np.diag([1.3, 1.3, 0.7]) @ np.linalg.qr(rng.standard_normal((3, 3)))[0]
F[5] = 0.0
for k in range(6, 40):                                  # inverted random
    F[k, :, 0] *= -1
F = np.ascontiguousarray(F)
U = np.zeros_like(F); S = np.zeros((n, 3)); V = np.zeros_like(F)
assert R.ref_svd(n, dp(F), dp(U), dp(S), dp(V)) == 0
out.update(svd_F=F, svd_U=U, svd_S=S, svd_V=V)

# ---- energies and P-hat (E=100, nu=0.4 as in Energy::unitTest_*, Energy.cpp:1283; plus 1e5)
m = 256
sig = 1.0 + 0.5 * (rng.random((m, 3)) - 0.5)
sig[0] = 1.0
sig[1] = 0.0
sig[2] = [1.2, 0.9, -0.4]
sig[3] = [1.0, 1.0, 1.0 + 1e-9]
YM = np.where(np.arange(m) % 2 == 0, 100.0, 1e5); PR = 0.4
mu = YM / 2 / (1 + PR); lam = YM * PR / (1 + PR) / (1 - 2 * PR)
sig = np.ascontiguousarray(sig)
for mat, name in ((0, "fcr"), (1, "snh")):
    psi = np.zeros(m); ph = np.zeros((m, 3))
    assert R.ref_energy_phat(mat, m, dp(mu), dp(lam), dp(sig), dp(psi), dp(ph)) == 0
    out[f"psi_{name}"] = psi
    out[f"phat_{name}"] = ph
out.update(mat_sigma=sig, mat_mu=mu, mat_lam=lam)

# ---- PSD clamps
k = 128
A3 = rng.standard_normal((k, 3, 3)); A3 = 0.5 * (A3 + A3.transpose(0, 2, 1))
A3[:16] += 4 * np.eye(3)                                # already PD -> untouched
A3[16] = np.diag([1.0, -2.0, 3.0])
A3 = np.ascontiguousarray(A3); P3 = A3.copy()
for i in range(k):
    R.ref_make_pd3(dp(P3[i]))
B2 = rng.standard_normal((k, 2, 2)); B2 = 0.5 * (B2 + B2.transpose(0, 2, 1))
B2[0] = [[1.0, 1.0], [1.0, 1.0]]                        # rest-state B block: eigenvalues (2, 0)
B2[1] = [[1.0 - 1e-17, 1.0], [1.0, 1.0 - 1e-17]]
B2[2] = [[-1.0, 0.0], [0.0, -2.0]]
B2[3] = [[2.0, 0.0], [0.0, -1.0]]
B2 = np.ascontiguousarray(B2); P2 = B2.copy()
for i in range(k):
    R.ref_make_pd2(dp(P2[i]))
out.update(pd3_in=A3, pd3_out=P3, pd2_in=B2, pd2_out=P2)

# ---- H = G M G^T through the reference's dF_div_dx_mult
h = 32
M9 = rng.standard_normal((h, 9, 9)); M9 = np.ascontiguousarray(0.5 * (M9 + M9.transpose(0, 2, 1)))
Ai = np.ascontiguousarray(rng.standard_normal((h, 3, 3)))
H = np.zeros((h, 12, 12))
for i in range(h):
    R.ref_hessian_from_dPdF(dp(M9[i]), dp(Ai[i]), dp(H[i]))
out.update(hess_M=M9, hess_A=Ai, hess_H=H)

np.savez_compressed(os.path.join(os.path.dirname(__file__), "ref_vectors.npz"), **out)
print({k: v.shape for k, v in out.items()})
