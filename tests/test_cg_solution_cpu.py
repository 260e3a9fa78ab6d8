"""
The host algebra of the short-recurrence sibk (eigd_amd.adjoint._cg_solution_coefficients) without a GPU: conjugate
gradients in the inner product of an SPD "factor" F on C = I - alpha K F (the operator of
eigenvector_derivatives.py:1246-1252, 1264-1269 without the projector), run in the three-term form the device kernels
implement (csrc/krylov.hip); the solution formed once at the end from the kept z = F r and the (gam, rho) log must be the
iterate of the solution's own recurrence, solve the system, and have positive coefficients.
"""
import numpy as np

from eigd_amd.adjoint import _cg_solution_coefficients


def _three_term_cg(F, K, alpha, b, steps, stop_after=None):
    """(psi by its own recurrence, z history, log) -- the updates of cg_coef_kernel / cg_update_kernel, one column"""
    n = b.shape[0]
    r, r_old = b.copy(), np.zeros(n)
    psi, psi_old = np.zeros(n), np.zeros(n)
    log = np.zeros((2 * (steps + 2), 64))
    Z = np.zeros((steps, n))
    rr_p = gam_p = rho_p = None
    for j in range(steps):
        moves = stop_after is None or j < stop_after
        z = F @ r
        y = K @ z
        Z[j] = z
        if not moves:
            log[2 * j, 0], log[2 * j + 1, 0] = 0.0, 1.0
            continue
        rr = r @ z
        gam = rr / (rr - alpha * (z @ y))
        rho = 1.0 if j == 0 else 1.0 / (1.0 - (gam / gam_p) * (rr / rr_p) / rho_p)
        r_new = rho * (r - gam * (r - alpha * y)) + (1.0 - rho) * r_old
        psi_new = rho * (psi + gam * z) + (1.0 - rho) * psi_old
        r_old, r, psi_old, psi = r, r_new, psi, psi_new
        rr_p, gam_p, rho_p = rr, gam, rho
        log[2 * j, 0], log[2 * j + 1, 0] = gam, rho
    return psi, Z, log, r


def _problem(seed=0, n=200):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    F = Q @ np.diag(rng.uniform(0.5, 2.0, n)) @ Q.T           # SPD "factor"
    S = rng.normal(size=(n, n))
    K = -(S @ S.T) / n                                         # K F has negative eigenvalues: C = I - alpha K F is SPD in <.,.>_F
    return F, K, 0.7, rng.normal(size=n)


def test_solution_from_the_z_history_is_the_iterate_of_the_recurrence():
    F, K, alpha, b = _problem()
    psi, Z, log, r = _three_term_cg(F, K, alpha, b, 25)
    S = _cg_solution_coefficients(log, 4)
    assert np.all(S[:25, 0] > 0.0) and not S[25:, 0].any() and not S[:, 1:].any()
    psi_h = S[:25, 0] @ Z
    assert np.linalg.norm(psi_h - psi) <= 1e-13 * np.linalg.norm(psi)
    # it solves C psi = b with r the residual: b - (psi - alpha K F psi)  ... psi is in the space of F r, the system is in r
    res = b - (np.eye(len(b)) - alpha * K @ F) @ np.linalg.solve(F, psi_h)
    assert np.linalg.norm(res - r) <= 1e-10 * np.linalg.norm(b)
    assert np.linalg.norm(r) <= 1e-8 * np.linalg.norm(b)


def test_a_column_that_stops_moving_keeps_its_iterate():
    F, K, alpha, b = _problem(seed=1)
    psi, Z, log, _ = _three_term_cg(F, K, alpha, b, 12, stop_after=7)     # frozen from step 8 on (gam = 0 in the log)
    ref, _, _, _ = _three_term_cg(F, K, alpha, b, 7)
    S = _cg_solution_coefficients(log, 1)
    assert not S[7:, 0].any()
    assert np.linalg.norm(S[:12, 0] @ Z - ref) <= 1e-13 * np.linalg.norm(ref)
    assert np.linalg.norm(psi - ref) <= 1e-13 * np.linalg.norm(ref)


def test_a_restart_of_the_recurrence_is_a_step_with_rho_one():
    """rho = 1 in the middle (the device's answer to a non-positive denominator): p restarts, the formula needs no case"""
    F, K, alpha, b = _problem(seed=2)
    n = len(b)
    r, r_old, psi, psi_old = b.copy(), np.zeros(n), np.zeros(n), np.zeros(n)
    log = np.zeros((2 * 14, 64)); Z = np.zeros((12, n))
    rr_p = gam_p = rho_p = None
    for j in range(12):
        z = F @ r; y = K @ z; Z[j] = z
        rr = r @ z
        gam = rr / (rr - alpha * (z @ y))
        rho = 1.0 if j in (0, 5) else 1.0 / (1.0 - (gam / gam_p) * (rr / rr_p) / rho_p)
        r, r_old = rho * (r - gam * (r - alpha * y)) + (1.0 - rho) * r_old, r
        psi, psi_old = rho * (psi + gam * z) + (1.0 - rho) * psi_old, psi
        rr_p, gam_p, rho_p = rr, gam, rho
        log[2 * j, 3], log[2 * j + 1, 3] = gam, rho
    S = _cg_solution_coefficients(log, 8)
    assert np.linalg.norm(S[:12, 3] @ Z - psi) <= 1e-13 * np.linalg.norm(psi)
