"""
sibk in short-recurrence form (conjugate gradients in the inner product of the factor, csrc/krylov.hip) against the
Arnoldi form that restates the reference's loop (eigd/eigenvector_derivatives.py:1246-1277), against the reference's
own psi in the golden fixtures, and against the CPU oracle.  Tolerance on psi: 1e-8 relative (north_star).
"""
import warnings

import numpy as np
import pytest

from conftest import corr_from, csr_from, index_sets, load_golden, relerr

pytestmark = pytest.mark.gpu

RTOL = 1e-8


def _adopt(s, g, prefix=""):
    m = int(g[prefix + "m"])
    if hasattr(s, "lam0"):
        s.lam0 = g[prefix + "lam"].copy()
    else:
        s.lam = g[prefix + "lam"].copy()
    s.Phi = g[prefix + "Phi"].copy()
    s.m = s._m = m
    s.V = g[prefix + "V"]
    s.Y, s.theta = g[prefix + "Y"].copy(), g[prefix + "theta"].copy()
    s.indices, s.T = g[prefix + "indices"].copy(), g[prefix + "T"].copy()


def _both_forms(monkeypatch, solve):
    """run ``solve()`` (-> psi, data, info, history) in both forms; checks which form ran"""
    import eigd_amd as eg
    from eigd_amd import adjoint as adj

    out = {}
    for form in ("auto", "arnoldi"):
        monkeypatch.setattr(eg.tuning, "recurrence", form)
        out[form] = solve()
        assert adj.LAST_ROUND["recurrence"] == ("short" if form == "auto" else "arnoldi"), adj.LAST_ROUND["recurrence"]
    # the short form once more with the solution's own three-term recurrence running along instead of psi formed at the end
    # from the z history: same iterates (result [0] = psi, [2] = steps per mode)
    monkeypatch.setattr(eg.tuning, "recurrence", "auto")
    monkeypatch.setattr(eg.tuning, "cg_solution_from_history", False)
    alt = solve()
    assert adj.LAST_ROUND["recurrence"] == "short" and adj.LAST_ROUND["cg_solution"] == "recurrence"
    monkeypatch.setattr(eg.tuning, "cg_solution_from_history", True)
    assert relerr(alt[0], out["auto"][0]) < 1e-11 and list(alt[2]) == list(out["auto"][2])
    return out["auto"], out["arnoldi"]


@pytest.mark.parametrize("case", ["g1_buckling", "g4_normal", "g4_buckling", "g2_normal", "g3_repeated"])
def test_short_recurrence_reproduces_the_reference_psi_and_the_arnoldi_form(monkeypatch, case):
    import eigd_amd as eg

    first = 0
    if case == "g1_buckling":
        g = load_golden("g1_buckling50_basiclanczos")
        A, B, mode, p = csr_from(g, "G"), csr_from(g, "K"), "buckling", ""
        sigma, N, rhs, want, corr = float(g["sigma"]), 6, g["Qrb"], g["psir"], corr_from(g, "corr")
    elif case.startswith("g4"):
        g = load_golden("g4_laplace900_basiclanczos")
        K, M = csr_from(g, "K"), csr_from(g, "M")
        mode = case.split("_")[1]
        A, B = (K, M) if mode == "normal" else ((-0.005 * M).tocsr(), K)
        p = mode + "_"
        sigma, N, rhs, want, corr = float(g[p + "sigma"]), 6, g["Phib"], g[p + "sibk_psi"], corr_from(g, p + "sibk_corr")
    elif case == "g2_normal":
        g = load_golden("g2_natfreq32x16_basiclanczos")       # (three rigid-body modes at lam ~ 1e-15, dropped afterwards)
        A, B, mode, p = csr_from(g, "K"), csr_from(g, "M"), "normal", ""
        sigma, N, rhs, want, corr = float(g["sigma"]), 13, g["Q0b"], g["psi"], corr_from(g, "corr")
        first = 3
    else:
        g = load_golden("g3_thermal32_eps1e-8_basiclanczos")
        A, B, mode, p = csr_from(g, "K"), csr_from(g, "M"), "normal", ""
        sigma, N, rhs, want, corr = float(g["sigma"]), 8, g["Qb"], g["psi"], corr_from(g, "corr")
    P = (A - sigma * B) if mode == "normal" else (B + sigma * A)
    fac = eg.SpLuOperator(P.tocsc())
    assert fac.negative_pivots == 0 and fac.static_pivots == 0
    s = eg.BasicLanczos(N=N, m=60, mode=mode)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s.solve(A, B, fac, sigma)
    _adopt(s, g, p)                                     # the reference's own eigenvectors and Lanczos data

    def solve():
        hist = []
        fac.count = 0
        psi, data = s.solve_adjoint(rhs.copy(), method="sibk", rtol=1e-12, update_guess=False, bs_target=1,
                                    callback=hist.append)
        return psi, data, list(s.last_info), hist, fac.count

    (psi_s, data_s, info_s, hist_s, cnt_s), (psi_a, data_a, info_a, hist_a, cnt_a) = _both_forms(monkeypatch, solve)
    assert index_sets(data_s) == index_sets(corr) == index_sets(data_a)
    assert relerr(psi_s[:, first:], want) < RTOL and relerr(psi_a[:, first:], want) < RTOL
    assert relerr(psi_s, psi_a) < 1e-9
    # same Krylov spaces: the step counts agree up to what the two minimisation properties differ by
    assert len(info_s) == len(info_a)
    assert sum(info_s) <= 1.1 * sum(info_a) + len(info_a), (info_s, info_a)
    # one factor application per step and mode (reference 19-22, 1248) on top of the first guess's
    assert cnt_s - sum(info_s) == cnt_a - sum(info_a) >= 0
    # the residuals handed to ``callback`` are true Euclidean residual norms: the last one of every mode meets 1275
    res, _ = s.eval_adjoint_residual_norm(rhs, psi_s, b_ortho=True)
    rn0 = np.sqrt(np.max(np.sum(rhs * rhs, axis=0)))
    assert res.max() < 1e-9 * rn0


def test_short_recurrence_on_a_column_with_deflated_extra_pairs_matches_the_arnoldi_form(monkeypatch):
    """buckling column, restarted block eigensolver with converged pairs beyond N (deflated by both forms), 24 modes of
    which one has a zero right-hand side; more than 32 modes in a second run (one block of 40 columns)"""
    import eigd_amd as eg
    from eigd_amd.device import default_context
    from eigd_amd.problems import BucklingColumn

    ctx = default_context()
    col = BucklingColumn(90, 90, seed=2)
    K = col.stiffness()
    u = col.full_vector(eg.SpLuOperator(K, ctx=ctx, check_symmetry=False)(col.f[col.reduced]))
    A, B, sigma = col.geometric_stiffness(u), K, 1.0
    fac = eg.SpLuOperator((B + sigma * A).tocsr(), ctx=ctx, check_symmetry=False)
    monkeypatch.setattr(eg.tuning, "iram_block", 4)
    for N in (24, 40):
        s = eg.IRAM(N=N, m=2 * N + 1, mode="buckling", ctx=ctx)
        s.solve(A, B, fac, sigma)
        Phib = np.random.default_rng(7).uniform(-1, 1, size=(B.shape[0], N))
        Phib[:, 3] = 0.0

        def solve():
            hist = []
            psi, data = s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1, callback=hist.append)
            return psi, data, list(s.last_info), hist

        (psi_s, data_s, info_s, hist_s), (psi_a, data_a, info_a, hist_a) = _both_forms(monkeypatch, solve)
        assert relerr(psi_s, psi_a) < 1e-9
        assert index_sets(data_s) == index_sets(data_a)
        assert info_s[3] == 0 == info_a[3]
        assert sum(info_s) <= 1.1 * sum(info_a) + N and max(info_s) <= 1.15 * max(info_a) + 1, (info_s, info_a)
        res, _ = s.eval_adjoint_residual_norm(Phib, psi_s, b_ortho=True)
        assert res.max() < 1e-9 * np.sqrt(np.max(np.sum(Phib * Phib, axis=0)))
        # a step limit inside the solve: the best iterate so far comes back, the unfinished modes are not in ``info``
        monkeypatch.setattr(eg.tuning, "recurrence", "auto")
        psi_c, _ = s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1, maxiter=4, nrestart=0)
        assert len(s.last_info) < N and np.all(np.isfinite(psi_c))
        assert relerr(psi_c, psi_s) < 0.5


def test_short_recurrence_keeps_the_residual_out_of_the_deflated_directions(monkeypatch):
    """
    Along a requested pair below lam_i the operator C_i is negative: what rounding leaves there grows by an order of
    magnitude per step relative to the residual and, left alone, turns <r, C r>_F negative (on the 1 M-dof column in step
    24, docs/LOG.md round 4).  The periodic projection takes it out of the residual AND of the previous residual, which
    the three-term recurrence brings back: the relative component stays below 1e-6 over chains of 20+ steps, with and
    without deflated extra pairs, and the short form runs to the end.
    """
    import eigd_amd as eg
    from eigd_amd import adjoint as adj
    from eigd_amd.device import default_context
    from eigd_amd.problems import BucklingColumn

    ctx = default_context()
    col = BucklingColumn(90, 90, seed=2)
    K = col.stiffness()
    u = col.full_vector(eg.SpLuOperator(K, ctx=ctx, check_symmetry=False)(col.f[col.reduced]))
    A, B, sigma, N = col.geometric_stiffness(u), K, 1.0, 24
    fac = eg.SpLuOperator((B + sigma * A).tocsr(), ctx=ctx, check_symmetry=False)
    monkeypatch.setattr(eg.tuning, "iram_block", 4)
    monkeypatch.setattr(eg.tuning, "recurrence", "auto")
    Phib = np.random.default_rng(3).uniform(-1, 1, size=(B.shape[0], N))
    seen = []

    def hook(prob, j, rv, lo, hi, projected):
        C = prob.Phi.tdot(rv)
        rel = np.abs(C) * prob.BPhi.colnorms()[:, None] / rv.colnorms()[None, :]
        seen.append((j, float(rel.max())))

    monkeypatch.setattr(adj, "_CG_TRACE_HOOK", hook)
    for extra in (0, 24):
        monkeypatch.setattr(eg.tuning, "iram_extra", extra)
        s = eg.IRAM(N=N, m=2 * N + 1, mode="buckling", ctx=ctx)
        s.solve(A, B, fac, sigma)
        assert s.n_extra == extra
        seen.clear()
        psi, _ = s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
        assert adj.LAST_ROUND["recurrence"] == "short", adj.LAST_ROUND
        assert len(seen) >= 12 and max(v for _, v in seen) < 1e-6, seen
        res, _ = s.eval_adjoint_residual_norm(Phib, psi, b_ortho=True)
        assert res.max() < 1e-9 * np.sqrt(np.max(np.sum(Phib * Phib, axis=0)))
    # the mechanism: with the residual alone projected the component grows from period to period (whatever form finishes)
    monkeypatch.setattr(eg.tuning, "cg_project_previous", False)
    seen.clear()
    s.solve_adjoint(Phib, method="sibk", rtol=1e-10, update_guess=False, bs_target=1)
    assert max(v for _, v in seen) > 1e-5, seen


def test_history_stacks_are_kept_for_the_two_last_block_shapes_only():
    import eigd_amd as eg
    from eigd_amd.device import default_context

    ctx = default_context()
    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    lam, Phi, Phib = g["normal_lam"], g["normal_Phi"], g["Phib"]
    fac = eg.SpLuOperator((K + 0.1 * M).tocsc(), ctx=ctx)
    for N in (6, 5, 4, 3):
        eg.sibk(Phib[:, :N], K, M, lam[:N], Phi[:, :N], factor=fac, sigma=-0.1, rtol=1e-10, ctx=ctx)
    tags = [t for t in ctx.__dict__.get("_ws", {}) if isinstance(t, tuple) and t[0] == "cg_z"]
    assert sorted({t[3] for t in tags}) == [3, 4], tags


def test_solution_coefficients_on_the_device_match_the_host_twin():
    """eigd_cg_solution_coefficients against adjoint._cg_solution_coefficients on a made-up log: columns that stop moving
    at different steps, a restart (rho = 1) in the middle, a column that never moves"""
    from eigd_amd._ffi import call
    from eigd_amd.adjoint import _cg_solution_coefficients
    from eigd_amd.device import default_context

    ctx = default_context()
    rng = np.random.default_rng(5)
    nsteps, k = 37, 29
    log = np.zeros((2 * (nsteps + 2), 64))
    for c in range(k):
        m = 0 if c == 4 else int(rng.integers(1, nsteps + 1))
        log[0:2 * m:2, c] = rng.uniform(0.3, 2.0, size=m)
        log[1:2 * m:2, c] = np.r_[1.0, rng.uniform(1.0, 2.5, size=max(m - 1, 0))][:m]
        if m > 6:
            log[2 * 5 + 1, c] = 1.0
    want = _cg_solution_coefficients(log, k)[:nsteps]
    dlog, dS = ctx.from_host(log), ctx.empty(nsteps, k)
    call("eigd_cg_solution_coefficients", ctx.h, k, dlog.ptr, nsteps, dS.ptr)
    got = dS.get()
    assert np.array_equal(got == 0.0, want == 0.0)
    assert np.allclose(got, want, rtol=1e-14, atol=0.0)


def test_short_recurrence_without_memory_for_its_history_hands_over(monkeypatch):
    """no room for another 16 slabs of the z history (here: from the second allocation on): the Arnoldi form redoes the solve"""
    import eigd_amd as eg
    from eigd_amd import _ffi, adjoint as adj, device as dev

    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    lam, Phi, Phib = g["normal_lam"], g["normal_Phi"], g["Phib"]
    sigma = -0.1
    fac = eg.SpLuOperator((K - sigma * M).tocsc())
    monkeypatch.setattr(eg.tuning, "recurrence", "auto")
    psi_ok, _, info_ok = eg.sibk(Phib, K, M, lam, Phi, factor=fac, sigma=sigma, rtol=1e-12)
    assert adj.LAST_ROUND["recurrence"] == "short" and max(info_ok) > 2
    monkeypatch.setattr(adj, "_CG_CHUNK", 2)
    orig = dev.Context.workspace_stack

    def stingy(self, tag, ns, n, k=1):
        if isinstance(tag, tuple) and tag[0] == "cg_z" and tag[1] >= 1:   # (tag: name, allocation number, n, k)
            raise _ffi.EigdHipError("out of memory (test)")
        return orig(self, tag, ns, n, k)

    monkeypatch.setattr(dev.Context, "workspace_stack", stingy)
    psi, _, info = eg.sibk(Phib, K, M, lam, Phi, factor=fac, sigma=sigma, rtol=1e-12)
    assert adj.LAST_ROUND["recurrence"].startswith("arnoldi (") and adj.LAST_ROUND["cg_exit"]["no_memory_at_step"] == [3]
    assert relerr(psi, psi_ok) < 1e-9


def test_short_recurrence_steps_aside_where_it_does_not_apply(monkeypatch):
    """an interior shift (indefinite factor) and an incomplete deflation set (eigenvalues below lam_i left in: the
    operator is indefinite in the deflated space, the breakdown is flagged on the device) both end in the Arnoldi form"""
    import eigd_amd as eg
    from eigd_amd import adjoint as adj
    from oracle import eigd_oracle as orc

    g = load_golden("g4_laplace900_basiclanczos")
    K, M = csr_from(g, "K"), csr_from(g, "M")
    lam, Phi, Phib = g["normal_lam"], g["normal_Phi"], g["Phib"]
    monkeypatch.setattr(eg.tuning, "recurrence", "auto")
    # (1) eigenvectors 2..5 only: modes 0 and 1 lie below and are not deflated
    sel = np.arange(2, 6)
    sigma = -0.1
    fac_d = eg.SpLuOperator((K - sigma * M).tocsc())
    fac_o = orc.SpLuOperator((K - sigma * M).tocsc())
    psi_d, data_d, info_d = eg.sibk(Phib[:, sel], K, M, lam[sel], Phi[:, sel], factor=fac_d, sigma=sigma, rtol=1e-12)
    # (the device reports it -- flag 2, the Arnoldi form takes over -- unless the recurrence got through with restarts,
    # flag 1, in which case the residual test has accepted the same psi)
    assert adj.LAST_ROUND["recurrence"].startswith("arnoldi (") or adj.LAST_ROUND["cg_restarted_modes"] > 0, adj.LAST_ROUND
    if adj.LAST_ROUND["recurrence"].startswith("arnoldi ("):
        # a column that broke down stops moving: the loop sees that in the residual norms it reads anyway and gives up
        # after a handful of steps (not after maxiter (nrestart + 1) = 150 sweeps), and the attempt's z history is gone
        # before the Arnoldi form allocates its stacks
        from eigd_amd.device import default_context

        assert adj.LAST_ROUND["cg_exit"]["steps"] <= 20, adj.LAST_ROUND["cg_exit"]
        tags = [t for t in default_context().__dict__.get("_ws", {}) if isinstance(t, tuple) and t[0] == "cg_z" and t[3] == len(sel)]
        assert not tags
    psi_o, data_o, info_o = orc.sibk(Phib[:, sel], K, M, lam[sel], Phi[:, sel], factor=fac_o, sigma=sigma, rtol=1e-12)
    assert relerr(psi_d, psi_o) < RTOL
    # (2) a shift between lam_1 and lam_2
    sig2 = 0.5 * (lam[1] + lam[2])
    fac2 = eg.SpLuOperator((K - sig2 * M).tocsc())
    assert fac2.negative_pivots == 2
    fac2_o = orc.SpLuOperator((K - sig2 * M).tocsc())
    psi_d, _, _ = eg.sibk(Phib, K, M, lam, Phi, factor=fac2, sigma=sig2, rtol=1e-12)
    assert adj.LAST_ROUND["recurrence"] == "arnoldi"
    psi_o, _, _ = orc.sibk(Phib, K, M, lam, Phi, factor=fac2_o, sigma=sig2, rtol=1e-12)
    assert relerr(psi_d, psi_o) < RTOL


def test_device_twins_of_host_arrays_follow_the_host_content(monkeypatch):
    """
    The numpy surface keeps the device block behind the psi it returned (Context.twin_adopt): handed back as it came, that
    array costs add_total_derivative no transfer.  By default it is returned READ-ONLY -- an in-place edit raises instead
    of leaving a stale device copy behind -- and the caller's own Phib is transferred every time, so an edit of a single
    unsampled entry of it is seen.  ``tuning.host_twins = True`` (opt-in) also keeps sampled copies of caller-owned
    arrays; the result always equals the one computed with the copies switched off.
    """
    import eigd_amd as eg
    from eigd_amd import device as dev
    from eigd_amd.problems import FreePlate

    pl = FreePlate(120, 120, seed=1)                    # 29 282 dof x 8 modes: 1.9 MB per array, above the twin threshold
    K, M = pl.stiffness(), pl.mass()
    sigma, N = -10.0, 8
    fac = eg.SpLuOperator((K - sigma * M).tocsr(), check_symmetry=False)
    s = eg.BasicLanczos(N=N, m=60)
    s.solve(K, M, fac, sigma)
    rng = np.random.default_rng(0)
    Phib, lamb = rng.uniform(size=(K.shape[0], N)), rng.uniform(size=N)
    Cm = rng.normal(size=(K.shape[0], 3))
    cb = lambda w, v: Cm.T @ (np.sum(w * v, axis=1) if w.ndim == 2 else w * v)   # noqa: E731

    def run():
        psi, data = s.solve_adjoint(Phib, method="sibk", rtol=1e-11)
        return psi, data

    def derivative(psi, data):
        return s.add_total_derivative(lamb, Phib, psi, cb, cb, np.zeros(3), adj_corr_data=data, deriv_type="tensor")

    uploads = []
    orig = dev.Context.from_host
    monkeypatch.setattr(dev.Context, "from_host", lambda self, a: (uploads.append(np.shape(a)), orig(self, a))[1])
    big = lambda: [u for u in uploads if len(u) == 2 and u[0] == K.shape[0]]   # noqa: E731
    assert eg.tuning.host_twins == "returned"           # the default
    psi, data = run()
    assert not psi.flags.writeable
    with pytest.raises(ValueError):
        psi[5, 2] = 1.0                                 # the edit a kept copy could not see raises
    n0 = len(big())
    d1 = derivative(psi, data)
    assert len(big()) == n0 + 1                         # Phib went over the bus again, psi did not
    # one entry of an UNSAMPLED row of the caller's Phib: seen, because Phib is read as it is given
    idx = set(dev._HostTwins._rows(Phib).tolist())
    row = next(r for r in range(1000, Phib.shape[0]) if r not in idx)
    Phib[row, 3] += 0.5
    d1b = derivative(psi, data)
    monkeypatch.setattr(eg.tuning, "host_twins", False)
    assert relerr(d1b, derivative(psi, data)) < 1e-13   # (equal to what is computed with every array transferred)
    Phib[row, 3] -= 0.5
    n1 = len(big())
    d0 = derivative(psi, data)
    assert len(big()) == n1 + 2 and relerr(d1, d0) < 1e-13
    # a caller that wants to edit psi makes it writable (or copies it): the next call transfers it
    monkeypatch.setattr(eg.tuning, "host_twins", "returned")
    psi, data = run()
    psi.flags.writeable = True
    psi[:, 2] *= -1.0
    n2 = len(big())
    d2 = derivative(psi, data)
    assert len(big()) == n2 + 2
    monkeypatch.setattr(eg.tuning, "host_twins", False)
    assert relerr(d2, derivative(psi, data)) < 1e-13
    # opt-in: sampled copies of caller-owned arrays as well
    monkeypatch.setattr(eg.tuning, "host_twins", True)
    psi, data = run()
    n3 = len(big())
    d3 = derivative(psi, data)
    assert len(big()) == n3                             # neither Phib nor psi went over the bus again
    np.multiply(Phib, 1.5, out=Phib)                    # every sampled row sees this
    d4 = derivative(psi, data)
    assert len(big()) == n3 + 1
    monkeypatch.setattr(eg.tuning, "host_twins", False)
    assert relerr(d4, derivative(psi, data)) < 1e-13
