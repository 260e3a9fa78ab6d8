#!/usr/bin/env python3
"""
Capture golden input/output vectors from the reference implementation.

Runs ONLY in the build container (needs /root/reference).  The reference is
imported unmodified; the one private scipy symbol it needs and that scipy
1.15 dropped (`_aslinearoperator_with_dtype`, arpack.py:4-10) is injected at
run time.  Output: small .npz fixtures in tests/golden/.
A fixture holds data only (matrices, vectors, scalars, index sets).

Every case runs in a fresh interpreter: ARPACK draws its start vector from
SAVEd state that advances per call, so only the first IRAM solve of a process
is reproducible.

usage:  python tools/make_golden.py            # all cases
        python tools/make_golden.py --case g4  # one case, in-process
"""

import argparse
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def import_reference():
    am = importlib.import_module("scipy.sparse.linalg._eigen.arpack.arpack")
    from scipy.sparse.linalg import aslinearoperator

    if not hasattr(am, "_aslinearoperator_with_dtype"):
        am._aslinearoperator_with_dtype = lambda m: aslinearoperator(m)
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "examples"))
    import eigd

    return eigd


def csr_fields(prefix, M):
    M = M.tocsr()
    M.sort_indices()
    return {
        prefix + "_indptr": M.indptr.astype(np.int32),
        prefix + "_indices": M.indices.astype(np.int32),
        prefix + "_data": M.data.astype(np.float64),
        prefix + "_shape": np.array(M.shape, dtype=np.int64),
    }


def corr_fields(prefix, data):
    """dict {i: [(j, xi, eta), ...]} -> flat arrays (order preserved)"""
    rows = [(i, j, xi, eta) for i in sorted(data) for (j, xi, eta) in data[i]]
    arr = np.array(rows, dtype=float).reshape(-1, 4)
    return {
        prefix + "_i": arr[:, 0].astype(np.int64),
        prefix + "_j": arr[:, 1].astype(np.int64),
        prefix + "_xi": arr[:, 2],
        prefix + "_eta": arr[:, 3],
    }


def save(name, **fields):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **fields)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB)")


def solver_fields(prefix, s, basic):
    f = {}
    if basic:
        f.update(
            {
                prefix + "lam": s.lam0,
                prefix + "Phi": s.Phi,
                prefix + "alpha": s.alpha,
                prefix + "beta": s.beta,
                prefix + "m": np.int64(s.m),
                prefix + "N": np.int64(s.N),
                prefix + "theta": s.theta,
                prefix + "indices": np.asarray(s.indices, dtype=np.int64),
                prefix + "V": s.V[:, : s.m],
                prefix + "Y": s.Y,
                prefix + "T": s.T,
                prefix + "eig_res": s.eig_res,
            }
        )
    else:
        f.update(
            {
                prefix + "lam": s.lam,
                prefix + "Phi": s.Phi,
                prefix + "theta": s.theta,
                prefix + "indices": np.asarray(s.indices, dtype=np.int64),
                prefix + "V": s.V,
                prefix + "Y": s.Y,
                prefix + "T": s.T,
                prefix + "m": np.int64(s.m),
            }
        )
    return f


# ---------------------------------------------------------------- G1 -------
def case_g1(solver_type):
    """C1: 50x50 buckling column, N=6, sigma=3, tanh eigenvector aggregate."""
    import_reference()
    import buckling

    np.random.seed(0)
    topo = buckling.make_model(
        nx=50, ny=50, Lx=1.0, Ly=1.0, N=6, sigma=3.0, solver_type=solver_type,
        adjoint_method="sibk",
        adjoint_options={"lanczos_guess": True, "update_guess": False, "bs_target": 1},
        deriv_type="tensor",
    )
    pert = np.random.uniform(size=topo.x.shape)  # the draw test_eigenvector_aggregate_derivatives would make itself
    node = (8 + 1) * 16 + 16
    d = topo.test_eigenvector_aggregate_derivatives(mode="tanh", rho=100.0, node=node, pert=pert)
    # test_* re-initialises at perturbed points; redo the base point for capture
    topo.initialize(store=True)
    topo.initialize_adjoint()
    h_agg = topo.get_eigenvector_aggregate(100.0, node, mode="tanh")
    topo.add_eigenvector_aggregate_derivative(1.0, 100.0, node, mode="tanh")
    Qrb = topo.Qrb.copy()
    lamb = topo.lamb.copy()
    # the two add_total_derivative calls of finalize_adjoint (buckling.py:930-972): d/du of the eigen part, then d/drho
    captured = []
    s = topo.eig_solver
    orig_add = s.add_total_derivative

    def recording_add(*a, **kw):
        out = orig_add(*a, **kw)
        captured.append(np.array(out))
        return out

    s.add_total_derivative = recording_add
    topo.finalize_adjoint()
    s.add_total_derivative = orig_add
    dfdu0, rhob_eig = captured
    f = {}
    f.update(csr_fields("K", topo.Kr))
    f.update(csr_fields("G", topo.Gr))
    f.update(solver_fields("", s, solver_type == "BasicLanczos"))
    f.update(corr_fields("corr", topo.profile["adjoint correction data"]))
    f.update(
        sigma=np.float64(topo.sigma), Qrb=Qrb, lamb=lamb, psir=topo.psir, BLF=topo.BLF,
        ans=np.float64(d["ans"]), cd=np.float64(d["cd"]), cd_err=np.float64(d["cd_err"]),
        adjoint_residuals=np.array(topo.profile["adjoint residuals"]),
        count_adjoint=np.int64(topo.profile["adjoint preconditioner count"]),
    )
    if "cs" in d:
        f.update(cs=np.float64(d["cs"]), cs_err=np.float64(d["cs_err"]))
    # the harness around the eigd calls: mesh, design variables, filter, fundamental path, the chain rule's stages
    f.update(csr_fields("F", topo.fltr.F))
    f.update(
        conn=topo.conn.astype(np.int32), X=topo.X, x=np.array(topo.x), rhoE=topo.rhoE, u=topo.u, f=topo.f,
        reduced=np.asarray(topo.reduced, dtype=np.int64), dvmap=np.asarray(topo.fltr.dvmap, dtype=np.int64),
        num_design_vars=np.int64(topo.fltr.num_design_vars), r0=np.float64(topo.fltr.r0),
        p=np.float64(topo.p), rho0_K=np.float64(topo.rho0_K), rho0_G=np.float64(topo.rho0_G),
        E=np.float64(topo.E), nu=np.float64(topo.nu),
        pert=pert, node=np.int64(node), agg_rho=np.float64(100.0), h_agg=np.float64(h_agg),
        dfdu0=dfdu0, rhob_eig=rhob_eig, rhob=topo.rhob, xb=topo.xb,
    )
    if solver_type == "BasicLanczos":
        # the other functionals of the harness at the same point (buckling.py:634-700): KS of the buckling loads and
        # the compliance, with their design gradients
        f.update(
            ks_rho=np.float64(30.0), ks=np.float64(topo.eval_ks_buckling(30.0)),
            ks_grad=topo.eval_ks_buckling_derivative(30.0),
            compliance=np.float64(topo.compliance()), compliance_grad=topo.compliance_derivative(),
        )
    save("g1_buckling50_" + solver_type.lower(), **f)


# ---------------------------------------------------------------- G2 -------
def case_g2(solver_type):
    """normal mode: small free-free plate, MinFreqOpt KS function."""
    import_reference()
    import natural_frequency as nf

    np.random.seed(0)
    topo = nf.make_model(
        nx=32, ny=16, Lx=2.0, Ly=1.0, N=10, solver_type=solver_type, adjoint_method="sibk",
        adjoint_options={"lanczos_guess": True, "update_guess": False, "bs_target": 1},
    )
    opt = nf.MinFreqOpt(topo)
    opt.initialize(store=True)
    opt.initialize_adjoint()
    opt.finalize_adjoint()
    res = topo.add_check_adjoint_residual(b_ortho=True)
    s = topo.eig_solver
    Q0b = np.zeros((topo.nvars, 3 + topo.N))
    Q0b[:, 3:] = topo.Qb
    lamb0 = np.zeros(3 + topo.N)
    lamb0[3:] = topo.lamb
    f = {}
    f.update(csr_fields("K", topo.K))
    f.update(csr_fields("M", topo.M))
    f.update(solver_fields("", s, solver_type == "BasicLanczos"))
    f.update(corr_fields("corr", topo.profile["adjoint correction data"]))
    f.update(
        sigma=np.float64(topo.sigma), Q0b=Q0b, lamb0=lamb0, psi=topo.psi, rhoEb=topo.rhoEb,
        res_bortho=np.asarray(res), adjoint_residuals=np.array(topo.profile["adjoint residuals"]),
    )
    f.update(csr_fields("F", topo.fltr.F))
    f.update(
        conn=topo.conn.astype(np.int32), X=topo.X, x=np.array(topo.x), rhoE=topo.rhoE, xb=topo.xb,
        dvmap=np.asarray(topo.fltr.dvmap, dtype=np.int64), num_design_vars=np.int64(topo.fltr.num_design_vars),
        r0=np.float64(topo.fltr.r0), p=np.float64(topo.p), rho0_K=np.float64(topo.rho0_K),
        density=np.float64(topo.density), E=np.float64(topo.E), nu=np.float64(topo.nu),
    )
    # the KS functional of MinFreqOpt (natural_frequency.py:700-807) whose seeds Q0b / lamb0 are: point-mass node sets
    names = sorted(topo.node_sets)
    f.update(
        ns_nodes=np.concatenate([topo.node_sets[k] for k in names]).astype(np.int64),
        ns_ptr=np.cumsum([0] + [len(topo.node_sets[k]) for k in names]).astype(np.int64),
        ks_param=np.float64(opt.ks_param), fixed_mass=np.float64(opt.fixed_mass), ks_min=np.float64(opt.ks_min),
        omegab=np.asarray(opt.omegab, dtype=float),
    )
    save("g2_natfreq32x16_" + solver_type.lower(), **f)


# ---------------------------------------------------------------- G3 -------
def case_g3(solver_type, epsilon, tag):
    """repeated-eigenvalue branch on the (nearly) square thermal problem."""
    import_reference()
    import thermal

    np.random.seed(0)
    topo = thermal.make_opt_model(
        nx=32, rfact=4.0, N=8, m=60, p=3, epsilon=epsilon, solver_type=solver_type,
        adjoint_method="sibk",
        adjoint_options={"lanczos_guess": True, "update_guess": False, "bs_target": 1},
        element_sets={}, eig_atol=1e-5, rtol=1e-12, deriv_type="tensor",
    )
    vec = np.random.uniform(size=topo.nnodes)
    d = topo.test_compliance_derivatives(vec=vec, dh_cs=1e-20)
    topo.initialize(store=True)
    topo.initialize_adjoint()
    topo.add_thermal_compliance_derivative(1.0, vec)
    topo.finalize_adjoint()
    res, ortho = topo.eig_solver.eval_adjoint_residual_norm(topo.Qb, topo.psi, b_ortho=True)
    s = topo.eig_solver
    f = {}
    f.update(csr_fields("K", topo.K))
    f.update(csr_fields("M", topo.M))
    f.update(solver_fields("", s, solver_type == "BasicLanczos"))
    f.update(corr_fields("corr", topo.profile["adjoint correction data"]))
    f.update(
        sigma=np.float64(topo.sigma), Qb=topo.Qb, lamb=topo.lamb, psi=topo.psi, rhoEb=topo.rhoEb,
        vec=vec, epsilon=np.float64(epsilon), res_bortho=res, ortho_bortho=ortho,
        ans=np.float64(d["ans"]), cd=np.float64(d["cd"]), cd_err=np.float64(d["cd_err"]),
        conn=topo.conn.astype(np.int32), X=topo.X, rhoE=topo.rhoE,
    )
    f.update(csr_fields("F", topo.fltr.F))
    f.update(
        x=np.array(topo.x), xb=topo.xb, dvmap=np.asarray(topo.fltr.dvmap, dtype=np.int64),
        num_design_vars=np.int64(topo.fltr.num_design_vars), r0=np.float64(topo.fltr.r0),
        p=np.float64(topo.p), th_beta=np.float64(topo.beta), kappa=np.float64(topo.kappa),
        density=np.float64(topo.density), heat_capacity=np.float64(topo.heat_capacity),
        compliance=np.float64(topo.get_thermal_compliance(vec)),
    )
    save(f"g3_thermal32_{tag}_{solver_type.lower()}", **f)


# ---------------------------------------------------------------- G4 -------
def laplacian_pair(nside=30, seed=3):
    from scipy import sparse

    I = sparse.identity(nside)
    T = sparse.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nside, nside))
    K = (sparse.kron(I, T) + sparse.kron(T, I)).tocsr()
    rng = np.random.default_rng(seed)
    M = sparse.diags(rng.uniform(0.5, 1.5, size=nside * nside)).tocsr()
    return K, M


def case_g4(solver_type):
    """method matrix on a 900-dof Laplacian / random-diagonal pair."""
    eigd = import_reference()
    K, M = laplacian_pair()
    sigma, N = -0.1, 6
    n = K.shape[0]
    rng = np.random.default_rng(7)
    Phib = rng.uniform(size=(n, N))
    lamb = rng.uniform(size=N)
    f = {}
    f.update(csr_fields("K", K))
    f.update(csr_fields("M", M))
    for mode in ("normal", "buckling"):
        # buckling convention: (B + lam A) phi = 0 with A = -0.005 M (a negative definite "G"), B = K;
        # lam_1 ~ 4.2 and the shift 3.0 keeps B + sigma A positive definite (as buckling.py:1421)
        A, B = (K, M) if mode == "normal" else ((-0.005 * M).tocsr(), K)
        sig = sigma if mode == "normal" else 3.0
        mat = (A - sig * B) if mode == "normal" else (B + sig * A)
        factor = eigd.SpLuOperator(mat.tocsc())
        if solver_type == "BasicLanczos":
            s = eigd.BasicLanczos(N=N, m=60, mode=mode)
        else:
            s = eigd.IRAM(N=N, m=40, mode=mode)
        s.solve(A, B, factor, sig)
        p = mode + "_"
        f.update(solver_fields(p, s, solver_type == "BasicLanczos"))
        f[p + "sigma"] = np.float64(sig)
        methods = ["laa", "sibk", "pcpg", "pgmres"] + (["dl"] if solver_type == "BasicLanczos" else [])
        for method in methods:
            factor.count = 0
            kw = {}
            if method == "sibk":
                kw = {"update_guess": False, "bs_target": 1}
            psi, data = s.solve_adjoint(Phib.copy(), method=method, rtol=1e-12, **kw)
            f[p + method + "_psi"] = psi
            f[p + method + "_count"] = np.int64(factor.count)
            f.update(corr_fields(p + method + "_corr", data))
            res, ortho = s.eval_adjoint_residual_norm(Phib, psi, b_ortho=False)
            f[p + method + "_res"] = res
        # extra sibk variants: block size 2 and update_guess
        for tag, kw in (("sibk_bs2", {"bs_target": 2}), ("sibk_ug", {"update_guess": True})):
            psi, data = s.solve_adjoint(Phib.copy(), method="sibk", rtol=1e-12, **kw)
            f[p + tag + "_psi"] = psi
    f.update(Phib=Phib, lamb=lamb)
    save("g4_laplace900_" + solver_type.lower(), **f)


# ---------------------------------------------------------------- G5 -------
def case_g5():
    """unit vectors for the small deterministic helpers."""
    eigd = import_reference()
    import eigd.eigenvector_derivatives as ed

    rng = np.random.default_rng(11)
    n, N, ndv = 40, 5, 7
    f = {}
    U, V, X = rng.normal(size=(n, N)), rng.normal(size=(n, N)), rng.normal(size=(n, 3))
    f.update(proj_U=U, proj_V=V, proj_X=X, proj_out=ed._project(U, V, X.copy()))
    x1 = rng.normal(size=n)
    f.update(proj_x1=x1, proj_out1=ed._project(U, V, x1.copy()))

    Phi = rng.normal(size=(n, N))
    Phib = rng.normal(size=(n, N))
    lamb = rng.normal(size=N)
    Ca, Cb = rng.normal(size=(n, ndv)), rng.normal(size=(n, ndv))

    def mk(C):
        def cb(w, v):
            if w.ndim == 1:
                return C.T @ (w * v)
            return C.T @ np.sum(w * v, axis=1)
        return cb

    f.update(Phi=Phi, Phib=Phib, lamb=lamb, Ca=Ca, Cb=Cb)
    for tag, lam in (("distinct", np.array([1.0, 2.0, 3.5, 4.0, 7.0])),
                     ("repeated", np.array([1.0, 1.0 + 2e-6, 3.5, 4.0, 4.0 + 5e-6]))):
        f[tag + "_lam"] = lam
        f[tag + "_repeated"] = np.bool_(eigd.are_eigenvalues_repeated(lam))
        for mode in ("normal", "buckling"):
            psi = rng.normal(size=(n, N))
            p = f"{tag}_{mode}_"
            f[p + "psi_in"] = psi.copy()
            data = eigd.generate_adjoint_correction(lam, Phi, psi, Phib=Phib, mode=mode)
            f[p + "psi_out"] = psi.copy()
            f.update(corr_fields(p + "corr", data))
            for dt in ("vector", "tensor"):
                dfdx = eigd.add_eig_total_derivative(
                    lam, Phi, lamb, Phib, psi, mk(Ca), mk(Cb), np.zeros(ndv),
                    adj_corr_data=data, mode=mode, deriv_type=dt)
                f[p + "dfdx_" + dt] = dfdx
    save("g5_units", **f)


# ---------------------------------------------------------------- G6 -------
def case_g6():
    """complex-step point of C1: the reference's own CS evaluation (buckling.py:1014-1023) on the 50x50
    column -- design variables x + i*1e-20*p, complex K/G, complex SuperLU, BasicLanczos with its complex _eigh."""
    import_reference()
    import buckling

    np.random.seed(0)
    topo = buckling.make_model(
        nx=50, ny=50, Lx=1.0, Ly=1.0, N=6, sigma=3.0, solver_type="BasicLanczos",
        adjoint_method="sibk",
        adjoint_options={"lanczos_guess": True, "update_guess": False, "bs_target": 1},
        deriv_type="tensor",
    )
    node = (8 + 1) * 16 + 16
    topo.initialize(store=True)
    x0 = np.array(topo.x)
    h0 = topo.get_eigenvector_aggregate(100.0, node, mode="tanh")
    lam_real = np.array(topo.eig_solver.lam0)
    pert = np.random.uniform(size=topo.x.shape)
    dh = 1e-20
    topo.x = np.array(x0).astype(complex)
    topo.x.imag += dh * pert
    topo.initialize()
    h1 = topo.get_eigenvector_aggregate(100.0, node, mode="tanh")
    s = topo.eig_solver
    K, G = topo.Kr.tocsr(), topo.Gr.tocsr()
    K.sort_indices()
    G.sort_indices()
    assert np.iscomplexobj(K.data) and np.iscomplexobj(G.data)
    f = {}
    for name, M in (("K", K), ("G", G)):
        f.update({name + "_indptr": M.indptr.astype(np.int32), name + "_indices": M.indices.astype(np.int32),
                  name + "_re": M.data.real.copy(), name + "_im": M.data.imag.copy(),
                  name + "_shape": np.array(M.shape, dtype=np.int64)})
    f.update(
        sigma=np.float64(topo.sigma), dh=np.float64(dh), node=np.int64(node), reduced=np.asarray(topo.reduced, dtype=np.int64),
        lam_real_point=lam_real, h_real_point=np.float64(np.real(h0)),
        lam=s.lam0, Phi=s.Phi, alpha=s.alpha, beta=s.beta, m=np.int64(s.m), N=np.int64(s.N), theta=s.theta,
        indices=np.asarray(s.indices, dtype=np.int64), Y=s.Y, eig_res=s.eig_res, h=np.complex128(h1),
        cs=np.float64(h1.imag / dh),
    )
    save("g6_buckling50_complexstep", **f)


# ---------------------------------------------------------------- G7 -------
def case_g7():
    """NodeFilter units (examples/node_filter.py:10-217): spatial and Helmholtz filters, with and without the tanh
    projection and a symmetric design-variable map, forward and gradient, on a small non-square mesh."""
    import_reference()
    from node_filter import NodeFilter

    nx, ny, Lx, Ly = 14, 9, 1.4, 0.8
    x1, y1 = np.linspace(0, Lx, nx + 1), np.linspace(0, Ly, ny + 1)
    nodes = np.arange((nx + 1) * (ny + 1)).reshape(nx + 1, ny + 1)
    X = np.zeros(((nx + 1) * (ny + 1), 2))
    X[nodes.ravel(), 0] = np.repeat(x1, ny + 1)
    X[nodes.ravel(), 1] = np.tile(y1, nx + 1)
    conn = np.zeros((nx * ny, 4), dtype=int)
    for j in range(ny):
        for i in range(nx):
            conn[i + nx * j] = [nodes[i, j], nodes[i + 1, j], nodes[i + 1, j + 1], nodes[i, j + 1]]
    dvmap = np.zeros((nx + 1, ny + 1), dtype=int)
    index = 0
    for i in range(nx // 2 + 1):                      # mirror symmetry left-right, as buckling.py:1327-1340
        for j in range(ny + 1):
            dvmap[i, j] = index
            dvmap[nx - i, j] = index
            index += 1
    dvmap = dvmap.flatten()
    dvmap[nodes[3, 4]] = -1                            # one node outside the design domain (x = 1 there)
    rng = np.random.default_rng(21)
    x = rng.uniform(0.05, 0.95, size=index)
    g = rng.normal(size=X.shape[0])
    r0 = 0.27
    f = dict(conn=conn.astype(np.int32), X=X, dvmap=dvmap.astype(np.int64), num_design_vars=np.int64(index),
             x=x, g=g, r0=np.float64(r0), beta=np.float64(8.0), eta=np.float64(0.4))
    for ftype in ("spatial", "helmholtz"):
        for proj in (False, True):
            for use_map in (False, True):
                kw = dict(dvmap=dvmap, num_design_vars=index) if use_map else {}
                flt = NodeFilter(conn, X, r0=r0, ftype=ftype, beta=8.0, eta=0.4, projection=proj, **kw)
                xin = x if use_map else rng.uniform(0.05, 0.95, size=X.shape[0])
                tag = f"{ftype}_{'proj' if proj else 'lin'}_{'map' if use_map else 'nomap'}_"
                f[tag + "x"] = xin
                f[tag + "rho"] = flt.apply(xin.copy())
                f[tag + "grad"] = flt.apply_gradient(g.copy(), xin.copy())
    save("g7_node_filter", **f)


CASES = {
    "g1_basic": lambda: case_g1("BasicLanczos"),
    "g1_iram": lambda: case_g1("IRAM"),
    "g2_basic": lambda: case_g2("BasicLanczos"),
    "g2_iram": lambda: case_g2("IRAM"),
    "g3_eps1e-1_basic": lambda: case_g3("BasicLanczos", 0.1, "eps1e-1"),
    "g3_eps1e-8_basic": lambda: case_g3("BasicLanczos", 1e-8, "eps1e-8"),
    "g3_eps1e-8_iram": lambda: case_g3("IRAM", 1e-8, "eps1e-8"),
    "g4_basic": lambda: case_g4("BasicLanczos"),
    "g4_iram": lambda: case_g4("IRAM"),
    "g5": case_g5,
    "g6": case_g6,
    "g7": case_g7,
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default=None)
    args = ap.parse_args()
    if args.case is not None:
        CASES[args.case]()
    else:
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
        for name in CASES:
            print("==", name, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--case", name], check=True, env=env)
