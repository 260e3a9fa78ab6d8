#!/usr/bin/env python3
"""development aid: spread of the G3 epsilon = 1e-8 IRAM chain's df/dx over start vectors of the eigensolver (how far
below the conditioning of the repeated-pair branch a gate against the reference's stored value can be)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    from conftest import corr_from, exact_pair_coefficients, load_golden, pair_rounding_in_dfdx, relerr
    from eigd_amd import design
    from eigd_amd.device import ElementBilinear, default_context
    from eigd_amd import tuning

    tuning.iram_seed = int(os.environ.get("SEED", "12345"))

    name = sys.argv[2]
    g = load_golden(name)
    ctx = default_context()
    flt = design.NodeFilter(g["conn"], g["X"], r0=float(g["r0"]), dvmap=g["dvmap"], num_design_vars=int(g["num_design_vars"]), ctx=ctx)
    an = design.ModalAnalysis(g["conn"], g["X"], kind="thermal", fltr=flt, N=8, m=60, sigma=float(g["sigma"]),
                              solver_type="IRAM" if "iram" in name else "BasicLanczos", tol=0.0, rtol=1e-12, p=float(g["p"]),
                              kappa=float(g["kappa"]), heat_capacity=float(g["heat_capacity"]), density=float(g["density"]),
                              beta=float(g["th_beta"]), ctx=ctx)
    lam, Q = an.initialize(g["x"])
    Qb, lamb = design.thermal_compliance_seeds(lam, Q, g["vec"])
    out = an.finalize_adjoint(Qb, lamb)
    ref_data = corr_from(g, "corr")
    dAdx = ElementBilinear.from_device(ctx, an.elem_dofs, an.Ke0, an._dKs)
    dBdx = ElementBilinear.from_device(ctx, an.elem_dofs, an.Me0, an._dMs)
    exact = exact_pair_coefficients(g["lam"], g["Phi"], g["Qb"], ref_data)
    d_rhoEb = pair_rounding_in_dfdx(ref_data, exact, g["Phi"], dAdx, dBdx)
    np.save(sys.argv[3], out["rhoEb"])
    print(f"seed {os.environ.get('SEED')}: vs corrected reference {relerr(out['rhoEb'], g['rhoEb'] + d_rhoEb):.3e}, raw "
          f"{relerr(out['rhoEb'], g['rhoEb']):.3e}", flush=True)
    sys.exit(0)
name = sys.argv[1] if len(sys.argv) > 1 else "g3_thermal32_eps1e-8_iram"
res = []
for seed in (12345, 1, 2, 3, 4, 5, 6, 7):
    f = f"/tmp/g3_{seed}.npy"
    subprocess.run([sys.executable, __file__, "one", name, f], env=dict(os.environ, SEED=str(seed)), check=True)
    res.append(np.load(f))
res = np.array(res)
mean = res.mean(axis=0)
print("spread of the runs around their mean:", [f"{np.linalg.norm(r - mean) / np.linalg.norm(mean):.2e}" for r in res])
