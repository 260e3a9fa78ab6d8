#!/usr/bin/env python3
"""
Headline benchmark: adjoint eigenvector-derivative modes per second on the 1M-dof buckling
problem (BASELINE.json configs[2]; SURVEY.md section 8d "C3").

  step   = solve_adjoint(Phib, method="sibk", rtol=1e-10) + add_total_derivative(...)  for N = 32 modes
           -- the reference's "adjoint solution time" + "total derivative time"
           (examples/buckling.py:905, 984) -- with every operand already resident in HBM.
  value  = N * steps / wall time  (whole job; with --gpus G the modes are sharded over the ranks and
           df/dx is all-reduced once per step: strong scaling of a fixed 32-mode job).

Untimed preamble per rank (reported in the JSON): synthetic K(rho), fundamental path u = K^-1 f,
G(u), shift selection, factorisation of K + sigma G, the IRAM eigensolve.

  python bench.py [--gpus G] [--steps 10] [--warmup 2]
      G > 1 without a launcher: this process starts G rank processes itself (before it touches the GPU; it never
      does) and relays rank 0's JSON line.
  python -m torch.distributed.run --nproc-per-node G ... bench.py --gpus G ...
      one rank per GPU from RANK / LOCAL_RANK / WORLD_SIZE; the collective is RCCL through the C ABI
      (eigd_comm_init / eigd_allreduce_sum of include/eigd_hip.h), no PyTorch in the rank processes.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md), GB/s


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def estimate_first_buckling_load(ctx, dK, dG, Kfac, n, iters=40):
    """power iteration on K^-1 (-G): the largest mu = 1 / BLF_1 (upper estimate of BLF_1)"""
    x = ctx.from_host(np.random.default_rng(3).uniform(-1, 1, size=n))
    t = ctx.empty(n, 1)
    mu = 0.0
    for _ in range(iters):
        dG.apply(x, t, alpha=-1.0)
        num = float(x.coldot(t)[0])
        dK.apply(x, t)
        den = float(x.coldot(t)[0])
        mu = num / den
        dG.apply(x, t, alpha=-1.0)
        Kfac.solve_device(t)
        nrm = float(t.colnorms()[0])
        x.assign_lincomb([(1.0 / nrm, t)])
    return 1.0 / mu


def _tail(path, nbytes=4000):
    try:
        with open(path, "rb") as fh:
            fh.seek(0, os.SEEK_END)
            fh.seek(max(0, fh.tell() - nbytes))
            return fh.read().decode("utf-8", "replace")
    except OSError:
        return ""


def launch_ranks(nranks, argv=None, limit_s=None, poll_s=0.05):
    """
    ``python bench.py --gpus N`` without a launcher: start N rank processes (fresh interpreters; this parent makes no
    HIP call before or after), one per GPU, and watch ALL of them: the first rank that exits non-zero ends the launch
    -- the others would wait for it in the collective for ever -- its stderr tail is relayed, the rest are terminated
    (their own process groups; nothing that touched a GPU is ever re-executed) and the exit code is non-zero.  The whole
    launch has a wall-clock limit (``EIGD_LAUNCH_TIMEOUT``, default 1500 s).  Rank 0's stdout (the JSON line) and
    stderr pass through.
    """
    import signal

    argv = sys.argv[1:] if argv is None else list(argv)
    limit_s = float(os.environ.get("EIGD_LAUNCH_TIMEOUT", "1500")) if limit_s is None else float(limit_s)
    rdv = tempfile.mkdtemp(prefix="eigd_comm_")
    procs, logs = [], []
    out0_path = os.path.join(rdv, "rank0.out")
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), EIGD_COMM_DIR=rdv,
                   EIGD_DEVICE=str(r), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        err_path = os.path.join(rdv, f"rank{r}.err")
        logs.append(err_path)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + argv, env=env, start_new_session=True,
            stdout=open(out0_path, "wb") if r == 0 else subprocess.DEVNULL,
            stderr=None if r == 0 else open(err_path, "wb")))

    def stop_all():
        for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            live = [p for p in procs if p.poll() is None]
            if not live:
                return
            for p in live:
                try:
                    os.killpg(p.pid, sig)          # the rank's own session: exactly the processes it started
                except (ProcessLookupError, PermissionError):
                    pass
            t_end = time.monotonic() + grace
            while time.monotonic() < t_end and any(p.poll() is None for p in live):
                time.sleep(poll_s)

    t0 = time.monotonic()
    verdict = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            r, c = bad[0]
            print(f"[bench] rank {r} of {nranks} exited with code {c}; stopping the other ranks", file=sys.stderr)
            if r != 0:
                sys.stderr.write(_tail(logs[r]))
            stop_all()
            verdict = 1
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() - t0 > limit_s:
            print(f"[bench] launch of {nranks} ranks exceeded {limit_s:.0f} s; stopping "
                  f"{[r for r, c in enumerate(codes) if c is None]}", file=sys.stderr)
            stop_all()
            verdict = 1
            break
        time.sleep(poll_s)
    if verdict == 0:
        sys.stdout.write(_tail(out0_path, 1 << 22))
        sys.stdout.flush()
    sys.stderr.flush()
    try:
        for fn in os.listdir(rdv):
            os.unlink(os.path.join(rdv, fn))
        os.rmdir(rdv)
    except OSError:
        pass
    return verdict


def measured_traffic(entry, sources):
    """
    HBM bytes per launch from the committed PMC passes (profiles/r03_traffic.json, written by tools/pmc_report.py from
    the rocprofv3 --pmc CSVs), valid only for the kernel sources they were taken with: a changed kernel file gives null.
    """
    rec = None
    for name in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):      # the newest round's passes first
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))[entry]
            break
        except (OSError, KeyError, ValueError):
            continue
    if rec is None:
        return None, "no PMC record"
    for fn in sources:
        sha = hashlib.sha256(open(os.path.join(ROOT, "eigd_amd", "csrc", fn), "rb").read()).hexdigest()[:16]
        if rec.get("source_sha16", {}).get(fn) != sha:
            return None, f"stale: {fn} changed since the PMC pass ({rec.get('files')})"
    return rec["traffic_bytes_per_launch"], rec.get("files")


MFMA_F64_PEAK_TFLOPS = 78.6   # v_mfma_f64_16x16x4 holds a SIMD's matrix pipe 64 cycles (PMC: BUSY_CYCLES / INSTS = 64.0 on
                              # every sweep kernel): 2048 flop / 64 cycles x 4 SIMDs x 256 CUs x 2.4 GHz


def measured_mfma(entry, sources, useful_flops, us_per_launch):
    """
    Compute ceiling of the sweep from the committed MFMA counter pass (same record and staleness rule as the traffic):
    fp64 MFMA flops issued against the useful ones (4 nnz(L) k: every entry of L meets every column once per direction),
    the time the issued ones need at the matrix pipe's peak, and the fraction of the launch the pipe was busy.
    """
    rec = None
    for name in ("r05_traffic.json", "r04_traffic.json"):      # the newest round's pass first
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))[entry]
            break
        except (OSError, KeyError, ValueError):
            pass
    if rec is None or "mfma_f64_insts_per_launch" not in rec:
        return None
    for fn in sources:
        sha = hashlib.sha256(open(os.path.join(ROOT, "eigd_amd", "csrc", fn), "rb").read()).hexdigest()[:16]
        if rec.get("source_sha16", {}).get(fn) != sha:
            return None
    issued = rec["mfma_f64_insts_per_launch"] * 2048.0
    pipe_us = issued / (MFMA_F64_PEAK_TFLOPS * 1e12) * 1e6
    return {"flops_useful": useful_flops, "flops_issued_mfma_f64": issued, "issued_over_useful": round(issued / useful_flops, 3),
            "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "achieved": round(useful_flops / (us_per_launch * 1e-6) / 1e12, 2),
            "frac": round(useful_flops / (us_per_launch * 1e-6) / 1e12 / MFMA_F64_PEAK_TFLOPS, 4),
            "pipe_time_us_if_perfectly_spread": round(pipe_us, 1), "pipe_time_over_launch": round(pipe_us / us_per_launch, 3),
            "hbm_time_us_at_peak": round(rec["traffic_bytes_per_launch"] / 8e12 * 1e6, 1),
            "note": "bound = whichever ceiling is nearer: the bytes really moved need hbm_time_us_at_peak at 8 TB/s, the MFMAs "
                    "really issued need pipe_time_us; the launch takes us_per_launch"}


class _LastOfMany:
    """stand-in communicator of the LAST rank of P (it owns the slowest mode): no collective, partial results"""

    def __init__(self, p):
        self.rank, self.size = p - 1, p

    def allreduce_sum(self, a):
        return a

    def allreduce_max(self, x):
        return x

    def barrier(self):
        pass


def scaling_model_of(ctx, rank_step, N, ms_per_step, what):
    """
    What mode sharding can give: the block-cyclic share of the slowest rank of P = 2, 4, 8, timed on this one GPU with a
    stand-in communicator (the 4 MB all-reduce of df/dx adds < 0.1 ms).  efficiency = ms_per_step / (P * rank_ms).
    """
    model = {"note": f"slowest rank's share of the same {N}-mode step ({what}), timed on this one GPU (no collective); "
                     "efficiency = ms_per_step / (P * rank_ms)", "ms_per_step_one_gpu": round(ms_per_step, 3), "ranks": {}}
    for P_ in (2, 4, 8):
        if P_ > N:
            continue
        cm = _LastOfMany(P_)
        rank_step(cm)
        ctx.sync()
        times = []
        for _ in range(3):                                 # the median of three: a share's first calls allocate its block shapes
            t0 = time.perf_counter()
            rank_step(cm)
            ctx.sync()
            times.append(time.perf_counter() - t0)
        t_r = sorted(times)[1]
        model["ranks"][str(P_)] = {"rank_ms": round(1e3 * t_r, 3), "modes_of_rank": int(len(range(P_ - 1, N, P_))),
                                   "predicted_efficiency": round(ms_per_step * 1e-3 / (P_ * t_r), 3)}
    return model


def c5_setup(args):
    """config C5 up to the eigenpairs: (box, dev, ctx, solver, sigma, fstats, lam, times)"""
    import eigd_amd as eg
    from eigd_amd.problems import ShellBox, ShellBoxOnDevice

    N, m = 64, 129
    t0 = time.perf_counter()
    box = ShellBox(832, 160, 40, nseg=2, seed=0)
    dev = ShellBoxOnDevice(box)
    ctx = dev.ctx
    dev.assemble()
    # a positive shift below the first buckling load, from the inertia of the factorisation (doubling + bisection)
    good, bad, sigma = 0.0, None, 0.25
    while bad is None:
        if dev.refactor(sigma) == 0:
            good, sigma = sigma, 2.0 * sigma
        else:
            bad = sigma
    for _ in range(3):
        mid = 0.5 * (good + bad)
        good, bad = (mid, bad) if dev.refactor(mid) == 0 else (good, mid)
    sigma = 0.9 * good
    assert dev.refactor(sigma) == 0
    ctx.sync()
    t_setup = time.perf_counter() - t0
    fstats = dev.factor.factor.stats()
    log(0, f"C5: n={box.n} nnz(K)={dev.dK.nnz} nnz(L)={fstats['nnzL']} sigma={sigma:.4f} set-up {t_setup:.1f}s")
    t0 = time.perf_counter()
    solver = eg.IRAM(N=N, m=m, mode="buckling", ctx=ctx)
    lam, Phi = solver.solve(dev.dG, dev.dK, dev.factor, sigma)
    ctx.sync()
    t_eig = time.perf_counter() - t0
    log(0, f"C5: eigensolve {t_eig:.2f}s, {solver.n_restarts} restarts; BLF = {lam[:3]} ... {lam[-1]:.3f}")
    rng = np.random.default_rng(1)
    Phib = rng.uniform(-1, 1, size=(box.n, N))
    w = rng.uniform(0.5, 1.5, size=N)
    dPhib = ctx.from_host(Phib)
    dAdx, dBdx = dev.callbacks()

    def step(comm=None):
        dpsi, data = solver.solve_adjoint(dPhib, method="sibk", rtol=args.rtol, update_guess=False, bs_target=1, maxiter=80,
                                          comm=comm)
        dfdx = solver.add_total_derivative(w, dPhib, dpsi, dAdx, dBdx, np.zeros(box.ngroups), adj_corr_data=data,
                                           deriv_type="tensor", comm=comm)
        return dpsi, data, dfdx

    return box, dev, ctx, solver, sigma, fstats, Phib, dPhib, step, (t_setup, t_eig), (N, m)


def c5_scaling_model(args):
    """the scaling model of config C5 (64 modes: 8 per rank at 8 GPUs) for the default run's JSON line"""
    box, dev, ctx, solver, sigma, fstats, Phib, dPhib, step, times, (N, m) = c5_setup(args)
    step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(2):
        step()
    ctx.sync()
    ms = 1e3 * (time.perf_counter() - t0) / 2
    model = scaling_model_of(ctx, lambda cm: step(cm), N, ms, f"config C5: shell box, {box.n / 1e6:.2f}M dof, {N} modes")
    ctx.release_workspaces()
    return model


def main_c5(args):
    """
    Config C5 (stand-in for the reference's CRM wingbox, examples/crm.py): thin-walled shell box, 6 dof per
    node, ~2.0 M dof, 64 modes, IRAM m = 129; step = solve_adjoint (sibk, rtol 1e-10, 80 Krylov vectors) +
    add_total_derivative w.r.t. the wall-thickness groups, operands resident in HBM.  One GPU.
    """
    box, dev, ctx, solver, sigma, fstats, Phib, dPhib, step, (t_setup, t_eig), (N, m) = c5_setup(args)
    rng = np.random.default_rng(2)

    for _ in range(max(args.warmup, 1)):  # (the first call allocates ~170 GB of Krylov workspace)
        step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dpsi, data, dfdx = step()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    res, _ = solver.eval_adjoint_residual_norm(dPhib, dpsi, b_ortho=True)
    Xs, Xo = ctx.from_host(rng.normal(size=(box.n, 32))), ctx.empty(box.n, 32)
    for _ in range(2):
        dev.factor.solve_device_to(Xs, Xo)
    ctx.sync()
    ctx.timer_start()
    for _ in range(5):
        dev.factor.solve_device_to(Xs, Xo)
    sweep_ms = ctx.timer_stop_ms() / 5
    sweep_bytes = dev.factor.factor.solve_bytes(32)
    achieved = sweep_bytes / (sweep_ms * 1e-3) / 1e9
    out = {
        "metric": "adjoint_mode_derivatives_per_sec", "value": round(N * args.steps / elapsed, 3), "unit": "modes/s",
        "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"shell box (stand-in for the CRM wingbox), {box.n / 1e6:.2f}M dof, 6 dof/node, {N} modes, "
                               f"IRAM m={m} + sibk rtol={args.rtol:g} (80 Krylov vectors) + derivative w.r.t. "
                               f"{box.ngroups} wall-thickness groups",
                   "n_dof": int(box.n), "nnz": int(dev.dK.nnz), "modes": N, "m": m, "sigma": round(float(sigma), 6)},
        "roofline": {"kernel": "one 32-column sweep of the factor, all tree levels", "bound": "hbm",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_launch": sweep_bytes,
                     "us_per_launch": round(sweep_ms * 1e3, 1), "columns": 32, "nnzL": fstats["nnzL"]},
        "cpu_baseline": None,
        "accuracy": {"adjoint_residual_rel_max": float(np.max(res) / np.linalg.norm(Phib, axis=0).max()),
                     "sibk_iterations_max": int(max(solver.last_info))},
        "preamble_s": {"setup_s": round(t_setup, 2), "eigensolve_s": round(t_eig, 2)},
        "scaling_model": (None if args.no_scaling_model else
                          scaling_model_of(ctx, lambda cm: step(cm), N, 1e3 * elapsed / args.steps, "config C5")),
    }
    print(json.dumps(out), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nx", type=int, default=706)
    ap.add_argument("--ny", type=int, default=706)
    ap.add_argument("--modes", type=int, default=32)
    ap.add_argument("--m", type=int, default=65)
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--cpu-sample", choices=("full", "none"), default="full",
                    help="CPU baseline: SuperLU factor of the same matrix + the oracle's laa + sibk on a sample of the modes")
    ap.add_argument("--cpu-modes", type=int, default=4, help="candidate modes of the CPU sample (spread over the spectrum)")
    ap.add_argument("--cpu-mode-list", default=None,
                    help="comma-separated modes the CPU sample must solve whatever the budget (e.g. 0,10,31: one of the slow "
                         "high modes, ~60 s of SuperLU solves, for the record under profiles/)")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0,
                    help="the CPU sample stops adding modes once its sibk time exceeds this (at least two modes run)")
    ap.add_argument("--numpy-steps", type=int, default=3, help="extra steps through the numpy-in / numpy-out call surface")
    ap.add_argument("--spmv-reps", type=int, default=200)
    ap.add_argument("--streams", type=int, default=None,
                    help="mode groups solved concurrently on separate HIP streams (default: EIGD_STREAMS or 1)")
    ap.add_argument("--workload", choices=("c3", "c5"), default="c3",
                    help="c3: the headline 1M-dof buckling column (BASELINE configs[2]); c5: the 2M-dof shell box, 64 modes "
                         "(stand-in for BASELINE configs[4], single GPU)")
    ap.add_argument("--ordering", choices=("geometric", "algebraic"), default="geometric",
                    help="nested dissection with the mesh coordinates as a hint, or purely from the matrix graph")
    ap.add_argument("--pyprofile", default=None, help="write a cProfile summary of one extra step to this file")
    ap.add_argument("--emulate-rank", default=None, help="r/P: time the mode share of rank r of P on this GPU (development aid)")
    ap.add_argument("--trace", default=None, help="write the per-iteration host timeline of one extra step to this file")
    ap.add_argument("--no-fd-check", action="store_true", help="skip the directional finite-difference check of df/dx")
    ap.add_argument("--no-arnoldi-leg", action="store_true", help="skip the untimed comparison solve in the Arnoldi form")
    ap.add_argument("--no-scaling-model", action="store_true", help="skip the slowest-rank timings for P = 2, 4, 8")
    ap.add_argument("--no-extras0-leg", action="store_true", help="skip the untimed step without extra deflated pairs")
    ap.add_argument("--no-c5-scaling-model", action="store_true", help="skip the scaling model of config C5 (64 modes)")
    ap.add_argument("--dump-dfdx", default=None, help="rank 0 saves the df/dx of the last timed step to this .npy file")
    ap.add_argument("--force-launch", action="store_true",
                    help="start the rank processes through the launcher even for --gpus 1 (test of the launcher)")
    args = ap.parse_args()

    if args.workload == "c5":
        return main_c5(args)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_launch):
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        os.environ["EIGD_DEVICE"] = str(local_rank)
    if args.gpus != world:
        log(rank, f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    comm = None
    if args.emulate_rank and world == 1:
        # development aid: time the share of ONE rank of a P-rank job on this GPU (no collective; results are partial)
        class _OneOfMany:
            def __init__(self, r, p):
                self.rank, self.size = r, p

            def allreduce_sum(self, a):
                return a

            def allreduce_max(self, x):
                return x

            def barrier(self):
                pass

        r_, p_ = (int(x) for x in args.emulate_rank.split("/"))
        comm = _OneOfMany(r_, p_)

    import eigd_amd as eg
    from eigd_amd.device import CSRMatrix, ElementBilinear, default_context
    from eigd_amd.problems import BucklingColumn

    ctx = default_context()
    if os.environ.get("EIGD_TEST_FAIL_RANK") == str(rank):   # launcher test: this rank dies before the communicator
        raise SystemExit(f"rank {rank}: EIGD_TEST_FAIL_RANK set, leaving before eigd_comm_init")
    if world > 1:
        from eigd_amd.comm import RcclComm

        comm = RcclComm(ctx, rank, world)           # RCCL over xGMI, one rank per GPU (collective call)
        log(rank, f"RCCL communicator up: {world} ranks")
    N = args.modes
    timing = {}

    # ------------------------------------------------------------------ problem (untimed preamble)
    t0 = time.perf_counter()
    col = BucklingColumn(args.nx, args.ny, Lx=1.0, Ly=1.0, seed=0)  # rhoE ~ U(0.3, 1), default_rng(0)
    K = col.stiffness()
    n = K.shape[0]
    timing["assemble_K_s"] = time.perf_counter() - t0
    log(rank, f"K assembled: n={n} nnz={K.nnz} ({timing['assemble_K_s']:.1f}s)")
    t0 = time.perf_counter()
    coords = None if args.ordering == "algebraic" else col.dof_coords()
    Kfac = eg.SpLuOperator(K, ctx=ctx, check_symmetry=False, coords=coords)
    ctx.sync()
    timing["factor_K_s"] = time.perf_counter() - t0
    ur = Kfac(col.f[col.reduced])
    u = col.full_vector(ur)
    t0 = time.perf_counter()
    G = col.geometric_stiffness(u)
    timing["assemble_G_s"] = time.perf_counter() - t0
    dK, dG = CSRMatrix(ctx, K), CSRMatrix(ctx, G)
    blf_est = estimate_first_buckling_load(ctx, dK, dG, Kfac, n)
    sigma = 0.7 * blf_est
    log(rank, f"G assembled ({timing['assemble_G_s']:.1f}s); BLF_1 ~ {blf_est:.4f}; shift sigma = {sigma:.4f}")
    t0 = time.perf_counter()
    while True:
        mat = (K + sigma * G).tocsr()
        factor = eg.SpLuOperator(mat, ctx=ctx, symbolic=Kfac.symbolic if mat.nnz == K.nnz else None,
                                 check_symmetry=False, coords=coords)
        if factor.negative_pivots == 0:  # inertia: no buckling load below the shift
            break
        sigma *= 0.5
        log(rank, f"shift not below BLF_1 ({factor.negative_pivots} negative pivots), retrying with sigma = {sigma:.4f}")
    ctx.sync()
    timing["factor_shifted_s"] = time.perf_counter() - t0
    fstats = factor.factor.stats()
    del Kfac
    t0 = time.perf_counter()
    solver = eg.IRAM(N=N, m=args.m, mode="buckling", ctx=ctx)
    lam, Phi = solver.solve(G, K, factor, sigma)
    ctx.sync()
    timing["eigensolve_s"] = time.perf_counter() - t0
    eig_count = factor.count
    # the same eigensolve once more, as every design point after the first sees it (work blocks, page-locked result
    # buffers and the coefficient slots come from the pools the first one filled)
    # (twice: the second large result of a size is the one that page-locks its buffer, the third finds it in the pool)
    del lam, Phi
    for _ in range(2):
        solver = lam = Phi = None
        solver = eg.IRAM(N=N, m=args.m, mode="buckling", ctx=ctx)
        ctx.sync()
        t0 = time.perf_counter()
        lam, Phi = solver.solve(dG, dK, factor, sigma)     # (the matrices as they sit in HBM: no upload, as after a device assembly)
        ctx.sync()
        timing["eigensolve_repeat_s"] = time.perf_counter() - t0
    factor.count = eig_count
    eig_info = {"block_size": int(getattr(solver, "block_size", 1)), "internal_basis": int(getattr(solver, "internal_basis", args.m)),
                "restarts": int(solver.n_restarts), "sweeps": int(getattr(solver, "sweeps", eig_count)),
                "factor_applications": int(eig_count), "extra_pairs_for_deflation": int(solver.n_extra),
                # the true-residual check of the returned pairs inside solve(): one N-column sweep + product, inside
                # eigensolve_s but counted neither in `sweeps` nor in `factor_applications` (a check, not a solve)
                "residual_check_sweeps": 1, "residual_check_columns": int(N)}
    log(rank, f"eigensolve: {timing['eigensolve_s']:.2f}s (repeated: {timing['eigensolve_repeat_s']:.2f}s), {eig_info}; "
              f"BLF = {lam[:4]} ... {lam[-1]:.4f}")

    rng = np.random.default_rng(1)
    Phib = rng.uniform(size=(n, N))
    lamb = rng.uniform(size=N)
    dPhib = ctx.from_host(Phib)
    scale = col.dK_scale()
    dBdx = ElementBilinear(ctx, col.elem_dofs, col.Ke0, scale=scale)            # d(w^T K v)/d rhoE
    dAdx = ElementBilinear(ctx, col.elem_dofs, col.Ge_unit, scale=col.dG_scale())  # d(w^T G v)/d rhoE at fixed u
    ndv = col.mesh.nelems

    import eigd_amd.adjoint as _adj

    def step():
        factor.count = 0
        _adj.LAST_ROUND["gs_cycles"] = _adj.LAST_ROUND["gs_correcting_passes"] = 0
        _adj.LAST_ROUND["post_gs_projections"] = _adj.LAST_ROUND["post_gs_updates_applied"] = 0
        _adj.LAST_ROUND["cycles_enqueued_for_nothing"] = _adj.LAST_ROUND["cycles_waited_for"] = 0
        dpsi, data = solver.solve_adjoint(dPhib, method="sibk", rtol=args.rtol, update_guess=False, bs_target=1,
                                          comm=comm, streams=args.streams)
        dfdx = solver.add_total_derivative(lamb, dPhib, dpsi, dAdx, dBdx, np.zeros(ndv), adj_corr_data=data,
                                           deriv_type="tensor", comm=comm)
        return dpsi, data, dfdx

    def fence():
        ctx.sync()
        if comm is not None and world > 1:
            comm.barrier()
            ctx.sync()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    step_times = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        dpsi, data, dfdx = step()
        step_times.append(time.perf_counter() - ts)
    fence()
    elapsed = time.perf_counter() - t0
    log(rank, "step times (s):", [round(t, 3) for t in step_times])
    if comm is not None:
        elapsed = comm.allreduce_max(elapsed)       # the slowest rank's wall time
    adj_count = factor.count
    last_round = dict(_adj.LAST_ROUND)              # the counters of the last timed step (later legs run more solves)
    sibk_iterations = [int(i) for i in solver.last_info]
    args.timed_sibk_iterations = sibk_iterations
    ms_per_step = 1e3 * elapsed / args.steps
    value = N * args.steps / elapsed

    if rank != 0:
        return
    if args.dump_dfdx:
        np.save(args.dump_dfdx, dfdx)
    if args.trace:
        import eigd_amd.adjoint as adj

        events, orig = [], adj._active_range

        def traced(done):
            r = orig(done)
            events.append((time.perf_counter(), r[0], r[1], int(np.count_nonzero(~done))))
            return r

        ratios, orig_lstsq = [], adj.solve_shifted_lstsq

        def lstsq_traced(alpha, H, r):
            j = H.shape[1]
            ratios.append((j, H[j, j - 1] / max(np.linalg.norm(H[:j, j - 1]), 1e-300)))
            return orig_lstsq(alpha, H, r)

        adj.solve_shifted_lstsq = lstsq_traced
        adj._active_range = traced
        ctx.sync()
        t_begin = time.perf_counter()
        step()
        ctx.sync()
        t_end = time.perf_counter()
        adj._active_range = orig
        adj.solve_shifted_lstsq = orig_lstsq
        with open(args.trace, "w") as fh:
            for j in sorted({q for q, _ in ratios}):
                rs = [x for q, x in ratios if q == j]
                fh.write(f"step {j:3d}: h_next/|h| over live modes: min {min(rs):.3e} median {np.median(rs):.3e} max {max(rs):.3e}\n")
            prev = t_begin
            for it, (t, lo, hi, live) in enumerate(events):
                fh.write(f"iter {it:3d}  +{1e3 * (t - prev):7.3f} ms  range [{lo:2d},{hi:2d})  live {live}\n")
                prev = t
            fh.write(f"tail (psi update, correction, derivative) {1e3 * (t_end - prev):.3f} ms; total {1e3 * (t_end - t_begin):.3f} ms\n")
    if args.pyprofile:
        import cProfile
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        step()
        ctx.sync()
        pr.disable()
        with open(args.pyprofile, "w") as fh:
            pstats.Stats(pr, stream=fh).sort_stats("cumulative").print_stats(45)
    # ------------------------------------------------------------------ the reference's call surface: numpy in, numpy out
    # (this leg runs BEFORE the Arnoldi and scaling-model legs: behind the Arnoldi leg's downloads and release_workspaces()
    # both transfers of the numpy step ran at half their rate on the same box -- 9.6 + 8.8 ms against 4.7 + 4.7 -- although
    # the caller's array was page-locked and the result came from the page-locked pool: tools/pin_probe.py, docs/LOG.md)
    # (the timed value keeps the operands resident in HBM; callers of the reference hand numpy arrays to solve_adjoint
    # and add_total_derivative, which adds the H2D of Phib and the D2H / H2D of psi: reported next to the value)
    numpy_api = None
    if world == 1 and comm is None and args.numpy_steps > 0:
        Phib_np = Phib.copy()                          # (the leg's own array: its content is changed from step to step)
        t_each = []
        for _ in range(args.numpy_steps + 2):          # two untimed: page-locked result buffers and Phib's registration
            # a design loop brings new right-hand sides every step: the content of the caller's array changes (outside
            # the timed region), so the device copy kept from the last step is found stale and Phib is uploaded again --
            # by both calls; what the kept copy saves is the upload of psi (tuning.host_twins = "returned")
            np.multiply(Phib_np, 1.0 + 1e-9, out=Phib_np)
            ctx.sync()
            t0 = time.perf_counter()
            psi_np, data_np = solver.solve_adjoint(Phib_np, method="sibk", rtol=args.rtol, update_guess=False, bs_target=1)
            t_mid = time.perf_counter()
            solver.add_total_derivative(lamb, Phib_np, psi_np, dAdx, dBdx, np.zeros(ndv), adj_corr_data=data_np,
                                        deriv_type="tensor")
            ctx.sync()
            t_each.append(time.perf_counter() - t0)
            t_solve_np = t_mid - t0
        t_np = float(np.mean(t_each[2:]))
        # where the host waits in one more such step: wall time inside every C-ABI entry point (blocking calls include the
        # wait for the kernels enqueued before them)
        from eigd_amd import _ffi as _ffi_mod
        from eigd_amd import device as _dev_mod
        from eigd_amd import adjoint as _adj_mod
        abi_ms, orig_call = {}, _ffi_mod.call

        def timed_call(name, *a):
            tq = time.perf_counter()
            try:
                return orig_call(name, *a)
            finally:
                abi_ms[name] = abi_ms.get(name, 0.0) + 1e3 * (time.perf_counter() - tq)

        for mod in (_ffi_mod, _dev_mod, _adj_mod):
            if hasattr(mod, "call"):
                setattr(mod, "call", timed_call)
        try:
            np.multiply(Phib_np, 1.0 + 1e-9, out=Phib_np)
            ctx.sync()
            psi_np, data_np = solver.solve_adjoint(Phib_np, method="sibk", rtol=args.rtol, update_guess=False, bs_target=1)
            solver.add_total_derivative(lamb, Phib_np, psi_np, dAdx, dBdx, np.zeros(ndv), adj_corr_data=data_np,
                                        deriv_type="tensor")
            ctx.sync()
        finally:
            for mod in (_ffi_mod, _dev_mod, _adj_mod):
                if hasattr(mod, "call"):
                    setattr(mod, "call", orig_call)
        abi_top = {k: round(v, 3) for k, v in sorted(abi_ms.items(), key=lambda kv: -kv[1])[:8]}
        # the two transfers by themselves (idle GPU): H2D of the caller's (page-locked) Phib, D2H of a block of psi's size
        ctx.sync()
        tq = time.perf_counter()
        blk_t = ctx.from_host(Phib_np)
        ctx.sync()
        t_h2d = time.perf_counter() - tq
        tq = time.perf_counter()
        host_t = blk_t.get()
        t_d2h = time.perf_counter() - tq
        del blk_t, host_t
        # the same two calls on device blocks, timed apart (the value's step is their sum)
        ctx.sync()
        t0 = time.perf_counter()
        dpsi_t, data_t = solver.solve_adjoint(dPhib, method="sibk", rtol=args.rtol, update_guess=False, bs_target=1)
        ctx.sync()
        t_solve_dev = time.perf_counter() - t0
        del dpsi_t, data_t
        numpy_api = {"value": round(N / t_np, 3), "unit": "modes/s", "ms_per_step": round(1e3 * t_np, 3),
                     "solve_adjoint_ms": round(1e3 * t_solve_np, 3), "solve_adjoint_on_device_blocks_ms": round(1e3 * t_solve_dev, 3),
                     "host_wait_by_entry_point_ms": abi_top,
                     "transfer_alone_ms": {"h2d_phib": round(1e3 * t_h2d, 3), "d2h_psi": round(1e3 * t_d2h, 3)},
                     "steps": args.numpy_steps, "first_calls_ms": [round(1e3 * t, 1) for t in t_each[:2]],
                     "host_twins": str(__import__("eigd_amd").tuning.host_twins),
                     "note": "same step with numpy arrays in and out: the caller's Phib is transferred by BOTH calls (it is "
                             "read as given, like the reference does), psi is downloaded once and comes back read-only: "
                             "handed to add_total_derivative as it is, the device block it came from is used again "
                             "(tuning.host_twins = 'returned'; an in-place edit of it raises).  Three transfers of "
                             "n x N doubles per step.  Results live in pooled page-locked memory, the caller's Phib is "
                             "page-locked in place from its second use; the first two calls (listed) pay for that once"}
        if args.pyprofile:
            import cProfile
            import pstats

            pr = cProfile.Profile()
            pr.enable()
            psi_np, data_np = solver.solve_adjoint(Phib, method="sibk", rtol=args.rtol, update_guess=False, bs_target=1)
            solver.add_total_derivative(lamb, Phib, psi_np, dAdx, dBdx, np.zeros(ndv), adj_corr_data=data_np,
                                        deriv_type="tensor")
            ctx.sync()
            pr.disable()
            with open(args.pyprofile + ".numpy", "w") as fh:
                pstats.Stats(pr, stream=fh).sort_stats("cumulative").print_stats(60)
        del psi_np
        log(rank, f"numpy-in / numpy-out step: {1e3 * t_np:.1f} ms ({N / t_np:.1f} modes/s)")

    # ------------------------------------------------------------------ the Arnoldi form of the same solve, for comparison
    # (outside the timed region: the reference's recurrence with full Gram-Schmidt, two steps per pass -- round 3's solver)
    arnoldi_form = None
    if world == 1 and comm is None and last_round.get("recurrence") == "short" and not args.emulate_rank \
            and not args.no_arnoldi_leg:
        import eigd_amd as _eg

        it_short = list(sibk_iterations)              # (of the last timed step)
        keep = _eg.tuning.recurrence
        _eg.tuning.recurrence = "arnoldi"
        try:
            step()                                       # (allocates the Krylov stacks)
            ctx.sync()
            t0 = time.perf_counter()
            dpsi_a, _, dfdx_a = step()
            ctx.sync()
            t_a = time.perf_counter() - t0
            it_arn = [int(i) for i in solver.last_info]
        finally:
            _eg.tuning.recurrence = keep
        da, ds = dpsi_a.get(), dpsi.get()
        arnoldi_form = {"ms_per_step": round(1e3 * t_a, 3), "sibk_iterations": it_arn,
                        "total_iterations": int(sum(it_arn)), "total_iterations_short_recurrence": int(sum(it_short)),
                        "psi_rel_diff": float(np.linalg.norm(da - ds) / np.linalg.norm(da)),
                        "dfdx_rel_diff": float(np.linalg.norm(dfdx_a - dfdx) / np.linalg.norm(dfdx_a))}
        del da, ds, dpsi_a
        ctx.release_workspaces()                         # the Krylov stacks of the Arnoldi form (26 GB at C3) go back
        log(rank, f"Arnoldi form: {arnoldi_form['ms_per_step']} ms/step, iterations {sum(it_arn)} against {sum(it_short)}, "
                  f"psi rel diff {arnoldi_form['psi_rel_diff']:.1e}")
    # ------------------------------------------------------------------ the reference's deflation set: no extra pairs
    # (untimed leg: the eigensolver keeps up to 32 converged pairs beyond N for the adjoint stage's deflation -- paid for
    # in the untimed eigensolve; with extra = 0 the Krylov solves deflate the N requested pairs only, as the reference does)
    extras0 = None
    if world == 1 and comm is None and not args.emulate_rank and not args.no_extras0_leg:
        import eigd_amd as _eg

        keep_x = _eg.tuning.iram_extra
        _eg.tuning.iram_extra = 0
        try:
            # (timed as eigensolve_repeat_s of the main leg is: the third call -- the first ones allocate the panels of this
            # basis size and free what the legs before this one left in the pools)
            for _ in range(3):
                solver0 = None
                solver0 = _eg.IRAM(N=N, m=args.m, mode="buckling", ctx=ctx)
                ctx.sync()
                t0 = time.perf_counter()
                solver0.solve(dG, dK, factor, sigma)
                ctx.sync()
                t_eig0 = time.perf_counter() - t0

            # (eigenvectors are determined up to their signs, and with a FIXED right-hand side block df/dx follows them:
            # the same design derivative needs Phib's columns flipped along)
            sg0 = np.sign(np.einsum("ij,ij->j", np.asarray(solver0.Phi), Phi))
            dPhib0 = ctx.from_host(Phib * sg0)

            def step0():
                dpsi0, data0 = solver0.solve_adjoint(dPhib0, method="sibk", rtol=args.rtol, update_guess=False, bs_target=1)
                return dpsi0, solver0.add_total_derivative(lamb, dPhib0, dpsi0, dAdx, dBdx, np.zeros(ndv), adj_corr_data=data0,
                                                           deriv_type="tensor")

            step0()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                dpsi0, dfdx0 = step0()
            ctx.sync()
            t_x0 = (time.perf_counter() - t0) / 3
            extras0 = {"ms_per_step": round(1e3 * t_x0, 3), "value": round(N / t_x0, 3), "eigensolve_s": round(t_eig0, 4),
                       "sibk_iterations": [int(i) for i in solver0.last_info], "extra_pairs_for_deflation": int(solver0.n_extra),
                       "recurrence": _adj.LAST_ROUND.get("recurrence"),
                       "dfdx_rel_diff": float(np.linalg.norm(dfdx0 - dfdx) / np.linalg.norm(dfdx)),
                       "lam_rel_diff": float(np.max(np.abs(np.asarray(solver0.lam) - lam) / np.abs(lam))),
                       "design_point_s": None}
            del dpsi0, solver0, dPhib0
            log(rank, f"extra = 0 (the reference's deflation set): {extras0['ms_per_step']} ms/step, eigensolve {t_eig0:.3f}s")
        finally:
            _eg.tuning.iram_extra = keep_x
            factor.count = adj_count
    # ------------------------------------------------------------------ what mode sharding can give: the slowest rank's share
    # (one GPU: the block-cyclic share of the LAST rank of P -- it owns the slowest mode -- timed with a stand-in
    # communicator; no collective: the 4 MB all-reduce of df/dx adds < 0.1 ms.  A prediction for the next multi-GPU run.)
    scaling_model = None
    if world == 1 and comm is None and not args.no_scaling_model:
        def rank_step(cm):
            dpsi_r, data_r = solver.solve_adjoint(dPhib, method="sibk", rtol=args.rtol, update_guess=False, bs_target=1,
                                                  comm=cm, streams=args.streams)
            solver.add_total_derivative(lamb, dPhib, dpsi_r, dAdx, dBdx, np.zeros(ndv), adj_corr_data=data_r,
                                        deriv_type="tensor", comm=cm)

        scaling_model = scaling_model_of(ctx, rank_step, N, ms_per_step, "config C3")
        log(rank, f"scaling model (slowest rank of P on one GPU): {scaling_model['ranks']}")
    # ------------------------------------------------------------------ accuracy of the timed result
    res, ortho = solver.eval_adjoint_residual_norm(dPhib, dpsi, b_ortho=False) if world == 1 else (None, None)
    accuracy = {}
    if res is not None:
        accuracy["adjoint_residual_max"] = float(np.max(res))
        accuracy["adjoint_residual_rel_max"] = float(np.max(res) / np.sqrt(np.max(np.sum(Phib**2, axis=0))))
        accuracy["ortho_max"] = float(np.max(ortho))

    # ------------------------------------------------------------------ df/dx against a central finite difference
    # f(rho) = lamb . ln(lam(rho)) + sum_i Phib_i . phi_i(rho) at frozen fundamental path u (what the callbacks
    # differentiate; ln(lam): the reference's buckling-mode convention for lamb), along a random
    # direction: (f(rho + h p) - f(rho - h p)) / 2h  vs  p . dfdx   (as examples/buckling.py:1025-1035)
    if world == 1 and not args.no_fd_check:
        from eigd_amd.problems import _assemble

        t0 = time.perf_counter()
        rho_base = col.rhoE.copy()
        pert = np.random.default_rng(5).uniform(size=ndv)
        hfd = 1e-5

        def functional(rho):
            col.rhoE = rho
            Kp = col.stiffness()
            Gp, _ = _assemble(col.mesh, (rho**col.p + col.rho0_G)[:, None, None] * col.Ge_unit, 2, col.free_map)
            factor.refactor((Kp + sigma * Gp).tocsr())
            sv = eg.IRAM(N=N, m=args.m, mode="buckling", ctx=ctx)
            lp, Pp = sv.solve(Gp, Kp, factor, sigma)
            sg = np.sign(np.einsum("ij,ij->j", Pp, Phi))
            return float(lamb @ np.log(lp) + np.einsum("ij,ij->", Phib, Pp * sg))

        fplus = functional(rho_base + hfd * pert)
        fminus = functional(rho_base - hfd * pert)
        col.rhoE = rho_base
        factor.refactor((K + sigma * G).tocsr())
        fd = (fplus - fminus) / (2 * hfd)
        ans = float(pert @ dfdx)
        accuracy["dfdx_directional"] = ans
        accuracy["dfdx_central_difference"] = fd
        accuracy["dfdx_fd_rel_err"] = abs(ans - fd) / abs(fd)
        log(rank, f"df/dx check: adjoint {ans:.10e}  central difference {fd:.10e}  rel-err {accuracy['dfdx_fd_rel_err']:.2e} "
                  f"({time.perf_counter() - t0:.1f}s)")

    # ------------------------------------------------------------------ roofline of the dominant kernels (HIP events)
    # The k-column triangular sweep (fwd_thin/fwd_level + bwd_thin/bwd_level kernels, one launch per tree level and
    # direction) is where a step spends most of its time.  One "launch" below = one sweep of N columns through the factor.
    # Algorithmic bytes: every entry of L once per direction (2 * 8 * nnz(L)) plus the block
    # read and written once (16 n N).  HBM traffic: FETCH_SIZE (doubled, the guide's gfx950 correction, calibrated on
    # a stream of known size in the same run) + WRITE_SIZE from the PMC passes under profiles/; C3 default only.
    Xs = ctx.from_host(rng.normal(size=(n, N)))
    Xo = ctx.empty(n, N)
    for _ in range(3):
        factor.solve_device_to(Xs, Xo)
    ctx.sync()
    ctx.timer_start()
    for _ in range(10):
        factor.solve_device_to(Xs, Xo)
    sweep_ms = ctx.timer_stop_ms() / 10
    sweep_bytes = factor.factor.solve_bytes(N)
    default_c3 = (args.nx, args.ny, N, args.ordering) == (706, 706, 32, "geometric")
    sweep_traffic, sweep_src = measured_traffic("sweep_k32_c3", ("factor.hip",)) if default_c3 else (None, "not the C3 default")
    achieved = sweep_bytes / (sweep_ms * 1e-3) / 1e9
    roofline = {"kernel": "fwd_thin_kernel + v1_assemble_kernel + fwd_level_kernel + bwd_thin_kernel + bwd_level_kernel (one sweep of the factor, "
                          "all tree levels)",
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": sweep_traffic, "traffic_source": sweep_src,
                "bytes_per_launch": sweep_bytes, "us_per_launch": round(sweep_ms * 1e3, 1), "columns": N,
                "nnzL": fstats["nnzL"],
                "compute": (measured_mfma("sweep_k32_c3", ("factor.hip",), 4.0 * fstats["nnzL"] * N, sweep_ms * 1e3)
                            if default_c3 else None)}
    # SpMV, the bit-exact CSR-stream kernel: K and G (same sparsity, 235 MB of traffic each) are applied alternately
    # so that consecutive launches cannot be served from the 256 MiB Infinity Cache
    x = ctx.from_host(rng.normal(size=n))
    y = ctx.empty(n, 1)
    y2 = ctx.empty(n, 1)
    for _ in range(5):
        dK.apply(x, y)
        dG.apply(x, y2)
    ctx.sync()
    ctx.timer_start()
    for _ in range(args.spmv_reps // 2):
        dK.apply(x, y)
        dG.apply(x, y2)
    spmv_ms = ctx.timer_stop_ms() / (2 * (args.spmv_reps // 2))
    spmv_bytes = dK.spmv_bytes(1)
    spmv_rate = spmv_bytes / (spmv_ms * 1e-3) / 1e9
    spmv_traffic, spmv_src = (measured_traffic("spmv_c3", ("sparse.hip",)) if (args.nx, args.ny) == (706, 706)
                              else (None, "not the C3 default"))
    spmv = {"kernel": "spmv_stream_kernel", "bound": "hbm", "achieved": round(spmv_rate, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(spmv_rate / HBM_PEAK_GBS, 4),
            "traffic": spmv_traffic, "traffic_source": spmv_src,
            "bytes_per_launch": spmv_bytes, "us_per_launch": round(spmv_ms * 1e3, 2)}

    # SpMM at the width of a lock-step Krylov step (N columns, the tiled kernel), K and G alternating as above
    Ys, Ys2 = ctx.empty(n, N), ctx.empty(n, N)
    for _ in range(3):
        dK.apply(Xs, Ys)
        dG.apply(Xs, Ys2)
    ctx.sync()
    ctx.timer_start()
    for _ in range(10):
        dK.apply(Xs, Ys)
        dG.apply(Xs, Ys2)
    spmm_ms = ctx.timer_stop_ms() / 20
    spmm_bytes = dK.spmv_bytes(N)
    spmm_rate = spmm_bytes / (spmm_ms * 1e-3) / 1e9
    spmm_traffic, spmm_src = (measured_traffic("spmm_k32_c3", ("sparse.hip",)) if default_c3 else (None, "not the C3 default"))
    spmm = {"kernel": "spmm_tiled_kernel", "bound": "hbm", "achieved": round(spmm_rate, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(spmm_rate / HBM_PEAK_GBS, 4), "traffic": spmm_traffic,
            "traffic_source": spmm_src, "bytes_per_launch": spmm_bytes, "us_per_launch": round(spmm_ms * 1e3, 2),
            "columns": N}
    del Ys, Ys2

    # ------------------------------------------------------------------ the same design point prepared on the device
    # (untimed preamble): K values, stress stiffness G(u), K + sigma G and the numeric refactorisation
    # without host arrays; compared with the host-assembled matrices.
    from eigd_amd.device import ElementAssembler, ElementLinearMatrices

    t0 = time.perf_counter()
    asm = ElementAssembler(ctx, col.elem_dofs, n)
    full_dofs, Lt, Qt = col.stress_stiffness_tables()
    elin = ElementLinearMatrices(ctx, full_dofs, Lt, Qt)
    timing["device_assembly_analysis_s"] = time.perf_counter() - t0
    sK = col.rhoE**col.p + col.rho0_K
    sG = col.rhoE**col.p + col.rho0_G
    u_dev = ctx.from_host(u)
    for rep in range(2):  # second pass: steady state (element data already resident)
        ctx.sync()
        t0 = time.perf_counter()
        vK = asm.assemble(col.Ke0, sK)
        vG = asm.assemble(elin(u_dev), sG)
        vS = ctx.empty(vK.n, 1).assign_lincomb([(1.0, vK), (float(sigma), vG)])
        factor.refactor_device(vS)
        ctx.sync()
        timing["device_assemble_and_refactor_s"] = time.perf_counter() - t0
    ref_vals = (K + sigma * G).tocsr()
    ref_vals.sort_indices()
    dev_vals = asm.values_to_host(vS)
    accuracy["device_assembly_rel_err"] = float(np.abs(dev_vals - ref_vals.data).max() / np.abs(ref_vals.data).max())
    log(rank, f"device assembly + refactorisation: {timing['device_assemble_and_refactor_s'] * 1e3:.1f} ms "
              f"(analysis {timing['device_assembly_analysis_s']:.2f} s once); values vs host assembly "
              f"{accuracy['device_assembly_rel_err']:.1e}")

    # ------------------------------------------------------------------ CPU baseline (oracle = port of the reference)
    cpu = None
    if args.cpu_sample != "none" and world == 1:
        cpu = cpu_baseline(args, K, G, sigma, lam, Phi, Phib, lamb, col, solver, dPhib, dpsi, data, dAdx, dBdx, ndv, log)
        accuracy["dfdx_rel_err_gpu_vs_cpu"] = cpu.pop("dfdx_rel_err_gpu_vs_cpu")
        accuracy["psi_rel_err_gpu_vs_cpu"] = cpu["psi_rel_err_gpu_vs_cpu"]

    out = {
        "metric": "adjoint_mode_derivatives_per_sec",
        "value": round(value, 3),
        "unit": "modes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"buckling {n / 1e6:.1f}M-dof Q4 column ({args.nx}x{args.ny} elements), {N} modes, "
                               f"IRAM m={args.m} + sibk rtol={args.rtol:g} + tensor total derivative w.r.t. element densities",
                   "n_dof": int(n), "nnz": int(K.nnz), "modes": N, "m": args.m, "sigma": round(float(sigma), 6),
                   "ordering": args.ordering,
                   "parallelism": f"modes sharded over {world} GPU(s), one RCCL all-reduce of df/dx per step"},
        "roofline": roofline,
        "spmv": spmv,
        "spmm": spmm,
        "cpu_baseline": cpu,
        "numpy_api": numpy_api,
        "accuracy": accuracy,
        "preamble_s": {k: round(v, 3) for k, v in timing.items()},
        "factor_sweeps_per_step": int(adj_count),
        "sibk_iterations": sibk_iterations,
        "arnoldi_form": arnoldi_form,
        "scaling_model": scaling_model,
        "scaling_model_c5": None,
        "extras0": extras0,
        "eigensolve_sweeps": int(eig_count),
        "eigensolver": eig_info,
        "lock_step": {"recurrence": last_round.get("recurrence"),
                      "cg": {key[3:]: last_round.get(key) for key in ("cg_steps", "cg_projections", "cg_projection_updates",
                                                                      "cg_sweeps_for_nothing", "cg_waited_for",
                                                                      "cg_restarted_modes", "cg_solution")},
                      "steps_per_gram_schmidt_pass": last_round.get("steps_per_pass"),
                      "inner_projections": last_round.get("inner_projections"),
                      "cycles": last_round.get("gs_cycles"),
                      "correcting_gram_schmidt_passes": last_round.get("gs_correcting_passes"),
                      "post_gs_projections_measured": last_round.get("post_gs_projections"),
                      "post_gs_updates_applied": last_round.get("post_gs_updates_applied"),
                      "cycles_enqueued_for_nothing": last_round.get("cycles_enqueued_for_nothing"),
                      "cycles_waited_for": last_round.get("cycles_waited_for")},
        # one design point of an optimisation loop as the reference's harness runs it (buckling.py:548-632, 874-986):
        # assembly + factorisation (device: K, G(u), K + sigma G, numeric factor) + eigensolve + the timed step
        # (a design point after the first: device assembly + refactorisation and the eigensolve as repeated)
        "design_point_s": round(timing["device_assemble_and_refactor_s"] + timing["eigensolve_repeat_s"] + ms_per_step * 1e-3, 4),
        "first_design_point_s": round(timing["device_assemble_and_refactor_s"] + timing["eigensolve_s"] + ms_per_step * 1e-3, 4),
    }
    out["design_point_modes_per_s"] = round(N / out["design_point_s"], 3)
    if extras0 is not None:
        extras0["design_point_s"] = round(timing["device_assemble_and_refactor_s"] + extras0["eigensolve_s"]
                                          + extras0["ms_per_step"] * 1e-3, 4)
        extras0["design_point_modes_per_s"] = round(N / extras0["design_point_s"], 3)
    if world == 1 and comm is None and default_c3 and not args.no_scaling_model and not args.no_c5_scaling_model:
        # config C5 (64 modes: 8 per rank at 8 GPUs) through the same model; the C3 objects go first
        try:
            solver = factor = dK = dG = dPhib = dpsi = Xs = None   # (the closures above hold these names, not the objects)
            ctx.release_workspaces()
            out["scaling_model_c5"] = c5_scaling_model(args)
            log(rank, f"scaling model, config C5: {out['scaling_model_c5']['ranks']}")
        except Exception as exc:                       # (a model for the record: it must not take the bench line with it)
            out["scaling_model_c5"] = {"error": repr(exc)}
    print(json.dumps(out), flush=True)


def cpu_baseline(args, K, G, sigma, lam, Phi, Phib, lamb, col, solver, dPhib, dpsi, data, dAdx, dBdx, ndv, log):
    """
    The CPU oracle (numpy/scipy restatement of the reference: SuperLU + laa guess + sibk, oracle/eigd_oracle.py) on the
    same matrices, eigenpairs, Lanczos data and right-hand sides, for a bounded sample of the modes spread over the
    spectrum (at least two, more while the time budget lasts); then those modes' share of df/dx on both sides.  One
    repetition: a single mode costs 16 ... 65 s of SuperLU solves at 1 M dof (SURVEY 8d asks for best of 3 on 4 modes,
    ~8 minutes here; the default bench run has to finish in a few).
    """
    from oracle import eigd_oracle as orc

    n, N = Phib.shape
    threads = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    t0 = time.perf_counter()
    fac = orc.SpLuOperator((K + sigma * G).tocsc())
    t_fac = time.perf_counter() - t0
    log(0, f"cpu: SuperLU factorisation {t_fac:.1f}s")
    ns = max(1, min(args.cpu_modes, N))
    want = sorted({int(round(q)) for q in np.linspace(0, N - 1, ns)})
    forced = args.cpu_mode_list is not None
    if forced:
        want = sorted({int(q) for q in args.cpu_mode_list.split(",") if 0 <= int(q) < N})
    V, Y, theta, indices = solver.V, np.asarray(solver.Y), np.asarray(solver.theta), np.asarray(solver.indices)
    t0 = time.perf_counter()
    psi0 = orc.laa(Phib, K, fac, sigma, lam, V, Y, theta, indices, b_ortho=True, mode="buckling", cols=want)
    t_laa_all = time.perf_counter() - t0
    # one mode at a time (the reference's own loop order), lowest first, until the time budget is used up: the high modes
    # need ~4x the iterations of the low ones (9 ... 36 at C3) at ~1.8 s per iteration of the 1M-dof SuperLU solve.
    # Every call starts from the laa guess and returns with the correction along the eigenvectors applied: its column.
    sample, iters, t_sibk = [], [], 0.0
    psi_c = np.zeros_like(psi0)
    data_c = {}
    for i in want:
        t0 = time.perf_counter()
        p_i, data_c, info = orc.sibk(Phib, G, K, lam, Phi, mode="buckling", psi=psi0.copy(), sigma=sigma, factor=fac,
                                     rtol=args.rtol, modes=[i])
        psi_c[:, i] = p_i[:, i]
        t_sibk += time.perf_counter() - t0
        sample.append(i)
        iters += list(info)
        log(0, f"cpu: mode {i}: {info} iterations, {time.perf_counter() - t0:.1f}s")
        if not forced and len(sample) >= 2 and t_sibk > args.cpu_budget_s:
            break
    t_laa = t_laa_all * len(sample) / len(want)
    # total derivative of the sampled modes: the reference's weight vectors (eigenvector_derivatives.py:118-134),
    # numpy einsum version of the two element callbacks (examples/buckling.py:178-218, 321-340)
    t0 = time.perf_counter()
    WA, WB = orc.derivative_weights(lam, Phi, lamb, Phib, psi_c, data_c, "buckling")
    ed = col.elem_dofs

    def gather(M):
        return np.where(ed[:, :, None] >= 0, M[np.maximum(ed, 0)], 0.0)

    wAe, wBe, pe = gather(WA[:, sample]), gather(WB[:, sample]), gather(Phi[:, sample])
    dfdx_ck = (col.dG_scale()[:, None] * np.einsum("nak,nab,nbk->nk", wAe, col.Ge_unit, pe)
               + col.dK_scale()[:, None] * np.einsum("nak,ab,nbk->nk", wBe, col.Ke0, pe))   # per element and sampled mode
    dfdx_c = dfdx_ck.sum(axis=1)
    t_der = time.perf_counter() - t0
    best = (t_laa + t_sibk + t_der, t_laa, t_sibk, t_der)
    log(0, f"cpu: {len(sample)} modes: laa {t_laa:.1f}s sibk {t_sibk:.1f}s derivative {t_der:.2f}s (iterations {iters})")
    # the same modes' share on the GPU: psi of the timed step, device callbacks, restricted to the sample
    import eigd_amd.adjoint as adj

    dfdx_g = adj._total_derivative_device(solver._prob.Phi, dPhib, dpsi, lam, lamb, dAdx, dBdx, np.zeros(ndv), data,
                                          "buckling", "tensor", np.asarray(sample))
    err_df = float(np.linalg.norm(dfdx_g - dfdx_c) / np.linalg.norm(dfdx_c))
    psi_g = dpsi.get()[:, sample]
    err_psi_k = np.linalg.norm(psi_g - psi_c[:, sample], axis=0) / np.linalg.norm(psi_c[:, sample], axis=0)
    err_psi = float(np.max(err_psi_k))
    per_mode = {}
    for q, i in enumerate(sample):                       # every sampled mode's own share of df/dx on both sides
        dg_i = adj._total_derivative_device(solver._prob.Phi, dPhib, dpsi, lam, lamb, dAdx, dBdx, np.zeros(ndv), data,
                                            "buckling", "tensor", np.asarray([i]))
        it_gpu = getattr(args, "timed_sibk_iterations", None)   # (of the timed step: later legs ran other solves)
        per_mode[str(i)] = {"sibk_iterations_cpu": int(iters[q]), "sibk_iterations_gpu": int(it_gpu[i]) if it_gpu else None,
                            "psi_rel_err": float(err_psi_k[q]),
                            "dfdx_rel_err": float(np.linalg.norm(dg_i - dfdx_ck[:, q]) / np.linalg.norm(dfdx_ck[:, q]))}
    log(0, f"cpu: GPU-vs-CPU on modes {sample}: psi rel-err {err_psi:.2e}, df/dx rel-err {err_df:.2e}; per mode {per_mode}")
    tot, t_laa, t_sibk, t_der = best
    return {"value": round(len(sample) / tot, 5), "unit": "modes/s", "cores": 1,
            "cores_note": "1 = threads that do work: SuperLU's solve and scipy's CSR products are sequential kernels (the "
                          "BLAS threads of numpy only see the small Hessenberg problems)", "host_cpus": os.cpu_count(),
            "thread_env": threads, "kind": "port",
            "sample": f"modes {sample} of {N} (candidates {want}, budget {args.cpu_budget_s:.0f}s) on the same 1M-dof matrices, "
                      f"eigenpairs, Lanczos basis and right-hand sides: oracle laa guess {t_laa:.1f}s + sibk {t_sibk:.1f}s "
                      f"({iters} iterations) + derivative {t_der:.2f}s, one repetition; SuperLU factor {t_fac:.0f}s untimed, "
                      f"like the GPU's; SuperLU / scipy CSR kernels are sequential whatever the thread settings",
            "factor_s": round(t_fac, 1), "psi_rel_err_gpu_vs_cpu": err_psi, "dfdx_rel_err_gpu_vs_cpu": err_df,
            "per_mode_gpu_vs_cpu": per_mode}


if __name__ == "__main__":
    main()
