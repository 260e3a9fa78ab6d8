"""
Solver switches of the host layer.  Plain module attributes: the defaults are the measured best (docs/LOG.md has the
experiments); tests and probes change them with ``monkeypatch.setattr(eigd_amd.tuning, name, value)``.  None of them
changes what is computed beyond rounding -- they select between equivalent forms of the same algorithm.

Environment variables the package reads (all optional): EIGD_DEVICE (device of the default context, else LOCAL_RANK),
EIGD_STREAMS (concurrent mode groups of the lock-step solvers, default 1), EIGD_COMM_DIR / EIGD_COMM_INIT_TIMEOUT
(rendezvous directory and watchdog of the RCCL communicator), EIGD_TRACE_IRAM (restart log of the block eigensolver);
bench.py adds EIGD_LAUNCH_TIMEOUT.  The library itself reads EIGD_PRE_MIN_WG when a factor is created (fewest row-tile
workgroups of a tree level whose right-hand sides are written once per level, default 512; csrc/factor.hip).
"""

# ---- sibk, lock-step form (adjoint.py)
recurrence = "auto"        # "auto": conjugate gradients in the factor inner product when the shift is positive definite
                           # (short recurrence, no Krylov history), else the Arnoldi form; "arnoldi": always the
                           # reference's form (full Gram-Schmidt against the history, Hessenberg least squares)
cg_solution_from_history = True   # short recurrence: psi = sum_k s_k z_k once at the end from the kept z of every step
                           # (four streaming passes per step); False: the three-term recurrence for psi runs along (eight)
cg_dots_in_spmm = True     # short recurrence: r.z and z.y from the SpMM's own pass (False: a dot kernel over z, r, y)
cg_projection_period = 0   # short recurrence: steps between projections of the residual (0: from the spectrum, at most 4)
cg_projection_tol = 1e-11  # ... applied where a coefficient exceeds this times the residual norm of its column
cg_project_previous = True # ... and the previous residual with it (False: the residual alone -- the deflated components
                           # come back through the three-term recurrence and grow from period to period)
cg_project_extra_pairs = False   # ... against the extra pairs as well (False: the N requested pairs only)
steps_per_pass = 2         # Arnoldi form: Krylov steps per Gram-Schmidt pass (1 = orthogonalise every step)
inner_projections = False  # Arnoldi form: True keeps the projections behind both operator applications (ref 1250-1252)
pair_defect_tol = 1e-10    # Arnoldi form: w_{j+1}.w_{j+2} above which a two-step solve is redone in the one-step form
predict_finish = True      # enqueue the next cycle only for the modes not expected to finish in the current one
reorth_tol = 1e-13         # measured second Gram-Schmidt pass: applied where |h2| > reorth_tol |h1|

# ---- the numpy-in / numpy-out surface (device.py: Context.twin_upload)
host_twins = "returned"    # "returned": the device block a returned psi was downloaded from is handed out again when the same
                           # array comes back (it is returned read-only: an in-place edit raises); caller-owned arrays (Phib)
                           # are transferred every time.  True: caller-owned arrays keep device copies too, validated against
                           # a content sample of 1024 rows + first and last page only (an edit between sampled rows goes
                           # unnoticed: opt-in).  False: every call transfers what it is given, results are writable

# ---- restarted block Lanczos (lanczos.py)
iram_block = 0             # block size (0: 8 for n >= 200 000, 4 for n >= 50 000, else the single-vector solver)
iram_extra = None          # converged pairs beyond N kept for the adjoint stage's deflation (None: min(N, 32) with blocks)
iram_basis = 0             # internal basis size (0: max(m, iram_basis_factor (N + extra) + block))
iram_basis_factor = 2.5    # (2 -> 2.5: C3 -5 %, C4 -7 %, C2 unchanged; beyond 2.5 nothing more)
iram_extra_tol = 1e-11     # convergence tolerance of the extra pairs (relative to |theta|).  1e-10 would save 7 % of the C3
                           # eigensolve with the same psi and residuals there, but takes the eigenvector part of the C5
                           # gradient check from < 1e-5 to 1.3e-5 (tools/extra_tol_probe.py, tests/test_gpu_shell.py)
iram_seed = 12345          # seed of the start block
lanczos_local_first_pass = True   # Gram-Schmidt of a Lanczos step: first pass against the last two blocks only (the
                           # recurrence), then ONE measured pass over the whole basis; False: both passes over all of it
lanczos_fused_restart = True      # V <- V S of a thick restart in one pass per 80 new columns (eigd_panels_times; bases of
                           # up to 192 vectors); False: panel by panel through the general product, 9 launches per 64 columns
deflate_extra = True       # the adjoint stage deflates the extra pairs and adds their share of psi in closed form
laa_internal = True        # the block run's own basis serves the first guess of solve_adjoint
laa_relation = True        # ... formed through the Lanczos relation of that basis (no factor application)
