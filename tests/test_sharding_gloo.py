"""
Mode sharding over torch.distributed with the gloo backend, world_size 2, on the CPU: the
collective plumbing (TorchDistComm) and the partition / all-reduce identities the sharded
solve_adjoint / add_total_derivative rely on.  The n-vector arithmetic itself needs the GPU
(tests/test_gpu_path.py::test_mode_sharding_* and test_two_rank_gpu_sharding).
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["EIGD_ROOT"])
    import torch.distributed as dist
    dist.init_process_group("gloo")
    from eigd_amd.comm import TorchDistComm, mode_columns
    from eigd_amd import adjoint as adj
    comm = TorchDistComm()
    rank, size = comm.rank, comm.size
    rng = np.random.default_rng(5)           # same stream on every rank: replicated inputs
    n, N, ndv = 200, 7, 11
    lam = np.sort(rng.uniform(1.0, 5.0, size=N)); lam[3] = lam[2] + 2e-6
    Phi, Phib, psi = rng.normal(size=(n, N)), rng.normal(size=(n, N)), rng.normal(size=(n, N))
    lamb = rng.normal(size=N)
    Ca, Cb = rng.normal(size=(n, ndv)), rng.normal(size=(n, ndv))
    G = -Phi.T @ Phib
    Cc, data = adj.correction_coefficients(lam, G, 1e-5, "normal")
    beta = 0.5 * np.einsum("ij,ij->j", Phi, Phib)
    CA, CB, sa, sb = adj.derivative_weight_coefficients(lam, lamb, beta, data, "normal", N)
    cols = mode_columns(N, rank, size)
    # each rank corrects and differentiates only its own modes ...
    psi_c = psi[:, cols] + Phi @ Cc[:, cols]
    WA = Phi @ CA[:, cols] + psi_c * sa[cols]
    WB = Phi @ CB[:, cols] + psi_c * sb[cols]
    part = Ca.T @ np.sum(WA * Phi[:, cols], axis=1) - Cb.T @ np.sum(WB * Phi[:, cols], axis=1)
    total = comm.allreduce_sum(part)          # ... and ONE all-reduce assembles df/dx
    # unsharded reference on every rank
    psi_f = psi + Phi @ Cc
    WA_f, WB_f = Phi @ CA + psi_f * sa, Phi @ CB + psi_f * sb
    full = Ca.T @ np.sum(WA_f * Phi, axis=1) - Cb.T @ np.sum(WB_f * Phi, axis=1)
    assert np.allclose(total, full, rtol=1e-12, atol=1e-12), (total, full)
    # G assembled from per-rank columns (pgmres / pcpg path)
    Gloc = np.zeros((N, N)); Gloc[:, cols] = G[:, cols]
    assert np.allclose(comm.allreduce_sum(Gloc), G)
    assert comm.allreduce_max(float(rank)) == size - 1
    comm.barrier()
    assert sorted(np.concatenate([mode_columns(N, r, size) for r in range(size)]).tolist()) == list(range(N))
    dist.destroy_process_group()
    print("rank", rank, "ok")
    """
)


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, EIGD_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_mode_columns_partition():
    from eigd_amd.comm import SerialComm, mode_columns

    for N in (1, 6, 32, 33):
        for P in (1, 2, 4, 8):
            allc = np.concatenate([mode_columns(N, r, P) for r in range(P)])
            assert sorted(allc.tolist()) == list(range(N))
            sizes = [len(mode_columns(N, r, P)) for r in range(P)]
            assert max(sizes) - min(sizes) <= 1
    c = SerialComm()
    a = np.arange(3.0)
    assert c.allreduce_sum(a) is a and c.size == 1
