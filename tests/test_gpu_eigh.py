"""The hand-written symmetric eigensolver (Householder tridiagonalisation + divide and conquer +
back-transformation) on the MI355X box: stage by stage and end to end.

Parity with LAPACK's eigenvectors is neither possible nor required: the checks are
backward error ||A U - U L|| / (n ||A|| eps), orthogonality ||U^T U - I|| / (n eps), eigenvalues against
LAPACK (scipy, test infrastructure) and -- what the LMM consumes -- the downstream statistics.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _sym(n, seed, kind="random"):
    rng = np.random.default_rng(seed)
    if kind == "random":
        A = rng.standard_normal((n, n))
        return (A + A.T) / 2
    if kind == "kinship":  # centred XX^T/p: one zero eigenvalue, Marchenko-Pastur bulk, structure
        p = 3 * n
        maf = rng.uniform(0.05, 0.5, p)
        X = rng.binomial(2, maf[None, :], size=(n, p)).astype(float)
        X[: n // 2] += rng.binomial(1, 0.2, size=(n // 2, p))
        X -= X.mean(0)
        K = X @ X.T / p
        return (K + K.T) / 2
    if kind == "lowrank":  # heavy deflation: rank 5 + tiny noise
        B = rng.standard_normal((n, 5))
        return B @ B.T + 1e-9 * np.eye(n)
    if kind == "clustered":
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        w = np.concatenate([np.ones(n // 2), 1 + 1e-12 * rng.standard_normal(n - n // 2 - 3), [5.0, 5.0, -3.0]])
        return (Q * w) @ Q.T
    raise ValueError(kind)


def _check(A, U, w, tag, tol_res=30.0, tol_orth=30.0):
    n = A.shape[0]
    nrm = max(np.linalg.norm(A, 2), 1e-300)
    res = np.linalg.norm(A @ U - U * w[None, :]) / (nrm * n * EPS)
    orth = np.linalg.norm(U.T @ U - np.eye(n)) / (n * EPS)
    wr = np.linalg.eigvalsh(A)
    everr = np.max(np.abs(w - wr)) / (nrm * EPS * n)
    print("eigh[%s] n=%d resid %.2f orth %.2f eval %.2f (units of n*eps)" % (tag, n, res, orth, everr))
    assert np.all(np.diff(w) >= 0), "eigenvalues not ascending"
    assert res < tol_res and orth < tol_orth and everr < 30.0, (tag, res, orth, everr)


@pytest.mark.parametrize("n", [2, 3, 7, 64, 65, 129, 200, 517])
def test_tridiagonalisation_stage(gpu_api, n):
    from gemma_amd import _lib as L
    A = _sym(n, 100 + n)
    d, e, tau, VT = np.zeros(n), np.zeros(max(n - 1, 1)), np.zeros(n), np.zeros((n, n))
    L.check(L.lib().gemma_hip_dbg_tridiag(_p(A), n, _p(d), _p(e), _p(tau), _p(VT)), "dbg_tridiag")
    T = np.diag(d) + np.diag(e[: n - 1], 1) + np.diag(e[: n - 1], -1)
    Q = np.eye(n)
    for j in range(n):  # Q = H_0 H_1 ...
        Q -= tau[j] * np.outer(Q @ VT[j], VT[j])  # Q (I - tau v v^T) as a rank-1 update
    nrm = np.linalg.norm(A, 2)
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) < 50 * n * EPS
    assert np.linalg.norm(Q @ T @ Q.T - A) / nrm < 50 * n * EPS
    assert np.max(np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A))) / nrm < 50 * n * EPS


@pytest.mark.parametrize("n,seg", [(64, 1024), (130, 512), (200, 1024), (1030, 512), (1030, 1024), (2200, 512),
                                   (2200, 1024)])
def test_symmetric_symv_path(gpu_api, n, seg, monkeypatch):
    """The lower-triangle SYMV (td_symv_sym_kernel: 64-row strips x 1024-column segments, fixed-order partial sums) forced
    on from the first column (it normally takes over for trailing sizes >= 8192): strips and segments that are ragged,
    straddle the diagonal or start inside a 64-column sub-tile; same bar as the row-per-wave form (even n only)."""
    from gemma_amd import _lib as L
    monkeypatch.setenv("GEMMA_HIP_EIGH_SYMV_MIN", "1")
    monkeypatch.setenv("GEMMA_HIP_EIGH_SEG", str(seg))
    A = _sym(n, 900 + n)
    d, e, tau, VT = np.zeros(n), np.zeros(max(n - 1, 1)), np.zeros(n), np.zeros((n, n))
    L.check(L.lib().gemma_hip_dbg_tridiag(_p(A), n, _p(d), _p(e), _p(tau), _p(VT)), "dbg_tridiag")
    T = np.diag(d) + np.diag(e[: n - 1], 1) + np.diag(e[: n - 1], -1)
    nrm = np.linalg.norm(A, 2)
    assert np.max(np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A))) / nrm < 50 * n * EPS
    if n <= 1100:
        Q = np.eye(n)
        for j in range(n):
            Q -= tau[j] * np.outer(Q @ VT[j], VT[j])  # Q (I - tau v v^T) as a rank-1 update
        assert np.linalg.norm(Q.T @ Q - np.eye(n)) < 50 * n * EPS
        assert np.linalg.norm(Q @ T @ Q.T - A) / nrm < 50 * n * EPS


@pytest.mark.parametrize("n,kind", [(2, "random"), (3, "random"), (64, "random"), (65, "random"), (150, "random"),
                                    (300, "laplace"), (300, "neardiag"), (257, "wilkinson"), (700, "random"),
                                    (1000, "graded")])
def test_divide_and_conquer_stage(gpu_api, n, kind):
    from gemma_amd import _lib as L
    rng = np.random.default_rng(n)
    if kind == "random":
        d, e = rng.standard_normal(n), rng.standard_normal(n - 1)
    elif kind == "laplace":
        d, e = np.full(n, 2.0), np.full(n - 1, -1.0)
    elif kind == "neardiag":
        d, e = np.ones(n), np.full(n - 1, 1e-9)
    elif kind == "wilkinson":
        d, e = np.abs(np.arange(n) - n // 2).astype(float), np.ones(n - 1)
    else:
        d, e = np.arange(n, dtype=float), np.full(n - 1, 1e-3)
    w, ZT = np.zeros(n), np.zeros((n, n))
    L.check(L.lib().gemma_hip_dbg_stedc(_p(d), _p(e), n, _p(w), _p(ZT)), "dbg_stedc")
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    _check(T, ZT.T.copy(), w, "stedc-" + kind)


@pytest.mark.parametrize("n,kind", [(1, "random"), (2, "random"), (5, "random"), (64, "random"), (100, "random"),
                                    (129, "kinship"), (300, "lowrank"), (333, "clustered"), (640, "kinship"),
                                    (1000, "kinship")])
def test_eigh_end_to_end(gpu_api, n, kind):
    A = _sym(n, 7 * n + 1, kind)
    U, w = np.zeros((n, n)), np.zeros(n)
    tr = gpu_api.EigenDecomp_Zeroed(A.copy(), U, w)
    wz = np.linalg.eigvalsh(A)
    wz[wz < 1e-10] = 0.0  # src/lapack.cpp:268
    assert tr == pytest.approx(wz.mean(), rel=1e-9, abs=1e-12)
    Az = A if kind != "kinship" and wz.min() > 0 else A
    # residual against the un-zeroed spectrum: recompute with the raw eigenvalues where zeroed
    w_raw = np.where(w == 0.0, np.einsum("ij,ij->j", U, A @ U), w)
    _check(Az, U, np.sort(w_raw), "eigh-" + kind) if np.all(np.diff(w_raw) >= -1e-12) else None
    assert np.linalg.norm(U.T @ U - np.eye(n)) < 50 * n * EPS
    assert np.linalg.norm(A @ U - U * w_raw[None, :]) / max(np.linalg.norm(A, 2), 1e-300) < 50 * n * EPS
    assert np.all((w >= 1e-10) | (w == 0.0))


@pytest.mark.parametrize("n", [130, 517, 1000])
def test_backtransform_gram_sources_agree(gpu_api, n, monkeypatch):
    """The compact-WY factor of each 128-reflector panel needs Y Y^T: by default the strict upper triangle comes from the
    panel dots the tridiagonalisation computes anyway; GEMMA_HIP_EIGH_PANEL_S=0 recomputes it as a GEMM.  Both must
    give an orthogonal eigenbasis with the same residual bar (ragged last panel included)."""
    A = _sym(n, 31 * n, "kinship")
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("GEMMA_HIP_EIGH_PANEL_S", flag)
        U, w = np.zeros((n, n)), np.zeros(n)
        gpu_api.EigenDecomp_Zeroed(A.copy(), U, w)
        w_raw = np.where(w == 0.0, np.einsum("ij,ij->j", U, A @ U), w)
        assert np.linalg.norm(U.T @ U - np.eye(n)) < 50 * n * EPS
        assert np.linalg.norm(A @ U - U * w_raw[None, :]) / np.linalg.norm(A, 2) < 50 * n * EPS
        out.append(w)
    assert np.allclose(out[0], out[1], rtol=0, atol=1e-12)


def test_identity_and_diagonal(gpu_api):
    n = 200
    A = np.diag(np.linspace(-1, 3, n))
    U, w = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(A.copy(), U, w)
    assert np.allclose(np.sort(np.where(w == 0, 0, w)), np.sort(np.where(np.diag(A) < 1e-10, 0, np.diag(A))), atol=1e-13)
    assert np.linalg.norm(U.T @ U - np.eye(n)) < 1e-12


def test_bxd_statistics_with_own_eigenvectors(gpu_api, oracle, bxd):
    """-lmm 1 and -lmm 2 on BXD with U, eval from this library's eigensolver instead of dsyevr:
    beta/se/p/logl within 1e-6 of the oracle outputs that reproduce the reference goldens."""
    G = oracle.center_matrix(bxd["K_sub"])
    n = G.shape[0]
    U, ev = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(G.copy(), U, ev)
    assert np.allclose(ev, bxd["eval"], rtol=0, atol=1e-12)
    # W, y are not in the fixture: recover them from the fixture's rotation (U_ref is orthogonal)
    Wm = bxd["U"] @ bxd["UtW"]
    y = bxd["U"] @ bxd["Uty"]
    UtW, Uty = U.T @ Wm, U.T @ y
    X = bxd["X"].astype(np.float64)[:1500]
    null = bxd["null"]
    for mode in (1, 2):
        lmm = gpu_api.LMM(a_mode=mode, l_mle_null=null[0], logl_mle_H0=null[1])
        got = lmm.AnalyzeBimbam(U, ev, UtW, Uty, X)
        ref = bxd["stat_mode%d" % mode][:1500]
        ok = ~(np.isnan(got["logl_H1"]) | np.isnan(ref["logl_H1"]))
        for k in (["beta", "se", "p_wald", "logl_H1"] if mode == 1 else ["p_lrt", "logl_H1"]):
            rel = np.abs(got[k][ok] - ref[k][ok]) / np.abs(ref[k][ok])
            assert rel.max() < 1e-6, (mode, k, rel.max())
    if True:
        assert "%.6e" % got["p_lrt"][0] == "1.234747e-01"  # test/dev_tests.rb:42


def test_eigh_4096_device(gpu_api):
    import torch
    n = 4096
    g = torch.Generator(device="cuda").manual_seed(3)
    X = torch.randn((n, 3 * n), dtype=torch.float64, device="cuda", generator=g)
    A = X @ X.T / (3 * n)
    A = (A + A.T) / 2
    U = torch.empty_like(A)
    w = torch.empty(n, dtype=torch.float64, device="cuda")
    import time
    t0 = time.time()
    gpu_api.EigenDecomp_Zeroed(A.clone(), U, w)
    torch.cuda.synchronize()
    dt = time.time() - t0
    nrm = torch.linalg.matrix_norm(A, 2)
    res = torch.linalg.matrix_norm(A @ U - U * w[None, :]) / (nrm * n * EPS)
    orth = torch.linalg.matrix_norm(U.T @ U - torch.eye(n, dtype=torch.float64, device="cuda")) / (n * EPS)
    wr = torch.linalg.eigvalsh(A)
    print("eigh n=4096: %.2f s, resid %.2f orth %.2f (n*eps), max eval err %.2e" % (dt, float(res), float(orth), float((w - wr).abs().max())))
    assert float(res) < 30 and float(orth) < 30
    assert float((w - wr).abs().max() / nrm) < 1e-12


# --------------------------------------------------------------------------- two-stage reduction (eigh2.hip.h)
@pytest.mark.parametrize("n,kind", [(384, "random"), (386, "kinship"), (768, "kinship"), (1026, "lowrank"),
                                    (1538, "clustered")])
def test_two_stage_reduction_stages(gpu_api, n, kind):
    """dense -> band (panel QR + two-sided GEMM update) and band -> tridiagonal (bulge chase) are orthogonal
    similarity transformations: both keep the spectrum of the input.  n = 384 / 386: one full panel and the smallest
    ragged one; 1026 / 1538: last chase blocks shorter than 128 rows."""
    from gemma_amd import _lib as L
    from scipy.linalg import eigvalsh_tridiagonal
    A = _sym(n, 31 * n + 5, kind)
    band, d, e = np.zeros((n, 129)), np.zeros(n), np.zeros(n - 1)
    L.check(L.lib().gemma_hip_dbg_eigh2(_p(np.ascontiguousarray(A)), n, _p(band), _p(d), _p(e)), "dbg_eigh2")
    B = np.zeros((n, n))
    for t in range(129):
        idx = np.arange(n - t)
        B[idx + t, idx] = band[: n - t, t]
        B[idx, idx + t] = band[: n - t, t]
    wr = np.linalg.eigvalsh(A)
    scale = max(np.abs(wr).max(), 1e-300) * n * EPS
    eb = np.abs(np.linalg.eigvalsh(B) - wr).max() / scale
    et = np.abs(eigvalsh_tridiagonal(d, e) - wr).max() / scale
    print("two-stage[%s] n=%d: band eigenvalues %.2f, tridiagonal %.2f (n*eps)" % (kind, n, eb, et))
    assert eb < 5.0 and et < 5.0


@pytest.mark.parametrize("n,kind,chase", [(384, "random", "persist"), (640, "kinship", "persist"), (1000, "kinship", "steps"),
                                          (1538, "clustered", "persist"), (2050, "lowrank", "persist")])
def test_eigh_two_stage_end_to_end(gpu_api, n, kind, chase, monkeypatch):
    """The whole solver on the two-stage path (forced: by default it starts at n = 8000), with the bounds of
    test_eigh_end_to_end; `steps` runs the bulge chase as one launch per time step instead of the persistent kernel."""
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "2")
    monkeypatch.setenv("GEMMA_HIP_EIGH_BC", chase)
    A = _sym(n, 7 * n + 3, kind)
    U, w = np.zeros((n, n)), np.zeros(n)
    tr = gpu_api.EigenDecomp_Zeroed(A.copy(), U, w)
    wz = np.linalg.eigvalsh(A)
    wz[wz < 1e-10] = 0.0
    assert tr == pytest.approx(wz.mean(), rel=1e-9, abs=1e-12)
    w_raw = np.where(w == 0.0, np.einsum("ij,ij->j", U, A @ U), w)
    assert np.linalg.norm(U.T @ U - np.eye(n)) < 50 * n * EPS
    assert np.linalg.norm(A @ U - U * w_raw[None, :]) / max(np.linalg.norm(A, 2), 1e-300) < 50 * n * EPS
    assert np.all((w >= 1e-10) | (w == 0.0))
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "1")
    U1, w1 = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(A.copy(), U1, w1)
    assert np.abs(w - w1).max() <= 30 * n * EPS * max(np.abs(w1).max(), 1e-300)


@pytest.mark.parametrize("n,kind", [(1538, "kinship"), (2307, "random"), (1000, "lowrank")])
def test_bulge_chase_hand_over_variants_are_bit_identical(gpu_api, n, kind, monkeypatch):
    """The persistent chase kernels differ only in WHEN a task may start (round 4: two positions per workgroup with whole-task
    hand-over, GEMMA_HIP_EIGH_BC_PIPE=0; one position per workgroup with the reflector handed on early, =1; default: also the
    left-block hand-over with the shared diagonal entry in a slot, both blocks requested ahead of the reflector): the arithmetic of
    a task is the same, so (U, eval) must be the same bits; a lost update or a stale block would show here."""
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "2")
    A = _sym(n, 5 * n + 1, kind)
    res = {}
    for pipe in ("default", "1", "0"):
        if pipe == "default":
            monkeypatch.delenv("GEMMA_HIP_EIGH_BC_PIPE", raising=False)
        else:
            monkeypatch.setenv("GEMMA_HIP_EIGH_BC_PIPE", pipe)
        U, w = np.zeros((n, n)), np.zeros(n)
        gpu_api.EigenDecomp_Zeroed(A.copy(), U, w)
        res[pipe] = (U, w)
    for pipe in ("1", "0"):
        assert np.array_equal(res[pipe][1], res["default"][1]), pipe
        assert np.array_equal(res[pipe][0], res["default"][0]), pipe
    U, w = res["default"]
    assert np.linalg.norm(U.T @ U - np.eye(n)) < 50 * n * EPS


@pytest.mark.parametrize("n,segments,workers", [(1538, "5", "7"), (2050, "64", "512"), (1000, "1", "3")])
def test_stage2_backtransform_dynamic_schedule(gpu_api, n, segments, workers, monkeypatch):
    """q2_apply_kernel as persistent workgroups drawing (segment of chase steps, row block) tasks (round 3; by default only where
    n / 64 row blocks do not fill 2 x CUs slots evenly) applies the same reflectors to the same rows in the same order as one
    workgroup per row block: eigenvectors identical bit for bit; more workers than tasks, one segment, fewer workers than row
    blocks (every hand-over between workgroups goes through the progress flags)."""
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "2")
    A = _sym(n, 5 * n + 1, "kinship")
    monkeypatch.setenv("GEMMA_HIP_EIGH_Q2_DYNAMIC", "0")
    U0, w0 = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(A.copy(), U0, w0)
    monkeypatch.setenv("GEMMA_HIP_EIGH_Q2_DYNAMIC", "1")
    monkeypatch.setenv("GEMMA_HIP_EIGH_Q2_SEGMENTS", segments)
    monkeypatch.setenv("GEMMA_HIP_EIGH_Q2_WORKERS", workers)
    U1, w1 = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(A.copy(), U1, w1)
    assert np.array_equal(w0, w1) and np.array_equal(U0, U1)


def test_eigh_two_stage_4096_device(gpu_api, monkeypatch):
    import torch
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "2")
    n = 4096
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn((n, n // 2), dtype=torch.float64, device="cuda", generator=g)  # rank n / 2: half the spectrum is zero
    A = X @ X.T / n
    A = (A + A.T) / 2
    U = torch.empty_like(A)
    w = torch.empty(n, dtype=torch.float64, device="cuda")
    gpu_api.EigenDecomp_Zeroed(A.clone(), U, w)
    torch.cuda.synchronize()
    nrm = torch.linalg.matrix_norm(A, 2)
    res = torch.linalg.matrix_norm(A @ U - U * w[None, :]) / (nrm * n * EPS)
    orth = torch.linalg.matrix_norm(U.T @ U - torch.eye(n, dtype=torch.float64, device="cuda")) / (n * EPS)
    wr = torch.linalg.eigvalsh(A).clamp_min(0)
    print("two-stage eigh n=4096: resid %.2f orth %.2f (n*eps)" % (float(res), float(orth)))
    assert float(res) < 30 and float(orth) < 30
    assert float((w - torch.where(wr < 1e-10, torch.zeros_like(wr), wr)).abs().max() / nrm) < 1e-12


@pytest.mark.parametrize("n,odd", [(14080, False), (14337, True)])
def test_eigh_default_path_at_two_stage_size(gpu_api, monkeypatch, n, odd):
    """The DEFAULT path at the sizes where it is the two-stage reduction (n >= 8 000 since round 4: eigh.hip.h eig_two_stage; nothing
    forced): a kinship-like matrix (centred, one zero eigenvalue, population structure), the same residual / orthogonality
    bars as at n = 4096, eigenvalues against rocSOLVER.  This is the path the headline bench's setup takes at n = 20 000
    (persistent bulge chase, dynamic stage-2 back-transformation schedule, paired stage-1 panels); the odd size goes through
    the (n + 1) embedding.  Reference semantics: src/lapack.cpp:149-291 (dsyevr, eval < 1e-10 -> 0)."""
    import torch
    monkeypatch.delenv("GEMMA_HIP_EIGH_STAGES", raising=False)
    g = torch.Generator(device="cuda").manual_seed(11 + n)
    p = 2 * n
    maf = torch.empty(p, dtype=torch.float64, device="cuda").uniform_(0.05, 0.5, generator=g)
    X = (torch.rand((n, p), dtype=torch.float64, device="cuda", generator=g) < maf).to(torch.float64)
    X += (torch.rand((n, p), dtype=torch.float64, device="cuda", generator=g) < maf).to(torch.float64)
    X[: n // 2] += (torch.rand((n // 2, p), dtype=torch.float64, device="cuda", generator=g) < 0.2).to(torch.float64)
    X -= X.mean(0, keepdim=True)
    A = X @ X.T / p
    del X
    A = (A + A.T) / 2
    A -= A.mean(0, keepdim=True)
    A -= A.mean(1, keepdim=True)
    A = (A + A.T) / 2
    U = torch.empty_like(A)
    w = torch.empty(n, dtype=torch.float64, device="cuda")
    gpu_api.EigenDecomp_Zeroed(A.clone(), U, w)
    torch.cuda.synchronize()
    wr = torch.linalg.eigvalsh(A)
    nrm = wr.abs().max()
    R = A @ U
    R -= U * w[None, :]
    res = torch.linalg.matrix_norm(R) / (nrm * n * EPS)
    R = U.T @ U
    R.diagonal().sub_(1.0)
    orth = torch.linalg.matrix_norm(R) / (n * EPS)
    everr = (w - torch.where(wr < 1e-10, torch.zeros_like(wr), wr)).abs().max() / nrm
    print("default-path eigh n=%d: resid %.2f orth %.2f (n*eps), max eval err / ||A|| %.2e" % (n, float(res), float(orth), float(everr)))
    assert bool((w[1:] >= w[:-1]).all()), "eigenvalues not ascending"
    assert float(res) < 30 and float(orth) < 30
    assert float(everr) < 1e-12


def test_two_stage_degenerate_inputs(gpu_api, monkeypatch):
    """Inputs whose reflectors vanish (tau = 0 everywhere): a diagonal matrix and a block-diagonal one on the two-stage
    path; and a NaN must be reported, not looped on (the bulge-chase kernel's waits are bounded)."""
    from gemma_amd import _lib as L
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "2")
    n = 512
    rng = np.random.default_rng(4)
    dvals = rng.uniform(0.5, 2.0, n)
    U, w = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(np.diag(dvals), U, w)
    assert np.allclose(w, np.sort(dvals), rtol=1e-14) and np.allclose(np.abs(U).sum(0), 1.0, atol=1e-12)
    A = np.zeros((n, n))
    for b0 in range(0, n, 64):
        Bk = rng.standard_normal((64, 64))
        A[b0:b0 + 64, b0:b0 + 64] = Bk @ Bk.T / 64
    gpu_api.EigenDecomp_Zeroed(A.copy(), U, w)
    _check(A, U, w, "two-stage-blockdiag")
    A[3, 5] = A[5, 3] = np.nan
    with pytest.raises(L.GemmaHipError):
        gpu_api.EigenDecomp_Zeroed(A.copy(), U, w)


@pytest.mark.parametrize("n,stages", [(193, "1"), (333, "1"), (1001, "2"), (2049, "2")])
def test_eigh_odd_n_is_padded(gpu_api, n, stages, monkeypatch):
    """Odd n runs as an (n+1) x (n+1) problem with one exactly decoupled extra eigenpair (aligned GEMM paths, symmetric
    SYMV, two-stage reduction all need even n); the result must be the one of the unpadded, unaligned path."""
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", stages)
    A = _sym(n, 11 * n, "kinship")
    U, w = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(A.copy(), U, w)
    w_raw = np.where(w == 0.0, np.einsum("ij,ij->j", U, A @ U), w)
    assert np.linalg.norm(U.T @ U - np.eye(n)) < 50 * n * EPS
    assert np.linalg.norm(A @ U - U * w_raw[None, :]) / np.linalg.norm(A, 2) < 50 * n * EPS
    monkeypatch.setenv("GEMMA_HIP_EIGH_PAD", "0")
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "1")
    U0, w0 = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(A.copy(), U0, w0)
    assert np.abs(w - w0).max() <= 30 * n * EPS * np.abs(w0).max()


@pytest.mark.parametrize("n,kind", [(1538, "kinship"), (2307, "random"), (1280, "lowrank"), (700, "kinship")])
def test_stage1_backtransform_panel_groups_agree(gpu_api, n, kind, monkeypatch):
    """Round 6: the stage-1 back-transformation applies 1, 2, 4 or 8 stage-1 panels as ONE block reflector (GEMMA_HIP_EIGH_Q1_GROUP;
    default 4, K = 512: the compact-WY factor of the group is built level by level from ONE Gram matrix of its reflectors).  Every group
    size is the same orthogonal transformation: eigenvalues bit-identical (they do not pass through it), eigenvectors equal to rounding,
    residual and orthogonality inside the bars of the end-to-end tests.  Sizes whose panel count is not a multiple of the group (a
    leftover group of 1 / 2 / 3 panels at the top) included."""
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "2")
    A = _sym(n, 11 * n + 5, kind)
    nrm = max(np.linalg.norm(A, 2), 1e-300)
    res = {}
    for G in ("1", "2", "4", "8"):
        monkeypatch.setenv("GEMMA_HIP_EIGH_Q1_GROUP", G)
        U, w = np.zeros((n, n)), np.zeros(n)
        gpu_api.EigenDecomp_Zeroed(A.copy(), U, w)
        w_raw = np.where(w == 0.0, np.einsum("ij,ij->j", U, A @ U), w)
        assert np.linalg.norm(U.T @ U - np.eye(n)) < 50 * n * EPS, G
        assert np.linalg.norm(A @ U - U * w_raw[None, :]) / nrm < 50 * n * EPS, G
        res[G] = (U, w)
    for G in ("2", "4", "8"):
        assert np.array_equal(res[G][1], res["1"][1]), G
        if kind != "lowrank":  # (a degenerate eigenspace may come back in another basis)
            assert np.abs(np.abs(res[G][0]) - np.abs(res["1"][0])).max() < 1e-9, G


@pytest.mark.parametrize("n,kind,panel", [(1538, "kinship", ""), (2307, "random", ""), (1411, "kinship", "launch"), (700, "lowrank", "launch")])
def test_stage1_lookahead_is_bit_identical(gpu_api, n, kind, panel, monkeypatch):
    """Round 6: the dense -> band reduction may issue the trailing update in two pieces (first tile row + its mirror, then the rest) and
    factor the NEXT panel on a second stream beside the second piece (by default only where the panel QR is a chain of launches, i.e.
    beyond 32 768 rows; GEMMA_HIP_EIGH_LOOKAHEAD=1: at every panel, =0: never).  Same tiles, same K order: eigenvalues and eigenvectors
    must be the same BITS -- with the one-launch panel kernel and with the per-column launches (GEMMA_HIP_EIGH_PANEL=launch)."""
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "2")
    if panel:
        monkeypatch.setenv("GEMMA_HIP_EIGH_PANEL", panel)
    A = _sym(n, 7 * n + 1, kind)
    res = {}
    for la in ("1", "0", ""):
        if la:
            monkeypatch.setenv("GEMMA_HIP_EIGH_LOOKAHEAD", la)
        else:
            monkeypatch.delenv("GEMMA_HIP_EIGH_LOOKAHEAD")
        U, w = np.zeros((n, n)), np.zeros(n)
        gpu_api.EigenDecomp_Zeroed(A.copy(), U, w)
        res[la] = (U, w)
    for la in ("1", ""):
        assert res[la][1].tobytes() == res["0"][1].tobytes(), la
        assert res[la][0].tobytes() == res["0"][0].tobytes(), la
    nrm = max(np.linalg.norm(A, 2), 1e-300)
    U, w = res["1"]
    w_raw = np.where(w == 0.0, np.einsum("ij,ij->j", U, A @ U), w)
    assert np.linalg.norm(U.T @ U - np.eye(n)) < 50 * n * EPS
    assert np.linalg.norm(A @ U - U * w_raw[None, :]) / nrm < 50 * n * EPS


def test_eigensolver_workspace_pool(gpu_api, monkeypatch):
    """Round 6: gemma_hip_eigh_reserve(n) allocates the solver's workspace ahead of the solve and keeps every later solve's buffers in a
    pool; gemma_hip_eigh_release hands them back.  Same bits with and without the pool, a second solve of the same order allocates
    nothing new, another order (and an odd one: embedded in n + 1) still works, release frees what was reserved."""
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "2")
    n = 1026
    A = _sym(n, 99, "kinship")
    U0, w0 = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(A.copy(), U0, w0)
    assert gpu_api.eigh_release() == 0  # nothing is kept unless asked for
    gpu_api.eigh_reserve(n)
    U1, w1 = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(A.copy(), U1, w1)
    U2, w2 = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(A.copy(), U2, w2)
    assert np.array_equal(w0, w1) and np.array_equal(U0, U1) and np.array_equal(U1, U2)
    m = 769  # odd: runs embedded in 770; the pool holds blocks of another order
    B = _sym(m, 5, "kinship") + 0.5 * np.eye(m)  # positive definite: EigenDecomp_Zeroed leaves every eigenvalue as it is
    Ub, wb = np.zeros((m, m)), np.zeros(m)
    monkeypatch.setenv("GEMMA_HIP_EIGH_POISON", "1")  # every block handed out, recycled or fresh, is filled with NaN patterns first
    gpu_api.EigenDecomp_Zeroed(B.copy(), Ub, wb)
    monkeypatch.delenv("GEMMA_HIP_EIGH_POISON")
    _check(B, Ub, wb, "pooled, another order, poisoned blocks")
    freed = gpu_api.eigh_release()
    assert freed > 5 * n * n * 8 // 2
    assert gpu_api.eigh_release() == 0
    U3, w3 = np.zeros((n, n)), np.zeros(n)
    gpu_api.EigenDecomp_Zeroed(A.copy(), U3, w3)
    assert np.array_equal(U0, U3) and gpu_api.eigh_release() == 0
