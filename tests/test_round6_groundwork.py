"""Numerical groundwork of two round-6 items (no device code involved; numpy restatements of what the kernels do).

(A) The digit planes of U (gemma_amd/csrc/i8gemm.hip.h: u_scale_kernel, u_digits_kernel): every column is scaled by
    q_j = 0.99 * 2^(8 D - 1) / max_k |U_kj| and cut into D balanced base-256 digits in [-128, 127].  The claims the kernels rest on:
    the scaled integer always fits D digits (0.99 * 2^(8 D - 1) < 127 / 255 * (256^D - 1)), the digits reproduce it exactly, U is
    recovered to 1.01 * 2^(-8 D) of the column maximum, int32 accumulation of a digit product cannot overflow at the sizes the path
    runs at, and the 7g6m form's mask product (the upper six of seven digits) is the 7-digit integer rounded at 2^8.
(B) bench.py --config 4: how p = 500 000 SNPs are cut over the ranks (steps x equal blocks of at most 20 000 SNPs)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digits(V, D):
    """u_digits_kernel's loop on Python integers: low byte as a signed digit, exact shift"""
    out = []
    v = int(V)
    for _ in range(D):
        dig = ((v & 0xFF) ^ 0x80) - 0x80
        out.append(dig)
        v = (v - dig) >> 8
    return out, v


def test_exact_maximum_scale_and_balanced_digits():
    rng = np.random.default_rng(6)
    for D in (6, 7):
        L = 0.99 * 2.0 ** (8 * D - 1)
        assert L < 127 / 255 * (256 ** D - 1)          # the largest scaled magnitude is representable ...
        assert -L > -128 / 255 * (256 ** D - 1)        # ... on both sides
        n = 400
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        U = Q * rng.choice([1.0, 1e-3, 7.5], size=n)[None, :]  # columns of very different scale
        cmax = np.abs(U).max(axis=0)
        q, qinv = L / cmax, cmax / L
        V = np.rint(U * q[None, :])
        assert np.abs(V).max() <= L + 0.5 + L * 1e-15 < 127 / 255 * (256 ** D - 1)  # rounding may add half a unit: still representable
        worst = 0.0
        for j in range(0, n, 37):
            for k in range(n):
                d, rest = _digits(V[k, j], D)
                assert rest == 0 and all(-128 <= x <= 127 for x in d)
                assert sum(x * 256 ** i for i, x in enumerate(d)) == int(V[k, j])
                if D == 7:  # the mask product of the 7g6m form multiplies digits 1..6: the integer rounded to a multiple of 256
                    top = sum(x * 256 ** i for i, x in enumerate(d) if i >= 1)
                    assert abs(top - int(V[k, j])) <= 128
            worst = max(worst, float(np.max(np.abs(V[:, j] - U[:, j] * q[j]))))  # in units of the last digit
        assert worst <= 0.5 + 2.0 ** -4                       # (the products U q carry their own fp64 rounding at this magnitude)
        assert 0.5 / L <= 1.011 * 2.0 ** (-8 * D)              # = half a unit in terms of the column maximum
        assert np.all(np.abs(q * qinv - 1.0) <= 2.3e-16)       # scale and inverse scale: one more relative rounding, no more
        worst = 0.5 / L
        # the power-of-two scale of rounds 1-5 for comparison: between 2x and 4x coarser
        e = np.frexp(cmax)[1]
        Vp = np.rint(np.ldexp(U, (8 * D - 2) - e[None, :]))
        errp = np.max(np.abs(np.ldexp(Vp, e[None, :] - (8 * D - 2)) - U) / cmax[None, :])
        if D == 6:  # (with seven digits both roundings are below what doubles resolve)
            assert 1.9 * worst <= errp * 1.05 and errp <= 4.1 * 2.0 ** (-8 * D), (worst, errp)
    # int32 accumulation: a digit product sums n terms of |g| <= 2 times |digit| <= 128; fused planes hold 256 C_hi + C_lo
    assert 32640 * 2 * 128 * 257 < 2 ** 31 <= 32641 * 2 * 128 * 257 + 2 ** 24   # the fuse bound of i8_begin (n <= 32 640)
    assert 65536 * 2 * 128 < 2 ** 31                                            # unfused planes: any n the path indexes


def test_config4_preset_cuts_p_over_the_ranks():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 3, 4, 8):
        sys.argv = ["bench.py", "--config", "4", "--gpus", str(world)]
        args = bench.parse()
        preset = bench.apply_config(args, world)
        assert preset["config"] == 4 and args.n == 50000 and args.kin_snps == 500000
        assert args.batch <= 20000 and args.batch * args.steps * world >= 500000
        assert args.batch * args.steps * world < 500000 + args.steps * world  # nothing but the rounding of the cut on top
        assert args.fp64_steps == 0 and args.e2e_snps == 0 and args.c4_leg == 0
    sys.argv = ["bench.py"]
    assert bench.apply_config(bench.parse(), 1) is None
