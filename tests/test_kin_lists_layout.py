"""The round-3 form of the integer kinship's correction (gemma_amd/csrc/kin_i8.hip.h: kin_i8_pack2_kernel, kin_i8_count /
scan / fill / sub kernels, kin_i8_corr2_kernel) restated in numpy, on the CPU: the CSR lists of the missing calls both ways,
the 2-bit copy of the block with its thread-major layout, the both-missing term in 2^-44 fixed point, and the upper-triangle
tile map of G^T G -- put together they must give the centred kinship of the oracle's restatement of PlinkKin.  The device side
is tests/test_gpu_parity.py::test_kinship_integer_path."""
import numpy as np

SEG = 4096          # KI8_SEG
FIX = 2.0 ** 44     # KI8_FIX


def pack2(G, nseg):
    """dword 256 seg + t of a row holds the individuals SEG seg + 256 q + t, q = 0 .. 15, at bits 2 q .. 2 q + 1"""
    l, n = G.shape
    A2 = np.zeros((l, 256 * nseg), dtype=np.uint32)
    for seg in range(nseg):
        for q in range(16):
            i = SEG * seg + 256 * q + np.arange(256)
            ok = i < n
            vals = np.zeros((l, 256), dtype=np.uint32)
            vals[:, ok] = G[:, i[ok]]
            A2[:, 256 * seg:256 * seg + 256] |= vals << np.uint32(2 * q)
    return A2


def csr(mask):
    """rows of a 0/1 matrix -> (offsets, ascending positions)"""
    cnt = mask.sum(1)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    return off, np.nonzero(mask)[1].astype(np.int64)


def test_lists_two_bit_copy_and_fixed_point_reproduce_the_centred_kinship(oracle):
    rng = np.random.default_rng(31)
    n, l = 4503, 260                       # two ranges of SEG individuals, ragged
    maf = rng.uniform(0.05, 0.5, l)
    g = (rng.random((l, n)) < maf[:, None]).astype(np.int64) + (rng.random((l, n)) < maf[:, None])
    miss = rng.random((l, n)) < 0.03
    miss[:, 17] = True                     # an individual that is never called
    miss[9, rng.random(n) < 0.9] = True    # a SNP missing for most: the "long list" walk of the both-missing pass
    X = np.where(miss, np.nan, g.astype(np.float64))
    K_ref = oracle.calc_kin(X, 1)

    G = np.where(miss, 0, g)
    mu = G.sum(1) / (n - miss.sum(1))
    nseg = (n + SEG - 1) // SEG
    A2 = pack2(G, nseg)
    offJ, listJ = csr(miss.T)              # per individual: the SNPs it is missing at
    offS, listS = csr(miss)                # per SNP: the individuals missing
    # sub[s][b] = entries of listS[s] below b SEG
    sub = np.stack([np.searchsorted(listS[offS[s]:offS[s + 1]], SEG * np.arange(nseg + 1)) for s in range(l)])
    assert np.all(sub[:, -1] == np.diff(offS))

    a = (mu[:, None] * G).sum(0)
    cj = np.array([(mu[listJ[offJ[j]:offJ[j + 1]]] ** 2).sum() for j in range(n)])
    S = np.zeros((n, n))
    worst_fix = 0.0
    for j in range(0, n, 97):              # a sample of rows j (the block (j, seg) of the kernel)
        sl = listJ[offJ[j]:offJ[j + 1]]
        for seg in range(nseg):
            i0 = SEG * seg
            acc = np.zeros((16, 256))
            for s in sl:                   # genotype term from the 2-bit copy: thread t, accumulator q <-> i0 + 256 q + t
                w = A2[s, 256 * seg:256 * seg + 256]
                for q in range(16):
                    acc[q] += mu[s] * ((w >> np.uint32(2 * q)) & np.uint32(3))
            trow = np.zeros(SEG, dtype=np.int64)   # both missing: integers in 2^-44 fixed point, any order
            exact = np.zeros(SEG)
            for s in sl:
                h = int(np.rint(0.5 * mu[s] * mu[s] * FIX))
                ii = listS[offS[s] + sub[s, seg]:offS[s] + sub[s, seg + 1]] - i0
                trow[ii] += h
                exact[ii] += 0.5 * mu[s] * mu[s]
            worst_fix = max(worst_fix, float(np.abs(trow / FIX - exact).max()))
            for q in range(16):
                i = i0 + 256 * q + np.arange(256)
                ok = i < n
                S[j, i[ok]] = acc[q][ok] + trow[256 * q + np.arange(256)][ok] / FIX - cj[j]
    assert worst_fix <= l * 2.0 ** -45     # <= 2^-45 per term
    # K = G^T G - a_i - a_j + sum mu^2 + S_ij + S_ji on the sampled rows j and every i (needs S_ij too: take i from the sample)
    rows = np.arange(0, n, 97)
    GtG = (G.T @ G).astype(np.float64)
    sub_ref = K_ref[np.ix_(rows, rows)] * l
    got = GtG[np.ix_(rows, rows)] - a[rows][:, None] - a[rows][None, :] + (mu ** 2).sum() + S[np.ix_(rows, rows)] + S[np.ix_(rows, rows)].T
    assert np.abs(got - sub_ref).max() <= 1e-12 * np.abs(sub_ref).max()


def test_upper_triangle_tile_map_covers_what_the_fold_reads():
    """kin_i8_tile_map: the 128 x 256 tiles (tm, tn) with tn >= tm >> 1, group by group of eight tile rows -- every entry with
    column >= row lies in a listed tile, no tile is listed twice, and the accumulate kernel's row bound per column block
    (rows < 256 (bx + 1)) stays inside the listed tiles."""
    for n in (20000, 4503, 300, 50000):
        tiles_m, tiles_n = (n + 127) // 128, (n + 255) // 256
        tiles = []
        for first in range(0, tiles_m, 8):
            gsz = min(8, tiles_m - first)
            for tn in range(first >> 1, tiles_n):
                for tm in range(first, first + gsz):
                    if tn >= (tm >> 1):
                        tiles.append((tm, tn))
        assert len(tiles) == len(set(tiles))
        have = set(tiles)
        for tm in range(tiles_m):
            for tn in range(tiles_n):
                meets = 256 * tn + 255 >= 128 * tm       # some column >= some row of the tile
                assert ((tm, tn) in have) == meets
        for bx in range(tiles_n):                        # accumulate kernel: rows i < min(n, 256 (bx + 1)) of column block bx
            for i in (0, min(n, 256 * (bx + 1)) - 1):
                assert (i // 128, bx) in have
        assert len(tiles) < 0.52 * tiles_m * tiles_n + tiles_m + tiles_n


def test_word_wise_plink_ingest_lut_and_ragged_tail(oracle):
    """ingest_i8_kernel's word path (gemma_amd/csrc/i8gemm.hip.h): code c of a .bed word -> packed byte (0x00011002 >> 8 c) & 0xFF
    = g | m << 4 with g = 2, 0 (missing), 1, 0 for c = 0, 1, 2, 3; calls past n inside the last word are ignored; the mean is
    sum g / (n - missing).  Against the oracle's bed_decode (the reference's PLINK convention, src/gemma_io.cpp:1665-1682)."""
    rng = np.random.default_rng(5)
    for n in (777, 800, 1003, 16, 17):
        p = 40
        codes = rng.choice([0, 1, 2, 3], size=(p, n), p=[0.3, 0.06, 0.34, 0.3]).astype(np.uint8)
        nb = (n + 3) // 4
        pad = np.ones((p, nb * 4), dtype=np.uint8)      # padding bits set to "missing": must not be counted
        pad[:, :n] = codes
        raw = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
        X = oracle.bed_decode(raw, n)                    # NaN where missing
        ldk = (n + 127) // 128 * 128
        A = np.zeros((p, ldk), dtype=np.uint8)
        mean = np.zeros(p)
        for s in range(p):
            itot = imiss = 0
            for k in range(ldk // 16):                   # one lane = one 32-bit word = 16 calls
                i0 = 16 * k
                if i0 >= n:
                    continue
                nvalid = min(16, n - i0)
                w = 0
                for b in range(4):
                    if i0 // 4 + b < nb:
                        w |= int(raw[s, i0 // 4 + b]) << (8 * b)
                for q in range(nvalid):
                    c = (w >> (2 * q)) & 3
                    byte = (0x00011002 >> (8 * c)) & 0xFF
                    A[s, i0 + q] = byte
                    itot += byte & 3
                    imiss += byte >> 4
            mean[s] = itot / (n - imiss) if n > imiss else np.nan
        g = (A & 3).astype(np.float64)[:, :n]
        m = (A >> 4)[:, :n].astype(bool)
        assert np.array_equal(m, np.isnan(X)) and np.array_equal(g[~m], X[~m]) and not g[m].any()
        assert not A[:, n:].any()
        with np.errstate(invalid="ignore"):
            ref_mean = np.nansum(X, axis=1) / (~np.isnan(X)).sum(1)
        assert np.array_equal(mean, ref_mean)
