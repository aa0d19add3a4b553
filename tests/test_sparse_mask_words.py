"""The mask-word encoding of gemma_amd/csrc/i8gemm_sparse.hip.h restated in numpy (CPU): one word per (SNP row, 32 individuals)
= eight index nibbles + sixteen "kept" bits; the 2:4 operand a word describes, plus the calls it drops (groups of four with
more than two missing calls keep their first two), must add up to the dense mask for any product.  The hardware side -- that
v_smfmac_i32_32x32x64_i8 reads the word this way -- is scripts/smfmac_probe.hip / scripts/i8_sparse_proto.hip
(profiles/r02_smfmac_i8_layout_probe.txt, r02_i8_sparse_proto.txt) and tests/test_gpu_parity.py on the GPU."""
import numpy as np
import pytest


def encode(mask32):
    """mask32: 32 zeros / ones -> (idx, bits, dropped positions), as sparse_meta_kernel does."""
    idx = bits = 0
    dropped = []
    for g in range(8):
        pos = [q for q in range(4) if mask32[4 * g + q]]
        cnt = len(pos)
        p0 = pos[0] if cnt >= 1 else 0
        p1 = pos[1] if cnt >= 2 else (2 if p0 == 3 else 3)
        idx |= (p0 | (p1 << 2)) << (4 * g)
        bits |= ((cnt >= 1) | ((cnt >= 2) << 1)) << (2 * g)
        dropped += [4 * g + q for q in pos[2:]]
    return idx, bits, dropped


def expand(bits):
    """the sixteen kept bytes from the sixteen bits: (nibble * 0x204081) & 0x01010101 per four bytes (sp_expand)."""
    out = []
    for d in range(4):
        w = (((bits >> (4 * d)) & 0xF) * 0x00204081) & 0x01010101
        out += [(w >> (8 * q)) & 0xFF for q in range(4)]
    return out


def decode(idx, bits):
    """what the sparse instruction multiplies: kept value 2 g + e sits at position (nibble g >> 2 e) & 3 of group g."""
    vals = expand(bits)
    dense = np.zeros(32, dtype=np.int64)
    for g in range(8):
        nib = (idx >> (4 * g)) & 0xF
        for e in range(2):
            dense[4 * g + ((nib >> (2 * e)) & 3)] += vals[2 * g + e]
    return dense


@pytest.mark.parametrize("miss", [0.01, 0.12, 0.5, 1.0])
def test_mask_words_reproduce_the_mask(miss):
    rng = np.random.default_rng(int(miss * 1000) + 3)
    u = rng.integers(-128, 128, size=32)
    for _ in range(400):
        m = (rng.random(32) < miss).astype(np.int64)
        idx, bits, dropped = encode(m)
        assert 0 <= idx < 2 ** 32 and 0 <= bits < 2 ** 16
        d = decode(idx, bits)
        assert set(np.unique(d)) <= {0, 1}           # never two kept values on one position
        d[dropped] += 1
        assert np.array_equal(d, m)                  # kept + dropped = the mask
        assert int(d @ u) == int(m @ u)
        assert all(expand(bits)[k] in (0, 1) for k in range(16))


# ---- round 3: the 16-byte records of gemma_amd/csrc/i8gemm_sparse2.hip.h (sparse2_meta_kernel) ------------------------------
def record(packed_tile_row, p, h):
    """packed_tile_row: the 128 packed bytes g | m << 4 of one SNP row and one K-tile -> the record of chunk 2 p + h:
    (genotype word of step 2p, of step 2p+1, index word, kept bits), as sparse2_meta_kernel writes it."""
    words = []
    for e in range(2):
        src = packed_tile_row[32 * (2 * p + e) + 16 * h: 32 * (2 * p + e) + 16 * h + 16]
        w = 0
        for i in range(4):
            for j in range(4):
                w |= (int(src[4 * i + j]) & 3) << (8 * j + 2 * i)
        words.append(w)
    m = [(int(b) >> 4) & 1 for b in packed_tile_row[32 * (2 * p + h): 32 * (2 * p + h) + 32]]
    idx, bits16, dropped = encode(m)
    bits = 0
    for e in range(16):  # kept element e -> bit 8 (e % 4) + e / 4
        bits |= ((bits16 >> e) & 1) << (8 * (e & 3) + (e >> 2))
    return words[0], words[1], idx, bits, dropped


def unpack_g(w):
    """s2_unpack_g: operand dword i = (w >> 2 i) & 0x03030303; operand byte 4 i + j = individual 4 i + j of the lane's 16."""
    out = []
    for i in range(4):
        d = (w >> (2 * i)) & 0x03030303
        out += [(d >> (8 * j)) & 0xFF for j in range(4)]
    return out


def expand2(bits):
    """s2_expand: operand dword i = (bits >> i) & 0x01010101; byte 4 i + j = kept element 4 i + j."""
    out = []
    for i in range(4):
        d = (bits >> i) & 0x01010101
        out += [(d >> (8 * j)) & 0xFF for j in range(4)]
    return out


@pytest.mark.parametrize("miss", [0.01, 0.3])
def test_records_reproduce_genotypes_and_mask(miss):
    rng = np.random.default_rng(77 + int(100 * miss))
    for _ in range(60):
        g = rng.integers(0, 3, size=128)
        m = (rng.random(128) < miss).astype(np.int64)
        g[m == 1] = 0
        row = (g | (m << 4)).astype(np.uint8)
        u = rng.integers(-128, 128, size=128)
        for p in range(2):
            for h in range(2):
                w0, w1, idx, bits, dropped = record(row, p, h)
                assert max(w0, w1, idx, bits) < 2 ** 32
                # lane half h of the dense instruction of K-step 2 p + e holds individuals 32 (2 p + e) + 16 h + 0..15
                for e, w in ((0, w0), (1, w1)):
                    k0 = 32 * (2 * p + e) + 16 * h
                    assert unpack_g(w) == list(g[k0:k0 + 16])
                # lane half h of the sparse instruction of the pair covers the 32 individuals of K-step 2 p + h
                k0 = 32 * (2 * p + h)
                vals = expand2(bits)
                dense = np.zeros(32, dtype=np.int64)
                for gq in range(8):
                    nib = (idx >> (4 * gq)) & 0xF
                    for e in range(2):
                        dense[4 * gq + ((nib >> (2 * e)) & 3)] += vals[2 * gq + e]
                dense[dropped] += 1
                assert np.array_equal(dense, m[k0:k0 + 32])
                assert int(dense @ u[k0:k0 + 32]) == int(m[k0:k0 + 32] @ u[k0:k0 + 32])
