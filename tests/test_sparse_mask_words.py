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
