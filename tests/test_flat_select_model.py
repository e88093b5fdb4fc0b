"""CPU model of svsx_flat.hip's `flat_select_kernel`, register path (lists of up to 512 keys, eight per lane): the kk-th smallest key of a
query's candidate lists by a radix descent over the order-encoded keys -- the bits all keys share are skipped, padding slots hold 0xFFFFFFFF and
must never be counted, and a bit is decided by how many keys that match the prefix so far have it clear.  The model follows the device code
(enc_key / dec_key, `diff` as the OR of key XOR key 0 over the valid slots, `want = prefix >> bit`, one compare per slot) and must equal
numpy's partition, including ties, negative keys, -0.0 / +0.0, infinities and the all-equal case."""
import numpy as np
import pytest


def enc_key(f):
    b = np.float32(f).view(np.uint32)
    return np.uint32(~b) if b & np.uint32(0x80000000) else np.uint32(b | np.uint32(0x80000000))


def dec_key(u):
    u = np.uint32(u)
    return (np.uint32(u & np.uint32(0x7FFFFFFF)) if u & np.uint32(0x80000000) else np.uint32(~u)).view(np.float32)


def select_kth(keys, kk, regs=8):
    total = len(keys)
    assert kk <= total <= 64 * regs
    v = np.full(64 * regs, 0xFFFFFFFF, np.uint32)              # slot e = lane + 64 r holds element e; the rest is padding
    v[:total] = [enc_key(x) for x in keys]
    v0 = v[0]
    diff = np.uint32(0)
    for e in range(total):
        diff |= v[e] ^ v0
    if diff == 0:
        return dec_key(v0)
    top = int(diff).bit_length() - 1
    prefix = np.uint32(0) if top == 31 else np.uint32(v0 & np.uint32(~((2 << top) - 1) & 0xFFFFFFFF))
    remaining = kk
    for bit in range(top, -1, -1):
        want = int(prefix) >> bit                                 # this bit still zero
        zeros = int(np.count_nonzero((v >> np.uint32(bit)) == np.uint32(want)))   # padding included: it must never match
        if remaining > zeros:
            remaining -= zeros
            prefix = np.uint32(prefix | np.uint32(1 << bit))
    return dec_key(prefix)


def test_order_encoding_is_monotone_and_invertible():
    xs = np.array([-np.inf, -3.5, -1e-30, -0.0, 0.0, 1e-30, 2.0, 7.25, np.inf], np.float32)
    enc = [int(enc_key(x)) for x in xs]
    assert enc == sorted(enc)
    for x in xs:
        assert dec_key(enc_key(x)).view(np.uint32) == np.float32(x).view(np.uint32)


@pytest.mark.parametrize("kk", [1, 12, 14, 40, 232])
def test_radix_descent_equals_partition(kk):
    rng = np.random.default_rng(11 * kk)
    for trial in range(8):
        total = int(rng.integers(kk, 513))
        keys = rng.standard_normal(total).astype(np.float32)
        if trial == 1:
            keys = np.abs(keys) + np.float32(0.75)                # a narrow positive range: many shared leading bits
        if trial == 2:
            keys = np.round(keys * 2) / 2                         # exact ties across the kk-th position
        if trial == 3:
            keys[:] = np.float32(-2.5)                            # every key the same
        if trial == 4:
            keys[: total // 3] = np.float32(np.inf)               # the coarsest level's +inf threshold leaves such keys behind
        if trial == 5:
            keys = -np.abs(keys)                                  # all negative: the encoding inverts them
        if trial == 6:
            keys[::7] = np.float32(-0.0)
            keys[3::7] = np.float32(0.0)
        want = np.partition(keys, kk - 1)[kk - 1]
        got = select_kth(list(keys), kk)
        # -0.0 and +0.0 compare equal as floats; the encoding orders -0.0 first, either is the same threshold
        assert got == want or (got == 0 and want == 0), (kk, trial, total, got, want)


def test_a_full_house_of_padding_is_never_counted():
    """total = kk = 1: 511 padding slots, one key -- the answer is that key (diff == 0 path); total = 65: one full register row + 1."""
    assert select_kth([np.float32(3.0)], 1) == np.float32(3.0)
    keys = list(np.linspace(-1, 1, 65).astype(np.float32))
    assert select_kth(keys, 65) == np.float32(1.0)
    assert select_kth(keys, 1) == np.float32(-1.0)


def test_accumulator_cells_of_a_block_cover_its_32_rows_once():
    """flat_thresh_wg_kernel: with the rows as the MFMA's A operand a lane (half = lane / 32, query = lane % 32) holds the cells of rows
    8 (r / 4) + 4 half + r % 4, r = 0 .. 15 (v_mfma_f32_32x32x16_f16's C / D layout); the per-row constants are read as four float4 at
    8 g + 4 half, the append path walks the cells in five groups of three plus cell 15.  Every row of the block must be owned exactly once,
    the float4 reads must line up with the cells, and the groups must cover all 16 cells."""
    rows = sorted(8 * (r >> 2) + 4 * half + (r & 3) for half in (0, 1) for r in range(16))
    assert rows == list(range(32))
    for half in (0, 1):
        for g in range(4):
            assert [8 * g + 4 * half + e for e in range(4)] == [8 * ((4 * g + e) >> 2) + 4 * half + ((4 * g + e) & 3) for e in range(4)]
    cells = sorted([3 * g + e for g in range(5) for e in range(3)] + [15])
    assert cells == list(range(16))
    assert 8 * (15 >> 2) + (15 & 3) == 27  # the literal the kernel uses for cell 15
