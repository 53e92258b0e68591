"""Kernel-body logic (the source hipcc compiles for gfx950) run through the sequential host simulator and
compared bit-for-bit with the oracle.  CPU-only pre-flight of the HIP path; the GPU tests repeat these
comparisons through the C ABI on the card."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import sim_lib as S
from dcdf_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
    G = json.load(f)


def check(a, **kw):
    okw = {k: v for k, v in kw.items() if k in ("fractional_bits", "round_")}
    ref, rs, rl, _ = O.chunk_build(a, want_snapshots=True, **okw)
    st, data, ns, nl, mm = S.encode(a, want_minmax=True, **kw)
    assert st == 0
    assert (ns, nl) == (rs, rl)
    if data != ref:
        n = min(len(data), len(ref))
        first = next((i for i in range(n) if data[i] != ref[i]), n)
        raise AssertionError("bytes differ: len %d vs %d, first diff at %d" % (len(data), len(ref), first))
    return data, mm


def array_n(n, T=12):
    a8 = np.array(G["array8"], dtype=np.int64)
    a = np.stack([a8[i % 3] for i in range(T)])
    idx = np.arange(n) % 8
    return a[:, idx][:, :, idx]


@pytest.mark.parametrize("n", [8, 16, 32, 64])
@pytest.mark.parametrize("dtype", [np.int64, np.int32])
def test_reference_fixture_tiled(n, dtype):  # testing.rs:242-249
    check(array_n(n).astype(dtype))


@pytest.mark.parametrize("shape", [(5, 8, 8), (7, 16, 16), (6, 32, 32), (5, 64, 64), (4, 128, 128)])
@pytest.mark.parametrize("kind", ["small", "wide", "noise", "const", "sparse"])
def test_random_unpadded(shape, kind):
    rng = np.random.default_rng(hash((shape, kind)) & 0xFFFF)
    T, R, Cc = shape
    if kind == "small":
        a = rng.integers(-3, 4, size=shape)
    elif kind == "wide":  # forces 2-4 byte DAC values
        a = rng.integers(-(2 ** 29), 2 ** 29, size=shape)
        a[1] = a[0] + rng.integers(-300, 300, size=(R, Cc))
    elif kind == "noise":
        a = rng.integers(0, 70000, size=shape)
    elif kind == "const":
        a = np.zeros(shape, dtype=np.int64) + 5
        a[2:] += 1
    else:
        base = rng.integers(-100, 100, size=(R, Cc))
        a = np.stack([base.copy() for _ in range(T)])
        for i in range(1, T):
            for _ in range(3):
                a[i, rng.integers(R), rng.integers(Cc)] += rng.integers(-500, 500)
            if i == 3:
                a[i, : R // 2, : Cc // 2] += 7  # "equal" quadrant (eqB = 1)
    a = a.astype(np.int64)
    check(a)
    check(a.astype(np.int32))
    check(a.astype(np.int32), force_novec=True)


@pytest.mark.parametrize("rows,cols", [(5, 3), (8, 7), (9, 9), (16, 9), (17, 4), (13, 31), (33, 20), (64, 1), (100, 77),
                                       (129, 130), (7, 8), (200, 256), (256, 255)])
def test_padded_shapes(rows, cols):
    rng = np.random.default_rng(rows * 1000 + cols)
    a = rng.integers(-9, 9, size=(5, rows, cols)).astype(np.int64)
    a[2] = a[1]
    a[3] = a[1] + 4
    a[4, : rows // 2] = a[1, : rows // 2]
    check(a)
    b = rng.integers(-40000, 40000, size=(4, rows, cols)).astype(np.int32)
    b[2, :, : max(1, cols // 2)] = b[0, :, : max(1, cols // 2)] - 3
    check(b)


def test_reference_padding_fixtures():  # snapshot.rs:560-572, log.rs:939-955
    data = np.zeros((3, 9, 9), dtype=np.int64) + 5
    data[:, :8, :8] = np.array(G["array8"], dtype=np.int64)
    data[0] = np.array(G["array9"], dtype=np.int64)[0]
    check(data)
    check(np.array(G["array9"], dtype=np.int64))


def test_synthetic_256():
    a = synth.cells(0xDCDF0001, 0, 6, 0, 256, 0, 256, np.int32)
    data, mm = check(a)
    assert (mm[:, 0] == a.reshape(6, -1).min(1)).all() and (mm[:, 1] == a.reshape(6, -1).max(1)).all()
    a64 = synth.cells(0xDCDF0001, 48, 52, 256, 512, 512, 768, np.int64)  # contains the repeated instant 49
    check(a64)


def test_non_contiguous_tile_view():
    big = synth.cells(7, 0, 4, 0, 96, 0, 160, np.int32)
    tile = big[:, 32:96, 64:128]
    assert not tile.flags["C_CONTIGUOUS"]
    check(tile)
    check(tile[:, ::1, ::2][:, :32, :32])  # non-unit column stride


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_float_fixture(dtype):  # testing.rs:251-339, mmarray.rs:1285,1403
    f8 = np.array([[[float("nan") if v == "nan" else v for v in row] for row in inst] for inst in G["farray8"]], dtype=dtype)
    a = np.stack([f8[i % 6] for i in range(12)])
    idx = np.arange(16) % 8
    a = np.ascontiguousarray(a[:, idx][:, :, idx])
    check(a, fractional_bits=3, round_=False)
    check(a, fractional_bits=2, round_=True)  # lossy
    st, data, _, _ = S.encode(a, fractional_bits=2, round_=False)
    assert st == -3 and data is None  # precision loss (fixed.rs:47-59)
    b = a.copy()
    b[3, 2, 2] = np.inf
    assert S.encode(b, fractional_bits=3)[0] == -2  # fixed.rs:39-41
    with pytest.raises(O.OracleError):
        O.chunk_build(b, fractional_bits=3)


def test_254_log_cap():  # chunk.rs:62
    base = (np.add.outer(np.arange(8), np.arange(8)) % 5).astype(np.int64)
    a = np.stack([base.copy() for _ in range(300)])
    for i in range(1, 300):
        a[i, i % 8, (i * 3) % 8] += 1
    check(a)


def test_value_range_contract():
    a = np.zeros((2, 8, 8), dtype=np.int64)
    a[1, 3, 3] = 2 ** 30
    assert S.encode(a)[0] == -8  # outside the fast path's |v| < 2^30 contract -> ST_UNSUPPORTED
    a[1, 3, 3] = 2 ** 30 - 1
    a[0, 0, 0] = -(2 ** 30)
    check(a)


def test_output_capacity_reported():
    a = np.random.default_rng(0).integers(0, 1000, size=(3, 16, 16)).astype(np.int32)
    ref = O.chunk_build(a)
    assert S.encode(a, cap=len(ref) - 1)[0] == -100
    st, data, _, _ = S.encode(a, cap=len(ref))
    assert st == 0 and data == ref


def _stash_logs():
    S.lib().sim_last_stash_logs.restype = __import__("ctypes").c_uint32
    return S.lib().sim_last_stash_logs()


def test_log_stash_used_and_fallback(monkeypatch):
    """Logs are normally emitted from the LDS stash built in phase 1; when the stash cannot hold an instant (forced
    here by shrinking it) or a value does not fit 16 bits, the re-reading passes produce the same bytes."""
    a = synth.cells(0xDCDF0002, 0, 5, 0, 256, 0, 256, np.int32)
    data, _ = check(a)
    _, _, ns, nl = S.encode(a)
    assert nl > 0 and _stash_logs() == nl  # every log came from the stash
    for words in ("1", "4000", "9000"):
        monkeypatch.setenv("K2R_SIM_STASH_WORDS", words)
        data2, _ = check(a)
        assert data2 == data
        assert _stash_logs() < nl
    monkeypatch.delenv("K2R_SIM_STASH_WORDS")
    # log differences beyond 16 bits (but a log still wins): not stashable -> fallback
    rng = np.random.default_rng(5)
    base = rng.integers(0, 50, size=(64, 64))
    b = np.stack([base, base, base]).astype(np.int64)
    b[1, 5, 7] += 100000
    b[2, 9, 9] -= 70000
    b[2, 40:44, 40:44] += rng.integers(-40000, 40000, size=(4, 4))
    check(b)
    _, _, ns, nl = S.encode(b)
    assert nl == 2 and _stash_logs() == 0
    # ...and the same shape with small differences does use it
    c = np.stack([base, base, base]).astype(np.int64)
    c[1, 5, 7] += 1000
    c[2, 40:44, 40:44] += rng.integers(-400, 400, size=(4, 4))
    check(c)
    _, _, ns, nl = S.encode(c)
    assert nl == 2 and _stash_logs() == 2


def test_value_range_contract_int32_rows():
    """The 16-byte row loader does not look at cells one by one: the block extremes enforce |v| < 2^30."""
    a = np.zeros((2, 16, 16), dtype=np.int32)
    a[1, 9, 9] = 2 ** 30
    assert S.encode(a)[0] == -8
    a[1, 9, 9] = -(2 ** 30) - 1
    assert S.encode(a)[0] == -8
    a[1, 9, 9] = 2 ** 30 - 1
    a[0, 1, 1] = -(2 ** 30)
    check(a)


@pytest.mark.parametrize("ftype", [np.float32, np.float64])
@pytest.mark.parametrize("side", [16, 64])
def test_float32_rows_fast_conversion(side, ftype):
    """float32 tiles with aligned rows take the converting row loader (fixed4_f32); everything its fast path does not
    cover (NaN, fractions beyond the chosen bits with rounding, negative fractions, huge values) must agree with
    to_fixed (fixed.rs:31-71) as restated by the oracle."""
    rng = np.random.default_rng(side)
    base = rng.integers(-2000, 2000, size=(side, side))
    a = np.stack([base + rng.integers(-30, 30, size=(side, side)) * (rng.random((side, side)) < 0.2) for _ in range(6)])
    f = (a / 8.0).astype(ftype)
    f[2, 3, 5] = np.nan
    f[4, 0, 0] = np.nan
    check(f, fractional_bits=3)
    check(f, fractional_bits=5)
    g = f.copy()
    g[1, 7, 7] = ftype(5.0 + 1 / 64.0)  # a positive value that needs 6 bits
    assert S.encode(g, fractional_bits=3)[0] == -3  # precision loss without rounding (fixed.rs:47-59)
    check(g, fractional_bits=3, round_=True)
    g[1, 7, 7] = ftype(-1.3)  # negative fraction: the reference neither rounds nor rejects it (fixed.rs:45)
    check(g, fractional_bits=3)
    h = f.copy()
    h[3, 2, 2] = ftype(3e8)  # stored value beyond the fast path's 2^30 contract
    assert S.encode(h, fractional_bits=3)[0] == -8
    h[3, 2, 2] = ftype(1e30)  # beyond i64: the reference panics (fixed.rs:62-68)
    assert S.encode(h, fractional_bits=3)[0] == -4
    h[3, 2, 2] = -np.inf
    assert S.encode(h, fractional_bits=3)[0] == -2


from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 6), st.integers(1, 40), st.integers(1, 40), st.integers(0, 2 ** 32 - 1),
       st.sampled_from(["tiny", "smooth", "blocks", "big"]), st.sampled_from([np.int32, np.int64]))
def test_property_random_tiles(instants, rows, cols, seed, kind, dtype):
    """Random small tiles of any shape (padding, 1-wide tiles, single instants), value distributions that exercise
    uniform subtrees, "equal" subtrees, multi-byte Dac planes and snapshot/log switches: kernel bodies == oracle."""
    if max(rows, cols) > 32 and instants > 4:
        instants = 4
    rng = np.random.default_rng(seed)
    shape = (instants, rows, cols)
    if kind == "tiny":
        a = rng.integers(-2, 3, size=shape)
    elif kind == "smooth":
        base = np.add.outer(np.arange(rows) * 3, np.arange(cols) * 5)
        a = np.stack([base + rng.integers(-1, 2, size=(rows, cols)) * (rng.random((rows, cols)) < 0.1) + 7 * t for t in range(instants)])
    elif kind == "blocks":
        a = np.zeros(shape, dtype=np.int64)
        for t in range(instants):
            a[t] = (rng.integers(0, 3, size=((rows + 3) // 4, (cols + 3) // 4)).repeat(4, 0).repeat(4, 1))[:rows, :cols] * 100
            if t and rng.random() < 0.5:
                a[t] = a[t - 1] + rng.integers(-3, 4)  # whole instant "equal" to its predecessor up to a constant
    else:
        a = rng.integers(-(2 ** 29), 2 ** 29, size=shape)
        if instants > 1:
            a[1:] = a[0] + rng.integers(-70000, 70000, size=(instants - 1, rows, cols))
            a = np.clip(a, -(2 ** 30) + 1, 2 ** 30 - 1)
    a = np.ascontiguousarray(a).astype(dtype)
    if max(rows, cols) < 5:  # sidelen < 8 is outside the fast path: reported, not encoded
        assert S.encode(a)[0] == -8
        return
    check(a)


def test_speculative_halves():
    """A chunk encoded as two or more work items (k2r_encode.h "speculative parts", spliced as k_stitch does): byte-identical to the
    sequential encode whenever the first half holds a single block; reported as -102 (re-encode whole) when a block boundary
    falls into the first half; errors of either half surface."""
    L = S.lib()
    try:
        rng = np.random.default_rng(9)                                       # one block: every log beats its snapshot
        a = (rng.integers(0, 4000, size=(64, 64))[None] + (rng.random((9, 64, 64)) < 0.03) * rng.integers(-300, 300, size=(9, 64, 64))).astype(np.int32)
        a[5] = a[4]
        assert S.encode(a)[2] == 1
        ref = check(a)[0]
        for m in (1, 2, 4, 5, 8):
            L.sim_set_split(m)
            assert check(a)[0] == ref
        for p in (2, 3, 4):                                                  # parts of decreasing length (k2r_capi_encode.hip part_bounds)
            L.sim_set_parts(p)
            assert check(a)[0] == ref
            big = np.concatenate([a, a[::-1], a])                            # 27 instants: parts at 14, 21, 24
            L.sim_set_parts(0)
            one_block = S.encode(big)[2] == 1
            L.sim_set_parts(p)
            if one_block:
                check(big)
            else:
                assert S.encode(big)[0] in (0, -102)
        L.sim_set_split(4)
        check(a[:, :50, :37])                                                # padded tile
        check(a.astype(np.int64) * 2 + 1)
        check((a / 8.0).astype(np.float32), fractional_bits=3)
        b = a.copy()                                                         # a block boundary INSIDE the second half
        b[6:] = np.random.default_rng(6).integers(0, 1 << 20, size=b[6:].shape)
        st, data, ns, nl = S.encode(b)
        assert st == 0 and ns >= 2
        ref_b = O.chunk_build(b)
        assert data == ref_b
        for m in (2, 4, 6):
            L.sim_set_split(m)
            assert S.encode(b)[1] == ref_b
        L.sim_set_split(7)                                                   # boundary at instant 6 < 7: the assumption fails
        assert S.encode(b)[0] == -102
        L.sim_set_parts(4)                                                   # 9 instants: parts at 5, 7, 8 -- the last two assume too much
        assert S.encode(b)[0] == -102
        w = a.copy()                                                         # a snapshot beyond 16 bits: no compact copy to share
        w[0, 0, 0] = 200000
        L.sim_set_parts(3)
        assert S.encode(w)[1] == O.chunk_build(w)
        c = array_n(16, T=12).astype(np.int64)                               # the reference fixture: blocks of 3 (a8 repeats)
        _, _, ns, _ = S.encode(c)
        L.sim_set_split(6)
        assert S.encode(c)[0] == (-102 if ns > 1 else 0)
        d = a.astype(np.int64)                                               # an out-of-contract value in the second half only
        d[7, 3, 3] = 2 ** 31
        L.sim_set_split(4)
        assert S.encode(d)[0] == -8
    finally:
        L.sim_set_split(0)
        L.sim_set_parts(0)


def test_stash_overflow_goes_to_global_scratch():
    """A snapshot instant with constant 64x64 blocks makes every quad under them internal in the logs that follow: more stash
    records than the LDS pool's shares hold (EncPool::CAPI_REC / CAPQ_REC).  The excess goes to global scratch and the logs
    are still emitted from the stash (no re-reading fallback), byte-identical."""
    a = synth.cells(0xDCDF0002, 0, 3, 0, 256, 0, 256, np.int32)
    a[0, :128, :128] = 77       # 4 of the 16 blocks of the snapshot instant constant: 4096 internal quads there alone
    a[2, 64:128, 64:192] = -5   # and a constant stretch in a later instant
    data, _ = check(a)
    _, _, ns, nl = S.encode(a)
    assert ns == 1 and nl == 2 and _stash_logs() == 2
