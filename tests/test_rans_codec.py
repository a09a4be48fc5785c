"""CPU: the host entropy coder (csrc/rans.cpp behind icm_amd.ans) against
  (1) oracle/rans_oracle.py -- an exact-integer Python restatement of the stream format (state machine of
      third_party/ryg_rans/rans64.h + CompressAI's published interface conventions), byte for byte;
  (2) oracle/_ref/librans64_ref.so -- a shim compiled from the reference's OWN rans64.h (oracle/build_ref.sh): the
      escape-free streams of one table must be identical word for word, and the reference decoder must read ours.
The reference holds no golden bitstreams and its coder binaries are never run: the interface layer above the state
machine is "parity unpinned" (DESIGN.md 2)."""
import ctypes
import os

import numpy as np
import pytest

from icm_amd import ans
from oracle import rans_oracle as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "librans64_ref.so")


def _tables(rng, n=5):
    cdfs, sizes, offs = [], [], []
    for k in range(n):
        m = int(rng.integers(2, 40))
        p = rng.random(m).astype(np.float32) ** 3
        if k == 1:
            p[: m // 2] = 0.0                      # zero-probability symbols: the cdf must still give them width
        p = p / p.sum()
        c = R.pmf_to_quantized_cdf(list(p) + [1e-4])
        cdfs.append(c)
        sizes.append(len(c))
        offs.append(-int(rng.integers(0, m)))
    return cdfs, sizes, offs


def test_pmf_to_quantized_cdf_matches_oracle_and_invariants():
    rng = np.random.default_rng(1)
    for trial in range(40):
        m = int(rng.integers(1, 120))
        p = rng.random(m).astype(np.float32) ** int(rng.integers(1, 9))
        if trial % 3 == 0:
            p[rng.random(m) < 0.5] = 0.0
        if p.sum() == 0:
            p[0] = 1.0
        p = (p / p.sum()).astype(np.float32)
        tail = np.float32(rng.random() * 1e-3)
        a = ans.pmf_to_quantized_cdf(list(p) + [tail])      # same input to both implementations
        b = R.pmf_to_quantized_cdf(list(p) + [tail])
        assert a == b
        assert a[0] == 0 and a[-1] == 65536 and all(y > x for x, y in zip(a, a[1:])), "every symbol needs width"
    with pytest.raises(ValueError):
        ans.pmf_to_quantized_cdf([0.5, -0.1])
    with pytest.raises(ValueError):
        ans.pmf_to_quantized_cdf([0.0, 0.0])
    with pytest.raises(ValueError):
        ans.pmf_to_quantized_cdf([float("nan")])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_stream_is_byte_identical_to_oracle_and_round_trips(seed):
    rng = np.random.default_rng(seed)
    cdfs, sizes, offs = _tables(rng)
    n = 6000
    idx = rng.integers(0, len(cdfs), n)
    sym = np.empty(n, dtype=np.int64)
    for i, k in enumerate(idx):
        r = rng.random()
        if r < 0.03:
            sym[i] = int(rng.integers(-5000, 5000))          # far outside the table: multi-nibble escapes
        elif r < 0.06:
            sym[i] = offs[k] + sizes[k] - 2 + int(rng.integers(0, 3))   # just above the table (raw = 0, 2, 4)
        elif r < 0.09:
            sym[i] = offs[k] - 1 - int(rng.integers(0, 3))
        else:
            sym[i] = offs[k] + int(rng.integers(0, sizes[k] - 2))
    mine = ans.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdfs, sizes, offs)
    ref = R.encode_with_indexes(sym.tolist(), idx.tolist(), cdfs, sizes, offs)
    assert mine == ref
    assert ans.RansDecoder().decode_with_indexes(mine, idx.tolist(), cdfs, sizes, offs) == sym.tolist()
    assert R.decode_with_indexes(mine, idx.tolist(), cdfs, sizes, offs) == sym.tolist()
    # RansDecoder.set_stream + several decode_stream calls consume ONE stream (cnn.py:300-318)
    d = ans.RansDecoder()
    d.set_stream(mine)
    cut = [0, 1, 777, 778, 4000, n]
    out = []
    for a, b in zip(cut, cut[1:]):
        out += d.decode_stream(idx[a:b].tolist(), cdfs, sizes, offs)
    assert out == sym.tolist()
    # BufferedRansEncoder: several encode calls, one flush == one encode of the concatenation (cnn.py:228,263-264)
    be = ans.BufferedRansEncoder()
    for a, b in zip(cut, cut[1:]):
        be.encode_with_indexes(sym[a:b].tolist(), idx[a:b].tolist(), cdfs, sizes, offs)
    assert be.flush() == mine


def test_edge_cases():
    cdfs, sizes, offs = [[0, 30000, 65535, 65536]], [4], [0]
    enc, dec = ans.RansEncoder(), ans.RansDecoder()
    empty = enc.encode_with_indexes([], [], cdfs, sizes, offs)
    assert len(empty) == 8 and empty == R.encode_with_indexes([], [], cdfs, sizes, offs)   # just the flushed state
    assert dec.decode_with_indexes(empty, [], cdfs, sizes, offs) == []
    big = [2 ** 30, -2 ** 30, 0, 1, 2, 3, -1]
    s = enc.encode_with_indexes(big, [0] * len(big), cdfs, sizes, offs)
    assert s == R.encode_with_indexes(big, [0] * len(big), cdfs, sizes, offs)
    assert dec.decode_with_indexes(s, [0] * len(big), cdfs, sizes, offs) == big
    with pytest.raises(ValueError):
        enc.encode_with_indexes([0], [3], cdfs, sizes, offs)          # index outside the tables
    with pytest.raises(ValueError):
        enc.encode_with_indexes([0, 1], [0], cdfs, sizes, offs)       # ragged inputs
    with pytest.raises(ValueError):
        enc.encode_with_indexes([0], [0], [[0, 0, 65536]], [3], [0])  # zero-width symbol in the table
    with pytest.raises(ValueError):
        dec.decode_with_indexes(b"abc", [0], cdfs, sizes, offs)       # not a whole number of words
    with pytest.raises(ValueError):
        dec.decode_with_indexes(s[:8], [0] * 50, cdfs, sizes, offs)   # truncated stream is detected, not over-read


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (oracle/build_ref.sh needs /root/reference)")
def test_state_machine_equals_the_reference_header():
    """escape-free coding with one table: every word equals what Rans64EncPut / Rans64EncFlush of the reference's
    vendored rans64.h produce, and Rans64DecInit / DecGet / DecAdvance read our stream back"""
    ref = ctypes.CDLL(REF_SO)
    u32p, i32p = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32)
    ref.ref_rans64_encode.restype = ctypes.c_int64
    ref.ref_rans64_encode.argtypes = [u32p, u32p, ctypes.c_int64, ctypes.c_uint32, u32p, ctypes.c_int64]
    ref.ref_rans64_decode.argtypes = [u32p, ctypes.c_int64, u32p, ctypes.c_int, ctypes.c_uint32, i32p, ctypes.c_int64]
    rng = np.random.default_rng(5)
    for trial in range(20):
        m = int(rng.integers(2, 200))
        p = rng.random(m).astype(np.float32) ** 4
        p = p / p.sum()
        cdf = np.asarray(ans.pmf_to_quantized_cdf(list(p) + [np.float32(1e-5)]), dtype=np.uint32)   # m + 1 symbols
        n = int(rng.integers(1, 5000))
        # skewed draws so that renormalisation happens at irregular intervals
        sym = np.minimum((rng.random(n) ** 3 * m).astype(np.int64), m - 1)
        start = np.ascontiguousarray(cdf[sym]).astype(np.uint32)
        freq = np.ascontiguousarray(cdf[sym + 1] - cdf[sym]).astype(np.uint32)
        out = np.zeros(n + 8, dtype=np.uint32)
        words = ref.ref_rans64_encode(start.ctypes.data_as(u32p), freq.ctypes.data_as(u32p), n, 16,
                                      out.ctypes.data_as(u32p), out.size)
        assert words > 0
        want = out[:words].tobytes()
        mine = ans.RansEncoder().encode_with_indexes(sym.tolist(), [0] * n, [cdf.astype(np.int64).tolist()],
                                                     [len(cdf)], [0])
        assert mine == want
        back = np.zeros(n, dtype=np.int32)
        sw = np.frombuffer(mine, dtype=np.uint32).copy()
        rc = ref.ref_rans64_decode(sw.ctypes.data_as(u32p), sw.size, cdf.ctypes.data_as(u32p), len(cdf) - 1, 16,
                                   back.ctypes.data_as(i32p), n)
        assert rc == 0 and back.tolist() == sym.tolist()
