"""compressai.ans mirror: RansEncoder / RansDecoder / BufferedRansEncoder over the C ABI (csrc/rans.cpp).

The reference imports these three classes from a binary-only pybind11 extension (``compressai/ans.cpython-38-*.so``,
used at entropy_models/entropy_models.py:32-36,200-208,268-276 and models/cnn.py:5,228,263-264,300-318); the
interface below keeps their call signatures -- Python lists (or anything ``numpy.asarray`` accepts) in, ``bytes`` /
``list[int]`` out -- and hands flat int32 arrays to ``icm_rans_*``.  Host-side, like the reference's coder."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib as L

_i32p = C.POINTER(C.c_int32)


def _arr(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.int32).reshape(-1))


class _Tables:
    """cdfs as one [ncdf, stride] int32 block + sizes + offsets (accepts the reference's list-of-lists)"""

    def __init__(self, cdfs, cdfs_sizes, offsets):
        if isinstance(cdfs, np.ndarray) and cdfs.ndim == 2:
            m = np.ascontiguousarray(cdfs.astype(np.int32, copy=False))
        else:
            rows = [np.asarray(r, dtype=np.int32).reshape(-1) for r in cdfs]
            stride = max((len(r) for r in rows), default=0)
            m = np.zeros((len(rows), stride), dtype=np.int32)
            for i, r in enumerate(rows):
                m[i, :len(r)] = r
        self.cdfs, self.sizes, self.offsets = m, _arr(cdfs_sizes), _arr(offsets)
        if m.ndim != 2 or m.shape[0] == 0 or len(self.sizes) != m.shape[0] or len(self.offsets) != m.shape[0]:
            raise ValueError("cdfs, cdfs_sizes and offsets must describe the same number of tables")

    def args(self):
        return (self.cdfs.ctypes.data_as(_i32p), int(self.cdfs.shape[1]), self.sizes.ctypes.data_as(_i32p),
                self.offsets.ctypes.data_as(_i32p), int(self.cdfs.shape[0]))


def _encode(symbols: np.ndarray, indexes: np.ndarray, t: _Tables) -> bytes:
    if symbols.shape != indexes.shape:
        raise ValueError("symbols and indexes must have the same length")
    lib = L.lib()
    n = int(symbols.size)
    sp, ip = symbols.ctypes.data_as(_i32p), indexes.ctypes.data_as(_i32p)
    cap = 4 * n + 64     # a regular symbol costs <= 16 bits; escapes are measured first
    buf = (C.c_uint8 * cap)()
    nb = lib.icm_rans_encode_with_indexes(sp, ip, n, *t.args(), buf, cap)
    if nb < 0:           # escape-heavy input or bad tables: measure, then encode into an exact buffer
        need = lib.icm_rans_encode_with_indexes(sp, ip, n, *t.args(), None, 0)
        if need < 0:
            raise ValueError("rANS encode: invalid symbols / indexes / CDF tables")
        buf = (C.c_uint8 * need)()
        nb = lib.icm_rans_encode_with_indexes(sp, ip, n, *t.args(), buf, need)
        if nb < 0:
            raise ValueError("rANS encode failed")
    return bytes(bytearray(buf)[:nb])


class RansEncoder:
    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes:
        return _encode(_arr(symbols), _arr(indexes), _Tables(cdfs, cdfs_sizes, offsets))


class BufferedRansEncoder:
    """symbols of several calls are concatenated and coded as ONE stream by flush() (cnn.py:228,263-264)"""

    def __init__(self):
        self._sym: List[np.ndarray] = []
        self._idx: List[np.ndarray] = []
        self._tables = None

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets) -> None:
        s, i = _arr(symbols), _arr(indexes)
        if s.shape != i.shape:
            raise ValueError("symbols and indexes must have the same length")
        self._sym.append(s)
        self._idx.append(i)
        self._tables = _Tables(cdfs, cdfs_sizes, offsets)

    def flush(self) -> bytes:
        if self._tables is None:
            raise ValueError("nothing to flush")
        out = _encode(np.concatenate(self._sym), np.concatenate(self._idx), self._tables)
        self._sym, self._idx, self._tables = [], [], None
        return out


class RansDecoder:
    def __init__(self):
        self._h = None
        self._keep = None

    def _close(self):
        if self._h:
            L.lib().icm_rans_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass

    def set_stream(self, stream: bytes) -> None:
        self._close()
        self._keep = (C.c_uint8 * len(stream)).from_buffer_copy(stream)
        self._h = L.lib().icm_rans_decoder_create(self._keep, len(stream))
        if not self._h:
            raise ValueError("rANS decode: not a stream (length must be a multiple of 4, at least 8 bytes)")

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets) -> List[int]:
        if not self._h:
            raise ValueError("set_stream() first")
        return self.decode_stream_np(_arr(indexes), _Tables(cdfs, cdfs_sizes, offsets)).tolist()

    def decode_stream_np(self, idx: np.ndarray, t: "_Tables") -> np.ndarray:
        out = np.empty(idx.size, dtype=np.int32)
        rc = L.lib().icm_rans_decoder_decode(self._h, idx.ctypes.data_as(_i32p), int(idx.size), *t.args(),
                                             out.ctypes.data_as(_i32p))
        if rc:
            raise ValueError("rANS decode: corrupt stream or invalid indexes / CDF tables")
        return out

    def decode_with_indexes(self, stream: bytes, indexes, cdfs, cdfs_sizes, offsets) -> List[int]:
        self.set_stream(stream)
        try:
            return self.decode_stream(indexes, cdfs, cdfs_sizes, offsets)
        finally:
            self._close()


def pmf_to_quantized_cdf(pmf: Sequence[float], precision: int = 16) -> List[int]:
    """compressai._CXX.pmf_to_quantized_cdf (entropy_models.py:13,60-63)"""
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32).reshape(-1))
    out = np.empty(p.size + 1, dtype=np.int32)
    rc = L.lib().icm_pmf_to_quantized_cdf(p.ctypes.data_as(C.POINTER(C.c_float)), int(p.size), int(precision),
                                          out.ctypes.data_as(_i32p))
    if rc:
        raise ValueError("pmf_to_quantized_cdf: invalid pmf (negative / non-finite / empty / all zero)")
    return out.tolist()
