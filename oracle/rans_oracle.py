"""CPU ORACLE for the entropy-coding back end -- TEST INFRASTRUCTURE ONLY.

Plain-Python (exact integers) restatement of what ``update() / compress() / decompress()`` delegate to in the
reference (compressai/entropy_models/entropy_models.py:60-63,172-290; compressai/models/cnn.py:210-332):

  * the rANS state machine of /root/reference/third_party/ryg_rans/rans64.h (Rans64EncPut :73-90, Rans64EncFlush
    :93-100, Rans64DecInit :102-110, Rans64DecGet :113-116, Rans64DecAdvance :121-138) -- additionally checked
    against that very header through oracle/_ref/librans64_ref.so (oracle/rans64_shim.c, oracle/build_ref.sh);
  * the interface layer of CompressAI 1.1.6dev0 (`compressai/cpp_exts/rans/rans_interface.cpp`,
    `compressai/cpp_exts/ops/ops.cpp`): NOT in the reference tree (only cp38 binaries, which are never run), so its
    published behaviour is restated here: 16-bit precision; value = symbol - offset; values outside [0, size-2) go
    through the last bin and are followed by 4-bit bypass groups (count nibbles, then the zig-zag payload
    raw = -2v-1 for v < 0, 2(v - max) above); symbols are pushed in reverse; pmf_to_quantized_cdf steals counts from
    the least frequent symbol to remove zero-width entries.

Parity status ("parity unpinned" for the interface layer): the reference holds no golden bitstreams
and its coder binaries may not be executed; what IS pinned: the state machine against rans64.h, round-trip identity,
and ``actual bpp ~ estimated bpp`` on the model.  Only tests/ import this module."""
from __future__ import annotations

import math
from typing import List, Sequence

PRECISION = 16
BYPASS_BITS = 4
BYPASS_MAX = (1 << BYPASS_BITS) - 1
RANS_L = 1 << 31
M32 = (1 << 32) - 1


def pmf_to_quantized_cdf(pmf: Sequence[float], precision: int = PRECISION) -> List[int]:
    import numpy as np
    one = 1 << precision
    p32 = np.asarray(pmf, dtype=np.float32)
    if not (np.isfinite(p32).all() and (p32 >= 0).all()):
        raise ValueError("invalid pmf")
    # std::round on a float product: round half away from zero of the f32 value p * 2^precision
    prod = (p32 * np.float32(one)).astype(np.float64)
    counts = [0] + [int(math.floor(v + 0.5)) for v in prod]
    total = sum(counts)
    if total == 0:
        raise ValueError("empty pmf")
    cdf, run = [], 0
    for c in counts:
        run += (one * c) // total
        cdf.append(run)
    cdf[-1] = one
    n = len(cdf) - 1
    for i in range(n):
        if cdf[i] != cdf[i + 1]:
            continue
        best, donor = None, -1
        for j in range(n):
            f = cdf[j + 1] - cdf[j]
            if f > 1 and (best is None or f < best):
                best, donor = f, j
        if donor < 0:
            raise ValueError("more symbols than counts")
        if donor < i:
            for j in range(donor + 1, i + 1):
                cdf[j] -= 1
        else:
            for j in range(i + 1, donor + 1):
                cdf[j] += 1
    return cdf


class _Enc:
    def __init__(self):
        self.x = RANS_L
        self.words: List[int] = []

    def put(self, start, freq, bits):        # rans64.h:73-90
        x_max = ((RANS_L >> bits) << 32) * freq
        if self.x >= x_max:
            self.words.append(self.x & M32)
            self.x >>= 32
        self.x = ((self.x // freq) << bits) + (self.x % freq) + start

    def put_bits(self, val, nbits):
        freq = 1 << (PRECISION - nbits)
        x_max = ((RANS_L >> PRECISION) << 32) * freq
        if self.x >= x_max:
            self.words.append(self.x & M32)
            self.x >>= 32
        self.x = (self.x << nbits) | val

    def finish(self) -> bytes:               # rans64.h:93-100 (stream written backwards)
        ws = [self.x & M32, self.x >> 32] + self.words[::-1]
        return b"".join(int(w).to_bytes(4, "little") for w in ws)


def encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets) -> bytes:
    items = []
    for s, idx in zip(symbols, indexes):
        cdf = cdfs[idx]
        mx = cdf_sizes[idx] - 2
        v = int(s) - offsets[idx]
        raw = 0
        if v < 0:
            raw, v = -2 * v - 1, mx
        elif v >= mx:
            raw, v = 2 * (v - mx), mx
        items.append((int(cdf[v]), int(cdf[v + 1]) - int(cdf[v]), False))
        if v == mx:
            g = 0
            while (raw >> (g * BYPASS_BITS)) != 0:
                g += 1
            c = g
            while c >= BYPASS_MAX:
                items.append((BYPASS_MAX, 1, True))
                c -= BYPASS_MAX
            items.append((c, 1, True))
            for j in range(g):
                items.append(((raw >> (j * BYPASS_BITS)) & BYPASS_MAX, 1, True))
    e = _Enc()
    for start, width, raw in reversed(items):
        if raw:
            e.put_bits(start, BYPASS_BITS)
        else:
            e.put(start, width, PRECISION)
    return e.finish()


class Decoder:
    def __init__(self, stream: bytes):
        self.w = [int.from_bytes(stream[i:i + 4], "little") for i in range(0, len(stream), 4)]
        self.x = self.w[0] | (self.w[1] << 32)        # rans64.h:102-110
        self.p = 2

    def _renorm(self):
        if self.x < RANS_L:
            self.x = (self.x << 32) | self.w[self.p]
            self.p += 1

    def _bits(self, n):
        v = self.x & ((1 << n) - 1)
        self.x >>= n
        self._renorm()
        return v

    def decode(self, indexes, cdfs, cdf_sizes, offsets) -> List[int]:
        out = []
        for idx in indexes:
            cdf = cdfs[idx]
            size = cdf_sizes[idx]
            mx = size - 2
            cum = self.x & ((1 << PRECISION) - 1)      # rans64.h:113-116
            s = 0
            while s + 1 < size and cdf[s + 1] <= cum:
                s += 1
            start, freq = int(cdf[s]), int(cdf[s + 1]) - int(cdf[s])
            self.x = freq * (self.x >> PRECISION) + (self.x & ((1 << PRECISION) - 1)) - start   # rans64.h:121-138
            self._renorm()
            v = s
            if s == mx:
                c = self._bits(BYPASS_BITS)
                g = c
                while c == BYPASS_MAX:
                    c = self._bits(BYPASS_BITS)
                    g += c
                raw = 0
                for j in range(g):
                    raw |= self._bits(BYPASS_BITS) << (j * BYPASS_BITS)
                v = raw >> 1
                v = -v - 1 if raw & 1 else v + mx
            out.append(v + offsets[idx])
        return out


def decode_with_indexes(stream, indexes, cdfs, cdf_sizes, offsets) -> List[int]:
    return Decoder(stream).decode(indexes, cdfs, cdf_sizes, offsets)
