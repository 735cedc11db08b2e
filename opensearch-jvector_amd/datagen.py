"""Bit-reproducible synthetic inputs.

* ``java_random_floats`` restates ``java.util.Random(seed).nextFloat()`` — the generator behind every
  random input of the reference's tests and JMH benchmark (src/testFixtures/.../TestUtils.java:108-124,
  benchmark-jmh/.../FormatBenchmarkQueryWithRandomVectors.java:77-86) — so KA8/KA12/KA13/KA15 inputs can
  be regenerated exactly without a JVM.
* ``splitmix_uniform`` is the counter-based generator for the large configs: element
  (i, j) of stream ``seed`` is splitmix64(seed, i*d + j) -> top 24 bits -> [0, 1).
"""
from __future__ import annotations

import numpy as np

_MASK48 = (1 << 48) - 1
_MULT = 0x5DEECE66D


def java_random_floats(seed: int, count: int) -> np.ndarray:
    """``count`` successive ``new java.util.Random(seed).nextFloat()`` values (float32)."""
    out = np.empty(count, dtype=np.float32)
    s = (seed ^ _MULT) & _MASK48
    # jump-free scalar loop is too slow in Python for millions of draws; vectorise with LCG powers
    # a_k = MULT^k, c_k = ADD * (MULT^(k-1) + ... + 1) (mod 2^48), computed blockwise.
    block = 1 << 16
    a = np.empty(block, dtype=object)
    c = np.empty(block, dtype=object)
    ak, ck = 1, 0
    for k in range(block):
        ak = (ak * _MULT) & _MASK48
        ck = (ck * _MULT + 0xB) & _MASK48
        a[k] = ak
        c[k] = ck
    # split 48-bit arithmetic into exact uint64 pieces: x*a mod 2^48 with a,x < 2^48
    a_lo = np.array([int(v) & 0xFFFFFF for v in a], dtype=np.uint64)
    a_hi = np.array([int(v) >> 24 for v in a], dtype=np.uint64)
    c_np = np.array([int(v) for v in c], dtype=np.uint64)
    pos = 0
    m48 = np.uint64(_MASK48)
    while pos < count:
        nb = min(block, count - pos)
        s_lo = np.uint64(s & 0xFFFFFF)
        s_hi = np.uint64(s >> 24)
        # (s_hi*2^24 + s_lo) * (a_hi*2^24 + a_lo) mod 2^48
        lo = s_lo * a_lo[:nb]
        mid = (s_lo * a_hi[:nb] + s_hi * a_lo[:nb]) & np.uint64(0xFFFFFF)
        states = (lo + (mid << np.uint64(24)) + c_np[:nb]) & m48
        out[pos:pos + nb] = (states >> np.uint64(24)).astype(np.float32) / np.float32(1 << 24)
        s = int(states[nb - 1])
        pos += nb
    return out


def java_random_vectors(seed: int, n: int, d: int) -> np.ndarray:
    return java_random_floats(seed, n * d).reshape(n, d)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def splitmix_uniform(seed: int, n: int, d: int, row_offset: int = 0) -> np.ndarray:
    """uniform [0,1) float32 matrix; element (i,j) depends only on (seed, (row_offset+i)*d + j)."""
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64)[:, None] + np.uint64(row_offset)) * np.uint64(d) + np.arange(d, dtype=np.uint64)[None, :]
        key = idx + np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
        z = _splitmix64(key)
    return ((z >> np.uint64(40)).astype(np.float32) / np.float32(1 << 24)).astype(np.float32)


def gaussian_mixture(seed: int, n: int, d: int, centres: int = 4096, sigma: float = 0.05,
                     centre_seed: int = 44, row_offset: int = 0) -> np.ndarray:
    """ANN-realistic distribution B of SURVEY §8(d): centres ~ U[0,1)^d, assignment i mod centres,
    Box-Muller noise from two splitmix streams."""
    c = splitmix_uniform(centre_seed, centres, d)
    u1 = splitmix_uniform(seed, n, d, row_offset)
    u2 = splitmix_uniform(seed + 1000003, n, d, row_offset)
    g = np.sqrt(-2.0 * np.log(np.maximum(u1, np.float32(1e-12)))) * np.cos(np.float32(2 * np.pi) * u2)
    assign = (np.arange(n) + row_offset) % centres
    return (c[assign] + np.float32(sigma) * g.astype(np.float32)).astype(np.float32)


def l2_normalize(x: np.ndarray) -> np.ndarray:
    nrm = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
    return (x / np.maximum(nrm, 1e-30)).astype(np.float32)
