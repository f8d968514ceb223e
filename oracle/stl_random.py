"""CPU ORACLE (test infrastructure): the reference's default start vector.

Reference: LanczosBase::setInitialVector()  lanczos.hpp:214-218 ->
makeRandomVector  :124-135 -> VectorDistribution::operator()  random.hpp:89-101
with NormalDistributionGen  util.hpp:132-148 (complex: ComplexNormalDistribution
util.hpp:76-97, real part drawn first).  The engine is a default-constructed
std::mt19937 (seed 5489); the distribution is the HOST STL's
std::normal_distribution<double>, i.e. on this toolchain libstdc++ (GCC 11):
Marsaglia polar method on generate_canonical<double,53> (two 32-bit draws per
uniform), returning y*mult first and caching x*mult for the next call.
The restatement is checked against g++'s real <random> in
tests/test_oracle_golden.py (a tiny program compiled at test time).
"""
from __future__ import annotations

import math

import numpy as np


class _LibstdcxxNormal:
    def __init__(self, seed: int = 5489):
        self._bg = np.random.MT19937()
        self._bg._legacy_seeding(seed)  # init_genrand(seed) == std::mt19937(seed)
        self._buf = np.zeros(0, dtype=np.uint64)
        self._pos = 0
        self._saved = None

    def _raw(self) -> int:
        if self._pos >= self._buf.size:
            self._buf = self._bg.random_raw(4096)
            self._pos = 0
        r = int(self._buf[self._pos])
        self._pos += 1
        return r

    def _canonical(self) -> float:
        # std::generate_canonical<double,53>: sum = x0 + x1 * 2^32 in double, / 2^64
        x0 = float(self._raw())
        x1 = float(self._raw())
        ret = (x0 + x1 * 4294967296.0) / 18446744073709551616.0
        if ret >= 1.0:
            ret = math.nextafter(1.0, 0.0)
        return ret

    def __call__(self) -> float:
        if self._saved is not None:
            r = self._saved
            self._saved = None
            return r
        while True:
            x = 2.0 * self._canonical() - 1.0
            y = 2.0 * self._canonical() - 1.0
            r2 = x * x + y * y
            if not (r2 > 1.0 or r2 == 0.0):
                break
        mult = math.sqrt(-2.0 * math.log(r2) / r2)
        self._saved = x * mult
        return y * mult


def libstdcxx_normal_vector(n: int, dtype=np.float64, seed: int = 5489):
    """n draws in index order (NOT normalised)."""
    g = _LibstdcxxNormal(seed)
    dtype = np.dtype(dtype)
    if dtype.kind == "c":
        out = np.empty(n, dtype=np.complex128)
        for i in range(n):
            re = g()
            im = g()
            out[i] = complex(re, im)
        return out.astype(dtype)
    out = np.empty(n, dtype=np.float64)
    for i in range(n):
        out[i] = g()
    return out.astype(dtype)
