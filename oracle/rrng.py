"""R's default random number stream (Mersenne-Twister + Inversion), restated.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference's only
known-answer vectors are the R snippets of /root/reference/README.md
(:47-53 Lasso/Enet/LAD data, :166-174 Basis-Pursuit data); they are stated
as `set.seed(123); runif(); rnorm(); sample()` calls, so regenerating the
inputs needs R's generator.  R itself is a third-party dependency that is
absent from /root/reference and from this image; what follows restates the
published algorithm of R >= 1.7 (src/main/RNG.c: `set.seed` scrambling,
MT19937 `MT_genrand`, `fixup`; src/nmath/snorm.c INVERSION branch;
src/main/unique.c / random.c `sample` for R < 3.6, "Rounding" sample.kind).

Pinned by: set.seed(123); runif(3) = 0.2875775 0.7883051 0.4089769 and
set.seed(123); rnorm(3) = -0.56047565 -0.23017749 1.55870831 (R documentation
examples reproduced in SURVEY.md Appendix A), asserted in
tests/test_oracle_readme.py.
"""
import numpy as np
from scipy.special import ndtri

_N, _M = 624, 397
_I2_32M1 = 2.328306437080797e-10  # 1/(2^32 - 1), fixup() bounds


class RRandom:
    """`set.seed(seed)` for RNGkind("Mersenne-Twister", "Inversion", "Rounding")."""

    def __init__(self, seed):
        s = np.uint32(seed & 0xFFFFFFFF)
        with np.errstate(over="ignore"):
            for _ in range(50):                       # RNG.c: initial scrambling
                s = np.uint32(np.uint32(69069) * s + np.uint32(1))
            iseed = np.empty(_N + 1, dtype=np.uint32)
            for j in range(_N + 1):                   # RNG_Init: fill i_seed[0..624]
                s = np.uint32(np.uint32(69069) * s + np.uint32(1))
                iseed[j] = s
        # FixupSeeds: i_seed[0] <- mti = 624; mt = i_seed[1:]
        self.mt = iseed[1:].astype(np.uint64)
        self.mti = _N

    # -- MT19937 block regeneration, vectorised in the standard three segments
    def _reload(self):
        mt = self.mt
        UP, LO = np.uint64(0x80000000), np.uint64(0x7FFFFFFF)
        MAG = np.uint64(0x9908B0DF)

        def tw(u, v):
            y = (u & UP) | (v & LO)
            return (y >> np.uint64(1)) ^ np.where((y & np.uint64(1)) != 0, MAG, np.uint64(0))

        # The sequential recurrence mt[kk] = mt[(kk+M) % N] ^ tw(mt[kk], mt[kk+1]) reads
        # only already-final words when done in these four slices (227 = N - M).
        a = _N - _M
        mt[0:a] = mt[_M:_N] ^ tw(mt[0:a], mt[1:a + 1])
        mt[a:2 * a] = mt[0:a] ^ tw(mt[a:2 * a], mt[a + 1:2 * a + 1])
        mt[2 * a:_N - 1] = mt[a:_N - 1 - a] ^ tw(mt[2 * a:_N - 1], mt[2 * a + 1:_N])
        mt[_N - 1] = mt[_M - 1] ^ tw(mt[_N - 1], mt[0])
        self.mti = 0

    def _int32_block(self, k):
        """k tempered 32-bit outputs."""
        out = np.empty(k, dtype=np.uint64)
        done = 0
        while done < k:
            if self.mti >= _N:
                self._reload()
            take = min(k - done, _N - self.mti)
            out[done:done + take] = self.mt[self.mti:self.mti + take]
            self.mti += take
            done += take
        y = out
        y = y ^ (y >> np.uint64(11))
        y = y ^ ((y << np.uint64(7)) & np.uint64(0x9D2C5680))
        y = y ^ ((y << np.uint64(15)) & np.uint64(0xEFC60000))
        y = y ^ (y >> np.uint64(18))
        return y & np.uint64(0xFFFFFFFF)

    def unif_rand(self, k):
        """k draws of unif_rand(): MT_genrand() * 2.3283064365386963e-10, fixup()."""
        u = self._int32_block(k).astype(np.float64) * 2.3283064365386963e-10
        u = np.where(u <= 0.0, 0.5 * _I2_32M1, u)
        u = np.where(1.0 - u <= 0.0, 1.0 - 0.5 * _I2_32M1, u)
        return u

    def runif(self, k):
        return self.unif_rand(k)

    def rnorm(self, k, mean=0.0, sd=1.0):
        """snorm.c INVERSION: u = floor(2^27 u1) + u2; qnorm(u / 2^27)."""
        u = self.unif_rand(2 * k)
        big = 134217728.0
        v = np.floor(big * u[0::2]) + u[1::2]
        return mean + sd * ndtri(v / big)

    def sample_perm(self, n):
        """sample.int(n) under R < 3.6 (0-based): j = floor(k * unif_rand())."""
        x = np.arange(n)
        out = np.empty(n, dtype=np.int64)
        u = self.unif_rand(n)
        k = n
        for i in range(n):
            j = int(k * u[i])
            out[i] = x[j]
            k -= 1
            x[j] = x[k]
        return out
