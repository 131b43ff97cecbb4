"""Oracle (test infrastructure): numpy's legacy ``np.random.seed`` / ``randn`` stream.

The pipeline draws its initial latents with ``np.random.seed(seed)`` +
``np.random.randn(...)`` (python_coreml_stable_diffusion/pipeline.py:331, :726).  The
algorithm is restated from the reference's own bit-exact re-implementation,
swift/StableDiffusion/pipeline/NumPyRandomSource.swift:28-102:
  * MT19937 seeded by the Knuth LCG 1812433253                       (:28-37)
  * 53-bit doubles from two draws: (a>>5)*2^26 + (b>>6)) / 2^53      (:77-81)
  * Marsaglia polar Box-Muller, caching the second deviate           (:84-102)
Golden vector pinned by the reference's unit test
swift/StableDiffusionTests/StableDiffusionTests.swift:52-62 (seed 12345, last 5 of 10000).
Pure Python on purpose (small cases only); numpy itself is the second witness.
"""
import math

M32 = 0xFFFFFFFF


class NumpyLegacyRandom:
    def __init__(self, seed):
        s = seed & M32
        key = []
        for i in range(624):
            key.append(s)
            s = (1812433253 * (s ^ (s >> 30)) + i + 1) & M32
        self.key = key
        self.pos = 624
        self.cached = None

    def _next_u32(self):
        n, m = 624, 397
        k = self.key
        if self.pos == n:
            for i in range(n):
                y = (k[i] & 0x80000000) | (k[(i + 1) % n] & 0x7FFFFFFF)
                k[i] = k[(i + m) % n] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.pos = 0
        y = k[self.pos]
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & M32

    def next_double(self):
        a = self._next_u32() >> 5
        b = self._next_u32() >> 6
        return (a * 67108864.0 + b) / 9007199254740992.0

    def next_gauss(self):
        if self.cached is not None:
            g, self.cached = self.cached, None
            return g
        while True:
            x1 = 2.0 * self.next_double() - 1.0
            x2 = 2.0 * self.next_double() - 1.0
            r2 = x1 * x1 + x2 * x2
            if 0.0 < r2 < 1.0:
                break
        f = math.sqrt(-2.0 * math.log(r2) / r2)
        self.cached = f * x1
        return f * x2

    def randn(self, n):
        return [self.next_gauss() for _ in range(n)]


# swift/StableDiffusionTests/StableDiffusionTests.swift:52-62
GOLDEN_SEED = 12345
GOLDEN_COUNT = 10000
GOLDEN_LAST5 = [-0.86285345, 2.15229409, -0.00670556, -1.21472309, 0.65498866]
