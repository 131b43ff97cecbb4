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


class TorchCpuRandom(NumpyLegacyRandom):
    """``torch.manual_seed(seed); torch.randn(n)`` on the CPU, restated from
    swift/StableDiffusion/pipeline/TorchRandomSource.swift:116-150 (which restates ATen's
    DistributionTemplates.h): n >= 16 fills an array of 24-bit uniforms, then Box-Muller over blocks of 16
    (first 8 = u1 -> radius, last 8 = u2 -> angle), the ragged tail recomputed from fresh 53-bit doubles over the
    LAST 16; n < 16 is scalar Box-Muller on 53-bit doubles with the sine cached.  Double arithmetic like the
    Swift code (torch itself evaluates the float32 path with vectorised float math, so it agrees to ~1e-6, not
    bit for bit - tests/test_oracle.py pins that)."""

    def next_u64(self):
        hi = self._next_u32()
        lo = self._next_u32()
        return (hi << 32) | lo

    def next_double53(self):
        return (self.next_u64() & 9007199254740991) * (1.0 / 9007199254740992.0)

    def next_float24(self):
        return (self._next_u32() & 16777215) * (1.0 / 16777216.0)

    def _gauss(self):
        if self.cached is not None:
            g, self.cached = self.cached, None
            return g
        u1 = self.next_double53()
        u2 = 1 - self.next_double53()
        radius = math.sqrt(-2.0 * math.log(u2))
        theta = 2.0 * math.pi * u1
        self.cached = radius * math.sin(theta)
        return radius * math.cos(theta)

    def randn(self, n):
        if n < 16:
            return [self._gauss() for _ in range(n)]
        data = [self.next_float24() for _ in range(n)]

        def fill16(i):
            for j in range(8):
                u1 = 1 - data[i + j]
                u2 = data[i + j + 8]
                radius = math.sqrt(-2.0 * math.log(u1))
                theta = 2.0 * math.pi * u2
                data[i + j] = radius * math.cos(theta)
                data[i + j + 8] = radius * math.sin(theta)

        for i in range(0, n - 15, 16):
            fill16(i)
        if n % 16:
            # torch redraws the last 16 as 24-bit floats; the Swift code uses 53-bit doubles here (:135-137), which
            # torch.randn(float32) does not - the two only differ off the path (latent counts are multiples of 16)
            for i in range(n - 16, n):
                data[i] = self.next_float24()
            fill16(n - 16)
        return data


def philox4x32_10(counter, key):
    """Philox4x32-10 block function (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator behind
    torch's CUDA RNG and NvRandomSource.swift:25-63): 4 x 32-bit counter, 2 x 32-bit key -> 4 x 32-bit output.  Pinned
    against the published known-answer vectors of the Random123 distribution (tests/test_oracle.py)."""
    m0, m1, w0, w1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c, k = [int(x) & M32 for x in counter], [int(x) & M32 for x in key]
    for r in range(10):
        v1 = c[0] * m0
        v2 = c[2] * m1
        c = [((v2 >> 32) ^ c[1] ^ k[0]) & M32, v2 & M32, ((v1 >> 32) ^ c[3] ^ k[1]) & M32, v1 & M32]
        if r < 9:
            k = [(k[0] + w0) & M32, (k[1] + w1) & M32]
    return c


def philox_randn(seed, offset, n):
    """torch.randn on a CUDA device as swift/StableDiffusion/pipeline/NvRandomSource.swift:25-80 restates it:
    Philox4x32-10 with counter (offset, 0, i, 0) and key (seed lo, seed hi) per element i, Box-Muller on the first
    two output words.  ``offset`` counts previous calls (the Swift struct increments it per array).  The block function is
    pinned against the published known-answer vectors; the counter layout and the Box-Muller step follow the Swift code and
    have no golden in the reference (no CUDA device here): that part stays PARITY UNPINNED."""
    out = []
    for i in range(n):
        c = philox4x32_10([offset & M32, 0, i & M32, 0], [seed & M32, (seed >> 32) & M32])
        u = c[0] / 4294967296.0 + (1.0 / 8589934592.0)
        v = c[1] * (math.pi / 2147483648.0) + (math.pi / 4294967296.0)
        out.append(math.sqrt(-2.0 * math.log(u)) * math.sin(v))
    return out
