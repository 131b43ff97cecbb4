"""Round-4 GPU tests: kernels added this round, called through the C ABI, against the oracle / torch on seeded inputs."""
import numpy as np
import pytest

from oracle import attention_ref
from python_hip_stable_diffusion import _lib
from test_ops_gpu import close, h16

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- attention8.hip (d = 64, S_k % 64 == 0)
@pytest.mark.parametrize("impl", ["ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"])
@pytest.mark.parametrize("shape", [(1, 1, 1, 64), (1, 3, 33, 64), (2, 2, 96, 448), (1, 2, 300, 192), (1, 1, 512, 1024), (2, 10, 1024, 1024)],
                         ids=lambda s: "x".join(map(str, s)))
def test_attention8_against_oracle_and_the_general_kernels(impl, shape):
    """The software-pipelined kernel (variant 0: the default for d = 64 and whole 64-key tiles) and the general kernels
    (variant 1) against the oracle; ragged query counts, one key tile, 4- and 8-wave workgroups."""
    b, h, sq, sk = shape
    rs = np.random.RandomState(abs(hash(shape)) % (2 ** 31))
    q, k, v = (h16(rs.randn(b, h * 64, 1, n)) for n in (sq, sk, sk))
    if impl == "SPLIT_EINSUM_V2" and sq >= 512 and sq % 512:
        pytest.skip("V2 rejects S_q % 512 != 0")
    ref = attention_ref.original(q.astype(np.float32), k.astype(np.float32), v.astype(np.float32), h, 64)
    new, _ = _lib.attention(impl, q, k, v, h, 64, variant=0)
    old, _ = _lib.attention(impl, q, k, v, h, 64, variant=1)
    close(new, ref, f"attention8 {impl} {shape}")
    close(old, ref, f"general kernel {impl} {shape}")
    close(new, old.astype(np.float32), f"attention8 vs general {impl} {shape}", min_psnr=55)


def test_attention8_large_scores_do_not_overflow_fp16_probabilities():
    """Lazy running max: P may reach 2^8 before a refresh; scores far above and far below the first tile's."""
    b, h, sq, sk = 1, 2, 128, 512
    rs = np.random.RandomState(3)
    q, k, v = (rs.randn(b, h * 64, 1, n).astype(np.float32) for n in (sq, sk, sk))
    k[:, :, :, :64] *= 0.05           # tiny first tile: the running max starts low
    k[:, :, :, 200:264] *= 9.0        # a tile far above it
    q[:, :64, :, 5] *= 20.0           # one query of head 0 with huge logits
    q, k, v = h16(q), h16(k), h16(v)
    ref = attention_ref.original(q.astype(np.float32), k.astype(np.float32), v.astype(np.float32), h, 64)
    for impl in ("ORIGINAL", "SPLIT_EINSUM"):
        out, _ = _lib.attention(impl, q, k, v, h, 64)
        close(out, ref, f"attention8 {impl} extreme scores")
