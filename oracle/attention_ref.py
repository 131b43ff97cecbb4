"""Oracle (test infrastructure): the three attention formulations.

Restates ``python_coreml_stable_diffusion/attention.py``:
  * ``original``        attention.py:147-168  (b,h,q,k) scores, softmax over k (dim=3)
  * ``split_einsum``    attention.py:24-72    per-head (b,k,1,q) scores, softmax over dim=1
  * ``split_einsum_v2`` attention.py:77-144   as split_einsum, query axis cut in 512-chunks,
                                              falls back to split_einsum for S_q < 512 and
                                              silently drops S_q % 512 queries (we raise).
Inputs follow the reference layout BC1S: q (B, h*d, 1, S_q); k, v (B, h*d, 1, S_k).
Scale d**-0.5 is applied to the *scores* (attention.py:49,123,159).  ``mask`` is always
None on the UNet path (unet.py:587-589) and is not modelled.

numpy inputs are computed in float64 (tighter than any fp32 run) and returned as numpy;
torch inputs are computed in their own dtype (used inside ``unet_ref`` at full size).

Pinned by ``oracle/pin_against_reference.py`` against the unmodified reference module
(imports torch only) and by ``tests/golden/attention_golden.npz``.
"""
import numpy as np
import torch

CHUNK_SIZE = 512  # attention.py:75


def _wrap(fn):
    def run(q, k, v, heads, dim_head):
        if isinstance(q, np.ndarray):
            args = [torch.from_numpy(np.asarray(a, np.float64)) for a in (q, k, v)]
            return fn(*args, heads, dim_head).numpy()
        return fn(q, k, v, heads, dim_head)
    run.__name__ = fn.__name__
    run.__wrapped__ = fn
    run.__doc__ = fn.__doc__
    return run


@_wrap
def original(q, k, v, heads, dim_head):
    """attention.py:147-168."""
    bs = q.shape[0]
    mh_q = q.reshape(bs, heads, dim_head, -1)           # (b,h,c,q)
    mh_k = k.reshape(bs, heads, dim_head, -1)           # (b,h,c,k)
    mh_v = v.reshape(bs, heads, dim_head, -1)
    w = torch.einsum("bhcq,bhck->bhqk", mh_q, mh_k) * dim_head ** -0.5
    w = w.softmax(dim=3)
    o = torch.einsum("bhqk,bhck->bhcq", w, mh_v)
    return o.reshape(bs, heads * dim_head, 1, -1)


@_wrap
def split_einsum(q, k, v, heads, dim_head):
    """attention.py:24-72: one (B, S_k, 1, S_q) score tile per head, softmax over dim=1."""
    outs = []
    kt = k.permute(0, 3, 2, 1)                           # (b, S_k, 1, C)
    for h in range(heads):
        sl = slice(h * dim_head, (h + 1) * dim_head)
        qi, ki, vi = q[:, sl], kt[:, :, :, sl], v[:, sl]
        aw = torch.einsum("bchq,bkhc->bkhq", qi, ki) * dim_head ** -0.5
        aw = aw.softmax(dim=1)
        outs.append(torch.einsum("bkhq,bchk->bchq", aw, vi))
    return torch.cat(outs, dim=1)


@_wrap
def split_einsum_v2(q, k, v, heads, dim_head):
    """attention.py:77-144.  The reference computes ``S_q // 512`` chunks and silently drops
    the tail (attention.py:86); that is a shape bug for S_q % 512 != 0, so the oracle (and the
    HIP path) reject it loudly instead of replicating it."""
    s_q = q.shape[3]
    n_chunks = s_q // CHUNK_SIZE
    if n_chunks == 0:                                    # attention.py:88-92
        return split_einsum.__wrapped__(q, k, v, heads, dim_head)
    if s_q % CHUNK_SIZE:
        raise ValueError(f"SPLIT_EINSUM_V2 needs S_q % {CHUNK_SIZE} == 0 (got {s_q})")
    parts = [split_einsum.__wrapped__(q[..., c * CHUNK_SIZE:(c + 1) * CHUNK_SIZE], k, v, heads, dim_head)
             for c in range(n_chunks)]
    return torch.cat(parts, dim=3)


IMPLS = {"ORIGINAL": original, "SPLIT_EINSUM": split_einsum, "SPLIT_EINSUM_V2": split_einsum_v2}
