"""Counter-based synthetic-input generator `u(seed, idx)` -> uniform [-1, 1) fp64.

Not reference code: the repo's own generator, chosen so that JS (oracle/gen_golden.js), C
(oracle/nd4_oracle.c), numpy (here) and HIP (csrc/nd4hip_util.hip `fill_uniform`) produce
bit-identical values from 32-bit integer ops only. SURVEY.md §8(d) asks for exactly this: any slice
of any BASELINE config can be regenerated on any device without shipping data.
"""
import numpy as np


def _fmix32(h):
    h = h.astype(np.uint32, copy=True)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def fill_uniform(seed, n, offset=0):
    """n values u(seed, offset..offset+n) as float64 in [-1, 1)."""
    with np.errstate(over="ignore"):
        idx = (np.arange(int(n), dtype=np.uint64) + np.uint64(offset)).astype(np.uint32)
        s = _fmix32(np.array([seed & 0xFFFFFFFF], dtype=np.uint32))[0]
        hi = _fmix32(idx ^ s)
        lo = _fmix32(hi + np.uint32(0x9E3779B9) + idx)
    m = (hi >> np.uint32(5)).astype(np.float64) * 67108864.0 + (lo >> np.uint32(6)).astype(np.float64)
    return m * 2.0 ** -52 - 1.0


def hash_idx(seed, i, mod):
    """Index sampler twin of gen_golden.js `hashIdx` (used to re-derive sampled positions)."""
    with np.errstate(over="ignore"):
        i = np.asarray(i, dtype=np.uint32)
        s = _fmix32(np.array([seed & 0xFFFFFFFF], dtype=np.uint32))[0]
        h = _fmix32(s + i * np.uint32(0x9E3779B1))
    return (h.astype(np.uint64) % np.uint64(mod)).astype(np.int64)


def matrix(seed, *shape, offset=0):
    n = int(np.prod(shape, dtype=np.int64))
    return fill_uniform(seed, n, offset).reshape(shape)
