"""Host-side evaluation of the counter hash of csrc/vf_common.h (``vf_dropout_hash``) for the few per-scene random numbers the
training step draws on the host (random pose multiplier); the dropout masks themselves are evaluated inside the kernels."""
import numpy as np

_M = 0xFFFFFFFF


def dropout_hash(seed, site, idx):
    """uint32 hash of (seed, site, 64-bit element index); ``idx`` array-like -> uint32 array"""
    out = []
    for i in np.asarray(idx, dtype=np.uint64).reshape(-1).tolist():
        h = (int(seed) ^ ((int(site) * 0x9E3779B9) & _M)) & _M
        h ^= i & _M
        h = (h * 0x85EBCA6B) & _M
        h ^= h >> 13
        h = (h + (((i >> 32) * 0xC2B2AE35) & _M) + 0x27D4EB2F) & _M
        h ^= h >> 16
        h = (h * 0x165667B1) & _M
        h ^= h >> 15
        h = (h * 0xD3A2646C) & _M
        h ^= h >> 16
        out.append(h)
    return np.asarray(out, dtype=np.uint32)
