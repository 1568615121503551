"""Host-side evaluation of the counter hashes of csrc/vf_common.h: ``vf_dropout_hash`` for the few per-scene random numbers the
training step draws on the host (random pose multiplier), and the dropout-mask definition itself (``vf_dropout_word`` /
``vf_dropout_keep``) for host-side checks; the masks of a training step are evaluated inside the kernels.

Mask definition (round 4): elements come in GROUPS of four that share one 32-bit word,
    word(seed, site, g) = lowbias32(lo32(g) ^ vf_dropout_hash(seed, site, hi32(g)))        g = 64-bit group index
    keep(element j of group g) = rotl32(word, 8 j) >= floor(rate * 2^32)
so every element's keep probability is exactly 1 - floor(rate 2^32) / 2^32 (a rotation of a uniform word is uniform) and a lane that
holds the four elements of a group pays for one hash.  Group index and position of an element:
    [M][N] activations (embedding / residual / MLP sites):  g = (m >> 2) * N + n,  j = m & 3    (four consecutive ROWS of a column:
        what a lane of the GEMM epilogue holds);
    attention weights (b, h, q, k) of a T-token sequence:  g = ((b H + h) << 32) | (q * ceil(T / 4) + (k >> 2)),  j = k & 3."""
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


def lowbias32(x):
    """Chris Wellons' 2-multiply integer hash (bias 0.17): uint32 array -> uint32 array"""
    x = np.asarray(x, dtype=np.uint64) & np.uint64(_M)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(_M)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(_M)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def dropout_word(seed, site, group):
    """the 32-bit word shared by the four elements of 64-bit group index ``group`` (array-like)"""
    g = np.asarray(group, dtype=np.uint64)
    hi = (g >> np.uint64(32)).reshape(-1)
    keys = {int(h): int(dropout_hash(seed, site, [h])[0]) for h in np.unique(hi).tolist()}
    key = np.asarray([keys[int(h)] for h in hi.tolist()], dtype=np.uint64).reshape(g.shape)
    return lowbias32((g & np.uint64(_M)) ^ key)


def dropout_keep(seed, site, group, sub, rate):
    """bool keep mask: rotl32(word(group), 8 * sub) >= floor(rate * 2^32)"""
    w = dropout_word(seed, site, group).astype(np.uint64)
    sh = (np.asarray(sub, dtype=np.uint64) & np.uint64(3)) * np.uint64(8)
    rot = ((w << sh) | (w >> (np.uint64(32) - sh))) & np.uint64(_M)
    rot = np.where(sh == 0, w, rot)
    return rot >= np.uint64(int(rate * 4294967296.0))


def elem_group(m, n, N):
    """(group, sub) of element (m, n) of an [M][N] activation"""
    m = np.asarray(m, dtype=np.uint64)
    g = (m >> np.uint64(2)) * np.uint64(N) + np.asarray(n, dtype=np.uint64)
    return g, np.broadcast_to(m & np.uint64(3), g.shape)


def attn_group(plane, q, k, T):
    """(group, sub) of attention weight (plane = b H + h, q, k) of a T-token sequence"""
    k = np.asarray(k, dtype=np.uint64)
    stride = np.uint64((T + 3) // 4)
    g = (np.asarray(plane, dtype=np.uint64) << np.uint64(32)) | (np.asarray(q, dtype=np.uint64) * stride + (k >> np.uint64(2)))
    return g, np.broadcast_to(k & np.uint64(3), g.shape)
