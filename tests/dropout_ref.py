"""numpy restatement of the counter-based dropout masks of csn_amd/csrc/csn_common.h — test infrastructure.
A keep decision is a pure function of (seed, position); the kernels regenerate it in the backward pass."""
import numpy as np


def _mix32(h):
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85ebca6b)) & np.uint64(0xffffffff)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xc2b2ae35)) & np.uint64(0xffffffff)
    h ^= h >> np.uint64(16)
    return h


def keep_mask(idx, seed: int, p: float) -> np.ndarray:
    """idx: integer array of element indices (< 2**63).  Returns a bool array: True = kept."""
    idx = np.asarray(idx, dtype=np.uint64)
    lo, hi = idx & np.uint64(0xffffffff), idx >> np.uint64(32)
    s0, s1 = np.uint64(seed & 0xffffffff), np.uint64((seed >> 32) & 0xffffffff)
    h = _mix32(lo ^ s0)
    h = _mix32((h + (hi ^ s1) + np.uint64(0x9e3779b9)) & np.uint64(0xffffffff))
    thr = np.uint64(int(np.float32(p) * np.float32(16777216.0)))
    return (h >> np.uint64(8)) >= thr


def _hash2(idx, seed: int):
    """the two-round hash of keep_mask, as a 32-bit value"""
    idx = np.asarray(idx, dtype=np.uint64)
    lo, hi = idx & np.uint64(0xffffffff), idx >> np.uint64(32)
    s0, s1 = np.uint64(seed & 0xffffffff), np.uint64((seed >> 32) & 0xffffffff)
    h = _mix32(lo ^ s0)
    return _mix32((h + (hi ^ s1) + np.uint64(0x9e3779b9)) & np.uint64(0xffffffff))


def attention_mask(E, H, nb, T, Tp, seed, p, Tq=None):
    """mask[e][h][blk][key][query] for the scores buffer geometry [E][H][nb][Tq][Tp]  (csn_block_salt / csn_pair_hash):
    every score block draws a salt from (seed, block id); one mixer round over (pair index ^ salt) decides the keys 2w
    (low 16 bits) and 2w+1 (high 16 bits) of query q, pair index = w * max(Tp, Tq) + q; keep <=> field >= p * 2^16.
    T keys, Tq queries per block (None: = T, the MID-FC layer)."""
    Tq = T if Tq is None else Tq
    salt = _hash2(np.arange(E * H * nb, dtype=np.uint64), seed).reshape(E, H, nb, 1, 1)
    key = np.arange(T, dtype=np.uint64).reshape(1, 1, 1, T, 1)
    q = np.arange(Tq, dtype=np.uint64).reshape(1, 1, 1, 1, Tq)
    pair = ((key >> np.uint64(1)) * np.uint64(max(Tp, Tq)) + q) & np.uint64(0xffffffff)
    h = _mix32(pair ^ salt)
    field = np.where((key & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xffff))
    thr = np.uint64(int(np.float32(p) * np.float32(65536.0)))
    return field >= thr


def fc_mask(E, C, N, seed, p, ld=None):
    """mask[e][c][n] of the fc-output dropout (csn_fc_pair / csn_keep16): evaluation e draws a salt from (seed, e); one mixer
    round over (pair index ^ salt) decides the channels 2w (low 16 bits) and 2w+1 (high 16 bits) of point n, pair index =
    w * ld + n with ld the row pitch of the xhat maps (None: = N); keep <=> field >= p * 2^16."""
    ld = N if ld is None else ld
    salt = _hash2(np.arange(E, dtype=np.uint64), seed).reshape(E, 1, 1)
    c = np.arange(C, dtype=np.uint64).reshape(1, C, 1)
    n = np.arange(N, dtype=np.uint64).reshape(1, 1, N)
    pair = ((c >> np.uint64(1)) * np.uint64(ld) + n) & np.uint64(0xffffffff)
    h = _mix32(pair ^ salt)
    field = np.where((c & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xffff))
    thr = np.uint64(int(np.float32(p) * np.float32(65536.0)))
    return field >= thr
