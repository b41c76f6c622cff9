import os
import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def math_golden():
    return np.load(os.path.join(GOLD, "math_golden.npz"))


def warp_golden():
    return np.load(os.path.join(GOLD, "warp_golden.npz"))


def blend_golden():
    return np.load(os.path.join(GOLD, "blend_golden.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def load_match_pairs():
    raw = np.fromfile(os.path.join(GOLD, "matchPairs.match"), np.uint8)
    n = int(raw[:4].view(np.int32)[0])
    rec = raw[4:].view(np.dtype([("ax", "<f4"), ("ay", "<f4"), ("aid", "<i4"), ("ai", "<i4"), ("af", "<i4"),
                                 ("bx", "<f4"), ("by", "<f4"), ("bid", "<i4"), ("bi", "<i4"), ("bf", "<i4")]))
    assert len(rec) == n
    return rec
