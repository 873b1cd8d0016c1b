import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_kat(name):
    raw = np.fromfile(os.path.join(GOLDEN, name), dtype=np.uint32)
    magic, func, n, ins, outs, _ = (int(v) for v in raw[:6])
    assert magic == 0x3154414B, "bad KAT magic in %s" % name
    data = raw[6:].view(np.float32)
    return func, data[:n * ins].reshape(n, ins).copy(), data[n * ins:n * ins + n * outs].reshape(n, outs).copy()


def bit_mismatch(expected, got):
    """Boolean mask of values that differ bitwise (NaN == NaN counts as equal)."""
    e, g = np.ascontiguousarray(expected, dtype=np.float32), np.ascontiguousarray(got, dtype=np.float32)
    return ~((e.view(np.uint32) == g.view(np.uint32)) | (np.isnan(e) & np.isnan(g)))
