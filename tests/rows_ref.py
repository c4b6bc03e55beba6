"""numpy restatement of the gradient-row segment format (include/riggs_hip.h: riggs_grad_rows_pack / _unpack) — test
infrastructure: the checker of csrc/exchange.hip on the GPU and the stand-in pack / unpack of the gloo protocol test."""
import numpy as np

from riggs_amd.dist import segment_words


def rows_offset(N):
    return (4 + (N + 255) // 256 + 1 + 3) // 4 * 4


def pack(grads, touched, scale, capacity, row_floats):
    """grads: list of (N, w) float32 arrays; touched: (N,) bool.  Returns the segment as an int32 array."""
    N = grads[0].shape[0]
    nb = (N + 255) // 256
    seg = np.zeros(segment_words(N, row_floats, capacity), np.int32)
    idx = np.nonzero(touched)[0]
    need = len(idx)
    seg[0], seg[1], seg[2], seg[3] = min(need, capacity), need, N, row_floats
    counts = np.bincount(idx // 256, minlength=nb)
    seg[4:4 + nb + 1] = np.concatenate([[0], np.cumsum(counts)])
    keep = idx[:capacity]
    rows = np.zeros((capacity, row_floats), np.float32)
    rows[:len(keep), 0] = keep.astype(np.int32).view(np.float32)
    o = 1
    for g in grads:
        w = g.shape[1]
        rows[:len(keep), o:o + w] = g[keep] * np.float32(scale)
        o += w
    ro = rows_offset(N)
    seg[ro:ro + capacity * row_floats] = rows.reshape(-1).view(np.int32)
    return seg


def unpack(grads, segments, capacity, row_floats):
    """In place on ``grads``; segments: (world, words) int32.  Returns (max rows needed, overflowed)."""
    N = grads[0].shape[0]
    world = segments.shape[0]
    need = int(segments[:, 1].max())
    bad = bool((segments[:, 1] > capacity).any() or (segments[:, 2] != N).any() or (segments[:, 3] != row_floats).any())
    if bad:
        return need, True
    ro = rows_offset(N)
    seen = np.zeros(N, bool)
    for r in range(world):
        n = int(segments[r, 0])
        rows = segments[r, ro:ro + capacity * row_floats].view(np.float32).reshape(capacity, row_floats)[:n]
        idx = rows[:, 0].copy().view(np.int32)
        first = ~seen[idx]
        o = 1
        for g in grads:
            w = g.shape[1]
            v = rows[:, o:o + w]
            g[idx[first]] = v[first]
            g[idx[~first]] = g[idx[~first]] + v[~first]
            o += w
        seen[idx] = True
    return need, False
