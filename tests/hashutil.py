"""Order-sensitive integer digests of output streams, computable the same way with numpy (reference side, worker processes)
and with torch on the device (product side) -- so that EVERY unit of a full-size batch (every instance of Tiger x10k, every
mesh of the 10k round-join polylines, every path of the 1 M cubics) can be compared with the reference without moving 9 GB
to the host. TEST INFRASTRUCTURE.

digest of a segment v[0..n) of 32-bit words = (sum lo, sum hi, sum lo*k, sum hi*k), lo / hi = the 16-bit halves of a word,
k = 1-based position inside the segment. All four stay below 2^63 for segments up to 2^30 words (no wrap-around involved
in the definition); any single changed word changes the digest, and so does any transposition of two unequal words."""
import numpy as np


def digest_uniform_np(words, nseg):
    """words: uint32 array of nseg equal segments. Returns int64 [nseg, 4]."""
    w = np.ascontiguousarray(words).reshape(nseg, -1).astype(np.int64)
    k = np.arange(1, w.shape[1] + 1, dtype=np.int64)[None, :]
    lo, hi = w & 0xFFFF, w >> 16
    return np.stack([lo.sum(1), hi.sum(1), (lo * k).sum(1), (hi * k).sum(1)], axis=1)


def digest_ragged_np(words, starts, counts):
    """words: uint32 array; segment i = words[starts[i] : starts[i] + counts[i]] (any order, may leave gaps). int64 [n, 4]."""
    out = np.zeros((len(starts), 4), dtype=np.int64)
    w = np.ascontiguousarray(words).astype(np.int64)
    for i, (s, c) in enumerate(zip(starts, counts)):
        seg = w[int(s):int(s) + int(c)]
        k = np.arange(1, seg.shape[0] + 1, dtype=np.int64)
        lo, hi = seg & 0xFFFF, seg >> 16
        out[i] = (lo.sum(), hi.sum(), (lo * k).sum(), (hi * k).sum())
    return out


def digest_uniform_torch(words_i32, nseg, chunk=256):
    """words_i32: int32 (or int16 for index streams) device tensor of nseg equal segments. Returns int64 [nseg, 4] on the host."""
    import torch
    w = words_i32.view(nseg, -1)
    L = w.shape[1]
    k = torch.arange(1, L + 1, dtype=torch.int64, device=w.device)[None, :]
    out = []
    for a in range(0, nseg, chunk):
        x = w[a:a + chunk].to(torch.int64)
        if words_i32.dtype == torch.int16:
            x = x & 0xFFFF
        else:
            x = x & 0xFFFFFFFF
        lo, hi = x & 0xFFFF, x >> 16
        out.append(torch.stack([lo.sum(1), hi.sum(1), (lo * k).sum(1), (hi * k).sum(1)], dim=1))
    return torch.cat(out).cpu().numpy()


def digest_ragged_torch(words, starts, counts, is_u16=False):
    """Segments given by int64 device tensors starts / counts (contiguous or not). Prefix sums in wrapping int64 arithmetic:
    differences of wrapped prefix sums are exact modulo 2^64 and the true digests are below 2^63. Returns int64 [n, 4] on the host."""
    import torch
    x = words.reshape(-1).to(torch.int64)
    x = x & (0xFFFF if is_u16 else 0xFFFFFFFF)
    n = x.shape[0]
    g = torch.arange(1, n + 1, dtype=torch.int64, device=x.device)  # global 1-based position
    res = []
    ends = starts + counts
    for part in (x & 0xFFFF, x >> 16):
        P = torch.zeros(n + 1, dtype=torch.int64, device=x.device)
        torch.cumsum(part, 0, out=P[1:])
        W = torch.zeros(n + 1, dtype=torch.int64, device=x.device)
        torch.cumsum(part * g, 0, out=W[1:])
        s0 = P[ends] - P[starts]
        s1 = (W[ends] - W[starts]) - starts * s0  # sum v * (global - start) with global 1-based = sum v * k
        res.append((s0, s1))
        del P, W
    return torch.stack([res[0][0], res[1][0], res[0][1], res[1][1]], dim=1).cpu().numpy()
