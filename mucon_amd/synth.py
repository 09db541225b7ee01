"""Seeded, platform-independent synthetic inputs (Breakfast-I3D-shaped tapes, emissions, transcripts).

Every value is produced by integer arithmetic (SplitMix64, counter based) followed by exact
power-of-two scaling and single IEEE adds, so the same seed gives the same bits on every host:
golden fixtures under tests/golden/ store only the expected OUTPUTS and the seeds.
Shapes follow SURVEY.md section 8d (tape [T x 2048] f32, emissions [T x 48] f32 log-prob-like).
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n 64-bit words of the SplitMix64 stream started at `seed` (vectorised, counter based)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.arange(1, n + 1, dtype=np.uint64) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, shape) -> np.ndarray:
    """float32 uniform in [0,1) with 24 random mantissa bits (exact conversion)."""
    n = int(np.prod(shape))
    bits = (splitmix64(seed, n) >> np.uint64(40)).astype(np.float32)  # < 2^24: exact
    return (bits * np.float32(2.0 ** -24)).reshape(shape)


def uniform_pm1(seed: int, shape) -> np.ndarray:
    """float32 uniform in [-1,1)."""
    return (uniform01(seed, shape) * np.float32(2.0) - np.float32(1.0)).astype(np.float32)


def integers(seed: int, n: int, lo: int, hi: int) -> np.ndarray:
    """n int64 in [lo, hi) (modulo reduction of 64-bit words; bias is irrelevant here)."""
    return (splitmix64(seed, n) % np.uint64(hi - lo)).astype(np.int64) + lo


def tape(seed: int, B: int, T: int, D: int = 2048) -> np.ndarray:
    """Feature tape [B x T x D] f32 in [-1,1) (reference layout: general_dataset.py:148-150)."""
    return uniform_pm1(seed, (B, T, D))


def transcript(seed: int, N: int, C: int = 48, allow_repeats: bool = True) -> np.ndarray:
    """N action labels in [0, C)."""
    t = integers(seed, N, 0, C)
    if not allow_repeats:
        for i in range(1, N):
            if t[i] == t[i - 1]:
                t[i] = (t[i] + 1) % C
    return t


def emissions(seed: int, T: int, C: int = 48, labels: np.ndarray = None, margin: float = 4.0) -> np.ndarray:
    """Log-prob-like frame scores [T x C] f32, all < 0, full 24-bit mantissas.

    lp = -(8*u) - margin*[c != labels[t]]  (labels optional).  Not normalised: the decoder never
    needs normalisation (reference viterbi.py:49-65 only sums them).
    """
    lp = -(uniform01(seed, (T, C)) * np.float32(8.0))
    if labels is not None:
        pen = np.full((T, C), np.float32(margin), dtype=np.float32)
        pen[np.arange(T), np.asarray(labels, dtype=np.int64)] = np.float32(0.0)
        lp = (lp - pen).astype(np.float32)
    return np.ascontiguousarray(lp, dtype=np.float32)


def segment_labels(seed: int, T: int, trans: np.ndarray) -> np.ndarray:
    """A plausible frame labelling of length T following `trans` (random positive segment lengths)."""
    N = len(trans)
    w = uniform01(seed, (N,)).astype(np.float64) + 0.25
    cuts = np.floor(np.cumsum(w) / w.sum() * T).astype(np.int64)
    cuts[-1] = T
    out = np.empty(T, dtype=np.int64)
    s = 0
    for n in range(N):
        e = max(int(cuts[n]), s)
        out[s:e] = trans[n]
        s = e
    out[s:] = trans[-1]
    return out


def mean_lengths(trans: np.ndarray, rel: np.ndarray, Tf: int, C: int = 48) -> np.ndarray:
    """Per-class mean lengths the way the evaluator builds them (reference evaluators.py:155-165)."""
    actions = np.eye(C)[np.asarray(trans).reshape(-1)]
    lengths = np.dot(np.asarray(rel, dtype=np.float32), actions)
    lengths *= Tf
    k = actions.sum(0)
    k[k == 0] = 1
    lengths /= k
    lengths[lengths == 0] = 1
    return lengths
