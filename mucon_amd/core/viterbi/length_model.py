"""Length models -- host-side mirror of reference src/core/viterbi/length_model.py.

PoissonModel (reference length_model.py:42-83) is the one the evaluator builds per video
(src/mucon/evaluators.py:167).  The table is computed on the host in float64 with NumPy's `log`,
vectorised but in the reference's exact operation order (sequential log-factorial sums, left to
right arithmetic), so its bits equal the reference's on the same host; only the rows the decoder
reads (lengths fs, 2fs, ...) are uploaded.  MultiPoissonModel / MeanLengthModel are out of scope:
the former's score() raises in the reference, the latter is unused."""
import numpy as np


_LOG_FACTORIALS = {}


def _log_factorials(n):
    """[log 0!, log 1!, ..., log (n-1)!] as the reference accumulates them (`logFak += np.log(k)`); one table per n (an
    evaluation builds a PoissonModel per video with the same max_length)."""
    t = _LOG_FACTORIALS.get(n)
    if t is None:
        with np.errstate(all="ignore"):
            t = _LOG_FACTORIALS[n] = np.concatenate(([0.0], np.cumsum(np.log(np.arange(1, n)))))
        t.setflags(write=False)
    return t


class LengthModel(object):
    def n_classes(self):
        return 0

    def score(self, length, label):
        return 0.0

    def max_length(self):
        return np.inf


class PoissonModel(LengthModel):
    def __init__(self, model, max_length=2000, renormalize=True):
        if isinstance(model, str):
            self.mean_lengths = np.loadtxt(model)
        else:
            self.mean_lengths = np.asarray(model)
        self.num_classes = self.mean_lengths.shape[0]
        self.max_len = int(max_length)
        mu = self.mean_lengths
        with np.errstate(all="ignore"):
            # log-factorial partial sums, accumulated sequentially from 0 like `logFak += np.log(k)`
            lf = _log_factorials(max(self.max_len, 2))  # lf[l] = sum_{k<=l} log k
            self.norms = np.zeros(mu.shape)
            if renormalize:
                self.norms = np.round(mu) * np.log(np.round(mu)) - np.round(mu)
                kmax = int(np.nanmax(np.where(np.isfinite(mu), mu, 0))) if mu.size else 0
                lf2 = np.concatenate(([0.0, 0.0], np.cumsum(np.log(np.arange(2, max(kmax + 1, 3))))))  # sum_{k=2..m} log k
                m = np.where(np.isfinite(mu), mu, 0).astype(np.int64)
                self.norms = self.norms - np.where(m >= 2, lf2[np.clip(m, 0, len(lf2) - 1)], 0)
            self._lf, self._logmu, self._poisson = lf, np.log(mu), None

    def _table(self, lengths, classes):
        """score(l, c) for l in lengths (int64 array), c in classes: the reference's expression, element by element
        (`l * log(mu_c) - mu_c - logFak(l) - norm_c`, left to right), so any subset carries the bits of the full table."""
        with np.errstate(all="ignore"):
            t = (lengths[:, None] * self._logmu[None, classes] - self.mean_lengths[None, classes] - self._lf[lengths, None]
                 - self.norms[None, classes])
        t[lengths == 0, :] = -np.inf  # length zero can not happen
        return t

    @property
    def poisson(self):
        """The full [max_len x classes] table of the reference (length_model.py:60-75), built on first use: the decoder only
        ever reads the rows fs, 2 fs, ... of the transcript's classes (rows_for), an evaluation builds one model per video."""
        if self._poisson is None:
            self._poisson = self._table(np.arange(self.max_len, dtype=np.int64), np.arange(self.num_classes))
        return self._poisson

    def n_classes(self):
        return self.num_classes

    def score(self, length, label):
        if length >= self.max_len:
            return -np.inf
        return self.poisson[length, label]

    def max_length(self):
        return self.max_len

    def rows_for(self, transcript, frame_sampling):
        """P[J x N] float64: P[j, n] = score((j+1)*fs, a_n), J = max_len // fs (what the kernel reads)."""
        J = self.max_len // frame_sampling
        lengths = (np.arange(J) + 1) * frame_sampling
        tr = np.asarray(transcript, dtype=np.int64)
        P = np.full((J, len(tr)), -np.inf, dtype=np.float64)
        ok = lengths < self.max_len
        P[ok, :] = self._table(lengths[ok].astype(np.int64), tr)
        return np.ascontiguousarray(P)


class PoissonRows(LengthModel):
    """A PoissonModel reduced to what the decoder reads of it: the rows P[j, n] = score((j+1) * fs, a_n) of ONE transcript (poisson_rows_for_many
    builds them for a whole evaluation chunk in a handful of array operations instead of a model object per video)."""

    def __init__(self, rows: np.ndarray, max_length: int, frame_sampling: int):
        self.rows, self.max_len, self.frame_sampling = rows, int(max_length), int(frame_sampling)

    def max_length(self):
        return self.max_len

    def rows_for(self, transcript, frame_sampling):
        if int(frame_sampling) != self.frame_sampling or self.rows.shape[1] != len(transcript):
            raise ValueError("PoissonRows was built for another transcript / frame_sampling")
        return self.rows

    def score(self, length, label):
        raise NotImplementedError("PoissonRows holds the decoder's rows only; build a PoissonModel for single scores")


def poisson_rows_for_many(mean_lengths, transcripts, frame_sampling: int, max_length: int = 2000):
    """[PoissonModel(mu).rows_for(tr, fs) for mu, tr in zip(mean_lengths, transcripts)], bit for bit, computed together: the same
    element-wise expressions (reference length_model.py:43-80: norms from round(mu), the table `l * log(mu_c) - mu_c - logFak(l) - norm_c`
    left to right) on the concatenation of all videos' transcript classes.  -> list of float64 [J x N_v] arrays, J = max_length // fs."""
    nv = len(transcripts)
    if nv == 0:
        return []
    mu = np.asarray(mean_lengths, dtype=np.float64).reshape(nv, -1)
    max_len, fs = int(max_length), int(frame_sampling)
    J = max_len // fs
    trs = [np.asarray(t, dtype=np.int64) for t in transcripts]
    counts = [len(t) for t in trs]
    vid = np.repeat(np.arange(nv), counts)
    cls = np.concatenate(trs) if sum(counts) else np.zeros(0, np.int64)
    with np.errstate(all="ignore"):
        lf = _log_factorials(max(max_len, 2))
        r = np.round(mu)
        norms = r * np.log(r) - r
        finite = np.where(np.isfinite(mu), mu, 0)
        kmax = int(np.nanmax(finite)) if mu.size else 0
        lf2 = np.concatenate(([0.0, 0.0], np.cumsum(np.log(np.arange(2, max(kmax + 1, 3))))))     # sum_{k=2..m} log k: a prefix of it is a video's own
        m = finite.astype(np.int64)
        norms = norms - np.where(m >= 2, lf2[np.clip(m, 0, len(lf2) - 1)], 0)
        logmu = np.log(mu)
        lengths = (np.arange(J, dtype=np.int64) + 1) * fs
        ok = lengths < max_len
        P = np.full((J, len(cls)), -np.inf, dtype=np.float64)
        lo = lengths[ok]
        t = lo[:, None] * logmu[vid, cls][None, :] - mu[vid, cls][None, :] - lf[lo, None] - norms[vid, cls][None, :]
        t[lo == 0, :] = -np.inf
        P[ok, :] = t
    out, at = [], 0
    for n in counts:
        out.append(np.ascontiguousarray(P[:, at: at + n]))
        at += n
    return out


class PoissonParams(LengthModel):
    """A PoissonModel reduced to what the DEVICE needs to build the decoder's rows itself (csrc/viterbi.hip: VitTab; include/mucon_hip.h,
    mucon_viterbi_decode_host_poisson): per transcript state n the three numbers of `l * np.log(mu) - mu - logFak - norms` that depend on the
    class -- params[0, n] = np.log(mu_c), params[1, n] = mu_c, params[2, n] = norms_c for c = a_n (reference length_model.py:54-71; NumPy's log,
    computed here on the host) -- and the shared running log-factorial at the lengths the decoder reads, log_fact[j] = logFak((j + 1) * fs).
    3 doubles per state cross PCIe instead of J = 66.  rows_for() builds the same rows on the host, bit for bit what PoissonModel.rows_for gives."""

    def __init__(self, params: np.ndarray, log_fact: np.ndarray, max_length: int, frame_sampling: int):
        self.params, self.log_fact, self.max_len, self.frame_sampling = params, log_fact, int(max_length), int(frame_sampling)

    def max_length(self):
        return self.max_len

    def row(self, j: int) -> np.ndarray:
        """P[j, :] (float64 [N]): the reference's expression, left to right; lengths >= max_length score -inf (length_model.py:76-80)."""
        length = (j + 1) * self.frame_sampling
        if length >= self.max_len:
            return np.full(self.params.shape[1], -np.inf)
        with np.errstate(all="ignore"):
            return length * self.params[0] - self.params[1] - self.log_fact[j] - self.params[2]

    def rows_for(self, transcript, frame_sampling):
        if int(frame_sampling) != self.frame_sampling or self.params.shape[1] != len(transcript):
            raise ValueError("PoissonParams was built for another transcript / frame_sampling")
        J = self.max_len // self.frame_sampling
        return np.ascontiguousarray(np.stack([self.row(j) for j in range(J)])) if J else np.zeros((0, len(transcript)))

    def score(self, length, label):
        raise NotImplementedError("PoissonParams holds one transcript's parameters only; build a PoissonModel for single scores")


def poisson_params_for_many(mean_lengths, transcripts, frame_sampling: int, max_length: int = 2000):
    """[PoissonParams of PoissonModel(mu) for the transcript tr, for mu, tr in zip(mean_lengths, transcripts)]: the norms and logs of
    poisson_rows_for_many (same element-wise expressions, reference length_model.py:43-63), without the J x N tables -- those are built on the
    device (bit for bit: tests/test_gpu_viterbi.py::test_device_built_length_rows).  All models share one log_fact array."""
    nv = len(transcripts)
    if nv == 0:
        return []
    mu = np.asarray(mean_lengths, dtype=np.float64).reshape(nv, -1)
    max_len, fs = int(max_length), int(frame_sampling)
    J = max_len // fs
    trs = [np.asarray(t, dtype=np.int64) for t in transcripts]
    counts = [len(t) for t in trs]
    vid = np.repeat(np.arange(nv), counts)
    cls = np.concatenate(trs) if sum(counts) else np.zeros(0, np.int64)
    with np.errstate(all="ignore"):
        lf = _log_factorials(max(max_len, 2))
        r = np.round(mu)
        norms = r * np.log(r) - r
        finite = np.where(np.isfinite(mu), mu, 0)
        kmax = int(np.nanmax(finite)) if mu.size else 0
        lf2 = np.concatenate(([0.0, 0.0], np.cumsum(np.log(np.arange(2, max(kmax + 1, 3))))))     # sum_{k=2..m} log k: a prefix of it is a video's own
        m = finite.astype(np.int64)
        norms = norms - np.where(m >= 2, lf2[np.clip(m, 0, len(lf2) - 1)], 0)
        logmu = np.log(mu)
    lengths = (np.arange(J, dtype=np.int64) + 1) * fs
    log_fact = np.ascontiguousarray(lf[np.minimum(lengths, len(lf) - 1)])
    log_fact.setflags(write=False)
    flat = np.stack([logmu[vid, cls], mu[vid, cls], norms[vid, cls]])          # [3, sum N]
    out, at = [], 0
    for n in counts:
        out.append(PoissonParams(np.ascontiguousarray(flat[:, at: at + n]), log_fact, max_len, fs))
        at += n
    return out
