"""Cosine scoring back end (the step after the path): global mean, mean-subtracted cosine per trial, EER,
cohort top-k statistics and adaptive S-norm.

Behaviour of the reference's scripts/compute_mean.py:9-33, scripts/cosine_score.py:52-68,
scripts/compute_eer.py:35-105, scripts/compute_topk_mean_std.py:10-23 and scripts/adaptive_snorm.py:14-40 (same file
formats, same numbers).  Two back ends with the same results to fp32 rounding: `host` (numpy, vectorised instead of a
per-trial Python/torch loop) and `hip` (csrc/score.hip through ops.py: embeddings stay in HBM, one launch per stage).
"""
import numpy as np

from . import kaldi_io


def read_embeddings(ark_path):
    """utt -> float64 vector from the ark written by decode (text 'utt [ v0 ... ]' lines or binary FV records): an
    vecark.EmbTable - a dict like the one kaldi_io.read_vec_flt_ark yields, parsed natively (libspkio), that also carries the
    [n][D] matrix the vectorised paths below work on."""
    from . import vecark
    return vecark.load(ark_path)


def compute_mean(ark_path, mean_path=None):
    """float32 mean over all vectors (compute_mean.py builds a FloatTensor and torch.mean's it)."""
    from . import vecark
    mat = vecark.load(ark_path).mat.astype(np.float32)
    import torch
    mean = torch.from_numpy(mat).mean(dim=0).numpy()   # same reduction as the reference (torch.mean over rows)
    if mean_path:
        with open(mean_path, "w") as f:
            f.write(" [ " + " ".join(map(str, mean)) + " ]\n")
    return mean


def _table(vecs, mean):
    """dict utt -> vector  =>  (key -> row index, float32 [N][D] of mean-subtracted vectors).  The subtraction is done
    in float64 and cast to float32, as the reference does (numpy float64 ark values minus the mean, then FloatTensor).
    An EmbTable (read_embeddings) is converted in one vectorised pass over its matrix."""
    m = np.zeros(1) if mean is None else np.asarray(mean, dtype=np.float64)
    if hasattr(vecs, "mat") and len(vecs) == len(vecs.keys_list):       # (no duplicate keys: dict and matrix agree)
        return vecs.index, (vecs.mat - m).astype(np.float32)
    keys = list(vecs)
    mat = np.stack([(np.asarray(vecs[k], dtype=np.float64) - m).astype(np.float32) for k in keys])
    return {k: i for i, k in enumerate(keys)}, mat


def _device_rows(mat, eps):
    import torch
    from . import ops
    return ops.center_normalize(torch.from_numpy(np.ascontiguousarray(mat)).cuda(), None, eps)


def cosine_score(enroll, test, trials_path, mean=None, score_path=None, backend="host"):
    """scores for '<enroll> <test> target|nontarget' lines; vectors are mean-subtracted in float64, cast to
    float32, cosine = a.b / (max(|a|,eps) * max(|b|,eps)) with eps 1e-8 (F.cosine_similarity)."""
    pairs, labels = [], []
    for line in open(trials_path):
        a, b, t = line.strip().split()
        pairs.append((a, b))
        labels.append(1 if t == "target" else 0)
    ie, me = _table(enroll, mean)
    it, mt = (ie, me) if test is enroll else _table(test, mean)
    ia = np.fromiter((ie[a] for a, _ in pairs), dtype=np.int32, count=len(pairs))    # KeyError = unknown utterance
    ib = np.fromiter((it[b] for _, b in pairs), dtype=np.int32, count=len(pairs))
    if backend == "hip":
        import torch
        from . import ops
        en = _device_rows(me, 1e-8)
        te = en if test is enroll else _device_rows(mt, 1e-8)
        scores = ops.trial_cosine(en, te, torch.from_numpy(ia).cuda(), torch.from_numpy(ib).cuda()).cpu().numpy()
    else:
        assert backend == "host", backend
        # the same arithmetic as before, gathered in chunks so that 10^7 trials do not materialise two [T][D] matrices at once
        ne = np.maximum(np.linalg.norm(me, axis=1), 1e-8)
        nt = ne if test is enroll else np.maximum(np.linalg.norm(mt, axis=1), 1e-8)
        scores = np.empty(len(pairs), dtype=np.float32)
        for lo in range(0, len(pairs), 1 << 18):
            ja, jb = ia[lo:lo + (1 << 18)], ib[lo:lo + (1 << 18)]
            num = (me[ja] * mt[jb]).sum(1, dtype=np.float32)
            scores[lo:lo + len(ja)] = (num / (ne[ja] * nt[jb])).astype(np.float32)
    if score_path:
        with open(score_path, "w") as f:
            for (a, b), s in zip(pairs, scores):
                f.write("{} {} {}\n".format(a, b, s))
    return scores, np.array(labels)


def compute_eer(scores, labels):
    """EER as a fraction: sort by score, cumulative miss / false-alarm rates, argmin |fnr - fpr|, max of the two."""
    order = np.argsort(np.asarray(scores), kind="stable")
    lab = np.asarray(labels, dtype=np.float64)[order]
    fnrs = np.cumsum(lab) / lab.sum()
    fprs = 1.0 - np.cumsum(1.0 - lab) / (len(lab) - lab.sum())
    i = int(np.nanargmin(np.abs(fnrs - fprs)))
    return float(max(fprs[i], fnrs[i]))


def topk_mean_std(vecs, cohort, mean=None, topk=300, backend="host"):
    """utt -> (mean, std) of the `topk` largest cosine scores of the utterance against the cohort vectors
    (compute_topk_mean_std.py:10-23: both sides mean-subtracted and L2-normalised, unbiased std)."""
    iv, mv = _table(vecs, mean)
    _, mc = _table(cohort, mean)
    if topk > len(mc):
        raise RuntimeError("selected index k out of range: topk=%d > %d cohort vectors" % (topk, len(mc)))
    if backend == "hip":
        from . import ops
        v, c = _device_rows(mv, 1e-12), _device_rows(mc, 1e-12)
        scores = ops.gemm(v, c, v.shape[0], c.shape[0], v.shape[1], v.stride(0), 1, 1, c.stride(0))
        mu, sd = ops.topk_mean_std(scores, topk)
        mu, sd = mu.cpu().numpy(), sd.cpu().numpy()
    else:
        assert backend == "host", backend
        v = mv / np.maximum(np.linalg.norm(mv, axis=1, keepdims=True), 1e-12)
        c = mc / np.maximum(np.linalg.norm(mc, axis=1, keepdims=True), 1e-12)
        top = -np.sort(-(v @ c.T), axis=1)[:, :topk]
        mu = top.mean(axis=1, dtype=np.float32)
        sd = top.std(axis=1, ddof=1, dtype=np.float32)
    return {k: (np.float32(mu[i]), np.float32(sd[i])) for k, i in iv.items()}


def write_mean_std(stats, path):
    with open(path, "w") as f:
        for k, (m, s) in stats.items():
            f.write("{} {} {}\n".format(k, m, s))


def read_mean_std(path):
    out = {}
    for line in open(path):
        k, m, s = line.strip().split()
        out[k] = (float(m), float(s))
    return out


def adaptive_snorm(enroll_stats, test_stats, score_in, score_out=None):
    """adaptive_snorm.py:28-36: half the enrol-side plus half the test-side z-normalised score, in Python floats."""
    lines, out = [], []
    for line in open(score_in):
        a, b, sc = line.strip().split()
        sc = float(sc)
        v = (sc - enroll_stats[a][0]) / max(enroll_stats[a][1], 1e-8) / 2 + (sc - test_stats[b][0]) / max(test_stats[b][1], 1e-8) / 2
        out.append(v)
        lines.append("{} {} {}".format(a, b, v))
    if score_out:
        with open(score_out, "w") as f:
            f.write("\n".join(lines) + "\n")
    return out
