"""Cosine scoring back end (the step after the path): global mean, mean-subtracted cosine per trial, EER.

Behaviour of the reference's scripts/compute_mean.py:9-33, scripts/cosine_score.py:52-68 and
scripts/compute_eer.py:35-105 (same file formats, same numbers), vectorised with numpy instead of a
per-trial Python/torch loop.
"""
import numpy as np

from . import kaldi_io


def read_embeddings(ark_path):
    """utt -> float64 vector from the text ark written by decode ('utt [ v0 ... ]')."""
    return {k: v for k, v in kaldi_io.read_vec_flt_ark(ark_path)}


def compute_mean(ark_path, mean_path=None):
    """float32 mean over all vectors (compute_mean.py builds a FloatTensor and torch.mean's it)."""
    mat = np.stack([np.asarray(v, dtype=np.float32) for _, v in kaldi_io.read_vec_flt_ark(ark_path)])
    import torch
    mean = torch.from_numpy(mat).mean(dim=0).numpy()   # same reduction as the reference (torch.mean over rows)
    if mean_path:
        with open(mean_path, "w") as f:
            f.write(" [ " + " ".join(map(str, mean)) + " ]\n")
    return mean


def cosine_score(enroll, test, trials_path, mean=None, score_path=None):
    """scores for '<enroll> <test> target|nontarget' lines; vectors are mean-subtracted in float64, cast to
    float32, cosine = a.b / (max(|a|,eps) * max(|b|,eps)) with eps 1e-8 (F.cosine_similarity)."""
    pairs, labels = [], []
    for line in open(trials_path):
        a, b, t = line.strip().split()
        pairs.append((a, b))
        labels.append(1 if t == "target" else 0)
    m = np.zeros(1) if mean is None else np.asarray(mean, dtype=np.float64)
    ea = np.stack([(np.asarray(enroll[a], dtype=np.float64) - m).astype(np.float32) for a, _ in pairs])
    tb = np.stack([(np.asarray(test[b], dtype=np.float64) - m).astype(np.float32) for _, b in pairs])
    num = (ea * tb).sum(1, dtype=np.float32)
    den = np.maximum(np.linalg.norm(ea, axis=1), 1e-8) * np.maximum(np.linalg.norm(tb, axis=1), 1e-8)
    scores = (num / den).astype(np.float32)
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
