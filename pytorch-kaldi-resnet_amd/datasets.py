"""Dataset classes over Kaldi feats.scp (host side, kept in Python per north_star).

Behaviour of the reference's scripts/datasets.py (SequenceDataset :7-72, SequenceDataset2 :74-146,
EmbeddingDataset :148-193), rewritten here: same class-balancing rule, same random crop, same [F, T] sample
layout (freq-major, time innermost) - and runnable on numpy >= 1.24 (the reference's np.int is gone).
"""
import numpy as np
from torch.utils.data import Dataset, Sampler

from . import kaldi_io


def _read_table(path):
    out = []
    for line in open(path):
        a, b = line.rstrip().split(None, 1)
        out.append((a, b))
    return out


def _crop(full, seq_len):
    """Random crop of seq_len frames, transposed to [F, T] (reference datasets.py:64-68)."""
    assert len(full) >= seq_len
    pin = np.random.randint(0, len(full) - seq_len + 1)
    return np.ascontiguousarray(full[pin:pin + seq_len, :].T)


def chunk_length_schedule(lo, hi, quantum, nbatches, seed, epoch):
    """One chunk length per batch for variable-length training (the reference draws lengths in [min, max] per sample,
    scripts/datasets.py:40-43, which default collation cannot batch; SURVEY.md section 5: one T per batch).  Lengths are
    lo, lo + quantum, ... <= hi; the draw depends on (seed, epoch) only, so every rank sees the same length for batch i."""
    lo, hi, quantum = int(lo), int(hi), max(1, int(quantum))
    assert 0 < lo <= hi
    choices = np.arange(lo, hi + 1, quantum)
    g = np.random.RandomState((int(seed) * 1000003 + int(epoch) * 7919 + 12345) % (2 ** 31 - 1))
    return choices[g.randint(0, len(choices), size=int(nbatches))]


class ChunkBatchSampler(Sampler):
    """Batch sampler for variable-length training: wraps an index sampler (RandomSampler / DistributedSampler) and yields
    batches of (index, T) pairs whose T comes from chunk_length_schedule - the length travels WITH the index, so DataLoader
    worker processes need no shared state (Dataset.set_chunk_size in the parent would not reach them)."""

    def __init__(self, sampler, batch_size, lo, hi, quantum=8, seed=0, drop_last=False):
        self.sampler, self.batch_size, self.drop_last = sampler, int(batch_size), drop_last
        self.lo, self.hi, self.quantum, self.seed, self.epoch = lo, hi, quantum, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch
        if hasattr(self.sampler, "set_epoch"):
            self.sampler.set_epoch(epoch)

    def __len__(self):
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        idx = list(self.sampler)
        lens = chunk_length_schedule(self.lo, self.hi, self.quantum, len(self), self.seed, self.epoch)
        for b in range(len(self)):
            sel = idx[b * self.batch_size:(b + 1) * self.batch_size]
            yield [(int(i), int(lens[b])) for i in sel]


def _index_and_len(index, default_len):
    """Dataset index: a plain int (chunk length from the dataset), or an (index, T) pair from ChunkBatchSampler"""
    if isinstance(index, (tuple, list)):
        return int(index[0]), int(index[1])
    return index, default_len


class SequenceDataset(Dataset):
    """Utterance-level training set with class-balanced repetition:
    rep(label) = max(1, min(500, (max_count + 1) // 2) // count[label]) copies of each scp line."""

    def __init__(self, scp_file, utt2spkid_file, chunk_size):
        self.utt2spkid = {u: int(s) for u, s in _read_table(utt2spkid_file)}
        count = {}
        for s in self.utt2spkid.values():
            count[s] = count.get(s, 0) + 1
        cap = min(500, int((max(count.values()) + 1) / 2))
        rx, lab = [], []
        for utt, rxfile in _read_table(scp_file):
            label = self.utt2spkid[utt]
            rep = max(1, cap // count[label])
            rx.extend([rxfile] * rep)
            lab.extend([label] * rep)
        self.rxfiles = np.array(rx)
        self.labels = np.array(lab, dtype=np.int64)
        if isinstance(chunk_size, int):
            self.seq_len = np.full(len(lab), chunk_size, dtype=np.int64)
        elif len(chunk_size) == 1:
            self.seq_len = np.full(len(lab), chunk_size[0], dtype=np.int64)
        else:
            self.seq_len = np.random.randint(min(chunk_size), max(chunk_size) + 1, size=len(lab))
        print("Totally " + str(len(self.rxfiles)) + " samples with at most " + str(cap) + " samples for one class")

    def __len__(self):
        return len(self.labels)

    def set_chunk_size(self, seq_len):
        self.seq_len = seq_len

    def __getitem__(self, index):
        index, T = _index_and_len(index, None)
        full = kaldi_io.read_mat(self.rxfiles[index])
        return _crop(full, int(self.seq_len[index]) if T is None else T), np.array(self.labels[index])


class SequenceDataset2(Dataset):
    """Speaker-uniform sampling: index -> speaker (round robin), utterance drawn at random."""

    def __init__(self, scp_file, utt2spkid_file, chunk_size):
        utt2spkid = {u: int(s) for u, s in _read_table(utt2spkid_file)}
        self.rxfiles = {}
        count = {}
        for utt, rxfile in _read_table(scp_file):
            s = utt2spkid[utt]
            count[s] = count.get(s, 0) + 1
            self.rxfiles.setdefault(s, []).append(rxfile)
        self.repetition = int((max(count.values()) + 1) / 2)
        print("id_count: {}".format(max(count.values())))
        self.labels = np.array(sorted(self.rxfiles.keys()))
        self.seq_len = chunk_size
        self.num_spk = len(self.rxfiles)
        print("Totally " + str(self.num_spk) + " speakers with at most " + str(self.repetition) + " samples for one class")

    def __len__(self):
        return len(self.labels) * self.repetition

    def set_chunk_size(self, seq_len):
        self.seq_len = seq_len

    def __getitem__(self, index):
        index, T = _index_and_len(index, self.seq_len)
        spk = self.labels[index % self.num_spk]
        files = self.rxfiles[spk]
        full = kaldi_io.read_mat(files[np.random.randint(0, len(files))])
        return _crop(full, T), np.array(spk)


class EmbeddingDataset(Dataset):
    """Extraction set: whole utterance when chunk_size == -1 (variable T -> batch size 1), else a random crop.
    Returns (features [F, T], utt-id); unlike the reference's `[utt]` wrapper (decode.py:204 only works at
    per-process batch size 1 because of it) the id is a plain string so default collation gives a list."""

    def __init__(self, scp_file, chunk_size=-1):
        tab = _read_table(scp_file)
        self.utts = [u for u, _ in tab]
        self.rxfiles = np.array([r for _, r in tab])
        self.seq_len = chunk_size
        print("Totally " + str(len(self.rxfiles)) + " samples")

    def __len__(self):
        return len(self.rxfiles)

    def set_chunk_size(self, seq_len):
        self.seq_len = seq_len

    def __getitem__(self, index):
        full = kaldi_io.read_mat(self.rxfiles[index])
        assert len(full) >= self.seq_len
        if self.seq_len > -1:
            chunk = _crop(full, self.seq_len)
        else:
            chunk = np.ascontiguousarray(full.T)
        return chunk, self.utts[index]
