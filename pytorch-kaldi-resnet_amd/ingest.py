"""Native batch ingest (libspkio): scp/ark -> pinned [B, F, T] staging, cropped at read time.

Same sampling semantics as the reference's SequenceDataset + DataLoader(shuffle / DistributedSampler)
(scripts/datasets.py:10-72, scripts/train_resnet.py:237-247): class-balanced repetition of scp lines, a fresh
permutation per epoch (seeded with the epoch like DistributedSampler.set_epoch), equal shards per rank (padded by
wrap-around), a uniform random crop start per sample.  What changes is the mechanics: one C++ call reads a whole
batch with pread() on a small thread pool, reading only the cropped frames, and transposes them straight into pinned
memory; a background thread keeps the next batch ready while the GPU works on the current one.
"""
import ctypes
import os
import queue
import threading

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libspkio.so")
        if not os.path.exists(path):
            raise RuntimeError("libspkio.so is missing at %s: run `python __graft_entry__.py build`" % path)
        l = ctypes.CDLL(path)
        l.spk_io_last_error.restype = ctypes.c_char_p
        cpp, i64p, i32p = ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)
        l.spk_ark_probe.argtypes = [ctypes.c_int, cpp, i64p, i32p, i32p, i64p]
        l.spk_ark_read_crop.argtypes = [ctypes.c_int, cpp, i64p, i32p, i32p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_int]
        l.spk_text_vectors_bound.argtypes = [ctypes.c_int, ctypes.c_int, cpp]
        l.spk_text_vectors_bound.restype = ctypes.c_int64
        l.spk_format_text_vectors.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, cpp, ctypes.c_char_p, ctypes.c_int64,
                                              ctypes.c_int]
        l.spk_format_text_vectors.restype = ctypes.c_int64
        _LIB = l
    return _LIB


def format_text_vectors(keys, vecs, nthreads=4):
    """bytes of the text ark 'key [ v0 v1 ... ]\\n' per row of float32 vecs [n][D], every value printed exactly like
    numpy's str(np.float32) (the reference's scripts/decode.py:206 line format), formatted natively on `nthreads` threads."""
    vecs = np.ascontiguousarray(vecs, dtype=np.float32)
    n, D = vecs.shape
    assert len(keys) == n
    arr = (ctypes.c_char_p * n)(*[k.encode() for k in keys])
    cap = lib().spk_text_vectors_bound(n, D, arr)
    buf = ctypes.create_string_buffer(cap)
    w = lib().spk_format_text_vectors(n, D, vecs.ctypes.data, arr, buf, cap, int(nthreads))
    if w < 0:
        raise RuntimeError("spk_format_text_vectors: buffer too small")
    return buf.raw[:w]


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, lib().spk_io_last_error().decode()))


def _split_rx(rx):
    path, off = rx.rsplit(":", 1)
    return path, int(off)


class ArkTable:
    """Parsed scp: per line the ark path, the payload offset and the frame count (headers probed once)."""

    def __init__(self, rxfiles):
        paths, offs = zip(*[_split_rx(r) for r in rxfiles])
        uniq = sorted(set(paths))
        self._cpaths = {p: ctypes.c_char_p(p.encode()) for p in uniq}
        self.paths = list(paths)
        n = len(paths)
        self.rows = np.zeros(n, dtype=np.int32)
        self.cols = np.zeros(n, dtype=np.int32)
        self.data_off = np.zeros(n, dtype=np.int64)
        arr = (ctypes.c_char_p * n)(*[self._cpaths[p].value for p in paths])
        offs = np.asarray(offs, dtype=np.int64)
        _check(lib().spk_ark_probe(n, arr, offs.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                   self.rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                   self.cols.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                   self.data_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))), "spk_ark_probe")

    def read_crop(self, idx, starts, T, out, nthreads=4):
        """out: contiguous float32 host tensor [B, F, T] (ideally pinned)."""
        B = len(idx)
        F = int(self.cols[idx[0]])
        assert out.is_contiguous() and tuple(out.shape) == (B, F, T) and out.dtype == torch.float32
        assert (self.cols[idx] == F).all()
        arr = (ctypes.c_char_p * B)(*[self._cpaths[self.paths[i]].value for i in idx])
        doff = np.ascontiguousarray(self.data_off[idx])
        rows = np.ascontiguousarray(self.rows[idx])
        st = np.ascontiguousarray(np.asarray(starts, dtype=np.int32))
        _check(lib().spk_ark_read_crop(B, arr, doff.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                       rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                       st.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), F, T, out.data_ptr(), nthreads),
               "spk_ark_read_crop")
        return out


class NativeTrainLoader:
    """Iterable over (features [B,F,T] pinned, labels [B] int64) batches of one epoch for this rank."""

    def __init__(self, scp_file, utt2spkid_file, chunk_size, batch_size, rank=0, world=1, seed=0, threads=4,
                 drop_last=False, prefetch=2, device=None, chunk_range=None):
        """chunk_range=(lo, hi, quantum): variable-length training - one chunk length per batch from
        datasets.chunk_length_schedule (seeded by (seed, epoch) only: the same length on every rank); chunk_size is then
        ignored and the pinned ring is sized for hi.
        device=None: batches are fresh pageable host tensors (the caller copies them; a copy from pageable memory is
        staged by the runtime before .cuda() returns, so nothing can overwrite it early).
        device=cuda:N: the loader owns a ring of PINNED staging buffers and the host->device copy: it issues the copy on
        its own copy stream (overlapping the previous step's kernels), records an event per ring slot, and the reader
        thread waits for that event before it refills the slot - the training loop never syncs with the host, so
        without this guard the reader could overwrite a slot whose asynchronous copy has not executed yet.  Yields
        device tensors, already ordered after the copy on the consumer's current stream."""
        self.device = torch.device(device) if device is not None else None
        utt2spk = {}
        for line in open(utt2spkid_file):
            u, s = line.split()
            utt2spk[u] = int(s)
        count = {}
        for s in utt2spk.values():
            count[s] = count.get(s, 0) + 1
        cap = min(500, int((max(count.values()) + 1) / 2))          # datasets.py:23-24
        rx, lab = [], []
        for line in open(scp_file):
            u, r = line.rstrip().split(None, 1)
            rep = max(1, cap // count[utt2spk[u]])
            rx.extend([r] * rep)
            lab.extend([utt2spk[u]] * rep)
        uniq = sorted(set(rx))
        pos = {r: i for i, r in enumerate(uniq)}
        self.table = ArkTable(uniq)
        self.sample_to_row = np.array([pos[r] for r in rx], dtype=np.int64)
        self.labels = np.array(lab, dtype=np.int64)
        self.chunk_range = tuple(int(v) for v in chunk_range) if chunk_range is not None else None
        self.T = int(chunk_size) if self.chunk_range is None else self.chunk_range[1]
        short = self.table.rows < self.T
        if short.any():
            raise AssertionError("%d utterances are shorter than the chunk size %d (reference: assert len(full_mat) >= seq_len)"
                                 % (int(short.sum()), self.T))
        self.bs, self.rank, self.world, self.seed, self.threads = batch_size, rank, world, seed, threads
        self.drop_last, self.prefetch = drop_last, prefetch
        self.epoch = 0
        print("Totally " + str(len(rx)) + " samples with at most " + str(cap) + " samples for one class")

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _indices(self):
        n = len(self.labels)
        g = np.random.RandomState(self.seed + self.epoch)
        perm = g.permutation(n)
        total = -(-n // self.world) * self.world
        perm = np.concatenate([perm, perm[: total - n]])               # DistributedSampler pads by wrap-around
        return perm[self.rank: total: self.world]

    def __len__(self):
        n = -(-len(self.labels) // self.world)
        return n // self.bs if self.drop_last else -(-n // self.bs)

    def __iter__(self):
        idx = self._indices()
        rng = np.random.RandomState((self.seed + self.epoch) * 7919 + self.rank)
        F = int(self.table.cols[0])
        q = queue.Queue(maxsize=self.prefetch)
        dev = self.device
        nslot = self.prefetch + 2
        ring = [torch.empty(self.bs * F * self.T).pin_memory() for _ in range(nslot)] if dev is not None else None
        copied = [None] * nslot          # per slot: event recorded after the H2D copy that last read the slot
        if self.chunk_range is not None:
            from .datasets import chunk_length_schedule
            lens = chunk_length_schedule(self.chunk_range[0], self.chunk_range[1], self.chunk_range[2], len(self) + 1, self.seed,
                                         self.epoch)
        else:
            lens = None

        def producer():
            try:
                k = 0
                for b0 in range(0, len(idx), self.bs):
                    sel = idx[b0:b0 + self.bs]
                    if len(sel) < self.bs and self.drop_last:
                        break
                    rows = self.sample_to_row[sel]
                    T = self.T if lens is None else int(lens[k])          # one chunk length per batch
                    starts = [int(rng.randint(0, int(self.table.rows[r]) - T + 1)) for r in rows]   # datasets.py:66
                    slot = k % nslot
                    if ring is not None:
                        ev = copied[slot]
                        if ev is not None:
                            ev.synchronize()            # the device has finished reading this pinned buffer
                        buf = ring[slot][:len(sel) * F * T].view(len(sel), F, T)
                    else:
                        buf = torch.empty(len(sel), F, T)
                    self.table.read_crop(rows, starts, T, buf, self.threads)
                    q.put((slot, buf, torch.from_numpy(self.labels[sel])))
                    k += 1
                q.put(None)
            except BaseException as e:      # surface reader errors in the training loop
                q.put(e)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        copy_stream = torch.cuda.Stream(device=dev) if dev is not None else None
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            slot, buf, lab = item
            if dev is None:
                yield buf, lab
                continue
            cur = torch.cuda.current_stream(dev)
            with torch.cuda.stream(copy_stream):
                xg = buf.to(dev, non_blocking=True)
                yg = lab.pin_memory().to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            copied[slot] = ev
            cur.wait_event(ev)
            xg.record_stream(cur)
            yg.record_stream(cur)
            yield xg, yg
        th.join()
