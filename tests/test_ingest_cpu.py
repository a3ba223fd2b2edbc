"""Native ingest (libspkio) against the Python reader: bit-exact crops, sampling semantics, error behaviour."""
import os
import re

import numpy as np
import pytest
import torch

import pytorch_kaldi_resnet_amd  # noqa: F401
from pytorch_kaldi_resnet_amd import ingest, kaldi_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _corpus(tmp_path, n_spk=4, per=5, F=12):
    rs = np.random.RandomState(3)
    ark = str(tmp_path / "f.ark")
    scp, u2s, mats = [], [], {}
    with open(ark, "wb") as f:
        for s in range(n_spk):
            for u in range(per if s else 2 * per):      # speaker 0 has twice the utterances -> balancing kicks in
                utt = "s%d-u%d" % (s, u)
                m = rs.randn(rs.randint(30, 50), F).astype(np.float32)
                off = kaldi_io.write_mat(f, m, key=utt)
                scp.append("%s %s:%d" % (utt, ark, off))
                u2s.append("%s %d" % (utt, s))
                mats[utt] = m
    open(str(tmp_path / "f.scp"), "w").write("\n".join(scp) + "\n")
    open(str(tmp_path / "u2s"), "w").write("\n".join(u2s) + "\n")
    return str(tmp_path / "f.scp"), str(tmp_path / "u2s"), mats, scp


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "spkio.h")).read()
    for name in set(re.findall(r"\b(spk_[a-z_]+)\s*\(", hdr)):
        assert hasattr(ingest.lib(), name), name


def test_read_crop_is_bit_exact(tmp_path):
    scp, u2s, mats, lines = _corpus(tmp_path)
    rx = [l.split()[1] for l in lines]
    tab = ingest.ArkTable(rx)
    for i, l in enumerate(lines):
        assert tab.rows[i] == mats[l.split()[0]].shape[0] and tab.cols[i] == 12
    idx = np.array([0, 5, 7, 11, 3])
    starts = [0, 4, 10, 2, int(tab.rows[3]) - 20]
    out = torch.empty(5, 12, 20)
    tab.read_crop(idx, starts, 20, out, nthreads=3)
    for b, (i, s) in enumerate(zip(idx, starts)):
        ref = kaldi_io.read_mat(rx[i])[s:s + 20].T
        np.testing.assert_array_equal(out[b].numpy(), ref)
    with pytest.raises(RuntimeError, match="outside utterance"):
        tab.read_crop(idx[:1], [int(tab.rows[0]) - 5], 20, torch.empty(1, 12, 20))
    with pytest.raises(RuntimeError, match="cannot open"):
        ingest.ArkTable(["/nonexistent/x.ark:0"])
    bad = str(tmp_path / "d.ark")
    with open(bad, "wb") as f:
        off = kaldi_io.write_mat(f, np.zeros((3, 2)), key="k")      # float64 'DM ' is refused by the native reader
    with pytest.raises(RuntimeError, match="float32"):
        ingest.ArkTable(["%s:%d" % (bad, off)])


def test_loader_semantics(tmp_path):
    scp, u2s, mats, lines = _corpus(tmp_path)
    # speaker 0: 10 utts, others 5 -> cap = min(500, (10+1)//2) = 5 -> rep = max(1, 5//10)=1 for spk0, 1 for others
    ld = ingest.NativeTrainLoader(scp, u2s, 16, batch_size=4, seed=1)
    assert len(ld.labels) == 25 and len(ld) == 7
    seen, nb = 0, 0
    for x, y in ld:
        assert x.shape[1:] == (12, 16) and x.dtype == torch.float32 and y.dtype == torch.int64
        seen += x.shape[0]
        nb += 1
    assert seen == 25 and nb == 7
    # two ranks: disjoint shards covering everything (+ wrap-around padding), reshuffled per epoch
    a = ingest.NativeTrainLoader(scp, u2s, 16, 4, rank=0, world=2, seed=1)
    b = ingest.NativeTrainLoader(scp, u2s, 16, 4, rank=1, world=2, seed=1)
    ia, ib = a._indices(), b._indices()
    assert len(ia) == len(ib) == 13 and set(ia) | set(ib) == set(range(25))
    a.set_epoch(1)
    assert not np.array_equal(a._indices(), ia)
    # every crop is a real window of its utterance
    ld2 = ingest.NativeTrainLoader(scp, u2s, 30, batch_size=25, seed=2)
    (x, y), = list(ld2)
    allm = list(mats.values())
    for i in range(25):
        w = x[i].numpy().T
        assert any(any(np.array_equal(w, m[s:s + 30]) for s in range(m.shape[0] - 29)) for m in allm if m.shape[0] >= 30)
    with pytest.raises(AssertionError):
        ingest.NativeTrainLoader(scp, u2s, 60, 4)


def test_text_vector_writer_is_byte_identical_to_numpy_str():
    """libspkio's embedding writer against the reference's line format (scripts/decode.py:206:
    utt + ' [ ' + ' '.join(map(str, row)) + ' ]\\n' with np.float32 elements) over random float32 bit patterns,
    the positional/scientific switch points, signed zero, denormals and non-finite values."""
    from pytorch_kaldi_resnet_amd import ingest
    rs = np.random.RandomState(3)
    x = rs.randint(0, 2 ** 32, size=(300, 256), dtype=np.uint64).astype(np.uint32).view(np.float32).copy()
    x[0, :14] = [0, -0.0, 1e-4, 9.9999e-5, 1.0001e-4, 1e16, 9.999999e15, 123456789.0, np.nan, np.inf, -np.inf, 1e-5, 1.0, 1e-45]
    x[1] = rs.randn(256).astype(np.float32)
    keys = ["spk%03d-utt%d" % (i, i * 7) for i in range(x.shape[0])]
    ref = "".join(k + " [ " + " ".join(map(str, r)) + " ]\n" for k, r in zip(keys, x))
    for nthreads in (1, 5):
        assert ingest.format_text_vectors(keys, x, nthreads).decode() == ref
    one = ingest.format_text_vectors(["a"], np.array([[1.5]], dtype=np.float32))
    assert one == b"a [ 1.5 ]\n"


def test_variable_chunk_lengths_one_per_batch_same_on_every_rank(tmp_path):
    """BASELINE configs[3] (variable-length batches; reference scripts/datasets.py:40-43,53-57): one chunk length per batch,
    drawn from (seed, epoch) only - identical on every rank - in the native loader and through ChunkBatchSampler + the
    Dataset classes (the length travels with the index, so DataLoader workers need no shared state)."""
    from pytorch_kaldi_resnet_amd.datasets import ChunkBatchSampler, SequenceDataset, chunk_length_schedule
    scp, u2s, mats, lines = _corpus(tmp_path)
    sched = chunk_length_schedule(16, 28, 4, 50, seed=3, epoch=0)
    assert set(sched) <= {16, 20, 24, 28} and len(set(sched)) > 1
    assert np.array_equal(sched, chunk_length_schedule(16, 28, 4, 50, seed=3, epoch=0))
    assert not np.array_equal(sched, chunk_length_schedule(16, 28, 4, 50, seed=3, epoch=1))
    # native loader: two ranks see the same length sequence, different samples; every crop is a window of its utterance
    la = ingest.NativeTrainLoader(scp, u2s, 0, 4, rank=0, world=2, seed=3, chunk_range=(16, 28, 4))
    lb = ingest.NativeTrainLoader(scp, u2s, 0, 4, rank=1, world=2, seed=3, chunk_range=(16, 28, 4))
    ta = [x.shape[2] for x, _ in la]
    tb = [x.shape[2] for x, _ in lb]
    assert ta == tb == [int(t) for t in sched[:len(ta)]] and len(ta) == len(la)
    allm = list(mats.values())
    for x, y in la:
        T = x.shape[2]
        assert x.is_contiguous()
        for i in range(x.shape[0]):
            w = x[i].numpy().T
            assert any(any(np.array_equal(w, m[s:s + T]) for s in range(m.shape[0] - T + 1)) for m in allm)
        break
    la.set_epoch(1)
    assert [x.shape[2] for x, _ in la] != ta
    with pytest.raises(AssertionError):                                   # utterances shorter than the longest chunk
        ingest.NativeTrainLoader(scp, u2s, 0, 4, chunk_range=(16, 60, 4))
    # Dataset + batch sampler (DataLoader path), with worker processes
    ds = SequenceDataset(scp, u2s, [28])
    bs = ChunkBatchSampler(torch.utils.data.SequentialSampler(ds), 4, 16, 28, 4, seed=3)
    dl = torch.utils.data.DataLoader(ds, batch_sampler=bs, num_workers=2)
    shapes = [tuple(x.shape) for x, _ in dl]
    assert [s[2] for s in shapes] == [int(t) for t in chunk_length_schedule(16, 28, 4, len(bs), 3, 0)]
    assert all(s[1] == 12 for s in shapes) and sum(s[0] for s in shapes) == len(ds)
    # a plain int index still means "the dataset's own chunk length"
    assert ds[0][0].shape == (12, 28)


def test_vector_ark_loader_matches_the_python_reader(tmp_path):
    """libspkio's vector-ark loader (the scoring back end's input) against kaldi_io.read_vec_flt_ark - the reference's reader
    semantics (scripts/kaldi_io.py:238-290): text 'key [ v ... ]' lines as decode writes them parse to the float64 numpy gives,
    binary FV records come back bit for bit; nan / inf / signed zero / extreme magnitudes included; a ragged or broken ark fails
    loudly.  The vectorised table path of the scoring back end must give the scores of the per-key dict path."""
    from pytorch_kaldi_resnet_amd import scoring, vecark
    rs = np.random.RandomState(3)
    M = (rs.randn(300, 24) * np.exp(rs.randn(300, 24) * 4)).astype(np.float32)
    M[0, :7] = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-38, 3e38]
    keys = ["spk%02d-u%03d" % (i % 7, i) for i in range(len(M))]
    tpath, bpath = str(tmp_path / "t.iv"), str(tmp_path / "b.iv")
    open(tpath, "wb").write(ingest.format_text_vectors(keys, M, 3))
    with open(bpath, "wb") as f:
        for k, v in zip(keys, M):
            kaldi_io.write_vec_flt(f, v, key=k)
    for path in (tpath, bpath):
        for nt in (1, 5):
            t = vecark.load(path, nthreads=nt)
            assert t.keys_list == keys and t.mat.shape == (300, 24) and t.mat.dtype == np.float64
            ref = list(kaldi_io.read_vec_flt_ark(path))
            assert [k for k, _ in ref] == keys
            for i, (_, v) in enumerate(ref):
                assert np.array_equal(np.asarray(v, dtype=np.float64), t.mat[i], equal_nan=True), (path, i)
                assert np.array_equal(t[keys[i]], t.mat[i], equal_nan=True)
    # the reference-written golden file
    g = vecark.load(os.path.join(ROOT, "tests", "golden", "io", "emb.iv"))
    for k, v in kaldi_io.read_vec_flt_ark(os.path.join(ROOT, "tests", "golden", "io", "emb.iv")):
        assert np.array_equal(np.asarray(v, dtype=np.float64), g[k])
    # errors: ragged rows, garbage, missing file
    open(str(tmp_path / "ragged.iv"), "w").write("a [ 1.0 2.0 ]\nb [ 1.0 ]\n")
    open(str(tmp_path / "junk.iv"), "w").write("a [ 1.0 x2 ]\n")
    for bad in ("ragged.iv", "junk.iv", "nope.iv"):
        with pytest.raises(RuntimeError):
            vecark.load(str(tmp_path / bad))
    # table path == dict path of the scoring back end (host)
    fin = vecark.load(str(tmp_path / "b.iv"))
    fin.mat[0, :7] = 0.5
    names = fin.keys_list
    tr = str(tmp_path / "trials")
    with open(tr, "w") as f:
        for _ in range(500):
            i, j = rs.randint(0, len(names), 2)
            f.write("%s %s %s\n" % (names[i], names[j], "target" if names[i][:5] == names[j][:5] else "nontarget"))
    mean = fin.mat.mean(axis=0)
    s_tab, l_tab = scoring.cosine_score(fin, fin, tr, mean)
    plain = {k: np.array(v) for k, v in fin.items()}
    s_dict, l_dict = scoring.cosine_score(plain, plain, tr, mean)
    assert np.array_equal(s_tab, s_dict) and np.array_equal(l_tab, l_dict)
