"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, host I/O and scoring
against the reference-generated fixtures, the NeuralSpeakerModel boundary (state_dict naming, loadParameters
semantics, flat arenas), optimizer (de)serialisation, tile selection invariants."""
import io
import json
import os
import re

import numpy as np
import pytest
import torch

import pytorch_kaldi_resnet_amd  # noqa: F401
from pytorch_kaldi_resnet_amd import datasets, hip, kaldi_io, scoring, tiling
from pytorch_kaldi_resnet_amd.model import NeuralSpeakerModel
from pytorch_kaldi_resnet_amd.optim import FlatSGD, cosine_lr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "spkhip.h")).read()
    declared = sorted(set(re.findall(r"\b(spk_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    lib = hip.lib()
    for name in declared:
        assert hasattr(lib, name), "libspkhip.so does not export %s" % name
    assert sorted(hip.exported_symbols()) == declared
    assert lib.spk_version() >= 100
    # argument validation happens on the host before any launch: no GPU needed to see the error convention
    rc = lib.spk_sgd_step(None, None, None, 0, 0.1, 0.9, 0.0, 1.0, 1, None)
    assert rc < 0 and b"spk_sgd_step" in lib.spk_last_error()


def test_flag_constants_agree_between_header_binding_and_kernels():
    """The launch flags live in three places - include/spkhip.h (#define), csrc/spk_common.h (enum) and hip.py (constants): every
    flag the header defines must have the same value in the other two."""
    def value(expr):
        expr = expr.split("/*")[0].strip().rstrip(",")
        assert re.fullmatch(r"[0-9()< ]+", expr), expr
        return eval(expr)
    hdr = open(os.path.join(ROOT, "include", "spkhip.h")).read()
    defines = {m.group(1): value(m.group(2)) for m in re.finditer(r"^#define (SPK_[A-Z0-9_]+)[ \t]+(\(?[0-9][0-9 <()]*)", hdr, re.M)}
    common = open(os.path.join(ROOT, "pytorch-kaldi-resnet_amd", "csrc", "spk_common.h")).read()
    enum = {m.group(1): value(m.group(2)) for m in re.finditer(r"^\s+(SPK_[A-Z0-9_]+) = ([0-9][0-9 <]*),?", common, re.M)}
    flags = {k: v for k, v in defines.items() if k in enum}
    assert len(flags) >= 12, sorted(flags)
    for name, v in flags.items():
        assert enum[name] == v, (name, enum[name], v)
        short = name[4:]
        if hasattr(hip, short):
            assert getattr(hip, short) == v, (name, getattr(hip, short), v)
    for short in ("CONV_PIPE", "CONV_M16", "IN_PRESPLIT", "SIDE_PRESPLIT", "DY_PRESPLIT", "IN_BNBWD", "EPI_BNBWD"):
        assert "SPK_" + short in flags and hasattr(hip, short), short


def test_kaldi_io_reads_reference_written_ark(gold_dir):
    d = os.path.join(gold_dir, "io")
    exp = np.load(os.path.join(d, "feats_expected.npz"))
    cwd = os.getcwd()
    os.chdir(ROOT)   # scp paths are relative to the repository root
    try:
        n = 0
        for key, mat in kaldi_io.read_mat_scp(os.path.join(d, "feats.scp")):
            np.testing.assert_array_equal(mat, exp[key])
            n += 1
        assert n == 6
        got = dict(kaldi_io.read_mat_ark(os.path.join(d, "feats.ark")))
        assert sorted(got) == sorted(exp.files)
    finally:
        os.chdir(cwd)


def test_kaldi_io_round_trips(tmp_path):
    m32 = np.random.RandomState(0).randn(7, 5).astype(np.float32)
    m64 = np.random.RandomState(1).randn(3, 4)
    p = str(tmp_path / "a.ark")
    with open(p, "wb") as f:
        o1 = kaldi_io.write_mat(f, m32, key="u1")
        o2 = kaldi_io.write_mat(f, m64, key="u2")
    np.testing.assert_array_equal(kaldi_io.read_mat("%s:%d" % (p, o1)), m32)
    np.testing.assert_array_equal(kaldi_io.read_mat("ark:%s:%d" % (p, o2)), m64)
    assert [k for k, _ in kaldi_io.read_mat_ark(p)] == ["u1", "u2"]
    v = np.arange(5, dtype=np.float32)
    q = str(tmp_path / "v.ark")
    with open(q, "wb") as f:
        kaldi_io.write_vec_flt(f, v, key="x")
    (k, r), = list(kaldi_io.read_vec_flt_ark(q))
    assert k == "x"
    np.testing.assert_array_equal(r, v)
    # empty ark, text matrix
    open(str(tmp_path / "e.ark"), "wb").close()
    assert list(kaldi_io.read_mat_ark(str(tmp_path / "e.ark"))) == []
    t = kaldi_io.read_mat(io.BytesIO(b" [\n 1 2 3\n 4 5 6 ]\n"))
    np.testing.assert_array_equal(t, np.array([[1, 2, 3], [4, 5, 6]], np.float32))
    with pytest.raises(kaldi_io.UnknownHeader):
        kaldi_io.read_mat(io.BytesIO(b"\0BXX \x04\x00\x00\x00\x00"))


def test_compressed_matrix_decode():
    # hand-built 'CM ' matrix: 2 columns, 3 rows
    import struct
    vmin, vrange, rows, cols = -1.0, 4.0, 3, 2
    hdr = np.array([[0, 16384, 49152, 65535], [0, 0, 65535, 65535]], dtype="<u2")
    data = np.array([[0, 64, 255], [10, 128, 192]], dtype=np.uint8)
    buf = b"\0BCM " + struct.pack("<ffii", vmin, vrange, rows, cols) + hdr.tobytes() + data.tobytes()
    m = kaldi_io.read_mat(io.BytesIO(buf))
    assert m.shape == (3, 2)
    q = vmin + vrange * 1.52590218966964e-05 * hdr.astype(np.float64)
    assert abs(m[0, 0] - q[0, 0]) < 1e-6 and abs(m[1, 0] - q[0, 1]) < 1e-6 and abs(m[2, 0] - q[0, 3]) < 1e-6
    assert abs(m[1, 1] - (q[1, 1] + (q[1, 2] - q[1, 1]) * 64 / 128.0)) < 1e-5


def test_scoring_matches_reference_outputs(gold_dir, tmp_path):
    d = os.path.join(gold_dir, "io")
    mean = scoring.compute_mean(os.path.join(d, "emb.iv"), str(tmp_path / "mean.vec"))
    ref_mean = kaldi_io.read_vec_flt(os.path.join(d, "mean.vec"))
    np.testing.assert_allclose(mean, ref_mean, rtol=1e-6, atol=1e-7)
    assert open(str(tmp_path / "mean.vec")).read().split() == open(os.path.join(d, "mean.vec")).read().split()
    emb = scoring.read_embeddings(os.path.join(d, "emb.iv"))
    scores, labels = scoring.cosine_score(emb, emb, os.path.join(d, "trials"), ref_mean, str(tmp_path / "scores"))
    ref = [float(l.split()[2]) for l in open(os.path.join(d, "scores"))]
    np.testing.assert_allclose(scores, ref, rtol=1e-6, atol=2e-7)
    assert "{0:.2%}".format(scoring.compute_eer(ref, labels)) == open(os.path.join(d, "eer.txt")).read().strip()
    # degenerate inputs: perfectly separable and perfectly wrong
    assert scoring.compute_eer([0.9, 0.8, 0.1, 0.0], [1, 1, 0, 0]) == 0.0
    assert scoring.compute_eer([0.0, 0.1, 0.8, 0.9], [1, 1, 0, 0]) == 1.0


def test_adaptive_snorm_matches_reference_outputs(gold_dir, tmp_path):
    """cohort top-300 statistics and adaptive S-norm against the files written by the reference's own scripts
    (scripts/compute_topk_mean_std.py, scripts/adaptive_snorm.py; tools/make_golden.py io)."""
    d = os.path.join(gold_dir, "io")
    mean = kaldi_io.read_vec_flt(os.path.join(d, "mean.vec"))
    emb = scoring.read_embeddings(os.path.join(d, "emb.iv"))
    coh = scoring.read_embeddings(os.path.join(d, "cohort.iv"))
    assert len(coh) == 320
    stats = scoring.topk_mean_std(emb, coh, mean, 300)
    ref = scoring.read_mean_std(os.path.join(d, "topk_mean_std"))
    assert list(stats) == list(ref)
    for k in ref:
        np.testing.assert_allclose(stats[k], ref[k], rtol=2e-6, atol=1e-7)
    scoring.write_mean_std(stats, str(tmp_path / "ms"))
    back = scoring.read_mean_std(str(tmp_path / "ms"))
    assert all(abs(back[k][0] - float(stats[k][0])) < 1e-9 for k in stats)
    # the S-norm itself is Python-float arithmetic on the three text files: byte-identical output
    scoring.adaptive_snorm(ref, ref, os.path.join(d, "scores"), str(tmp_path / "snorm"))
    assert open(str(tmp_path / "snorm")).read() == open(os.path.join(d, "scores_snorm")).read()
    # the reference's topk(300) raises when the cohort is smaller than k
    with pytest.raises(RuntimeError):
        scoring.topk_mean_std(emb, emb, mean, 300)
    # zero-variance guard of adaptive_snorm.py:33-34 (max(std, 1e-8))
    flat = {k: (0.0, 0.0) for k in ref}
    out = scoring.adaptive_snorm(flat, flat, os.path.join(d, "scores"))
    first = float(open(os.path.join(d, "scores")).readline().split()[2])
    assert out[0] == first / 1e-8 / 2 + first / 1e-8 / 2


def test_datasets_balance_and_crop(gold_dir, tmp_path):
    d = os.path.join(gold_dir, "io")
    u2s = str(tmp_path / "utt2spkid")
    utts = [l.split()[0] for l in open(os.path.join(d, "feats.scp"))]
    # speaker 0 has 2 utts, speaker 1 has 4 -> cap = min(500, (4+1)//2) = 2 -> rep(spk0) = 1, rep(spk1) = max(1, 2//4) = 1
    with open(u2s, "w") as f:
        for i, u in enumerate(utts):
            f.write("%s %d\n" % (u, 0 if i < 2 else 1))
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        ds = datasets.SequenceDataset(os.path.join(d, "feats.scp"), u2s, [16])
        assert len(ds) == 6
        np.random.seed(0)
        x, y = ds[3]
        assert x.shape == (8, 16) and x.dtype == np.float32 and int(y) == 1 and x.flags["C_CONTIGUOUS"]
        ds2 = datasets.SequenceDataset2(os.path.join(d, "feats.scp"), u2s, 12)
        assert len(ds2) == 2 * 2 and ds2[1][0].shape == (8, 12)
        ed = datasets.EmbeddingDataset(os.path.join(d, "feats.scp"))
        x, utt = ed[2]
        assert utt == utts[2] and x.shape == (8, 26)
        with pytest.raises(AssertionError):
            datasets.SequenceDataset(os.path.join(d, "feats.scp"), u2s, [999])[0]
    finally:
        os.chdir(cwd)


def test_datasets_match_the_reference_classes(gold_dir):
    """SequenceDataset / SequenceDataset2 against fixtures produced by the reference's own classes
    (tools/make_dataset_golden.py: imported scripts/datasets.py with the `numpy.int = int` shim): class-balanced
    repetition (scripts/datasets.py:23-31), sample order, the per-sample length draw, and - with the global numpy RNG
    seeded the same way - the very same crops (utterance draw, then crop start: :128-137, :64-68)."""
    g = np.load(os.path.join(gold_dir, "datasets.npz"))
    scp, u2s = os.path.join(gold_dir, "io", "unbalanced.scp"), os.path.join(gold_dir, "io", "unbalanced.utt2spkid")
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        ds = datasets.SequenceDataset(scp, u2s, [16])
        assert len(ds) == int(g["v1_len"])
        np.testing.assert_array_equal(ds.labels, g["v1_labels"])
        assert [str(r) for r in ds.rxfiles] == [str(r) for r in g["v1_rxfiles"]]
        for i in g["v1_idx"]:
            np.random.seed(100 + int(i))
            x, y = ds[int(i)]
            np.testing.assert_array_equal(x, g["v1_x%d" % i])
            assert int(y) == int(g["v1_y%d" % i]) and x.dtype == np.float32
        np.random.seed(5)
        dsv = datasets.SequenceDataset(scp, u2s, [12, 20])
        np.testing.assert_array_equal(dsv.seq_len, g["v1_var_seq_len"])
        ds2 = datasets.SequenceDataset2(scp, u2s, 14)
        assert len(ds2) == int(g["v2_len"]) and ds2.repetition == int(g["v2_repetition"])
        np.testing.assert_array_equal(ds2.labels, g["v2_labels"])
        for i in g["v2_idx"]:
            np.random.seed(200 + int(i))
            x, y = ds2[int(i)]
            np.testing.assert_array_equal(x, g["v2_x%d" % i])
            assert int(y) == int(g["v2_y%d" % i])
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("loss,arch", [("AAM", "resnet34"), ("softmax", "resnet34"), ("AAM-v1", "resnet34"),
                                       ("AAM", "resnet101")])
def test_state_dict_matches_reference_keys(gold_dir, loss, arch):
    keys = json.load(open(os.path.join(gold_dir, "state_keys_%s_%s.json" % (arch, loss))))
    m = NeuralSpeakerModel(7, 80, "mean+std", loss, arch=arch)
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _ in keys]
    assert [list(v.shape) for v in sd.values()] == [s for _, s in keys]
    if arch == "resnet34":
        assert len(sd) == {"AAM": 219, "softmax": 225, "AAM-v1": 224}[loss]


def test_flat_arena_and_load_parameters(capsys):
    m = NeuralSpeakerModel(5, 40, "mean", "softmax")
    flat = m.flat_parameters()
    assert flat.numel() >= sum(p.numel() for p in m.parameters())
    for p, o in zip(m.parameters(), m._offsets):
        assert p.data_ptr() == flat.data_ptr() + 4 * o and o % 4 == 0
    # parameters are views: writing the arena changes the parameter
    flat.zero_()
    assert float(m.fc1.weight.abs().sum()) == 0.0
    # loadParameters: 'module.' prefix stripped, unknown and mismatching entries reported and skipped
    st = {"module." + k: torch.ones_like(v) for k, v in m.state_dict().items()}
    st["module.bogus"] = torch.zeros(1)
    st["module.last.weight"] = torch.zeros(3, 256)
    m.loadParameters(st)
    out = capsys.readouterr().out
    assert "module.bogus is not in the model." in out
    assert "Wrong parameter length: module.last.weight" in out
    assert float(m.fc1.weight.min()) == 1.0 and float(m.last.weight.abs().sum()) == 0.0
    assert float(flat[:10].sum()) == 10.0     # still the same arena
    with pytest.raises(RuntimeError):
        m.predict(torch.zeros(1, 40, 64))     # no CPU path
    with pytest.raises(NotImplementedError):
        NeuralSpeakerModel(5, 40, "mean", "nope")


def test_flat_sgd_state_dict_speaks_torch_sgd():
    m = NeuralSpeakerModel(4, 40, "mean", "AAM")
    opt = FlatSGD(m, 0.1, momentum=0.9, weight_decay=5e-4)
    ref = torch.optim.SGD(m.parameters(), 0.1, momentum=0.9, weight_decay=5e-4)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    ref.step()
    sd = ref.state_dict()
    opt.load_state_dict(sd)
    back = opt.state_dict()
    assert set(back["state"].keys()) == set(sd["state"].keys())
    for k in sd["state"]:
        assert torch.equal(back["state"][k]["momentum_buffer"], sd["state"][k]["momentum_buffer"])
    assert back["param_groups"][0]["lr"] == 0.1 and back["param_groups"][0]["params"] == sd["param_groups"][0]["params"]
    ref2 = torch.optim.SGD(m.parameters(), 0.5)
    ref2.load_state_dict(back)               # torch accepts what FlatSGD emits
    assert ref2.param_groups[0]["momentum"] == 0.9
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 30, eta_min=1e-4)
    for e in range(5):
        assert abs(opt.param_groups[0]["lr"] - cosine_lr(e, 30, 0.1, 1e-4)) < 1e-9
        sch.step()


@pytest.mark.parametrize("shape", [(80, 300), (40, 150), (20, 75), (10, 38), (10, 25), (5, 13), (1, 1), (80, 1000), (3, 7)])
def test_tile_selection_invariants(shape):
    OH, OW = shape
    for IS, ks, ntaps in [(1, 3, 9), (2, 3, 9), (2, 1, 1), (1, 1, 1), (1, 2, 4)]:
        for Cout in (32, 64, 256):
            TH, TW, MT, NT = tiling.conv_tile(OH, OW, IS, ks, ks, ntaps, Cout)
            assert 1 <= TH and 1 <= TW and TH * TW <= 128 * MT and MT in (1, 2, 3, 4) and Cout % (32 * NT) == 0
            assert ((TH - 1) * IS + ks) * ((TW - 1) * IS + ks) * 144 <= 160 * 1024
    for k, s in [(3, 1), (3, 2), (1, 2), (1, 1)]:
        TH, TW, WN = tiling.wgrad_tile(OH, OW, 64, 64, k, s)
        assert TW % 2 == 0 and WN in (1, 2, 4)
        import ctypes
        mh, mt = ctypes.c_int(), ctypes.c_int()
        assert hip.lib().spk_conv_wgrad_limits(WN, ctypes.byref(mh), ctypes.byref(mt)) == 0
        assert ((TH - 1) * s + k) * ((TW - 1) * s + k) <= mh.value == tiling.WGRAD_MAX_HALO
        assert TH * TW <= mt.value == tiling.WGRAD_MAX_TILE[WN]


def test_tile_lookup_by_mode_and_operand_mode(monkeypatch):
    """Host-side tile selection: the measured table wins over the heuristic, the fused-BN-backward mode and the
    bf16-split operand mode have their own keys with fall-back to the plain entry, every result fits the kernel limits."""
    from pytorch_kaldi_resnet_amd import tiling
    key = (20, 75, 1, 3, 3, 9, 128)
    assert tiling.conv_tile(*key) == tiling.FORCE_CONV[key]
    assert tiling.conv_tile(*key, mode=1) == tiling.FORCE_CONV.get(key + (1,), tiling.FORCE_CONV[key])
    assert tiling.conv_tile(*key, split=6) == tiling.FORCE_CONV_SPLIT.get(key, tiling.FORCE_CONV[key])
    monkeypatch.setitem(tiling.FORCE_CONV_SPLIT, key, (10, 25, 2, 2))
    monkeypatch.setitem(tiling.FORCE_CONV_SPLIT, key + (1,), (5, 25, 1, 2))
    assert tiling.conv_tile(*key, split=6) == (10, 25, 2, 2)
    assert tiling.conv_tile(*key, mode=1, split=6) == (5, 25, 1, 2)
    assert tiling.conv_tile(*key) == tiling.FORCE_CONV[key]
    for shape in [(7, 9, 1, 3, 3, 9, 32), (33, 77, 2, 3, 3, 9, 64), (5, 3, 1, 1, 1, 1, 256), (80, 401, 1, 3, 3, 9, 32)]:
        TH, TW, MT, NT = tiling.conv_tile(*shape)
        assert TH * TW <= 128 * MT and shape[6] % (32 * NT) == 0
        halo = ((TH - 1) * shape[2] + shape[3]) * ((TW - 1) * shape[2] + shape[4])
        assert halo * tiling.LDS_PIX_BYTES <= tiling.LDS_HARD
    # weight-gradient tiles: the split kernel has its own measured table (tiles of 64 pixels = four 16-pixel MFMA steps)
    wkey = (20, 75, 128, 128, 3, 1)
    assert tiling.wgrad_tile(*wkey) == tiling.FORCE_WGRAD[wkey]
    assert tiling.wgrad_tile(*wkey, split=6) == tiling.FORCE_WGRAD_SPLIT.get(wkey, tiling.FORCE_WGRAD[wkey])
    for key, (TH, TW, WN) in list(tiling.FORCE_WGRAD.items()) + list(tiling.FORCE_WGRAD_SPLIT.items()):
        k, st = key[4], key[5]
        assert TH * TW <= tiling.WGRAD_MAX_TILE[WN] and ((TH - 1) * st + k) * ((TW - 1) * st + k) <= tiling.WGRAD_MAX_HALO and TW % 2 == 0
    for c in tiling.conv_candidates(20, 75, 1, 3, 3, 9, 128, split=6):
        assert (c[2], c[3]) in ((1, 1), (2, 1), (3, 1), (4, 1), (1, 2), (2, 2), (3, 2), (1, 4))


def test_operand_modes_are_declared():
    from pytorch_kaldi_resnet_amd import ops
    assert ops.MFMA_MODES == {"f32": 0, "bf16x6": 6, "bf16x9": 9, "f16x3": 3}
    assert ops.SPLIT in ops.MFMA_MODES.values()
    # 1x1 convolutions use fp32 operands in the bf16-term modes and the fp16 terms in f16x3
    assert ops.split_for(1) == (3 if ops.SPLIT == 3 and ops.SPLIT_1X1 else 0) and ops.split_for(3) == ops.SPLIT
    assert ops.split_for(3, bwd=True) == (ops.SPLIT if ops.SPLIT_BWD is None else ops.SPLIT_BWD)
    import torch
    w = torch.zeros(64, 32, 3, 3)
    old = (ops.SPLIT, ops.SPLIT_BWD)
    try:
        for mode, n in (("f32", w.numel()), ("bf16x6", w.numel() * 3 // 2), ("f16x3", w.numel() + 4)):
            ops.SPLIT, ops.SPLIT_BWD = ops.MFMA_MODES[mode], None
            assert ops.packed_numel(w) == n and ops.packed_numel(w, bwd=True) == n       # 4 / 6 / 4 (+16 header) bytes per weight
        ops.SPLIT, ops.SPLIT_BWD = 3, 6       # forward on fp16 terms, gradients on bf16 terms: the two packs differ
        assert ops.packed_numel(w) == w.numel() + 4 and ops.packed_numel(w, bwd=True) == w.numel() * 3 // 2
    finally:
        ops.SPLIT, ops.SPLIT_BWD = old


def test_lds_bank_model_of_the_weight_gradient_x_image():
    """tools/lds_bank_model.py restates the lane -> LDS address formulas of conv_wgrad_wm16_kernel's transposed reads and the bank
    rule of ds_read_b64_tr_b16 (per 32-lane half, 64 banks of 4 bytes).  The kernel's X layout must keep the tiles of the training
    step that are 8 or 16 pixels wide free of bank conflicts on their X reads (the first layout - pixel pitch 384 / 192 bytes - read
    them four / two deep: SQ_LDS_BANK_CONFLICT 9.2e7 of 1.30e8 LDS cycles per launch, profiles/r04_sq_counters), and the constants the
    model uses must be the ones in the source."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_bank_model", os.path.join(os.path.dirname(__file__), "..", "tools", "lds_bank_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    src = open(os.path.join(os.path.dirname(__file__), "..", "pytorch-kaldi-resnet_amd", "csrc", "conv_wgrad_wm16.hip")).read()
    assert "#define WM16_PX_WIDE (WM16_ODD_PITCH ? 416 : 384)" in src and "#define WM16_PX_C32 (WM16_ODD_PITCH ? 160 : 192)" in src
    assert "#define WM16_ODD_PITCH 1" in src
    for TH, TW, c32 in ((8, 8, False), (4, 16, False), (8, 16, True)):
        for j in range(-(-(TH * TW) // 32)):
            for t in range(9):
                assert m.x_read_cycles(TH, TW, 1, True, c32, j, (t // 3, t % 3)) == 4          # one cycle per half, two reads
        assert m.x_read_cycles(TH, TW, 1, False, c32) == (8 if c32 else 16)                      # the first layout: two / four deep
    assert m.dy_read_cycles(True, False) == 8 and m.dy_read_cycles(True, True) == 8              # dY: two deep by construction of the DMA image
    assert abs(m.kstep_ratio(8, 8, 1, False, False) - 3.8) < 1e-9 and abs(m.kstep_ratio(8, 8, 1, True, False) - 1.1) < 1e-9


def test_lds_bank_model_of_the_pipelined_convolution():
    """The same tool restates conv_pipe_kernel's 16x16x32 form: the lane -> LDS address of an A-fragment ds_read_b128 (lbase[] of
    conv_body, pix_of_row16, pixel pitch 80 bytes) and of a staging store.  A row tile of 16 consecutive pixels whose pixels sit in
    two tile rows reads two deep (the halo's two extra pixels shift the second row by 10 granules, so pixel 15 lands on pixel 1's
    bank group): 1.79 x the conflict-free cycles on 20 x 19 tiles, 1.38 x on 10 x 38 - with the staging stores and the epilogue slab
    the model lands within 1 / 7 / 3 % of the measured SQ_INSTS_LDS / SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT of a launch
    (profiles/r04_sq_counters/final_kernels_after_odd_pitch_lds_vmem.md).  Six more granules per halo row (96 bytes: shift = 16
    granules) is the only row pad below 16 that removes the read conflicts; the constants the model uses must be the source's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_bank_model", os.path.join(os.path.dirname(__file__), "..", "tools", "lds_bank_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    src = open(os.path.join(os.path.dirname(__file__), "..", "pytorch-kaldi-resnet_amd", "csrc", "conv_kernel.h")).read()
    assert "static constexpr int LP4 = SPLIT ? (NTERM * CK * 2 + 16) / 16 : 9;" in src and "#define SPK_SPLIT_CK 16" in src     # 2 * 16 * 2 + 16 = 80 bytes
    assert "const int q = (row + 12) & 15;" in src and "return ((q & 7) << 1) | (q >> 3);" in src
    assert "lbase[i] = ((ly * a.IS) * a.halo_w + lx * a.IS) * LP4 + 2 * ((lane >> 4) & 1);" in src
    assert "uint2* dst = (uint2*)plane + p * (LP4 * 2) + (qd >> 1) * 4 + (qd & 1);" in src
    assert m.PIPE_LP4 == 5 and sorted(m.pix_of_row16(r) for r in range(16)) == list(range(16))
    # the tiles the step launches this kernel with are the ones the model walks
    import pytorch_kaldi_resnet_amd  # noqa: F401
    from pytorch_kaldi_resnet_amd import tiling
    for (OH, OW, C), (_, _, TH, TW, cin, blocks) in zip(((40, 150, 64), (20, 75, 128), (10, 38, 256)), m.PIPE_LAUNCHES):
        assert tiling.conv_tile(OH, OW, 1, 3, 3, 9, C, mode=0, split=3) == (TH, TW, 3, 2) and cin == C
        assert blocks == 256 * -(-OH // TH) * -(-OW // TW) * (C // 64)
    # a row tile inside one tile row is conflict-free, one that straddles two rows is two deep in every lane group
    assert m.pipe_a_read_cycles(20, 19, 1, 0) == 4 and m.pipe_a_read_cycles(20, 19, 1, 1) == 8
    assert abs(m.pipe_a_read_ratio(20, 19) - 1.7917) < 1e-3 and abs(m.pipe_a_read_ratio(10, 38) - 1.375) < 1e-9
    assert [p for p in range(16) if m.pipe_a_read_ratio(20, 19, row_pad=p) < 1.1] == [6]
    assert [p for p in range(16) if m.pipe_a_read_ratio(10, 38, row_pad=p) < 1.1] == [6]
    assert m.pipe_stage_write_cycles() == 16                                  # 80-byte pixels: 4 pixels of a 16-lane group, two deep
    insts, cycles, conflicts, read_conf = m.pipe_launch_model()
    assert abs(insts / 5.288e6 - 1) < 0.02 and abs(cycles / 4.007e7 - 1) < 0.08 and abs(conflicts / 1.654e7 - 1) < 0.04
    assert 0.55 < read_conf / conflicts < 0.65                                 # the A reads are six tenths of the conflict cycles
    # the padded halo rows still fit two blocks per CU (2 slots + the dump pixel <= 80 KB), and the address arithmetic of the prepared
    # variant (tools/patches/pipe_row_pad.patch: store, read base, tap offsets) is consistent with and without the pad
    for TH, TW in ((20, 19), (10, 38)):
        assert 2 * (TH + 2) * ((TW + 2) * 80 + 96) + 80 <= 80 * 1024
        assert m.pipe_row_pad_consistent(TH, TW, 1, 0) == (TH + 2) * (TW + 2) * 80
        assert m.pipe_row_pad_consistent(TH, TW, 1, 6) == (TH + 2) * ((TW + 2) * 80 + 96)
    patch = open(os.path.join(os.path.dirname(__file__), "..", "tools", "patches", "pipe_row_pad.patch")).read()
    for line in ("lbase[i] += ly * a.IS * PIPE_ROW_PAD;", "dst += (int)__umulhi((unsigned)p, a.halo_w_magic) * (PIPE_ROW_PAD * 2);",
                 "a.tap_off[t] += (tap_dy[t] - mindy) * PIPE_ROW_PAD;", "lds2 += (size_t)2 * a.halo_h * PIPE_ROW_PAD * 16;"):
        assert line in patch
