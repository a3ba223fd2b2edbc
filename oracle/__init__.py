"""CPU oracle for the ResNet-on-fbank speaker-embedding hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``pytorch-kaldi-resnet_amd``) never imports this package and
fails loudly when its HIP library is missing.

The oracle is a torch-CPU fp32 restatement (own code, functional style) of the
reference algorithm in ``scripts/model.py``, ``scripts/train_resnet.py`` and
``scripts/decode.py`` of ZihanLiao/pytorch-kaldi-resnet.  Parity pinning: the
reference holds no tests or golden vectors, so the
restatement is pinned against outputs of the reference itself, generated in the
build container by ``tools/make_golden.py`` (which imports the reference's
``scripts/model.py`` on CPU) and committed as arrays under ``tests/golden/``.
"""
