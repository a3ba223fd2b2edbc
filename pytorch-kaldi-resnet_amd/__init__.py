"""MI355X-native speaker-embedding hot path (ResNet-on-fbank + statistics pooling + AAM-softmax).

Python host side mirroring the reference's NeuralSpeakerModel interface over libspkhip.so
(hand-written HIP kernels for gfx950).  Import as `pytorch_kaldi_resnet_amd` (see the shim at the
repository root)."""
__version__ = "0.1.0"
