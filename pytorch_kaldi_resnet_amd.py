"""Import shim: the product package lives in the directory `pytorch-kaldi-resnet_amd/` (not a valid Python
identifier), this module loads it under the importable name `pytorch_kaldi_resnet_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pytorch-kaldi-resnet_amd")
_spec = importlib.util.spec_from_file_location(
    "pytorch_kaldi_resnet_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pytorch_kaldi_resnet_amd"] = _mod
_spec.loader.exec_module(_mod)
