"""kaldi_amd: MI355X-native acoustic-likelihood engine behind Kaldi's interfaces.

The compute path is libamlike.so (hand-written gfx950 HIP kernels behind the C
ABI of include/amlike.h).  The Python modules here only marshal arguments.
"""
from . import _lib  # noqa: F401
