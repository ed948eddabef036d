"""tethys-speech_amd: MI355X (gfx950) data-parallel speech training step.

Drop-in for the hot path of hyunnnchoi/tethys-speech (speech_jobs/whisper_dist.py):
forward + backward + gradient all-reduce + Adam, as hand-written HIP kernels behind the
C ABI of ``include/tethys_mi.h`` (``libtethys_mi.so``), driven from Python.  PyTorch is
used for device memory, streams and ``torch.distributed`` (RCCL) only.
"""
from . import _lib  # noqa: F401  (does not load the .so until first use)

__all__ = ["_lib"]
