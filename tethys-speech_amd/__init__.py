"""tethys-speech_amd: MI355X (gfx950) data-parallel speech training step.

Drop-in for the hot path of hyunnnchoi/tethys-speech (speech_jobs/whisper_dist.py):
forward + backward + gradient all-reduce + Adam, as hand-written HIP kernels behind the
C ABI of ``include/tethys_mi.h`` (``libtethys_mi.so``), driven from Python.  PyTorch is
used for device memory, streams and ``torch.distributed`` (RCCL) only.
"""
import os as _os

# The pool's host driver only supports dmabuf IPC: without this RCCL (and any sharing of device tensors between processes)
# fails with "hipIpcGetMemHandle: invalid argument".  Must be in the environment before the HIP runtime initialises; the
# launching shell normally exports it already.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import _lib  # noqa: E402,F401  (does not load the .so until first use)

__all__ = ["_lib"]
