#!/usr/bin/env python3
"""Drop-in for the reference's ``speech_jobs/whisper_dist_tensorsize.py`` command line: the same
training run as whisper_dist.py plus the tensor-size / skewness report in
``$TETHYS_WORKSPACE/tensor_logs`` (reference: /workspace/tensor_logs, :23).  The sizes are computed
from the model configuration and the batch shape (tethys-speech_amd/tensorsize.py), not by
instrumenting the step."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import whisper_dist  # noqa: E402


def main(argv=None, model_overrides=None, train_kw=None):
    """``model_overrides`` / ``train_kw`` are for tests (tiny dimensions), as in whisper_dist.main."""
    logs = os.path.join(os.environ.get("TETHYS_WORKSPACE", "/workspace"), "tensor_logs")
    argv = list(sys.argv[1:] if argv is None else argv)
    return whisper_dist.main(argv + ["--tensor_logs", logs], model_overrides=model_overrides, train_kw=train_kw)


if __name__ == "__main__":
    sys.exit(main())
