#!/usr/bin/env python3
"""Drop-in for the reference's ``stable_jobs/whisper_dist.py`` command line: that file is byte-identical to
``speech_jobs/whisper_dist.py``, so this is the same entry point."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from speech_jobs.whisper_dist import main  # noqa: E402,F401

if __name__ == "__main__":
    sys.exit(main())
