"""Import shim: the package directory is named ``tethys-speech_amd`` (a hyphen is not a
legal module name), so ``import tethys_speech_amd`` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tethys-speech_amd")
_spec = importlib.util.spec_from_file_location(
    "tethys_speech_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tethys_speech_amd"] = _mod
_spec.loader.exec_module(_mod)
