"""Import alias: ``import sdlcm_amd`` loads the package directory
``stable-diffusion-1.5-lcm-onnx-rknn2_amd/`` (whose name is not a Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stable-diffusion-1.5-lcm-onnx-rknn2_amd")
_spec = importlib.util.spec_from_file_location(
    "sdlcm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sdlcm_amd"] = _mod
_spec.loader.exec_module(_mod)
