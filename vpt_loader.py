"""Import helper: the package directory is named `volumetric-path-tracer_amd` (a dash is not a
valid Python identifier), so it is loaded by path and registered as module `vpt_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PACKAGE_DIR = os.path.join(_ROOT, "volumetric-path-tracer_amd")


def load():
    if "vpt_amd" in sys.modules:
        return sys.modules["vpt_amd"]
    spec = importlib.util.spec_from_file_location(
        "vpt_amd", os.path.join(PACKAGE_DIR, "__init__.py"), submodule_search_locations=[PACKAGE_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["vpt_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
