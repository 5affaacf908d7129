"""Loader for the `fastintercu-vvc_amd/` package (hyphenated dir => not a plain `import`)."""
import importlib.util
import os
import sys

_NAME = "fastintercu_vvc_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg_dir = os.path.join(ROOT, "fastintercu-vvc_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
