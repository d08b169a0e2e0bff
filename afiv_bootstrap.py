"""Import helper: the package directory name contains hyphens, so it cannot be imported with a
plain ``import`` statement.  ``load()`` registers it in ``sys.modules`` as ``afiv_amd``."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "arbitrage-free-iv-surfaces-using-vae_amd")


def load():
    mod = sys.modules.get("afiv_amd")
    if mod is not None:
        return mod
    spec = importlib.util.spec_from_file_location(
        "afiv_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["afiv_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def dropin_paths():
    """Directories to put on sys.path so that the reference's import statements
    (``from heston_model_ql import HestonModelQL``, ``from cvae_model import ...``) resolve to
    the MI355X implementations."""
    base = os.path.join(_PKG_DIR, "dropin")
    return [base, os.path.join(base, "condtional_vae")]
