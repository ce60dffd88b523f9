"""Glue for running this package next to a checkout of the reference (deepsphere/deepsphere-weather).

Both trees have a regular package called ``modules``; Python resolves a regular package from the FIRST ``sys.path``
entry that has it, so "put ours first, the reference second" alone hides every reference module we do not replace.
Two explicit, side-effect-visible helpers instead:

``merge_reference_package(root)``
    appends ``<root>/modules`` to ``modules.__path__``: the hot-path modules this package provides (``layers``,
    ``models``, ``utils_models``, ``my_models_graph``, ``loss``) keep coming from here, everything else
    (``utils_config``, ``utils_io``, ``predictions_autoregressive`` ...) resolves from the reference checkout.

``load_reference_modules(root, names)``
    loads the REFERENCE's own files for the given submodules (default: its model definitions) under the ``modules.*``
    names, on top of THIS package's ``modules.layers`` - the call-compatibility setup of SURVEY.md 8(a8)/(b): the
    reference's ``UNetSpherical`` then runs unchanged on the HIP kernels.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys


def merge_reference_package(reference_root: str):
    import modules

    ref = os.path.join(reference_root, "modules")
    if not os.path.isdir(ref):
        raise FileNotFoundError(f"{ref} is not a directory")
    if ref not in list(modules.__path__):
        modules.__path__.append(ref)
    return modules


def load_reference_modules(reference_root: str, names=("utils_models", "models", "my_models_graph")):
    """Import ``<reference_root>/modules/<name>.py`` as ``modules.<name>`` for every name (in order), replacing this
    package's counterparts of the same name; ``modules.layers`` stays this package's.  Returns the loaded modules."""
    import modules
    import modules.layers  # noqa: F401  (ours, and it must be in sys.modules before the reference files import it)

    out = []
    for name in names:
        path = os.path.join(reference_root, "modules", name + ".py")
        spec = importlib.util.spec_from_file_location("modules." + name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["modules." + name] = mod
        setattr(modules, name, mod)
        spec.loader.exec_module(mod)
        out.append(mod)
    return out
