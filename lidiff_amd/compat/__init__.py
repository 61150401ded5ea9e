"""Aliases that let the reference's own Python import this library in place of its third-party wheels.

    import lidiff_amd.compat as compat
    compat.install()                     # before `import lidiff.models.minkunet`

registers in ``sys.modules``:
  ``MinkowskiEngine``      -> lidiff_amd.MinkowskiEngine      (minkunet.py:6, models.py:6, models_refine.py:6,
                                                               tools/diff_completion_pipeline.py:2, train.py:11)
  ``pykeops.torch``        -> lidiff_amd.compat.keops         (minkunet.py:8: ``LazyTensor``)
  ``diffusers``            -> lidiff_amd.compat.diffusers_alias (pipeline:6, models.py:17: ``DPMSolverMultistepScheduler``)
so /root/reference/lidiff/models/minkunet.py runs UNMODIFIED on the HIP kernels (INTEGRATION.md section 2a;
tests/test_reference_exec.py).  Nothing is registered when the real package is importable, unless force=True.
"""
from __future__ import annotations

import importlib
import importlib.util
import sys
import types


def install(force: bool = False) -> list[str]:
    """Returns the names that were aliased."""
    from .. import MinkowskiEngine as ME
    from . import diffusers_alias, keops
    done = []

    def real(top):                       # is the genuine package importable?
        if top in done:
            return False
        try:
            return importlib.util.find_spec(top) is not None
        except (ValueError, ImportError):
            return top in sys.modules

    def put(name, mod):
        if force or (name not in sys.modules and not real(name.split(".")[0])):
            sys.modules[name] = mod
            done.append(name)

    put("MinkowskiEngine", ME)
    put("MinkowskiEngine.utils", ME.utils)
    top = types.ModuleType("pykeops")
    top.torch = keops
    put("pykeops", top)
    put("pykeops.torch", keops)
    put("diffusers", diffusers_alias)
    return done


def uninstall(names):
    for n in names:
        sys.modules.pop(n, None)
