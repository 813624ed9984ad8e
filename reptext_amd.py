"""Import alias for the product package.

The package directory is named ``arabic-text-image-generation-reptext_amd`` (the project's name), which is not
a valid Python identifier. This module makes it importable as ``reptext_amd``: it points ``__path__`` at that
directory, so ``import reptext_amd.ops`` resolves to ``arabic-text-image-generation-reptext_amd/ops.py``.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "arabic-text-image-generation-reptext_amd")
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"), globals())
