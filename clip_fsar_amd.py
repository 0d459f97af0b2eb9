"""Import shim: exposes the source directory ``clip-fsar_amd/`` (a hyphen is not
a legal Python identifier) as the importable package ``clip_fsar_amd``.

``import clip_fsar_amd`` executes ``clip-fsar_amd/__init__.py`` in this module's
namespace and points ``__path__`` at that directory, so
``clip_fsar_amd.hip``, ``clip_fsar_amd.models.base.builder`` ... resolve to the
files under ``clip-fsar_amd/``.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "clip-fsar_amd")
__path__ = [_PKG_DIR]
__file__ = _os.path.join(_PKG_DIR, "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
