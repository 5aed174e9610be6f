"""`umbrella` -> `umbrella_amd` alias: lets the reference's front-ends run unchanged.

The reference's examples / app / API server import ``umbrella.speculation.auto_engine``,
``umbrella.models.auto_model``, ``umbrella.api.server`` ... (examples/spec_generate.py:3-4, app/chatbot.py,
umbrella/api/server.py:7).  This package holds no code of its own: every ``umbrella.X`` import is resolved to the
module object of ``umbrella_amd.X`` (same object, so state such as the loaded HIP library is shared).
"""
import importlib
import importlib.abc
import importlib.util
import sys

import umbrella_amd as _impl

_PREFIX, _TARGET = __name__ + ".", _impl.__name__ + "."


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)       # the umbrella_amd module itself

    def exec_module(self, module):
        pass                                               # already executed under its own name


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _TARGET + fullname[len(_PREFIX):]
        try:
            real_spec = importlib.util.find_spec(real)
        except ModuleNotFoundError:
            return None
        if real_spec is None:
            return None
        spec = importlib.util.spec_from_loader(fullname, _AliasLoader(real), is_package=real_spec.submodule_search_locations is not None)
        return spec


sys.meta_path.insert(0, _AliasFinder())
__path__ = []                                               # a package with no files of its own
__all__ = getattr(_impl, "__all__", [])


def __getattr__(name):
    try:
        return getattr(_impl, name)
    except AttributeError:
        pass
    try:
        return importlib.import_module(_PREFIX + name)
    except ModuleNotFoundError as e:                        # hasattr / inspect.unwrap / mock.patch probe with AttributeError
        if e.name not in (_PREFIX + name, _TARGET + name):
            raise                                           # the submodule exists and ITS import failed: a real error
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None
