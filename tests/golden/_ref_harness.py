"""Loader for the reference's *own* numeric helpers (this container only).

TEST INFRASTRUCTURE -- used only by ``tests/golden/make_golden.py`` to generate
the committed ``.npz`` fixtures.  Nothing here is imported by the product, by
the ``-m gpu`` tests, by ``bench.py`` or by ``smoke()``; ``/root/reference``
does not exist on the GPU box.

The reference package cannot be imported whole (xarray, dask, numba, cv2,
catalogue, napari ... are not installed), so the two files
that hold the deterministic numeric helpers (``src/magnify/utils.py`` and
``src/magnify/find.py``) are loaded *by path* with inert stand-in modules placed
in ``sys.modules`` first: ``numba.njit/jit`` become identity decorators and
``prange`` becomes ``range`` so the reference's own Python source runs as plain
NumPy.  No reference source is copied: the files are executed where they lie.
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get("MAGNIFY_REFERENCE", "/root/reference")


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


class _Anything:
    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()

    def __or__(self, other):
        return self

    __ror__ = __or__


class _Registry:
    def register(self, name):
        return lambda f: f

    def get(self, name):
        raise KeyError(name)


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """Return (utils, find) modules of the reference, executed in place."""
    sys.dont_write_bytecode = True
    if "magnify.utils" in sys.modules:
        return sys.modules["magnify.utils"], sys.modules["magnify.find"]
    _module("numba", njit=_identity_decorator, jit=_identity_decorator, prange=range)
    _module("cv2")
    nap = _module("napari")
    nap.types = _module("napari.types", LayerDataTuple=tuple)
    nap.settings = _module("napari.settings", get_settings=_Anything())
    nap.Viewer = _Anything()
    nap.run = _Anything()
    _module("magicgui", magicgui=_Anything())
    q = _module("qtpy")
    q.QtCore = _module("qtpy.QtCore", QEventLoop=_Anything(), QTimer=_Anything())
    _module("xarray", Dataset=_Anything(), DataArray=_Anything())
    d = _module("dask")
    d.array = _module("dask.array")
    _module("tqdm", tqdm=lambda it, **k: it)
    _module("catalogue", create=lambda *a, **k: _Registry())
    pkg = _module("magnify")
    pkg.__path__ = [os.path.join(REF, "src", "magnify")]
    pkg.registry = _module("magnify.registry", components=_Registry(), readers=_Registry(),
                           component=lambda name: (lambda f: f))
    plot = _module("magnify.plot")
    plot.__path__ = [os.path.join(REF, "src", "magnify", "plot")]

    def _load(modname, rel):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, "src", "magnify", rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    vis = _load("magnify.plot.vis", os.path.join("plot", "vis.py"))
    plot.vis = vis
    utils = _load("magnify.utils", "utils.py")
    pkg.utils = utils
    find = _load("magnify.find", "find.py")
    pkg.find = find
    return utils, find
