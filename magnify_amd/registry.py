"""Component / reader registries and the predefined pipelines.

Mirrors the reference's plugin API (src/magnify/registry.py): ``readers`` and ``components`` map a
name to a *factory*; ``factory(**kwargs)`` returns ``callable(Dataset) -> Dataset``.  Two
registration idioms exist and both are kept: ``@component("name")`` on ``f(xp, **kw)``
(registry.py:16-29) and ``@components.register("name")`` on a ``make(**kw)`` that returns a callable
object (stitch.py:48-50, find.py:404-442, 607-629).  The pipeline builders keep the reference's
keyword names and defaults verbatim (registry.py:32-59, 196-222, 454-470, 568-583, 615-621, 672-677).
"""
from __future__ import annotations

import functools
import inspect


class Registry:
    """Minimal stand-in for a ``catalogue`` registry (register / get)."""

    def __init__(self, *namespace):
        self.namespace = namespace
        self._items = {}

    def register(self, name, func=None):
        def deco(f):
            self._items[name] = f
            return f

        return deco(func) if func is not None else deco

    def get(self, name):
        try:
            return self._items[name]
        except KeyError:
            avail = ", ".join(sorted(self._items))
            raise KeyError(f"Cant't find '{name}' in registry {' -> '.join(self.namespace)}. Available names: {avail}") from None

    def get_all(self):
        return dict(self._items)

    def __contains__(self, name):
        return name in self._items


readers = Registry("magnify", "readers")
components = Registry("magnify", "components")


class _Factory:
    """What ``component(name)`` puts into the registry for ``func(xp, **settings)``: calling it with the settings
    gives the pipeline step ``step(xp)``; it presents ``func``'s name, docstring and signature without the dataset
    parameter, which is what ``Pipeline.add_pipe(name, **settings)`` and introspection see (registry.py:16-29)."""

    def __init__(self, func):
        self._func = func
        functools.update_wrapper(self, func)
        dataset_param, *settings = inspect.signature(func).parameters.values()
        self.__signature__ = inspect.Signature(settings)

    def __call__(self, *args, **settings):
        func = self._func

        def step(xp, *more, **late):
            return func(*args, xp, *more, **{**settings, **late})

        step.__name__ = getattr(func, "__name__", "step")
        step.func, step.args, step.keywords = func, args, settings  # (what functools.partial would expose)
        return step


def component(name):
    """Decorator: register ``func(xp, **settings)`` under ``name``; ``func`` itself is returned unchanged, so a
    module can keep calling it directly."""

    def register(func):
        components.register(name)(_Factory(func))
        return func

    return register


from .pipeline import Pipeline  # noqa: E402  (Pipeline looks the registries up lazily)


def microfluidic_chip(data, shape=(8, 8), pinlist=None, blank=None, overlap=102, rotation=0, row_dist=375 / 1.61,
                      col_dist=400 / 1.61, chip_type=None, min_button_diameter=8, max_button_diameter=30,
                      chamber_diameter=60, top_chamber=None, left_chamber=None, low_edge_quantile=0.1,
                      high_edge_quantile=0.9, num_iter=5000000, min_roundness=0.2, cluster_penalty=50, roi_length=None,
                      progress_bar=False, search_timestep=0, search_channel=None, roi_only=False, drop_tiles=True,
                      interactive=False):
    """registry.py:32-110."""
    kw = dict(locals())
    kw.pop("data")
    return microfluidic_chip_pipe(**kw)(data=data)


def microfluidic_chip_pipe(shape=(8, 8), pinlist=None, blank=None, overlap=102, rotation=0, row_dist=375 / 1.61,
                           col_dist=400 / 1.61, chip_type=None, min_button_diameter=8, max_button_diameter=30,
                           chamber_diameter=60, top_chamber=None, left_chamber=None, low_edge_quantile=0.1,
                           high_edge_quantile=0.9, num_iter=5000000, min_roundness=0.2, cluster_penalty=50,
                           roi_length=None, progress_bar=False, search_timestep=0, search_channel=None,
                           roi_only=False, drop_tiles=True, interactive=False):
    """registry.py:196-271."""
    if chip_type is not None:
        if chip_type == "minichip":
            row_dist, col_dist = 375 / 1.61, 400 / 1.61
        elif chip_type == "pc":
            row_dist, col_dist = 406 / 3.22, 750 / 3.22
        elif chip_type == "ps":
            row_dist, col_dist = 375 / 3.22, 655 / 3.22
        else:
            raise ValueError(f"Invalid chip type: {chip_type}. Must be one of ['pc', 'ps', 'minichip']")
    pipe = Pipeline("read")
    pipe.add_pipe("standardize_format")
    pipe.add_pipe("identify_buttons", shape=shape, pinlist=pinlist, blank=blank)
    pipe.add_pipe("stitch", overlap=overlap)
    pipe.add_pipe("rotate", rotation=rotation)
    pipe.add_pipe("find_buttons", row_dist=row_dist, col_dist=col_dist, min_button_diameter=min_button_diameter,
                  max_button_diameter=max_button_diameter, chamber_diameter=chamber_diameter, top_chamber=top_chamber,
                  left_chamber=left_chamber, low_edge_quantile=low_edge_quantile,
                  high_edge_quantile=high_edge_quantile, num_iter=num_iter, min_roundness=min_roundness,
                  cluster_penalty=cluster_penalty, roi_length=roi_length, progress_bar=progress_bar,
                  search_timestep=search_timestep, search_channel=search_channel, interactive=interactive)
    pipe.add_pipe("drop", roi_only=roi_only, drop_tiles=drop_tiles)
    pipe.add_pipe("restore_format")
    return pipe


def mrbles(data, spectra, codes, flatfield=1.0, darkfield=0.0, overlap=102, min_bead_diameter=10, max_bead_diameter=50,
           low_edge_quantile=0.1, high_edge_quantile=0.9, num_iter=5000000, min_roundness=0.3, roi_length=None,
           search_channel=None, reference="eu", roi_only=False, drop_tiles=True, interactive=False):
    """registry.py:274-399."""
    kw = dict(locals())
    kw.pop("data")
    return mrbles_pipe(**kw)(data=data)


def mrbles_pipe(spectra, codes, flatfield=1.0, darkfield=0.0, overlap=102, min_bead_diameter=10, max_bead_diameter=50,
                low_edge_quantile=0.1, high_edge_quantile=0.9, num_iter=5000000, min_roundness=0.3, roi_length=None,
                search_channel=None, reference="eu", roi_only=False, drop_tiles=True, interactive=False):
    """registry.py:402-451."""
    pipe = Pipeline("read")
    pipe.add_pipe("standardize_format")
    pipe.add_pipe("flatfield_correct", flatfield=flatfield, darkfield=darkfield)
    pipe.add_pipe("stitch", overlap=overlap)
    pipe.add_pipe("find_beads", min_bead_diameter=min_bead_diameter, max_bead_diameter=max_bead_diameter,
                  low_edge_quantile=low_edge_quantile, high_edge_quantile=high_edge_quantile, num_iter=num_iter,
                  min_roundness=min_roundness, roi_length=roi_length, search_channel=search_channel,
                  interactive=interactive)
    pipe.add_pipe("identify_mrbles", spectra=spectra, codes=codes, reference=reference)
    pipe.add_pipe("drop", roi_only=roi_only, drop_tiles=drop_tiles)
    pipe.add_pipe("restore_format")
    return pipe


def beads(data, flatfield=1.0, darkfield=0.0, overlap=102, min_bead_diameter=10, max_bead_diameter=50,
          low_edge_quantile=0.1, high_edge_quantile=0.9, num_iter=5000000, min_roundness=0.3, roi_length=None,
          search_channel=None, roi_only=False, drop_tiles=True, interactive=False):
    """registry.py:454-565."""
    kw = dict(locals())
    kw.pop("data")
    return beads_pipe(**kw)(data=data)


def beads_pipe(flatfield=1.0, darkfield=0.0, overlap=102, min_bead_diameter=5, max_bead_diameter=25,
               low_edge_quantile=0.1, high_edge_quantile=0.9, num_iter=5000000, min_roundness=0.3, roi_length=None,
               search_channel=None, roi_only=False, drop_tiles=True, interactive=False):
    """registry.py:568-612 (note the 5/25 defaults here versus 10/50 in ``beads``)."""
    pipe = Pipeline("read")
    pipe.add_pipe("standardize_format")
    pipe.add_pipe("flatfield_correct", flatfield=flatfield, darkfield=darkfield)
    pipe.add_pipe("stitch", overlap=overlap)
    pipe.add_pipe("find_beads", min_bead_diameter=min_bead_diameter, max_bead_diameter=max_bead_diameter,
                  low_edge_quantile=low_edge_quantile, high_edge_quantile=high_edge_quantile, num_iter=num_iter,
                  min_roundness=min_roundness, roi_length=roi_length, search_channel=search_channel,
                  interactive=interactive)
    pipe.add_pipe("drop", roi_only=roi_only, drop_tiles=drop_tiles)
    pipe.add_pipe("restore_format")
    return pipe


def image(data, overlap=102, rotation=0, roi_only=False, drop_tiles=True):
    """registry.py:615-669."""
    return image_pipe(overlap=overlap, rotation=rotation, roi_only=roi_only, drop_tiles=drop_tiles)(data=data)


def image_pipe(overlap=102, rotation=0, roi_only=False, drop_tiles=True):
    """registry.py:672-693."""
    pipe = Pipeline("read")
    pipe.add_pipe("standardize_format")
    pipe.add_pipe("stitch", overlap=overlap)
    pipe.add_pipe("rotate", rotation=rotation)
    pipe.add_pipe("drop", roi_only=roi_only, drop_tiles=drop_tiles)
    pipe.add_pipe("restore_format")
    return pipe
