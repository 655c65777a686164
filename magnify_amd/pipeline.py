"""Pipeline runner with the reference's contract (src/magnify/pipeline.py:9-87): an ordered
sequence of named stages fed by a registered reader.

Behaviour kept: stages are added by registry name (factory called with the keyword arguments) or as
a bare callable ``f(xp, **kwargs)``; placement by ``first`` / ``last`` / ``before`` / ``after`` (name
or integer position), default append; more than one placement selector or a duplicate stage name is a
``ValueError``; ``remove_pipe`` raises ``ValueError`` for an empty pipeline or an unknown name;
calling the pipeline returns one result, or a list when the reader yields several assays.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable


@dataclass
class _Stage:
    name: str
    run: Callable[[Any], Any]


class Pipeline:
    def __init__(self, reader: str):
        from . import registry

        self.reader = registry.readers.get(reader)()
        self._stages: list[_Stage] = []

    # The reference exposes the stage list as (name, callable) pairs.
    @property
    def components(self):
        return [(st.name, st.run) for st in self._stages]

    def _names(self):
        return [st.name for st in self._stages]

    def __call__(self, data):
        results = []
        for assay in self.reader(data=data):
            for stage in self._stages:
                assay = stage.run(assay)
            results.append(assay)
        return results[0] if len(results) == 1 else results

    def _resolve(self, component, name, kwargs):
        from . import registry

        if isinstance(component, str):
            stage_fn = registry.components.get(component)(**kwargs)
            return name or component, stage_fn
        return name or component.__name__, (lambda xp, _f=component, _kw=dict(kwargs): _f(xp, **_kw))

    def _slot(self, after, before, first, last):
        selectors = [after is not None, before is not None, bool(first), bool(last)]
        if sum(selectors) > 1:
            raise ValueError("Only one of after, before, first, and last can be set.")
        if first:
            return 0
        if last or not any(selectors):
            return len(self._stages)
        anchor, shift = (before, 0) if before is not None else (after, 1)
        if isinstance(anchor, bool) or not isinstance(anchor, (int, str)):
            raise ValueError("before/after must be a string or int.")
        return (anchor if isinstance(anchor, int) else self._names().index(anchor)) + shift

    def add_pipe(self, component, name=None, after=None, before=None, first=False, last=False, **kwargs):
        stage_name, stage_fn = self._resolve(component, name, kwargs)
        where = self._slot(after, before, first, last)
        if stage_name in self._names():
            raise ValueError(f"A component with the name '{stage_name}' already exists in the pipeline.")
        self._stages.insert(where, _Stage(stage_name, stage_fn))

    def remove_pipe(self, name: str) -> None:
        if not self._stages:
            raise ValueError(f"Cannot remove pipe '{name}': pipeline has no components")
        if name not in self._names():
            raise ValueError(f"Component '{name}' not found in pipeline")
        del self._stages[self._names().index(name)]
