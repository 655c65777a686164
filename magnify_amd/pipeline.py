"""Pipeline runner (reference: src/magnify/pipeline.py:9-87): an ordered list of
``(name, callable)``; ``add_pipe`` by registry name or bare callable with the
after/before/first/last placement rules; ``pipe(data)`` loops assays x components."""
from __future__ import annotations


class Pipeline:
    def __init__(self, reader: str):
        from . import registry

        self.reader = registry.readers.get(reader)()
        self.components = []

    def __call__(self, data):
        assays = []
        for assay in self.reader(data=data):
            for _, component in self.components:
                assay = component(assay)
            assays.append(assay)
        if len(assays) == 1:
            assays = assays[0]
        return assays

    def add_pipe(self, component, name=None, after=None, before=None, first=False, last=False, **kwargs):
        from . import registry

        if isinstance(component, str):
            if name is None:
                name = component
            func = registry.components.get(component)(**kwargs)
        else:
            name = component.__name__ if name is None else name

            def func(xp, _component=component):
                return _component(xp, **kwargs)

        if after is None and before is None and not first and not last:
            last = True
        if (after is not None) + (before is not None) + first + last > 1:
            raise ValueError("Only one of after, before, first, and last can be set.")
        names = [n for n, _ in self.components]
        if name in names:
            raise ValueError(f"A component with the name '{name}' already exists in the pipeline.")
        if first:
            idx = 0
        elif last:
            idx = len(self.components)
        elif isinstance(before, int):
            idx = before
        elif isinstance(before, str):
            idx = names.index(before)
        elif isinstance(after, int):
            idx = after + 1
        elif isinstance(after, str):
            idx = names.index(after) + 1
        else:
            raise ValueError("before/after must be a string or int.")
        self.components.insert(idx, (name, func))

    def remove_pipe(self, name: str) -> None:
        if not self.components:
            raise ValueError(f"Cannot remove pipe '{name}': pipeline has no components")
        names = [n for n, _ in self.components]
        if name not in names:
            raise ValueError(f"Component '{name}' not found in pipeline")
        self.components.pop(names.index(name))
