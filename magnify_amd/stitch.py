"""Stitcher (reference: src/magnify/stitch.py:7-50): crop ``overlap // 2`` (+ the odd remainder on
the far side) from every tile edge and butt the tiles together; no blending.  One HIP kernel does
the crop/concat, fused with a pending flat-field correction and with the per-plane min/max that
``to_uint8`` needs later."""
from __future__ import annotations

from . import hotpath, preprocess, registry
from .xr_lite import DataArray


class Stitcher:
    def __init__(self, overlap: int = 102):
        if overlap < 0:
            raise ValueError("Overlap must be non-negative.")
        self.overlap = overlap

    def __call__(self, assay):
        if "tile" not in assay:
            raise AttributeError("Dataset must contain 'tile' data variable.")
        sizes = assay.data_vars["tile"].sizes
        if self.overlap >= sizes["tile_y"] or self.overlap >= sizes["tile_x"]:
            raise ValueError(f"Overlap ({self.overlap}) must be smaller than tile size "
                             f"({sizes['tile_y']}x{sizes['tile_x']}).")
        tile = assay.data_vars["tile"].transpose("channel", "time", "tile_row", "tile_col", "tile_y", "tile_x")
        raw = tile.raw
        lazy = isinstance(raw, preprocess.LazyFlatfield)
        if self.overlap == 0 and sizes["tile_row"] == 1 and sizes["tile_col"] == 1 and (not lazy or raw.max2 is None):
            # One tile, nothing to crop and (integer pixels, flat 1, dark 0: LazyFlatfield.max2 is None) nothing to
            # correct: the image IS the tile array -- no pass over it at all (a 4 x 4096^2 assay: 70 us of copying).
            # The finders take the min / max of the planes they search themselves.  The image then shares its memory
            # with the tile array it was given (deliberate differences).
            tiles = raw.tiles if lazy else preprocess.to_device(raw)
            if tiles.is_contiguous():
                image = tiles.view(tiles.shape[0], tiles.shape[1], tiles.shape[4], tiles.shape[5])
                assay["image"] = DataArray(image, ("channel", "time", "im_y", "im_x"))
                assay._cache["image_minmax"] = (image.data_ptr(), None)
                return assay
        if isinstance(raw, preprocess.LazyFlatfield):
            image, minmax = hotpath.flatfield_stitch(raw.tiles, self.overlap, raw.flatfield, raw.darkfield,
                                                     max2=raw.max2)
        else:
            image, minmax = hotpath.flatfield_stitch(preprocess.to_device(raw), self.overlap, apply_flatfield=False)
        assay["image"] = DataArray(image, ("channel", "time", "im_y", "im_x"))
        assay._cache["image_minmax"] = (image.data_ptr(), minmax)
        return assay

    @registry.components.register("stitch")
    def make(overlap: int = 102):
        return Stitcher(overlap=overlap)
