"""BASELINE.json's configs C1 and C3 at full size through the drop-in API, checked against the oracle
(C2 is tests/test_gpu_fullsize.py::test_public_api_c2_matches_c_oracle, C4 is bench.py's workload and
test_full_size_assays_match_c_oracle, C5's path is the streamed / tiled tests).  The work is
tests/config_table.py's, which also times it for DESIGN.md's table."""
import argparse

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _args():
    return argparse.Namespace(num_iter=5_000_000, repeat=1, no_cpu=False)


def test_c1_single_plane_through_mg_beads():
    """C1: one 2048 x 2048 single-channel plane (seed 1000, 503 drawn beads), reference-default 5e6
    iterations, through mg.beads: bead positions, ROI pixels and fg masks equal the C oracle's."""
    import config_table

    rec = config_table.bead_config("C1", 1, 2048, 1000, _args())
    assert rec["same_beads_roi_fg_as_gpu"] is True
    assert rec["markers"] >= 0.8 * rec["drawn_beads"] and rec["shape"] == [1, 1, 2048, 2048]


def test_c3_tile_stack_through_mg_microfluidic_chip():
    """C3: the 8 x 8 stack of 1024^2 tiles (overlap 102 -> 7376^2), 28 x 28 buttons, through
    mg.microfluidic_chip: the stitched image equals the oracle's stitch, all 784 positions equal the
    oracle's find_centers / find_rois (its C port as the circle search) and sit on the drawn grid."""
    import config_table

    rec = config_table.chip_config(_args())
    assert rec["same_image_as_oracle_stitch"] is True
    assert rec["same_xy_as_gpu"] is True
    assert rec["markers"] == 784 and rec["stitched"] == [7376, 7376]
    assert rec["max_centre_error_px"] <= 1.0
