"""The voxelization oracle (oracle/voxel_ref.py) against closed-form cases and an independent formulation of
its flood fill.  (The reference has no vectors for this path; the pin to its own kernels is tests/test_gpu_reference_pin_aux.py.)"""
import numpy as np
from scipy import ndimage

from oracle import voxel_ref as V
from tests.voxel_scenes import box_faces, nested_shells, sphere_faces


def test_axis_aligned_box_closed_form():
    # box [4,12]^3 in a 16^3 grid: every coordinate is exact in fp32, so the hit set can be written down.
    # A ray at integer (y, x) in [4,12]^2 hits the planes at 4 and 12 and marks (y, y-1) x (x, x-1): rows 3..12.
    vox = V.surface(box_faces(), 16)[0]
    shell = np.zeros((16, 16, 16), np.int32)
    for axis in range(3):
        for plane in (4, 12):
            sl = [slice(3, 13)] * 3
            sl[axis] = plane
            shell[tuple(sl)] = 1
    np.testing.assert_array_equal(vox, shell)
    solid = shell.copy()
    solid[5:12, 5:12, 5:12] = 1
    np.testing.assert_array_equal(V.voxelization(box_faces(), 16)[0], solid)


def test_fill_equals_connected_components_formulation():
    rng = np.random.default_rng(0)
    for vs, density in ((8, 0.3), (12, 0.45), (17, 0.6)):
        occ = (rng.random((4, vs, vs, vs)) < density).astype(np.int32)
        got = V.fill(occ)
        for b in range(4):
            lab, _ = ndimage.label(occ[b] == 0)                      # 6-connectivity by default
            edge = np.ones_like(lab, bool)
            edge[1:-1, 1:-1, 1:-1] = False
            outside = np.isin(lab, np.unique(lab[edge & (lab > 0)])) & (lab > 0)
            np.testing.assert_array_equal(got[b], 1 - outside.astype(np.int32))


def test_open_surface_encloses_nothing_and_shells_fill_solid():
    quad = np.array([[[[0.2, 0.2, 0.5], [0.8, 0.2, 0.5], [0.8, 0.8, 0.5]],
                      [[0.2, 0.2, 0.5], [0.8, 0.8, 0.5], [0.2, 0.8, 0.5]]]], np.float32)
    np.testing.assert_array_equal(V.voxelization(quad, 16), V.surface(quad, 16))
    one = V.voxelization(sphere_faces(1, 2, 0.9, 1, 0.0), 24)          # icosphere() has radius 0.5 -> 0.45
    two = V.voxelization(nested_shells(), 24)
    np.testing.assert_array_equal(one, two)                              # the inner shell is already inside the solid
    r = 0.45 * 24
    ball = 4.0 / 3.0 * np.pi * r ** 3
    assert 0.9 * ball < one.sum() < 1.35 * ball                          # solid, plus a one-voxel surface layer


def test_faces_outside_the_grid_and_empty_input():
    far = box_faces(2.0, 3.0)
    assert V.voxelization(far, 8).sum() == 0
    assert V.voxelization(np.zeros((2, 0, 3, 3), np.float32), 8).sum() == 0
