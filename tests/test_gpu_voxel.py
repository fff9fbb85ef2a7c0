"""HIP voxelization against the numpy oracle: integer grids, so the bar is bit-exact."""
import numpy as np
import pytest
import torch

import gendr_amd as gendr
from gendr_amd import functional as Fn
from gendr_amd.synthetic import icosphere
from oracle import voxel_ref as V
from tests.voxel_scenes import box_faces, nested_shells, soup, sphere_faces

pytestmark = pytest.mark.gpu


def hip(faces, size, normalize=False):
    return Fn.voxelization(torch.from_numpy(faces).cuda(), size, normalize).cpu().numpy()


@pytest.mark.parametrize('size', [1, 2, 7, 16, 32, 33, 64])
def test_spheres_bit_exact_lds_path(size):
    faces = sphere_faces(3, 2)
    out = hip(faces, size)
    assert out.dtype == np.int32 and out.shape == (3, size, size, size)
    np.testing.assert_array_equal(out, V.voxelization(faces, size))


@pytest.mark.parametrize('size', [65, 80, 130])
def test_spheres_bit_exact_global_path(size):
    faces = sphere_faces(2, 1)
    np.testing.assert_array_equal(hip(faces, size), V.voxelization(faces, size))


def test_box_closed_form_and_soup_and_shells():
    solid = np.zeros((16, 16, 16), np.int32)
    for axis in range(3):
        for plane in (4, 12):
            sl = [slice(3, 13)] * 3
            sl[axis] = plane
            solid[tuple(sl)] = 1
    solid[5:12, 5:12, 5:12] = 1
    np.testing.assert_array_equal(hip(box_faces(), 16)[0], solid)
    s = soup()
    np.testing.assert_array_equal(hip(s, 24), V.voxelization(s, 24))
    n = nested_shells()
    np.testing.assert_array_equal(hip(n, 32), V.voxelization(n, 32))


def test_winding_cavity_needs_many_sweeps():
    # a thin-walled box with a hole in one corner: the flood has to travel the whole interior
    faces = box_faces(0.1, 0.9)[:, 1:]                     # drop one triangle -> open
    np.testing.assert_array_equal(hip(faces, 48), V.voxelization(faces, 48))


def test_normalize_flag_and_mesh_method_and_errors():
    faces = sphere_faces(1, 2)
    np.testing.assert_array_equal(hip(faces * 32, 32, normalize=True), hip(faces, 32))
    v0, f0 = icosphere(2)
    mesh = gendr.Mesh(v0[None] * 0.4, f0[None])
    got = mesh.voxelize(32).cpu().numpy()
    fv = (v0 * 0.4)[f0][None].astype(np.float32)
    np.testing.assert_array_equal(got, V.voxelization(fv * np.float32(32 / 31.0) + np.float32(0.5), 32))
    with pytest.raises(TypeError):
        Fn.voxelization(torch.from_numpy(faces), 32)
    assert hip(np.zeros((2, 0, 3, 3), np.float32), 8).sum() == 0
