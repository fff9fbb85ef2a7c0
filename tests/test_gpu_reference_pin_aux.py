"""The pin of SURVEY rows f-2 / f-3: the REFERENCE's own voxelization and texture-atlas kernels (oracle/_ref, built by
oracle/build_ref.py from voxelization_cuda_kernel.cu, load_textures_cuda_kernel.cu and create_texture_image_cuda_kernel.cu;
launched by oracle/ref_gpu.py) against the numpy restatements (oracle/voxel_ref.py, oracle/texture_ref.py) and the HIP
kernels, stage by stage.  Integer grids and single-rounding float arithmetic: the bar is bit-exact."""
import numpy as np
import pytest
import torch

from gendr_amd import functional as Fn
from oracle import ref_gpu
from oracle import texture_ref as T
from oracle import voxel_ref as V
from tests.voxel_scenes import box_faces, nested_shells, soup, sphere_faces

pytestmark = pytest.mark.gpu


def _need(name):
    import parity
    parity.require_reference(name)


SCENES = {'spheres': lambda: sphere_faces(3, 2), 'box': box_faces, 'shells': nested_shells, 'soup': soup}


@pytest.mark.parametrize('size', [1, 7, 16, 33, 64])
@pytest.mark.parametrize('scene', sorted(SCENES))
def test_voxel_stages_match_reference_kernels(scene, size):
    _need('voxelization')
    faces = SCENES[scene]()
    f = faces.astype(np.float32) * np.float32(size)
    for dim in range(3):
        np.testing.assert_array_equal(ref_gpu.voxel_sub1(f, size, dim), V._sub1(f, size, dim), err_msg='sub1 dim %d' % dim)
    np.testing.assert_array_equal(ref_gpu.voxel_sub2(f, size), V._sub2(f, size))
    surface = V.surface(faces, size)
    filled, sweeps = ref_gpu.voxel_fill(surface)
    np.testing.assert_array_equal(filled, V.fill(surface))
    whole = ref_gpu.voxelization(faces, size)
    np.testing.assert_array_equal(whole, V.voxelization(faces, size))
    hip = Fn.voxelization(torch.from_numpy(faces).cuda(), size).cpu().numpy()
    np.testing.assert_array_equal(hip, whole, err_msg='HIP product vs reference kernels')


def test_voxel_large_grid_and_winding_cavity():
    _need('voxelization')
    faces = sphere_faces(2, 1)
    np.testing.assert_array_equal(ref_gpu.voxelization(faces, 130), V.voxelization(faces, 130))
    # a serpentine cavity: the reference needs many sub4 sweeps; the fixpoint is what counts
    vs = 24
    vox = np.ones((1, vs, vs, vs), np.int32)
    for x in range(1, vs - 1):
        ys = range(1, vs - 1) if x % 4 == 1 else ([vs - 2] if x % 4 == 2 else (range(1, vs - 1) if x % 4 == 3 else [1]))
        for y in ys:
            vox[0, x, y, 1:vs - 1] = 0
    vox[0, 0, 1, 1] = 0                                         # the mouth of the cavity on the boundary
    got, sweeps = ref_gpu.voxel_fill(vox)
    np.testing.assert_array_equal(got, V.fill(vox))
    assert sweeps >= 1


@pytest.mark.parametrize('R,H,W,nf', [(1, 8, 8, 5), (2, 9, 11, 33), (4, 64, 48, 320), (7, 31, 130, 77)])
def test_load_textures_matches_reference_kernel(R, H, W, nf):
    _need('load_textures')
    rng = np.random.default_rng(R)
    img = rng.random((H, W, 3)).astype(np.float32)
    uv = rng.uniform(0, 1, (nf, 3, 2)).astype(np.float32)
    # keep the samples off the last row / column, where the reference's weight-0 neighbour lies outside the image
    # (the restatement and the HIP kernel clamp that index; tests/test_gpu_texture.py covers it)
    uv *= np.float32(0.97)
    upd = (rng.random(nf) < 0.7).astype(np.int32)
    tex0 = rng.random((nf, R * R, 3)).astype(np.float32)
    got = ref_gpu.load_textures(img, uv, upd, tex0)
    np.testing.assert_array_equal(got, T.load_textures(img, uv, upd, tex0))
    from test_gpu_texture import hip_load_textures
    np.testing.assert_array_equal(hip_load_textures(img, uv, upd, tex0, R), got)


@pytest.mark.parametrize('nf,R,res', [(1, 1, 2), (10, 4, 16), (17, 3, 8), (320, 2, 6), (1280, 4, 16)])
def test_create_texture_image_matches_reference_kernel(nf, R, res):
    _need('create_texture_image')
    tex = np.random.default_rng(nf).random((nf, R * R, 3)).astype(np.float32)
    tile_width, tile_height, uv = T.atlas_layout(nf, res)
    image = np.ones((tile_height * res, tile_width * res, 3), np.float32)
    got = ref_gpu.create_texture_image_kernel(uv, tex, image, 1e-5)
    np.testing.assert_array_equal(got, T.create_texture_image_kernel(uv, tex, image, tile_width, 1e-5))
    from gendr_amd.functional import obj_io
    img, _ = obj_io.create_texture_image(torch.from_numpy(tex), res)
    np.testing.assert_array_equal(np.asarray(img), got[::-1], err_msg='HIP product vs reference kernel')
