"""HIP texture-atlas kernels against the numpy oracle (same fp32 operations in the same order:
the bar is bit-exact), and OBJ + MTL + PNG round trips through the public API."""
import os

import numpy as np
import pytest
import torch

import gendr_amd as gendr
from gendr_amd import _native
from gendr_amd.functional import obj_io
from gendr_amd.functional.renderer import check
from gendr_amd.synthetic import icosphere
from oracle import texture_ref as T

pytestmark = pytest.mark.gpu


def hip_load_textures(image, uv, upd, tex, R):
    lib = _native.lib()
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to('cuda', dt).contiguous()
    image, uv, upd, tex = d(image, torch.float32), d(uv, torch.float32), d(upd, torch.int32), d(tex, torch.float32)
    check(lib.gendr_load_textures(image.data_ptr(), uv.data_ptr(), upd.data_ptr(), tex.data_ptr(), tex.shape[0], R,
                                  image.shape[0], image.shape[1], torch.cuda.current_stream().cuda_stream), 'load')
    return tex.cpu().numpy()


@pytest.mark.parametrize('R,H,W,nf', [(1, 8, 8, 5), (2, 9, 11, 33), (4, 64, 48, 320), (7, 31, 130, 77)])
def test_load_textures_bit_exact(R, H, W, nf):
    rng = np.random.default_rng(R)
    img = rng.random((H, W, 3)).astype(np.float32)
    uv = rng.uniform(0, 1, (nf, 3, 2)).astype(np.float32)
    uv[0] = 1.0
    uv[1] = 0.0
    upd = (rng.random(nf) < 0.7).astype(np.int32)
    tex0 = rng.random((nf, R * R, 3)).astype(np.float32)
    np.testing.assert_array_equal(hip_load_textures(img, uv, upd, tex0, R), T.load_textures(img, uv, upd, tex0))


@pytest.mark.parametrize('nf,R,res', [(1, 1, 2), (10, 4, 16), (17, 3, 8), (320, 2, 6), (1280, 4, 16)])
def test_create_texture_image_bit_exact(nf, R, res):
    tex = np.random.default_rng(nf).random((nf, R * R, 3)).astype(np.float32)
    img, uv = obj_io.create_texture_image(torch.from_numpy(tex), res)
    ref_img, ref_uv = T.create_texture_image(tex, res)
    np.testing.assert_array_equal(img, ref_img)
    np.testing.assert_allclose(uv, ref_uv, rtol=0, atol=1e-7)


def test_obj_round_trip_with_surface_textures(tmp_path):
    v0, f0 = icosphere(1)
    nf = f0.shape[0]
    colours = (np.random.default_rng(0).integers(0, 256, (nf, 1, 3)) / 255.0).astype(np.float32)
    tex = np.broadcast_to(colours, (nf, 16, 3)).copy()
    mesh = gendr.Mesh(v0[None], f0[None], tex[None], texture_type='surface')
    path = os.path.join(str(tmp_path), 'ball.obj')
    mesh.save_obj(path, save_texture=True, texture_res_out=8)
    assert os.path.exists(path[:-4] + '.png') and os.path.exists(path[:-4] + '.mtl')
    back = gendr.Mesh.from_obj(path, load_texture=True, texture_res=4, texture_type='surface')
    assert back.textures.shape == (1, nf, 16, 3)
    np.testing.assert_allclose(back.vertices.cpu().numpy(), v0[None], atol=1e-6)
    np.testing.assert_array_equal(back.faces.cpu().numpy(), f0[None])
    np.testing.assert_allclose(back.textures.cpu().numpy()[0], tex, atol=1.5 / 255)
    # and the loaded mesh renders
    images = gendr.GenDR(image_size=32)(gendr.LookAt()(back))
    assert images.shape == (1, 4, 32, 32) and torch.isfinite(images).all()


def test_materials_without_images_use_kd(tmp_path):
    d = str(tmp_path)
    with open(os.path.join(d, 'm.mtl'), 'w') as fh:
        fh.write('newmtl red\nKd 1.0 0.0 0.0\nnewmtl blue\nKd 0.0 0.0 1.0\n')
    with open(os.path.join(d, 'm.obj'), 'w') as fh:
        fh.write('mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvt 0 1\n'
                 'usemtl red\nf 1/1 2/2 3/3\nusemtl blue\nf 2/1 4/2 3/3\n')
    v, f, t = obj_io.load_obj(os.path.join(d, 'm.obj'), load_texture=True, texture_res=2)
    assert t.shape == (2, 4, 3)
    np.testing.assert_array_equal(t[0].cpu().numpy(), np.broadcast_to(np.float32([1, 0, 0]), (4, 3)))
    np.testing.assert_array_equal(t[1].cpu().numpy(), np.broadcast_to(np.float32([0, 0, 1]), (4, 3)))
