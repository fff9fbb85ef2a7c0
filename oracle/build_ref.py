"""oracle/_ref: the REFERENCE's own device code, compiled for gfx950.  TEST INFRASTRUCTURE ONLY.

What this is.  /root/reference/gendr/cuda/generalized_renderer_cuda_kernel.cu holds the three kernels behind
`forward_render` / `backward_render` (kernel.cu:620 face preprocessing, :680 forward, :866 backward); the voxelization
and the two texture-atlas files beside it hold the kernels of SURVEY rows f-2 / f-3.  They are CUDA sources; this image
has no CUDA headers, so the files as they stand do not compile, and the HOST half of the render file does not compile
under HIP either (the scalar exports at :1237-1271 call `normcdf`, which HIP declares for the device only -- making the
host half build would need a stand-in for CUDA's host math library, which the oracle rules forbid).  The DEVICE halves need
nothing the image lacks:

  1. PyTorch-ROCm's own source translator (`torch.utils.hipify`, the tool `CUDAExtension` runs when the upstream package
     is installed on an AMD machine) rewrites the two CUDA includes; it reads the reference files where they lie (through
     symlinks in a temporary directory) and writes its output into that temporary directory;
  2. clang compiles the DEVICE side only (`--cuda-device-only`, `--offload-arch=gfx950`), against the ATen headers of the
     installed PyTorch.  No file of the reference is edited, no header, library or function is supplied by this repo;
  3. the code objects go to oracle/_ref/gendr_ref_<name>.co (git-ignored, they travel to the GPU box like the built .so
     files) together with a manifest (kernel symbols, flags, sha256 of the reference files); the temporary directory,
     including the translated sources, is deleted.

oracle/ref_gpu.py loads the code objects with hipModuleLoad and launches the reference's kernels with the launch shapes
of the reference's host code (kernel.cu:1099-1150 / :1186-1222 and the like), so the `-m gpu` tests can hold the
restatements (oracle/gendr_oracle_body.inc, voxel_ref.py, texture_ref.py) against OUTPUTS OF THE REFERENCE'S OWN KERNELS
on the same inputs, in float and in double.

Floating-point contraction.  nvcc fuses a*b+c into an FMA where it likes (-fmad=true is its default), clang does so with
different choices; neither is a property of the source.  The pin build uses -ffp-contract=off: every operation of the
source rounds once, which is the arithmetic the restatement states.  A second render code object with clang's default
contraction (`render_fma`) is built beside it; the tests report how far the two reference builds are from each other (that
distance is what "the reference's results" are uncertain by on any machine).

    python -m oracle.build_ref        # needs /root/reference; a no-op (keeps the prebuilt files) where it is absent
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, '_ref')
REF_CUDA_DIR = '/root/reference/gendr/cuda'
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
# code object -> (reference file, extra flags, kernels expected in float and double)
OBJECTS = {
    'render': ('generalized_renderer_cuda_kernel.cu', ['-ffp-contract=off'],
               ('forward_render_inv_cuda_kernel', 'forward_render_cuda_kernel', 'backward_render_cuda_kernel')),
    'render_fma': ('generalized_renderer_cuda_kernel.cu', [],
                   ('forward_render_inv_cuda_kernel', 'forward_render_cuda_kernel', 'backward_render_cuda_kernel')),
    'voxelization': ('voxelization_cuda_kernel.cu', ['-ffp-contract=off'],
                     ('voxelize_sub1_kernel', 'voxelize_sub2_kernel', 'voxelize_sub3_kernel', 'voxelize_sub4_kernel')),
    'load_textures': ('load_textures_cuda_kernel.cu', ['-ffp-contract=off'], ('load_textures_cuda_kernel',)),
    'create_texture_image': ('create_texture_image_cuda_kernel.cu', ['-ffp-contract=off'], ('create_texture_image_cuda_kernel',)),
}
REF_SOURCE = os.path.join(REF_CUDA_DIR, OBJECTS['render'][0])


def manifest_path():
    return os.path.join(REF_DIR, 'manifest.json')


def _manifest():
    try:
        with open(manifest_path()) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def available(name=None):
    """Is the code object `name` (default: every one of OBJECTS) built?"""
    man = _manifest()
    if man is None:
        return False
    names = OBJECTS if name is None else (name,)
    return all(n in man.get('objects', {}) and os.path.exists(os.path.join(REF_DIR, man['objects'][n]['file'])) for n in names)


def _sha256(path):
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def _symbols(co, kernels):
    out = subprocess.run([READELF, '-s', '-W', co], check=True, capture_output=True, text=True).stdout
    table = {}
    for line in out.splitlines():
        parts = line.split()
        if len(parts) != 8 or parts[3] != 'FUNC' or parts[6] == 'UND':
            continue
        for k in kernels:
            for tag, scalar in (('f', 'float'), ('d', 'double')):
                # _Z...<len><name>I<scalar>E...: the template instantiation on float / double
                if '%d%sI%sE' % (len(k), k, tag) in parts[7]:
                    table['%s<%s>' % (k, scalar)] = parts[7]
    return table


def build(verbose=False):
    """-> path of the manifest, or None when there is neither a reference tree nor a prebuilt set of code objects."""
    if not os.path.isdir(REF_CUDA_DIR):
        return manifest_path() if available() else None
    shas = {f: _sha256(os.path.join(REF_CUDA_DIR, f)) for f in sorted({o[0] for o in OBJECTS.values()})}
    man = _manifest()
    if available() and man.get('reference_sha256') == shas:
        return manifest_path()
    import torch
    from torch.utils import cpp_extension
    from torch.utils.hipify import hipify_python
    os.makedirs(REF_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix='gendr_ref_')
    try:
        links = {}
        for f in shas:
            links[f] = os.path.join(tmp, f)
            os.symlink(os.path.join(REF_CUDA_DIR, f), links[f])     # read where they lie; nothing is written next to them
        res = hipify_python.hipify(project_directory=tmp, output_directory=tmp, includes=[os.path.join(tmp, '*')],
                                   extra_files=list(links.values()), show_progress=verbose, is_pytorch_extension=True)
        inc = []
        for d in cpp_extension.include_paths(device_type='cuda'):
            inc += ['-isystem', d]
        manifest = dict(reference_dir=REF_CUDA_DIR, reference_sha256=shas, torch=torch.__version__, arch='gfx950',
                        translator='torch.utils.hipify (is_pytorch_extension=True)', objects={})
        for name, (f, extra, kernels) in OBJECTS.items():
            translated = res[links[f]].hipified_path
            assert translated and os.path.dirname(os.path.abspath(translated)) == tmp, translated
            co = os.path.join(REF_DIR, 'gendr_ref_%s.co' % name)
            cmd = [CLANG, '-x', 'hip', '--offload-arch=gfx950', '--cuda-device-only', '--no-gpu-bundle-output', '-O3',
                   '-std=c++17', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1', '-w'] + extra + inc + ['-c', translated, '-o', co]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('reference device code of %s did not compile:\n%s' % (f, r.stderr[-4000:]))
            syms = _symbols(co, kernels)
            missing = [k + '<%s>' % t for k in kernels for t in ('float', 'double') if k + '<%s>' % t not in syms]
            if missing:
                raise RuntimeError('kernels missing from %s: %s' % (co, missing))
            manifest['objects'][name] = dict(file=os.path.basename(co), source=f, kernels=syms,
                                             flags=[c for c in cmd[1:] if not c.startswith('/') and c != '-isystem'])
        stale = set(os.listdir(REF_DIR)) - {o['file'] for o in manifest['objects'].values()} - {'manifest.json'}
        for f in stale:
            os.remove(os.path.join(REF_DIR, f))
        with open(manifest_path(), 'w') as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)             # the translated sources do not stay
    return manifest_path()


if __name__ == '__main__':
    p = build(verbose='-v' in sys.argv)
    print(p if p else 'no reference tree and no prebuilt oracle/_ref: nothing built')
