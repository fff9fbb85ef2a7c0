"""oracle/_ref: the REFERENCE's own device code for the render path, compiled for gfx950.  TEST INFRASTRUCTURE ONLY.

What this is.  /root/reference/gendr/cuda/generalized_renderer_cuda_kernel.cu holds the three kernels behind
`forward_render` / `backward_render` (kernel.cu:620 face preprocessing, :680 forward, :866 backward).  It is CUDA
source; this image has no CUDA headers, so the file as it stands does not compile, and its HOST half does not compile
under HIP either (the scalar exports at :1237-1271 call `normcdf`, which HIP declares for the device only -- making the
host half build would need a stand-in for CUDA's host math library, which the oracle rules forbid).  The DEVICE half needs
nothing the image lacks:

  1. PyTorch-ROCm's own source translator (`torch.utils.hipify`, the tool `CUDAExtension` runs when the upstream package
     is installed on an AMD machine) rewrites the two CUDA includes; it reads the reference file where it lies (through a
     symlink in a temporary directory) and writes its output into that temporary directory;
  2. clang compiles the DEVICE side only (`--cuda-device-only`, `--offload-arch=gfx950`), against the ATen headers of the
     installed PyTorch.  No file of the reference is edited, no header, library or function is supplied by this repo;
  3. the code object goes to oracle/_ref/gendr_ref_kernels.co (git-ignored, travels to the GPU box like the built .so
     files) together with a manifest (kernel symbols, flags, sha256 of the reference file); the temporary directory,
     including the translated source, is deleted.

oracle/ref_gpu.py loads that code object with hipModuleLoad and launches the reference's kernels with the launch shapes
of kernel.cu:1099-1150 / :1186-1222, so the `-m gpu` tests can hold the C restatement (oracle/gendr_oracle_body.inc)
against OUTPUTS OF THE REFERENCE'S OWN KERNELS on the same inputs, in float and in double.

Floating-point contraction.  nvcc fuses a*b+c into an FMA where it likes (-fmad=true is its default), clang does so with
different choices; neither is a property of the source.  The pin build uses -ffp-contract=off: every operation of the
source rounds once, which is the arithmetic the restatement states.  A second code object with clang's default
contraction (`_fma`) is built beside it; the tests report how far the two reference builds are from each other (that
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
REF_SOURCE = '/root/reference/gendr/cuda/generalized_renderer_cuda_kernel.cu'
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
VARIANTS = {'gendr_ref_kernels': ['-ffp-contract=off'], 'gendr_ref_kernels_fma': []}
KERNELS = ('forward_render_inv_cuda_kernel', 'forward_render_cuda_kernel', 'backward_render_cuda_kernel')


def manifest_path():
    return os.path.join(REF_DIR, 'manifest.json')


def available():
    return os.path.exists(manifest_path()) and all(os.path.exists(os.path.join(REF_DIR, v + '.co')) for v in VARIANTS)


def _sha256(path):
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def _symbols(co):
    out = subprocess.run([READELF, '-s', '-W', co], check=True, capture_output=True, text=True).stdout
    table = {}
    for line in out.splitlines():
        parts = line.split()
        if len(parts) != 8 or parts[3] != 'FUNC' or parts[6] == 'UND':
            continue
        for k in KERNELS:
            for tag, scalar in (('f', 'float'), ('d', 'double')):
                # _Z...<len><name>I<scalar>E...: the template instantiation on float / double
                if '%d%sI%sE' % (len(k), k, tag) in parts[7]:
                    table['%s<%s>' % (k, scalar)] = parts[7]
    return table


def build(verbose=False):
    """-> path of the manifest, or None when there is neither a reference tree nor a prebuilt code object."""
    if not os.path.exists(REF_SOURCE):
        return manifest_path() if available() else None
    sha = _sha256(REF_SOURCE)
    if available():
        with open(manifest_path()) as f:
            if json.load(f).get('reference_sha256') == sha:
                return manifest_path()
    import torch
    from torch.utils import cpp_extension
    from torch.utils.hipify import hipify_python
    os.makedirs(REF_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix='gendr_ref_')
    try:
        link = os.path.join(tmp, os.path.basename(REF_SOURCE))
        os.symlink(REF_SOURCE, link)                       # read where it lies; nothing is written next to it
        res = hipify_python.hipify(project_directory=tmp, output_directory=tmp, includes=[os.path.join(tmp, '*')],
                                   extra_files=[link], show_progress=verbose, is_pytorch_extension=True)
        translated = res[link].hipified_path
        assert translated and os.path.dirname(os.path.abspath(translated)) == tmp, translated
        inc = []
        for d in cpp_extension.include_paths(device_type='cuda'):
            inc += ['-isystem', d]
        manifest = dict(reference_file=REF_SOURCE, reference_sha256=sha, torch=torch.__version__, arch='gfx950',
                        translator='torch.utils.hipify (is_pytorch_extension=True)', variants={})
        for name, extra in VARIANTS.items():
            co = os.path.join(REF_DIR, name + '.co')
            cmd = [CLANG, '-x', 'hip', '--offload-arch=gfx950', '--cuda-device-only', '--no-gpu-bundle-output', '-O3',
                   '-std=c++17', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1', '-w'] + extra + inc + ['-c', translated, '-o', co]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('reference device code did not compile:\n' + r.stderr[-4000:])
            syms = _symbols(co)
            missing = [k + '<%s>' % s for k in KERNELS for s in ('float', 'double') if k + '<%s>' % s not in syms]
            if missing:
                raise RuntimeError('kernels missing from %s: %s' % (co, missing))
            manifest['variants'][name] = dict(file=os.path.basename(co), flags=[c for c in cmd[1:] if not c.startswith('/') and c != '-isystem'],
                                              kernels=syms)
        with open(manifest_path(), 'w') as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)             # the translated source does not stay
    return manifest_path()


if __name__ == '__main__':
    p = build(verbose='-v' in sys.argv)
    print(p if p else 'no reference tree and no prebuilt oracle/_ref: nothing built')
