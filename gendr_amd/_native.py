"""ctypes binding of ``libgendr_hip.so`` (C ABI in ``include/gendr_hip.h``).

This is the only place the Python layer touches native code; it plays the role
of the pybind11 module ``gendr.cuda.generalized_renderer`` of the reference
(``gendr/cuda/generalized_renderer_cuda.cpp:230-237``).  There is no fallback:
if the library is missing the import of any render entry point raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgendr_hip.so")

ABI_VERSION = 7


class GendrParams(ctypes.Structure):
    """``struct gendr_params`` of include/gendr_hip.h (field order is ABI)."""
    _fields_ = [
        ("image_size", ctypes.c_int),
        ("dist_func", ctypes.c_int),
        ("dist_scale", ctypes.c_float),
        ("dist_squared", ctypes.c_int),
        ("dist_shape", ctypes.c_float),
        ("dist_shift", ctypes.c_float),
        ("dist_eps", ctypes.c_float),
        ("aggr_alpha_func", ctypes.c_int),
        ("aggr_alpha_t_conorm_p", ctypes.c_float),
        ("aggr_rgb_func", ctypes.c_int),
        ("aggr_rgb_eps", ctypes.c_float),
        ("aggr_rgb_gamma", ctypes.c_float),
        ("near_", ctypes.c_float),
        ("far_", ctypes.c_float),
        ("double_side", ctypes.c_int),
        ("texture_type", ctypes.c_int),
        ("background", ctypes.c_float * 3),
        ("background_from_buffer", ctypes.c_int),
        ("texel_mode", ctypes.c_int),
        ("cull", ctypes.c_int),
        ("clear_ptr", ctypes.c_void_p),
        ("clear_floats", ctypes.c_ulonglong),
        ("deterministic", ctypes.c_int),
        ("skip_unlisted_aux", ctypes.c_int),
        ("pool_entries_max", ctypes.c_ulonglong),
        ("pair_hints", ctypes.c_int),
        ("loose_faces", ctypes.c_int),
        ("team", ctypes.c_int),
    ]


MAX_DIRECTIONAL = 4


class GendrLightParams(ctypes.Structure):
    """``gendr_light_params`` of include/gendr_hip.h."""
    _fields_ = [
        ("ambient_intensity", ctypes.c_float),
        ("ambient_color", ctypes.c_float * 3),
        ("n_directional", ctypes.c_int),
        ("intensity", ctypes.c_float * MAX_DIRECTIONAL),
        ("color", (ctypes.c_float * 3) * MAX_DIRECTIONAL),
        ("direction", (ctypes.c_float * 3) * MAX_DIRECTIONAL),
    ]


EXPORTS = (
    "gendr_abi_version", "gendr_params_size", "gendr_error_string", "gendr_workspace_bytes", "gendr_validate",
    "gendr_face_setup", "gendr_forward", "gendr_backward", "gendr_face_info",
    "gendr_sigmoid_forward", "gendr_sigmoid_backward", "gendr_t_conorm_forward", "gendr_t_conorm_backward",
    "gendr_cull_radius", "gendr_project_faces", "gendr_project_faces_backward",
    "gendr_camera_rotation", "gendr_camera_rotation_backward",
    "gendr_silhouette_workspace_bytes", "gendr_silhouette_forward", "gendr_silhouette_backward", "gendr_workspace_bytes_f64", "gendr_forward_f64", "gendr_backward_f64", "gendr_selftest", "gendr_light_faces", "gendr_light_faces_backward", "gendr_voxelize_workspace_bytes", "gendr_voxelize", "gendr_load_textures", "gendr_create_texture_image",
    "gendr_uses_team", "gendr_uses_team_cover",
)

_libs = {}
# Build variant the render entry points use: "default" (libgendr_hip.so) or "exact" (libgendr_hip_exact.so, the backward
# kernels with the reference's rounding on the gradient side, gendr_amd/build.py).  Process-wide default from
# GENDR_VARIANT; tests switch it with `use_variant`.
_active = os.environ.get("GENDR_VARIANT", "default")


class NativeLibraryError(RuntimeError):
    pass


def variant_path(variant):
    return LIB_PATH if variant == "default" else os.path.join(_HERE, "libgendr_hip_%s.so" % variant)


class use_variant(object):
    """``with use_variant('exact'): ...`` routes every native call inside through that build variant."""
    def __init__(self, variant):
        self.variant = variant

    def __enter__(self):
        global _active
        self.prev, _active = _active, self.variant
        return lib()

    def __exit__(self, *exc):
        global _active
        _active = self.prev
        return False


def lib(variant=None):
    """Loads the library (of the active build variant) once.  Raises NativeLibraryError (never falls back) if it is
    absent or stale."""
    variant = variant or _active
    if variant in _libs:
        return _libs[variant]
    path = variant_path(variant)
    if not os.path.exists(path):
        raise NativeLibraryError(
            "gendr_amd: %s not found. Build it with `python -m gendr_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU or PyTorch fallback." % path)
    L = ctypes.CDLL(path)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise NativeLibraryError("gendr_amd: %s does not export %s" % (path, name))
    i, f, vp = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
    pp = ctypes.POINTER(GendrParams)
    L.gendr_abi_version.restype = i
    L.gendr_abi_version.argtypes = []
    L.gendr_error_string.restype = ctypes.c_char_p
    L.gendr_error_string.argtypes = [i]
    L.gendr_workspace_bytes.restype = ctypes.c_ulonglong
    L.gendr_workspace_bytes.argtypes = [i, i, i, pp]
    L.gendr_validate.restype = i
    L.gendr_validate.argtypes = [pp, i, i, i]
    L.gendr_face_setup.restype = i
    L.gendr_face_setup.argtypes = [vp, vp, vp, i, i, i, pp, vp]
    L.gendr_forward.restype = i
    L.gendr_forward.argtypes = [vp, vp, vp, vp, vp, i, i, i, pp, vp]
    L.gendr_backward.restype = i
    L.gendr_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, pp, vp]
    L.gendr_silhouette_workspace_bytes.restype = ctypes.c_ulonglong
    L.gendr_silhouette_workspace_bytes.argtypes = [i, i, pp]
    L.gendr_silhouette_forward.restype = i
    L.gendr_silhouette_forward.argtypes = [vp, vp, vp, vp, vp, i, i, pp, vp]
    L.gendr_silhouette_backward.restype = i
    L.gendr_silhouette_backward.argtypes = [vp, vp, vp, vp, vp, vp, i, i, pp, vp]
    L.gendr_workspace_bytes_f64.restype = ctypes.c_ulonglong
    L.gendr_workspace_bytes_f64.argtypes = [i, i, i, pp]
    L.gendr_forward_f64.restype = i
    L.gendr_forward_f64.argtypes = [vp, vp, vp, vp, vp, i, i, i, pp, vp]
    L.gendr_backward_f64.restype = i
    L.gendr_backward_f64.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, pp, vp]
    L.gendr_selftest.restype = i
    L.gendr_selftest.argtypes = [i, vp, vp]
    L.gendr_face_info.restype = i
    L.gendr_face_info.argtypes = [vp, vp, i, i, vp]
    for name in ("gendr_sigmoid_forward", "gendr_sigmoid_backward"):
        getattr(L, name).restype = f
        getattr(L, name).argtypes = [i, f, f, f, f, f]
    for name in ("gendr_t_conorm_forward", "gendr_t_conorm_backward"):
        getattr(L, name).restype = f
        getattr(L, name).argtypes = [i, f, f, i, f]
    L.gendr_light_faces.restype = i
    L.gendr_light_faces.argtypes = [vp, vp, vp, vp, i, i, i, i, i, ctypes.POINTER(GendrLightParams), vp]
    L.gendr_light_faces_backward.restype = i
    L.gendr_light_faces_backward.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, ctypes.POINTER(GendrLightParams), vp]
    L.gendr_load_textures.restype = i
    L.gendr_load_textures.argtypes = [vp, vp, vp, vp, i, i, i, i, vp]
    L.gendr_create_texture_image.restype = i
    L.gendr_create_texture_image.argtypes = [vp, vp, vp, i, i, i, i, i, f, vp]
    L.gendr_voxelize_workspace_bytes.restype = ctypes.c_size_t
    L.gendr_voxelize_workspace_bytes.argtypes = [i, i]
    L.gendr_voxelize.restype = i
    L.gendr_voxelize.argtypes = [vp, vp, vp, i, i, i, vp]
    L.gendr_camera_rotation.restype = i
    L.gendr_camera_rotation.argtypes = [vp, vp, vp, vp, i, i, vp]
    L.gendr_camera_rotation_backward.restype = i
    L.gendr_camera_rotation_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.gendr_project_faces.restype = i
    L.gendr_project_faces.argtypes = [vp, vp, vp, vp, i, i, i, i, i, f, vp]
    L.gendr_project_faces_backward.restype = i
    L.gendr_project_faces_backward.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, f, vp]
    L.gendr_uses_team.restype = i
    L.gendr_uses_team.argtypes = [i, i, i, pp, i]
    L.gendr_uses_team_cover.restype = i
    L.gendr_uses_team_cover.argtypes = [i, i, i, pp]
    L.gendr_cull_radius.restype = f
    L.gendr_cull_radius.argtypes = [pp]
    L.gendr_params_size.restype = i
    L.gendr_params_size.argtypes = []
    if L.gendr_params_size() != ctypes.sizeof(GendrParams):
        raise NativeLibraryError("gendr_amd: gendr_params layout mismatch (library %d bytes, python %d bytes)"
                                 % (L.gendr_params_size(), ctypes.sizeof(GendrParams)))
    if L.gendr_abi_version() != ABI_VERSION:
        raise NativeLibraryError("gendr_amd: ABI version mismatch (library %d, python %d); rebuild"
                                 % (L.gendr_abi_version(), ABI_VERSION))
    _libs[variant] = L
    return L


def error_string(code):
    return lib().gendr_error_string(int(code)).decode()


# ---- the C++ autograd node (gendr_amd/csrc/gendr_torch.cpp -> _gendr_torch.so) -------------------------------------------------------
TORCH_EXT_PATH = os.path.join(_HERE, "_gendr_torch.so")
_torch_ext = None
_torch_slots = {}


def torch_ext():
    """The compiled autograd node, or None when it is not built (the ctypes-based GenDRFunction then does the same work: both paths
    end in the same two C-ABI calls, neither is a fallback off the HIP kernels)."""
    global _torch_ext
    if _torch_ext is None:
        if not os.path.exists(TORCH_EXT_PATH):
            _torch_ext = False
        else:
            import importlib.util
            import torch  # noqa: F401  (its libraries must be loaded first)
            spec = importlib.util.spec_from_file_location("_gendr_torch", TORCH_EXT_PATH)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _torch_ext = mod
    return _torch_ext or None


def torch_slot(variant=None):
    """Slot of the active library variant's entry points inside the C++ node (bound once per variant)."""
    variant = variant or _active
    if variant not in _torch_slots:
        L = lib(variant)
        addr = lambda f: ctypes.cast(f, ctypes.c_void_p).value
        _torch_slots[variant] = torch_ext().bind(addr(L.gendr_validate), addr(L.gendr_workspace_bytes), addr(L.gendr_forward),
                                                 addr(L.gendr_backward), addr(L.gendr_error_string), L.gendr_params_size(),
                                                 L.gendr_abi_version())
    return _torch_slots[variant]
