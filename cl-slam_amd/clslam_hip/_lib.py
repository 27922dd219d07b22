"""ctypes binding of include/clslam_hip.h.

The product loads exactly one library: ``cl-slam_amd/lib/libclslam_hip.so`` built for gfx950 by
``cl-slam_amd/csrc/build.py``.  There is NO CPU fallback: if the library is missing, or is not a
device build, loading raises.  (The host-logic tests install the kernel sources compiled against
the CPU emulator through ``install_library_for_tests``; nothing in the product calls that.)
"""
import ctypes as C
from pathlib import Path

import torch  # noqa: F401  -- FIRST: the library must bind to the HIP runtime torch ships (same libamdhip64)
from typing import Optional

LIB_PATH = Path(__file__).resolve().parents[1] / 'lib' / 'libclslam_hip.so'

OK = 0
ACT_NONE, ACT_RELU, ACT_ELU, ACT_HSWISH, ACT_HSIGMOID = 0, 1, 2, 3, 4
PAD_ZERO, PAD_REFLECT = 0, 1

fptr = C.c_void_p
i32 = C.c_int32


class ConvDesc(C.Structure):
    _fields_ = [('src_a', fptr), ('src_b', fptr), ('weight', fptr), ('scale', fptr), ('shift', fptr),
                ('residual', fptr), ('out', fptr),
                ('batch', i32), ('in_h', i32), ('in_w', i32), ('ch_a', i32), ('ch_b', i32),
                ('out_h', i32), ('out_w', i32), ('ch_out', i32),
                ('ksize', i32), ('stride', i32), ('pad', i32), ('pad_mode', i32), ('upsample_a', i32),
                ('act', i32), ('config', i32), ('actgrad_src', fptr), ('actgrad_kind', i32),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('weight_wino', fptr), ('cu_limit', i32)]


class TransposeItem(C.Structure):
    _fields_ = [('w', fptr), ('wt', fptr), ('ch_out', i32), ('taps', i32), ('ch_in', i32), ('ch_in_sel', i32)]


class CopyItem(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('bytes', C.c_size_t)]


class LossDesc(C.Structure):
    _fields_ = [('partial', fptr * 4), ('disp', fptr * 4), ('rgb0', fptr * 4), ('means', fptr * 4),
                ('pose', fptr), ('dist0', fptr), ('dist1', fptr), ('sample_w', fptr), ('smooth_w', fptr),
                ('losses', fptr), ('smooth_aux', fptr),
                ('batch', i32), ('nblk', i32), ('H', i32), ('W', i32), ('n_smooth', i32),
                ('smooth_scale', C.c_float), ('vel_scale', C.c_float)]


# name -> argtypes (restype is always int unless listed in _RESTYPES)
_SIGNATURES = {
    'clslam_version': [],
    'clslam_is_device_build': [],
    'clslam_conv2d': [C.POINTER(ConvDesc), C.c_void_p],
    'clslam_conv2d_pick_config': [C.POINTER(ConvDesc)],
    'clslam_wino_weight_transform': [fptr, fptr, i32, i32, C.c_void_p],
    'clslam_weight_transpose': [fptr, fptr, i32, i32, i32, i32, C.c_void_p],
    'clslam_weight_transpose_multi': [C.POINTER(TransposeItem), i32, C.c_void_p],
    'clslam_fold_blocks': [i32, i32, i32, i32, i32],
    'clslam_fold_act_grad': [fptr, fptr, fptr, fptr, i32, i32, i32, i32, i32, i32, i32, i32, fptr, fptr, C.c_void_p],
    'clslam_wgrad_splits': [C.POINTER(ConvDesc), i32],
    'clslam_conv_wgrad': [C.POINTER(ConvDesc), fptr, fptr, i32, C.c_void_p],
    'clslam_wgrad_patch_supported': [C.POINTER(ConvDesc)],
    'clslam_wgrad_patch_splits': [C.POINTER(ConvDesc), i32],
    'clslam_conv_wgrad_patch': [C.POINTER(ConvDesc), fptr, fptr, i32, C.c_void_p],
    'clslam_reduce_partials': [fptr, fptr, C.c_size_t, i32, C.c_float, C.c_void_p],
    'clslam_reduce_multi': [fptr, i32, i32, C.c_void_p],
    'clslam_reduce_multi_adam': [fptr, i32, i32, fptr, fptr, fptr, fptr, C.c_double, C.c_double, C.c_double, C.c_double, i32, fptr, C.c_void_p],
    'clslam_colsum_blocks': [i32],
    'clslam_colsum': [fptr, fptr, i32, i32, C.c_void_p],
    'clslam_stem_packed_size': [i32],
    'clslam_stem_pack_weight': [fptr, fptr, i32, C.c_void_p],
    'clslam_stem_conv': [fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32, i32, C.c_void_p],
    'clslam_maxpool3x3s2': [fptr, fptr, i32, i32, i32, i32, C.c_void_p],
    'clslam_dispconv_fwd': [fptr, fptr, fptr, fptr, i32, i32, i32, i32, C.c_void_p],
    'clslam_dispconv_bwd_data': [fptr, fptr, fptr, i32, i32, i32, i32, i32, C.c_void_p],
    'clslam_dispconv_wgrad_blocks': [i32],
    'clslam_dispconv_wgrad': [fptr, fptr, fptr, i32, i32, i32, i32, C.c_void_p],
    'clslam_pose_head_fwd': [fptr, fptr, fptr, fptr, fptr, i32, i32, C.c_void_p],
    'clslam_pose_head_bwd': [fptr, fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, C.c_float, C.c_void_p],
    'clslam_pose_to_proj': [fptr, fptr, fptr, fptr, i32, C.c_void_p],
    'clslam_warp_fwd': [fptr, i32, i32, fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32, C.c_float, C.c_float, C.c_void_p],
    'clslam_warp_fwd_pyramid': [C.POINTER(fptr), fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32, C.c_float, C.c_float, C.c_void_p],
    'clslam_warp_coords_pyramid': [C.POINTER(fptr), fptr, fptr, fptr, i32, i32, i32, C.c_float, C.c_float, C.c_void_p],
    'clslam_warp_cells_pyramid': [C.POINTER(fptr), fptr, fptr, C.c_void_p, i32, i32, i32, C.c_float, C.c_float, C.c_void_p],
    'clslam_warp_bwd_blocks': [i32, i32],
    'clslam_automask_pyramid': [fptr, fptr, fptr, fptr, fptr, i32, i32, i32, i32, C.c_void_p],
    'clslam_disp_mean_pyramid': [C.POINTER(fptr), fptr, i32, i32, i32, C.c_void_p],
    'clslam_photo_automask_pyramid': [fptr, fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32, C.c_void_p],
    'clslam_warp_fwd_pyramid_range': [C.POINTER(fptr), fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32, C.c_float, C.c_float, i32, i32,
                                      C.c_void_p],
    'clslam_photo_automask_pyramid_range': [fptr, fptr, fptr, fptr, C.c_uint64, C.c_uint64, fptr, fptr, fptr, i32, i32, i32, i32, i32,
                                            C.c_void_p],
    'clslam_loss_bwd2_pyramid_range': [C.POINTER(fptr), fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32,
                                       C.c_float, C.c_float, i32, i32, C.c_void_p],
    'clslam_disp_grad_pyramid_range': [fptr, C.POINTER(fptr), fptr, i32, C.POINTER(fptr), i32, i32, i32, i32, i32, C.c_void_p],
    'clslam_photo_automask_pyramid_rng': [fptr, fptr, fptr, C.c_uint64, C.c_uint64, fptr, fptr, fptr, i32, i32, i32, C.c_void_p],
    'clslam_tie_break_noise': [fptr, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p],
    'clslam_smooth_intended_chunks': [],
    'clslam_smooth_intended_fwd': [C.POINTER(fptr), C.POINTER(fptr), fptr, i32, i32, i32, C.c_void_p],
    'clslam_smooth_intended_finalize': [fptr, fptr, fptr, fptr, fptr, i32, i32, i32, C.c_float, C.c_void_p],
    'clslam_smooth_intended_bwd': [C.POINTER(fptr), C.POINTER(fptr), fptr, fptr, C.POINTER(fptr), i32, i32, i32, C.c_float, C.c_void_p],
    'clslam_loss_bwd2_blocks': [i32, i32],
    'clslam_loss_bwd2_pyramid': [C.POINTER(fptr), fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32,
                                 C.c_float, C.c_float, C.c_void_p],
    'clslam_loss_bwd_blocks': [i32, i32],
    'clslam_loss_bwd_pyramid': [C.POINTER(fptr), fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32,
                                C.c_float, C.c_float, C.c_void_p],
    'clslam_disp_grad_pyramid': [fptr, C.POINTER(fptr), fptr, i32, C.POINTER(fptr), i32, i32, i32, C.c_void_p],
    'clslam_warp_bwd': [fptr, fptr, i32, i32, fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32, C.c_float, C.c_float,
                        C.c_void_p],
    'clslam_pose_bwd': [fptr, i32, i32, fptr, fptr, fptr, fptr, fptr, C.c_float, fptr, i32, C.c_void_p],
    'clslam_photo_map': [fptr, fptr, fptr, fptr, i32, i32, i32, i32, C.c_void_p],
    'clslam_automask_blocks': [i32, i32],
    'clslam_automask': [fptr, fptr, fptr, fptr, fptr, i32, i32, i32, C.c_void_p],
    'clslam_disp_mean_chunks': [],
    'clslam_disp_mean': [fptr, fptr, i32, i32, C.c_void_p],
    'clslam_loss_finalize': [C.POINTER(LossDesc), C.c_void_p],
    'clslam_photo_grad': [fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32, C.c_void_p],
    'clslam_mbv3_stem': [fptr, fptr, fptr, fptr, fptr, i32, i32, i32, C.c_void_p],
    'clslam_dwconv': [fptr, fptr, fptr, fptr, fptr, i32, i32, i32, i32, i32, i32, i32, C.c_void_p],
    'clslam_lanczos_ksize': [i32, i32],
    'clslam_lanczos_plan': [i32, i32, C.c_void_p, C.c_void_p],
    'clslam_resize_pass_u8': [C.c_void_p, C.c_void_p, fptr, C.c_void_p, C.c_void_p, i32, i32, i32, i32, i32, i32, i32, C.c_void_p],
    'clslam_u8_to_planar_f32': [C.c_void_p, fptr, i32, i32, i32, i32, C.c_void_p],
    'clslam_color_jitter_u8': [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, i32, i32, i32, C.POINTER(C.c_int), i32,
                               C.POINTER(C.c_double), C.c_void_p],
    'clslam_color_jitter_f32_blocks': [i32, i32],
    'clslam_color_jitter_f32': [fptr, fptr, C.c_void_p, fptr, i32, i32, i32, C.c_void_p],
    'clslam_avgpool_chunks': [i32],
    'clslam_global_avgpool': [fptr, fptr, fptr, i32, i32, i32, C.c_void_p],
    'clslam_se_gate': [fptr, fptr, fptr, fptr, fptr, fptr, i32, i32, i32, C.c_void_p],
    'clslam_channel_scale': [fptr, fptr, i32, i32, i32, C.c_void_p],
    'clslam_adam_step': [fptr, fptr, fptr, fptr, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, i32, C.c_float,
                         fptr, C.c_void_p],
    'clslam_disp_grad': [fptr, fptr, fptr, i32, fptr, i32, i32, i32, i32, i32, C.c_void_p],
    'clslam_copy_multi': [C.c_void_p, i32, C.c_void_p],
    'clslam_handoff_arm': [C.c_void_p],
    'clslam_handoff_wait': [C.c_void_p, C.c_void_p, C.c_void_p],
    'clslam_conv_profile_begin': [i32],
    'clslam_conv_profile_end': [C.c_void_p, i32, C.c_void_p],
    'clslam_ip_scores': [fptr, fptr, fptr, i32, i32, i32, C.c_void_p],
    'clslam_topk_chunks': [i32],
    'clslam_topk_desc': [fptr, i32, i32, i32, fptr, C.c_void_p, fptr, C.c_void_p, C.c_void_p],
    'clslam_l2_normalize_rows': [fptr, i32, i32, C.c_void_p],
    'clslam_diversity_commit': [fptr, fptr, i32, C.c_void_p, i32, i32, i32, i32, C.c_float, fptr, fptr, C.c_void_p,
                                fptr, C.c_void_p],
}
_RESTYPES = {'clslam_last_error': C.c_char_p, 'clslam_last_error_string': C.c_char_p, 'clslam_build_id': C.c_char_p}
ABI_VERSION = 104          # include/clslam_hip.h CLSLAM_ABI_VERSION: struct layouts / pointer types this binding was written for
_SIZE_FNS = {'clslam_wino_weight_size': [i32, i32]}      # return size_t
_PTR_FNS = {'clslam_handoff_event_create': []}             # return void*
_VOID_FNS = {'clslam_handoff_event_destroy': [C.c_void_p]}


class ClslamError(RuntimeError):
    pass


class Library:
    def __init__(self, path: Path, require_device: bool = True) -> None:
        if not Path(path).exists():
            raise ClslamError(
                f'{path} not found: build it with `python cl-slam_amd/csrc/build.py` '
                '(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.')
        self.path = Path(path)
        self.cdll = C.CDLL(str(path))
        # the version FIRST: a stale library lacks the newer entry points, and resolving those before the check would end in an
        # AttributeError about one symbol instead of the instruction to rebuild
        try:
            got = int(self.cdll.clslam_version())
        except AttributeError:
            raise ClslamError(f'{path} does not export clslam_version: not a clslam kernel library') from None
        if got != ABI_VERSION:      # a stale library would read the descriptors with another layout: refuse instead of corrupting memory
            raise ClslamError(f'{path} implements ABI version {got}, this binding needs {ABI_VERSION}: rebuild it with '
                              '`python cl-slam_amd/csrc/build.py`')
        for name, rt in _RESTYPES.items():
            fn = getattr(self.cdll, name)
            fn.restype, fn.argtypes = rt, []
        for name, argtypes in _SIZE_FNS.items():
            fn = getattr(self.cdll, name)
            fn.restype, fn.argtypes = C.c_size_t, argtypes
        for name, argtypes in _PTR_FNS.items():
            fn = getattr(self.cdll, name)
            fn.restype, fn.argtypes = C.c_void_p, argtypes
        for name, argtypes in _VOID_FNS.items():
            fn = getattr(self.cdll, name)
            fn.restype, fn.argtypes = None, argtypes
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = C.c_int
        self.is_device = bool(self.cdll.clslam_is_device_build())
        if require_device and not self.is_device:
            raise ClslamError(f'{path} is not a gfx950 device build')

    def call(self, name: str, *args) -> None:
        rc = getattr(self.cdll, name)(*args)
        if rc != OK:
            raise ClslamError(f'{name} failed ({rc}): {self.cdll.clslam_last_error().decode()}')

    @property
    def device_type(self) -> str:
        return 'cuda' if self.is_device else 'cpu'


_LIB: Optional[Library] = None


def get_lib() -> Library:
    global _LIB
    if _LIB is None:
        _LIB = Library(LIB_PATH, require_device=True)
    return _LIB


def install_library_for_tests(path) -> Library:
    """TESTS ONLY: bind the kernel sources compiled against tests/emu (CPU emulator)."""
    global _LIB
    _LIB = Library(Path(path), require_device=False)
    return _LIB


def exported_symbols():
    return list(_RESTYPES) + list(_SIZE_FNS) + list(_PTR_FNS) + list(_VOID_FNS) + list(_SIGNATURES)


def build_id() -> str:
    """Identity of the kernel sources the LOADED library was built from (embedded at link time by csrc/build.py, read through
    clslam_build_id()).  Where the sources are present it is checked against them: 'stale:<lib>/<src>' when they differ (a
    library older than its sources); an install that ships only the .so just reports the embedded id."""
    built = get_lib().cdll.clslam_build_id().decode()
    built = built.split(':', 1)[1] if built.startswith('clslam-build-id:') else built
    csrc = Path(__file__).resolve().parents[1] / 'csrc'
    if not (csrc / 'build.py').exists():
        return built
    import importlib.util
    spec = importlib.util.spec_from_file_location('_clslam_build', csrc / 'build.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    src = mod.source_id()
    return built if built == src else f'stale:{built}/{src}'
