"""ctypes loader + typed prototypes for every symbol declared in include/jckgan.h."""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(HERE), "lib", "libjckgan_hip.so")
PREC_BF16, PREC_F32 = 0, 1

vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double, C.c_size_t


class StepInputs(C.Structure):
    _fields_ = [("real_nchw", vp), ("noise_real", vp), ("z", vp), ("noise_fake", vp), ("alpha", vp),
                ("lr", f32), ("grad_scale", f32), ("step", i32), ("labels", vp), ("drop_mask", vp * 4),
                ("real_u8", vp), ("real_idx", vp)]


# name -> (restype, argtypes)      (keep in sync with include/jckgan.h; tests/test_abi.py checks the symbol list)
PROTOS = {
    "jck_last_error": (C.c_char_p, []),
    "jck_version": (i32, []),
    "jck_pad_rows": (i32, [i32]),
    "jck_pad_chan": (i32, [i32]),
    "jck_packed_bytes": (sz, [i32, i64]),
    "jck_stats_floats": (sz, [i64, i32, i32]),
    "jck_pack_down": (i32, [i32, vp, i32, i32, vp, vp]),
    "jck_pack_up": (i32, [i32, vp, i32, i32, vp, vp]),
    "jck_pack_g1": (i32, [i32, vp, i32, i32, i32, vp, vp]),
    "jck_pack_head": (i32, [vp, i32, vp, vp]),
    "jck_conv_down": (i32, [i32, vp, vp, vp, vp, C.POINTER(i32), i32, i32, i32, i32, i32, vp]),
    "jck_conv_up": (i32, [i32, vp, vp, vp, vp, C.POINTER(i32), i32, i32, i32, i32, i32, i32, vp]),
    "jck_conv_wgrad_ws_bytes": (sz, [i32, i32, i32, i32, i32]),
    "jck_conv_wgrad": (i32, [i32, vp, vp, vp, sz, vp, i32, i32, i32, i32, i32, i32, vp]),
    "jck_g1_fwd": (i32, [i32, vp, vp, vp, vp, C.POINTER(i32), i32, i32, i32, vp]),
    "jck_g1_wgrad_ws_bytes": (sz, [i32, i32, i32]),
    "jck_g1_wgrad": (i32, [i32, vp, vp, vp, sz, vp, i32, i32, i32, i32, i32, vp]),
    "jck_bn_finalize": (i32, [vp, i32, f32, vp, vp, vp, vp, vp, f32, f32, vp, i32, vp]),
    "jck_bn_act_fwd": (i32, [i32, vp, vp, f32, vp, i64, i32, vp]),
    "jck_bn_bwd_ws_floats": (sz, [i32]),
    "jck_bn_act_bwd": (i32, [i32, vp, vp, vp, f32, vp, vp, vp, vp, i64, i32, vp]),
    "jck_bn_finalize_grouped": (i32, [vp, i32, f32, vp, vp, f32, vp, vp, i32, i32, vp]),
    "jck_bn_act_fwd_grouped": (i32, [i32, vp, vp, f32, vp, i64, i32, i32, vp]),
    "jck_bn_fwd": (i32, [i32, vp, vp, i32, f32, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, f32, i64, i32, i32, vp]),
    "jck_conv_down_grouped": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "jck_conv_up_grouped": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "jck_bn_act_bwd_grouped": (i32, [i32, vp, vp, vp, f32, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "jck_grid_sync_bytes": (sz, []),
    "jck_debug_bnres_stamps": (i32, [vp]),
    "jck_grid_sync_error": (i32, [vp]),
    "jck_bn_act_bwd_res": (i32, [i32, vp, vp, vp, f32, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp]),
    "jck_img_prep": (i32, [i32, vp, vp, f32, f32, vp, i32, i32, vp]),
    "jck_resize_norm": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, f32, f32, vp, vp, vp]),
    "jck_img_prep_u8": (i32, [i32, vp, vp, vp, f32, f32, vp, vp, i32, i32, i32, vp]),
    "jck_nhwc4_to_nchw": (i32, [i32, vp, vp, i32, i32, vp]),
    "jck_axpy_noise": (i32, [i32, vp, vp, f32, f32, vp, i32, i32, vp]),
    "jck_interp": (i32, [i32, vp, vp, vp, vp, i32, i32, vp]),
    "jck_mix_interp": (i32, [i32, vp, vp, vp, i32, f32, f32, vp, vp, vp, vp, i32, i32, vp]),
    "jck_gp_norm": (i32, [i32, vp, i32, i32, vp, i32, i32, vp, vp]),
    "jck_tanh_bwd": (i32, [i32, vp, vp, f32, vp, i64, vp]),
    "jck_head_fwd": (i32, [i32, vp, vp, vp, i32, i32, f32, i32, vp, vp, vp, i32, i32, i32, vp]),
    "jck_head_fwd_grouped": (i32, [i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "jck_head_bwd_ws_floats": (sz, [i32]),
    "jck_pack_linear": (i32, [i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "jck_linear_fwd": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "jck_linear_finish": (i32, [i32, vp, i32, vp, vp, f32, vp, vp, i32, i32, vp]),
    "jck_linear_wgrad_ws_bytes": (sz, [i32, i32, i32]),
    "jck_linear_wgrad": (i32, [i32, vp, i32, vp, i32, vp, sz, vp, i32, i32, i32, vp]),
    "jck_unperm_linear_grad": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "jck_label_embed_fwd": (i32, [i32, vp, vp, vp, f32, i32, i32, i32, vp, i32, i32, vp, vp]),
    "jck_label_embed_bwd": (i32, [i32, vp, i32, i32, vp, vp, f32, i32, i32, i32, vp, vp, vp]),
    "jck_label_embed_fwd_tiled": (i32, [i32, vp, vp, vp, f32, i32, i32, i32, vp, i32, i32, vp, i32, vp]),
    "jck_label_embed_bwd_tiled": (i32, [i32, vp, i32, i32, vp, vp, f32, i32, i32, i32, vp, vp, i32, vp]),
    "jck_concat_rows": (i32, [i32, vp, i32, vp, i32, i32, vp]),
    "jck_split_rows": (i32, [i32, vp, i32, i32, vp, i32, vp]),
    "jck_dropout": (i32, [i32, vp, vp, f32, vp, i64, vp]),
    "jck_colsum": (i32, [i32, vp, i32, i32, i32, vp, vp]),
    "jck_sum_vec": (i32, [vp, i32, vp, vp]),
    "jck_cgan_z": (i32, [i32, vp, vp, i32, i32, i32, i32, vp, vp]),
    "jck_gp_grad": (i32, [i32, vp, vp, f32, i32, i32, vp, vp]),
    "jck_gp_head2": (i32, [i32, vp, vp, vp, i32, i32, vp, vp, vp, vp]),
    "jck_bn2_ws_floats": (sz, [i32]),
    "jck_bn2_vchain": (i32, [i32, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, i64, i32, vp]),
    "jck_bn2_reverse": (i32, [i32, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, i64, i32, vp]),
    "jck_head_bwd": (i32, [i32, vp, vp, vp, i32, i32, vp, vp, i32, vp, vp]),
    "jck_head_unpack_grad": (i32, [vp, i32, vp, i32, vp]),
    "jck_head_bwd_conv": (i32, [i32, vp, vp, vp, i32, i32, vp, vp, vp, vp]),
    "jck_head_bwd_conv2": (i32, [i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "jck_adam": (i32, [vp, vp, vp, vp, i64, f64, f64, f64, f64, i32, f32, vp]),
    "jck_engine_create": (i32, [C.POINTER(vp), i32, i32, i32]),
    "jck_engine_create_sized": (i32, [C.POINTER(vp), i32, i32, i32, i32]),
    "jck_engine_image_size": (i32, [vp]),
    "jck_engine_num_tensors_of": (i32, [vp, i32]),
    "jck_engine_tensor_info_of": (i32, [vp, i32, i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64),
                                        C.POINTER(i32)]),
    "jck_engine_arena_numel_of": (i64, [vp, i32, i32]),
    "jck_engine_destroy": (None, [vp]),
    "jck_engine_num_tensors": (i32, [i32, i32]),
    "jck_engine_tensor_info": (i32, [i32, i32, i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64),
                                     C.POINTER(i32)]),
    "jck_engine_workspace_bytes": (sz, [vp]),
    "jck_engine_arena_numel": (i64, [i32, i32, i32]),
    "jck_engine_bind": (i32, [vp, vp, sz] + [vp] * 12),
    "jck_engine_repack": (i32, [vp, i32, vp]),
    "jck_engine_phase": (i32, [vp, i32, C.POINTER(StepInputs), vp]),
    "jck_engine_order_after_tail": (i32, [vp, vp]),
    "jck_engine_drop_prefetch": (i32, [vp, vp]),
    "jck_engine_grad_tail": (i64, [vp, i32]),
    "jck_engine_check": (i32, [vp]),
    "jck_engine_scalars": (vp, [vp]),
    "jck_engine_scalars_at": (vp, [vp, i32]),
    "jck_engine_sample": (i32, [vp, vp, vp, i32, vp, vp]),
    "jck_engine_tensor": (vp, [vp, C.c_char_p, C.POINTER(i64)]),
    "jck_debug_tr_read": (i32, [vp, i32, vp, vp]),
    "jck_debug_wgrad_stamps": (i32, [vp, i32]),
    "jck_tune": (i32, [C.c_char_p, i32]),
    "jck_conv2d_nhwc_f32": (i32, [vp, vp, vp, vp, vp] + [i32] * 14 + [vp]),
    "jck_pool2d_nhwc_f32": (i32, [vp, vp] + [i32] * 10 + [vp]),
    "jck_global_avgpool_nhwc_f32": (i32, [vp, vp, i32, i32, i32, vp]),
    "jck_nchw_to_nhwc_f32": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "jck_mean_cov_f64": (i32, [vp, vp, vp, i32, i32, vp]),
    "jck_engine_set_step": (i32, [vp, i32, f32, vp]),
    "jck_engine_set_noise_seed": (i32, [vp, C.c_ulonglong]),
    "jck_step_rng": (i32, [vp, i32, C.c_ulonglong, vp, C.c_longlong, vp, C.c_longlong, vp, C.c_longlong, f32, vp]),
    "jck_img_prep_rng": (i32, [i32, vp, vp, i32, f32, f32, vp, i32, i32, vp]),
    "jck_img_prep_u8_rng": (i32, [i32, vp, vp, vp, i32, f32, f32, vp, i32, i32, i32, vp]),
    "jck_axpy_noise_rng": (i32, [i32, vp, vp, i32, f32, f32, vp, i32, i32, vp]),
    "jck_engine_capture_begin": (i32, [vp, vp]),
    "jck_engine_capture_end": (i32, [vp, vp, C.POINTER(vp)]),
    "jck_engine_capture_abort": (i32, [vp, vp]),
    "jck_graph_launch": (i32, [vp, vp]),
    "jck_graph_destroy": (None, [vp]),
    "jck_comm_unique_id": (i32, [vp]),
    "jck_comm_create": (i32, [C.POINTER(vp), vp, i32, i32]),
    "jck_comm_world": (i32, [vp]),
    "jck_comm_allreduce_enqueue": (i32, [vp, vp, sz, vp, C.POINTER(i32)]),
    "jck_comm_wait": (i32, [vp, i32, vp]),
    "jck_comm_destroy": (i32, [vp]),
    "jck_prof_enable": (i32, [i32]),
    "jck_prof_collect": (i32, [i32, C.POINTER(C.c_char_p), C.POINTER(i32), C.POINTER(f64), C.POINTER(f64), C.POINTER(f64), C.POINTER(vp)]),
}


class JckError(RuntimeError):
    pass


_cdll = None


def load_library(path=None):
    """Loads the shared library and attaches the prototypes.  Raises (never falls back) if missing."""
    global _cdll
    if _cdll is not None:
        return _cdll
    path = path or os.environ.get("JCKGAN_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise JckError(f"{path} not found: build it with `python __graft_entry__.py build` "
                       f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
    dll = C.CDLL(path)
    partial = os.environ.get("JCKGAN_ALLOW_PARTIAL") == "1"      # development only
    for name, (res, args) in PROTOS.items():
        if partial and not hasattr(dll, name):
            continue
        fn = getattr(dll, name)          # AttributeError here = header / library mismatch
        fn.restype, fn.argtypes = res, args
    if os.path.abspath(path) == os.path.abspath(LIB_PATH) and os.environ.get("JCKGAN_ALLOW_STALE") != "1":
        from . import build as _build
        want, got = _build.source_id(), dll.jck_version()
        if got != want:
            raise JckError(f"{path} was built from other sources than the ones beside it (jck_version {got}, sources {want}): "
                           f"rebuild with `python __graft_entry__.py build`")
    _cdll = dll
    return dll


def _arg(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


class _Lib:
    """`lib.jck_conv_down(prec, big, w_hi, ...)`: tensors become device pointers, a non-zero return raises."""

    def __getattr__(self, name):
        dll = load_library()
        fn = getattr(dll, name)

        def call(*args):
            for a in args:
                if isinstance(a, torch.Tensor):
                    if not a.is_cuda:
                        raise JckError(f"{name}: tensor argument is on {a.device}; the HIP path needs device memory")
                    if not a.is_contiguous():
                        raise JckError(f"{name}: non-contiguous tensor argument")
            r = fn(*[_arg(a) for a in args])
            if fn.restype is i32 and name not in ("jck_version", "jck_pad_rows", "jck_pad_chan", "jck_engine_num_tensors", "jck_prof_collect", "jck_grid_sync_error", "jck_comm_world") \
                    and r != 0:
                raise JckError(f"{name} failed ({r}): {dll.jck_last_error().decode()}")
            return r
        call.__name__ = name
        setattr(self, name, call)
        return call


lib = _Lib()


def cur_stream():
    return torch.cuda.current_stream().cuda_stream
