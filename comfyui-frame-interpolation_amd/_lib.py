"""ctypes binding of libvfi_hip.so (include/vfi_hip.h).

There is deliberately no fallback: if the library is missing or a call fails, a
``RuntimeError`` is raised — the product path never routes through torch ops or the oracle.

Two builds of the same sources exist (csrc/build.py): ``libvfi_hip.so`` — the product, exactly the entry points of
include/vfi_hip.h — and ``libvfi_hip_test.so``, which adds the test taps of include/vfi_hip_test.h (A/B switches between
two correct kernel forms, read-back of internal tensors).  The package loads the product library; ``use_test_build()`` —
called by tests/conftest.py and tools/, never by the package — selects the other one before the first ``load()``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvfi_hip.so")
TEST_LIB_PATH = os.path.join(_HERE, "libvfi_hip_test.so")

_lib = None
_test_build = False


def use_test_build():
    """Select libvfi_hip_test.so for this process (tests/ and tools/ only).  One process uses ONE library — it owns per-device
    workspaces — so this must come before the first load()."""
    global _test_build
    if _lib is not None and not _test_build:
        raise RuntimeError("use_test_build(): the product library is already loaded in this process")
    _test_build = True


def is_test_build():
    return _test_build

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)

# name -> (restype, argtypes); PROTOTYPES mirrors include/vfi_hip.h one to one, TEST_PROTOTYPES include/vfi_hip_test.h
TEST_NAMES = ("vfi_conv3x3_naive", "vfi_test_conv_algo", "vfi_test_pack_wino3x3", "vfi_test_pack_deconv3x3", "vfi_test_set_option", "vfi_test_variant_override", "vfi_test_wino_probe_read", "vfi_rife_debug_keep", "vfi_rife_debug_read", "vfi_test_film_schedule", "vfi_test_linspace01", "vfi_film_debug_read_flow", "vfi_m2m_debug_read")
PROTOTYPES = {
    "vfi_init": (C.c_int, [C.c_int]),
    "vfi_last_error": (C.c_char_p, []),
    "vfi_device_info": (C.c_int, [C.c_char_p, C.c_int, c_int_p]),
    "vfi_trace_enable": (C.c_int, [C.c_int]),
    "vfi_trace_reset": (C.c_int, []),
    "vfi_trace_report": (C.c_int, [C.c_char_p, C.c_int]),
    "vfi_clock_probe": (C.c_int, [C.c_void_p, C.c_int]),
    "vfi_clock_probe_names": (C.c_int, [C.c_char_p, C.c_int]),
    "vfi_warp_border": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_conv3x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "vfi_conv3x3_naive": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "vfi_test_conv_algo": (C.c_int, [C.c_int]),
    "vfi_test_pack_wino3x3": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, c_int_p, C.c_int, C.c_void_p, C.c_int64]),
    "vfi_test_pack_deconv3x3": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]),
    "vfi_test_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "vfi_test_variant_override": (C.c_int, [C.c_char_p]),
    "vfi_test_wino_probe_read": (C.c_int, [C.POINTER(C.c_uint32)]),
    "vfi_deconv4x4_ps2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p]),
    "vfi_conv_create": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "vfi_conv_destroy": (None, [C.c_void_p]),
    "vfi_conv_create_up2x2": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "vfi_conv_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_float, C.c_void_p]),
    "vfi_avgpool2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_upsample_nearest": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_void_p]),
    "vfi_resize_bilinear": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_float, C.c_void_p]),
    "vfi_warp_film": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_void_p]),
    "vfi_axpby": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_float,
                            C.c_float, C.c_void_p]),
    "vfi_softsplat_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_costvol9x9": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_conv_create_ex": (C.c_void_p, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p,
                                        C.c_int, C.c_void_p]),
    "vfi_conv_forward_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "vfi_m2m_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vfi_warp_m2m": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_void_p]),
    "vfi_pool_mean": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_m2m_cube_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_m2m_photo": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                C.c_void_p]),
    "vfi_m2m_splat_inputs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_int, C.c_void_p]),
    "vfi_m2m_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p]),
    "vfi_rife40_prep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vfi_warp_rife": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p]),
    "vfi_absmax": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "vfi_rife40_output": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_ifrnet_prep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vfi_ifrnet_center": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "vfi_conv7x7s2_prelu": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p]),
    "vfi_sigmoid": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "vfi_fill_items": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, c_float_p, C.c_void_p]),
    "vfi_resize_bilinear_ratio": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "vfi_ifrnet_output": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_pad_rgb": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_normalize_channels": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, c_float_p, c_float_p, C.c_void_p]),
    "vfi_prelu_scalar": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p]),
    "vfi_instnorm_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vfi_instnorm_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vfi_layernorm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vfi_layernorm_add": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int, C.c_void_p]),
    "vfi_gelu": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "vfi_window_partition": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_bmm_nt": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "vfi_bmm_nn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_softmax_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vfi_flow_sample": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_resize_bilinear_ac": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "vfi_local_match": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_local_propagate": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_convex_upsample": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_gmfss_metric_inputs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_tanh_scale": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p]),
    "vfi_splat_prep": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_void_p]),
    "vfi_splat_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "vfi_pixel_shuffle2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_clamp_crop": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_channel_pool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vfi_cbam_gate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vfi_cbam_scale_compress": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "vfi_cbam_spatial": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_convex_upsample_c": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_lerp_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "vfi_add_clamp01": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "vfi_ifunet_blend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_fill_channels": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p]),
    "vfi_rife_create": (C.c_void_p, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int]),
    "vfi_rife_destroy": (None, [C.c_void_p]),
    "vfi_rife_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "vfi_rife_load_frame": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vfi_rife_interpolate": (C.c_int, [C.c_void_p, C.c_int, c_int_p, c_int_p, c_float_p, C.c_void_p, C.c_void_p]),
    "vfi_rife_debug_keep": (C.c_int, [C.c_void_p, C.c_int]),
    "vfi_rife_debug_read": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "vfi_rife_work": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vfi_memcpy_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "vfi_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "vfi_stream_destroy": (C.c_int, [C.c_void_p]),
    "vfi_stream_spin": (C.c_int, [C.c_void_p, C.c_int]),
    "vfi_set_reserved_cus": (C.c_int, [C.c_int]),
    "vfi_get_reserved_cus": (C.c_int, []),
    "vfi_film_create": (C.c_void_p, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int]),
    "vfi_film_destroy": (None, [C.c_void_p]),
    "vfi_film_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vfi_film_release_workspace": (C.c_int, [C.c_void_p]),
    "vfi_film_two_streams": (C.c_int, [C.c_void_p, C.c_int]),
    "vfi_film_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, C.c_int, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_int64)]),
    "vfi_m2m_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, C.c_int, C.c_void_p, C.c_void_p,
                              C.POINTER(C.c_int64)]),
    "vfi_test_film_schedule": (C.c_int, [C.c_int, c_int_p, C.c_int]),
    "vfi_test_linspace01": (C.c_int, [C.c_int, C.c_void_p]),
    "vfi_film_debug_read_flow": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "vfi_m2m_create": (C.c_void_p, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int]),
    "vfi_m2m_destroy": (None, [C.c_void_p]),
    "vfi_m2m_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_m2m_render": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "vfi_m2m_release_workspace": (C.c_int, [C.c_void_p]),
    "vfi_m2m_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "vfi_m2m_debug_read": (C.c_int64, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "vfi_m2m_image4": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vfi_m2m_warp_image4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vfi_m2m_photo_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_void_p]),
    "vfi_m2m_render_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]),
    "vfi_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "vfi_window_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "vfi_rife_load_frame_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vfi_rife_load_frames": (C.c_int, [C.c_void_p, C.c_int, c_int_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]),
    "vfi_f32_to_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vfi_rife_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p,
                               C.POINTER(C.c_int64)]),
    "vfi_rife_clone_empty": (C.c_void_p, [C.c_void_p]),
    "vfi_rife_weights": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "vfi_comm_create": (C.c_void_p, [C.c_int, c_int_p]),
    "vfi_comm_destroy": (None, [C.c_void_p]),
    "vfi_comm_size": (C.c_int, [C.c_void_p]),
    "vfi_comm_broadcast": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    "vfi_comm_all_gather_v": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]),
    "vfi_comm_plan_all_gather": (C.c_int64, [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64]),
    "vfi_comm_all_gather_mode": (C.c_int, []),
}

TEST_PROTOTYPES = {k: PROTOTYPES.pop(k) for k in TEST_NAMES}


# The runtime variables this package honours — all of them select resources or diagnostics, none changes a frame's values
# (INTEGRATION.md "Environment").  Kernel A/B switches are NOT environment variables: include/vfi_hip_test.h, vfi_test_set_option.
SUPPORTED_ENV = {
    "VFI_DEVICES":        "devices one node call may use: 'current' (default) | 'all' | '0,1,2,3' (multidev.py)",
    "VFI_ALLGATHER":      "device-side all-gather of new frames: 'rccl' (default) | 'direct' (csrc/comm.hip)",
    "VFI_MODEL_CACHE":    "'0': do not keep engines between node calls (ckpt.py)",
    "VFI_PAIR_LANES":     "frame pairs in flight per GPU in the M2M / FILM / GMFSS / IFUNet / IFRNet nodes, one engine + HIP stream each (lanes.py; default per model 2-3, 1 = one stream; frames are bit-identical)",
    "VFI_RIFE_MIN_BATCH": "smallest tasks-per-launch the RIFE node uses whatever its batch_size widget says (default 8)",
    "VFI_HOST_WORKERS":   "host copy worker threads 'upload,download,...' (hostpipe.py)",
    "VFI_HOST_THP":       "'0': no transparent-huge-page advice on the pinned staging ring (hostpipe.py)",
    "VFI_HOST_PROFILE":   "'1': per-phase wall-clock accounting of the host pipeline (hostpipe.py)",
    "VFI_TRACE_SHAPES":   "per-shape rows in vfi_trace_report (profiling)",
}
_TEST_HARNESS_ENV = {"VFI_TEST_OPTIONS", "VFI_HOSTCHECK", "VFI_CHILD", "VFI_REAL_CKPTS"}      # read by tests/, never by the package or the library


def audit_environment():
    """Names of ``VFI_*`` variables in the environment that nothing in this package reads.  They are IGNORED (the library has no
    experiment switches in its environment any more); saying so loudly beats a user believing a stale switch is in force."""
    stray = sorted(k for k in os.environ if k.startswith("VFI_") and k not in SUPPORTED_ENV and k not in _TEST_HARNESS_ENV)
    if stray:
        import warnings

        warnings.warn("ignored environment variable(s) " + ", ".join(stray) + ": not one of " + ", ".join(sorted(SUPPORTED_ENV)) +
                      " (kernel A/B switches are set through vfi_test_set_option, include/vfi_hip_test.h)", RuntimeWarning, stacklevel=3)
    return stray


def load():
    """Load the shared library (building nothing: run ``__graft_entry__.build()`` first)."""
    global _lib
    if _lib is not None:
        return _lib
    audit_environment()
    path = TEST_LIB_PATH if _test_build else LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the VFI hot path.")
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7 — the same SONAME as the ROCm
    # install this library was linked against).  Whichever is mapped first serves both; if ours came first, torch would
    # run on a runtime it was not built with and the second initialisation finds no device.  So: torch first, always.
    import torch  # noqa: F401

    lib = C.CDLL(path)
    protos = dict(PROTOTYPES)
    if _test_build:
        protos.update(TEST_PROTOTYPES)
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def test_tap(name):
    """A test tap of include/vfi_hip_test.h, or a RuntimeError that says why it is not there (the product library has none)."""
    lib = load()
    if not _test_build:
        raise RuntimeError(f"{name} is a test tap (include/vfi_hip_test.h): only libvfi_hip_test.so has it — call "
                           "cfi_amd._lib.use_test_build() before the library is first loaded (tests/conftest.py does)")
    return getattr(lib, name)


def last_error():
    return (load().vfi_last_error() or b"").decode(errors="replace")


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def stream_ptr():
    """hipStream_t of torch's current stream, so library launches order with torch copies."""
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_own_streams = {}      # device index -> idle streams made by vfi_stream_create (kept for the life of the process, handed out exclusively)


class OwnStream:
    """A HIP stream made by the library for ONE owner at a time (a pair lane, an engine's graph capture), as a torch stream object.
    torch.cuda.Stream() draws from a shared pool of 32 per device, round robin: two live engines could be handed the same stream —
    and with it the same (device, stream)-keyed library scratch, whose addresses their captured graphs have baked in."""

    def __init__(self, device):
        import torch

        self.index = torch.device(device).index or 0
        idle = _own_streams.setdefault(self.index, [])
        if idle:
            self.ptr = idle.pop()
        else:
            out = C.c_void_p()
            with torch.cuda.device(self.index):
                check(load().vfi_stream_create(C.byref(out)), "vfi_stream_create")
            self.ptr = out.value
        self.stream = torch.cuda.ExternalStream(self.ptr, device=torch.device("cuda", self.index))

    def release(self):
        """back to the idle list (after the owner has drained it)"""
        if self.ptr is not None:
            _own_streams.setdefault(self.index, []).append(self.ptr)
            self.ptr = self.stream = None


def streams_share_queue(a, b, spin_us=300):
    """Do torch streams a and b (same device) run on ONE hardware queue?  The HIP runtime binds a stream to one of a handful of hardware
    queues (4 by default) at its first use; two streams of one queue execute strictly in order — two pair lanes on them overlap nothing
    (measured: M2M with three lanes 6.75 instead of 6.17 ms per pair whenever two of the lanes' streams had landed on one queue).
    Probe: one spinning workgroup on each; together they take 1x the spin on different queues, 2x on the same."""
    import time

    lib = load()
    pa, pb = C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream)
    for s in (pa, pb):                      # first use binds the queue (and loads the kernel)
        check(lib.vfi_stream_spin(s, 1), "vfi_stream_spin")
    votes = 0
    for _ in range(3):
        a.synchronize(); b.synchronize()
        t0 = time.perf_counter()
        check(lib.vfi_stream_spin(pa, spin_us), "vfi_stream_spin")
        check(lib.vfi_stream_spin(pb, spin_us), "vfi_stream_spin")
        a.synchronize(); b.synchronize()
        votes += (time.perf_counter() - t0) * 1e6 > 1.6 * spin_us
    return votes >= 2


def own_streams_apart(device, k, avoid=(), tries=12):
    """k OwnStreams of `device` that pairwise do not share a hardware queue, and none of which shares one with a busy stream in `avoid`
    (best effort: with 4 hardware queues and copy streams beside the lanes not every wish can be met — the mutual condition comes
    first, `avoid` is dropped stream by stream from its end when the candidates run out).  Rejected candidates return to the idle list."""
    import torch

    if torch.device(device).type != "cuda" or (k <= 1 and not avoid):
        return [OwnStream(device) for _ in range(k)]
    avoid = list(avoid)
    while True:
        chosen, rejected = [], []
        for _ in range(tries + k):
            if len(chosen) == k:
                break
            c = OwnStream(device)
            if any(streams_share_queue(c.stream, o.stream) for o in chosen) or any(streams_share_queue(c.stream, o) for o in avoid):
                rejected.append(c)
            else:
                chosen.append(c)
        if len(chosen) == k or not avoid:
            while len(chosen) < k:            # no luck at all: lanes that share a queue still give right frames
                chosen.append(rejected.pop() if rejected else OwnStream(device))
            for r in rejected:
                r.release()
            return chosen
        for r in rejected + chosen:
            r.release()
        avoid.pop()


def clock_probe_names():
    buf = C.create_string_buffer(1 << 18)
    n = load().vfi_clock_probe_names(buf, len(buf))
    if n < 0:
        raise RuntimeError(f"vfi_clock_probe_names failed: {last_error()}")
    return buf.value.decode().splitlines()


def trace_report():
    buf = C.create_string_buffer(1 << 16)
    check(load().vfi_trace_report(buf, len(buf)), "vfi_trace_report")
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.split()
        out[name] = (int(calls), float(ms))
    return out
