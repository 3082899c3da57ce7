"""ctypes binding of ``csrc/libneube_hip.so`` (the C ABI declared in ``include/neube_hip.h``).

There is no CPU fallback: if the library is missing or a symbol cannot be resolved this raises, and
every product-path op goes through :func:`lib`.  (The reference silently falls back to slow
``_ref`` ops when its plugin build fails, ``torch_utils/ops/bias_act.py:47-50``; this build must not.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading

# torch MUST be imported before the library is dlopen'ed: the torch wheel bundles its own HIP runtime
# (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  Loaded first, it satisfies this library's
# NEEDED libamdhip64.so.7, so kernels, device pointers and streams all live in ONE runtime.  Loaded
# second, /opt/rocm's copy would come in beside it and launches fail with "no ROCm-capable device".
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libneube_hip.so")
ABI_VERSION = 11

_lock = threading.Lock()
_lib = None

f32p = C.POINTER(C.c_float)
vp = C.c_void_p


class NbLayerDesc(C.Structure):
    """Mirror of ``struct NbLayerDesc`` (include/neube_hip.h)."""
    _fields_ = [
        ("affine_w", C.c_uint64), ("affine_b", C.c_uint64), ("wsq", C.c_uint64), ("styles", C.c_uint64),
        ("dcoefs", C.c_uint64), ("noise_const", C.c_uint64), ("noise_lin", C.c_uint64), ("noise_out", C.c_uint64),
        ("noise_strength", C.c_uint64),
        ("c_aff", C.c_int32), ("n_plain", C.c_int32), ("c_out", C.c_int32), ("w_index", C.c_int32),
        ("res", C.c_int32), ("style_scale", C.c_float), ("pad_", C.c_int32 * 2),
    ]


# name -> (restype, argtypes); must list EVERY symbol declared in include/neube_hip.h
PROTOTYPES = {
    "nb_last_error": (C.c_char_p, []),
    "nb_abi_version": (C.c_int, []),
    "nb_bias_act_f32": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp]),
    "nb_bias_act_grad_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                       C.c_float, C.c_float, vp]),
    "nb_upfirdn2d_f32": (C.c_int, [vp, vp, vp] + [C.c_int] * 14 + [C.c_float, vp]),
    "nb_modconv3x3_up1_small_h3": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_float, C.c_float, C.c_float, vp]),
    "nb_modconv3x3_up2_small_h3": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_float, C.c_float, C.c_float, vp]),
    "nb_conv2d_f32": (C.c_int, [vp, vp, vp, vp, vp] + [C.c_int] * 9 + [vp]),
    "nb_conv2d_wgrad_f32": (C.c_int, [vp, vp, vp] + [C.c_int] * 9 + [vp]),
    "nb_conv2d_wgrad_h3": (C.c_int, [vp, vp, vp, vp] + [C.c_int] * 9 + [vp]),
    "nb_conv2d_wgrad_h3_ws_bytes": (C.c_longlong, [C.c_int] * 5),
    "nb_conv2d_wgrad_h3_ws": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, C.c_longlong] + [C.c_int] * 10 + [vp]),
    "nb_modconv_bwd_dot_f32": (C.c_int, [vp, vp, vp, C.c_longlong, vp, C.c_int, C.c_int, C.c_int, vp]),
    "nb_modconv_bwd_finish_f32": (C.c_int, [vp, C.c_longlong, C.c_longlong, C.c_longlong, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "nb_absmax_f32": (C.c_int, [vp, C.c_longlong, vp, C.c_longlong, vp, C.c_longlong, vp, vp]),
    "nb_pack_h2_ranged_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_float, vp, vp, C.c_int, vp]),
    "nb_mapping_f32": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "nb_mapping_ws_f32": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp]),
    "nb_styles_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "nb_styles_fast_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "nb_styles_noise_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp]),
    "nb_demod_coefs_f32": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "nb_noise_f32": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp]),
    "nb_norm_positions_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp]),
    "nb_modconv3x3_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int64, vp, vp,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp]),
    "nb_modconv3x3_variant": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "nb_calibrate_mfma_f16": (C.c_int, [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "nb_modconv3x3_up2_h3_variant": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "nb_pack_h2_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp]),
    "nb_pack_conv_weight_h3": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "nb_pack_conv_weight_h3_dev": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "nb_conv3x3_s2_valid_h3": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, vp] + [C.c_int] * 4 + [vp]),
    "nb_modconv3x3_up1_h3": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_float, C.c_float, C.c_float, vp]),
    "nb_modconv3x3_up2_h3": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_float, C.c_float, C.c_float, vp]),
    "nb_modconv3x3_up2_f32_h2": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int64, vp, vp, vp,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp]),
    "nb_torgb_triad_f32": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp,
                                     C.c_int, C.c_int, C.c_int, vp]),
    "nb_blend_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "nb_modconv3x3_up1_h3_ex": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int64, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp]),
    "nb_modconv3x3_up2_h3_ex": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int64, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp]),
    "nb_pack_h2f8_part_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "nb_pack_h2f8_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp]),
    "nb_pack_h2f6_part_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "nb_pack_h2f6_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp]),
    "nb_modconv3x3_up1_h3_torgb": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_float, C.c_float, C.c_float, vp, vp]),
    "nb_modconv3x3_up1_h3_h2": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int64, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp]),
    "nb_modconv3x3_up2_h3_h2": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int64, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp]),
    "nb_pack_h2_part_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "nb_geom_tiles_f32": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]),
    "nb_canvas_replay_f32": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int,
                                       vp, vp, vp]),
    "nb_canvas_replay_box_f32": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int,
                                       vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "nb_canvas_replay_pieces_f32": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp,
                                              C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "nb_paste_tiles_u8": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp]),
    "nb_enc_stem7x7_f32_h2": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "nb_enc_conv3x3_h3": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "nb_enc_conv3x3_h3_handoff": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_float, vp]),
    "nb_enc_upsample2x_h2": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "nb_enc_stem7x7_f32_h2_ex": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "nb_enc_conv3x3_ex": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "nb_enc_upsample2x_h2_ex": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "nb_enc_stem_conv3x3_f8": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "nb_pack_conv_weight": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
}


class NbTilePiece(C.Structure):
    """``struct NbTilePiece`` of include/neube_hip.h."""
    _fields_ = [("data", C.c_uint64), ("cstride", C.c_int32), ("rstride", C.c_int32), ("cy", C.c_int32), ("cx", C.c_int32),
                ("h", C.c_int32), ("w", C.c_int32), ("ly0", C.c_int32), ("lx0", C.c_int32)]


class NbNoiseSrc(C.Structure):
    """``struct NbNoiseSrc`` of include/neube_hip.h (noise computed inside the split-f16 convolutions)."""
    _fields_ = [("noise_const_t", vp), ("noise_lin", vp), ("noise_strength", vp), ("norm_pos", vp), ("positions", vp),
                ("res", C.c_int), ("img_resolution", C.c_int)]


NB_NOISE_IN_KERNEL = -1


class NbTorgbArgs(C.Structure):
    """``struct NbTorgbArgs`` of include/neube_hip.h."""
    _fields_ = [("styles", vp), ("w", vp), ("bias", vp), ("color_bias", vp), ("logits", vp), ("uvs", vp), ("img", vp),
                ("colors_out", vp), ("user_colors", vp), ("sfactor", vp), ("rgba_f32", vp), ("rgba_u8", vp),
                ("styles_stride_n", C.c_int), ("render_mode", C.c_int), ("clamp", C.c_float)]


class NeubeHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        from . import build as _build
        override = os.environ.get("NEUBE_LIB_PATH")          # developer A/B switch: load this prebuilt library instead
        if override:
            globals()["LIB_PATH"] = override
        if not override and os.path.exists(LIB_PATH) and not os.path.exists(_build.STAMP):
            # a prebuilt library without a build stamp (packaged deployment, older checkout, built by hand with other
            # flags): its sources cannot be compared, so it is loaded as it is -- the export and ABI-version checks below
            # still apply -- and says so
            import warnings
            warnings.warn(f"{LIB_PATH} has no build stamp ({os.path.basename(_build.STAMP)}): loading it without checking that "
                          f"it was built from the kernel sources in this tree (python -m brushstroke_engine_amd.build rebuilds)")
        elif not override and _build.is_stale():
            # missing, or built from other sources than the tree holds now (kernel edits without an ABI bump would
            # otherwise run the old code silently): rebuild if a compiler is here, else fail loudly.  Multi-process
            # callers build BEFORE init_process_group (bench.py, paint_image_main, tools/bench_*.py call build.build()
            # first), so that no rank sits in a collective timeout while another compiles.
            if os.path.exists(_build.HIPCC) and os.environ.get("NEUBE_NO_AUTOBUILD") != "1":
                try:
                    _build.build(verbose=False)
                except Exception as e:                                   # noqa: BLE001
                    raise NeubeHipError(f"rebuilding the stale HIP kernel library failed: {e}") from e
            else:
                raise NeubeHipError(
                    f"HIP kernel library at {LIB_PATH} is missing or was built from different sources. Build it with "
                    f"`python -m brushstroke_engine_amd.build` (needs hipcc, gfx950). There is no CPU fallback.")
        l = C.CDLL(globals()["LIB_PATH"])
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(l, name)
            except AttributeError as e:
                raise NeubeHipError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
            fn.restype = res
            fn.argtypes = args
        v = l.nb_abi_version()
        if v != ABI_VERSION:
            raise NeubeHipError(f"ABI version mismatch: library {v}, python {ABI_VERSION}; rebuild the library")
        _lib = l
        return _lib


def check(code: int, what: str) -> None:
    """Translate a negative NB_E* return code into the exception type the reference would raise
    (TORCH_CHECK failures surface as RuntimeError, bias_act.cpp:35-52)."""
    if code != 0:
        msg = lib().nb_last_error().decode("utf-8", "replace")
        raise NeubeHipError(f"{what} failed ({code}): {msg}")
