"""ctypes binding of include/demonet_hip.h. Fails loudly when the HIP library is missing: there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEMONET_HIP_LIB") or os.path.join(_HERE, "lib", "libdemonet_hip.so")      # (DEMONET_HIP_LIB: a dev build of the SAME C ABI, e.g. `python -m demonet_amd.build --stamps`)
DN_ABI_VERSION = 1

DN_OP = dict(stem=1, pw=2, dw=3, se=4, conv=5, maxpool=6, l2norm=7)
DN_T = dict(act=0, image=1, vec=2, pool=3)


class TensorDesc(C.Structure):
    _fields_ = [("c", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("kind", C.c_int32)]


class OpDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("inp", C.c_int32), ("out", C.c_int32), ("residual", C.c_int32),
                ("se", C.c_int32), ("pool", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("dil", C.c_int32), ("act", C.c_int32),
                ("head", C.c_int32), ("level", C.c_int32), ("squeeze", C.c_int32), ("ceil_mode", C.c_int32),
                ("pool_pixels", C.c_int32), ("reserved", C.c_int32 * 2),
                ("w_off", C.c_int64), ("b_off", C.c_int64), ("w2_off", C.c_int64), ("b2_off", C.c_int64)]


class ModelDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_tensors", C.c_int32), ("n_ops", C.c_int32),
                ("tensors", C.POINTER(TensorDesc)), ("ops", C.POINTER(OpDesc)),
                ("input_tensor", C.c_int32), ("image_h", C.c_int32), ("image_w", C.c_int32),
                ("mean", C.c_float * 3), ("std", C.c_float * 3),
                ("num_classes", C.c_int32), ("n_levels", C.c_int32),
                ("level_tensor", C.c_int32 * 8), ("anchors_per_loc", C.c_int32 * 8),
                ("num_anchors", C.c_int32), ("anchors", C.POINTER(C.c_float)),
                ("score_thresh", C.c_float), ("nms_thresh", C.c_float),
                ("detections_per_img", C.c_int32), ("topk_candidates", C.c_int32)]


_SIGNATURES = {
    "dn_abi_version": (C.c_int, []),
    "dn_last_error": (C.c_char_p, []),
    "dn_create": (C.c_int, [C.POINTER(ModelDesc), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dn_destroy": (None, [C.c_void_p]),
    "dn_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "dn_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dn_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dn_forward_heads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dn_head_outputs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "dn_tensor_ptr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "dn_postprocess_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dn_postprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                 C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dn_pointwise_conv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "dn_depthwise_conv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dn_dense_conv": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 10 + [C.c_void_p]),
    "dn_expand_depthwise": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 11 + [C.c_void_p]),
    "dn_expand_depthwise_tiles": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "dn_ssd_loss_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dn_ssd_loss": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_float, C.c_float] + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dn_set_graph_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "dn_set_packed_output": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dn_profile_begin": (C.c_int, [C.c_void_p]),
    "dn_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "dn_batch_split": (C.c_int, [C.c_void_p, C.c_int]),
    "dn_set_chains": (C.c_int, [C.c_void_p, C.c_int]),
    "dn_profile_op_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int32)]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Loads libdemonet_hip.so (built in-tree by `python -m demonet_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m demonet_amd.build` (hipcc, gfx950). "
            "demonet_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.dn_abi_version() != DN_ABI_VERSION:
        raise HipLibraryMissing(f"ABI mismatch: library {L.dn_abi_version()} vs binding {DN_ABI_VERSION}; rebuild")
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc < 0:
        msg = lib().dn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"demonet_hip {what} failed ({rc}): {msg}")
    return rc
