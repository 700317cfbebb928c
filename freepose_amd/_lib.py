"""ctypes binding of libfreepose_hip.so (the C ABI in include/freepose_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is
raised.  torch is only used by callers for device memory and streams (tensor.data_ptr(), current stream).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libfreepose_hip.so"
# the lab build (python -m freepose_amd.build --lab): the same sources with -DFP_LAB, i.e. plus the measurement variants and hooks of
# tools/ (alternative GEMM loops, attention ring depths, FP_* environment toggles).  Never loaded by the product: a tool opts in by
# calling use_lab() before the first load().
LAB_LIB_PATH = _HERE / "lib" / "libfreepose_hip_lab.so"

_lib = None
_use_lab = False


def use_lab():
    """tools/ only: make load() open libfreepose_hip_lab.so (must be called before anything loaded the product library)"""
    global _use_lab
    if _lib is not None and not _use_lab:
        raise RuntimeError("use_lab() after the product library was loaded")
    _use_lab = True


def is_lab() -> bool:
    return _use_lab

c_void_p, c_int, c_float, c_double, c_size_t, c_char_p = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_char_p
P = C.POINTER


class VitArch(C.Structure):
    _fields_ = [("dim", c_int), ("depth", c_int), ("heads", c_int), ("mlp_dim", c_int), ("patch", c_int),
                ("n_reg", c_int), ("pos_grid", c_int), ("ln_eps", c_float)]


# name -> (restype, argtypes); mirrors include/freepose_hip.h one to one
SIGNATURES = {
    "fp_last_error": (c_char_p, []),
    "fp_version": (c_int, []),
    "fp_ctx_create": (c_int, [c_int, P(c_void_p)]),
    "fp_ctx_destroy": (c_int, [c_void_p]),
    "fp_ctx_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "fp_ctx_workspace_bytes": (c_size_t, [c_void_p]),
    "fp_vit_create": (c_int, [c_void_p, P(VitArch), P(c_void_p)]),
    "fp_vit_destroy": (c_int, [c_void_p]),
    "fp_vit_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, c_size_t, c_void_p]),
    "fp_vit_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fp_vit_flops": (c_double, [c_void_p, c_int, c_int, c_int, c_int]),
    "fp_ffa": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fp_bank_prepare": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fp_bank_topk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fp_topk_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fp_rerank_views": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fp_l2_normalize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fp_template_score": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fp_template_score_normed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fp_crop_resize_pad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float,
                                   c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "fp_roi_align": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float,
                             c_void_p, c_void_p]),
    "fp_generate_rotations": (c_int, [c_int, c_void_p]),
    "fp_geodesic_select": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_double, c_void_p, P(c_int), c_void_p]),
    "fp_mesh_upload": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, P(c_void_p)]),
    "fp_mesh_upload_textured": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                        P(c_void_p)]),
    "fp_mesh_set_ambient": (c_int, [c_void_p, c_float]),
    "fp_mesh_set_shading": (c_int, [c_void_p, c_int]),
    "fp_mesh_set_filter": (c_int, [c_void_p, c_int]),
    "fp_mesh_set_cull": (c_int, [c_void_p, c_int]),
    "fp_project_vertices": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float, c_void_p,
                                    c_void_p, c_void_p]),
    "fp_mesh_destroy": (c_int, [c_void_p]),
    "fp_rasterize": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float, c_int,
                             c_int, c_void_p, c_void_p, c_void_p]),
    "fp_rasterize_extents": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_float, c_int,
                                     c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fp_depth_extents": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_void_p,
                                 c_void_p]),
    "fp_comm_unique_id": (c_int, [c_void_p]),
    "fp_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "fp_comm_destroy": (c_int, [c_void_p]),
    "fp_comm_size": (c_int, [c_void_p]),
    "fp_comm_rank": (c_int, [c_void_p]),
    "fp_allgather_bytes": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "fp_allgather_topk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fp_allgather_poses": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fp_op_gemm": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                           c_int, c_int, c_int, c_void_p]),
    "fp_op_gemm_vt": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                              c_void_p]),
    "fp_op_ln_linear": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p, c_int, c_int,
                                c_int, c_int, c_float, c_void_p, c_void_p]),
    "fp_op_gemm_stats": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_int, c_int, c_float, c_void_p, c_void_p]),
    "fp_op_gelu": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fp_op_attention": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fp_op_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "fp_op_im2col_norm": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fp_timer_create": (c_int, [P(c_void_p)]),
    "fp_timer_start": (c_int, [c_void_p, c_void_p]),
    "fp_timer_stop": (c_int, [c_void_p, c_void_p]),
    "fp_timer_elapsed_ms": (c_int, [c_void_p, P(c_float)]),
    "fp_timer_destroy": (c_int, [c_void_p]),
    "fp_vit_profile": (c_int, [c_void_p, c_int]),
    "fp_vit_profile_read": (c_int, [c_void_p, P(c_float), P(c_float), P(c_float), P(c_double)]),
    "fp_vit_profile_gemm_launches": (C.c_long, [c_void_p]),
}


def load(path: os.PathLike | None = None):
    """dlopen the library and attach prototypes.  Raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else (LAB_LIB_PATH if _use_lab else LIB_PATH)
    if not p.exists():
        raise RuntimeError(
            f"{p} not found: the HIP extension is not built (run `python -m freepose_amd.build`). "
            "freepose_amd has no CPU fallback.")
    lib = C.CDLL(str(p), mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if hasattr(lib, "fp_lab_set_option"):   # lab build only
        lib.fp_lab_set_option.restype = c_int
        lib.fp_lab_set_option.argtypes = [c_char_p, c_int]
        lib.fp_lab_read_scratch.restype = c_int
        lib.fp_lab_read_scratch.argtypes = [c_void_p, c_void_p, c_size_t]
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().fp_last_error()
        raise RuntimeError(f"libfreepose_hip {what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """device (or host) address of a torch tensor / numpy array, None -> NULL"""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
