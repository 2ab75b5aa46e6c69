"""ctypes binding of ``libdadd_hip.so`` (C ABI declared in ``include/dadd_hip.h``).

There is no fallback: if the library is missing or a symbol is absent, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdadd_hip.so")
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ("igemm.hip", "igemm_dma.hip", "conv_halo.hip", "attn2_fused.hip", "ffn_block.hip", "tf_head.hip", "norm.hip", "attention.hip", "elementwise.hip",
           "conditioning.hip", "api.hip")

DADD_OK, DADD_EINVAL, DADD_EHIP, DADD_ESTATE = 0, -1, -2, -3
EPI_BIAS, EPI_ROWVEC, EPI_RESIDUAL, EPI_GEGLU = 1, 2, 4, 8
EPI_LNFOLD, EPI_QUICKGELU, EPI_GELU, EPI_SIGMOID, EPI_GNSTAT, EPI_LNSTAT = 128, 256, 512, 1024, 2048, 4096
PRE_GN, PRE_GN_SILU = 8192, 16384
EPI_GNAPPLY, EPI_GNAPPLY_SILU = 32768, 65536
TUNE_SHALLOW, TUNE_NODMA, TUNE_PERSIST = 16, 32, 64
XATTN_SPLIT, XATTN_BASELINE = 0, 1
GN_MAX_CHUNKS = 256

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class IgemmDesc(C.Structure):
    """Mirror of ``dadd_igemm_desc``."""
    _fields_ = [(n, vp) for n in ("x", "x2", "w", "out", "partial", "bias", "rowvec", "residual")] + \
               [(n, i32) for n in ("B", "Hi", "Wi", "C1", "C2", "Ho", "Wo", "N", "taps", "stride",
                                   "ups", "pad", "ldo", "ldr", "ld_rowvec", "splitk", "flags",
                                   "tile_n", "tile_m")] + [("counters", vp), ("ln_c1", vp), ("ln_eps", f32), ("gn_ws", vp), ("gn_nchunk", i32), ("gn_cg", i32),
                                                           ("ln_stats_out", vp), ("ln_stats_in", vp), ("ln_parts_out", i32), ("ln_parts_in", i32),
                                                           ("gn_in_ws", vp), ("gn_in_gamma", vp), ("gn_in_beta", vp), ("gn_in_nchunk", i32), ("gn_in_eps", f32),
                                                           ("gn_in_ws2", vp), ("gn_in_nchunk2", i32),
                                                           ("gn_out", vp), ("gn_out_gamma", vp), ("gn_out_beta", vp), ("gn_out_eps", f32)]


# name -> (restype, argtypes); every symbol include/dadd_hip.h declares
PROTOTYPES = {
    "dadd_last_error": (C.c_char_p, []),
    "dadd_version": (C.c_int, []),
    "dadd_init": (C.c_int, []),
    "dadd_device_info": (C.c_int, [C.c_int, C.POINTER(i64)]),
    "dadd_conv_igemm_f16": (C.c_int, [C.POINTER(IgemmDesc), vp]),
    "dadd_conv3x3_cin8_f16": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "dadd_conv_in_nchw_f16": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "dadd_conv3x3_cout4_f16": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, vp]),
    "dadd_conv_out_ddim_f16": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "dadd_pack_nchw_f32_to_nhwc8_f16": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp,
                                                  vp, vp]),
    "dadd_q_sample_f32": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, i64, vp]),
    "dadd_mse_rows_f32": (C.c_int, [vp, vp, vp, C.c_int, i64, vp]),
    "dadd_frames_to_u8": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "dadd_gaussian_sample_f32": (C.c_int, [vp, vp, vp, f32, vp, i64, vp]),
    "dadd_groupnorm_f16": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int,
                                     C.c_int, f32, C.c_int, C.c_int, vp]),
    "dadd_layernorm_f16": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, f32, vp]),
    "dadd_self_attn_f16": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, vp]),
    "dadd_attn2_fused_f16": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, f32, C.c_int, C.c_int, C.c_int, vp]),
    "dadd_ffn_block_f16": (C.c_int, [vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "dadd_ffn_block_bytes": (C.c_int, []),
    "dadd_tf_head_f16": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, f32, vp, vp, vp, f32, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "dadd_tf_head_bytes": (C.c_int, []),
    "dadd_tf_head_debug": (C.c_int, [vp]),
    "dadd_tri_xattn_f16": (C.c_int, [vp, vp, vp, vp, f32, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, vp]),
    "dadd_timestep_features_f32": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "dadd_linear_rows_f32": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "dadd_attn_f16": (C.c_int, [vp, vp, vp, vp] + [C.c_int] * 8 + [vp]),
    "dadd_clip_patch_rows_f16": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "dadd_aoe_interp_f32": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "dadd_purifier_tail_f16": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, f32, vp]),
    "dadd_begin_step": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "dadd_ddim_update_f32": (C.c_int, [vp, vp, vp, f32, vp, vp, i64, vp]),
    "dadd_prefetch": (C.c_int, [vp, i64, vp]),
    "dadd_prefetch_join": (C.c_int, [vp]),
    "dadd_graph_begin": (C.c_int, [vp]),
    "dadd_graph_end": (C.c_int, [vp, C.POINTER(vp)]),
    "dadd_graph_launch": (C.c_int, [vp, vp]),
    "dadd_graph_destroy": (C.c_int, [vp]),
    "dadd_prof_begin": (C.c_int, []),
    "dadd_prof_end": (C.c_int, [C.POINTER(C.c_int)]),
    "dadd_prof_record": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double)]),
}

_lib: Optional[C.CDLL] = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into ``libdadd_hip.so`` (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    import glob
    deps = sorted(set(srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")))) + \
        [os.path.join(os.path.dirname(_HERE), "include", "dadd_hip.h")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps)):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs.  With the default AGPR form the
    # attention softmax paid 144 v_accvgpr_read/write per 64-key tile (more than half of its VALU).
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
           "-amdgpu-mfma-vgpr-form=1", *os.environ.get("DADD_EXTRA_CFLAGS", "").split(), *srcs, "-o", LIB_PATH]   # (extra flags: diagnostics builds)
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def load() -> C.CDLL:
    """Load the library and bind every prototype; raises RuntimeError when it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  torch's bundled HIP runtime must be the one in the process: load it first
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is required (run __graft_entry__.build()); "
            "there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libdadd_hip.so does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Map a status code to the exception the reference would raise for the same condition."""
    if rc == DADD_OK:
        return
    msg = (_lib.dadd_last_error() or b"").decode() if _lib is not None else ""
    if rc == DADD_EINVAL:
        raise ValueError(msg)
    raise RuntimeError(f"libdadd_hip error {rc}: {msg}")
