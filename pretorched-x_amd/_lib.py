"""ctypes binding of libptx_amd.so (the C ABI declared in include/ptx_amd.h).

There is deliberately no fallback: if the shared library is missing or a symbol cannot be
resolved this module raises, and every model in the package is unusable (SURVEY.md 8b:
"the product path must fail loudly when the HIP extension is missing").
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptx_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ptx_amd.h")

PTX_EPI_RELU = 1
PTX_EPI_RES_ADD = 2
PTX_EPI_RES_PADA = 4
PTX_PRO_RELU = 8
PTX_EPI_ACCUM = 16
PTX_EPI_RES_UP = 64
PTX_F16_OPERANDS = 128
PTX_F16X3_OPERANDS = 0x8000
PTX_SPLITK_FUSED = 0x10000
PTX_ACT_OUT_F16 = 0x100
PTX_EPI_OUT_F16, PTX_EPI_AFFINE, PTX_EPI_DUAL_RAW, PTX_RES_F16, PTX_PRO_UP2, PTX_EPI_TANH = 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000


class PtxError(RuntimeError):
    """`status` carries the library's status code (PTX_ERR_*: 1 invalid, 2 unsupported, 3 HIP, 4 workspace) when the
    error came from a libptx_amd call, else None."""
    status = None


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("N", "Ti", "Hi", "Wi", "Ci", "ldx", "To", "Ho", "Wo", "Co", "ldy",
                 "kT", "kH", "kW", "sT", "sH", "sW", "pT", "pH", "pW", "Kc", "Co_pad")] + \
               [("flags", C.c_uint32)] + \
               [(n, C.c_int32) for n in
                ("ldr", "res_C", "res_T", "res_H", "res_W", "res_sT", "res_sH", "res_sW",
                 "x2_C", "x2_ld", "x2_T", "x2_H", "x2_W", "x2_sT", "x2_sH", "x2_sW", "groups")]

    def key(self):
        """Identity of a conv PROBLEM for the tuned-tile table: every field, minus flag bits that do not change which
        tile is fastest by construction of the table (PTX_SPLITK_FUSED arrived after the table was keyed)."""
        return tuple((getattr(self, f) & ~0x10000) if f == "flags" else getattr(self, f) for f, _ in self._fields_)


class ConvStage(C.Structure):
    """ptx_conv_stage: one convolution of a conv program (ptx_conv_program_*)."""
    _fields_ = [("desc", ConvDesc), ("x", C.c_void_p), ("x2", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p),
                ("res", C.c_void_p), ("y", C.c_void_p), ("tile", C.c_int32), ("split_k", C.c_int32)]


class ConvProgramInfo(C.Structure):
    """ptx_conv_program_info: sizes of a planned / built conv program."""
    _fields_ = [("n_stages", C.c_int32), ("total_items", C.c_int32), ("ctrl_words", C.c_int32), ("lds_bytes", C.c_int32),
                ("launches_replaced", C.c_int32), ("n_chunks", C.c_int32), ("image_bytes", C.c_uint64),
                ("workspace_bytes", C.c_uint64)]


class ConvFusedExt(C.Structure):
    """ptx_conv_fused_ext: operands of the fused generator-stage epilogue."""
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("ld_affine", C.c_int32), ("ld_raw", C.c_int32),
                ("y_raw", C.c_void_p)]


class PackDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("Co", "Ci", "kT", "kH", "kW", "Kc", "Co_pad", "fold_kw",
                                         "ld_k", "k_off", "bias_accumulate", "sub_groups", "co_per_super", "f16")]


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("N", "Ti", "Hi", "Wi", "C", "ld", "To", "Ho", "Wo",
                 "kT", "kH", "kW", "sT", "sH", "sW", "pT", "pH", "pW", "ldy")] + [("flags", C.c_uint32)]


class NormDesc(C.Structure):
    """ptx_norm_desc: TransformImage's tensor half (reference transforms/utils.py:72-75)."""
    _fields_ = [("mean", C.c_float * 4), ("std", C.c_float * 4), ("swap_rb", C.c_int32), ("to_255", C.c_int32)]

    @classmethod
    def make(cls, mean, std, input_space="RGB", input_range=(0, 1)):
        d = cls()
        for i in range(4):
            d.mean[i] = float(mean[i]) if i < len(mean) else 0.0
            d.std[i] = float(std[i]) if i < len(std) else 1.0
        d.swap_rb = int(input_space == "BGR")
        d.to_255 = int(max(input_range) == 255)
        return d


PTX_POOL_SAME, PTX_POOL_PAD_ZERO = 1, 2
PTX_REL_MAX_SETS, PTX_REL_MAX_FRAMES = 8, 16


class RelationDesc(C.Structure):
    """ptx_relation_desc: frame subsets of one TRN relation scale (reference trn.py:101-110)."""
    _fields_ = [("B", C.c_int32), ("n_sets", C.c_int32), ("n_frames", C.c_int32), ("frame_len", C.c_int32),
                ("idx", (C.c_int32 * PTX_REL_MAX_FRAMES) * PTX_REL_MAX_SETS)]


class NonlocalDesc(C.Structure):
    """ptx_nonlocal_desc: fused theta^T phi -> softmax -> . g (reference nonlocalnet.py:143-166)."""
    _fields_ = [(n, C.c_int32) for n in ("batch", "Nq", "Nk", "d", "dv", "ld_theta", "ld_phi", "ld_g", "ld_y")] + \
               [(n, C.c_int64) for n in ("bs_theta", "bs_phi", "bs_g", "bs_y")] + [("mode", C.c_int32)]


PTX_NL_SOFTMAX, PTX_NL_SCALE, PTX_NL_F16, PTX_NL_X3, PTX_NL_RELU, PTX_NL_OUT_F16 = 0, 1, 2, 4, 8, 16


class RgbConvDesc(C.Structure):
    """ptx_rgb_conv_desc: the generator's output layer BN -> ReLU -> conv3x3(C -> 3) -> tanh in one launch."""
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "C", "ldx", "ldy", "ld_affine")] + [("flags", C.c_uint32)]


_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_U = C.c_uint32
_Z = C.c_size_t

# name -> (restype, argtypes); must list every function declared in include/ptx_amd.h
SIGNATURES = {
    "ptx_version": (C.c_char_p, []),
    "ptx_last_error": (C.c_char_p, []),
    "ptx_conv3d_num_configs": (C.c_int, []),
    "ptx_conv3d_config_name": (C.c_char_p, [C.c_int]),
    "ptx_conv3d_config_supported": (C.c_int, [C.POINTER(ConvDesc), C.c_int]),
    "ptx_conv3d_pick_config": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int)]),
    "ptx_conv3d_workspace_bytes": (_Z, [C.POINTER(ConvDesc), C.c_int]),
    "ptx_conv3d_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _Z, C.c_int, C.c_int, _P]),
    "ptx_conv3d_fused_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, C.POINTER(ConvFusedExt), _P, _Z, C.c_int,
                                       C.c_int, _P]),
    "ptx_conv3d_dual_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _Z, C.c_int, C.c_int, _P]),
    "ptx_conv3d_chain_num_configs": (C.c_int, []),
    "ptx_conv3d_chain_config_name": (C.c_char_p, [C.c_int]),
    "ptx_conv3d_chain_supported": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_int]),
    "ptx_conv3d_chain_pick_config": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
    "ptx_conv3d_chain_fwd": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "ptx_conv_body_f32_supported": (C.c_int, [C.POINTER(ConvDesc), C.c_int]),
    "ptx_conv_body_f32_weight_elems": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "ptx_pack_conv_body_f32_weight": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P]),
    "ptx_conv_body_f32_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, C.c_int, _P]),
    "ptx_conv_body_chain_f32_supported": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_int]),
    "ptx_conv_body_tail_f32_weight_elems": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "ptx_pack_conv_body_tail_f32_weight": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P]),
    "ptx_conv_body_chain_f32_fwd": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "ptx_conv_program_num_tiles": (C.c_int, []),
    "ptx_conv_program_tile_name": (C.c_char_p, [C.c_int]),
    "ptx_conv_program_plan": (C.c_int, [C.POINTER(ConvStage), _I, C.POINTER(ConvProgramInfo)]),
    "ptx_conv_program_describe": (C.c_int, [C.POINTER(ConvStage), _I, C.c_char_p, _Z]),
    "ptx_conv_program_build": (C.c_int, [C.POINTER(ConvStage), _I, _P, _Z, _P, _Z, C.POINTER(ConvProgramInfo)]),
    "ptx_conv_program_fwd": (C.c_int, [C.POINTER(ConvProgramInfo), _P, _P, _I, _P]),
    "ptx_conv_program_trace_fwd": (C.c_int, [C.POINTER(ConvProgramInfo), _P, _P, _I, _P, _Z, _P]),
    "ptx_conv_program_error": (C.c_int, [_P, C.POINTER(C.c_int32), _P]),
    "ptx_ncdhw_to_split4": (C.c_int, [_P, _P, _I, _I, _L, _P]),
    "ptx_conv_stem_x3_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "ptx_conv_stem_x3_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "ptx_ncdhw_to_split_planes": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ptx_conv_stem_x3p_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "ptx_stem_x3p_weight_elems": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "ptx_pack_stem_x3p_weight": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P]),
    "ptx_conv_stem_x3p_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "ptx_conv_stem_f32_supported": (C.c_int, [C.POINTER(ConvDesc), _L, _L, _L]),
    "ptx_stem_f32_weight_elems": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "ptx_pack_stem_f32_weight": (C.c_int, [C.POINTER(ConvDesc), _P, _I, _P, _P]),
    "ptx_conv_stem_f32_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _L, _L, _L, _P, _P, _P, _P]),
    "ptx_packed_weight_elems": (_Z, [C.POINTER(PackDesc)]),
    "ptx_pack_conv_weight": (C.c_int, [C.POINTER(PackDesc), _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P]),
    "ptx_checksum_f32": (C.c_int, [_P, _I, _P, _P]),
    "ptx_ncdhw_to_ndhwc": (C.c_int, [_P, _P, _I, _I, _L, _I, _P]),
    "ptx_ndhwc_to_ncdhw": (C.c_int, [_P, _P, _I, _I, _L, _I, _P]),
    "ptx_fold_kw_ncdhw": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ptx_fold_kw_strided": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _L, _L, _L, _I, _I, _I, _I, _I, _P]),
    "ptx_pad_rows": (C.c_int, [_P, _P, _L, _I, _I, _P]),
    "ptx_frames_u8_to_ncdhw": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, C.POINTER(NormDesc), _P]),
    "ptx_fold_kw_frames_u8": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(NormDesc), _P]),
    "ptx_maxpool3d_fwd": (C.c_int, [C.POINTER(PoolDesc), _P, _P, _P]),
    "ptx_cbn_fold": (C.c_int, [_P, _P, _P, _P, C.c_float, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ptx_affine_act_upsample": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ptx_outer_sum_relu": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ptx_copy2d": (C.c_int, [_P, _P, _L, _I, _L, _L, _P]),
    "ptx_window_mean": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ptx_global_avgpool": (C.c_int, [_P, _P, _I, _I, _L, _I, _I, _P]),
    "ptx_linear_fwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _U, _P]),
    "ptx_relation_linear_fwd": (C.c_int, [C.POINTER(RelationDesc), _P, _I, _P, _P, _P, _I, _I, _U, _P]),
    "ptx_linear_setsum_fwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _U, _P]),
    "ptx_nonlocal_supported": (C.c_int, [C.POINTER(NonlocalDesc)]),
    "ptx_nonlocal_fwd": (C.c_int, [C.POINTER(NonlocalDesc), _P, _P, _P, _P, _P]),
    "ptx_nonlocal_workspace_bytes": (C.c_size_t, [C.POINTER(NonlocalDesc)]),
    "ptx_nonlocal_ws_fwd": (C.c_int, [C.POINTER(NonlocalDesc), _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "ptx_bgemm_nt": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _P]),
    "ptx_softmax_rows": (C.c_int, [_P, _L, _I, _I, _I, _P]),
    "ptx_transpose_last2": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ptx_conv1x1_skip_f16_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "ptx_conv1x1_skip_f16_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, C.POINTER(ConvFusedExt), _P]),
    "ptx_conv1x1_pro_f16_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "ptx_conv1x1_pro_f16_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, C.POINTER(ConvFusedExt), _P, _P, _P, C.POINTER(ConvFusedExt), _P]),
    "ptx_conv3x3_f16_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "ptx_conv3x3_f16_fwd": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, C.POINTER(ConvFusedExt), _P]),
    "ptx_rgb_conv3x3_f16_supported": (C.c_int, [C.POINTER(RgbConvDesc)]),
    "ptx_rgb_conv_weight_elems": (_Z, [_I]),
    "ptx_pack_rgb_conv_weight": (C.c_int, [_P, _I, _P, _P]),
    "ptx_rgb_conv3x3_f16_fwd": (C.c_int, [C.POINTER(RgbConvDesc), _P, _P, _P, _P, _P, _P, _P]),
}


def header_symbols(path=HEADER_PATH):
    """Every function name declared in include/ptx_amd.h (used by the no-GPU ABI test)."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptx_[a-z0-9_]+)\s*\(", text)))


def experimental_symbols(path=HEADER_PATH):
    """Function names declared with PTX_EXPERIMENTAL_API in include/ptx_amd.h: built, tested and exported, but off the
    default path (each measured slower than what the engine runs) -- their ABI is not part of the drop-in contract."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"PTX_EXPERIMENTAL_API\s+[a-z_ \*]+?\b(ptx_[a-z0-9_]+)\s*\(", text)))


# the experimental part of SIGNATURES (kept in step with the header by tests/test_abi_and_host.py)
EXPERIMENTAL = ("ptx_conv_program_num_tiles", "ptx_conv_program_tile_name", "ptx_conv_program_plan", "ptx_conv_program_describe",
                "ptx_conv_program_build", "ptx_conv_program_fwd", "ptx_conv_program_trace_fwd", "ptx_conv_program_error",
                "ptx_nonlocal_workspace_bytes", "ptx_nonlocal_ws_fwd")

_lib = None


def lib():
    """Load (once) and return the shared library with typed entry points."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtxError(
                "libptx_amd.so not found at %s -- build it with "
                "`python pretorched-x_amd/csrc/build.py` (or __graft_entry__.build()); "
                "this package has no CPU / eager fallback" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is missing: loud by design
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def source_hash():
    """sha256 of the library's sources as they are in THIS tree (csrc/build.py:source_hash -- the same function build.py
    compiles into ptx_version()).  Equal to `binary_source_hash()` iff the loaded .so was built from this tree."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ptx_build", os.path.join(_HERE, "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash()


def binary_source_hash():
    """The source hash the loaded libptx_amd.so carries (the part of ptx_version() after "src:")."""
    return lib().ptx_version().decode().rsplit("src:", 1)[-1]


def check(status, what=""):
    if status != 0:
        msg = lib().ptx_last_error().decode(errors="replace")
        err = PtxError("%s failed (status %d): %s" % (what or "libptx_amd call", status, msg))
        err.status = int(status)
        raise err
