"""ctypes binding of libtad_hip.so (the C ABI declared in include/tad_hip.h).

The library is the only compute path of this package: if it is missing the loader
raises, there is no fallback.  Build it with ``python __graft_entry__.py`` (or
``make -C timeautodiff_amd/csrc``)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAD_HIP_LIB") or os.path.join(_HERE, "libtad_hip.so")   # override: A/B builds of the same ABI

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SILU, ACT_DSILU, ACT_SILU_Z32 = 0, 1, 2, 3, 4
CELL_LSTM, CELL_GRU = 0, 1

_p = C.c_void_p
_i64 = C.c_int64
_i = C.c_int
_f = C.c_float


class RnnDir(C.Structure):
    _fields_ = [("xproj", _p), ("ld_xproj", _i64), ("w_packed", _p), ("b_hh", _p), ("out", _p), ("ld_out", _i64),
                ("gates", _p), ("ld_gates", _i64), ("cstate", _p), ("ld_c", _i64), ("reverse", C.c_int32), ("pad_", C.c_int32),
                ("out16", _p), ("ld_out16", _i64)]


class RnnXin(C.Structure):
    _fields_ = [("x16", _p), ("ldx", _i64), ("w_ih16", _p), ("ldw", _i64), ("bias", _p), ("kin", C.c_int32), ("pad_", C.c_int32)]


class RnnBwdDir(C.Structure):
    _fields_ = [("dout", _p), ("ld_dout", _i64), ("gates", _p), ("ld_gates", _i64), ("cstate", _p), ("ld_c", _i64),
                ("out", _p), ("ld_out", _i64), ("w_packed_t", _p), ("dgates", _p), ("ld_dgates", _i64),
                ("reverse", C.c_int32), ("pad_", C.c_int32)]


class TnJob(C.Structure):
    _fields_ = [("B16", _p), ("ldb", _i64), ("nb", _i64), ("shift", _i64), ("C", _p), ("ldc", _i64), ("db", _p), ("db2", _p),
                ("a0", _i64), ("na", _i64), ("gate_map", C.c_int32 * 6), ("A16", _p), ("lda", _i64)]


class RnnPackJob(C.Structure):
    _fields_ = [("w_hh", _p), ("packed_fwd", _p), ("packed_bwd", _p)]


class PackW16Job(C.Structure):
    _fields_ = [("src", _p), ("ld_src", _i64), ("n", _i64), ("k", _i64), ("dst", _p), ("ld_dst", _i64),
                ("transpose", C.c_int32), ("pad_", C.c_int32)]


# name -> (restype, argtypes); must list every function declared in include/tad_hip.h
PROTOTYPES = {
    "tad_abi_version": (_i, []),
    "tad_last_error": (C.c_char_p, []),
    "tad_last_kernel": (C.c_char_p, []),
    "tad_linear_fwd": (_i, [_i, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64, _i, _p]),
    "tad_linear_bwd_data": (_i, [_i, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i, _p]),
    "tad_linear_bwd_weight": (_i, [_i, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _i64, C.POINTER(C.c_int32), _p]),
    "tad_linear_fwd_b16": (_i, [_i64, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _i, _i64, _i, _p]),
    "tad_linear_fwd_b16x": (_i, [_i64, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _i, _i64, _i, _p]),
    "tad_linear_bwd_data_b16": (_i, [_i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i, _p]),
    "tad_linear_bwd_data_b16o": (_i, [_i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p]),
    "tad_linear_bwd_weight_b16": (_i, [_i64, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _i64, C.POINTER(C.c_int32), _p]),
    "tad_mm16_nt": (_i, [_i64, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _i, _i64, _p, _i, _i64, _p, _i, _i64, _p, _i64, _i, _p]),
    "tad_mm16_tn": (_i, [_i64, _i64, _p, _i64, _i, C.POINTER(TnJob), _p]),
    "tad_cvt_bf16": (_i, [_i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p]),
    "tad_pack_w16": (_i, [_i, C.POINTER(PackW16Job), _p]),
    "tad_colsum": (_i, [_i64, _i64, _p, _i64, _p, _i64, _f, _p, _p, _p, _p]),
    "tad_enc_grad_b16": (_i, [_i64, _i, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i, _i, _i, _f, _f, _i, _p, _p, _p, _p, _p, _p]),
    "tad_rnn_fwd": (_i, [_i, _i, _i, _i64, _i64, _i64, _i, C.POINTER(RnnDir), _p]),
    "tad_rnn_workspace_bytes": (_i64, [_i, _i, _i, _i64, _i64, _i64, _i]),
    "tad_rnn_fwd_ws": (_i, [_i, _i, _i, _i64, _i64, _i64, _i, C.POINTER(RnnDir), _p, _i64, _p]),
    "tad_rnn_fwd_ws_x": (_i, [_i, _i, _i, _i64, _i64, _i64, _i, C.POINTER(RnnDir), C.POINTER(RnnXin), _p, _i64, _p]),
    "tad_rnn_ws_status": (_i, [_p, _p]),
    "tad_rnn_bwd": (_i, [_i, _i, _i, _i64, _i64, _i64, _i, C.POINTER(RnnBwdDir), _p]),
    "tad_rnn_pack_input": (_i, [_i, _i64, _i64, _i, _p, _p, _p, _p, _i, _i64, _p, _p]),
    "tad_rnn_packed_bytes": (_i64, [_i, _i64, _i]),
    "tad_rnn_pack_weights": (_i, [_i, _i64, _i, _p, _p, _p, _p]),
    "tad_rnn_pack_weights_multi": (_i, [_i, _i64, _i, _i, C.POINTER(RnnPackJob), _p]),
    "tad_add_noise": (_i, [_i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "tad_sampler_update": (_i, [_i64, _p, _p, _p, _f, _f, _f, _i, _p]),
    "tad_sampler_tail": (_i, [_i, _i64, _i, _i, _i, _p, _i64, _p, _p, _f, _p, _p, _p, _p, _f, _f, _f, _i, _p, _p, _p, _i64, _p]),
    "tad_sampler_tail_b16": (_i, [_i, _i64, _i, _i, _i, _p, _i64, _p, _p, _f, _p, _p, _p, _p, _f, _f, _f, _i, _p, _p, _p, _i64, _p]),
    "tad_sampler_tail_proj_b16": (_i, [_i, _i64, _i, _i, _i, _p, _i64, _p, _p, _f, _p, _p, _p, _p, _f, _f, _f, _i, _p, _p, _p, _p, _i64, _p,
                                       _i64, _p]),
    "tad_sampler_tail_proj_cn_b16": (_i, [_i, _i64, _i, _i, _i, _p, _i64, _p, _p, _f, _p, _p, _p, C.c_uint64, C.c_uint64, _i64, _f, _f, _f, _i,
                                          _p, _p, _p, _p, _i64, _p, _i64, _p]),
    "tad_mse_loss": (_i, [_i64, _p, _p, _f, _p, _p, _p]),
    "tad_grad_norm_sq": (_i, [_i64, _p, _p, _p]),
    "tad_ln_fc_fwd": (_i, [_i, _i64, _i, _i, _p, _i64, _p, _p, _f, _p, _p, _p, _i64, _p, _p, _p]),
    "tad_ln_fc_bwd": (_i, [_i, _i64, _i, _i, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p]),
    "tad_l1_loss": (_i, [_i64, _p, _p, _f, _p, _p, _p]),
    "tad_sincos_features": (_i, [_i64, _i, _p, _i64, _p, _i64, _p]),
    "tad_sincos_features_b16": (_i, [_i64, _i, _p, _i64, _p, _i64, _p]),
    "tad_embed_fwd": (_i, [_i64, _i, _i, _p, _i64, _p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _p, _i64, _p]),
    "tad_embed_fwd_b16": (_i, [_i64, _i, _i, _p, _i64, _p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _p, _i64, _p]),
    "tad_embed_bwd": (_i, [_i64, _i, _i, _p, _i64, _p, _i64, _p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _p]),
    "tad_posenc_fwd": (_i, [_i64, _i, _p, _i64, _f, _p, _p, _p, _p, _p, _i64, _p]),
    "tad_posenc_fwd_b16": (_i, [_i64, _i, _p, _i64, _f, _p, _p, _p, _p, _p, _i64, _p]),
    "tad_silu_fwd": (_i, [_i64, _p, _p, _p]),
    "tad_silu_bwd": (_i, [_i64, _p, _p, _p, _p]),
    "tad_silu_fwd_b16": (_i, [_i64, _p, _p, _p]),
    "tad_silu_bwd_b16": (_i, [_i64, _p, _p, _p, _p]),
    "tad_layernorm_fwd": (_i, [_i64, _i, _p, _i64, _p, _p, _f, _p, _i64, _p, _p]),
    "tad_layernorm_bwd": (_i, [_i64, _i, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p, _p, _p]),
    "tad_gather_rows": (_i, [_i64, _i64, _p, _p, _p, _p]),
    "tad_draw_batch": (_i, [C.c_uint64, C.c_uint64, _i64, _i64, _i64, _i64, _i, _p, _p, _p, _p, _p]),
    "tad_sampler_noise": (_i, [C.c_uint64, C.c_uint64, _i64, _i64, _i, _p, _p]),
    "tad_zero_2d": (_i, [_i64, _i64, _p, _i64, _p]),
    "tad_copy_2d": (_i, [_i64, _i64, _p, _i64, _p, _i64, _p]),
    "tad_add_vec": (_i, [_i64, _p, _p, _p, _p]),
    "tad_gather_windows": (_i, [_i64, _i64, _i64, _i64, _p, _i64, _p, _p, _p]),
    "tad_adam_step": (_i, [_i64, _p, _p, _p, _p, _f, _f, _f, _f, _f, _i, _p]),
    "tad_adam_coefficients": (None, [_f, _f, _f, _i, C.POINTER(C.c_float)]),
    "tad_adam_step_dev": (_i, [_i64, _p, _p, _p, _p, _f, _f, _f, _f, _p, _p]),
    "tad_adam_tick": (_i, [_f, _f, _f, _p, _p]),
    "tad_fill_f32": (_i, [_i64, _p, _f, _p]),
    "tad_sigmoid_f32": (_i, [_i64, _p, _p]),
    "tad_vae_reparam_fwd": (_i, [_i64, _p, _p, _p, _p, _p, _p]),
    "tad_vae_reparam_bwd": (_i, [_i64, _p, _p, _p, _p, _f, _p, _p, _p]),
    "tad_vae_heads_loss": (_i, [_i64, _i, _i, _i, C.POINTER(C.c_int32), _p, _i64, _p, _p, _i64, _f, _f, _f, _p, _p, _p]),
    "tad_vae_prepare": (_i, [_i64, _p, _p, _p, _p, _p, _p]),
    "tad_metadata_noise": (_i, [_i64, _i64, _i, C.POINTER(C.c_int32), _p, _i64, _p, _f, _p, _p, _i64, _p]),
    "tad_metadata_noise_static": (_i, [_i64, _i64, _i, C.POINTER(C.c_int32), _p, _i64, _p, _f, _p, _i64, _p]),
    "tad_flip_categories": (_i, [_i64, _i64, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _p, _i64, _p, _p, _f, _p, _i64, _p]),
    "tad_heads_to_tensor": (_i, [_i64, _i, _i, _i, C.POINTER(C.c_int32), _p, _i64, _p, _i64, _p]),
}

_lib = None


class TadError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TadError(f"{LIB_PATH} not found: the HIP extension is not built "
                           f"(run `python __graft_entry__.py` at the repository root); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load()
        raise TadError(lib.tad_last_error().decode())
