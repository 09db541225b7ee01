"""ctypes binding of libmucon_hip.so -- the C ABI declared in include/mucon_hip.h.

The product path has NO fallback: if the library cannot be loaded (or built with hipcc), every
entry point raises.  Nothing under oracle/ is ever imported from here.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmucon_hip.so")
MAX_LAYERS = 16
ABI_VERSION = 8
METRICS_MAX_RUNS = 1024   # MUCON_METRICS_MAX_RUNS

OK, E_ARG, E_WORKSPACE, E_HIP = 0, -1, -2, -3
VIT_OK, VIT_INDEX_ERROR, VIT_NO_HYPOTHESIS, VIT_TRUNCATED = 0, 1, 2, 3
VIT_BEAM_MAX_ITEMS = 4096           # MUCON_VIT_BEAM_MAX_ITEMS: max_hypotheses + N of mucon_viterbi_decode_beam
VIT_LABELS_I32, VIT_LABELS_U8, VIT_LABELS_NONE = 0, 1, 2   # MUCON_VIT_LABELS_*
VIT_LATENCY_VIDEOS = 8                                    # MUCON_VIT_LATENCY_VIDEOS

_f32p = ctypes.POINTER(ctypes.c_float)


class EncoderCfg(ctypes.Structure):
    _fields_ = [
        ("B", ctypes.c_int32), ("T", ctypes.c_int32), ("D", ctypes.c_int32), ("H", ctypes.c_int32),
        ("n_layers", ctypes.c_int32),
        ("dilation", ctypes.c_int32 * MAX_LAYERS),
        ("pool_after", ctypes.c_int32 * MAX_LAYERS),
        ("pool_type", ctypes.c_int32),
        ("leaky", ctypes.c_int32),
        ("last_gn", ctypes.c_int32), ("gn_groups", ctypes.c_int32), ("gn_eps", ctypes.c_float),
        ("last_relu", ctypes.c_int32),
        ("training", ctypes.c_int32),
        ("p_drop_layer", ctypes.c_float), ("p_drop_last", ctypes.c_float),
        ("seed", ctypes.c_uint64),
    ]


class EncoderParams(ctypes.Structure):
    _fields_ = [
        ("first_w", ctypes.c_void_p), ("first_b", ctypes.c_void_p),
        ("dil_w", ctypes.c_void_p * MAX_LAYERS), ("dil_b", ctypes.c_void_p * MAX_LAYERS),
        ("pw_w", ctypes.c_void_p * MAX_LAYERS), ("pw_b", ctypes.c_void_p * MAX_LAYERS),
        ("last_w", ctypes.c_void_p), ("last_b", ctypes.c_void_p),
        ("gn_w", ctypes.c_void_p), ("gn_b", ctypes.c_void_p),
    ]


class ViterbiJob(ctypes.Structure):
    _fields_ = [
        ("lp", ctypes.c_void_p), ("tr_off", ctypes.c_int64), ("p_off", ctypes.c_int64),
        ("label_off", ctypes.c_int64), ("seg_off", ctypes.c_int64), ("ws_off", ctypes.c_int64),
        ("T", ctypes.c_int32), ("N", ctypes.c_int32), ("force_n", ctypes.c_int32), ("force_j", ctypes.c_int32),
    ]


class ViterbiVideo(ctypes.Structure):
    _fields_ = [("lp", ctypes.c_void_p), ("transcript", ctypes.c_void_p), ("table", ctypes.c_void_p),
                ("T", ctypes.c_int32), ("N", ctypes.c_int32), ("force_n", ctypes.c_int32), ("force_j", ctypes.c_int32)]


class LstmParams(ctypes.Structure):
    _fields_ = [("w_ih", ctypes.c_void_p * 2), ("w_hh", ctypes.c_void_p * 2),
                ("b_ih", ctypes.c_void_p * 2), ("b_hh", ctypes.c_void_p * 2)]


class DecoderCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("Tz", "ME", "D", "NC", "n_emb", "max_steps", "teacher_forcing",
                                              "stop_on_eos", "eos")]


DECODER_PARAM_FIELDS = (
    "hidden_out_w", "hidden_out_b", "cn_out_w", "cn_out_b", "attention_W1", "attention_l2_w", "attention_l2_b",
    "attention_V", "embedding", "attn_combine_w", "attn_combine_b", "lstm_w_ih", "lstm_w_hh", "lstm_b_ih", "lstm_b_hh",
    "transcript0_w", "transcript0_b", "transcript2_w", "transcript2_b", "length0_w", "length0_b", "length2_w", "length2_b")


class DecoderParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in DECODER_PARAM_FIELDS]


class LossCfg(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ("T", "M", "N", "S", "NC", "mucon_type", "smoothing_clamp", "transcript_average")]
                + [(n, ctypes.c_float) for n in ("overlap", "clamp_min", "clamp_max", "length_width", "mul_transcript",
                                                 "mul_length", "mul_mucon", "mul_smoothing")]
                + [("align_corners", ctypes.c_int32)])


class SgdTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("momentum_buf", ctypes.c_void_p),
                ("n", ctypes.c_int64), ("group", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class AdamTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("max_exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_int64), ("group", ctypes.c_int32), ("reserved", ctypes.c_int32)]


# every symbol include/mucon_hip.h declares: (restype, argtypes)
_vp, _i32, _i64, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
SYMBOLS = {
    "mucon_abi_version": (ctypes.c_int, []),
    "mucon_encoder_bwd_overlap": (ctypes.c_int, [_vp, _i32]),
    "mucon_last_error": (ctypes.c_char_p, []),
    "mucon_encoder_out_length": (_i32, [ctypes.POINTER(EncoderCfg)]),
    "mucon_encoder_workspace_bytes": (_sz, [ctypes.POINTER(EncoderCfg)]),
    "mucon_encoder_fwd": (ctypes.c_int, [ctypes.POINTER(EncoderCfg), ctypes.POINTER(EncoderParams), _vp, _vp, _vp, _sz, _vp]),
    "mucon_encoder_bwd": (ctypes.c_int, [ctypes.POINTER(EncoderCfg), ctypes.POINTER(EncoderParams), _vp, _vp, _vp, _sz,
                                         ctypes.POINTER(EncoderParams), _vp]),
    "mucon_linear_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "mucon_linear_fwd": (ctypes.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mucon_linear_bwd": (ctypes.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mucon_metrics_overlap": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mucon_metrics_segmental": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mucon_conv128_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "mucon_conv128_fwd": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "mucon_mstcn_fuse_fwd": (ctypes.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_uint64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mucon_mstcn_tail_bwd": (ctypes.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp]),
    "mucon_conv128_dgrad": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "mucon_conv128_wgrad": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mucon_head_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "mucon_head_fwd": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mucon_head_bwd": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mucon_decoder_bwd_defer": (ctypes.c_int, [_i32]),
    "mucon_decoder_bwd_flush": (ctypes.c_int, []),
    "mucon_head_fwd_defer": (ctypes.c_int, [_i32]),
    "mucon_head_fwd_flush": (ctypes.c_int, []),
    "mucon_head_bwd_defer": (ctypes.c_int, [_i32]),
    "mucon_head_bwd_flush": (ctypes.c_int, []),
    "mucon_viterbi_job_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "mucon_viterbi_decode_batch": (ctypes.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mucon_viterbi_decode_host": (ctypes.c_int, [_i32, ctypes.POINTER(ViterbiVideo), _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "mucon_viterbi_decode_host_poisson": (ctypes.c_int, [_i32, ctypes.POINTER(ViterbiVideo), _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "mucon_viterbi_decode_batch_poisson": (ctypes.c_int, [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mucon_viterbi_decode_beam": (ctypes.c_int, [_i32, ctypes.POINTER(ViterbiVideo), _i32, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "mucon_encoder_saved_view": (ctypes.c_int, [ctypes.POINTER(EncoderCfg), _i32, _i32, ctypes.POINTER(ctypes.c_size_t),
                                                ctypes.POINTER(ctypes.c_int32)]),
    "mucon_test_gemm_nt": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mucon_test_gemm_tn": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    "mucon_test_vit_rows": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "mucon_test_dropout_mask": (ctypes.c_int, [_vp, _i64, ctypes.c_uint64, _i32, ctypes.c_float, _vp]),
    "mucon_test_set_knob": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "mucon_test_get_knob": (ctypes.c_int, [ctypes.c_char_p]),
    "mucon_test_mfma_probe": (ctypes.c_int, [_i32, _i32, _i32, _vp, _sz, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                             ctypes.POINTER(ctypes.c_float), _vp]),
    "mucon_test_read_cs_stamps": (ctypes.c_int, [ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_int32), _i32]),
    "mucon_test_read_clock": (ctypes.c_int, [_i32, ctypes.POINTER(ctypes.c_longlong), _i32]),
    "mucon_test_vit_host_phases": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double)]),
    "mucon_test_read_stamps": (ctypes.c_int, [ctypes.POINTER(ctypes.c_longlong), _i32]),
    "mucon_profile_begin": (ctypes.c_int, [_i32]),
    "mucon_profile_stride": (ctypes.c_int, [_i32]),
    "mucon_profile_end": (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)]),
    "mucon_test_first_conv_split": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _i32,
                                                  ctypes.POINTER(ctypes.c_float), _vp]),
    "mucon_bench_first_conv": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, ctypes.POINTER(ctypes.c_float), _vp]),
    "mucon_lstm_workspace_bytes": (_sz, [_i32, _i32]),
    "mucon_lstm_fwd": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, ctypes.POINTER(LstmParams), _vp, _vp, _vp, _vp, _sz, _vp]),
    "mucon_lstm_bwd": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, ctypes.POINTER(LstmParams), _vp, _vp, _vp, _vp, _vp, _vp,
                                      ctypes.POINTER(LstmParams), _vp, _sz, _vp]),
    "mucon_loss_workspace_bytes": (_sz, [ctypes.POINTER(LossCfg)]),
    "mucon_loss_fwd_bwd": (ctypes.c_int, [ctypes.POINTER(LossCfg)] + [_vp] * 15 + [_sz, _vp]),
    "mucon_sgd_workspace_bytes": (_sz, [_i32, _i64]),
    "mucon_sgd_clip_step": (ctypes.c_int, [_i32, ctypes.POINTER(SgdTensor), _i32, ctypes.POINTER(ctypes.c_float), ctypes.c_float,
                                           ctypes.c_float, ctypes.c_float, _vp, _vp, _sz, _vp]),
    "mucon_clip_grads": (ctypes.c_int, [_i32, ctypes.POINTER(SgdTensor), _i32, ctypes.POINTER(ctypes.c_float), _vp, _vp, _sz, _vp]),
    "mucon_adam_workspace_bytes": (_sz, [_i32, _i64]),
    "mucon_adam_clip_step": (ctypes.c_int, [_i32, ctypes.POINTER(AdamTensor), _i32, ctypes.POINTER(ctypes.c_float), ctypes.c_double,
                                            ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _i64, _vp, _vp, _sz, _vp]),
    "mucon_decoder_workspace_bytes": (_sz, [ctypes.POINTER(DecoderCfg)]),
    "mucon_decoder_fwd": (ctypes.c_int, [ctypes.POINTER(DecoderCfg), ctypes.POINTER(DecoderParams), _vp, _vp, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _vp, _sz, _vp]),
    "mucon_decoder_bwd": (ctypes.c_int, [ctypes.POINTER(DecoderCfg), _i32, ctypes.POINTER(DecoderParams), _vp, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(DecoderParams), _vp, _sz, _vp]),
}

_lib = None
_pyhost = None
PYHOST_PATH = os.path.join(HERE, "libmucon_pyhost.so")


class MuconHipError(RuntimeError):
    pass


def pyhost():
    """libmucon_pyhost.so (csrc/pyhost.c): the per-video part of the binding's list handling as a C loop.  ctypes.PyDLL: the calls
    keep the GIL and Python exceptions raised inside propagate."""
    global _pyhost
    if _pyhost is None:
        if not os.path.exists(PYHOST_PATH):
            from . import build as _build

            _build.build_pyhost()
        lib = ctypes.PyDLL(PYHOST_PATH)
        po = ctypes.py_object
        lib.mucon_py_viterbi_decode.restype = po
        lib.mucon_py_viterbi_decode.argtypes = [po, po, po, po, po, po] + [ctypes.c_long] * 6 + [ctypes.c_ulonglong] * 3
        _pyhost = lib
    return _pyhost


def load(build_if_missing: bool = True):
    """Load libmucon_hip.so (building it with hipcc when absent).  Raises if that is impossible.

    torch must be imported first so that the HIP runtime torch ships (libamdhip64.so.7) is the
    one the library binds to -- one runtime per process."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's libamdhip64 before ours resolves its DT_NEEDED)

    path = LIB_PATH
    if os.environ.get("MUCON_LIB_VARIANT") == "stamp":
        # the timing build with the in-kernel stamps compiled in (mucon_amd/build.py: build_stamp): tools/kernel_cycles.py only -- never the product path
        path = os.path.join(os.path.dirname(LIB_PATH), "libmucon_hip_stamp.so")
        if not os.path.exists(path):
            raise MuconHipError(f"{path} is missing: run `python -m mucon_amd.build --stamp`")
    elif not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise MuconHipError(f"{LIB_PATH} is missing: run `python -m mucon_amd.build`")
        from . import build as _build

        _build.build()
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.mucon_abi_version() != ABI_VERSION:
        raise MuconHipError(f"ABI version mismatch: library {lib.mucon_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


_MFMA16_DEFAULT = None


def mfma16_default() -> int:
    """The MUCON_MFMA16 value this process started with (environment, else the library's default): what tests restore."""
    global _MFMA16_DEFAULT
    if _MFMA16_DEFAULT is None:
        _MFMA16_DEFAULT = int(load().mucon_test_get_knob(b"MUCON_MFMA16"))   # (the library read the environment when it was first used)
    return _MFMA16_DEFAULT


def set_knob(name: str, value) -> None:
    """Tuning / regression knob by its environment name (include/mucon_hip_test.h); tests switch code paths with it."""
    check(load().mucon_test_set_knob(name.encode(), str(value).encode()), f"set_knob({name})")


def check(rc: int, what: str):
    if rc != OK:
        msg = load().mucon_last_error()
        raise MuconHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def current_stream_raw() -> int:
    """The current HIP stream of the current device as an integer handle (torch._C._cuda_getCurrentRawStream: a third of the cost of
    torch.cuda.current_stream().cuda_stream, which builds a Stream object first)."""
    import torch

    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def current_stream_ptr():
    return ctypes.c_void_p(current_stream_raw())


def ptr(t):
    """Device pointer of a contiguous tensor (or NULL for None)."""
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_contiguous(), "tensor handed to the C ABI must be contiguous"
    return ctypes.c_void_p(t.data_ptr())
