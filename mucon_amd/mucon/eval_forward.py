"""The evaluation forward of one video (MuCon.forward_deferred: reference models.py:319-358 in eval mode, greedy decode with the
EOS stop) as four calls into the C ABI with the per-call Python reduced to pointer arithmetic.

The generic path (ops.encoder_forward / lstm_forward / decoder_forward_deferred / head_forward) re-derives, per call, what does not
change between test videos: 50 parameter pointers into four ctypes structs, contiguity / dtype checks of each, four workspace
size queries, a dozen device allocations.  At batch 1 that Python is ~0.18 ms per video next to ~0.1 ms of launch calls, and the
evaluation is host-bound (tools/eval_host_profile.py).  Here

  * the parameter structs are built once and reused while every parameter still lives at the address it was built for (one
    data_ptr() per parameter and call: .to(), load_state_dict() into new storage or a replaced Parameter all change it);
  * configs, workspace sizes and the offsets of everything inside ONE byte slab are cached per tape length;
  * one device allocation per video holds the four workspaces and every output; the outputs are views of it.

Same launches, same kernels, same results as the generic path (tests/test_gpu_eval_batched.py compares the two bit for bit)."""
import ctypes
import weakref
from typing import Dict, List

import torch

from .. import _lib, ops

_ALIGN = 256


def _up(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


class _Plan:
    """Everything that depends on the tape length only."""
    __slots__ = ("enc_cfg", "dec_cfg", "Tz", "nb_enc", "nb_lstm", "nb_dec", "nb_head", "off", "total", "f32_words")


_INSTANCES = weakref.WeakKeyDictionary()      # model -> EvalForward (kept out of the module's __dict__: it holds ctypes objects)


def for_model(model) -> "EvalForward":
    ef = _INSTANCES.get(model)
    if ef is None:
        ef = _INSTANCES[model] = EvalForward(model)
    return ef


class EvalForward:
    def __init__(self, model):
        self._model = weakref.ref(model)
        self.lib = _lib.load()
        self._ptrs: List[int] = []
        self._plans: Dict[int, _Plan] = {}
        self._structs = None
        self.trusted = False      # (r6) set by MuConEvaluator around a chunk of videos: the parameters were validated by the chunk's first call and no code between
                                  # the calls can replace them -- the ~100 (data_ptr, dtype, numel) reads per video (~40 us of a host-bound ~300) are skipped

    # ------------------------------------------------------------------ parameters
    def _param_list(self):
        m = self._model()
        return (m.ft.ordered_parameters() + [m.ft_last_gn.weight, m.ft_last_gn.bias], list(m.fs_encoder_lstm._flat_weights),
                m._decoder_param_list(), [m.conv_classifier.weight, m.conv_classifier.bias])

    def _bind(self, groups) -> None:
        m = self._model()
        enc, lstm, dec, head = groups
        for t in enc + lstm + dec + head:
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.MuconHipError("mucon_amd ops need contiguous float32 device tensors: there is no CPU fallback")
        if len(dec) != len(_lib.DECODER_PARAM_FIELDS):
            raise ValueError(f"expected {len(_lib.DECODER_PARAM_FIELDS)} decoder parameter tensors, got {len(dec)}")
        spec = m._encoder_spec()
        ndir = 2 if m.fs_encoder_lstm.bidirectional else 1
        if len(lstm) != 4 * ndir:
            raise ValueError(f"expected {4 * ndir} LSTM weight tensors, got {len(lstm)}")
        self._structs = {"spec": spec, "enc": ops._pack_params(spec, enc), "lstm": ops._lstm_params(lstm, ndir), "dec": ops._decoder_params(dec),
                         "ndir": ndir, "H": int(lstm[1].shape[1]), "I": int(lstm[0].shape[1]),
                         "D": int(dec[7].shape[0]), "NC": int(dec[17].shape[0]), "n_emb": int(dec[8].shape[0]),
                         "head_w": head[0].data_ptr(), "head_b": head[1].data_ptr(), "C": int(head[0].shape[0]), "Hh": int(head[0].shape[1])}
        self._plans.clear()

    def _plan(self, T: int) -> _Plan:
        s, lib, m = self._structs, self.lib, self._model()
        p = _Plan()
        p.enc_cfg = s["spec"].to_c(1, T, False, 0)
        p.nb_enc = lib.mucon_encoder_workspace_bytes(ctypes.byref(p.enc_cfg))
        if p.nb_enc == 0:
            _lib.check(_lib.E_ARG, "mucon_encoder_workspace_bytes")
        p.Tz = lib.mucon_encoder_out_length(ctypes.byref(p.enc_cfg))
        ME, S = s["ndir"] * s["H"], int(m.max_decoding_steps)
        p.dec_cfg = _lib.DecoderCfg(Tz=p.Tz, ME=ME, D=s["D"], NC=s["NC"], n_emb=s["n_emb"], max_steps=S, teacher_forcing=0,
                                    stop_on_eos=1, eos=int(m.EOS_token_id))
        p.nb_lstm = lib.mucon_lstm_workspace_bytes(p.Tz, s["ndir"])
        p.nb_dec = lib.mucon_decoder_workspace_bytes(ctypes.byref(p.dec_cfg))
        if p.nb_dec == 0:
            _lib.check(_lib.E_ARG, "mucon_decoder_workspace_bytes")
        p.nb_head = lib.mucon_head_workspace_bytes(1, p.Tz, s["Hh"], s["C"])
        # the slab: float32 results first (one typed view serves all of them), then the workspaces
        off, at = {}, 0
        for name, words in (("logp", T * s["C"]), ("logits", T * s["C"]), ("tlogp", S * s["NC"]), ("lens", S), ("n_steps", 1),
                            ("enc", p.Tz * s["spec"].hidden), ("mem", p.Tz * ME), ("hn", ME), ("cn", ME)):
            off[name] = at
            at = _up(at + 4 * words)
        p.f32_words = at // 4
        for name, nb in (("ws_enc", p.nb_enc), ("ws_lstm", p.nb_lstm), ("ws_dec", p.nb_dec), ("ws_head", p.nb_head)):
            off[name] = at
            at = _up(at + max(nb, 1))
        p.off, p.total = off, at
        return p

    # ------------------------------------------------------------------ one video
    @torch.no_grad()
    def __call__(self, feats: torch.Tensor, tf_input: torch.Tensor) -> dict:
        """feats [1 x T x D] float32 on the device, tf_input int64 on the device (its first entry is the start token) ->
        the dict of MuCon.forward_deferred."""
        if not (self.trusted and self._structs is not None):
            groups = self._param_list()
            ptrs = [(t.data_ptr(), t.dtype, t.numel()) for g in groups for t in g]
            if ptrs != self._ptrs:
                self._bind(groups)
                self._ptrs = ptrs
        s, lib = self._structs, self.lib
        if not feats.is_cuda or feats.dtype != torch.float32 or not tf_input.is_cuda:
            raise _lib.MuconHipError("mucon_amd ops need device tensors: there is no CPU fallback")
        feats = feats.contiguous()
        tf_input = tf_input.contiguous().to(torch.int64)
        if tf_input.numel() < 1:
            raise ValueError("forward_deferred: tf_input needs the start token")
        _, T, D = feats.shape
        if D != s["spec"].in_dim:
            raise ValueError(f"tape feature dim {D} != {s['spec'].in_dim}")
        p = self._plans.get(T)
        if p is None:
            p = self._plans[T] = self._plan(T)
        slab = torch.empty(p.total, dtype=torch.uint8, device=feats.device)
        base, off = slab.data_ptr(), p.off
        stream = _lib.current_stream_raw()
        chk = _lib.check
        chk(lib.mucon_encoder_fwd(ctypes.byref(p.enc_cfg), ctypes.byref(s["enc"]), feats.data_ptr(), base + off["enc"], base + off["ws_enc"],
                                  p.nb_enc, stream), "mucon_encoder_fwd")
        chk(lib.mucon_lstm_fwd(p.Tz, s["I"], s["H"], s["ndir"], base + off["enc"], ctypes.byref(s["lstm"]), base + off["mem"], base + off["hn"],
                               base + off["cn"], base + off["ws_lstm"], p.nb_lstm, stream), "mucon_lstm_fwd")
        chk(lib.mucon_decoder_fwd(ctypes.byref(p.dec_cfg), ctypes.byref(s["dec"]), base + off["mem"], base + off["hn"], base + off["cn"],
                                  tf_input.data_ptr(), None, base + off["tlogp"], base + off["lens"], base + off["n_steps"],
                                  base + off["ws_dec"], p.nb_dec, stream), "mucon_decoder_fwd")
        chk(lib.mucon_head_fwd(1, p.Tz, T, s["Hh"], s["C"], base + off["enc"], s["head_w"], s["head_b"], base + off["logits"],
                               base + off["logp"], base + off["ws_head"], p.nb_head, stream), "mucon_head_fwd")
        f = slab[: 4 * p.f32_words].view(torch.float32)
        C, S, NC = s["C"], p.dec_cfg.max_steps, s["NC"]
        o = off["logits"] // 4
        t0, l0, n0 = off["tlogp"] // 4, off["lens"] // 4, off["n_steps"]
        return {"logp": f[: T * C].view(T, C), "segmentation": f[o: o + T * C].view(T, C), "transcript": f[t0: t0 + S * NC].view(S, NC),
                "lengths": f[l0: l0 + S], "n_steps": slab[n0: n0 + 4].view(torch.int32)}
