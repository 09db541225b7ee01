"""MuCon model with the reference's surface (src/mucon/models.py): create_model, MuCon.forward /
predict / loss / set_teacher_forcing, encode_params / decode_params, the MuConLoss /
MuConForwardOut / MuConPredictOut records, and the reference's parameter names, so a reference
state_dict loads unchanged.

What runs where
  * temporal_modeling_forward (reference models.py:746-773), frame_classifier_forward (:567-582)
    and the per-frame log-softmax (:368, :405): the hand-written gfx950 kernels, via mucon_amd.ops.
  * the s-head (biLSTM encoder + attention LSTM decoder, :585-744) and the losses (:376-565):
    PyTorch-ROCm ops.  They are rows "next" of SURVEY.md 8f, kept here so that forward/loss are
    complete; their arithmetic follows the reference statement by statement.
"""
import math
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn import functional as F

from .. import ops
from ..core.datasets import Batch
from ..core.modules.temporal import MSTCNPPFirstStage, NoFt, WaveNetBlock
from .masks import create_masks, project_lengths_softmax


@dataclass(repr=False)
class GeneralLoss:
    main: Tensor


@dataclass(repr=False)
class MuConLoss(GeneralLoss):
    transcript_loss: Tensor
    mucon_loss: Tensor
    length_loss: Tensor
    smoothing_loss: Tensor


@dataclass(repr=False)
class MuConForwardOut:
    transcript: Tensor    # [(N + 1) x (M + 1)] log-probs of the s-head, EOS included
    lengths: Tensor       # [N] un-normalised log length estimates
    segmentation: Tensor  # [Tf x M] y-head logits


@dataclass(repr=False)
class MuConPredictOut:
    transcript: List[int]         # N + 1 labels, EOS included
    lengths: Tensor               # [N] softmaxed relative lengths
    segmentation_logits: Tensor   # [T x M] log-softmaxed y-head output (Viterbi input)


def rand_t(*sz):
    return torch.randn(sz) / math.sqrt(sz[0])


def rand_p(*sz):
    return nn.Parameter(rand_t(*sz), requires_grad=True)


def create_fully_supervised_model(cfg, num_classes: int, max_decoding_steps: int, input_feature_size: int) -> "MuConFullySupervised":
    """reference models.py:49-64"""
    if cfg.model.name != "mucon":
        raise Exception("Invalid model name")
    return MuConFullySupervised(cfg, input_feature_size=input_feature_size, num_classes=num_classes,
                                max_decoding_steps=max_decoding_steps)


def create_mixed_supervision_model(cfg, num_classes: int, max_decoding_steps: int, input_feature_size: int) -> "MuConMixedSupervision":
    """reference models.py:67-82"""
    if cfg.model.name != "mucon":
        raise Exception("Invalid model name")
    return MuConMixedSupervision(cfg, input_feature_size=input_feature_size, num_classes=num_classes,
                                 max_decoding_steps=max_decoding_steps)


def create_model(cfg, num_classes: int, max_decoding_steps: int, input_feature_size: int) -> "MuCon":
    if cfg.model.name == "mucon":
        return MuCon(cfg=cfg, input_feature_size=input_feature_size, num_classes=num_classes,
                     max_decoding_steps=max_decoding_steps)
    raise Exception("Invalid model name")


class EmptyTranscriptError(RuntimeError):
    """The s-head emitted EOS as its first word: there is no length to stack.  The reference fails in torch.stack([])
    (models.py:351) with this RuntimeError text; the subclass lets the evaluator tell it from every other RuntimeError."""


class MuCon(nn.Module):
    def __init__(self, cfg, input_feature_size: int, num_classes: int, max_decoding_steps: int):
        super().__init__()
        self.cfg = cfg
        self.input_feature_size, self.num_classes, self.max_decoding_steps = input_feature_size, num_classes, max_decoding_steps
        self.teacher_forcing = True
        self.EOS_token_id = num_classes
        m = cfg.model
        self.loss_mul_mucon, self.loss_mul_transcript = m.loss.mul_mucon, m.loss.mul_transcript
        self.loss_mul_smoothing, self.loss_mul_length = m.loss.mul_smoothing, m.loss.mul_length

        H = m.ft.hidden_size
        if m.ft.type == "wavenet":
            self.ft = WaveNetBlock(in_channels=input_feature_size, stages=m.ft.stages, out_dims=H, pooling=m.ft.pooling,
                                   pooling_type=m.ft.pooling_type, pooling_layers=m.ft.pooling_layers,
                                   leaky=m.ft.leaky_relu, dropout_rate=m.ft.dropout_rate)
        elif m.ft.type == "mstcnpp":   # reference models.py:172-179
            self.ft = MSTCNPPFirstStage(input_dim=input_feature_size, num_layers=len(m.ft.stages), output_dim=H, num_f_maps=H,
                                        pooling_layers=m.ft.pooling_layers)
        elif m.ft.type == "noft":      # reference models.py:180-184
            self.ft = NoFt(in_chnnels=input_feature_size, out_dims=H)
        else:
            raise Exception(f"Invalid ft type ({m.ft.type})")
        self.ft_last_gn = nn.GroupNorm(num_groups=m.ft.last_gn_num_groups, num_channels=H)
        self.ft_last_dropout = nn.Dropout(p=m.ft.last_dropout_rate)

        E, Dd = m.fs.encoder.hidden_size, m.fs.decoder.hidden_size
        bi = m.fs.encoder.bidirectional
        enc_out = 2 * E if bi else E
        self.fs_encoder_lstm = nn.LSTM(input_size=H, hidden_size=E, batch_first=True, dropout=m.fs.encoder.dropout,
                                       bidirectional=bi)
        self.fs_encoder_hidden_out = nn.Linear(enc_out, E)
        self.fs_encoder_cn_out = nn.Linear(enc_out, E)
        self.fs_decoder_attention_W1 = rand_p(E * 2, Dd)
        self.fs_decoder_attention_l2 = nn.Linear(Dd, Dd)
        self.fs_decoder_attention_l3 = nn.Linear(Dd + Dd, Dd)  # present (unused) in the reference too
        self.fs_decoder_attention_V = rand_p(Dd)
        self.fs_decoder_embedding = nn.Embedding(num_embeddings=num_classes + 2, embedding_dim=Dd)
        self.fs_decoder_embedding_drop = nn.Dropout(p=m.fs.decoder.embedding_dropout)
        self.fs_decoder_attn_combine = nn.Linear(enc_out + Dd, Dd)
        self.fs_decoder_lstm = nn.LSTM(input_size=Dd, hidden_size=Dd, dropout=m.fs.decoder.dropout)
        self.fs_decoder_transcript = nn.Sequential(nn.Linear(Dd, Dd), nn.ReLU(), nn.Linear(Dd, num_classes + 1))
        self.fs_decoder_length = nn.Sequential(nn.Linear(Dd + num_classes + 1, int(Dd / 2)), nn.ReLU(),
                                               nn.Linear(int(Dd / 2), 1))
        self.conv_classifier = nn.Conv1d(H, num_classes, kernel_size=1)

        # two parameter sets, clipped separately by the trainer (reference models.py:284-317)
        def flat(items):
            out = []
            for it in items:
                out.extend(list(it.parameters()) if isinstance(it, nn.Module) else [it])
            return out

        self.encode_params = flat([self.ft, self.ft_last_gn, self.fs_encoder_lstm, self.fs_encoder_hidden_out,
                                   self.fs_encoder_cn_out])
        self.decode_params = flat([self.fs_decoder_attention_W1, self.fs_decoder_attention_l2,
                                   self.fs_decoder_attention_l3, self.fs_decoder_attention_V, self.fs_decoder_embedding,
                                   self.fs_decoder_attn_combine, self.fs_decoder_lstm, self.fs_decoder_transcript,
                                   self.fs_decoder_length, self.conv_classifier])
        self._step = 0  # dropout stream counter for the HIP encoder
        self.native_lstm = True     # s-head biLSTM through the HIP kernels (False: torch's nn.LSTM / MIOpen)
        self.native_decoder = True  # s-head decoding loop as one persistent HIP kernel (False: the torch loop)
        self.native_loss = True     # the four losses + gradients in the fused HIP kernels (False: the torch formulation)
        self.fast_eval_forward = True   # forward_deferred through mucon/eval_forward.py (False: the generic ops, call by call)

    def get_params(self, original_lr):  # fandak.Model.get_params
        return [{"params": self.parameters(), "lr": original_lr}]

    # ------------------------------------------------------------------------------ hot path
    def _encoder_spec(self) -> ops.EncoderSpec:
        ft = self.cfg.model.ft
        return self.ft.spec(last_gn=ft.last_gn, last_gn_num_groups=ft.last_gn_num_groups, last_relu=ft.last_relu,
                            last_dropout=ft.last_dropout, last_dropout_rate=ft.last_dropout_rate)

    def temporal_modeling_forward(self, input: Tensor) -> Tensor:
        """[B x T x D] -> [B x T' x D'] (reference models.py:746-773): one fused HIP pipeline --
        no permute, no transposed copy of the tape."""
        if not isinstance(self.ft, WaveNetBlock):
            # non-default encoders (NoFt, MSTCNPPFirstStage): library ops, then the wrapper of models.py:759-768
            ft = self.cfg.model.ft
            z = (self.ft.forward_time_major(input).permute(0, 2, 1) if isinstance(self.ft, NoFt)
                 else self.ft(input.permute(0, 2, 1)))                      # [B x D' x T']
            if ft.last_gn:
                z = self.ft_last_gn(z)
            if ft.last_relu:
                z = F.relu(z)
            if ft.last_dropout:
                z = self.ft_last_dropout(z)
            return z.permute(0, 2, 1).contiguous()
        self._step += 1
        seed = self._dropout_seed() if self.training else 0
        return self.ft.forward_time_major(input, self.ft_last_gn.weight, self.ft_last_gn.bias, self._encoder_spec(), seed)

    dropout_rank = 0   # data-parallel rank (set by SimpleTrainer): every rank draws its own masks

    def _dropout_seed(self) -> int:
        """64-bit key of this step's dropout masks: (cfg seed, step counter) with the rank folded in by a golden-ratio
        multiple, so ranks do not share masks; the kernels hash the whole word (csrc/common.hpp: make_drop)."""
        seed = ((int(self.cfg.system.seed) << 32) ^ self._step) & 0xFFFFFFFFFFFFFFFF
        return seed ^ ((int(self.dropout_rank) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)

    def frame_classifier_forward(self, temporal_encoded: Tensor, target_length: int) -> Tensor:
        """[1 x Ds x Tz] -> [1 x num_classes x Tf] (reference models.py:567-582)."""
        enc = temporal_encoded.permute(0, 2, 1)
        if not enc.is_cuda:      # cfg.system.device = "cpu": plumbing path (mucon_amd/cpu_plumbing.py)
            from .. import cpu_plumbing
            return cpu_plumbing.head_forward(enc, self.conv_classifier.weight, self.conv_classifier.bias, target_length)[0].permute(0, 2, 1)
        logits, _ = ops.head_forward(enc, self.conv_classifier.weight, self.conv_classifier.bias, target_length,
                                     want_logits=True, want_logp=False)
        return logits.permute(0, 2, 1)

    def _segmentation_and_logp(self, temporal_encoded: Tensor, Tf: int):
        """logits [Tf x M] and log-probs [Tf x M] from one kernel launch."""
        if not temporal_encoded.is_cuda:      # cfg.system.device = "cpu": plumbing path (mucon_amd/cpu_plumbing.py)
            from .. import cpu_plumbing
            logits, logp = cpu_plumbing.head_forward(temporal_encoded, self.conv_classifier.weight, self.conv_classifier.bias, Tf)
            return logits[0], logp[0]
        logits, logp = ops.head_forward(temporal_encoded, self.conv_classifier.weight, self.conv_classifier.bias, Tf)
        return logits[0], logp[0]

    # ------------------------------------------------------------------------------ forward
    def forward(self, batch: Batch) -> MuConForwardOut:
        features = batch.feats                       # [1 x T x D]
        Tf = features.shape[1]
        temporal_encoded = self.temporal_modeling_forward(input=features)      # [1 x Tz x Dt]
        transcripts, lengths = self.sequence_generation_forward(
            temporal_encoded=temporal_encoded, tf_transcript_target_length=batch.transcript_tf_target.shape[0],
            transcript_tf_input=batch.transcript_tf_input, transcript_tf_target=batch.transcript_tf_target)
        segmentation, logp = self._segmentation_and_logp(temporal_encoded, Tf)  # [Tf x M] each
        fused = self.__dict__.pop("_decoder_tensors", None)
        if fused is not None:      # the persistent decoder kernel's outputs, without the list round trip
            if fused[1].shape[0] < 2:
                raise EmptyTranscriptError("stack expects a non-empty TensorList")   # EOS first: what torch.stack([]) raises (models.py:351)
            out = MuConForwardOut(transcript=fused[0], lengths=fused[1][:-1], segmentation=segmentation)
        else:
            if len(lengths) < 2:
                raise EmptyTranscriptError("stack expects a non-empty TensorList")
            out = MuConForwardOut(transcript=torch.cat(transcripts, dim=0), lengths=torch.stack(lengths[:-1]),
                                  segmentation=segmentation)
        out._logp = logp  # the kernel's log-softmax, reused by predict() and the smoothing loss
        return out

    # ------------------------------------------------------------------------------ the training step without autograd
    def can_fuse_step(self, batch: Batch) -> bool:
        lc = self.cfg.model.loss
        return (type(self).loss is MuCon.loss and self.training and isinstance(self.ft, WaveNetBlock) and batch.feats.is_cuda
                and batch.feats.shape[0] == 1 and self.native_lstm and self.native_decoder and self.native_loss
                and self.fs_encoder_lstm.input_size == 128 and self.fs_encoder_lstm.hidden_size == 128
                and self.fs_encoder_lstm.num_layers == 1 and lc.mucon.type in ("flint", "arithmetic")
                and batch.feats.shape[1] >= 2 and self.num_classes <= 64 and 1 <= batch.transcript_tf_target.shape[0] - 1 <= 64
                and batch.feats.shape[1] // 16 <= 4096)

    @torch.no_grad()
    def fused_train_step(self, batch: Batch) -> MuConLoss:
        """forward + loss + backward of one video as a straight line of kernel launches: the same autograd Functions
        (same C entry points, same arithmetic) called directly in dependency order, gradients assigned to `.grad`.
        Equivalent to `loss = self.loss(batch, self.forward(batch)); loss.main.backward()` (tests/test_gpu_fused_step.py);
        what it saves is the autograd graph walk: ~40 nodes with a Python round trip each -- half of the step's host time.
        (Since r3 the step is bound by its ~50 launches on the GPU, 0.82 ms; the y-head's four launches on a side stream beside
        the s-head's persistent kernels were measured: 0.85-0.91 ms -- the stream hand-offs cost more than the ~30 us they hide.)"""
        F_ = ops
        lc = self.cfg.model.loss
        feats = batch.feats
        Tf = feats.shape[1]
        self._step += 1
        seed = self._dropout_seed()
        enc_params = self.ft.ordered_parameters() + [self.ft_last_gn.weight, self.ft_last_gn.bias]
        enc, c_enc = F_.run_forward(F_._EncoderFn, feats, self._encoder_spec(), True, int(seed), *enc_params)
        # the y-head's forward in front of the s-head's (it only needs enc): (r6) its kernel then rides in the LSTM's forward recurrence launch, which follows on this
        # stream (ops.DEFER_NEXT_HEAD_FORWARD -> mucon_head_fwd_defer); logits / logp are written by THAT launch and first read by the losses below
        wc = self.conv_classifier.weight
        F_.DEFER_NEXT_HEAD_FORWARD = bool(getattr(self, "fused_step_deferrals", True))
        (logits, logp), c_head = F_.run_forward(F_._HeadFn, enc, wc.reshape(wc.shape[0], wc.shape[1]), self.conv_classifier.bias,
                                                 int(Tf), True, True)
        F_.DEFER_NEXT_HEAD_FORWARD = False
        lstm = self.fs_encoder_lstm
        lstm_w = list(lstm.parameters())
        ndir = 2 if lstm.bidirectional else 1
        (memory, h_n, c_n), c_lstm = F_.run_forward(F_._LstmFn, enc[0], ndir, *lstm_w)
        steps = batch.transcript_tf_target.shape[0]
        mask = self._embedding_drop_mask(steps, feats.device)
        dec_params = self._decoder_param_list()
        (tlogp, lens), c_dec = F_.run_forward(F_._DecoderFn, memory, h_n.reshape(-1), c_n.reshape(-1), batch.transcript_tf_input,
                                               mask, (int(steps), bool(self.teacher_forcing), False, int(self.EOS_token_id)),
                                               *dec_params)
        seg = logits[0]
        sx = logp[0] if lc.smoothing.log_softmax_before else seg
        fo = MuConForwardOut(transcript=tlogp, lengths=lens[:-1], segmentation=seg)
        if self.teacher_forcing:
            target = batch.transcript
        else:
            target = tlogp[:-1].argmax(dim=1)
            target = torch.where(target >= self.num_classes, torch.zeros_like(target), target)
        self._loss_constants(seg.device)
        (main, parts), c_loss = F_.run_forward(F_._LossFn, seg, sx, tlogp, lens[:-1].contiguous(), self._loss_spec(), target.to(torch.int64),
                                                batch.transcript_tf_target.to(torch.int64), self._loss_tmpl, self._loss_mw, self._loss_tw)
        d_seg, d_sx, d_tlogp, d_len = c_loss.saved_tensors            # d main / d input, computed with the loss itself
        if getattr(self, "fused_step_deferrals", True):      # (an instance attribute set to False: the same step without them -- tests compare the two bit for bit)
            c_head.defer_reduce = True     # (r6) d_wc / d_bc are summed inside the encoder backward's first launch, which follows on this stream below
            c_head.defer_kernel = True     # (r6) ... and the y-head's z-level backward kernel rides in the decoder backward's eight-workgroup launch, which follows on this stream below
            c_dec.defer_outer = True       # (r6) the decoder's weight-gradient outer products ride in the LSTM backward's recurrence launch, which follows on this stream below
            c_head.reuse_grads = c_dec.reuse_grads = c_lstm.reuse_grads = True   # (r6) the previous video's gradient tensors serve again: the optimizer step that consumed them is behind us
        if lc.smoothing.log_softmax_before:
            d_enc, d_wc, d_bc = F_.run_backward(F_._HeadFn, c_head, d_seg.unsqueeze(0), d_sx.unsqueeze(0))[:3]
        else:
            d_enc, d_wc, d_bc = F_.run_backward(F_._HeadFn, c_head, (d_seg + d_sx).unsqueeze(0), None)[:3]
        d_lens = c_loss.d_len_full                                      # [steps]: the loss kernel wrote the unused last entry as 0
        d_mem, d_hn, d_cn, _, _, _, *g_dec = F_.run_backward(F_._DecoderFn, c_dec, d_tlogp, d_lens)
        c_lstm.dx_accumulate = d_enc[0]                                 # the LSTM's input gradient is added onto the y-head's in its kernel
        _, _, *g_lstm = F_.run_backward(F_._LstmFn, c_lstm, d_mem, d_hn.view(ndir, -1), d_cn.view(ndir, -1))
        c_enc.reuse_grads = bool(getattr(self, "fused_step_deferrals", True))   # (r6) the optimizer step of the previous video is behind us: its gradient buffer and views serve again (ops._EncoderFn.backward)
        g_enc = F_.run_backward(F_._EncoderFn, c_enc, d_enc)[4:]
        for prm, g in zip(enc_params, g_enc):
            prm.grad = g
        for prm, g in zip(lstm_w, g_lstm):
            prm.grad = g
        for prm, g in zip(dec_params, g_dec):
            prm.grad = g
        wc.grad, self.conv_classifier.bias.grad = d_wc.view_as(wc), d_bc
        loss = MuConLoss(main=main, transcript_loss=parts[0], length_loss=parts[1], mucon_loss=parts[2], smoothing_loss=parts[3])
        return loss, fo

    def can_defer_eval(self, batch: Batch, on_device: Optional[bool] = None) -> bool:
        """The evaluation forward without host round trips exists for the all-HIP configuration (device tensors, the native LSTM and
        decoder at the reference's sizes, one video per batch); everything else takes forward() / predict().  Depends on the VIDEO
        (its encoded length): callers ask per video.  on_device: the answer for this batch once it has been moved to the GPU."""
        is_cuda = batch.feats.is_cuda if on_device is None else on_device
        key = (self.training, self.teacher_forcing, self.native_lstm, self.native_decoder)
        st = self.__dict__.get("_defer_static")
        if st is None or st[0] != key:
            # what does not depend on the video: the module sizes the kernels are built for, and how often the encoder halves the tape
            # (module sizes are fixed at construction; the four flags above are the things a caller flips)
            ft = self.cfg.model.ft
            halvings = sum(1 for i in range(len(ft.stages)) if ft.pooling and i in ft.pooling_layers)
            lstm, d = self.fs_encoder_lstm, self.fs_decoder_lstm
            ok = (not self.training and not self.teacher_forcing and isinstance(self.ft, WaveNetBlock) and self.native_lstm
                  and self.native_decoder and lstm.input_size == 128 and lstm.hidden_size == 128 and lstm.num_layers == 1
                  and d.input_size == 128 and d.hidden_size == 128 and d.num_layers == 1 and self.num_classes + 1 <= 128
                  and self.fs_decoder_embedding.embedding_dim == 128 and (2 if lstm.bidirectional else 1) * lstm.hidden_size <= 256)
            st = self.__dict__["_defer_static"] = (key, bool(ok), halvings)
        shape = batch.feats.shape
        return st[1] and is_cuda and shape[0] == 1 and 1 <= (shape[1] >> st[2]) <= 4096

    @torch.no_grad()
    def forward_deferred(self, batch: Batch) -> dict:
        """forward() of the evaluation (eval mode, greedy decoding with the EOS stop: reference models.py:319-358 as
        evaluators.py:316-318 drives it) as the same launches, but nothing is read back: the number of decoded words stays on
        the device.  -> {"logp" [T x M] log-softmaxed y-head output, "segmentation" [T x M] logits, "transcript" [S x (M+1)]
        log-probs of all S = max_decoding_steps rows (rows >= n_steps unwritten), "lengths" [S], "n_steps" int32 [1]}."""
        if self.fast_eval_forward and self.ft.out_dims == 128:
            # the same four library calls with the per-call Python trimmed (parameter structs and per-length plans cached, one
            # allocation per video): mucon/eval_forward.py
            from .eval_forward import for_model
            self._step += 1           # as temporal_modeling_forward counts it
            return for_model(self)(batch.feats, batch.transcript_tf_input)
        Tf = batch.feats.shape[1]
        temporal_encoded = self.temporal_modeling_forward(input=batch.feats)
        enc_out, h_n, c_n = self._sequence_encoder(temporal_encoded)
        tlogp, lens, n_steps = ops.decoder_forward_deferred(enc_out[0], h_n, c_n, batch.transcript_tf_input, self._decoder_param_list(),
                                                            self.max_decoding_steps, self.EOS_token_id)
        segmentation, logp = self._segmentation_and_logp(temporal_encoded, Tf)
        return {"logp": logp, "segmentation": segmentation, "transcript": tlogp, "lengths": lens, "n_steps": n_steps}

    def predict(self, batch: Batch, forward_out: MuConForwardOut) -> MuConPredictOut:
        if self.teacher_forcing:
            transcript = batch.transcript_tf_target.detach().cpu().numpy().tolist()
        else:
            transcript = forward_out.transcript.argmax(dim=1).tolist()   # one host sync, not one per word
        logp = getattr(forward_out, "_logp", None)
        if logp is None:
            logp = F.log_softmax(forward_out.segmentation, dim=1)
        # softmax over a handful of length logits, written out: F.softmax goes through MIOpen on ROCm (~1 ms of host time)
        e = torch.exp(forward_out.lengths - forward_out.lengths.max())
        return MuConPredictOut(transcript=transcript, lengths=e / e.sum(), segmentation_logits=logp)

    # ------------------------------------------------------------------------------ s-head
    def sequence_generation_forward(self, temporal_encoded: Tensor, tf_transcript_target_length: int,
                                    transcript_tf_input: Tensor, transcript_tf_target: Tensor):
        """biLSTM encoder over [1 x Tz x D'], additive attention, LSTM decoder (reference models.py:585-728)."""
        enc_out, h_n, c_n = self._sequence_encoder(temporal_encoded)
        steps = tf_transcript_target_length if (self.teacher_forcing or self.training) else self.max_decoding_steps
        if self._native_decoder_ok(enc_out):
            return self._native_decoder(enc_out[0], h_n, c_n, steps, transcript_tf_input)
        dec_h = self.fs_encoder_hidden_out(h_n.view(1, -1)).unsqueeze(0)   # [1 x 1 x D'']
        dec_c = self.fs_encoder_cn_out(c_n.view(1, -1)).unsqueeze(0)
        memory = enc_out[0]                                               # [Tz x 2D']
        memory_proj = memory @ self.fs_decoder_attention_W1               # [Tz x D'']
        lengths, transcripts = [], []
        dec_in = transcript_tf_input[0].unsqueeze(0)
        for step in range(steps):
            if self.teacher_forcing:
                dec_in = transcript_tf_input[step].unsqueeze(0)
            emb = self.fs_decoder_embedding_drop(F.relu(self.fs_decoder_embedding(dec_in)))      # [1 x Ds]
            attn = self._calculate_attention(dec_h, memory_proj)                                  # [Tz]
            context = (attn.unsqueeze(1) * memory).sum(dim=0, keepdim=True)                       # [1 x 2Ds]
            mixed = F.relu(self.fs_decoder_attn_combine(torch.cat((emb, context), 1)).unsqueeze(0))  # [1 x 1 x Ds]
            dec_out, (dec_h, dec_c) = self.fs_decoder_lstm(mixed, (dec_h, dec_c))
            word_logits = self.fs_decoder_transcript(dec_out)                                     # [1 x 1 x M+1]
            length = self.fs_decoder_length(F.relu(torch.cat((mixed, word_logits), 2))).squeeze()
            word_logp = F.log_softmax(word_logits.squeeze(0), dim=1)                              # [1 x M+1]
            transcripts.append(word_logp)
            lengths.append(length)
            word = word_logp.argmax(dim=1)
            if not self.teacher_forcing and not self.training and word.item() == self.EOS_token_id:
                break
            if not self.teacher_forcing:
                dec_in = word
        return transcripts, lengths

    def _decoder_param_list(self):
        """The 23 tensors of _lib.DECODER_PARAM_FIELDS, in that order (cached like WaveNetBlock.ordered_parameters: every entry's
        identity is checked against the _parameters dict it came from, so a replaced Parameter object is noticed)."""
        cache = self.__dict__.get("_decoder_params_cache")
        if cache is not None:
            owners = self.__dict__["_decoder_param_owners"]
            if all(o[k] is t for (o, k), t in zip(owners, cache)):
                return cache
        owners = [(m._parameters, k) for m, k in self._decoder_param_sites()]
        cache = [o[k] for o, k in owners]
        self.__dict__["_decoder_params_cache"] = cache
        self.__dict__["_decoder_param_owners"] = owners
        return cache

    def _decoder_param_sites(self):
        """(module, parameter name) per entry of _lib.DECODER_PARAM_FIELDS."""
        lstm = self.fs_decoder_lstm
        sites = [(self.fs_encoder_hidden_out, "weight"), (self.fs_encoder_hidden_out, "bias"), (self.fs_encoder_cn_out, "weight"),
                 (self.fs_encoder_cn_out, "bias"), (self, "fs_decoder_attention_W1"), (self.fs_decoder_attention_l2, "weight"),
                 (self.fs_decoder_attention_l2, "bias"), (self, "fs_decoder_attention_V"), (self.fs_decoder_embedding, "weight"),
                 (self.fs_decoder_attn_combine, "weight"), (self.fs_decoder_attn_combine, "bias")]
        sites += [(lstm, n) for n, _ in lstm.named_parameters()]
        for m in (self.fs_decoder_transcript[0], self.fs_decoder_transcript[2], self.fs_decoder_length[0], self.fs_decoder_length[2]):
            sites += [(m, "weight"), (m, "bias")]
        return sites

    def _native_decoder_ok(self, enc_out: Tensor) -> bool:
        d = self.fs_decoder_lstm
        return (self.native_decoder and enc_out.is_cuda and d.input_size == 128 and d.hidden_size == 128
                and d.num_layers == 1 and enc_out.shape[2] <= 256 and self.num_classes + 1 <= 128
                and self.fs_decoder_embedding.embedding_dim == 128 and enc_out.shape[1] <= 4096)

    def _embedding_drop_mask(self, steps: int, device):
        """[steps x 128] keep-mask / (1 - p) of the decoder's embedding dropout (None for p = 0), in ONE launch: dropout of a
        cached tensor of ones (rand, compare, cast and divide were four).  Drawn with torch's generator."""
        p = self.fs_decoder_embedding_drop.p
        if p <= 0:
            return None
        ones = getattr(self, "_drop_ones", None)
        if ones is None or ones.shape[0] < steps or ones.device != device:
            ones = self._drop_ones = torch.ones((max(steps, 64), 128), device=device)
        return torch.nn.functional.dropout(ones[:steps], p, training=True)

    def _native_decoder(self, memory: Tensor, h_n: Tensor, c_n: Tensor, steps: int, transcript_tf_input: Tensor):
        """The decoding loop below as one persistent HIP kernel (ops.decoder_forward, csrc/decoder.hpp); returns
        the same per-step lists.  The embedding dropout mask is drawn with torch's generator."""
        mask = self._embedding_drop_mask(steps, memory.device) if self.training else None
        stop = not self.teacher_forcing and not self.training
        logp, lengths = ops.decoder_forward(memory, h_n, c_n, transcript_tf_input, self._decoder_param_list(), steps,
                                            self.teacher_forcing, stop, self.EOS_token_id, mask)
        n = logp.shape[0]
        # forward() takes the two tensors as they are; the per-step lists below only serve callers of the reference's
        # list interface (each slice is an autograd node: ~4 tiny launches apiece in the backward)
        self._decoder_tensors = (logp, lengths)
        return [logp[i:i + 1] for i in range(n)], [lengths[i] for i in range(n)]

    def _sequence_encoder(self, temporal_encoded: Tensor):
        """fs_encoder_lstm over [1 x Tz x D'] (reference models.py:605-611).  On the GPU, at the reference's
        sizes (input = hidden = 128, one layer), this is the persistent HIP LSTM (ops.lstm_forward, csrc/lstm.hpp)
        on the module's own parameters; other sizes -- and the CPU-side surface tests -- run torch's nn.LSTM."""
        lstm = self.fs_encoder_lstm
        if (temporal_encoded.is_cuda and temporal_encoded.shape[0] == 1 and lstm.input_size == 128
                and lstm.hidden_size == 128 and lstm.num_layers == 1 and self.native_lstm):
            out, h_n, c_n = ops.lstm_forward(temporal_encoded[0], list(lstm.parameters()), lstm.bidirectional)
            return out.unsqueeze(0), h_n, c_n
        enc_out, (h_n, c_n) = lstm(temporal_encoded)
        return enc_out, h_n, c_n

    def _calculate_attention(self, current_hidden_state: Tensor, encoder_result_ready_for_attention: Tensor) -> Tensor:
        q = self.fs_decoder_attention_l2(current_hidden_state.view(1, -1))
        u = torch.tanh(encoder_result_ready_for_attention + q)
        return F.softmax(u @ self.fs_decoder_attention_V, dim=0)

    # ------------------------------------------------------------------------------ losses
    def loss(self, batch: Batch, forward_out: MuConForwardOut) -> MuConLoss:
        if self._native_loss_ok(forward_out):
            return self._native_loss(batch, forward_out)
        t = self.transcript_loss(batch, forward_out)
        ln = self.length_loss(batch, forward_out)
        mu = self.mucon_loss(batch, forward_out)
        sm = self.smoothing_loss(batch, forward_out)
        main = self.loss_mul_transcript * t + self.loss_mul_length * ln + self.loss_mul_mucon * mu + self.loss_mul_smoothing * sm
        return MuConLoss(main=main, transcript_loss=t, length_loss=ln, mucon_loss=mu, smoothing_loss=sm)

    def _native_loss_ok(self, forward_out: MuConForwardOut) -> bool:
        seg = forward_out.segmentation
        return (self.native_loss and seg.is_cuda and seg.shape[0] >= 2 and seg.shape[1] <= 64
                and 1 <= forward_out.lengths.shape[0] <= 64 and self.cfg.model.loss.mucon.type in ("flint", "arithmetic"))

    def _native_loss(self, batch: Batch, forward_out: MuConForwardOut) -> MuConLoss:
        """All four losses and their gradients through the fused HIP kernels (ops.losses_forward, csrc/loss.hpp); the
        torch formulation below stays as the path for CPU tensors and unsupported sizes."""
        lc = self.cfg.model.loss
        seg, dev = forward_out.segmentation, forward_out.segmentation.device
        if self.teacher_forcing:
            target = batch.transcript
        else:   # reference models.py:417-428
            target = forward_out.transcript[:-1].argmax(dim=1)
            target = torch.where(target >= self.num_classes, torch.zeros_like(target), target)
        if lc.smoothing.log_softmax_before:
            logp = getattr(forward_out, "_logp", None)
            sx = logp if logp is not None else F.log_softmax(seg, dim=1)
        else:
            sx = seg
        self._loss_constants(dev)
        spec = self._loss_spec()
        main, parts = ops.losses_forward(seg, sx, forward_out.transcript, forward_out.lengths, spec, target.to(torch.int64),
                                         batch.transcript_tf_target.to(torch.int64), self._loss_tmpl, self._loss_mw,
                                         self._loss_tw)
        return MuConLoss(main=main, transcript_loss=parts[0], length_loss=parts[1], mucon_loss=parts[2],
                         smoothing_loss=parts[3])

    def _loss_constants(self, dev):
        """Mask template and class-weight vectors on the device (built once per template / device)."""
        lc = self.cfg.model.loss
        key = (lc.mucon.template, str(dev))
        if getattr(self, "_loss_consts_key", None) != key:
            from .masks import _template
            self._loss_tmpl = _template(lc.mucon.template, 1, torch.zeros(1, device=dev)).reshape(-1).contiguous()
            self._loss_mw = (self._bg_weight(self.num_classes, lc.mucon_weight_background_index,
                                             lc.mucon_weight_background_value, dev) if lc.mucon_weight_background else None)
            self._loss_tw = (self._bg_weight(self.num_classes + 1, lc.transcript_weight_background_index,
                                             lc.transcript_weight_background_value, dev)
                             if lc.transcript_weight_background else None)
            self._loss_consts_key = key

    def _loss_spec(self) -> ops.LossSpec:
        lc = self.cfg.model.loss
        return ops.LossSpec(mucon_type=lc.mucon.type, overlap=float(lc.mucon.overlap), smoothing_clamp=bool(lc.smoothing.clamp),
                            clamp_min=float(lc.smoothing.clamp_min), clamp_max=float(lc.smoothing.clamp_max),
                            length_width=float(lc.length_width), transcript_average=bool(lc.transcript_average),
                            mul_transcript=float(self.loss_mul_transcript), mul_length=float(self.loss_mul_length),
                            mul_mucon=float(self.loss_mul_mucon), mul_smoothing=float(self.loss_mul_smoothing),
                            align_corners=bool(lc.mucon.get("align_corners", True)))

    def smoothing_loss(self, batch: Batch, forward_out: MuConForwardOut) -> Tensor:
        sm = self.cfg.model.loss.smoothing
        logp = getattr(forward_out, "_logp", None) if sm.log_softmax_before else None
        if logp is not None:
            return self._smoothing_from(logp)
        return self.calculate_smoothing_loss_for_logits(forward_out.segmentation)

    def _smoothing_from(self, x: Tensor) -> Tensor:
        sm = self.cfg.model.loss.smoothing
        values = F.mse_loss(x[1:, :], x[:-1, :].detach())
        if sm.clamp:
            values = torch.clamp(values, min=sm.clamp_min, max=sm.clamp_max)
        return torch.mean(values)

    def calculate_smoothing_loss_for_logits(self, logits):
        if self.cfg.model.loss.smoothing.log_softmax_before:
            logits = F.log_softmax(logits, dim=1)
        return self._smoothing_from(logits)

    def mucon_loss(self, batch: Batch, forward_out: MuConForwardOut) -> Tensor:
        if self.teacher_forcing:
            target = batch.transcript
        else:
            target = forward_out.transcript[:-1].argmax(dim=1)           # stays on the device
            target = torch.where(target >= self.num_classes, torch.zeros_like(target), target)
        T = forward_out.segmentation.shape[0]
        absolute_lengths = project_lengths_softmax(T=T, L=forward_out.lengths)
        mc = self.cfg.model.loss.mucon
        masks = create_masks(T=T, L=absolute_lengths, template=mc.template, overlap=mc.overlap,
                             align_corners=bool(mc.get("align_corners", True)))   # [N x T]
        return self.calculate_mucon_loss_using_masks(absolute_lengths, masks, forward_out.segmentation, target)

    def _bg_weight(self, n, index, value, device):
        w = torch.ones(n, dtype=torch.float32, device=device)
        w[index] = value
        return w

    def calculate_mucon_loss_using_masks(self, absolute_lengths, masks, segmentation, target_transcript):
        lc = self.cfg.model.loss
        weight = (self._bg_weight(self.num_classes, lc.mucon_weight_background_index, lc.mucon_weight_background_value,
                                  absolute_lengths.device) if lc.mucon_weight_background else None)
        kind = lc.mucon.type
        if kind == "flint":
            # mean of the masked logits per segment == (masks @ segmentation) / length, then log-softmax
            windows = (masks @ segmentation) / absolute_lengths.unsqueeze(1)                 # [N x M]
            return F.nll_loss(F.log_softmax(windows, dim=1), target_transcript, weight=weight, reduction="mean")
        if kind == "arithmetic":
            T = segmentation.size(0)
            total = 0
            for i in range(absolute_lengths.shape[0]):
                tgt = target_transcript[i].clone().detach().repeat(T).long().to(segmentation.device)
                total = total + (F.cross_entropy(segmentation, tgt, reduction="none", weight=weight) * masks[i]).sum()
            return total / T
        raise Exception(f"Invalid mucon type ({kind})")

    def length_loss(self, batch: Batch, forward_out: MuConForwardOut) -> Tensor:
        w = self.cfg.model.loss.length_width
        s = forward_out.lengths
        return F.relu(s - w).sum() + F.relu(-w - s).sum()

    def transcript_loss(self, batch: Batch, forward_out: MuConForwardOut) -> Tensor:
        lc = self.cfg.model.loss
        weight = (self._bg_weight(self.num_classes + 1, lc.transcript_weight_background_index,
                                  lc.transcript_weight_background_value, forward_out.transcript.device)
                  if lc.transcript_weight_background else None)
        return F.nll_loss(input=forward_out.transcript, target=batch.transcript_tf_target,
                          reduction="mean" if lc.transcript_average else "sum", weight=weight)

    def set_teacher_forcing(self, teacher_forcing: bool = True):
        self.teacher_forcing = teacher_forcing


@dataclass(repr=False)
class MuConFullySupervisedLoss(MuConLoss):
    classification_loss: Tensor = None
    supervised_length_loss: Tensor = None


class MuConFullySupervised(MuCon):
    """+ frame classification and length regression against the ground truth (reference models.py:781-868)."""

    def __init__(self, cfg, input_feature_size: int, num_classes: int, max_decoding_steps: int):
        super().__init__(cfg, input_feature_size=input_feature_size, num_classes=num_classes, max_decoding_steps=max_decoding_steps)
        fs = self.cfg.model.loss.fully_supervised
        self.loss_mul_classification, self.loss_mul_supervised_length = fs.mul_classification, fs.mul_supervised_length

    def classification_loss(self, batch, forward_out: MuConForwardOut) -> Tensor:
        return self.calculate_classification_loss_for_logit(forward_out.segmentation, batch.gt_label, batch.gt_label.shape[0],
                                                            logp=getattr(forward_out, "_logp", None))

    def calculate_classification_loss_for_logit(self, segmentation, target_labels, target_length, logp=None):
        if segmentation.shape[0] != target_length:
            segmentation = F.interpolate(segmentation.transpose(0, 1).unsqueeze(0), size=target_length).squeeze(0).transpose(0, 1)
            logp = None
        if logp is not None:      # the y-head kernel's own log-softmax: cross-entropy = nll of it
            return F.nll_loss(logp, target_labels, reduction="mean")
        return F.cross_entropy(segmentation, target_labels, reduction="mean")

    def supervised_length_loss(self, batch, forward_out: MuConForwardOut) -> Tensor:
        relative = batch.absolute_lengths / batch.absolute_lengths.sum()
        return F.mse_loss(relative, torch.softmax(forward_out.lengths, dim=0), reduction="mean")

    def _with_supervision(self, batch, forward_out: MuConForwardOut, supervised: bool) -> MuConFullySupervisedLoss:
        base = super().loss(batch, forward_out)          # the four weakly supervised losses (fused HIP kernels on the GPU)
        cls, sl = self.classification_loss(batch, forward_out), self.supervised_length_loss(batch, forward_out)
        main = base.main
        if supervised:
            main = main + self.loss_mul_classification * cls + self.loss_mul_supervised_length * sl
        return MuConFullySupervisedLoss(main=main, transcript_loss=base.transcript_loss, length_loss=base.length_loss,
                                        mucon_loss=base.mucon_loss, smoothing_loss=base.smoothing_loss,
                                        classification_loss=cls, supervised_length_loss=sl)

    def loss(self, batch, forward_out: MuConForwardOut) -> MuConFullySupervisedLoss:
        return self._with_supervision(batch, forward_out, True)


class MuConMixedSupervision(MuConFullySupervised):
    """Full supervision only on the videos the dataset marks (reference models.py:871-911)."""

    def loss(self, batch, forward_out: MuConForwardOut) -> MuConFullySupervisedLoss:
        return self._with_supervision(batch, forward_out, bool(batch.fully_supervised))
