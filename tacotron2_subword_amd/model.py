"""Drop-in host module for the reference's ``model.py``: same classes, constructor arguments,
``state_dict`` keys and call signatures (``BERT_Tacotron2.parse_batch / forward / inference``),
with the decoder hot loop running in the hand-written HIP library behind the C ABI.

What runs where:
  * Decoder (both prenets, both attention LSTMs, SMA / LSA attention, decoder LSTM, mel/gate
    projections; teacher-forced forward + backward, autoregressive inference): fused HIP drivers.
  * Embeddings, encoder conv/BN stacks, BiLSTMs, linear converters, postnet conv/BN stack:
    HIP building blocks (blocks.py -> conv.hip, lstm.hip, gemm.hip), forward and backward.
  * torch is used for tensor storage, autograd bookkeeping, a few layout copies (cat / transpose)
    and the loss reductions.
Activations are channels-last [B,T,C] inside; the reference's [B,C,T] appears only at the API.
There is no CPU path: calling forward() with CPU tensors raises.
"""
from math import sqrt

import torch
from torch import nn

from . import _lib as L
from . import blocks
from . import ops
from .attention import (DynamicConvolutionAttention, ForwardAttentionV2, GMMAttention, LocationSensitiveAttention,
                        StepwiseMonotonicAttention)
from .layers import ConvNorm, LinearNorm
from .utils import get_mask_from_lengths, to_gpu


class Prenet(nn.Module):
    """Parameter container for model.py:13-24 (two bias-free linears; ReLU + always-on dropout
    are fused into the GEMM epilogues of the HIP decoder)."""

    def __init__(self, in_dim, sizes):
        super().__init__()
        ins = [in_dim] + sizes[:-1]
        self.layers = nn.ModuleList([LinearNorm(i, o, bias=False) for i, o in zip(ins, sizes)])


class Postnet(nn.Module):
    """model.py:27-70: five Conv1d(k=5)+BatchNorm1d blocks, tanh on all but the last."""

    def __init__(self, hparams):
        super().__init__()
        n, k = hparams.postnet_n_convolutions, hparams.postnet_kernel_size
        dims = [hparams.n_mel_channels] + [hparams.postnet_embedding_dim] * (n - 1) + [hparams.n_mel_channels]
        self.convolutions = nn.ModuleList()
        for i in range(n):
            self.convolutions.append(nn.Sequential(
                ConvNorm(dims[i], dims[i + 1], kernel_size=k, stride=1, padding=int((k - 1) / 2), dilation=1,
                         w_init_gain="tanh" if i < n - 1 else "linear"),
                nn.BatchNorm1d(dims[i + 1])))

    def forward_btc(self, x_btc, seed, residual=True):
        """x [B,T,n_mel] channels-last -> x + postnet(x) (model.py:557-558) as one fused stack."""
        n = len(self.convolutions)
        layers = [(blk[0].conv, blk[1]) for blk in self.convolutions]
        return blocks.conv_bn_stack(x_btc, layers, [blocks.ACT_TANH] * (n - 1) + [blocks.ACT_NONE], training=self.training,
                                    drop_p=0.5, seed=seed, site0=L.SITE["POSTNET0"], residual=residual)

    def forward(self, x):
        """Reference signature: [B,n_mel,T] -> [B,n_mel,T] (without the residual)."""
        y = self.forward_btc(x.transpose(1, 2).contiguous(), _next_seed(self), residual=False)
        return y.transpose(1, 2)


class Encoder(nn.Module):
    """model.py:73-125: three Conv1d(k=5)+BN+ReLU blocks and a BiLSTM."""

    def __init__(self, hparams):
        super().__init__()
        E, k = hparams.encoder_embedding_dim, hparams.encoder_kernel_size
        self.convolutions = nn.ModuleList([
            nn.Sequential(ConvNorm(E, E, kernel_size=k, stride=1, padding=int((k - 1) / 2), dilation=1, w_init_gain="relu"),
                          nn.BatchNorm1d(E))
            for _ in range(hparams.encoder_n_convolutions)])
        self.lstm = nn.LSTM(E, int(E / 2), 1, batch_first=True, bidirectional=True)

    def forward_btc(self, x_btc, input_lengths, site0, seed):
        """x [B,T,E] channels-last embedded input -> [B,T,E] (conv stack + packed / unpacked BiLSTM)."""
        layers = [(blk[0].conv, blk[1]) for blk in self.convolutions]
        h = blocks.conv_bn_stack(x_btc, layers, [blocks.ACT_RELU] * len(layers), training=self.training, drop_p=0.5,
                                 seed=seed, site0=site0)
        return blocks.bilstm(h, input_lengths, self.lstm)

    def forward(self, x, input_lengths):
        """Reference signature: x [B,E,T] (model.py:97-114)."""
        return self.forward_btc(x.transpose(1, 2).contiguous(), input_lengths, L.SITE["ENC0"], _next_seed(self))

    def inference(self, x):
        return self.forward_btc(x.transpose(1, 2).contiguous(), None, L.SITE["ENC0"], _next_seed(self))


def _next_seed(module):
    """Per-call dropout seed: (hparams.seed, call counter, rank) -> 64-bit."""
    module._t2_calls = getattr(module, "_t2_calls", 0) + 1
    rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
    return ((getattr(module, "base_seed", 1234) * 1000003 + module._t2_calls) * 64 + rank) & ((1 << 63) - 1)


class _FinalizeFn(torch.autograd.Function):
    """[B,T,C] -> [B,C,T] with frames >= length filled (parse_decoder_outputs + parse_output,
    model.py:290-320,531-541).  The reference fills in place on .data, i.e. the fill does not block
    gradients: backward is the plain transpose."""

    @staticmethod
    def forward(ctx, x_btc, lengths, fill):
        return ops.finalize_bct(x_btc.contiguous(), lengths, fill)

    @staticmethod
    def backward(ctx, d):
        return d.transpose(1, 2).contiguous(), None, None


class _DecoderFn(torch.autograd.Function):
    """Teacher-forced decoder pass as ONE autograd node: forward = t2_decoder_forward,
    backward = t2_decoder_backward (hand-written BPTT)."""

    @staticmethod
    def forward(ctx, memory, memory_sub, mels, mem_lengths, sub_lengths, cfg, *params):
        dec, keys = cfg["decoder"], cfg["keys"]
        P = {"decoder." + k: p.detach() for k, p in zip(keys, params)}
        P.update({"decoder." + k: v for k, v in dec.named_buffers()})          # DCA prior taps
        dims = dec.dims
        W = L.decoder_weights(P, dims.attention_kind, single=dec.single)
        memory, mels = memory.contiguous(), mels.contiguous()
        memory_sub = None if memory_sub is None else memory_sub.contiguous()
        pre = cfg.get("pre")
        if pre is not None:                     # Decoder.prologue ran the memory-independent part on its own stream
            torch.cuda.current_stream().wait_event(pre["event"])
        for f in list(ops.PRE_PERSISTENT):
            f()
        dp = ops.decoder_forward(W, dims, memory, memory_sub, mem_lengths, sub_lengths, mels,
                                 training=cfg["training"], prenet_dropout=cfg["prenet_dropout"], seed=cfg["seed"],
                                 dp=None if pre is None else pre["dp"])
        cfg.pop("pre", None)
        # The outputs leave the pass object: kept on ctx they would close a cycle output -> grad_fn -> ctx -> dp -> output that
        # runs through C++ shared_ptrs, which Python's GC cannot collect, so a grad-enabled forward that is never
        # backpropagated (validation without no_grad, an exception before backward) leaked the whole workspace.  What the
        # backward needs of them (the alignments) goes through save_for_backward, which autograd handles without a cycle.
        mel, gate, align, align_sub = dp.mel, dp.gate, dp.align, dp.align_sub
        dp.mel = dp.gate = dp.align = dp.align_sub = None
        ctx.cfg, ctx.dp, ctx.P, ctx.W = cfg, dp, P, W
        ctx.out_meta = (mel.shape, gate.shape, mel.dtype, mel.device)
        ctx.has_sub = memory_sub is not None
        ctx.save_for_backward(*((memory, align, align_sub) if memory_sub is None else (memory, memory_sub, align, align_sub)))
        ctx.set_materialize_grads(False)
        return mel, gate, align, align_sub

    @staticmethod
    def backward(ctx, d_mel, d_gate, d_align, d_align_sub):
        cfg, dp = ctx.cfg, ctx.dp
        saved = ctx.saved_tensors
        memory, memory_sub = saved[0], (saved[1] if ctx.has_sub else None)
        dp.align, dp.align_sub = saved[-2], saved[-1]
        mel_shape, gate_shape, dt, dev = ctx.out_meta
        c = lambda g, shape: torch.zeros(shape, dtype=dt, device=dev) if g is None else g.contiguous()
        cz = lambda g: None if g is None else g.contiguous()
        # Only d(memory) feeds further backward nodes (the encoders); the weight gradients are leaves.  With
        # defer_weight_grads the library leaves them on its side stream underneath the encoders' backward, and the
        # join is queued for the end of this backward pass (before anything reads .grad).
        # (only when every decoder gradient slot is empty: AccumulateGrad then takes the new tensor without reading it;
        # an accumulation into an existing .grad would read it on this stream too early)
        dec = cfg["decoder"]
        keep = [] if (getattr(dec, "defer_weight_grads", False) and all(p.grad is None for p in dec.parameters())) else None
        for f in list(ops.PRE_PERSISTENT):      # (e.g. gradient reductions launched by earlier backward nodes: see ops.PRE_PERSISTENT)
            f()
        G, dm, dms = ops.decoder_backward(ctx.W, ctx.P, cfg["decoder"].dims, dp, memory, memory_sub, c(d_mel, mel_shape),
                                          c(d_gate, gate_shape), training=cfg["training"], prenet_dropout=cfg["prenet_dropout"],
                                          seed=cfg["seed"], d_align=cz(d_align), d_align_sub=cz(d_align_sub), defer=keep)
        if keep is not None:
            def _join(keep=keep):
                ops.side_join()
                keep.clear()
            torch.autograd.Variable._execution_engine.queue_callback(_join)
        grads = tuple(G.get("decoder." + k) for k in cfg["keys"])
        ctx.dp = None
        return (dm, dms, None, None, None, None) + grads


class Decoder(nn.Module):
    """model.py:128-492.  Holds the parameters under the reference's names; the step loop lives in
    csrc/ (lstm.hip, attention.hip, c_api.hip)."""

    def __init__(self, hparams, single=False):
        super().__init__()
        hp = hparams
        self.single = bool(single)          # classic one-stream decoder (Tacotron2 class below)
        ns = 1 if single else 2
        self.n_mel_channels, self.n_frames_per_step = hp.n_mel_channels, hp.n_frames_per_step
        self.encoder_embedding_dim, self.attention_rnn_dim = hp.encoder_embedding_dim, hp.attention_rnn_dim
        self.decoder_rnn_dim, self.prenet_dim = hp.decoder_rnn_dim, hp.prenet_dim
        self.max_decoder_steps, self.gate_threshold = hp.max_decoder_steps, hp.gate_threshold
        self.p_attention_dropout, self.p_decoder_dropout = hp.p_attention_dropout, hp.p_decoder_dropout
        E, Ha, Hd, Pn = hp.encoder_embedding_dim, hp.attention_rnn_dim, hp.decoder_rnn_dim, hp.prenet_dim
        M = hp.n_mel_channels * hp.n_frames_per_step
        self.prenet = Prenet(M, [Pn, Pn])
        if not single:
            self.prenet_bert = Prenet(M, [Pn, Pn])
        self.attention_rnn = nn.LSTMCell(Pn + E, Ha)
        if not single:
            self.attention_rnn_bert = nn.LSTMCell(Pn + E, Ha)
        # The reference builds attention_layer_bert only for SMA and then uses it unconditionally
        # (model.py:158-191 vs :261,356); here both streams always get their module.
        att_cls = {"StepwiseMonotonicAttention": StepwiseMonotonicAttention, "ForwardAttentionV2": ForwardAttentionV2,
                   "GMMAttention": GMMAttention, "DynamicConvolutionAttention": DynamicConvolutionAttention,
                   "LSA": LocationSensitiveAttention, "LocationSensitiveAttention": LocationSensitiveAttention}.get(hp.attention)
        if att_cls is None:                                               # model.py:182-191: anything else is LSA
            att_cls = LocationSensitiveAttention
        print({"SMA": "Use SMA", "LSA": "Use LSA", "FWD2": "Use ForwardAttention", "GMM": "Use GMMA",
               "DCA": "Use DCA"}[att_cls.kind])                          # model.py:159-191
        args = (Ha, E, hp.attention_dim, hp.attention_location_n_filters, hp.attention_location_kernel_size)
        self.attention_layer = att_cls(*args)
        if not single:
            self.attention_layer_bert = att_cls(*args)
        self.decoder_rnn = nn.LSTMCell(ns * (Ha + E), Hd, 1)
        if not single:
            self.decoder_rnn_bert = nn.LSTMCell(Ha + E, Hd, 1)  # dead in the reference too (model.py:197-199, :375-378)
        self.linear_projection = LinearNorm(Hd + ns * E, M)
        self.gate_layer = LinearNorm(Hd + ns * E, 1, bias=True, w_init_gain="sigmoid")
        self._dims = L.dims_from_hparams(hp, ns)
        self.prenet_dropout = True          # model.py:23 (always on); tests switch it off for deterministic parity
        self.defer_weight_grads = False     # True: weight gradients finish on the library's side stream (see _DecoderFn); measured: +0.5 %
        self.base_seed, self._calls = int(getattr(hp, "seed", 1234)), 0

    # -- helpers ---------------------------------------------------------------------------
    @property
    def dims(self):
        """Model dimensions for the C ABI, with the attention modules' score_mask_value as it stands now: callers poke it
        from outside (train.py:77-78 sets decoder.attention_layer.score_mask_value for fp16 runs)."""
        d = self._dims
        d.score_mask_value = float(self.attention_layer.score_mask_value)
        d.score_mask_given = 1                                              # (an intentional 0.0 stays 0.0)
        if not self.single:
            d.score_mask_value_sub = float(self.attention_layer_bert.score_mask_value)
        return d

    def _param_keys(self):
        return L.decoder_param_keys(self.dims.attention_kind, self.single)

    def _params(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in self._param_keys()]

    def _next_seed(self):
        return _next_seed(self)

    def _weights(self):
        P = {"decoder." + k: v.detach() for k, v in self.named_parameters()}
        P.update({"decoder." + k: v for k, v in self.named_buffers()})           # DCA prior taps
        return P, L.decoder_weights(P, self.dims.attention_kind, single=self.single)

    def prologue(self, decoder_inputs, B, Tin, Tsub):
        """Start the part of forward() that needs no encoder output — teacher inputs, both prenets, the hoisted
        attention-LSTM input GEMMs (about 1 ms of chip-filling GEMMs at B=64, T=400) — on a side stream, so that it runs
        underneath the encoders' chains of small launches.  Returns the handle forward(pre=...) takes."""
        seed = self._next_seed()
        mels = decoder_inputs.contiguous()
        _, W = self._weights()
        dp = ops.DecoderPass(self.dims, B, mels.shape[2], Tin, Tsub, mels.device)      # allocated on the caller's stream
        cur = torch.cuda.current_stream()
        if getattr(self, "_pro_stream", None) is None:
            self._pro_stream = torch.cuda.Stream(device=mels.device)
        self._pro_stream.wait_stream(cur)
        with torch.cuda.stream(self._pro_stream):
            ops.decoder_prologue(W, self.dims, dp, mels, training=self.training, prenet_dropout=self.prenet_dropout, seed=seed)
            event = torch.cuda.Event()
            event.record()
        return dict(dp=dp, seed=seed, event=event, mels=mels, training=self.training, prenet_dropout=self.prenet_dropout)

    # -- reference surface -----------------------------------------------------------------
    def forward(self, memory, embeddings, decoder_inputs, memory_lengths, bert_lengths, channels_last=False, pre=None):
        """Decoder.forward (model.py:392-428): returns mel [B,n_mel,T] ([B,T,n_mel] if channels_last),
        gate [B,T], align [B,T,Tin], align_bert [B,T,Tsub].  pre: handle of prologue() for the same inputs."""
        if pre is not None:
            dp = pre["dp"]
            same = (dp.B, dp.T, dp.Tin) == (memory.shape[0], decoder_inputs.shape[2], memory.shape[1]) and \
                   (embeddings is None or dp.Tsub == embeddings.shape[1]) and pre["training"] == self.training and \
                   pre["prenet_dropout"] == self.prenet_dropout
            if not same:                        # shapes changed in between: wait the side work out and run the whole pass
                torch.cuda.current_stream().wait_event(pre["event"])
                pre = dict(seed=pre["seed"])
        cfg = dict(decoder=self, keys=self._param_keys(), training=self.training, prenet_dropout=self.prenet_dropout,
                   seed=self._next_seed() if pre is None else pre["seed"])
        if pre is not None and "dp" in pre:
            cfg["pre"] = pre
        mel, gate, al, alb = _DecoderFn.apply(memory, embeddings, decoder_inputs, memory_lengths, bert_lengths, cfg,
                                              *self._params())
        return (mel if channels_last else mel.transpose(1, 2)), gate, al, alb

    def inference(self, memory, embeddings, channels_last=False):
        """Decoder.inference (model.py:430-492) for any batch size; per-item stop rule of SURVEY §8a A17.
        Returns mel [B,n_mel,T'], gate [B,T',1], align, align_bert, INFER_FLAG."""
        P, W = self._weights()
        with torch.no_grad():
            dp, steps, stop = ops.decoder_infer(W, self.dims, memory.contiguous(), None if embeddings is None else embeddings.contiguous(),
                                                max_steps=int(self.max_decoder_steps), gate_threshold=float(self.gate_threshold),
                                                prenet_dropout=self.prenet_dropout, seed=self._next_seed())
        stop = stop.cpu()
        ops.check_chain_status()                # (the copy above has waited for the whole loop: the status word is final)
        flag = bool((stop >= 0).all())
        n = int(stop.max()) + 1 if flag else steps
        self.last_stop_index = stop
        # decoder steps actually run / run past the last stop (the stop test is polled without draining the queue:
        # include/t2amd.h t2_decoder_infer); what a latency-sensitive caller pays beyond the frames it gets
        self.last_steps_run, self.last_overshoot = steps, steps - n
        mel = dp.mel[:, :n]
        return ((mel if channels_last else mel.transpose(1, 2)), dp.gate[:, :n].unsqueeze(-1), dp.align[:, :n], dp.align_sub[:, :n], flag)


class BERT_Tacotron2(nn.Module):
    """model.py:494-582: phone + sub-word dual-stream Tacotron2."""

    def __init__(self, hparams):
        super().__init__()
        hp = hparams
        self.mask_padding, self.fp16_run = hp.mask_padding, hp.fp16_run
        self.n_mel_channels, self.n_frames_per_step = hp.n_mel_channels, hp.n_frames_per_step
        self.embedding = nn.Embedding(hp.n_symbols, hp.symbols_embedding_dim)
        self.embedding_sub = nn.Embedding(hp.sub_n_symbols, hp.symbols_embedding_dim)
        val = sqrt(3.0) * sqrt(2.0 / (hp.n_symbols + hp.symbols_embedding_dim))     # model.py:503-506
        self.embedding.weight.data.uniform_(-val, val)
        self.embedding_sub.weight.data.uniform_(-val, val)
        self.encoder = Encoder(hp)
        self.encoder_sub = Encoder(hp)
        self.linear_converter = LinearNorm(hp.encoder_embedding_dim + hp.BERT_embedding_dim, hp.encoder_embedding_dim)
        self.linear_converter_sub = LinearNorm(hp.encoder_embedding_dim + hp.BERT_embedding_dim, hp.encoder_embedding_dim)
        self.decoder = Decoder(hp)
        self.postnet = Postnet(hp)
        self.overlap_encoders, self._t2_side = True, None

    def parse_batch(self, batch):
        """model.py:517-529: 10-tuple -> (x 9-tuple, y 3-tuple) on the GPU."""
        text, il, ilb, mel, gate, ol, sub, pcls, bcls, align = batch
        # The two maxima (model.py:521-524) are host numbers the loader already knows: data_utils.batch_to_device hands them
        # over as `host_max`, and lengths that are still host tensors are read before they go up.  Only bare device
        # tensors cost the reference's two .item() round trips, each of which drains the whole launch queue.
        hm = getattr(batch, "host_max", None)
        if hm is None and not (il.is_cuda or ilb.is_cuda or ol.is_cuda):
            hm = (int(torch.max(torch.cat((il, ilb), 0))), int(torch.max(ol)))
        text, il, ilb = to_gpu(text).long(), to_gpu(il).long(), to_gpu(ilb).long()
        if hm is None:
            hm = (int(torch.max(torch.cat((il, ilb), 0)).item()), int(torch.max(ol).item()))
        max_in, max_out = int(hm[0]), int(hm[1])
        mel, gate, ol, align = to_gpu(mel).float(), to_gpu(gate).float(), to_gpu(ol).long(), to_gpu(align).float()
        return ((text, il, ilb, mel, (max_in, max_out), ol, to_gpu(sub), to_gpu(pcls), to_gpu(bcls)), (mel, gate, align))

    def parse_output(self, outputs, output_lengths=None):
        """model.py:531-541 — including the in-place fill on .data (the postnet's first conv has
        already saved mel as its input, so its weight gradient sees the masked mel)."""
        if self.mask_padding and output_lengths is not None:
            mask = ~get_mask_from_lengths(output_lengths, outputs[0].size(2))
            outputs[0].data.masked_fill_(mask[:, None, :], 0.0)
            outputs[1].data.masked_fill_(mask[:, None, :], 0.0)
            outputs[2].data.masked_fill_(mask, 1e3)
        return outputs

    def _front(self, ids, lengths, cls, sub, seed):
        """embedding -> encoder -> cat CLS -> linear converter (model.py:546-554 / 563-572), channels-last."""
        emb, enc, conv = ((self.embedding_sub, self.encoder_sub, self.linear_converter_sub) if sub
                          else (self.embedding, self.encoder, self.linear_converter))
        x = blocks.embedding(ids, emb.weight)                                   # [B,T,E]
        h = enc.forward_btc(x, lengths, L.SITE["ENCSUB0" if sub else "ENC0"], seed)
        cat = torch.cat([h, cls[:, :h.size(1)]], 2)
        return blocks.linear(cat, conv.linear_layer.weight, conv.linear_layer.bias)

    def _fronts(self, text, tl, pcls, sub_ids, bl, bcls, seed):
        """Both front ends.  They are independent until the decoder and each is a chain of small launches (three
        conv+BN layers, a BiLSTM recurrence), so the sub-word one runs on a side stream next to the phone one;
        autograd replays each backward on its forward stream.  `overlap_encoders = False` keeps one stream."""
        if not (self.overlap_encoders and text.is_cuda):
            return self._front(text, tl, pcls, False, seed), self._front(sub_ids, bl, bcls, True, seed)
        cur = torch.cuda.current_stream()
        if self._t2_side is None:
            self._t2_side = torch.cuda.Stream(device=text.device)
        side = self._t2_side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            memory_sub = self._front(sub_ids, bl, bcls, True, seed)
        memory = self._front(text, tl, pcls, False, seed)
        cur.wait_stream(side)
        memory_sub.record_stream(cur)
        return memory, memory_sub

    def forward(self, inputs):
        text, tl, bl, mels, _, ol, sub_ids, pcls, bcls = inputs
        tl, bl, ol = tl.data, bl.data, ol.data
        seed = _next_seed(self)
        pre = None
        if self.overlap_encoders and text.is_cuda:                              # decoder prologue underneath the encoders
            pre = self.decoder.prologue(mels, text.shape[0], text.shape[1], sub_ids.shape[1])
        memory, memory_sub = self._fronts(text, tl, pcls, sub_ids, bl, bcls, seed)
        mel_btc, gate, al, alb = self.decoder(memory, memory_sub, mels, tl, bl, channels_last=True, pre=pre)
        post_btc = self.postnet.forward_btc(mel_btc, seed)                      # mel + postnet(mel), [B,T,n_mel]
        if self.mask_padding and ol is not None:
            mel = _FinalizeFn.apply(mel_btc, ol, 0.0)
            post = _FinalizeFn.apply(post_btc, ol, 0.0)
            gate = gate.clone()                                                 # keep the decoder's own buffer intact
            ops.mask_bt_(gate.data, ol, 1e3)
            # reference quirk (model.py:537): the fill happens in place on the tensor the postnet's first conv
            # saved as its input, so that conv's weight gradient sees the masked mel
            ops.mask_btc_(mel_btc.data, ol, 0.0)
        else:
            mel = _FinalizeFn.apply(mel_btc, None, 0.0)
            post = _FinalizeFn.apply(post_btc, None, 0.0)
        return [mel, post, gate, al, alb]

    def inference(self, inputs, embeddings, phoneme_embeddings_cls, bert_embeddings_cls):
        seed = _next_seed(self)
        memory, memory_sub = self._fronts(inputs, None, phoneme_embeddings_cls, embeddings, None, bert_embeddings_cls, seed)
        mel_btc, gate, al, alb, flag = self.decoder.inference(memory, memory_sub, channels_last=True)
        mel_btc = mel_btc.contiguous()
        post_btc = self.postnet.forward_btc(mel_btc, seed)
        return [ops.finalize_bct(mel_btc, None, 0.0), ops.finalize_bct(post_btc, None, 0.0), gate, al, alb, flag]


class Tacotron2(nn.Module):
    """Classic single-stream Tacotron2 with the API the reference's stale callers expect (GTA.py:6,21,57-59,
    inference.py:302,334, streamlitNews.py:8): forward((text, in_len, mel, max_len, out_len)) ->
    [mel, mel_postnet, gate, align]; inference(ids) -> [mel, mel_postnet, gate, align].  The reference itself
    has no such class (model.py exports only BERT_Tacotron2), so this class has no reference oracle: its parity
    is pinned by the CPU restatement only (oracle.forward_single).  Same kernels, n_streams = 1; the attention
    type follows hparams.attention (SMA by default in this repo's hparams, LSA in the classic recipe)."""

    def __init__(self, hparams):
        super().__init__()
        hp = hparams
        self.mask_padding, self.fp16_run = hp.mask_padding, hp.fp16_run
        self.n_mel_channels, self.n_frames_per_step = hp.n_mel_channels, hp.n_frames_per_step
        self.embedding = nn.Embedding(hp.n_symbols, hp.symbols_embedding_dim)
        val = sqrt(3.0) * sqrt(2.0 / (hp.n_symbols + hp.symbols_embedding_dim))
        self.embedding.weight.data.uniform_(-val, val)
        self.encoder = Encoder(hp)
        self.decoder = Decoder(hp, single=True)
        self.postnet = Postnet(hp)

    def parse_batch(self, batch):
        text, il, mel, gate, ol = batch
        max_len = None if il.is_cuda else int(torch.max(il))                # host lengths: no device round trip
        text, il = to_gpu(text).long(), to_gpu(il).long()
        if max_len is None:
            max_len = int(torch.max(il).item())
        mel, gate, ol = to_gpu(mel).float(), to_gpu(gate).float(), to_gpu(ol).long()
        return ((text, il, mel, max_len, ol), (mel, gate))

    def forward(self, inputs):
        text, tl, mels, _, ol = inputs
        tl, ol = tl.data, ol.data
        seed = _next_seed(self)
        memory = self.encoder.forward_btc(blocks.embedding(text, self.embedding.weight), tl, L.SITE["ENC0"], seed)
        mel_btc, gate, al, _ = self.decoder(memory, None, mels, tl, None, channels_last=True)
        post_btc = self.postnet.forward_btc(mel_btc, seed)
        use_mask = self.mask_padding and ol is not None
        mel = _FinalizeFn.apply(mel_btc, ol if use_mask else None, 0.0)
        post = _FinalizeFn.apply(post_btc, ol if use_mask else None, 0.0)
        if use_mask:
            gate = gate.clone()
            ops.mask_bt_(gate.data, ol, 1e3)
            ops.mask_btc_(mel_btc.data, ol, 0.0)
        return [mel, post, gate, al]

    def inference(self, inputs):
        seed = _next_seed(self)
        memory = self.encoder.forward_btc(blocks.embedding(inputs, self.embedding.weight), None, L.SITE["ENC0"], seed)
        mel_btc, gate, al, _, flag = self.decoder.inference(memory, None, channels_last=True)
        mel_btc = mel_btc.contiguous()
        post_btc = self.postnet.forward_btc(mel_btc, seed)
        return [ops.finalize_bct(mel_btc, None, 0.0), ops.finalize_bct(post_btc, None, 0.0), gate, al]
