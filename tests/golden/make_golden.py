#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE model.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  The reference's ``model.py`` is imported unmodified; three harness-side shims
 make it runnable on CPU:
  1. a stub ``librosa`` module (imported at module top by layers.py/utils.py, unused by the path);
  2. ``model.get_mask_from_lengths`` rebound to a device-agnostic version (utils.py:12
     allocates a torch.cuda tensor);
  3. for LSA only: ``decoder.attention_layer_bert`` constructed harness-side
     (model.py:164-167 only builds it in the SMA branch, yet :261,356 use it always).
Randomness is replayed, not drawn: ``F.dropout`` / ``Tensor.normal_`` are patched during
the reference call so that the recipe's keep-masks / noise (oracle/recipe.py) are used.

Only inputs-by-recipe + reference OUTPUTS are stored (weights are rebuilt from the recipe).
Usage:  python tests/golden/make_golden.py
"""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import recipe  # noqa: E402
from oracle import tacotron2_oracle as O  # noqa: E402


def _stub(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
    return m


def import_reference():
    lib, f, u = _stub("librosa"), _stub("librosa.filters"), _stub("librosa.util")
    lib.filters, lib.util = f, u
    f.mel = lambda *a, **k: None
    u.normalize = u.pad_center = u.tiny = lambda *a, **k: None
    sys.path.insert(0, "/root/reference")
    import attention as ref_attention  # noqa
    import hparams as ref_hparams  # noqa
    import loss_function as ref_loss  # noqa
    import model as ref_model  # noqa
    ref_model.get_mask_from_lengths = lambda lengths: O.get_mask_from_lengths(lengths)
    return ref_model, ref_attention, ref_hparams, ref_loss


class Replay:
    """Patches F.dropout and Tensor.normal_ while the reference runs."""

    def __init__(self, drop_queue=None, noise_queue=None):
        self.dq, self.nq = drop_queue, noise_queue

    def __enter__(self):
        import torch.nn.functional as F
        self._F, self._drop, self._normal = F, F.dropout, torch.Tensor.normal_
        dq, nq = self.dq, self.nq

        def dropout(x, p=0.5, training=True, inplace=False):
            if dq is None:
                return x                                            # deterministic set: identity
            if not training:
                return x
            keep = dq.pop(0)
            assert keep.shape == x.shape, (keep.shape, x.shape)
            return x * keep * (1.0 / (1.0 - p))

        def normal_(self_, *a, **k):
            n = nq.pop(0)
            assert n.shape == self_.shape
            return self_.copy_(n)

        F.dropout = dropout
        if nq is not None:
            torch.Tensor.normal_ = normal_
        return self

    def __exit__(self, *exc):
        self._F.dropout = self._drop
        torch.Tensor.normal_ = self._normal
        if exc[0] is None:
            assert not self.dq, "unused dropout masks: call order mismatch"
            assert not self.nq, "unused noise tensors"


def build_reference(ref_model, ref_attention, ref_hparams, attention, seed=1234):
    hps = ref_hparams.create_hparams()
    hps.attention = attention
    m = ref_model.BERT_Tacotron2(hps)
    if attention != "StepwiseMonotonicAttention":
        # model.py:158-191 builds attention_layer_bert only for SMA and then uses it unconditionally (:261,356): the
        # harness supplies the missing module of the same class
        cls = {"ForwardAttentionV2": ref_attention.ForwardAttentionV2, "GMMAttention": ref_attention.GMMAttention,
               "DynamicConvolutionAttention": ref_attention.DynamicConvolutionAttention}.get(
            attention, ref_attention.LocationSensitiveAttention)
        m.decoder.attention_layer_bert = cls(
            hps.attention_rnn_dim, hps.encoder_embedding_dim, hps.attention_dim,
            hps.attention_location_n_filters, hps.attention_location_kernel_size)
    hp = O.default_hparams()
    hp["attention"] = attention
    spec = recipe.state_dict_spec(hp)
    sd = m.state_dict()
    if attention == "StepwiseMonotonicAttention":
        assert [k for k, _, _ in spec] == list(sd.keys()), "state_dict key contract mismatch"
    else:   # the harness-side attention_layer_bert registers last, so only the key SET can match
        assert sorted(k for k, _, _ in spec) == sorted(sd.keys()), "state_dict key contract mismatch"
    for k, s, _ in spec:
        assert tuple(sd[k].shape) == s, (k, tuple(sd[k].shape), s)
    W = recipe.make_weights(hp, seed)
    m.load_state_dict(W)
    return m, hp, W


def drop_queue_from_rnd(rnd, B, training):
    """Reference call order of F.dropout in one forward (model.py:99,23,341-373,67-68)."""
    q = []
    if training:
        q += list(rnd["enc_keep"]) + list(rnd["encsub_keep"])
    ones = lambda k: torch.cat([k, torch.ones(1, *k.shape[1:])], 0)   # frame T+1 of the prenet is unused
    q += [ones(k) for k in rnd["prenet_keep"]] + [ones(k) for k in rnd["prenet_bert_keep"]]
    T = rnd["att_h_keep"].shape[0]
    if training:
        for t in range(T):
            q += [rnd["att_h_keep"][t], rnd["att_c_keep"][t], rnd["att_h_bert_keep"][t], rnd["att_c_bert_keep"][t],
                  rnd["dec_h_keep"][t], rnd["dec_c_keep"][t]]
        q += list(rnd["post_keep"])
    return q


def trace_steps(m):
    """Record decoder state after every decode() call of the reference."""
    rec = []
    orig = m.decoder.decode

    def decode(a, b):
        out = orig(a, b)
        d = m.decoder
        rec.append(dict(att_h=d.attention_hidden, att_c=d.attention_cell, att_h_bert=d.attention_hidden_bert,
                        ctx=d.attention_context, ctx_bert=d.attention_context_bert,
                        dec_h=d.decoder_hidden, dec_c=d.decoder_cell))
        return out

    m.decoder.decode = decode
    return rec


def np_(t):
    return t.detach().cpu().numpy()


def gen_forward(refs, attention, name, B, Tin, Tsub, T, training, with_grads=False, steps_to_keep=(0, 1, 2, 3, -1)):
    ref_model, ref_attention, ref_hparams, ref_loss = refs
    m, hp, W = build_reference(ref_model, ref_attention, ref_hparams, attention)
    batch = recipe.make_batch(hp, B, Tin, Tsub, T, seed=1234)
    m.train(training)
    rec = trace_steps(m)
    rnd = recipe.make_rnd(hp, B, Tin, Tsub, T, seed=7) if training else None
    dq = drop_queue_from_rnd(rnd, B, training) if training else None
    nq = None
    if training and attention == "StepwiseMonotonicAttention":
        nq = []
        for t in range(T):
            nq += [rnd["sma_noise"][t], rnd["sma_noise_bert"][t]]
    x, y = m.parse_batch(batch)
    out = {}
    ctxmgr = Replay(dq, nq)
    with ctxmgr, torch.set_grad_enabled(with_grads):
        y_pred = m(x)
        if with_grads:
            crit = ref_loss.Tacotron2Loss("")
            loss, mel_loss, gate_loss, _, _ = crit(y_pred, y, x, 0)
            loss.backward()
            out["loss"] = np_(loss); out["mel_loss"] = np_(mel_loss); out["gate_loss"] = np_(gate_loss)
    for k, v in zip(("mel", "mel_postnet", "gate", "align", "align_bert"), y_pred):
        out[k] = np_(v)
    for i in steps_to_keep:
        for k, v in rec[i].items():
            out[f"step{i if i >= 0 else 'last'}_{k}"] = np_(v)
    out["memory"] = np_(m.decoder.memory); out["memory_sub"] = np_(m.decoder.bert)
    if with_grads:
        keys, stats = [], []
        small = {}
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            g = p.grad
            keys.append(k)
            stats.append([float(g.double().norm()), float(g.double().sum()), float(g.double().abs().max())])
            if g.numel() <= 4096:
                small["grad/" + k] = np_(g)
            else:
                small["gradhead/" + k] = np_(g.reshape(-1)[:256])
        out["grad_keys"] = np.array(keys)
        out["grad_stats"] = np.array(stats, dtype=np.float64)
        out.update(small)
        assert all(p.grad is None for k, p in m.named_parameters() if k.startswith("decoder.decoder_rnn_bert"))
    if training:
        sd = m.state_dict()
        for k in sd:
            if "running_" in k or "num_batches" in k:
                out["bn/" + k] = np_(sd[k])
    out["meta"] = np.array([B, Tin, Tsub, T, int(training)])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_inference(refs, attention, name, Tin, Tsub, fixed_steps=48):
    ref_model, ref_attention, ref_hparams, ref_loss = refs
    m, hp, W = build_reference(ref_model, ref_attention, ref_hparams, attention)
    m.eval()
    batch = recipe.make_batch(hp, 1, Tin, Tsub, 8, seed=4321, ragged=False)
    ids, sub, pcls, bcls = batch[0], batch[6], batch[7], batch[8]
    out = {}
    with Replay(None, None), torch.no_grad():
        m.decoder.gate_threshold, m.decoder.max_decoder_steps = 2.0, fixed_steps   # never stops
        r = m.inference(ids, sub, pcls, bcls)
        assert r[5] is False and r[0].shape[2] == fixed_steps
        for k, v in zip(("mel", "mel_postnet", "gate", "align", "align_bert"), r[:5]):
            out["fixed_" + k] = np_(v)
        sg = torch.sigmoid(r[2].reshape(-1))
        # data-driven stop (random weights give sigmoid(gate) ~ 0.5 >> hparams' 0.001): put the
        # threshold midway between frame 5 and frame 6 of the (rising) gate curve -> stop index 6
        s_ = 6
        assert float(sg[s_]) > float(sg[:s_].max()), "gate curve not rising at the chosen frame"
        thr = 0.5 * (float(sg[:s_].max()) + float(sg[s_]))
        stop = s_
        m.decoder.gate_threshold, m.decoder.max_decoder_steps = thr, 1000
        r2 = m.inference(ids, sub, pcls, bcls)
        assert r2[5] is True and r2[0].shape[2] == stop + 1, (r2[0].shape, stop)
        out["stop_threshold"] = np.array(thr, dtype=np.float64)
        out["stop_index"] = np.array(stop)
        for k, v in zip(("mel", "mel_postnet", "gate"), r2[:3]):
            out["stop_" + k] = np_(v)
    out["meta"] = np.array([1, Tin, Tsub, fixed_steps])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "stop index", stop, "thr", thr)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    refs = import_reference()
    SMA, LSA = "StepwiseMonotonicAttention", "LSA"
    gen_forward(refs, SMA, "sma_small_eval", 2, 13, 8, 12, training=False)
    gen_forward(refs, LSA, "lsa_small_eval", 2, 13, 8, 12, training=False)
    gen_forward(refs, SMA, "sma_small_train", 3, 13, 8, 12, training=True, with_grads=True)
    gen_forward(refs, LSA, "lsa_small_train", 3, 13, 8, 12, training=True, with_grads=True)
    gen_forward(refs, SMA, "sma_baseline_eval", 2, 100, 60, 400, training=False, steps_to_keep=(0, 199, -1))
    gen_inference(refs, SMA, "sma_infer", 21, 11)
    gen_inference(refs, LSA, "lsa_infer", 21, 11)
    gen_fa2(refs)
    gen_gmm(refs)
    gen_dca(refs)


def gen_fa2(refs):
    """ForwardAttentionV2 — added after the first fixtures; `python make_golden.py fa2` writes only these."""
    FA2 = "ForwardAttentionV2"
    gen_forward(refs, FA2, "fa2_small_eval", 2, 13, 8, 12, training=False)
    gen_forward(refs, FA2, "fa2_small_train", 3, 13, 8, 12, training=True, with_grads=True)


def gen_gmm(refs):
    """GMMAttention; `python make_golden.py gmm` writes only these."""
    GMM = "GMMAttention"
    gen_forward(refs, GMM, "gmm_small_eval", 2, 13, 8, 12, training=False)
    gen_forward(refs, GMM, "gmm_small_train", 3, 13, 8, 12, training=True, with_grads=True)


def gen_dca(refs):
    """DynamicConvolutionAttention; `python make_golden.py dca` writes only these.  Its init_attention
    hard-codes `.cuda()` (attention.py:234); on this GPU-less box the harness makes Tensor.cuda the identity."""
    DCA = "DynamicConvolutionAttention"
    torch.Tensor.cuda = lambda self, *a, **k: self
    gen_forward(refs, DCA, "dca_small_eval", 2, 13, 8, 12, training=False)
    gen_forward(refs, DCA, "dca_small_train", 3, 13, 8, 12, training=True, with_grads=True)


if __name__ == "__main__":
    if sys.argv[1:] == ["dca"]:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        gen_dca(import_reference())
    elif sys.argv[1:] == ["gmm"]:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        gen_gmm(import_reference())
    elif sys.argv[1:] == ["fa2"]:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        gen_fa2(import_reference())
    else:
        main()
