"""The parts of the reference's train.py that belong to the hot path: ``load_model``
(train.py:75-83), ``init_distributed`` (:30-42), ``reduce_tensor`` (:23-27), plus the training
iteration itself (:293-340) as a function, and a synthetic batch generator that
stands in for the reference's dataset files (absent: .gitignore:1-9 of the reference).

Not rebuilt here (out of scope, SURVEY.md §2 rows 9,10,12): DataLoader/collate, tensorboard
logging, checkpoint rotation, validation loop.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L
from .distributed import apply_gradient_allreduce
from .loss_function import Tacotron2Loss
from .model import BERT_Tacotron2
from .optim import FusedAdam
from .utils import fit_cpu_threads


def reduce_tensor(tensor, n_gpus):
    """train.py:23-27: mean over ranks (logging only)."""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= n_gpus
    return rt


def init_distributed(hparams, n_gpus, rank, group_name=None):
    """train.py:30-42.  backend "nccl" is RCCL on ROCm; under torchrun the env:// rendezvous is
    used instead of hparams.dist_url."""
    assert torch.cuda.is_available(), "Distributed mode requires a GPU."
    torch.cuda.set_device(rank % torch.cuda.device_count())
    if "MASTER_ADDR" in os.environ and "RANK" in os.environ:
        dist.init_process_group(backend=hparams.dist_backend, init_method="env://", world_size=n_gpus, rank=rank)
    else:
        dist.init_process_group(backend=hparams.dist_backend, init_method=hparams.dist_url, world_size=n_gpus, rank=rank)


def load_model(hparams):
    """train.py:75-83.  (Also sizes torch's CPU thread pool to the container's CPU quota: utils.fit_cpu_threads.)"""
    fit_cpu_threads()
    model = BERT_Tacotron2(hparams).cuda()
    if hparams.fp16_run:
        # train.py:77-78 and :213-216: the reference switches apex amp (O2) on.  Here the flag moves the mask value only: the
        # arithmetic type of this implementation is chosen with _lib.set_precision ("f32" exact / "bf16" operands), and its
        # reduced-precision mode is bf16 with fp32 master weights, state and accumulation — no loss scaling, no fp16.
        print("fp16_run: no apex amp here — arithmetic stays " + L.get_precision() + " (set with tacotron2_subword_amd._lib.set_precision); "
              "only attention_layer.score_mask_value follows the flag")
        model.decoder.attention_layer.score_mask_value = float(np.finfo("float16").min)
    if hparams.distributed_run:
        model = apply_gradient_allreduce(model)
    return model


def warm_start_model(checkpoint_path, model, ignore_layers):
    """train.py:84-96.  `weights_only=True`: a checkpoint is a dict of tensors / numbers; nothing is unpickled into code."""
    assert os.path.isfile(checkpoint_path)
    print("Warm starting model from checkpoint '{}'".format(checkpoint_path))
    checkpoint_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    model_dict = checkpoint_dict["state_dict"]
    if len(ignore_layers) > 0:
        model_dict = {k: v for k, v in model_dict.items() if k not in ignore_layers}
        dummy_dict = model.state_dict()
        dummy_dict.update(model_dict)
        model_dict = dummy_dict
    model.load_state_dict(model_dict)
    return model


def load_checkpoint(checkpoint_path, model, optimizer):
    """train.py:99-112: same dict keys as the reference writes ('state_dict', 'optimizer', 'learning_rate',
    'iteration', optional 'val_loss'), so its checkpoints load here and vice versa (state_dict keys are unchanged)."""
    assert os.path.isfile(checkpoint_path)
    print("Loading checkpoint '{}'".format(checkpoint_path))
    checkpoint_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    model.load_state_dict(checkpoint_dict["state_dict"])
    optimizer.load_state_dict(checkpoint_dict["optimizer"])
    learning_rate = checkpoint_dict["learning_rate"]
    iteration = checkpoint_dict["iteration"]
    print("Loaded checkpoint '{}' from iteration {}".format(checkpoint_path, iteration))
    return model, optimizer, learning_rate, iteration


def save_checkpoint(model, optimizer, learning_rate, iteration, val_loss, filepath):
    """train.py:115-122."""
    print("Saving model and optimizer state at iteration {} to {}".format(iteration, filepath))
    torch.save({"iteration": iteration, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict(),
                "val_loss": val_loss, "learning_rate": learning_rate}, filepath)


def synthetic_batch(hparams, B, Tin=100, Tsub=60, T=400, seed=1234, ragged=True):
    """LJSpeech-shaped phone -> mel batch in the 10-tuple layout parse_batch takes
    (text, input_lengths, input_lengths_bert, mel [B,80,T], gate [B,T], output_lengths, sub_ids,
    phoneme_cls [B,Tin,768], bert_cls [B,Tsub,768], align)."""
    g = np.random.Generator(np.random.PCG64([seed, B, Tin, Tsub, T]))

    def lens(n):
        l = g.integers(max(1, int(0.7 * n)), n + 1, size=B) if ragged else np.full(B, n)
        l[0] = n
        return np.sort(l)[::-1].copy()

    tl, bl, ol = lens(Tin), lens(Tsub), lens(T)
    text = g.integers(1, hparams.n_symbols, size=(B, Tin))
    sub = g.integers(1, hparams.sub_n_symbols, size=(B, Tsub))
    M = hparams.n_mel_channels
    mel = np.clip(g.normal(-5.0, 2.0, size=(B, M, T)), -11.5, 2.0).astype(np.float32)
    gate = np.zeros((B, T), np.float32)
    for b in range(B):
        text[b, tl[b]:] = 0
        sub[b, bl[b]:] = 0
        mel[b, :, ol[b]:] = 0.0
        gate[b, ol[b] - 1:] = 1.0
    C = hparams.BERT_embedding_dim
    cls = np.repeat(g.normal(0, 1, size=(B, 1, C)).astype(np.float32), Tin, 1)
    bcls = np.repeat(g.normal(0, 1, size=(B, 1, C)).astype(np.float32), Tsub, 1)
    t = torch.from_numpy
    return (t(text).long(), t(tl).long(), t(bl).long(), t(mel), t(gate), t(ol).long(), t(sub).long(),
            t(cls.copy()), t(bcls.copy()), torch.zeros(B, T, Tin))


def train_step(model, criterion, optimizer, x, y, hparams, iteration=0):
    """One iteration of the reference loop (train.py:293-340): zero grads, forward, loss,
    backward (+ bucketed RCCL all-reduce when wrapped), clip, Adam step.  Returns the loss tensor
    (no host sync here; the reference's per-iteration .item() is the caller's choice)."""
    model.zero_grad()
    y_pred = model(x)
    loss, mel_loss, gate_loss, _, _ = criterion(y_pred, y, x, iteration)
    loss.backward()
    if hasattr(optimizer, "last_norm"):                        # optim.FusedAdam: clip + Adam in two HIP passes
        optimizer.step(max_norm=hparams.grad_clip_thresh)
    else:
        torch.nn.utils.clip_grad_norm_(model.parameters(), hparams.grad_clip_thresh)
        optimizer.step()
    return loss


def make_training_objects(hparams):
    """Model + Adam + loss as train.py:205-221 builds them."""
    torch.manual_seed(hparams.seed)
    torch.cuda.manual_seed(hparams.seed)
    model = load_model(hparams)
    # train.py:210-211 builds torch.optim.Adam(lr, weight_decay); FusedAdam is that optimizer with a HIP step()
    optimizer = FusedAdam(model.parameters(), lr=hparams.learning_rate, weight_decay=hparams.weight_decay)
    criterion = Tacotron2Loss(hparams.alignloss)
    return model, optimizer, criterion
