"""Input pipeline in the reference's on-disk formats (data_utils.py:47-206, train.py:286-316):

    <dataset_root>/<split>/ljspeech-mel-%05d.npy   float [T, 80]      (index = item + 1)
    text list file: one path per line -> .npy [T_in, 2], column 0 = phone id (column 1 = duration in frames)
    <embedding_path>/<item>.npy                    int   [T_sub]       sub-word token ids
    <embedding_cls_path>/<item>.npy                float [1, 768]      BERT CLS vector (repeated along time)

`collate_fn` keeps the reference's contract: the DataLoader batch holds batch_size**2 items, which are sorted by
phone count and cut into batch_size lists of batch_size items (so a loader step yields a LIST of model batches).
`batch_to_device` replaces the ten synchronous `.to(device)` copies per iteration of train.py:295-314 by one pinned
staging buffer per field and non-blocking copies on the current stream; the tuple it returns is what
`BERT_Tacotron2.parse_batch` takes.  No tokenizer / BERT model is needed here: the dataset reads their precomputed
outputs, exactly like the reference's Dataset."""
from __future__ import annotations

import math
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .utils import create_alignment


def process_text(train_text_path):
    with open(train_text_path, "r", encoding="utf-8") as f:
        return [line.strip() for line in f.readlines()]


class BERTTacotron2Dataset(Dataset):
    def __init__(self, dataset_path="train", text_path=None, embedding_path=None, embedding_cls_path=None,
                 dataset_root="dataset", alignloss=""):
        self.dataset_path = os.path.join(dataset_root, dataset_path)
        self.text_path = text_path
        self.text = process_text(self.text_path)
        self.embedding_path = embedding_path
        self.embedding_cls_path = embedding_cls_path
        self.alignloss = alignloss

    def __len__(self):
        return len(self.text)

    def __getitem__(self, idx):
        mel_target = np.load(os.path.join(self.dataset_path, "ljspeech-mel-%05d.npy" % (idx + 1)))
        phone_file = np.load(self.text[idx]).astype(int)
        phoneme = torch.from_numpy(phone_file)[:, 0]
        bert_embedding = torch.from_numpy(np.load(os.path.join(self.embedding_path, str(idx) + ".npy")))
        embedding_cls = torch.from_numpy(np.load(os.path.join(self.embedding_cls_path, str(idx) + ".npy")))
        stop_token = np.array([0. for _ in range(mel_target.shape[0])])
        stop_token[-1] = 1.
        sample = {"text": phoneme, "mel_target": mel_target, "bert_embedding": bert_embedding,
                  "bert_embedding_cls": embedding_cls.repeat(bert_embedding.size(0), 1),
                  "phoneme_embedding_cls": embedding_cls.repeat(phoneme.size(0), 1), "stop_token": stop_token}
        if self.alignloss != "":
            sample["duration"] = torch.from_numpy(phone_file)[:, 1]
        return sample


def get_alignment(filename):
    """Hard alignment [T, T_in] from the duration column of a phone file (data_utils.py:135-144)."""
    dur = torch.from_numpy(np.load(filename).astype(int))[:, 1].unsqueeze(0)
    alignment = torch.zeros(1, int(dur.sum()), dur.size(1))
    return create_alignment(alignment, dur)[0]


# ---------------------------------------------------------------------------------------------------------------
# Collation straight into host staging.  One model batch = one _Stage: a flat buffer per field, page-locked when the
# process has a CUDA context (a DataLoader worker has none: pageable there), written once per batch by the loops below —
# no per-item pad / stack temporaries — and handed to the GPU by batch_to_device with one non-blocking copy per field.
# A ring of stages keeps the batches of two loader steps apart, so batch k+1 is collated and uploaded while step k runs;
# each stage carries the event recorded behind its last upload and is refilled only after that event.
# ---------------------------------------------------------------------------------------------------------------
class _Stage:
    def __init__(self):
        self.buf, self.event = {}, None

    def wait(self):
        if self.event is not None:
            self.event.synchronize()
            self.event = None

    def take(self, name, shape, dtype, fill=None):
        n = 1
        for d in shape:
            n *= int(d)
        b = self.buf.get(name)
        if b is None or b.numel() < n or b.dtype != dtype:
            b = torch.empty(max(n, 1), dtype=dtype)
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                b = b.pin_memory()
            self.buf[name] = b
        v = b[:n].view(*shape)
        if fill is not None:
            v.fill_(fill)
        return v


class StagedBatch(dict):
    """A collated model batch (the reference's dict keys) whose arrays live in `stage`."""
    stage = None


class _Ring:
    def __init__(self):
        self.stages, self.next = [], 0

    def get(self, per_step):
        want = max(2 * per_step, 2)                       # the batches of two loader steps
        while len(self.stages) < want:
            self.stages.append(_Stage())
        st = self.stages[self.next % len(self.stages)]
        self.next += 1
        st.wait()
        return st


_RING = _Ring()


def _collate_cut(batch, cut, stage):
    """One model batch (data_utils.py:98-133), every field written in place: zero (or pad-value) fill, then one slice
    assignment per item.  Dtypes and values are those of the reference's pad_normal / pad_mel / pad_emb."""
    items = [batch[int(i)] for i in cut]
    n = len(items)
    n_text = [int(it["text"].shape[0]) for it in items]
    n_sub = [int(it["bert_embedding"].shape[0]) for it in items]
    n_mel = [int(it["mel_target"].shape[0]) for it in items]
    Tin, Tsub, T = max(n_text), max(n_sub), max(n_mel)
    mel0 = items[0]["mel_target"]
    text = stage.take("text", (n, Tin), items[0]["text"].dtype, 0)
    sub = stage.take("sub", (n, Tsub), items[0]["bert_embedding"].dtype, 0)
    mel = stage.take("mel", (n, T, mel0.shape[1]), torch.from_numpy(mel0[:0]).dtype, 0)
    stop = stage.take("stop", (n, T), torch.float64, 1.0)
    C = items[0]["bert_embedding_cls"].shape[1]
    pcls = stage.take("pcls", (n, Tin, C), items[0]["phoneme_embedding_cls"].dtype, 0)
    bcls = stage.take("bcls", (n, Tsub, C), items[0]["bert_embedding_cls"].dtype, 0)
    for k, it in enumerate(items):
        text[k, :n_text[k]] = it["text"]
        sub[k, :n_sub[k]] = it["bert_embedding"]
        mel[k, :n_mel[k]] = torch.from_numpy(np.ascontiguousarray(it["mel_target"]))
        stop[k, :n_mel[k]] = torch.from_numpy(np.asarray(it["stop_token"], dtype=np.float64))
        pcls[k, :n_text[k]] = it["phoneme_embedding_cls"]
        bcls[k, :n_sub[k]] = it["bert_embedding_cls"]
    lt = stage.take("length_text", (n,), torch.float64)
    lb = stage.take("length_bert", (n,), torch.float64)
    lm = stage.take("length_mel", (n,), torch.float64)
    lt.copy_(torch.tensor(n_text, dtype=torch.float64)); lb.copy_(torch.tensor(n_sub, dtype=torch.float64)); lm.copy_(torch.tensor(n_mel, dtype=torch.float64))
    out = StagedBatch(text=text.numpy(), mel_target=mel.numpy(), stop_token=stop.numpy(), bert_embeddings=sub.numpy(),
                      bert_embeddings_cls=bcls, phoneme_embeddings_cls=pcls, length_mel=lm.numpy(), length_text=lt.numpy(),
                      length_bert=lb.numpy())
    out.stage = stage
    if all("duration" in it for it in items):
        # alignloss != "": the reference's collate calls get_alignment(texts) against a (self, filename) signature and
        # raises; what it is after is the padded hard alignment [B, T_max, T_in_max] built from the duration column
        durs = torch.zeros(n, Tin, dtype=items[0]["duration"].dtype)
        for k, it in enumerate(items):
            durs[k, :n_text[k]] = it["duration"]
        al = create_alignment(torch.zeros(n, max(T, int(durs.sum(1).max())), Tin), durs)
        out["align"] = stage.take("align", (n, T, Tin), al.dtype).copy_(al[:, :T]).numpy()
    else:
        out["align"] = out["text"]
    return out


def collate_fn(batch):
    """data_utils.py:146-206: the loader batch holds batch_size**2 items; sorted by phone count (descending) and cut into
    batch_size model batches, returned as a list."""
    order = np.argsort(-np.array([d["text"].shape[0] for d in batch]))
    per_step = int(math.sqrt(len(batch)))
    return [_collate_cut(batch, order[i * per_step:(i + 1) * per_step], _RING.get(per_step)) for i in range(per_step)]


def batch_to_device(data_of_batch, device="cuda"):
    """train.py:295-316: one collate_fn batch -> the 10-tuple `BERT_Tacotron2.parse_batch` takes.  Each field goes up
    as collated (one non-blocking copy out of the batch's page-locked stage, on the current stream); dtype changes and
    the mel transpose happen on the device.  A batch that did not come out of collate_fn is staged here first."""
    device = torch.device(device)
    d = data_of_batch
    stage = getattr(d, "stage", None)
    own = stage is None
    if own:
        stage = _RING.get(1)

    def up(name, dtype):
        v = d[name]
        t = v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))
        if device.type != "cuda":
            return t.to(dtype)
        if not t.is_pinned():
            t = stage.take("_" + name, tuple(t.shape), t.dtype).copy_(t)
        return t.to(device, non_blocking=True).to(dtype)

    character = up("text", torch.long)
    mel_target = up("mel_target", torch.float32).contiguous().transpose(1, 2)
    stop_target = up("stop_token", torch.float32)
    embeddings = up("bert_embeddings", torch.long)
    phoneme_cls = up("phoneme_embeddings_cls", torch.float32)
    bert_cls = up("bert_embeddings_cls", torch.float32)
    il, ilb, ol = up("length_text", torch.long), up("length_bert", torch.long), up("length_mel", torch.long)
    align = character if d["align"] is d["text"] else up("align", torch.long)
    if device.type == "cuda":
        stage.event = torch.cuda.Event()
        stage.event.record(torch.cuda.current_stream(device))
    return character, il, ilb, mel_target, stop_target, ol, embeddings, phoneme_cls, bert_cls, align
