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
                  "bert_embedding_cls": embedding_cls.expand(bert_embedding.size(0), -1),
                  "phoneme_embedding_cls": embedding_cls.expand(phoneme.size(0), -1), "stop_token": stop_token}
        # (the reference materialises the CLS rows with .repeat(); the stride-0 views hold the same values, cost nothing per
        #  item and let collate_fn see that the rows of an item are one vector: it then ships [B, 768] instead of [B, T, 768])
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
# process has a CUDA context, written once per batch by the loops below — no per-item pad / stack temporaries — and
# handed to the GPU by batch_to_device with one non-blocking copy per field.  A ring of stages keeps the batches of two
# loader steps apart, so batch k+1 is collated and uploaded while step k runs; each stage carries the event recorded
# behind its last upload and is refilled only after that event.
#
# The ring exists only in the process that trains.  Inside a DataLoader worker (train.py:236-240 runs 8-16 of them) a
# collated batch is pickled to the parent, which moves tensor storage into shared memory IN PLACE: a reused buffer would
# be refilled by the worker while the parent still holds the batch made from it two loader steps earlier.  Workers
# therefore collate into fresh tensors and attach no stage; batch_to_device stages such a batch itself.
#
# CLS vectors: an item's phoneme_embedding_cls / bert_embedding_cls is ONE BERT vector repeated along time
# (data_utils.py:76-79 of the reference).  When collate sees that (stride-0 rows, or a single row) it keeps [B, 768]
# rows plus the item lengths; batch_to_device uploads those (0.4 MB instead of 31 MB at B=64) and expands them on the
# device with the reference's zero padding.  The dict still answers d["phoneme_embeddings_cls"] with the full
# [B, T, 768] tensor (built on first access), so the reference's own train loop reads what it expects.
# ---------------------------------------------------------------------------------------------------------------
class _Stage:
    def __init__(self, pinned=True):
        self.buf, self.event, self.pinned = {}, None, pinned

    def wait(self):
        if self.event is not None:
            self.event.synchronize()
            self.event = None

    def take(self, name, shape, dtype, fill=None):
        n = 1
        for d in shape:
            n *= int(d)
        b = self.buf.get(name) if self.pinned else None           # (an unpinned stage never reuses: see above)
        if b is None or b.numel() < n or b.dtype != dtype:
            b = torch.empty(max(n, 1), dtype=dtype)
            if self.pinned and torch.cuda.is_available() and torch.cuda.is_initialized():
                b = b.pin_memory()
            if self.pinned:
                self.buf[name] = b
        v = b[:n].view(*shape)
        if fill is not None:
            v.numpy().fill(fill)         # numpy on purpose: one thread (see utils.fit_cpu_threads for what torch's pool costs here)
        return v


_CLS_KEYS = {"phoneme_embeddings_cls": "length_text", "bert_embeddings_cls": "length_bert"}


class StagedBatch(dict):
    """A collated model batch (the reference's dict keys).  `stage`: the page-locked stage its arrays live in (None for a
    batch made in a DataLoader worker).  `cls_rows`: {key: [B, 768] tensor} for CLS fields kept as one row per item; the
    dict entry of such a key is None until somebody asks for it and is then the full zero-padded [B, T, 768] tensor.
    `host_max`: (max input length over both streams, max output length), what parse_batch would otherwise .item()."""
    stage = None
    cls_rows = None
    host_max = None

    def _full(self, key):
        rows = self.cls_rows[key]
        lens = np.asarray(dict.__getitem__(self, _CLS_KEYS[key])).astype(np.int64)
        T = int(dict.__getitem__(self, "text" if key == "phoneme_embeddings_cls" else "bert_embeddings").shape[1])
        full = torch.zeros(rows.shape[0], T, rows.shape[1], dtype=rows.dtype)
        for k, n in enumerate(lens):
            full[k, :int(n)] = rows[k]
        return full

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if v is None and self.cls_rows and key in self.cls_rows:
            v = self._full(key)
            dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __reduce__(self):            # (the default walks items(), which would materialise the CLS tensors for the trip to the parent)
        return (_rebuild_staged, (dict(dict.items(self)), self.cls_rows, self.host_max))

    def items(self):
        return [(k, self[k]) for k in self]

    def values(self):
        return [self[k] for k in self]


def _rebuild_staged(raw, cls_rows, host_max):
    out = StagedBatch(raw)
    out.cls_rows, out.host_max = cls_rows, host_max
    return out


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


def _in_worker():
    info = torch.utils.data.get_worker_info()
    return info is not None


def _one_row(x):
    """True when every time row of x [T, C] is the same vector by construction: a stride-0 expand or a single row."""
    return x.dim() == 2 and (x.shape[0] == 1 or x.stride(0) == 0)


def collate_batch(items, stage=None):
    """One model batch (data_utils.py:98-133) from a list of dataset items, every field written in place: zero (or
    pad-value) fill, then one slice assignment per item.  Dtypes and values are those of the reference's pad_normal /
    pad_mel / pad_emb.  stage: a page-locked _Stage of the ring (in-process collation), None = fresh pageable tensors."""
    own = stage is None
    if own:
        stage = _Stage(pinned=False)
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
    compact = all(_one_row(it["phoneme_embedding_cls"]) and _one_row(it["bert_embedding_cls"]) for it in items)
    if compact:
        pcls = stage.take("pcls", (n, C), items[0]["phoneme_embedding_cls"].dtype)
        bcls = stage.take("bcls", (n, C), items[0]["bert_embedding_cls"].dtype)
    else:
        pcls = stage.take("pcls", (n, Tin, C), items[0]["phoneme_embedding_cls"].dtype, 0)
        bcls = stage.take("bcls", (n, Tsub, C), items[0]["bert_embedding_cls"].dtype, 0)
    # numpy views of the stage buffers: plain single-threaded copies (torch's CPU ops would wake its whole intra-op pool)
    tn, sn, mn, stn, pn, bn = (t.numpy() for t in (text, sub, mel, stop, pcls, bcls))
    as_np = lambda v: v.numpy() if torch.is_tensor(v) else np.asarray(v)
    for k, it in enumerate(items):
        tn[k, :n_text[k]] = as_np(it["text"])
        sn[k, :n_sub[k]] = as_np(it["bert_embedding"])
        mn[k, :n_mel[k]] = it["mel_target"]
        stn[k, :n_mel[k]] = it["stop_token"]
        if compact:
            pn[k] = as_np(it["phoneme_embedding_cls"][0])
            bn[k] = as_np(it["bert_embedding_cls"][0])
        else:
            pn[k, :n_text[k]] = as_np(it["phoneme_embedding_cls"])
            bn[k, :n_sub[k]] = as_np(it["bert_embedding_cls"])
    lt = stage.take("length_text", (n,), torch.float64)
    lb = stage.take("length_bert", (n,), torch.float64)
    lm = stage.take("length_mel", (n,), torch.float64)
    lt.numpy()[:] = n_text; lb.numpy()[:] = n_sub; lm.numpy()[:] = n_mel
    out = StagedBatch(text=text.numpy(), mel_target=mel.numpy(), stop_token=stop.numpy(), bert_embeddings=sub.numpy(),
                      bert_embeddings_cls=None if compact else bcls, phoneme_embeddings_cls=None if compact else pcls,
                      length_mel=lm.numpy(), length_text=lt.numpy(), length_bert=lb.numpy())
    if compact:
        out.cls_rows = {"phoneme_embeddings_cls": pcls, "bert_embeddings_cls": bcls}
    out.host_max = (max(Tin, Tsub), T)
    if not own:
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
        dict.__setitem__(out, "align", dict.__getitem__(out, "text"))
    return out


def collate_fn(batch):
    """data_utils.py:146-206: the loader batch holds batch_size**2 items; sorted by phone count (descending) and cut into
    batch_size model batches, returned as a list."""
    order = np.argsort(-np.array([d["text"].shape[0] for d in batch]))
    per_step = int(math.sqrt(len(batch)))
    ring = None if _in_worker() else _RING
    return [collate_batch([batch[int(i)] for i in order[i * per_step:(i + 1) * per_step]], None if ring is None else ring.get(per_step))
            for i in range(per_step)]


class DeviceBatch(tuple):
    """The 10-tuple `BERT_Tacotron2.parse_batch` takes, plus what the host already knows about it: `host_max` =
    (max input length over both streams, max output length), so that parse_batch need not ask the device."""
    host_max = None


def batch_to_device(data_of_batch, device="cuda"):
    """train.py:295-316: one collate_fn batch -> the 10-tuple `BERT_Tacotron2.parse_batch` takes.  Each field goes up
    as collated (one non-blocking copy out of the batch's page-locked stage, on the current stream); dtype changes, the
    mel transpose and the expansion of per-item CLS rows happen on the device.  A batch that did not come out of an
    in-process collate_fn (a DataLoader worker's, a hand-made dict) is staged here first.  No host synchronisation."""
    device = torch.device(device)
    d = data_of_batch
    raw = (lambda k: dict.__getitem__(d, k)) if isinstance(d, dict) else (lambda k: d[k])
    stage = getattr(d, "stage", None)
    if stage is None and device.type == "cuda":
        stage = _RING.get(1)

    def up(v, name, dtype):
        t = v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v))
        if device.type != "cuda":
            return t.to(dtype)
        if not t.is_pinned():
            st = stage.take("_" + name, tuple(t.shape), t.dtype)
            np.copyto(st.numpy(), t.numpy() if t.is_contiguous() else t.contiguous().numpy())       # (one thread: see collate_batch)
            t = st
        return t.to(device, non_blocking=True).to(dtype)

    character = up(raw("text"), "text", torch.long)
    mel_target = up(raw("mel_target"), "mel_target", torch.float32).contiguous().transpose(1, 2)
    stop_target = up(raw("stop_token"), "stop_token", torch.float32)
    embeddings = up(raw("bert_embeddings"), "bert_embeddings", torch.long)
    il, ilb, ol = (up(raw(k), k, torch.long) for k in ("length_text", "length_bert", "length_mel"))
    rows = getattr(d, "cls_rows", None) or {}

    def cls(key, lengths, T):
        if key not in rows or raw(key) is not None:
            return up(d[key], key, torch.float32)
        r = up(rows[key], key, torch.float32)                            # [B, C] -> [B, T, C], zero past each item's length
        live = torch.arange(T, device=r.device)[None, :] < lengths[:, None]
        return r[:, None, :] * live[:, :, None].to(r.dtype)

    phoneme_cls = cls("phoneme_embeddings_cls", il, character.shape[1])
    bert_cls = cls("bert_embeddings_cls", ilb, embeddings.shape[1])
    align = character if raw("align") is raw("text") else up(raw("align"), "align", torch.long)
    if device.type == "cuda":
        stage.event = torch.cuda.Event()
        stage.event.record(torch.cuda.current_stream(device))
    out = DeviceBatch((character, il, ilb, mel_target, stop_target, ol, embeddings, phoneme_cls, bert_cls, align))
    hm = getattr(d, "host_max", None)
    if hm is None:                                                       # lengths are host arrays here: no device round trip
        hm = (int(max(np.max(np.asarray(raw("length_text"))), np.max(np.asarray(raw("length_bert"))))), int(np.max(np.asarray(raw("length_mel")))))
    out.host_max = (int(hm[0]), int(hm[1]))
    return out
