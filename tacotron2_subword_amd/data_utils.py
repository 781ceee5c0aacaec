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
from torch.nn import functional as F
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


def pad_normal(inputs, PAD=0):
    max_len = max((len(x) for x in inputs))
    return np.stack([np.pad(x, (0, max_len - x.shape[0]), mode="constant", constant_values=PAD) for x in inputs])


def pad_mel(inputs):
    def pad(x, max_len):
        if np.shape(x)[0] > max_len:
            raise ValueError("not max_len")
        s = np.shape(x)[1]
        return np.pad(x, (0, max_len - np.shape(x)[0]), mode="constant", constant_values=0)[:, :s]
    max_len = max(np.shape(x)[0] for x in inputs)
    return np.stack([pad(x, max_len) for x in inputs])


def pad_emb(inputs):
    max_len = max(x.size(0) for x in inputs)
    return torch.stack([F.pad(x, (0, 0, 0, max_len - x.size(0))) for x in inputs])


def get_alignment(filename):
    """Hard alignment [T, T_in] from the duration column of a phone file (data_utils.py:135-144)."""
    dur = torch.from_numpy(np.load(filename).astype(int))[:, 1].unsqueeze(0)
    alignment = torch.zeros(1, int(dur.sum()), dur.size(1))
    return create_alignment(alignment, dur)[0]


def reprocess(batch, cut_list):
    texts = [batch[ind]["text"] for ind in cut_list]
    bert_embeddings = [batch[ind]["bert_embedding"] for ind in cut_list]
    bert_embeddings_cls = [batch[ind]["bert_embedding_cls"] for ind in cut_list]
    phoneme_embeddings_cls = [batch[ind]["phoneme_embedding_cls"] for ind in cut_list]
    mel_targets = [batch[ind]["mel_target"] for ind in cut_list]
    stop_tokens = [batch[ind]["stop_token"] for ind in cut_list]
    length_text = np.array([float(t.shape[0]) for t in texts])
    length_bert = np.array([float(e.shape[0]) for e in bert_embeddings])
    length_mel = np.array([float(m.shape[0]) for m in mel_targets])
    have_dur = all("duration" in batch[ind] for ind in cut_list)
    texts = pad_normal(texts)
    out = {"text": texts, "mel_target": pad_mel(mel_targets), "stop_token": pad_normal(stop_tokens, PAD=1.),
           "bert_embeddings": pad_normal(bert_embeddings), "bert_embeddings_cls": pad_emb(bert_embeddings_cls),
           "phoneme_embeddings_cls": pad_emb(phoneme_embeddings_cls), "length_mel": length_mel, "length_text": length_text,
           "length_bert": length_bert}
    if have_dur:
        # alignloss != "": the reference's collate calls get_alignment(texts) against a (self, filename) signature and
        # raises; what it is after is the padded hard alignment [B, T_max, T_in_max] built from the duration column
        durs = pad_normal([batch[ind]["duration"].numpy() for ind in cut_list])
        T = out["mel_target"].shape[1]
        al = create_alignment(torch.zeros(len(cut_list), max(T, int(durs.sum(1).max())), durs.shape[1]), torch.from_numpy(durs))
        out["align"] = al[:, :T].numpy()
    else:
        out["align"] = texts
    return out


def collate_fn(batch):
    len_arr = np.array([d["text"].shape[0] for d in batch])
    index_arr = np.argsort(-len_arr)
    real_batchsize = int(math.sqrt(len(batch)))
    cut_list = [index_arr[i * real_batchsize:(i + 1) * real_batchsize] for i in range(real_batchsize)]
    return [reprocess(batch, cut_list[i]) for i in range(real_batchsize)]


class _Stager:
    """One pinned host buffer per field, grown on demand; copies are non-blocking on the current stream.  An event
    recorded after a batch's copies guards the buffers against being refilled while those copies are in flight."""

    def __init__(self):
        self.buf, self.event = {}, None

    def put(self, name, t, dtype, device):
        t = t if torch.is_tensor(t) else torch.from_numpy(np.ascontiguousarray(t))
        t = t.to(dtype)
        if device.type != "cuda":
            return t.to(device)
        b = self.buf.get(name)
        if b is None or b.numel() < t.numel() or b.dtype != dtype:
            b = torch.empty(max(t.numel(), 1), dtype=dtype).pin_memory()
            self.buf[name] = b
        v = b[:t.numel()].view(t.shape)
        v.copy_(t)
        return v.to(device, non_blocking=True)


_STAGER = _Stager()


def batch_to_device(data_of_batch, device="cuda"):
    """train.py:295-316: one collate_fn batch -> the 10-tuple `BERT_Tacotron2.parse_batch` takes."""
    device = torch.device(device)
    if _STAGER.event is not None:
        _STAGER.event.synchronize()                      # the previous batch's H2D copies have left the pinned buffers
    d, put = data_of_batch, _STAGER.put
    character = put("text", d["text"], torch.long, device)
    mel_target = put("mel", d["mel_target"], torch.float32, device).contiguous().transpose(1, 2)
    stop_target = put("stop", d["stop_token"], torch.float32, device)
    embeddings = put("sub", d["bert_embeddings"], torch.long, device)
    phoneme_cls = put("pcls", d["phoneme_embeddings_cls"], torch.float32, device)
    bert_cls = put("bcls", d["bert_embeddings_cls"], torch.float32, device)
    il = put("il", d["length_text"], torch.long, device)
    ilb = put("ilb", d["length_bert"], torch.long, device)
    ol = put("ol", d["length_mel"], torch.long, device)
    align = put("align", d["align"], torch.long, device)
    if device.type == "cuda":
        _STAGER.event = torch.cuda.Event()
        _STAGER.event.record(torch.cuda.current_stream(device))
    return character, il, ilb, mel_target, stop_target, ol, embeddings, phoneme_cls, bert_cls, align
