"""Input pipeline against batches collated by the reference's own data_utils.py from the same
synthetic on-disk dataset (tests/golden/make_golden_data.py writes the files from a seed; so does this test)."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)


def _collated(tmp_path, alignloss=""):
    from make_golden_data import write_dataset
    from tacotron2_subword_amd import data_utils as D
    listing, emb, cls = write_dataset(str(tmp_path))
    ds = D.BERTTacotron2Dataset("train", listing, emb, cls, dataset_root=os.path.join(str(tmp_path), "dataset"), alignloss=alignloss)
    assert len(ds) == 9
    return D, D.collate_fn([ds[i] for i in range(len(ds))])


def test_dataset_and_collate_match_the_reference(tmp_path):
    g = load_golden("data_collate")
    D, batches = _collated(tmp_path)
    assert len(batches) == int(g["n_batches"]) == 3                  # sqrt(9) model batches per loader step
    for i, b in enumerate(batches):
        assert set(b) == {k[len(f"b{i}_"):] for k in g.files if k.startswith(f"b{i}_")}
        for k, v in b.items():
            ref = g[f"b{i}_{k}"]
            got = v.numpy() if torch.is_tensor(v) else np.asarray(v)
            assert got.shape == ref.shape and got.dtype == ref.dtype, (i, k, got.dtype, ref.dtype)
            assert np.array_equal(got, ref), (i, k)


def test_batch_tuple_feeds_parse_batch_layout(tmp_path):
    D, batches = _collated(tmp_path)
    t = D.batch_to_device(batches[0], "cpu")
    text, il, ilb, mel, gate, ol, sub, pcls, bcls, align = t
    B, Tin = text.shape
    assert mel.shape[:2] == (B, 80) and gate.shape == (B, mel.shape[2]) and sub.shape[0] == B
    assert pcls.shape == (B, Tin, 768) and bcls.shape == (B, sub.shape[1], 768)
    assert text.dtype == torch.long and mel.dtype == torch.float32 and il.dtype == torch.long
    assert int(il.max()) == Tin and int(ol.max()) == mel.shape[2]
    assert torch.all(gate[torch.arange(B), ol - 1] == 1) and torch.all(gate.sum(1) >= 1)      # 1 from the last real frame on (pad = 1)
    assert torch.equal(align, text)                                  # alignloss == "": the reference passes the texts through


def test_collate_with_alignment_targets(tmp_path):
    """alignloss != "": hard alignments from the duration column (the reference's own call raises a TypeError here)."""
    D, batches = _collated(tmp_path, alignloss="L2")
    b = batches[0]
    al = b["align"]
    B, T, Tin = al.shape
    assert (B, T, Tin) == (b["text"].shape[0], b["mel_target"].shape[1], b["text"].shape[1])
    for i in range(B):
        n = int(b["length_mel"][i])
        assert np.all(al[i, :n].sum(1) == 1) and np.all(al[i, n:] == 0)          # one phone per real frame, none on padding
        assert np.all(np.diff(al[i, :n].argmax(1)) >= 0)                          # monotonic


@pytest.mark.gpu
def test_pinned_staging_to_gpu(tmp_path):
    D, batches = _collated(tmp_path)
    cpu = [D.batch_to_device(b, "cpu") for b in batches]
    for b, want in zip(batches, cpu):                                 # buffers are reused batch after batch
        got = D.batch_to_device(b, "cuda")
        torch.cuda.synchronize()
        for a, w in zip(got, want):
            assert a.is_cuda and torch.equal(a.cpu(), w)


def _check_cls_rows_follow_their_texts(batches, root, n_items):
    """Every item of every collated batch carries the CLS vector of ITS utterance (identified by its phone ids)."""
    by_text = {}
    for i in range(n_items):
        ids = np.load(os.path.join(root, "txt", f"{i}.npy"))[:, 0]
        by_text[tuple(int(v) for v in ids)] = np.load(os.path.join(root, "cls", f"{i}.npy"))[0]
    seen = 0
    for b in batches:
        text, lt, lb = b["text"], b["length_text"].astype(int), b["length_bert"].astype(int)
        pcls, bcls = b["phoneme_embeddings_cls"], b["bert_embeddings_cls"]
        assert pcls.shape == (text.shape[0], text.shape[1], 768) and bcls.shape[:2] == b["bert_embeddings"].shape
        for k in range(text.shape[0]):
            want = by_text[tuple(int(v) for v in text[k, :lt[k]])]
            assert np.array_equal(pcls[k, :lt[k]].numpy(), np.broadcast_to(want, (lt[k], 768))), "phoneme CLS rows of another item"
            assert np.array_equal(bcls[k, :lb[k]].numpy(), np.broadcast_to(want, (lb[k], 768))), "sub-word CLS rows of another item"
            assert not pcls[k, lt[k]:].any() and not bcls[k, lb[k]:].any()                  # zero padding (pad_emb)
            seen += 1
    return seen


@pytest.mark.parametrize("workers", [0, 1, 2])
def test_dataloader_workers_keep_items_apart(tmp_path, workers):
    """train.py:236-240 runs the collate in DataLoader workers.  A worker's batch reaches the parent through shared
    memory, so nothing it is made of may be reused by the worker afterwards: all batches of several loader steps are held
    here while the workers keep producing, and only then checked (with a reused staging ring the CLS rows
    of a later batch showed up under an earlier batch's texts)."""
    from make_golden_data import write_dataset
    from tacotron2_subword_amd import data_utils as D
    n_items = 45
    listing, emb, cls = write_dataset(str(tmp_path), n_items=n_items, seed=5)
    ds = D.BERTTacotron2Dataset("train", listing, emb, cls, dataset_root=os.path.join(str(tmp_path), "dataset"))
    loader = torch.utils.data.DataLoader(ds, batch_size=9, shuffle=False, collate_fn=D.collate_fn, drop_last=True, num_workers=workers)
    if workers:
        held = [b for step in loader for b in step]
        assert all(b.stage is None for b in held)                                           # no ring stage crosses a process boundary
        assert all(b.cls_rows is not None and dict.__getitem__(b, "phoneme_embeddings_cls") is None for b in held)   # [B, 768] rows travel, not [B, T, 768]
        assert _check_cls_rows_follow_their_texts(held, str(tmp_path), n_items) == n_items
    else:
        # in-process collation writes into the page-locked ring: a batch is valid until the loader step after the next
        # (the training loop has long uploaded it by then), so the batches are checked as they arrive
        held, seen, prev = [], 0, []
        for step in loader:
            assert all(b.stage is not None for b in step)
            seen += _check_cls_rows_follow_their_texts(prev + step, str(tmp_path), n_items) - 3 * len(prev)
            prev = step
            held += step
        assert seen == n_items
    assert len(held) == 15 and all(b["text"].shape[0] == 3 for b in held)
    tup = D.batch_to_device(held[-1], "cpu")
    assert tup.host_max == (max(held[-1]["text"].shape[1], held[-1]["bert_embeddings"].shape[1]), held[-1]["mel_target"].shape[1])
    assert torch.equal(tup[7], held[-1]["phoneme_embeddings_cls"]) and torch.equal(tup[8], held[-1]["bert_embeddings_cls"])


def test_collate_keeps_full_cls_tensors_when_rows_differ(tmp_path):
    """Items whose CLS rows are not one repeated vector (a caller's own Dataset) are collated as full [B, T, 768] tensors."""
    D, _ = _collated(tmp_path)
    g = torch.Generator().manual_seed(0)
    items = []
    for n_text, n_sub, n_mel in ((5, 4, 11), (3, 6, 9)):
        items.append({"text": torch.randint(1, 300, (n_text,), generator=g), "mel_target": np.random.default_rng(n_mel).normal(size=(n_mel, 80)).astype(np.float32),
                      "bert_embedding": torch.randint(1, 5000, (n_sub,), generator=g).int(), "stop_token": np.zeros(n_mel),
                      "bert_embedding_cls": torch.randn(n_sub, 768, generator=g), "phoneme_embedding_cls": torch.randn(n_text, 768, generator=g)})
    b = D.collate_batch(items)
    assert b.cls_rows is None and b["phoneme_embeddings_cls"].shape == (2, 5, 768)
    for k, it in enumerate(items):
        assert torch.equal(b["phoneme_embeddings_cls"][k, :it["text"].shape[0]], it["phoneme_embedding_cls"])
        assert torch.equal(b["bert_embeddings_cls"][k, :it["bert_embedding"].shape[0]], it["bert_embedding_cls"])
    t = D.batch_to_device(b, "cpu")
    assert torch.equal(t[7], b["phoneme_embeddings_cls"]) and t.host_max == (6, 11)
