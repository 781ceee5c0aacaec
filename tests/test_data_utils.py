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
