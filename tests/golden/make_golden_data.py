#!/usr/bin/env python3
"""Golden vectors for the input pipeline (data_utils.py:47-206): a small synthetic dataset in the reference's on-disk
formats is written to a temp directory, read by the REFERENCE's BERTTacotron2Dataset and collate_fn, and the collated
batches are saved to data_collate.npz.  tests/test_data_utils.py rebuilds the same files from the same seed."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def write_dataset(root, n_items=9, seed=77):
    """Shared with the test: deterministic synthetic items in the reference's file formats."""
    g = np.random.Generator(np.random.PCG64(seed))
    split, emb, cls, txt = os.path.join(root, "dataset", "train"), os.path.join(root, "emb"), os.path.join(root, "cls"), os.path.join(root, "txt")
    for d in (split, emb, cls, txt):
        os.makedirs(d, exist_ok=True)
    lines = []
    for i in range(n_items):
        tin, tsub = int(g.integers(4, 12)), int(g.integers(3, 9))
        dur = g.integers(1, 4, size=tin)
        np.save(os.path.join(split, "ljspeech-mel-%05d.npy" % (i + 1)), g.normal(-5, 2, size=(int(dur.sum()), 80)).astype(np.float32))
        p = os.path.join(txt, f"{i}.npy")
        np.save(p, np.stack([g.integers(1, 313, size=tin), dur], 1))
        lines.append(p)
        np.save(os.path.join(emb, f"{i}.npy"), g.integers(1, 5500, size=tsub).astype(np.int32))
        np.save(os.path.join(cls, f"{i}.npy"), g.normal(0, 1, size=(1, 768)).astype(np.float32))
    listing = os.path.join(root, "train.txt")
    with open(listing, "w") as f:
        f.write("\n".join(lines) + "\n")
    return listing, emb, cls


def main():
    import_reference()
    import data_utils as rd  # the reference's (sys.path has /root/reference)
    with tempfile.TemporaryDirectory() as root:
        listing, emb, cls = write_dataset(root)
        cwd = os.getcwd()
        os.chdir(root)                                   # the reference hard-codes os.path.join("dataset", split)
        try:
            ds = rd.BERTTacotron2Dataset("train", listing, emb, cls)
            items = [ds[i] for i in range(len(ds))]
            batches = rd.collate_fn(items)               # 9 items -> 3 batches of 3
        finally:
            os.chdir(cwd)
    out = {"n_batches": np.array(len(batches))}
    for i, b in enumerate(batches):
        for k, v in b.items():
            out[f"b{i}_{k}"] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "data_collate.npz"), **out)
    print("wrote data_collate.npz:", len(batches), "batches", {k: out[k].shape for k in out if k.startswith("b0_")})


if __name__ == "__main__":
    main()
