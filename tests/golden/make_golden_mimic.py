#!/usr/bin/env python3
"""f19_mimic.npz: expert-demonstration data sets in the reference's on-disk format + what the reference reads back.

The reference's own writers (USTC_lab/data/mimic_exp.py:17-140, IMPORTED) write a classical (observations + .npy
labels) and an atari (one frame file per stacked frame, action as text) data set into temp dirs; the files are stored in
the fixture as name -> bytes so that the test can re-materialise the directories; the reference's readers + the
DataLoader(shuffle=True) the discriminator wraps around them (GAIL.py:49-57) give the expected samples and the first
batches under torch.manual_seed.  Usage: python tests/golden/make_golden_mimic.py"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
REF = os.environ.get("DDRL_REFERENCE", "/root/reference")


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, REF)
    from torch.utils.data import DataLoader
    from USTC_lab.data import MimicExpFactory
    out = {}
    rng = np.random.default_rng(19)
    for kind in ("classical", "atari"):
        d = tempfile.mkdtemp(prefix="ddrl_mimic_%s_" % kind) + "/"
        w = MimicExpFactory().mimic_writer(kind, "golden", d, 2, 1 if kind == "classical" else 4)
        for step in range(5):
            for proc in range(2):
                if kind == "classical":
                    w.put(rng.normal(size=(3, 4)).astype(np.float32), rng.integers(0, 2, size=(3, 1)).astype(np.float32), proc)
                else:
                    w.put(rng.integers(0, 256, size=(3, 4, 12, 12)).astype(np.uint8), rng.integers(0, 6, size=3), proc)
        w.write()
        w.sf.close()
        files = sorted(os.listdir(d))
        out[kind + "/files"] = np.array(files)
        for f in files:
            out["%s/file/%s" % (kind, f)] = np.frombuffer(open(os.path.join(d, f), "rb").read(), np.uint8)
        ds = MimicExpFactory().mimic_reader(kind, d, torch.float32, "cpu")
        out[kind + "/len"] = np.int64(len(ds))
        xs, ys = zip(*[ds[i] for i in range(len(ds))])
        out[kind + "/x"] = np.stack(xs)
        out[kind + "/y"] = np.stack([np.asarray(y, np.float32) for y in ys])
        torch.manual_seed(190)
        dl = DataLoader(ds, batch_size=8, shuffle=True)
        for epoch in range(2):
            for bi, batch in enumerate(dl):
                out["%s/epoch%d/batch%d/x" % (kind, epoch, bi)] = batch[0].numpy()
                out["%s/epoch%d/batch%d/y" % (kind, epoch, bi)] = batch[1].numpy()
                if bi == 1:
                    break
    np.savez_compressed(os.path.join(HERE, "f19_mimic.npz"), **out)
    print("f19_mimic.npz", os.path.getsize(os.path.join(HERE, "f19_mimic.npz")), "B;", {k: int(out[k + "/len"]) for k in ("classical", "atari")})


if __name__ == "__main__":
    main()
