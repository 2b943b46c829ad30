#!/usr/bin/env python
"""Seeded, format-identical stand-in for the reference's missing
rawdata-classification/DataSet.tar.gz (`.MISSING_LARGE_BLOBS:2`; README.md:98-108:
571 classes, `TrainPairs`/`EvalPairs` = `title \\t classId`, `targetIDs` =
`category path \\t classId`).  SURVEY 8d C1: category-path targets, titles of
5-12 words from a Zipf vocabulary, one label each.

    python tools/make_standin_dataset.py --out rawdata-classification [--targets 571 --train 20000 --eval 2000]
"""
import argparse
import io
import os
import tarfile

import numpy as np

SYL = ["ka", "to", "mi", "ra", "ne", "so", "lu", "vi", "pe", "da", "gor", "an", "el", "ix", "ur", "bo", "shi", "qua", "zen", "fy"]


def _words(rng, n):
    seen, out = set(), []
    while len(out) < n:
        w = "".join(rng.choice(SYL, size=rng.randint(2, 4)))
        if w not in seen:
            seen.add(w)
            out.append(w)
    return out


def generate(n_targets=571, n_train=20000, n_eval=2000, n_vocab=5000, seed=0):
    rng = np.random.RandomState(seed)
    vocab = _words(rng, n_vocab)
    zipf = 1.0 / np.arange(1, n_vocab + 1)
    zipf /= zipf.sum()
    tops = _words(rng, 12)
    targets, keywords = [], []
    for c in range(n_targets):
        depth = rng.randint(2, 5)
        path = [tops[c % len(tops)]] + [vocab[rng.randint(50, n_vocab)] for _ in range(depth - 1)]
        targets.append((":".join(w.capitalize() for w in path), "c%04d" % c))
        keywords.append(path[1:] + [vocab[rng.randint(50, n_vocab)] for _ in range(3)])

    def title(c):
        n = rng.randint(5, 13)
        k = keywords[c]
        words = [k[rng.randint(0, len(k))] for _ in range(max(2, n // 2))]
        words += [vocab[i] for i in rng.choice(n_vocab, size=n - len(words), p=zipf)]
        rng.shuffle(words)
        return " ".join(words).capitalize()

    def pairs(n):
        cls = rng.randint(0, n_targets, size=n)
        return ["%s\t%s" % (title(c), targets[c][1]) for c in cls]

    return {"TrainPairs": pairs(n_train), "EvalPairs": pairs(n_eval),
            "targetIDs": ["%s\t%s" % t for t in targets]}


def write_tar(files, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "DataSet.tar.gz")
    with tarfile.open(path, "w:gz") as tar:
        for name, lines in files.items():
            data = ("\n".join(lines) + "\n").encode("utf-8")
            info = tarfile.TarInfo(name)
            info.size = len(data)
            tar.addfile(info, io.BytesIO(data))
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="rawdata-classification")
    ap.add_argument("--targets", type=int, default=571)
    ap.add_argument("--train", type=int, default=20000)
    ap.add_argument("--eval", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    print(write_tar(generate(a.targets, a.train, a.eval, seed=a.seed), a.out))


if __name__ == "__main__":
    main()
