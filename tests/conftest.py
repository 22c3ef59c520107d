import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def known():
    with open(os.path.join(GOLDEN, "known_answers.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def trec():
    """The reference's bundled example (examples/trec_news_2018.{train,test}) as arrays."""
    d = np.load(os.path.join(GOLDEN, "trec_news_2018.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def qrel_dict():
    with open(os.path.join(GOLDEN, "newsir18_entity_qrel.json")) as fh:
        return json.load(fh)


def synth_dataset(seed, n, d, q, max_len=None):
    """Small-scale twin of bench.py's MSLR-shaped generator:
    integer-heavy columns (ties), heavy tails, sparse columns, label signal."""
    rng = np.random.default_rng(seed)
    lens = np.clip(rng.lognormal(np.log(max(n / q, 1.0)), 0.6, q), 1, max_len or 1300)
    lens = np.maximum(1, np.floor(lens * (n / lens.sum()))).astype(np.int64)
    diff = n - lens.sum()
    i = 0
    while diff != 0:
        k = i % q
        if diff > 0:
            lens[k] += 1
            diff -= 1
        elif lens[k] > 1:
            lens[k] -= 1
            diff += 1
        i += 1
    qid = np.repeat(np.arange(1, q + 1, dtype=np.int64), lens)
    y = rng.choice(5, size=n, p=[0.515, 0.324, 0.134, 0.019, 0.008]).astype(np.float64)
    X = np.empty((n, d), dtype=np.float32)
    for j in range(d):
        m = j % 4
        if m == 0:
            col = rng.random(n)
        elif m == 1:
            col = np.floor(rng.exponential(2.0, n))
        elif m == 2:
            col = rng.lognormal(0.0, 2.0, n)
        else:
            col = np.where(rng.random(n) < 0.7, 0.0, rng.random(n))
        if j % 8 == 0:
            col = col + 0.3 * y
        X[:, j] = col.astype(np.float32)
    return X, y, qid


def ranksvm_presence(path, d):
    """present[i, f] for a ranksvm / libsvm file as the reference holds it (src/instance.rs:104-130): a row whose listed
    features cover at least half of 1..max index becomes Dense32 of length max + 1 and HOLDS every index below that
    (unlisted ones as 0.0); a sparser row holds exactly what it lists.  Independent of the product's loader."""
    rows = []
    with open(path) as fh:
        for line in fh:
            data = line.split("#", 1)[0].split()
            if not data:
                continue
            ids = [int(tok.split(":", 1)[0]) for tok in data[1:] if not tok.startswith("qid:")]
            mx = max(ids) if ids else 1
            held = np.zeros(d, dtype=bool)
            if len(ids) / mx >= 0.5:
                held[: mx + 1] = True
            else:
                held[ids] = True
            rows.append(held)
    return np.array(rows)
