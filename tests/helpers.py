"""Shared test helpers: golden-vector loading and seeded case generation."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def reference_cases(layouts=("dense", "gathered")):
    doc = json.load(open(os.path.join(GOLDEN, "reference_vectors.json")))
    return [c for c in doc["cases"] if c["layout"] in layouts]


def reference_doc():
    return json.load(open(os.path.join(GOLDEN, "reference_vectors.json")))


def np_log_softmax32(x):
    """fp32 log-softmax with torch's association ((x-max)-log(sum))."""
    x = np.asarray(x, dtype=np.float32)
    m = x.max(axis=-1, keepdims=True)
    s = np.exp(x - m, dtype=np.float32).sum(axis=-1, keepdims=True, dtype=np.float32)
    return ((x - m) - np.log(s, dtype=np.float32)).astype(np.float32)


def make_case(seed, N, T, U, V, ragged=False, blank=0):
    """Seeded synthetic case following pytorch_binding/benchmark.py:9-28 (N(0,1)
    logits, labels never blank, full lengths or the reference's ragged rule)."""
    rng = np.random.RandomState(seed)
    logits = rng.randn(N, T, U, V).astype(np.float32)
    choices = np.array([v for v in range(V) if v != blank], dtype=np.int32)
    labels = choices[rng.randint(0, len(choices), size=(N, max(U - 1, 0)))].astype(np.int32)
    if ragged:
        xn = rng.randint(max(T // 2, 1), T + 1, size=(N,)).astype(np.int32)
        yn = rng.randint(U // 2, U, size=(N,)).astype(np.int32) if U > 1 else np.zeros((N,), np.int32)
        xn = (xn + T - xn.max()).astype(np.int32)
        yn = (yn + (U - 1) - yn.max()).astype(np.int32)
    else:
        xn = np.full((N,), T, dtype=np.int32)
        yn = np.full((N,), U - 1, dtype=np.int32)
    return logits, labels, xn, yn
