"""Helpers shared by the golden-vector tests (CPU oracle and GPU engine are checked by the same comparisons)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IF_HZ = 4092000


def fnv1a32(buf) -> int:
    h = 0x811C9DC5
    for b in np.asarray(buf).view(np.uint8).reshape(-1).tolist():
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return h


def load(name):
    """Load an .npz fully into a dict (NpzFile would re-decompress an array on every [] access)."""
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)
