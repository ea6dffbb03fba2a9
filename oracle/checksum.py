"""TEST ORACLE (numpy) of the parameter-guard checksum, vatl_checksum_multi (include/vatl_hip.h, csrc/checksum.hip).

The reference has no counterpart: it runs torch modules directly, so a parameter written through `.data`
(alphapose/models/layers/dcn/deform_conv.py:232,255) is simply seen by the next forward.  The HIP path packs
weights into plans, and this checksum is how it notices such writes; the oracle restates the published formula
so the GPU test can compare the kernel bit for bit.  Only tests import this file.
"""
from __future__ import annotations

import numpy as np

K = np.uint64(0x9E3779B97F4A7C15)


def checksum_words(words: np.ndarray) -> int:
    """sum_i (w_i + 1) * (K + 2 i)  mod 2^64 over the 32-bit words of a tensor."""
    w = np.ascontiguousarray(words).view(np.uint32).reshape(-1)
    i = np.arange(w.size, dtype=np.uint64)
    with np.errstate(over="ignore"):
        terms = (w.astype(np.uint64) + np.uint64(1)) * (K + np.uint64(2) * i)
        return int(terms.sum(dtype=np.uint64))


def checksum_tensor(t) -> int:
    """Same for a torch CPU tensor / numpy array of any dtype whose byte size is a multiple of 4."""
    a = t.detach().cpu().contiguous().numpy() if hasattr(t, "detach") else np.ascontiguousarray(t)
    return checksum_words(a.reshape(-1).view(np.uint8).view(np.uint32))
