"""Hadamard factors for non-power-of-two dimensions.

The reference ships literal +-1 tables (`get_had12 .. get_had172`,
third-party/QuaRot/quarot/functional/hadamard.py:175-3010) and picks one with
`get_hadK(n)` (:6-56).  A real QSpec checkpoint carries the matrix it was
rotated with (`...mlp.online_hadamard.had_rem_dim`, vllm/worker/model_runner.py:1132-1141),
so at run time the table comes from the checkpoint.  For synthetic weights any
Hadamard matrix of the right order is equivalent; this module CONSTRUCTS them
(Paley I / Paley II / Sylvester doubling) instead of carrying tables, with the
same `get_hadK` selection rule.  Orders covered: 12, 20, 28, 36, 40, 44, 60,
108, 140 (the configs need 28 for Llama-3-8B/70B, 108 for Llama-2-13B, 44 for
TinyLlama-1.1B whose 5632 = 44*128 has no table in the reference).
"""
from __future__ import annotations

import functools

import numpy as np
import torch


def is_pow2(n: int) -> bool:
    return n > 0 and (n & (n - 1)) == 0


def _jacobsthal(q: int) -> np.ndarray:
    """Q[i,j] = chi(i - j) over GF(q), q an odd prime."""
    residues = {(x * x) % q for x in range(1, q)}
    chi = np.zeros(q, dtype=np.int64)
    for a in range(1, q):
        chi[a] = 1 if a in residues else -1
    idx = (np.arange(q)[:, None] - np.arange(q)[None, :]) % q
    return chi[idx]


def _is_prime(q: int) -> bool:
    return q > 1 and all(q % p for p in range(2, int(q ** 0.5) + 1))


def paley1(q: int) -> np.ndarray:
    """Order q+1, q prime, q = 3 (mod 4)."""
    assert _is_prime(q) and q % 4 == 3
    n = q + 1
    S = np.zeros((n, n), dtype=np.int64)
    S[0, 1:] = 1
    S[1:, 0] = -1
    S[1:, 1:] = _jacobsthal(q)
    return S + np.eye(n, dtype=np.int64)


def paley2(q: int) -> np.ndarray:
    """Order 2(q+1), q prime, q = 1 (mod 4); symmetric."""
    assert _is_prime(q) and q % 4 == 1
    n = q + 1
    C = np.zeros((n, n), dtype=np.int64)
    C[0, 1:] = 1
    C[1:, 0] = 1
    C[1:, 1:] = _jacobsthal(q)
    H = np.zeros((2 * n, 2 * n), dtype=np.int64)
    plus = np.array([[1, 1], [1, -1]])
    minus = -plus
    diag = np.array([[1, -1], [-1, -1]])
    for i in range(n):
        for j in range(n):
            blk = diag if i == j else (plus if C[i, j] == 1 else minus)
            H[2 * i:2 * i + 2, 2 * j:2 * j + 2] = blk
    return H


@functools.lru_cache(maxsize=None)
def hadamard_matrix(K: int) -> np.ndarray:
    if K == 1:
        return np.ones((1, 1), dtype=np.int64)
    if K % 4 == 0 and _is_prime(K - 1) and (K - 1) % 4 == 3:
        H = paley1(K - 1)
    elif K % 4 == 0 and K // 2 - 1 > 2 and _is_prime(K // 2 - 1) and (K // 2 - 1) % 4 == 1:
        H = paley2(K // 2 - 1)
    elif K % 2 == 0:
        h = hadamard_matrix(K // 2)
        H = np.block([[h, h], [h, -h]])
    else:
        raise ValueError(f"no Hadamard construction for order {K}")
    assert np.array_equal(H @ H.T, K * np.eye(K, dtype=np.int64)), K
    return H


_ORDERS = (172, 156, 140, 108, 60, 52, 44, 36, 28, 40, 20, 12)  # reference order (:6-56) plus 44


def get_hadK(n: int, transpose: bool = False):
    """Same contract as quarot.functional.hadamard.get_hadK: (hadK fp32 [K,K] or None, K)."""
    for K in _ORDERS:
        if n % K == 0 and is_pow2(n // K):
            try:
                H = hadamard_matrix(K)
            except ValueError:
                continue
            t = torch.from_numpy(H.T.copy() if transpose else H.copy()).to(torch.float32)
            return t, K
    assert is_pow2(n), f"no Hadamard factorisation for {n}"
    return None, 1
