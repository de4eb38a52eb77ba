"""CPU emulation of csrc/decode_attn.hip's arithmetic (dec_attn_chunk): every 128-key chunk reduces its CACHED keys to a partial {o[D], max, sum} in the log2 domain, the
head's last chunk merges the partials in chunk order and folds the NEWEST key in as one more term,

    s = scale log2(e) (q . k_new),  M' = max(M, s),  out = (2^(M - M') o + 2^(s - M') v_new) / (2^(M - M') l + 2^(s - M'))

— checked against dense softmax attention over keys 0 .. pos in float64 (the single-token branch of llava_arch.py:103-112 feeding
HF5:models/llama/modeling_llama.py:191-214), incl. an empty cache, a position on a chunk boundary (the merger's own chunk is empty), a chunk with one key, masked tail keys
holding garbage.  The GPU kernels are checked against fp64 and against each other in tests/test_ops_gpu.py / tests/test_decode_splitq_gpu.py; this pins the FORMULA."""
import numpy as np
import pytest

CHUNK = 128
LOG2E = 1.4426950408889634


def emulate(q, k_new, v_new, Kc, Vt, pos, scale):
    D = q.shape[0]
    n_split = pos // CHUNK + 1
    parts = []
    for split in range(n_split):
        k0 = split * CHUNK
        nk = min(pos - k0, CHUNK)                                  # cached keys of this chunk (the last chunk: pos - k0 in 0 .. 127)
        sc = np.full(CHUNK, -np.inf)
        sc[:nk] = (Kc[k0:k0 + nk] @ q) * (scale * LOG2E)
        mx = sc.max() if nk > 0 else -np.inf
        e = np.where(np.isneginf(sc), 0.0, np.exp2(sc - (mx if nk > 0 else 0.0)))
        vt = Vt[:, k0:k0 + CHUNK].copy()
        vt[:, nk:] = 0.0                                           # the kernel clears the V^T bytes past the cached keys (they may hold anything)
        parts.append((vt @ e, mx, e.sum()))
    M = max(p[1] for p in parts)
    l = sum(np.exp2(p[1] - M) * p[2] for p in parts if not np.isneginf(p[1]))
    o = sum((np.exp2(p[1] - M) * p[0] for p in parts if not np.isneginf(p[1])), np.zeros(D))
    s = float(q @ k_new) * (scale * LOG2E)
    M2 = max(M, s)
    wa = np.exp2(M - M2) if not np.isneginf(M) else 0.0
    wb = np.exp2(s - M2)
    return (wa * o + wb * v_new) / (wa * l + wb)


@pytest.mark.parametrize("pos", [0, 1, 127, 128, 129, 255, 256, 1087, 1150, 2047])
def test_chunk_partials_merge_and_new_key_fold_equal_dense_attention(pos):
    rng = np.random.default_rng(pos + 7)
    D, s_max = 128, 2048
    scale = 1.0 / np.sqrt(D)
    q = rng.standard_normal(D); k_new = rng.standard_normal(D); v_new = rng.standard_normal(D)
    Kc = rng.standard_normal((s_max, D)); Vt = rng.standard_normal((D, s_max))
    Vt[:, pos:] = np.nan                                           # anything past the cached keys must never reach the sum
    Kc[pos:] = 1e30
    got = emulate(q, k_new, v_new, Kc, Vt, pos, scale)
    keys = np.concatenate([Kc[:pos], k_new[None]], 0)
    vals = np.concatenate([Vt[:, :pos].T, v_new[None]], 0)
    sc = keys @ q * scale
    p = np.exp(sc - sc.max()); p /= p.sum()
    ref = p @ vals
    assert np.all(np.isfinite(got))
    assert np.abs(got - ref).max() < 1e-12, np.abs(got - ref).max()
