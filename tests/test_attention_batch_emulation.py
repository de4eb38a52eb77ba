"""Lane-for-lane emulation (numpy, CPU) of csrc/attention_batch.h's decode_attn_wave_kernel — the decode-batch attention written at the end of round 4 and not yet run on
a GPU: the index arithmetic of its two register layouts (K: 16 lanes per key, 4 keys per load instruction; V^T: 8 lanes per d-row, 8 d-rows per instruction), the
16-lane dot-product butterflies, the wave-wide online softmax, the probabilities' trip through the wave-private LDS row, the 8-lane d-row sums, the merge of the NWV wave
states with the new key — against dense softmax attention over [cached keys | new key] (HF5:models/llama/modeling_llama.py:191-214).  The emulation is a transcription of the
kernel statement by statement (same names); it checks the ALGORITHM and its layouts, the GPU run still has to check the transcription."""
import numpy as np
import pytest

D, NWV, PIECE = 128, 8, 64


def kernel_emulation(q, k_new, v_new, Kc, Vt, pos, scale):
    """q, k_new (already rotated), v_new: [D]; Kc [s_max, D] rows; Vt [D, s_max]; returns o [D]."""
    scl = scale * 1.4426950408889634
    lanes = np.arange(64)
    sub, kslot = lanes & 15, lanes >> 4
    s8, drow8 = lanes & 7, lanes >> 3
    part = np.zeros((NWV + 1, D + 2))
    n_piece = (pos + PIECE - 1) // PIECE
    for wave in range(NWV):
        m_run, l_run = -np.inf, 0.0
        acc = np.zeros((64, 16))
        qv = np.stack([q[sub * 8 + e] for e in range(8)], axis=1)                    # [lane, 8]
        for pc in range(wave, n_piece, NWV):
            k0 = pc * PIECE
            nk = min(PIECE, pos - k0)
            mine = np.zeros(64)
            for u in range(16):
                kl = 4 * u + kslot
                key = k0 + np.where(kl < nk, kl, nk - 1)
                kraw = np.stack([Kc[key, sub * 8 + e] for e in range(8)], axis=1)      # each lane: 8 dims of its key
                dot = (qv * kraw).sum(1)
                for o in (8, 4, 2, 1):                                                 # butterfly inside the 16-lane group
                    dot = dot + dot[lanes ^ o]
                mine = np.where(u == sub, dot, mine)
            my_key = 4 * sub + kslot
            live = my_key < nk
            s = np.where(live, mine * scl, -np.inf)
            m_new = max(m_run, s.max())
            p = np.where(live, np.exp2(s - m_new), 0.0)
            alpha = np.exp2(m_run - m_new) if np.isfinite(m_run) else 0.0
            l_run = l_run * alpha + p.sum()
            m_run = m_new
            p_lds = np.zeros(PIECE)
            p_lds[my_key] = p                                                          # one float per lane, all 64 keys covered
            assert len(set(my_key.tolist())) == 64
            pp = np.stack([p_lds[s8 * 8 + e] for e in range(8)], axis=1)               # [lane, 8]: the lane's 8 consecutive keys
            for i in range(16):
                vraw = np.stack([Vt[8 * i + drow8, k0 + s8 * 8 + e] for e in range(8)], axis=1)
                acc[:, i] = acc[:, i] * alpha + (pp * vraw).sum(1)
        for i in range(16):
            t = acc[:, i].copy()
            for o in (1, 2, 4):
                t = t + t[lanes ^ o]
            for ln in lanes[s8 == 0]:
                part[wave, 8 * i + drow8[ln]] = t[ln]
        part[wave, D], part[wave, D + 1] = m_run, l_run
    part[NWV, D] = float(q @ k_new) * scl
    part[NWV, D + 1] = 1.0
    part[NWV, :D] = v_new
    M = part[:, D].max()
    l = o = 0.0
    o = np.zeros(D)
    for w in range(NWV + 1):
        m = part[w, D]
        if np.isfinite(m):
            f = np.exp2(m - M)
            l += f * part[w, D + 1]
            o += f * part[w, :D]
    return o / l


@pytest.mark.parametrize("pos", [0, 1, 63, 64, 65, 511, 512, 1087, 1214, 2047])
def test_wave_kernel_algorithm_equals_dense_attention(pos):
    rng = np.random.default_rng(pos)
    s_max = 2048
    q, k_new, v_new = rng.standard_normal(D), rng.standard_normal(D), rng.standard_normal(D)
    Kc = rng.standard_normal((s_max, D)); Vt = rng.standard_normal((D, s_max))
    Kc[pos:] = 1e3; Vt[:, pos:] = -1e3                       # rows past the cached keys must never contribute (masked score, zero probability)
    scale = 1.0 / np.sqrt(D)
    got = kernel_emulation(q, k_new, v_new, Kc, Vt, pos, scale)
    keys = np.concatenate([Kc[:pos], k_new[None]], 0)
    vals = np.concatenate([Vt[:, :pos].T, v_new[None]], 0)
    sc = keys @ q * scale
    w = np.exp(sc - sc.max()); w /= w.sum()
    ref = w @ vals
    assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
