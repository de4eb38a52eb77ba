"""CPU: the host half of a beam-sample step (llava_mi355x/beam.py::rank_draws) — which of the per-row candidates are the 2 * num_beams draws and in what
order GenerationMixin.beam_sample (transformers 4.31: multinomial without replacement over the num_beams x V block, then torch.sort by score) walks them.
The check re-states the step with torch on the flat block: top-K of (score + the same noise) = the draws, then a descending sort of their scores."""
import torch


def test_rank_draws_equals_flat_topk_then_sort():
    from llava_mi355x.beam import rank_draws
    g = torch.Generator().manual_seed(3)
    B, V, K = 3, 50, 6
    for trial in range(20):
        score = torch.randn(B, V, generator=g) * 2 - 3
        noise = -torch.log(-torch.log(torch.rand(B, V, generator=g).clamp(1e-7, 1 - 1e-7)))
        keep = torch.rand(B, V, generator=g) > 0.3
        key = (score + noise).masked_fill(~keep, -float("inf"))
        # what the device returns: per row the K largest keys (rows with fewer survivors pad with id -1)
        kk, ii = torch.topk(key, K, dim=1)
        ids = torch.where(torch.isinf(kk), torch.full_like(ii, -1), ii).to(torch.int32)
        sc = torch.gather(score, 1, ii)
        got = rank_draws(kk, sc, ids, V, K)
        # flat re-statement
        fk, fi = torch.topk(key.reshape(-1), K)
        fs = score.reshape(-1)[fi]
        order = sorted(range(K), key=lambda j: (-float(fs[j]), int(fi[j])))
        want = [(float(fs[j]), int(fi[j]) % V, int(fi[j]) // V) for j in order]
        assert got == want


def test_rank_draws_skips_padding_and_handles_short_rows():
    from llava_mi355x.beam import rank_draws
    keys = torch.tensor([[5.0, 1.0, -float("inf")], [4.0, 3.0, 2.0]])
    scores = torch.tensor([[-1.0, -2.0, -float("inf")], [-0.5, -3.0, -4.0]])
    ids = torch.tensor([[7, 2, -1], [9, 1, 0]], dtype=torch.int32)
    got = rank_draws(keys, scores, ids, 10, 4)
    assert got == [(-0.5, 9, 1), (-1.0, 7, 0), (-3.0, 1, 1), (-4.0, 0, 1)]        # draws = keys 5, 4, 3, 2; ranked by score
