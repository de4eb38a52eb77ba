"""One rank of a TP group whose ranks share ONE GPU, at the REAL hidden width (test helper of test_tp_p2p_gpu.py::test_two_shot_allreduce_real_width).

The two-shot (reduce-scatter + all-gather) kernel of csrc/p2p.hip on messages of the LLaVA-1.5-7B prefill's size: [rows, 4096] of the model dtype, exchange
region sized by the engine for hidden 4096 (2048 slices per chunk, 16.8 MB per source and parity).  Only all-reduces run here: while a rank's 500-odd
workgroups wait for their peer on a SHARED GPU no kernel that needs a whole CU (the prefill GEMMs) could start, which is why the bench's one-GPU dry run keeps
prefill-sized sums on 32-row launches (LMX_TP_P2P_BIG=0); between GPUs every rank has its own CUs.  usage: p2p_big_worker.py rank world port out.json"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as dist
    from llava_mi355x import _C
    from synthetic import build as harness
    from synthetic import recipes as synth
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = {"rank": rank}
    try:
        cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 1, 1)
        dt = torch.bfloat16
        model = harness.build_model(cfg, dtype=dt, seed=0, device_rng=True, tp_rank=rank, tp_world=world, max_position=2048)
        model.init_tensor_parallel(rccl=False, p2p=True)
        res["p2p_active"] = bool(model.p2p_active)
        H = cfg.hidden_size
        # <= 1536 rows: a rank's launch is rows x H / world / 4096 workgroups (768 here) and BOTH ranks' launches must be resident on this one GPU at the same
        # time (2048 workgroup slots) — a 4096-row message (2048 workgroups per rank) lets the first rank's launch fill the chip and wait for a peer that can
        # never start (30 s time-out per message; between GPUs each rank has its own chip)
        ok, sizes = True, (1087, 1087, 33, 1536, 577, 1152, 1087, 40)             # 1152 = 2 x 576: an image-feature gather piece of two images
        for it, rows in enumerate(sizes):
            col = torch.arange(H, device="cuda") % 11
            base = (col[None, :] + torch.arange(rows, device="cuda")[:, None] * 3 + it) % 5                # values 0..4: every partial sum is exact in bf16
            mine = (base * (rank + 1)).to(dt).contiguous()
            want = (base * (world * (world + 1) // 2)).to(dt)
            _C.check(_C.lib.lmx_op_allreduce(model._h, _C.ptr(mine), rows * H, _C.stream_handle()))
            torch.cuda.synchronize()
            good = bool(torch.equal(mine, want))
            ok = ok and good
            res.setdefault("by_rows", []).append([rows, good])
        res["big_ok"] = ok
        res["status"] = int(_C.lib.lmx_tp_p2p_status(model._h, None))
        # the image-feature all-gather as model._run_tower issues it: each rank fills its own images' rows of a zeroed buffer
        n_img, P = 5, cfg.tokens_per_image
        feats = torch.zeros((n_img, P, H), dtype=dt, device="cuda")
        g = torch.Generator(device="cuda").manual_seed(7)
        full = torch.randn((n_img, P, H), dtype=torch.float32, device="cuda", generator=g).to(dt)
        mine_idx = torch.arange(rank, n_img, world, device="cuda")
        feats.index_copy_(0, mine_idx, full.index_select(0, mine_idx))
        per = 2                                             # model._run_tower takes 4096 // P = 7 images per piece; 2 keeps both ranks' launches resident here
        for i0 in range(0, n_img, per):
            piece = feats[i0:i0 + per]
            _C.check(_C.lib.lmx_op_allreduce(model._h, _C.ptr(piece), piece.numel(), _C.stream_handle()))
        torch.cuda.synchronize()
        res["gather_ok"] = bool(torch.equal(feats, full))
        res["status_gather"] = int(_C.lib.lmx_tp_p2p_status(model._h, None))
        # latency of one 1087 x 4096 message (8.9 MB) between two processes on one GPU: protocol + the copies through this GPU's own memory
        buf = torch.ones((1087, H), dtype=dt, device="cuda")
        for rows, key in ((1087, "us_per_allreduce_1087x4096"), (1536, "us_per_allreduce_1536x4096")):
            buf = torch.ones((rows, H), dtype=dt, device="cuda")
            for _ in range(3):
                _C.check(_C.lib.lmx_op_allreduce(model._h, _C.ptr(buf), buf.numel(), _C.stream_handle())); buf.fill_(1)
            torch.cuda.synchronize(); dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _C.check(_C.lib.lmx_op_allreduce(model._h, _C.ptr(buf), buf.numel(), _C.stream_handle()))
            e1.record(); torch.cuda.synchronize()
            res[key] = e0.elapsed_time(e1) / 20 * 1e3
        res["status_end"] = int(_C.lib.lmx_tp_p2p_status(model._h, None))
        res["ok"] = True
    except Exception as e:  # noqa: BLE001
        import traceback
        res["ok"] = False; res["error"] = repr(e); res["trace"] = traceback.format_exc()[-1500:]
    json.dump(res, open(out, "w"))
    try:
        dist.barrier(); dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


if __name__ == "__main__":
    main()
