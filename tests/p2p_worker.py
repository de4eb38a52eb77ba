"""One rank of a TP group whose ranks share ONE GPU (test helper, run as a subprocess by test_tp_p2p_gpu.py).

RCCL refuses two ranks on one device, so this configuration has no RCCL communicator at all: every all-reduce of the engine
(prefill-sized ones too: LMX_TP_P2P_ALL=1) goes through the one-shot peer-to-peer kernel, whose exchange buffers are mapped
between the two PROCESSES with HIP IPC exactly as they are between GPUs.  usage: p2p_worker.py rank world port dtype out.json [config]"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))


def main():
    rank, world, port, dts, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    name = sys.argv[6] if len(sys.argv) > 6 else "tiny"
    os.environ["LMX_TP_P2P_ALL"] = "1"
    import torch.distributed as dist
    from golden_util import case_inputs, load
    from synthetic import build as harness
    dt = {"f32": torch.float32, "bf16": torch.bfloat16}[dts]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = {"rank": rank}
    try:
        z, meta = load(name)
        cfg, cm, ids, mask, labels, pix = case_inputs(z, meta, "single")
        model = harness.build_model(cfg, dtype=dt, seed=0, tp_rank=rank, tp_world=world)
        model.init_tensor_parallel(rccl=False, p2p=True)
        res["p2p_active"] = bool(model.p2p_active)
        ids_t = torch.from_numpy(ids).cuda(); pix_t = torch.from_numpy(pix).cuda().to(dt)
        logits = model.forward(input_ids=ids_t, images=pix_t, use_cache=False).logits.float().cpu().numpy()
        gen = model.generate(inputs=ids_t, images=pix_t, do_sample=False, max_new_tokens=6, eos_token_id=-1, run_ahead=3).cpu().numpy()
        # a decode batch of 3 sequences: [3, H] rows per all-reduce
        prompts = [ids_t[0], ids_t[0, :9], ids_t[0]]
        outs = model.generate_batch(prompts, [pix_t, pix_t, pix_t], max_new_tokens=5, eos_token_id=-1, run_ahead=2)
        # the batched steps' RMSNorms ride in the all-reduce launches (p2p.hip, round 6): only the first layer's input norm of a step is still its own launch
        model.profile(True)
        outs_p = model.generate_batch(prompts, [pix_t, pix_t, pix_t], max_new_tokens=5, eos_token_id=-1, run_ahead=2)
        prof = model.profile_read(); model.profile(False)
        res["batch_rmsnorm_launches"] = int(prof.get("decode_batch.rmsnorm", (0, 0))[1]); res["batch_linear_launches"] = int(prof.get("decode_batch.linear", (0, 0))[1])
        res["batch_profiled_equal"] = all(torch.equal(a, b) for a, b in zip(outs, outs_p))
        # sampled generation: the ranks' CPU generators are seeded DIFFERENTLY on purpose; rank 0's sampler seed is broadcast
        # (LlavaLlamaForCausalLM._draw_seed), so every rank must still draw the same ids
        torch.manual_seed(1000 + 17 * rank)
        samp = model.generate(inputs=ids_t, images=pix_t, do_sample=True, temperature=0.9, top_p=0.95, max_new_tokens=8, eos_token_id=-1)
        res["sampled"] = samp.cpu().tolist()
        res["vocab_split"] = bool(cfg.vocab_size % (8 * world) == 0)
        ref = z["single.logits"]
        res["logits_err"] = float(np.abs(logits - ref).max())
        res["logits_scale"] = float(np.abs(ref).max())
        res["gen"] = gen.tolist()
        res["gen_ref"] = z["single.generate"][:, : ids.shape[1] + 6].tolist()
        res["batch0"] = outs[0].cpu().tolist()
        res["status"] = int(__import__("llava_mi355x")._C.lib.lmx_tp_p2p_status(model._h, None))
        # prefill-sized messages: the two-shot (reduce-scatter + all-gather) kernel of p2p.hip, exact integer data, odd sizes, back-to-back launches (slot parity)
        _C = __import__("llava_mi355x")._C
        Hm = cfg.hidden_size
        big_ok = True
        for it, rows in enumerate((33, 64, 577, 1087, 1087, 40, 2047)):
            col = torch.arange(Hm, device="cuda") % 11
            base = (col[None, :] + torch.arange(rows, device="cuda")[:, None] * 3 + it) % 5                # values 0..4
            mine = (base * (rank + 1)).to(dt).contiguous()
            want = (base * (world * (world + 1) // 2)).to(dt)
            _C.check(_C.lib.lmx_op_allreduce(model._h, _C.ptr(mine), rows * Hm, _C.stream_handle()))
            torch.cuda.synchronize()
            big_ok = big_ok and bool(torch.equal(mine, want))
        res["big_ok"] = big_ok
        res["status_big"] = int(_C.lib.lmx_tp_p2p_status(model._h, None))
        buf = torch.ones((1087, Hm), dtype=dt, device="cuda")
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            _C.check(_C.lib.lmx_op_allreduce(model._h, _C.ptr(buf), buf.numel(), _C.stream_handle())); buf.fill_(1)
        torch.cuda.synchronize(); dist.barrier()
        e0.record()
        for _ in range(50):
            _C.check(_C.lib.lmx_op_allreduce(model._h, _C.ptr(buf), buf.numel(), _C.stream_handle()))
        e1.record(); torch.cuda.synchronize()
        res["us_per_allreduce_1087_rows"] = e0.elapsed_time(e1) / 50 * 1e3
        # latency of the decode-sized all-reduce (both ranks on one GPU here: protocol cost without the xGMI hop)
        H = cfg.hidden_size
        for Hn, key in ((H, "us_per_allreduce_tinyH"), (4096, "us_per_allreduce_H4096")):
            if Hn != H:
                break                      # the exchange buffer is sized for the model's H
            buf = torch.ones((1, Hn), dtype=dt, device="cuda")
            _C = __import__("llava_mi355x")._C
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(20):
                _C.check(_C.lib.lmx_op_allreduce(model._h, _C.ptr(buf), Hn, _C.stream_handle())); buf.fill_(1)
            torch.cuda.synchronize(); dist.barrier()
            e0.record()
            for _ in range(200):
                _C.check(_C.lib.lmx_op_allreduce(model._h, _C.ptr(buf), Hn, _C.stream_handle()))
            e1.record(); torch.cuda.synchronize()
            res[key] = e0.elapsed_time(e1) / 200 * 1e3
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
