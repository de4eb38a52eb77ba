"""One rank of a TP=2 serving group whose ranks share ONE GPU (test helper, run as a subprocess by test_tp_serving_gpu.py).

rank 0 = leader: continuous-batching scheduler with a command channel, N concurrent request threads calling model.generate (greedy with
a streamer and a stopping criterion, and sampled) exactly as the worker's threads do; rank 1 = follower: tp_serving.serve_follower.
usage: tp_serving_worker.py rank world port dtype out.json config"""
import json
import os
import sys
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))


def requests_for(ids, n):
    """n request variants of one golden prompt: (prompt ids [1, L], max_new_tokens, sampled?)."""
    out = []
    L = ids.shape[1]
    for i in range(n):
        cut = L - (i % 3) * 2                       # different prompt lengths (the image marker sits early in the prompt)
        out.append((ids[:, :cut].clone(), 5 + (i % 4) * 3, i % 3 == 2))
    return out


class Collect:
    """Streamer stand-in (put/end), as TextIteratorStreamer is fed."""
    def __init__(self):
        self.chunks = []

    def put(self, t):
        self.chunks.append(t.reshape(-1).tolist())

    def end(self):
        self.chunks.append(None)


def main():
    rank, world, port, dts, out, name = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
    os.environ["LMX_TP_P2P_ALL"] = "1"
    import torch.distributed as dist
    from golden_util import case_inputs, load
    from synthetic import build as harness
    from llava_mi355x import tp_serving
    dt = {"f32": torch.float32, "bf16": torch.bfloat16}[dts]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = {"rank": rank}
    try:
        z, meta = load(name)
        cfg, cm, ids, mask, labels, pix = case_inputs(z, meta, "single")
        model = harness.build_model(cfg, dtype=dt, seed=0, tp_rank=rank, tp_world=world)
        model.init_tensor_parallel(rccl=False, p2p=True)
        chan = tp_serving.CommandChannel(dist.new_group(backend="gloo"))
        n_req = 7
        if rank == 0:
            ids_t = torch.from_numpy(ids).cuda(); pix_t = torch.from_numpy(pix).cuda().to(dt)
            reqs = requests_for(ids_t, n_req)
            model.enable_batching(capacity=4, channel=chan)             # capacity < requests: some wait for a slot
            outs, streams, errs = [None] * n_req, [Collect() for _ in range(n_req)], []

            def run(i):
                p, new, sampled = reqs[i]
                try:
                    torch.manual_seed(100 + i)
                    stop_at = lambda full, scores: full.shape[1] >= p.shape[1] + new - 1 if i == 1 else False      # a stopping criterion
                    o = model.generate(inputs=p, images=pix_t, do_sample=sampled, temperature=0.8 if sampled else 0.0, top_p=0.9 if sampled else None,
                                       max_new_tokens=new, eos_token_id=-1, streamer=streams[i], stopping_criteria=[stop_at])
                    outs[i] = o[0, p.shape[1]:].cpu().tolist()
                except Exception as e:  # noqa: BLE001
                    errs.append((i, repr(e)))

            ths = [threading.Thread(target=run, args=(i,)) for i in range(n_req)]
            for t in ths: t.start()
            for t in ths: t.join(timeout=180)
            b = model._batcher
            res.update({"outs": outs, "errs": errs, "steps": b.steps, "member_steps": b.member_steps, "max_live": b.max_live, "sent": chan.sent, "prefill_batches": b.prefill_batches, "prefilled": b.prefilled,
                        "streamed": [[t for c in s.chunks[1:] if c for t in c] for s in streams],
                        "stream_ended": [s.chunks[-1] is None for s in streams]})
            model.disable_batching()                                     # -> ("stop",)
        else:
            stats = tp_serving.serve_follower(model, chan, capacity=4, record_tokens=True)
            res.update({k: v for k, v in stats.items() if k != "tokens"})
            res["tokens"] = sorted(stats.get("tokens", {}).values())
        res["status"] = int(__import__("llava_mi355x")._C.lib.lmx_tp_p2p_status(model._h, None))
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
