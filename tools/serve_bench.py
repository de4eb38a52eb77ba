"""Concurrent-request throughput through model.generate (the worker's thread-per-request model, llava/serve/model_worker.py:174-185)
with the continuous-batching scheduler on and off.  7B geometry, every request = 1 image + 512-token prompt, 128 new tokens.
usage: python tools/serve_bench.py [n_requests] [capacity] [new_tokens]"""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))


CALLBACK_US = float(os.environ.get("CALLBACK_US", "0"))


def run(model, reqs, new_tokens, sample):
    lat, first = [None] * len(reqs), [None] * len(reqs)

    class FirstToken:                     # a streamer that only records when the first generated token arrived
        def __init__(self, i): self.i, self.n = i, 0
        def put(self, v):
            self.n += 1
            if self.n == 2: first[self.i] = time.perf_counter()
            if CALLBACK_US:                  # stand-in for TextIteratorStreamer's tokenizer.decode + KeywordsStoppingCriteria per token
                t_end = time.perf_counter() + CALLBACK_US * 1e-6
                while time.perf_counter() < t_end:
                    pass
        def end(self): pass

    def one(i):
        ids, pix = reqs[i]
        with torch.cuda.stream(torch.cuda.Stream()):
            kw = dict(do_sample=True, temperature=0.7, top_p=0.9) if sample else dict(do_sample=False)
            model.generate(inputs=ids, images=pix, max_new_tokens=new_tokens, eos_token_id=-1, streamer=FirstToken(i), **kw)
        lat[i] = time.perf_counter()

    t0 = time.perf_counter()
    ths = [threading.Thread(target=one, args=(i,)) for i in range(len(reqs))]
    for t in ths: t.start()
    for t in ths: t.join()
    wall = time.perf_counter() - t0
    return {"wall_s": wall, "tokens_per_s": len(reqs) * new_tokens / wall, "ttft_ms_median": sorted((f - t0) * 1e3 for f in first)[len(reqs) // 2],
            "ttft_ms_max": max((f - t0) * 1e3 for f in first)}


def main():
    from synthetic import build as harness, recipes as synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    new_tokens = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    cfg = synth.CONFIGS["llava15_7b"]
    dev = torch.device("cuda:0")
    model = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, device_rng=True, device=dev, max_position=2048)
    reqs = []
    for i in range(n):
        ids = torch.from_numpy(synth.make_prompt(cfg, 512, image_positions=(35,), seed=100 + i))[None].to(dev)
        pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=200 + i)).to(dev, torch.bfloat16)
        reqs.append((ids, pix))
    run(model, reqs[:2], 8, False)                        # warm-up
    out = {"requests": n, "new_tokens": new_tokens, "capacity": cap, "callback_us": CALLBACK_US}
    if not os.environ.get("SKIP_PLAIN"):
        out["threads_no_batching"] = run(model, reqs, new_tokens, False)
    # packed: the waiting requests are prefilled together — on the prefill thread beside the decode steps (round 5's default), or between two decode steps on the
    # scheduler's own stream (round 4: prefill_thread=0); thread_prefill: one prefill per request thread
    for tag, packed, pt in (("packed_prefill", True, 2), ("packed_prefill_between_steps", True, 0), ("thread_prefill", False, 0)):
        if os.environ.get("ONLY") and tag not in os.environ["ONLY"].split(","):
            continue
        model.enable_batching(capacity=cap, packed_prefill=packed, prefill_thread=pt)
        run(model, reqs[:4], 8, False)
        out["continuous_batching_" + tag] = run(model, reqs, new_tokens, False)
        out["continuous_batching_sampled_" + tag] = run(model, reqs, new_tokens, True)
        out["scheduler_" + tag] = {"steps": model._batcher.steps, "member_steps": model._batcher.member_steps, "max_live": model._batcher.max_live,
                                   "prefill_batches": model._batcher.prefill_batches, "prefilled": model._batcher.prefilled}
        model.disable_batching()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
