"""bench.py's bookkeeping (CPU): the algorithmic work it prices the rooflines with equals BASELINE.md §2 / SURVEY §8(d) (decoder linears 14.079 TFLOP,
causal attention 0.310, vision tower 0.366, projector 0.024 per 1087-position prefill of LLaVA-1.5-7B; 13.214 GB of weights per decoded token), the
command line keeps the driver's contract (--gpus / --steps / --warmup, defaults that finish within minutes), and the timed region never touches the oracle."""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_algorithmic_work_matches_the_baseline_figures():
    import bench
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["llava15_7b"]
    T = 512 - 1 + cfg.tokens_per_image
    assert T == 1087
    fl = bench.flops_prefill(cfg, T, 1)
    assert abs(fl["linear"] / 1e12 - 14.079) < 0.005
    assert abs(fl["attention"] / 1e12 - 0.310) < 0.005
    assert abs(fl["vision"] / 1e12 - 0.366) < 0.005
    assert abs(fl["projector"] / 1e12 - 0.024) < 0.002
    assert abs(fl["total"] / 1e12 - 14.78) < 0.02
    H, I, L, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size
    assert abs((4 * H * H * 2 * L + 3 * H * I * 2 * L + V * H * 2) / 1e9 - 13.214) < 0.002


def test_command_line_contract(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps >= 1 and a.warmup >= 0 and a.workload == "config2" and a.model == "llava15_7b" and a.dtype == "bf16"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2)


def test_oracle_is_only_imported_by_the_cpu_baseline_leg():
    """Every `oracle` import of bench.py sits inside cpu_baseline / cpu_baseline_full; nothing under the product package imports it at all."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            uses = [n for n in ast.walk(node) if isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n)]
            if uses:
                assert node.name.startswith("cpu_baseline"), node.name
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n)]
    assert not top
    pkg = os.path.join(ROOT, "llava-plus-codebase_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(d, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "/root/reference" not in text, os.path.join(d, f)
