"""bench.py contract checks that need no GPU: flags, defaults, workload shapes, JSON keys it promises."""
import ast
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def test_bench_flags_and_defaults():
    src = (ROOT / "bench.py").read_text()
    tree = ast.parse(src)
    flags = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args:
            name = node.args[0].value
            default = next((ast.literal_eval(k.value) for k in node.keywords if k.arg == "default"), None)
            flags[name] = default
    assert flags["--gpus"] == 1 and flags["--steps"] == 10 and flags["--warmup"] == 3
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"',
                '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"',
                '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"', '"cores"', '"kind"', '"sample"'):
        assert key in src, key


def test_synthetic_batch_matches_cfg3_shapes():
    import bench

    b = bench.synthetic_batch(torch.device("cpu"), 0)
    assert b["retriever_query_input_ids"].shape == (18, 50)
    assert b["retriever_passage_input_ids"].shape == (18, 128)
    assert b["generator_input_input_ids"].shape == (18, 256)
    assert b["query_passage_input_len"].shape == (18,) and b["query_passage_input_len"].dtype == torch.int64
    m = b["generator_input_attention_mask"]
    assert int(m[:, -1].min()) == 1 and 60 <= int(m.sum(1).min())   # left-padded, 60..256 live tokens
    assert abs(bench.A100_README_PAIRS_PER_S - 7.9365) < 1e-3
    assert bench.HBM_PEAK_GBPS == 8000.0
