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


def test_bench_self_spawns_and_refuses_clearly_without_enough_gpus():
    """`python bench.py --gpus N` (the driver's form, no torchrun): bench.py spawns the ranks itself; on a box with
    fewer GPUs it says so and exits 2 instead of hanging or dying in rendezvous."""
    import os
    import subprocess

    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])
    assert "8 ranks requested but only" in r.stderr and "GPU(s) are visible" in r.stderr
    assert r.stdout.strip() == ""                                   # no JSON line pretending to be a result


def test_bench_workloads_and_cfg5_shapes():
    import bench

    src = (ROOT / "bench.py").read_text()
    for w in ('"cfg3"', '"cfg5"', '"cfg2"', '"cfg1"', '"--dtype"', '"fp32"', "ranks_seen_by_process_group", "traffic_source"):
        assert w in src, w
    assert bench.GENERATORS["falcon-7b"][1] == 65024 and bench.GENERATORS["llama-2-7b"][1] == 32000
    fc = bench.GENERATORS["falcon-7b"][0](32)
    assert (fc.hidden_size, fc.num_hidden_layers, fc.num_attention_heads, fc.vocab_size) == (4544, 32, 71, 65024)
    lc = bench.GENERATORS["llama-2-7b"][0](32)
    assert (lc.hidden_size, lc.num_hidden_layers, lc.intermediate_size, lc.vocab_size) == (4096, 32, 11008, 32000)
    b = bench.synthetic_batch(torch.device("cpu"), 0, V=65024)
    assert int(b["generator_input_input_ids"].max()) > 32000 and int(b["generator_input_input_ids"].max()) < 65024
    # a tiny cfg5-architecture model assembles (LoRA lands on Falcon's fused query_key_value)
    m = bench.build_models(torch.device("cpu"), torch.float32, bert_layers=1, llama_layers=1, generator="falcon-7b")
    assert m.generator_model._dalm_lora_config["target_modules"] == ["query_key_value"]


def test_bucketed_data_path_is_opt_in_and_keeps_every_row():
    """--data-path bucketed (an extra line, never the default): batches of B rows, trimmed widths that vary with the bucket,
    no live token lost, and fewer generator tokens than the fixed-shape batches."""
    import bench

    src = (ROOT / "bench.py").read_text()
    assert 'ap.add_argument("--data-path", default="fixed"' in src
    bs = bench.bucketed_batches(torch.device("cpu"), 100, 32000, n_batches=6)
    pool = [bench.synthetic_batch(torch.device("cpu"), 100 + i, V=32000) for i in range(6)]
    live_in = sum(int(b["generator_input_attention_mask"].sum()) for b in pool)
    live_out = sum(int(b["generator_input_attention_mask"].sum()) for b in bs)
    assert live_in == live_out
    widths = {b["generator_input_input_ids"].shape[1] for b in bs}
    assert all(b["generator_input_input_ids"].shape[0] == bench.CFG["B"] for b in bs) and len(widths) > 1
    assert sum(b["generator_input_input_ids"].numel() for b in bs) < 0.85 * 6 * bench.CFG["B"] * bench.CFG["Tg"]


def test_kernel_bench_printer_takes_label_values(capsys):
    """tools/kernel_bench.py crashed in round 5 formatting the string "bf16 x3" with :.4g."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("kernel_bench", ROOT / "tools" / "kernel_bench.py")
    kb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kb)
    kb.print_results({"sim 4096x4096 D1024": {"rowstats": {"s": 1.25e-4, "pipe": "bf16 x3", "frac": 0.47, "n": 3}}})
    out = capsys.readouterr().out
    assert "pipe=bf16 x3" in out and "frac=0.47" in out
