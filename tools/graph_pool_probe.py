"""Memory of the shared hipGraph pool as GraphedStep captures one graph per batch shape (bench.py --data-path bucketed / packed):
allocated / reserved bytes after every new capture, for a depth-reduced cfg3 model.
    python tools/graph_pool_probe.py [--layers 8] [--mode bucketed|packed]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--mode", default="bucketed")
    a = ap.parse_args()
    import bench
    from dalm_amd.packed import add_pack_plans
    from dalm_amd.training.graphed import GraphedStep, TensorLRScheduler, make_capturable_adam
    from dalm_amd.training.step import RagE2EStep
    from dalm_amd.tuning import enable_tuned_gemms

    enable_tuned_gemms()
    dev = torch.device("cuda:0")
    model = bench.build_models(dev, torch.bfloat16, a.layers, a.layers)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = make_capturable_adam(params, 1e-4, dev)
    sched = TensorLRScheduler(opt, 1e-4, lambda o: torch.optim.lr_scheduler.LambdaLR(o, lambda s: 1.0))
    step = GraphedStep(RagE2EStep(model, opt, sched, 100, autocast_dtype=torch.bfloat16, inplace_grad=True), max_graphs=32)
    if a.mode == "bucketed":
        batches = bench.bucketed_batches(dev, 100, 32000)
    else:
        batches = [{k: v.to(dev) for k, v in add_pack_plans(bench.synthetic_batch(torch.device("cpu"), 100 + i)).items()} for i in range(12)]
    print(f"after build: allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB")
    for i, b in enumerate(batches):
        step(b)
        torch.cuda.synchronize()
        shp = {k: tuple(v.shape) for k, v in b.items() if k.endswith("input_ids") or k.endswith("pack_rows")}
        print(f"batch {i:2d} graphs {len(step.graphs):2d} failed {step.failed}  allocated {torch.cuda.memory_allocated() / 2**30:7.2f} GiB  "
              f"reserved {torch.cuda.memory_reserved() / 2**30:7.2f} GiB  {shp}", flush=True)


if __name__ == "__main__":
    main()
