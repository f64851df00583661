"""world_size-2 `gloo` tests of the sharded-negatives host logic (dalm_amd.fused + dalm_amd.sharded).

The product kernels need a GPU; here the SAME host code (collectives, coefficient assembly, autograd
wiring) runs on CPU with the checker backend `OracleOps` injected through the `ops=` argument, and must
reproduce the single-process result at the global batch: W ranks x B_l == 1 rank x W*B_l, loss and all grads.
"""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist

    import dalm_oracle as O
    from dalm_amd.fused import GatherHandle, TorchDistComm, contrastive_loss, rag_e2e_loss
    from dalm_amd.sharded import GradBucket, allreduce_grads
    from helpers import synth_batch

    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = TorchDistComm()
    ops = O.OracleOps()
    B_l, D, Tg, V = 3, 16, 10, 37
    q, p, logits, ids, mask, qlen = synth_batch(42, world * B_l, D, Tg, V, pad_side="left", logit_gain=2.0)
    sl = slice(rank * B_l, (rank + 1) * B_l)
    w = torch.nn.Parameter(torch.eye(D) + 0.01 * torch.arange(D * D, dtype=torch.float32).reshape(D, D) / (D * D))
    bucket = GradBucket([w], comm) if mode == "e2e" else None   # e2e: bucket views; contrastive: flatten path
    ql, pl = (q[sl] @ w), (p[sl] @ w)          # a shared "tower" parameter, replicated on every rank
    lg = logits[sl].clone().requires_grad_(True)
    if mode == "e2e":
        ph = GatherHandle(pl, comm)             # the early all-gather the trainer starts after the passage tower
        loss = rag_e2e_loss(ql, pl, lg, ids[sl], mask[sl], qlen[sl], 100, comm=comm, ops=ops, p_gather=ph)
    else:
        loss = contrastive_loss(ql, pl, 100, comm=comm, ops=ops)
    loss.backward()
    if bucket is not None:
        assert w.grad.data_ptr() == bucket.flat.data_ptr()
        bucket.all_reduce()                     # SUM over ranks, in the bucket
    else:
        allreduce_grads([w], comm)              # SUM over ranks
    total = loss.detach().clone()
    dist.all_reduce(total)
    torch.save({"loss_share": loss.detach(), "loss_total": total, "dw": w.grad.clone(),
                "dlogits": lg.grad.clone() if lg.grad is not None else None}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("e2e", 2), ("contrastive", 2), ("e2e", 3)])
def test_ranks_equal_one_process_at_global_batch(tmp_path, mode, world):
    import dalm_oracle as O
    from helpers import synth_batch

    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]

    # single process at the global batch, reference op sequence in fp64
    B_l, D, Tg, V = 3, 16, 10, 37
    q, p, logits, ids, mask, qlen = synth_batch(42, world * B_l, D, Tg, V, pad_side="left", logit_gain=2.0)
    w = (torch.eye(D) + 0.01 * torch.arange(D * D, dtype=torch.float32).reshape(D, D) / (D * D)).double().requires_grad_(True)
    lg = logits.double().requires_grad_(True)
    out = O.ref_step_loss(q.double() @ w, p.double() @ w, lg if mode == "e2e" else None, ids, mask, qlen, 100)
    out["loss"].backward()

    assert abs(float(res[0]["loss_total"]) - float(out["loss"])) <= 1e-5 * abs(float(out["loss"]))
    assert abs(sum(float(r["loss_share"]) for r in res) - float(out["loss"])) <= 1e-5 * abs(float(out["loss"]))
    for r in range(world):  # every rank holds the same, fully reduced parameter gradient
        torch.testing.assert_close(res[r]["dw"].double(), w.grad, rtol=2e-4, atol=1e-6)
    if mode == "e2e":
        got = torch.cat([res[r]["dlogits"] for r in range(world)]).double()
        torch.testing.assert_close(got, lg.grad, rtol=2e-4, atol=1e-8)


def test_local_comm_is_identity():
    from dalm_amd.fused import GatherHandle, LocalComm

    c = LocalComm()
    t = torch.arange(6.0).reshape(3, 2)
    assert c.all_gather_rows(t) is t and c.all_reduce_sum_(t) is t
    assert torch.equal(GatherHandle(t, c).wait(), t)
