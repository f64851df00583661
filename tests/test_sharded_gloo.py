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


def _spawn(fn, world, *rest):
    """mp.spawn(fn, (world, <free port>, *rest)); one retry on a fresh port: the port found by _free_port() is released
    before the ranks bind it, and on a busy host something else can take it in between (seen once in ~30 suite runs)."""
    for attempt in (0, 1):
        try:
            mp.spawn(fn, args=(world, _free_port()) + tuple(rest), nprocs=world, join=True)
            return
        except Exception as e:                      # ProcessRaisedException / ProcessExitedException
            text = str(e).lower()
            if attempt == 1 or not any(w in text for w in ("address already in use", "socket", "connection", "tcpstore")):
                raise


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
    bucket = GradBucket([w], comm) if mode.startswith("e2e") else None   # e2e: bucket views; contrastive: flatten path
    ql, pl = (q[sl] @ w), (p[sl] @ w)          # a shared "tower" parameter, replicated on every rank
    lg = logits[sl].clone().requires_grad_(True)
    if mode == "e2e_hidden":
        # the fused lm_head path on every rank, over its own live rows: logits = hidden @ head^T are never formed
        from dalm_amd.fused import live_row_index, rag_e2e_loss_from_hidden

        g = torch.Generator().manual_seed(7)
        hidden_all = torch.randn(world * B_l, Tg, 12, generator=g)
        head = 0.5 * torch.randn(V, 12, generator=g)
        lg = hidden_all[sl].clone().requires_grad_(True)
        loss = rag_e2e_loss_from_hidden(ql, pl, lg, head, ids[sl], mask[sl], qlen[sl], 100, comm=comm, ops=ops,
                                        chunk_samples=2, live_rows=live_row_index(mask[sl], 4))
    elif mode == "e2e_packed":
        # round 6: every rank runs its generator on its own PACKED rows (dalm_amd/packed.py); the loss gets the hidden states of
        # those rows + their shifted labels, the token count M and the softmax statistics still span the global batch
        from dalm_amd import packed
        from dalm_amd.fused import rag_e2e_loss_packed

        g = torch.Generator().manual_seed(7)
        hidden_all = torch.randn(world * B_l, Tg, 12, generator=g)
        head = 0.5 * torch.randn(V, 12, generator=g)
        rows, _cu = packed.pack_plan(mask[sl], shifted=True, multiple=4)
        lg = hidden_all[sl].clone().requires_grad_(True)
        hp = lg.reshape(-1, 12).index_select(0, rows.clamp_min(0))      # what the packed generator would hand over
        y, wts = packed.packed_labels(ids[sl], mask[sl], rows)
        loss = rag_e2e_loss_packed(ql, pl, hp, head, y, wts, mask[sl], qlen[sl], 100, comm=comm, ops=ops)
    elif mode == "e2e":
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


@pytest.mark.parametrize("mode,world", [("e2e", 2), ("contrastive", 2), ("e2e", 3), ("e2e", 8), ("e2e_hidden", 2), ("e2e_packed", 2)])
def test_ranks_equal_one_process_at_global_batch(tmp_path, mode, world):
    import dalm_oracle as O
    from helpers import synth_batch

    _spawn(_worker, world, mode, str(tmp_path))
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]

    # single process at the global batch, reference op sequence in fp64
    B_l, D, Tg, V = 3, 16, 10, 37
    q, p, logits, ids, mask, qlen = synth_batch(42, world * B_l, D, Tg, V, pad_side="left", logit_gain=2.0)
    w = (torch.eye(D) + 0.01 * torch.arange(D * D, dtype=torch.float32).reshape(D, D) / (D * D)).double().requires_grad_(True)
    lg = logits.double().requires_grad_(True)
    full = lg
    if mode in ("e2e_hidden", "e2e_packed"):
        g = torch.Generator().manual_seed(7)
        lg = torch.randn(world * B_l, Tg, 12, generator=g).double().requires_grad_(True)   # the hidden states
        full = lg @ (0.5 * torch.randn(V, 12, generator=g)).double().t()
    out = O.ref_step_loss(q.double() @ w, p.double() @ w, full if mode.startswith("e2e") else None, ids, mask, qlen, 100)
    out["loss"].backward()

    assert abs(float(res[0]["loss_total"]) - float(out["loss"])) <= 1e-5 * abs(float(out["loss"]))
    assert abs(sum(float(r["loss_share"]) for r in res) - float(out["loss"])) <= 1e-5 * abs(float(out["loss"]))
    for r in range(world):  # every rank holds the same, fully reduced parameter gradient
        torch.testing.assert_close(res[r]["dw"].double(), w.grad, rtol=2e-4, atol=1e-6)
    if mode.startswith("e2e"):
        got = torch.cat([res[r]["dlogits"] for r in range(world)]).double()
        torch.testing.assert_close(got, lg.grad, rtol=2e-4, atol=1e-8)


def test_local_comm_is_identity():
    from dalm_amd.fused import GatherHandle, LocalComm

    c = LocalComm()
    t = torch.arange(6.0).reshape(3, 2)
    assert c.all_gather_rows(t) is t and c.all_reduce_sum_(t) is t
    assert torch.equal(GatherHandle(t, c).wait(), t)


def _bucket_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from dalm_amd.fused import TorchDistComm
    from dalm_amd.sharded import GradBucket

    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = TorchDistComm()
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 7, 3, 9, 4, 6)]   # same values on every rank
    bucket = GradBucket(ps, comm, bucket_bytes=40)                           # 10 floats per bucket -> 4 buckets
    assert [(b.lo, b.hi, b.count) for b in bucket.buckets] == [(0, 12, 2), (12, 24, 2), (24, 34, 2)]
    logs = []
    for step in range(2):
        # rank-dependent data; the LAST parameter is unused on step 0 (its bucket must be flushed by all_reduce),
        # and the graph touches the parameters in a different order on the two ranks' steps
        x = float(rank + 1 + step)
        order = [0, 1, 2, 3, 4] if step == 0 else [4, 2, 0, 5, 3, 1]
        loss = sum(((i + 1) * x) * (ps[i] * ps[i]).sum() for i in order)
        loss.backward()
        bucket.all_reduce()
        logs.append(list(bucket.last_launch_log))
        got = [p.grad.clone() for p in ps]
        sx = sum(float(r + 1 + step) for r in range(world))
        for i, p in enumerate(ps):
            want = 2 * (i + 1) * sx * p.detach() if i in order else torch.zeros_like(p)
            torch.testing.assert_close(got[i], want, rtol=1e-6, atol=1e-6)
        bucket.zero()
        assert float(bucket.flat.abs().sum()) == 0.0
    torch.save(logs, os.path.join(out_dir, f"log{rank}.pt"))
    dist.destroy_process_group()


def test_overlapped_grad_buckets_fixed_launch_order(tmp_path):
    """Hook-driven bucketed all-reduce: SUM over ranks in every view, unused parameters flushed, and the same
    launch order (descending bucket index) on every rank and every step whatever order the hooks fire in."""
    _spawn(_bucket_worker, 2, str(tmp_path))
    l0, l1 = torch.load(tmp_path / "log0.pt"), torch.load(tmp_path / "log1.pt")
    assert l0 == l1 == [[2, 1, 0], [2, 1, 0]]


def _accum_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from dalm_amd.fused import TorchDistComm
    from dalm_amd.training.step import _StepBase

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Linear(4, 3)                                  # same weights on every rank
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    step = _StepBase(model, opt, None, 100, comm=TorchDistComm(), grad_accum=2, track_grad_norm=True)
    assert step.bucket is not None and step.bucket.overlap is False   # N > 1: ONE all-reduce per optimizer step
    w0 = model.weight.detach().clone()
    seen = []
    for micro in range(4):
        x = torch.full((2, 4), float(rank + 1 + micro))
        loss = model(x).sum()
        step._finish(loss)
        seen.append((step.synced, model.weight.detach().clone()))
    torch.save({"seen": seen, "w0": w0, "grad_norm": float(step.grad_norm)}, os.path.join(out_dir, f"acc{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_accumulation_with_two_ranks_reduces_once_per_optimizer_step(tmp_path):
    """`--gradient_accumulation_steps 2` at W = 2 (reference: accelerate's accumulate(), train_rage2e.py:431): micro-batch
    gradients (each scaled 1/N) add up locally, the cross-rank SUM runs once, on the step that takes the update - an
    all-reduce per micro-batch would count the earlier micro-batches W times."""
    _spawn(_accum_worker, 2, str(tmp_path))
    r0, r1 = torch.load(tmp_path / "acc0.pt"), torch.load(tmp_path / "acc1.pt")
    assert [s for s, _ in r0["seen"]] == [False, True, False, True]
    # d(sum(model(x)))/dW[o, i] = sum_b x[b, i] = 2 * value; optimizer step 1 sees micro 0 + 1 of both ranks, each / 2
    w = r0["w0"].clone()
    for first in (0, 2):
        g = sum(2.0 * float(rank + 1 + micro) / 2.0 for rank in range(2) for micro in (first, first + 1))
        w = w - 0.5 * g
        for r in (r0, r1):
            torch.testing.assert_close(r["seen"][first][1], w + 0.5 * g)      # no update on the first micro-batch
            torch.testing.assert_close(r["seen"][first + 1][1], w)
    assert r0["grad_norm"] == r1["grad_norm"] > 0


def test_launcher_spawns_gloo_ranks(tmp_path):
    """dalm_amd.launch (the torchrun-free spawner bench.py and the trainers use for N > 1): two CPU ranks
    rendezvous over 127.0.0.1, rank 0 keeps stdout, a failing rank takes the job down with its exit code."""
    import subprocess

    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(dist.get_rank() + 1)])\n"
        "dist.all_reduce(t)\n"
        "print('SUM', float(t), os.environ['LOCAL_RANK'], os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'])\n"
        "dist.destroy_process_group()\n"
        "sys.exit(int(sys.argv[1]) if os.environ['RANK'] == '1' else 0)\n")
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    ok = subprocess.run([sys.executable, "-m", "dalm_amd.launch", "--nproc", "2", "--cpu", str(script), "0"],
                        capture_output=True, text=True, env=env, timeout=300)
    assert ok.returncode == 0, ok.stderr
    lines = [ln for ln in ok.stdout.splitlines() if ln.startswith("SUM")]    # (gloo prints a banner on stdout)
    assert lines == ["SUM 3.0 0 2 127.0.0.1"]                      # rank 0's stdout only
    assert "SUM 3.0 1 2 127.0.0.1" in ok.stderr                    # rank 1's stdout is folded into stderr
    bad = subprocess.run([sys.executable, "-m", "dalm_amd.launch", "--nproc", "2", "--cpu", str(script), "7"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert bad.returncode == 7 and "rank 1 exited with code 7" in bad.stderr
    # a rank that ignores SIGTERM (stuck in a collective) is killed after the grace period instead of hanging the launcher
    stuck = tmp_path / "stuck.py"
    stuck.write_text(
        "import os, signal, sys, time\n"
        "if os.environ['RANK'] == '0':\n"
        "    signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
        "    time.sleep(600)\n"
        "time.sleep(0.5); sys.exit(5)\n")
    from dalm_amd.launch import spawn_ranks
    import time as _t

    t0 = _t.time()
    rc = spawn_ranks([sys.executable, str(stuck)], 2, require_gpus=False, term_grace_s=1.0)
    assert rc == 5 and _t.time() - t0 < 60
    # asking for more GPU ranks than GPUs is refused before anything is spawned (no GPU in the CPU suite)
    if not torch.cuda.is_available():
        no = subprocess.run([sys.executable, "-m", "dalm_amd.launch", "--nproc", "2", str(script), "0"],
                            capture_output=True, text=True, env=env, timeout=300)
        assert no.returncode == 2 and "only 0 GPU(s) are visible" in no.stderr


def _ckpt_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from dalm_amd.training import common

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                   # replicas: identical parameters and optimizer state
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 7, 3, 9, 4)]
    opt = torch.optim.Adam(ps, lr=1e-3)
    for _ in range(3):
        sum((p * p).sum() for p in ps).backward()
        opt.step(); opt.zero_grad()
    saver = common.AsyncSaver()
    d = os.path.join(out_dir, "step_3")
    if rank == 0:
        # the directory is RE-USED - an earlier run (same output_dir, same world size) left a complete-looking
        # checkpoint behind.  Its files must be gone before rank 0 starts waiting for the new shards.
        os.makedirs(d, exist_ok=True)
        for name in ("optimizer-00000-of-00002.pt", "optimizer-00001-of-00002.pt"):
            torch.save({"state": {}, "param_groups": None, "num_state": 0, "stamp": 999}, os.path.join(d, name))
        torch.save({"optimizer": None, "sharded": 2, "stamp": 999, "scheduler": None, "extra": {"completed_steps": 999}},
                   os.path.join(d, "trainer_state.pt"))
    dist.barrier()
    common.save_training_state(d, None, opt, None, {"completed_steps": 3},
                               lambda path: open(os.path.join(path, "models_written_by_rank0"), "w").close(),
                               rank=rank, world=world, saver=saver, barrier=dist.barrier)
    with torch.no_grad():                                  # training moves on while the writer works: the snapshot must not see it
        for p in ps:
            p.add_(100.0)
        for st in opt.state.values():
            st["exp_avg"].add_(100.0)
    saver.wait()
    dist.barrier()
    if rank == 0:
        torch.save(opt.state_dict(), os.path.join(out_dir, "live.pt"))
    dist.destroy_process_group()


def test_sharded_async_checkpoint_roundtrip(tmp_path):
    """with W ranks every rank writes 1/W of the (replicated) optimizer state from a background thread;
    loading merges the shards; values are those at submit time, not whatever training did afterwards."""
    from dalm_amd.training import common

    _spawn(_ckpt_worker, 2, str(tmp_path))
    d = tmp_path / "step_3"
    assert sorted(p.name for p in d.iterdir()) == ["models_written_by_rank0", "optimizer-00000-of-00002.pt",
                                                   "optimizer-00001-of-00002.pt", "trainer_state.pt"]
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 7, 3, 9, 4)]
    ref = torch.optim.Adam(ps, lr=1e-3)
    for _ in range(3):
        sum((p * p).sum() for p in ps).backward()
        ref.step(); ref.zero_grad()
    fresh = torch.optim.Adam([torch.nn.Parameter(torch.zeros(n)) for n in (5, 7, 3, 9, 4)], lr=1e-3)
    extra = common.load_training_state(str(d), fresh, None)
    assert extra == {"completed_steps": 3}
    for k, st in ref.state_dict()["state"].items():
        got = fresh.state_dict()["state"][k]
        torch.testing.assert_close(got["exp_avg"], st["exp_avg"])
        torch.testing.assert_close(got["exp_avg_sq"], st["exp_avg_sq"])
    live = torch.load(tmp_path / "live.pt")                 # sanity: the live state had moved on by +100
    assert float((live["state"][0]["exp_avg"] - ref.state_dict()["state"][0]["exp_avg"]).mean()) > 99
    os.remove(d / "optimizer-00001-of-00002.pt")
    with pytest.raises(FileNotFoundError):
        common.load_training_state(str(d), fresh, None)
    # commit point: trainer_state.pt is written LAST, after every rank's shard exists - a directory with shards
    # but without it (crash between the two, or still being written) is refused with a clear message
    os.remove(d / "trainer_state.pt")
    with pytest.raises(RuntimeError, match="never committed"):
        common.load_training_state(str(d), fresh, None)
    assert not [p for p in d.iterdir() if ".tmp" in p.name]        # every file arrived by atomic rename


def test_rank0_does_not_commit_a_checkpoint_whose_shards_are_missing(tmp_path):
    """Rank 0 of a 2-rank job whose rank 1 never writes its shard: no trainer_state.pt appears, the writer reports it."""
    from dalm_amd.training import common

    ps = [torch.nn.Parameter(torch.randn(4))]
    opt = torch.optim.Adam(ps, lr=1e-3)
    (ps[0] * ps[0]).sum().backward()
    opt.step()
    d = tmp_path / "step_1"
    with pytest.raises(RuntimeError, match="did not appear"):
        common.save_training_state(str(d), None, opt, None, {"completed_steps": 1}, lambda path: None, rank=0, world=2,
                                   shard_wait_s=0.3, barrier=lambda: None)
    assert (d / "optimizer-00000-of-00002.pt").exists() and not (d / "trainer_state.pt").exists()
    with pytest.raises(ValueError, match="barrier"):
        common.save_training_state(str(d), None, opt, None, {"completed_steps": 1}, lambda path: None, rank=0, world=2)


def test_mixed_saves_are_refused_by_the_stamp(tmp_path):
    """A shard written at another step than the commit file (two saves mixed in one directory) does not load."""
    from dalm_amd.training import common

    ps = [torch.nn.Parameter(torch.randn(4)), torch.nn.Parameter(torch.randn(3))]
    opt = torch.optim.Adam(ps, lr=1e-3)
    sum((p * p).sum() for p in ps).backward()
    opt.step()
    d = tmp_path / "step_5"
    d.mkdir()
    sd = opt.state_dict()
    for r, stamp in ((0, 5), (1, 4)):
        sh = common.shard_optimizer_state(sd, r, 2)
        sh["stamp"] = stamp
        torch.save(sh, d / f"optimizer-{r:05d}-of-00002.pt")
    torch.save({"optimizer": None, "sharded": 2, "stamp": 5, "scheduler": None, "extra": {"completed_steps": 5}},
               d / "trainer_state.pt")
    with pytest.raises(RuntimeError, match="another step"):
        common.load_training_state(str(d), opt, None)


# ---------------------------------------------------------------------------------------------------------------------
# round 4: the native communicator is the W > 1 default - its rendezvous (unique id over a TCPStore, no id file) and the
# ALL-OR-NONE fallback agreement, on CPU with two processes and a fake communicator (no RCCL here)
# ---------------------------------------------------------------------------------------------------------------------
class _FakeComm:
    def __init__(self, rdv, fail_on=None):
        if fail_on == "rank0-early" and rdv.rank == 0:
            raise RuntimeError("rank 0 died before it could publish the unique id")
        self.uid = rdv.exchange(lambda: bytes([rdv.rank + 7]) * 128)       # only rank 0's lambda runs
        self.rank, self.fail_on, self.closed = rdv.rank, fail_on, False

    def self_test(self):
        if self.fail_on == self.rank:
            raise RuntimeError("self-test failed on this rank")
        if self.fail_on == "stall-1":
            # rank 1 stalls inside its first collective (it never returns, it does not raise); rank 0's collective waits for it
            # - what a hung ncclCommInitRank / all-reduce looks like from both sides
            import time

            time.sleep(3600)

    def close(self):
        self.closed = True


def _rdv_worker(rank, world, port, out_dir, fail_on):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.pop("DALM_COMM_ID_FILE", None)
    os.environ.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    if fail_on == "stall-1":
        os.environ["DALM_COMM_BRINGUP_TIMEOUT_S"] = "3"
    import warnings

    import torch.distributed as dist

    from dalm_amd.sharded import native_comm_or_none

    made = []

    def make(rdv):
        c = _FakeComm(rdv, fail_on)
        made.append(c)
        return c

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        comm = native_comm_or_none(rank, world, make=make)
    res = {"native": comm is not None, "uid0": made[0].uid[0] if made else None, "closed": made[0].closed if made else True}
    if comm is None:        # the fallback every rank takes together: MASTER_PORT must be free again for the process group
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        res["fallback_sum"] = float(t)
        dist.destroy_process_group()
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))


@pytest.mark.parametrize("fail_on", [None, 1, 0, "rank0-early", "stall-1"])
def test_native_comm_rendezvous_and_all_or_none_fallback(tmp_path, fail_on):
    """Both ranks get rank 0's unique id through the store; when ONE rank's communicator fails its self-test, BOTH ranks
    drop the native communicator (the healthy one is closed) and meet again in torch.distributed on the same port.
    "stall-1": the bring-up does not fail, it HANGS (both ranks sit in the first collective) - the watchdog
    (sharded._bring_up_with_deadline) times both out and both fall back together."""
    _spawn(_rdv_worker, 2, str(tmp_path), fail_on)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(2)]
    if fail_on == "stall-1":
        assert not any(r["native"] for r in res) and all(r["fallback_sum"] == 3.0 for r in res)
        return
    if fail_on == "rank0-early":       # rank 1 is told at once that no id will come (no 120 s timeout), both fall back
        assert not any(r["native"] for r in res) and all(r["fallback_sum"] == 3.0 for r in res)
        return
    assert all(r["uid0"] == 7 for r in res)                      # rank 0's payload (bytes of value 0 + 7) on both ranks
    if fail_on is None:
        assert all(r["native"] and not r["closed"] for r in res)
    else:
        assert not any(r["native"] for r in res) and all(r["closed"] for r in res)
        assert all(r["fallback_sum"] == 3.0 for r in res)


def test_bench_bring_up_path_with_two_gloo_ranks(tmp_path):
    """The calls `bench.py --gpus N` makes around its timed region: `sharded.init_distributed()` from the
    launcher's environment, `barrier`, `max_over_ranks` (the slowest rank's time), an all-gather of embeddings through the
    communicator it returned, and the one JSON line from rank 0 only - with two CPU ranks (gloo), as the driver launches N GPU ranks."""
    import json
    import subprocess

    script = tmp_path / "bring_up.py"
    script.write_text(
        "import json, os, torch\n"
        "from dalm_amd.sharded import barrier, init_distributed, max_over_ranks\n"
        "comm, dev = init_distributed()\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert comm.world_size == world == 2 and comm.rank == rank and dev.type == 'cpu'\n"
        "barrier(comm)\n"
        "slowest = max_over_ranks(comm, 1.0 + rank)\n"
        "rows = comm.all_gather_rows(torch.full((3, 4), float(rank)))\n"
        "assert rows.shape == (6, 4) and float(rows[:3].sum()) == 0.0 and float(rows[3:].sum()) == 12.0\n"
        "barrier(comm)\n"
        "if rank == 0:\n"
        "    print(json.dumps({'value': 1.0, 'n_gpus': world, 'slowest': slowest,\n"
        "                      'config': {'ranks_seen_by_process_group': comm.world_size, 'collective_backend': 'gloo'}}), flush=True)\n"
        "import torch.distributed as dist\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    env.pop("DALM_NATIVE_COMM", None)
    run = subprocess.run([sys.executable, "-m", "dalm_amd.launch", "--nproc", "2", "--cpu", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stderr
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                          # one JSON line, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["slowest"] == 2.0 and out["config"]["ranks_seen_by_process_group"] == 2
