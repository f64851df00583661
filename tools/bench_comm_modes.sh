#!/bin/bash
# W > 1 code path on ONE GPU (a live one-rank RCCL group): the step through each communicator / launch mode, next to the
# single-process whole-step hipGraph.  -> gpurun_out/<tag>_comm_modes.txt   (bash tools/bench_comm_modes.sh r04)
F=gpurun_out/${1:-r04}_comm_modes.txt
run() { echo "## $1"; shift; env "$@" 2>&1 | grep '^{"metric"' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); c=d['config']
    print(f\"   {d['value']:.2f} pairs/s  {d['ms_per_step']:.2f} ms/step   launch: {c['launch']}   backend: {c['collective_backend']}   hw_queues: {c['gpu_max_hw_queues']}\")
"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc"
{
echo "# bench.py cfg3 on one MI355X, $(date -u +%F): communicator / launch modes of the W > 1 code path with one rank"
run "single process, LocalComm, whole-step hipGraph (the headline mode)" $B
run "DALM_FORCE_DIST=1 (the W > 1 DEFAULT since round 4): the library's own RCCL binding dalm_comm_*_on, TCPStore rendezvous + self-test, graphed towers" DALM_FORCE_DIST=1 $B
run "DALM_FORCE_DIST=1 DALM_NATIVE_COMM=0: torch.distributed(nccl = RCCL), graphed towers + eager collectives/loss/optimizer" DALM_FORCE_DIST=1 DALM_NATIVE_COMM=0 $B
run "DALM_FORCE_DIST=1 DALM_NATIVE_COMM=1 --graph-collectives: whole step incl. the RCCL collectives in ONE hipGraph" DALM_FORCE_DIST=1 DALM_NATIVE_COMM=1 $B --graph-collectives
} > $F 2>&1
cat $F
