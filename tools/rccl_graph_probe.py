#!/usr/bin/env python
"""Can this stack capture RCCL collectives in a hipGraph and replay them?  Run as one process per GPU
(`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 ... tools/rccl_graph_probe.py`, or plainly for one rank).

bench.py spawns it (with its own rendezvous port and a timeout) before it decides whether the N > 1 mapper iteration is captured as a graph
or driven eagerly: a capture that hangs must not take the benchmark down with it.  Exit code 0 = an all_gather_into_tensor and an all_reduce
were captured, replayed twice and produced the right values on this rank; the JSON line also carries the replay time.
"""
import json
import os
import sys
import time


def main():
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n = 1 << 20
    x = torch.full((n,), float(rank + 1), device=dev)
    mine = torch.full((n // 4,), float(rank + 1), device=dev)
    gathered = torch.zeros((world, n // 4), device=dev)
    for _ in range(2):                      # eager warm-up: communicator setup happens outside the capture
        dist.all_reduce(x)
        dist.all_gather_into_tensor(gathered, mine)
    torch.cuda.synchronize()
    x.fill_(float(rank + 1))
    # thread_local capture + a drained watchdog: ProcessGroupNCCL's watchdog thread polls the warm-up collectives with hipEventQuery, which a
    # GLOBAL-mode capture forbids from any thread (gs_icp_slam_amd/graph.py: drain_process_group_watchdog)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gs_icp_slam_amd.graph import capture_mode, drain_process_group_watchdog
    drain_process_group_watchdog(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode=capture_mode()):
        y = x * 2.0
        dist.all_reduce(y)
        dist.all_gather_into_tensor(gathered, mine)
        z = y + gathered.sum(0).repeat(4)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        graph.replay()
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / reps
    s = world * (world + 1) / 2.0
    expect = 2.0 * s + s
    ok = bool(torch.all(z == expect).item())
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ok = bool(flag.item() == 1.0)
    if rank == 0:
        print(json.dumps({"rccl_graph_capture": ok, "world_size": world, "replay_us": round(us, 1), "bytes_all_reduce": 4 * n,
                          "bytes_all_gather_per_rank": n}))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
