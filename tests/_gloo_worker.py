"""Worker for tests/test_shard_gloo.py: one rank of the N/P sharded step, on the CPU, over gloo.

Mirrors the data movement of the sharded pipeline (nbody_amd/csrc/pipeline.hip: shard plan from the C
library, own massive + massless slices as receivers, all-gather of the new massive positions every step,
re-assembly into partitioned order) with the ORACLE doing the arithmetic -- this is a test of the plan and
of the exchange pattern, not a product path.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import nbody_amd as nb          # noqa: E402  (only nb_hip_shard_plan: pure host code)
import oracle_binding as ob     # noqa: E402


def main():
    out_path, n, steps, dt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    ic = np.fromfile(os.path.join(HERE, "golden", f"ic_{n}.bin"), dtype=np.float32).reshape(-1, 8)
    part, M = ob.partition(ic)
    N = part.shape[0]
    plan = nb.shard_plan(N, M, rank, world)
    Mc = plan["mass_chunk"]

    # this rank's receivers: its massive slice and its massless slice (global indices)
    own_m = np.arange(plan["mass_begin"], plan["mass_begin"] + plan["mass_count"])
    own_z = np.arange(plan["zero_begin"], plan["zero_begin"] + plan["zero_count"])
    mine_m, mine_z = part[own_m].copy(), part[own_z].copy()
    sources = part[:M].copy()   # every rank starts from the full array, like SetSimulationData

    for _ in range(steps):
        # local world = all sources (as the gathered array) + own massless; the oracle steps it, we keep only
        # what this rank owns -- a receiver's result depends on the sources alone (Jacobi), so this equals
        # stepping just the owned receivers against the gathered sources
        local = np.concatenate([sources, mine_z]) if len(mine_z) else sources.copy()
        stepped = ob.step(local, M, dt, 1)
        mine_m = stepped[own_m]
        mine_z = stepped[M:]
        # uniform-count all-gather of the new massive slices (padded to Mc rows like the device buffer)
        send = torch.zeros((Mc, 8), dtype=torch.float32)
        send[:len(mine_m)] = torch.from_numpy(mine_m)
        recv = [torch.zeros((Mc, 8), dtype=torch.float32) for _ in range(world)]
        dist.all_gather(recv, send)
        gathered = torch.cat(recv).numpy()          # index q*Mc + i == global massive index
        sources = gathered[:M].copy()

    # re-assembly (GetSimulationData of the sharded pipeline): gather massless slices too
    Zc = plan["zero_chunk"]
    send = torch.zeros((Zc, 8), dtype=torch.float32)
    send[:len(mine_z)] = torch.from_numpy(mine_z)
    recv = [torch.zeros((Zc, 8), dtype=torch.float32) for _ in range(world)]
    dist.all_gather(recv, send)
    full = np.zeros_like(part)
    full[:M] = sources
    for q in range(world):
        pq = nb.shard_plan(N, M, q, world)
        full[pq["zero_begin"]:pq["zero_begin"] + pq["zero_count"]] = recv[q].numpy()[:pq["zero_count"]]
    if rank == 0:
        full.tofile(out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
