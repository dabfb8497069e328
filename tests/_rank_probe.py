"""Rank program for tests: started N times by waveforms_amd.bert.spawn_ranks (torchrun env).
Joins a gloo group, runs its shard of a sweep with a deterministic stand-in for the GPU block
(or, with --gpu, the real device-resident link with every rank on cuda:0) and lets rank 0
print the reduced counter table as one JSON line."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--nsym", type=int, default=1000)
    ap.add_argument("--blocks", type=int, default=5)
    a = ap.parse_args()
    import torch.distributed as dist

    from waveforms_amd.bert import SweepPlan, ber_sweep, dist_env

    rank, world, _ = dist_env()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = SweepPlan(ebn0_db=[0.0, 4.0, 8.0], blocks_per_point=a.blocks, nsym=a.nsym)
    runner = None
    if not a.gpu:
        def runner(plan):
            acc = np.zeros((3, 3), dtype=np.int64)

            def run(point, block):
                acc[point] += (100 * point + block, 10 * point + block, plan.nsym)

            return run, lambda: acc
    else:
        import torch

        torch.cuda.set_device(0)
    res = ber_sweep(plan, runner=runner)
    if rank == 0:
        print(json.dumps({"world": world, "counts": res.tolist()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
