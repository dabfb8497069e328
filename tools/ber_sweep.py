"""SOQPSK-TG BER sweep on the device-resident link (BASELINE config 4).

    python tools/ber_sweep.py [--ebn0 0:12] [--symbols-per-point 1e8] [--block 10000000] [--detector PT]
    python tools/ber_sweep.py --gpus 8 ...          (starts the 8 ranks itself, as child processes)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/ber_sweep.py ...

Independent trial blocks are dealt round-robin to the ranks; the only collective is one
all-reduce of the error-counter table at the end.  Rank 0 prints the table, the Eb/N0 at
BER 1e-3 / 1e-4 and (if tests/golden/ber_golden*.csv are present) the offset in dB from the
reference's curve.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def golden_curve(detector: str):
    """(ebn0, ber, bit_errors, symbols) per point from the committed reference runs."""
    import csv

    acc = {}
    key = "pt" if detector == "PT" else "pam"
    for name in ("ber_golden.csv", "ber_golden_hi.csv", "ber_golden_hi2.csv"):
        path = ROOT / "tests" / "golden" / name
        if not path.exists():
            continue
        for r in csv.DictReader(open(path)):
            a = acc.setdefault(int(r["ebn0_db"]), [0, 0])
            a[0] += int(r[f"{key}_compared"])
            a[1] += int(r[f"{key}_bit_err"])
    e = sorted(acc)
    return (np.array(e, dtype=float), np.array([acc[k][1] / acc[k][0] for k in e]),
            np.array([acc[k][1] for k in e]), np.array([acc[k][0] for k in e]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ebn0", default="0:12")
    ap.add_argument("--symbols-per-point", type=float, default=1e8,
                    help="BASELINE configs[3] says 1e8: 1.3e9 symbols = 0.065 s of ONE MI355X, so on 8 GPUs a pass is 8 ms of work "
                         "per rank against seconds of process start and RCCL bootstrap (reported as init_seconds) — for a "
                         "strong-scaling curve that measures the GPUs use >= 1e10 per point (>= 6.5 s of work in all)")
    ap.add_argument("--block", type=int, default=10_000_000, help="symbols per trial block (1e7: BASELINE configs[1]'s block, 10 per point)")
    ap.add_argument("--detector", default="PT")
    ap.add_argument("--waveform", default="soqpsk", choices=["soqpsk", "multih", "pcmfm"])
    ap.add_argument("--states", type=int, default=16, choices=[16, 64, 256], help="--waveform multih: the 16-state design of BASELINE configs[2], the 64-state one or the full 256-state trellis")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--gpus", type=int, default=0, help="N > 1 without torchrun: start N ranks as child processes")
    ap.add_argument("--passes", type=int, default=2, help="run the sweep this many times; `seconds` is the last pass")
    ap.add_argument("--json-out", default=None, help="rank 0 also writes its result object to this file")
    a = ap.parse_args()
    lo, hi = (int(v) for v in a.ebn0.split(":"))
    ebn0 = list(range(lo, hi + 1))

    from waveforms_amd.bert import SweepPlan, ber_sweep, ber_table, ebn0_at_ber, init_ranks, spawn_ranks

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python tools/ber_sweep.py --gpus N`: the ranks are child processes of this one,
        # started before anything here has touched the GPU
        raise SystemExit(spawn_ranks(str(Path(__file__).resolve()), a.gpus, sys.argv[1:]))
    import torch

    t_start = time.perf_counter()
    # WF_BENCH_REHEARSAL=1: every rank on cuda:0, gloo collectives (N > 1 on a one-GPU box)
    rank, world, dist, coll_dev = init_ranks(rehearsal=os.environ.get("WF_BENCH_REHEARSAL") == "1")
    init_seconds = time.perf_counter() - t_start       # process-group rendezvous (+ RCCL bootstrap): reported apart from the sweep
    blocks = max(1, int(round(a.symbols_per_point / a.block)))
    plan = SweepPlan(ebn0_db=ebn0, blocks_per_point=blocks, nsym=a.block, seed=a.seed, detector=a.detector, waveform=a.waveform, states=a.states)
    # two passes: the first also pays the process's first-use costs (workspace allocations of ~0.8 GB per block in
    # flight, code-object loads, table uploads); `seconds` is the second pass, the first is reported beside it
    passes = []
    for _ in range(max(1, a.passes)):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()               # every rank starts the pass together ...
        t0 = time.perf_counter()
        counts = ber_sweep(plan)
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        if dist is not None:             # ... and the pass lasts as long as its slowest rank
            t = torch.tensor([mine], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            mine = float(t.item())
        passes.append(mine)
    dt = passes[-1]
    if rank == 0:
        tab = ber_table(ebn0, counts)
        for row in tab:     # errors are counted per symbol and per bit; symbols carry bits_per_symbol bits
            row["ber"] = row["bit_errors"] / max(row["symbols"] * plan.bits_per_symbol, 1)
        for row in tab:
            print(f"Eb/N0 {row['ebn0_db']:5.1f} dB  symbols {row['symbols']:>12d}  SER {row['ser']:.3e}  BER {row['ber']:.3e}")
        out = {"detector": a.detector, "n_gpus": world, "seconds": round(dt, 3), "seconds_first_pass": round(passes[0], 3),
               "seconds_are": "barrier before the clock, max over ranks", "init_seconds": round(init_seconds, 3),
               "collective_backend": dist.get_backend() if dist is not None else None, "world_seen": dist.get_world_size() if dist is not None else 1,
               "Msym_per_s": round(int(counts[:, 2].sum()) / dt / 1e6, 1), "table": tab}
        ber = [r["ber"] for r in tab]
        ge, gb = (np.array([]), np.array([]))
        if a.waveform == "soqpsk":
            ge, gb, _, _ = golden_curve(a.detector)
        else:
            # no reference detector exists for these waveforms: the yardstick is the minimum-distance term
            # Q(sqrt(d^2 Eb/N0)) with d^2 recomputed from the reference's pulse (ARTM CPM: 1.2957)
            import math
            d2 = {"multih": 1.2957297551846658}.get(a.waveform)
            if d2:
                out["min_distance_bound"] = [0.5 * math.erfc(math.sqrt(d2 * 10 ** (e / 10.0) / 2.0)) for e in ebn0]
            out["detector"] = f"generic CPM trellis detector ({a.waveform}{f', {a.states} states' if a.states != 16 else ''})"
        for target in (1e-3, 1e-4):
            try:
                mine = ebn0_at_ber(ebn0, ber, target)
                out[f"ebn0_at_{target:g}"] = round(mine, 4)
                if ge.size:
                    out[f"delta_db_vs_reference_at_{target:g}"] = round(mine - ebn0_at_ber(ge, gb, target), 4)
            except ValueError:
                pass
        out["counts"] = [[int(v) for v in row] for row in counts]
        print(json.dumps(out))
        if a.json_out:
            Path(a.json_out).write_text(json.dumps(out) + "\n")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
