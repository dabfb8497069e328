"""Monte-Carlo BER sweep harness (the reference's empty ``waveforms/bert`` placeholder is
where it intended this to live).

Unit of work = one independent trial block (Eb/N0 point x block of PRBS bits): block ``b``
transmits PN bits ``[b*nsym, (b+1)*nsym)`` and draws its noise from Philox subsequence
``point * 2**32 + b``.  Blocks are dealt round-robin to the ranks (one process per GPU);
there is NO data-path collective — the only exchange is one all-reduce(SUM) of the
int64 counter table at the end (RCCL over xGMI when the backend is ``nccl``; ``gloo`` in
the CPU tests).  A failed shard is idempotent: re-run the same (seed, point, block).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Callable, Sequence

import numpy as np


@dataclass
class SweepPlan:
    ebn0_db: Sequence[float]
    blocks_per_point: int
    nsym: int                      # symbols per block
    seed: int = 1
    detector: str = "PT"
    sps: int = 8
    pn_degree: int = 23
    waveform: str = "soqpsk"       # "soqpsk" (detector PT / PAM), or "multih" / "pcmfm" through the generic CPM detector
    warmup: int = 0                # detector chunk warm-up (0: per point, waveforms_amd.link.operating_point_warmup): speed only, never the counts
    states: int = 16               # waveform "multih": 16 (ARTM_16, BASELINE configs[2]), 64 (ARTM_64: every phase state of the 2-symbol pulse) or 256 (ARTM_256: the full trellis, notes/cpm/cpm.md:128-140)
    jobs: list[tuple[int, int]] = field(default_factory=list)   # (point index, block index)

    def __post_init__(self):
        # block-major so that every rank touches every Eb/N0 point (even tail latency)
        self.jobs = [(p, b) for b in range(self.blocks_per_point) for p in range(len(self.ebn0_db))]

    def shard(self, rank: int, world: int) -> list[tuple[int, int]]:
        return self.jobs[rank::world]

    def stream_id(self, point: int, block: int) -> int:
        return (point << 32) | block

    @property
    def bits_per_symbol(self) -> int:
        return 2 if self.waveform == "multih" else 1

    def skip_bits(self, block: int) -> int:
        return block * self.nsym * self.bits_per_symbol


def dist_env() -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment (1 process per GPU)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def spawn_ranks(script: str, ngpus: int, argv: Sequence[str]) -> int:
    """Start ``ngpus`` rank processes of ``script`` (one per GPU) under ``torch.distributed.run``
    as CHILD processes and return their exit status.  For entry points that were started as a
    plain ``python script --gpus N`` with no rendezvous in the environment.  The caller must not
    have touched the GPU: the parent only waits (a process that has initialised HIP must never
    be replaced by exec on these machines, and is never needed here); the ranks' stdout is
    inherited, so rank 0's result line is the caller's."""
    import socket
    import subprocess
    import sys

    with socket.socket() as s:   # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(ngpus)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]
    return subprocess.run(cmd, env=env).returncode


def init_ranks(rehearsal: bool = False):
    """Join the job's process group (if WORLD_SIZE > 1) and select this rank's GPU.  Returns
    (rank, world, dist-or-None, device for collectives).  ``rehearsal``: every rank on cuda:0 and
    gloo for the collectives, so the N > 1 path can be exercised on a one-GPU box."""
    import torch

    rank, world, local = dist_env()
    torch.cuda.set_device(0 if rehearsal else local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    return rank, world, dist, ("cpu" if rehearsal else "cuda")


def gpu_block_runner(plan: SweepPlan, streams: int | None = None, fuse: int | None = None):
    """Default runner: device-resident links; returns (run(point, block), finish()).
    Consecutive trial blocks overlap on the GPU either INSIDE one link (`fuse` bit 5: a block's detector and error
    count on the context's side stream, beside the next block's front end) or as `streams` blocks in flight on
    separate HIP streams (own workspace, own wf_ctx, own counter table each).  Defaults, from same-box runs of the
    13 x 1e8-symbol sweep (tools/sweep_lanes.py; identical counts in every arrangement): SOQPSK one lane with the
    pipelined link (0.0638 s; three lanes of sequential links 0.0685, three lanes of pipelined links 0.076); ARTM three
    lanes of sequential links (0.165 - 0.169 s; one pipelined lane 0.168 - 0.172), PCM/FM — since the lane detector and the
    3 dB floor of its operating-point warm-up — one pipelined lane (0.089 - 0.092 s; three sequential lanes 0.093 - 0.100;
    round 3, row-form detector: 0.107): profiles/r04_sweep_lanes.log."""
    import ctypes

    from waveforms_amd import _hip, device as dev
    from waveforms_amd.link import CPMLink, SOQPSKLink, sigma_for_ebn0
    from waveforms_amd.viterbi.cpm import sigma_for_ebn0 as cpm_sigma

    torch = _hip.torch()
    cpm = plan.waveform != "soqpsk"
    # (SOQPSK blocks under 2^23 symbols: three lanes again — 2^22-symbol blocks 0.0735 s against 0.0762 for one pipelined lane;
    #  1e7-symbol blocks, which the front-end kernel's launch shape is tuned for, are the fastest way to run the sweep)
    # (round 6: ARTM's 16-state link with the matched filters inside the detector — fuse bit 7, blocks of >= 2^21 calls — pipelines
    #  like PCM/FM's: one lane, 0.85 ms per 1e7-symbol block against 1.2 for three lanes of the rows form; the 256-state link in that
    #  form runs its blocks one after the other on one lane: whatever shares the chip with its detector costs more than it hides)
    artm16 = plan.waveform == "multih" and plan.states == 16 and plan.nsym >= 6_000_000
    artm256 = plan.waveform == "multih" and plan.states == 256
    one_pipelined = (not cpm and plan.nsym >= (1 << 23)) or (plan.waveform == "pcmfm" and plan.nsym >= 6_500_000) or artm16
    n = max(1, int(streams)) if streams is not None else (1 if one_pipelined or artm256 else 3)
    if cpm:
        from waveforms_amd.viterbi.cpm import ARTM_64, ARTM_256

        big = {64: ARTM_64, 256: ARTM_256}.get(plan.states) if plan.waveform == "multih" else None
        links = [CPMLink(plan.nsym, plan.sps, waveform=plan.waveform, spec=big, pn_degree=plan.pn_degree, private_ctx=n > 1,
                         warmup=plan.warmup, fuse=((42 if n == 1 and one_pipelined else 10) | (128 if plan.waveform == "multih" else 0)) if fuse is None else fuse)
                 for _ in range(n)]
        run_fn = _hip.lib().wf_cpm_link_run
    else:
        # (several lanes: sequential links with fuse bit 4 — the link's one-launch PRBS + precoder kernel is few, long
        #  workgroups, the right trade for ONE block at a time; with three blocks in flight the generic kernels fill the
        #  machine better: 13 x 1e8 symbols in 0.074 - 0.076 s against 0.078 - 0.079)
        links = [SOQPSKLink(plan.nsym, plan.sps, detector=plan.detector, pn_degree=plan.pn_degree, private_ctx=n > 1, warmup=plan.warmup,
                            fuse=(31 if n > 1 else 47) if fuse is None else fuse)
                 for _ in range(n)]
        run_fn = _hip.lib().wf_link_run
    lanes = [torch.cuda.Stream() for _ in range(n)] if n > 1 else [torch.cuda.current_stream()]
    npts = len(plan.ebn0_db)
    # Counters of a lane accumulate on the device.  Every launch of the chunk-parallel detectors proves on the device
    # that each chunk started from bitwise the sequential detector's state and repairs the chunks that did not
    # (cascading where needed), so a block's counts never depend on the warm-up it ran with; the proof word of each
    # lane's context is read once, at the end, as a consistency check (non-zero only with the repairs switched off).
    tables = [_hip.zeros((npts, 2), "int64") for _ in range(n)]
    compared = np.zeros(npts, dtype=np.int64)
    issued = [0]
    stats = {"repaired_chunks": 0, "cascaded_chunks": 0}
    if n > 1:   # the tables were zeroed on the current stream
        torch.cuda.current_stream().synchronize()

    # no warm-up asked for: each Eb/N0 point runs at its own (waveforms_amd.link.operating_point_warmup; 0 = the library's
    # default where the table has no shorter one)
    from waveforms_amd.link import operating_point_warmup, soqpsk_warmup_param
    point_warmup = [(0 if plan.states != 16 else operating_point_warmup(plan.waveform, float(e))) if cpm
                    else soqpsk_warmup_param(operating_point_warmup("soqpsk", float(e))) for e in plan.ebn0_db]

    def launch(k: int, point: int, block: int, warmup: int) -> int:
        link, c = links[k], links[k].cfg
        c.sigma = (cpm_sigma(plan.ebn0_db[point], plan.sps, plan.bits_per_symbol) if cpm
                   else sigma_for_ebn0(plan.ebn0_db[point], plan.sps))
        c.seed = plan.seed
        c.stream_id, c.skip, c.event_slot = plan.stream_id(point, block), plan.skip_bits(block), -1
        keep, c.warmup = c.warmup, warmup
        m = ctypes.c_int64(0)
        try:
            with torch.cuda.stream(lanes[k]):
                _hip.check(run_fn(link._ctx, ctypes.byref(c), link.workspace.data_ptr(), link.workspace_bytes,
                                  tables[k].data_ptr() + 16 * point, ctypes.byref(m), _hip.stream()))
        finally:
            c.warmup = keep
        return m.value

    def run(point: int, block: int) -> None:
        k = issued[0] % n
        issued[0] += 1
        compared[point] += launch(k, point, block, links[k].cfg.warmup or point_warmup[point])

    def finish() -> np.ndarray:
        out = np.zeros((npts, 3), dtype=np.int64)
        for k in range(n):
            with torch.cuda.stream(lanes[k]):
                _hip.check(_hip.lib().wf_ctx_check(links[k]._ctx, _hip.stream()))
                unproven = dev.viterbi_unmerged(reset=True, ctx=links[k]._ctx)
                if unproven:
                    raise RuntimeError(f"{unproven} detector chunk(s) were left unproven (the repairs are switched off on this context)")
                stats["repaired_chunks"] += dev.viterbi_repaired(reset=True, ctx=links[k]._ctx)
                stats["cascaded_chunks"] += dev.viterbi_cascaded(reset=True, ctx=links[k]._ctx)
                out[:, :2] += tables[k].cpu().numpy()
        out[:, 2] = compared
        return out

    run.stats = stats                   # type: ignore[attr-defined]
    run.links = links                   # type: ignore[attr-defined]
    return run, finish


def ber_sweep(plan: SweepPlan, rank: int | None = None, world: int | None = None,
              runner: Callable | None = None, reduce: bool = True) -> np.ndarray:
    """Run this rank's shard and (if a process group exists) all-reduce the counters.

    Returns int64[n_points, 3]: symbol errors, bit errors, symbols compared — identical
    on every rank after the reduce.
    """
    env_rank, env_world, _ = dist_env()
    rank = env_rank if rank is None else rank
    world = env_world if world is None else world
    run, finish = (runner or gpu_block_runner)(plan)
    for point, block in plan.shard(rank, world):
        run(point, block)
    local = finish()
    return all_reduce_counts(local) if reduce and world > 1 else local


def all_reduce_counts(local: np.ndarray) -> np.ndarray:
    """SUM-all-reduce an int64 counter table across the default process group."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return local
    backend = dist.get_backend()
    t = torch.from_numpy(np.ascontiguousarray(local))
    if backend == "nccl":  # RCCL: the tensor must live on this rank's GPU
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def ber_table(ebn0_db: Sequence[float], counts: np.ndarray) -> list[dict]:
    return [dict(ebn0_db=float(e), symbols=int(c[2]), sym_errors=int(c[0]), bit_errors=int(c[1]),
                 ser=float(c[0]) / max(int(c[2]), 1), ber=float(c[1]) / max(int(c[2]), 1))
            for e, c in zip(ebn0_db, counts)]


def ebn0_at_ber(ebn0_db: Sequence[float], ber: Sequence[float], target: float) -> float:
    """Eb/N0 (dB) where log10(BER) crosses log10(target), linear interpolation in dB."""
    e = np.asarray(ebn0_db, dtype=np.float64)
    y = np.log10(np.maximum(np.asarray(ber, dtype=np.float64), 1e-300))
    t = np.log10(target)
    for k in range(len(e) - 1):
        if y[k] >= t >= y[k + 1] and y[k] != y[k + 1]:
            return float(e[k] + (y[k] - t) * (e[k + 1] - e[k]) / (y[k] - y[k + 1]))
    raise ValueError(f"BER curve does not cross {target}")
