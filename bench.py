"""Benchmark: SOQPSK-TG modulate + AWGN + matched filter + 4-state Viterbi detect @ 8 sps.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one block of synthetic PN23 bits, entirely in HBM
(`wf_link_run`: PRBS -> encode -> upsample+FIR -> phase scan+cexp -> derotate+AWGN ->
decimating PT bank -> Viterbi -> error count).  N > 1: one process per GPU under torchrun,
every rank runs its own independent trial blocks (weak scaling, no data-path collective);
the only exchange is the final all-reduce of the error counters (RCCL over xGMI).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# algorithmic HBM bytes per symbol of each materialising stage at sps = 8, fp64 / complex128
# (DESIGN.md "Kernels"): what the stage must read + write at minimum.
def stage_bytes_per_symbol(sps: int, nfilt: int = 3) -> dict:
    return {
        "prbs": 1, "encode": 2,
        "fir": 1 + 8 * sps,                 # i8 symbol in, sps f64 out
        "phase": 8 * sps + 16 * sps,        # f64 in, c128 out
        "awgn": 16 * sps + 16 * sps,        # c128 in, c128 out
        "mfbank": 16 * sps + 16 * nfilt,    # c128 in, nfilt c128 per symbol out
        "viterbi": 16 * nfilt + 2,          # 3 c128 in, bit + symbol out
        "count": 4,
    }


HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)

# stage -> kernel that dominates it (names as rocprofv3 prints them)
STAGE_KERNEL = {"fir": "fir_kernel<9>", "phase": "phase_kernel", "modulate": "mod_main_kernel<9, true, false>",
                "awgn": "awgn_kernel", "mfbank": "mf_bank_kernel<3, false, 8, 9>", "awgn+mfbank": "mf_bank_kernel<3, true, 8, 9>",
                "viterbi": "viterbi_batch_kernel<false>", "count": "count_errors_kernel", "prbs": "lfsr_kernel",
                "encode": "enc_reduce_kernel"}


def profile_record(stage: str, nsym: int, sps: int):
    """The committed rocprofv3 record of the stage's kernel (profiles/*_summary.json, produced by
    tools/profile.sh + tools/pmc_summary.py from separate `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and
    `--pmc SQ_*` passes over this same command): HBM bytes and VALU instructions per launch.  The
    newest summary that matches the workload wins; `current` tells whether it was taken from the
    kernels as they are built now (source digest of csrc/)."""
    try:
        from waveforms_amd.csrc.build import _digest
        now = _digest()
    except Exception:   # noqa: BLE001
        now = None
    best = None
    for path in sorted((ROOT / "profiles").glob("*_summary.json")):
        try:
            d = json.loads(path.read_text())
        except (OSError, ValueError):
            continue
        k = d.get("kernels", {}).get(STAGE_KERNEL.get(stage, ""))
        if d.get("nsym") == nsym and d.get("sps") == sps and k:
            rec = {"traffic": int(k["hbm_traffic_bytes"]), "source": path.name, "valu_insts": k.get("valu_insts"),
                   "shader_cycles": k.get("shader_cycles"), "avg_ns": k.get("avg_ns"),
                   "valu_active_quad_cycles": k.get("valu_active_quad_cycles"),
                   "mfma_insts": k.get("mfma_f64_insts"), "mfma_busy_cycles": k.get("mfma_busy_cycles"),
                   "mfma_pass_shader_cycles": k.get("mfma_pass_shader_cycles"),
                   "current": bool(now and d.get("build_digest") == now)}
            if best is None or rec["current"] or not best["current"]:
                best = rec
    return best


def _current_digest():
    try:
        from waveforms_amd.csrc.build import _digest
        return _digest()
    except Exception:   # noqa: BLE001
        return None


def valu_issue(stage: str, rec, launch_ms: float):
    """Vector-issue side of the roofline for kernels whose HBM traffic is already minimal: measured VALU
    wave-instructions per launch (SQ_INSTS_VALU) x average issue cycles per instruction (static class mix of
    the kernel, per-class costs measured by tools/valu_probe.hip) / (1024 SIMDs x shader cycles of the launch).
    The shader cycles are those of the SAME counter pass (GRBM_GUI_ACTIVE / 8 XCDs), so no clock is assumed;
    profiles without that counter fall back to 2.2 GHz x the live launch time."""
    mix, mix_name = None, None
    for path in sorted((ROOT / "profiles").glob("r*_valu_mix.json")):       # newest round last
        try:
            mix, mix_name = json.loads(path.read_text()), path.name
        except (OSError, ValueError):
            continue
    if mix is None:
        return None
    k = mix["kernels"].get(STAGE_KERNEL.get(stage, ""))
    if not (k and rec and rec.get("valu_insts") and launch_ms > 0):
        return None
    # mean issue cost of what the loops execute (static mix weighted by loop depth), not of the whole kernel text
    avg = k.get("avg_cycles_per_valu_loop", k["avg_cycles_per_valu"])
    # (SQ_INSTS_VALU counts the matrix instructions too: they are priced below by the cycles the matrix pipe was busy)
    cycles = (rec["valu_insts"] - (rec.get("mfma_insts") or 0)) * avg
    shader_cycles = rec.get("shader_cycles") or 2.2e9 * launch_ms * 1e-3
    frac = cycles / (mix["simds"] * shader_cycles)
    out = {"valu_issue_frac": round(frac, 4), "valu_wave_insts_per_launch": rec["valu_insts"],
           "avg_issue_cycles_per_valu": avg, "avg_issue_cycles_whole_kernel_text": k["avg_cycles_per_valu"],
           "mix": "static class mix weighted by loop depth (tools/valu_mix.py)" if "avg_cycles_per_valu_loop" in k else "static, whole kernel",
           "class_mix_static": {c: k[c] for c in ("full", "fast", "trans64", "trans32") if c in k},
           "simds": mix["simds"], "issue_cycles_per_launch": int(cycles), "shader_cycles_per_launch": int(shader_cycles),
           "shader_cycles_source": "GRBM_GUI_ACTIVE / 8 in the profile's counter pass" if rec.get("shader_cycles") else "2.2 GHz x live launch time",
           "mix_matches_build": bool(mix.get("build_digest") and mix.get("build_digest") == _current_digest()),
           "source": f"{rec['source']} + {mix_name} + {mix.get('probe', 'r02_valu_probe.json')}"}
    if rec.get("valu_active_quad_cycles") and rec.get("shader_cycles"):
        # the same quantity straight from the counters of that profile (no instruction-class model): SQ_ACTIVE_INST_VALU counts
        # quad-cycles, x 4 = SIMD cycles with a vector instruction in execution, over 1024 SIMDs x the launch's shader cycles
        out["valu_active_frac_counter"] = round(4.0 * rec["valu_active_quad_cycles"] / (mix["simds"] * rec["shader_cycles"]), 4)
    if rec.get("shader_cycles") and rec.get("avg_ns"):
        out["shader_clock_ghz_in_profile"] = round(rec["shader_cycles"] / rec["avg_ns"], 3)
    if rec.get("mfma_insts") and rec.get("mfma_busy_cycles") and rec.get("mfma_pass_shader_cycles"):
        # kernels with fp64 matrix tiles (PAM bank, ARTM bank): the matrix pipe's busy cycles of its own counter pass.  A
        # SIMD does not issue fp64 vector work of its other waves under a running v_mfma_f64 (profiles/r03_mfma_f64_probe.json,
        # the PAM ablation builds), so the two fractions ADD to the share of the SIMDs' time that is spoken for.
        mf = rec["mfma_busy_cycles"] / (mix["simds"] * rec["mfma_pass_shader_cycles"])
        out.update(matrix_pipe_frac=round(mf, 4), mfma_f64_insts_per_launch=rec["mfma_insts"],
                   mfma_busy_cycles_per_launch=rec["mfma_busy_cycles"], simd_busy_frac=round(frac + mf, 4))
    return out


def _cpu_pool_ready(_):
    import oracle

    oracle.build_c_oracle()
    return os.getpid()


def _cpu_trial(job):
    """One independent trial block on one host core: own PRBS segment, own PCG64 noise stream."""
    kind, idx, nsym, sps, ebn0 = job
    import numpy as np

    import oracle

    t0 = time.perf_counter()
    state = ((idx + 1) * 2654435761) & 0x7FFFFF or 1
    rng = np.random.Generator(np.random.PCG64(seed=1 + idx))
    pulse, sigma = oracle.freq_pulse_soqpsk_tg(sps), oracle.sigma_for_ebn0(ebn0, sps)
    if kind.startswith("cpm:"):   # generic CPM detector chain (build-defined; no reference form exists)
        waveform = kind[4:]
        states = 256 if waveform.endswith("256") else (64 if waveform.endswith("64") else 16)   # "multih64" / "multih256": the bigger ARTM designs (same modulator)
        waveform = waveform[:-len(str(states))] if states != 16 else waveform
        spec = {16: oracle.ARTM_16, 64: oracle.ARTM_64, 256: oracle.ARTM_256}[states] if waveform == "multih" else oracle.PCMFM_SPEC
        bits, _ = oracle.glfsr_bits(0x420000, state, nsym * spec.lgM)
        sym = oracle.multih_mapper(bits)[0] if waveform == "multih" else oracle.pcmfm_mapper(bits)
        pulse = oracle.freq_pulse_multih_irig(sps) if waveform == "multih" else oracle.freq_pulse_pcmfm(sps)
        res = oracle.cpm_detection_run(sym, pulse, sps, spec, sigma=oracle.cpm_sigma_for_ebn0(ebn0, sps, spec.lgM), rng=rng)
    elif kind == "port":      # compiled / vectorised restatement (C loops + numpy)
        bits, _ = oracle.glfsr_bits(0x420000, state, nsym)
        res = oracle.detection_run(bits, pulse, 0.25, sps, sigma, rng=rng)
    else:                   # the reference's execution form: interpreted per-symbol / per-sample loops
        from oracle import faithful_loops as fl

        bits, _ = fl.lfsr_bits_loop(0x420000, state, nsym)
        res = fl.detection_run_loop(bits, pulse, 0.25, sps, sigma, rng)
    return nsym, time.perf_counter() - t0, res["bit_errors"], res["compared"]


def cpu_baseline(sps: int, ebn0: float, nsym_port: int, nsym_loop: int, waveform: str = "soqpsk", gpus: int = 1) -> dict:
    """The oracle timed on the host cores of this box, one process per core on independent trial
    blocks, in two forms: `port` = the CPU port of the reference algorithm (C loops +
    numpy), and `faithful_loop` = the reference's own execution form (interpreted per-symbol
    detector and per-sample modulator loops, oracle/faithful_loops.py).  Outside the timed region."""
    import multiprocessing as mp

    host_cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host_cores
    # a 16-core SHARE of the host per GPU of the job (what a one-GPU box grants; an N-GPU job owns N shares), never more than
    # this process may run on
    cores = max(1, min(usable, 16 * max(1, int(gpus))))
    ctx = mp.get_context("spawn")            # never fork a process that has initialised HIP
    out = {}
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_pool_ready, range(cores), chunksize=1)     # imports + C build outside the clock
        kinds = (("port", nsym_port), ("faithful_loop", nsym_loop)) if waveform == "soqpsk" else ((f"cpm:{waveform}", nsym_port // 2),)
        for kind, n in kinds:
            one = _cpu_trial_timed(pool, [(kind, 0, n, sps, ebn0)])
            many = _cpu_trial_timed(pool, [(kind, 1 + k, n, sps, ebn0) for k in range(cores)])
            out[kind] = {"value": round(many["rate"], 4), "unit": "Msym/s", "cores": cores,
                         "single_core": round(one["rate"], 4),
                         "sample": f"{cores} independent blocks of {n} PN23 symbols, one process per core, "
                                   f"{many['wall']:.1f} s wall (1 block on 1 core: {one['wall']:.1f} s); "
                                   f"bit errors {many['bit_errors']}/{many['compared']}"}
    if waveform != "soqpsk":
        o = out[f"cpm:{waveform}"]
        return {**o, "host_cores": host_cores, "kind": "port",
                "sample": f"same {waveform} @{sps}sps chain, Eb/N0 {ebn0:.1f} dB, sequential C detector + numpy (the reference has no "
                          f"detector for this waveform: build-defined oracle), PCG64 noise: {o['sample']}"}
    return {"value": out["port"]["value"], "unit": "Msym/s", "cores": cores, "host_cores": host_cores, "kind": "port",
            # SURVEY 8(d) asks for two CPU variants: its "vectorised NumPy" one is this line's top level (the oracle's NumPy for the array
            # stages with the two sequential loops — encoder, detector — compiled, on a bounded 2^21-symbol sample per core instead of
            # N = 1e7: per-symbol cost is linear in N), its "faithful-loop" one is `faithful_loop` below
            "survey_variant": "vectorised (NumPy array stages + compiled sequential loops), bounded sample", 
            "cores_note": f"min(cores this process may run on, 16 per GPU of the job) = {cores} of the host's {host_cores} cores, one process per core",
            "single_core": out["port"]["single_core"],
            "sample": "same SOQPSK-TG @%dsps chain, Eb/N0 %.1f dB, oracle C loops + numpy, PCG64 noise: %s"
                      % (sps, ebn0, out["port"]["sample"]),
            "faithful_loop": {**out["faithful_loop"], "kind": "port",
                              "note": "the reference's execution form (interpreted per-symbol Viterbi, per-sample phase loop, "
                                      "full-rate np.convolve bank): what the reference costs on these cores"}}


def _cpu_trial_timed(pool, jobs) -> dict:
    t0 = time.perf_counter()
    res = pool.map(_cpu_trial, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    return {"rate": sum(r[0] for r in res) / wall / 1e6, "wall": wall,
            "bit_errors": sum(r[2] for r in res), "compared": sum(r[3] for r in res)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nsym", type=int, default=10_000_000)
    ap.add_argument("--sps", type=int, default=8)
    ap.add_argument("--ebn0", type=float, default=10.0)
    ap.add_argument("--detector", default="PT")
    ap.add_argument("--waveform", default="soqpsk", choices=["soqpsk", "multih", "pcmfm"],
                    help="soqpsk: BASELINE configs[1] (the headline metric); multih: configs[2], ARTM multi-h CPM through the "
                         "16-state generic CPM trellis detector; pcmfm: PCM/FM through the same detector family")
    ap.add_argument("--states", type=int, default=16, choices=[16, 64, 256],
                    help="--waveform multih: 16 = the reduced design BASELINE configs[2] names (Lp 2, NC 4); 64 = every phase state "
                         "for the two-symbol pulse (Lp 2, NC 16, N_S = p M^(Lp-1): notes/cpm/cpm.md:128-140), 0.2 dB better, one wave per detector; "
                         "256 = the full trellis (3-symbol pulse, 64 matched filters per symbol: 1 KB of rows per symbol), one workgroup per detector")
    ap.add_argument("--fuse", type=int, default=175,
                    help="bit 0: fused modulator (FIR + phase scan in one pass); bit 1: AWGN inside the MF bank; "
                         "bit 2: detector-packed 32 B rows between bank and detector; bit 3: modulator + channel + bank in one "
                         "kernel (no baseband samples in HBM); bit 4: PRBS + precoder through the generic kernels; bit 5: the "
                         "detector and the error count of a block on the context's side stream, beside the next block's front "
                         "end (15 = one block after the other); 0 = every stage its own kernel")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent trial blocks in flight on separate HIP streams (own workspace + context each)")
    ap.add_argument("--overlap-streams", type=int, default=3,
                    help="after the timed region (single GPU, --streams 1): the same K steps again with this many independent "
                         "blocks in flight on separate HIP streams, reported as `overlapped` (0 = skip)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="record the per-stage HIP events on every E-th timed step only (the last step always)")
    ap.add_argument("--vit-warmup", type=int, default=-1,
                    help="detector chunk warm-up in rows (= detector calls before a chunk's own first call), any waveform; "
                         "-1: by Eb/N0 (waveforms_amd.link.operating_point_warmup), 0: library default")
    ap.add_argument("--cpu-sample", type=int, default=1 << 21, help="symbols per core, compiled port")
    ap.add_argument("--cpu-loop-sample", type=int, default=1 << 16, help="symbols per core, faithful-loop form")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="a wf_ctx option for every context of this run (include/wfhip.h wf_option, lower-case name without "
                         "the WF_OPT_ prefix): e.g. --opt cpm_form=1 (row form), --opt cpm_chunk_calls=320, --opt det_final_verify=1")
    ap.add_argument("--ber-points", default="0:12",
                    help="SOQPSK at 8 sps: after the timed region, the BER sweep of BASELINE configs[3] (these Eb/N0 points x "
                         "--ber-symbols) through the same link, reported as `ber_curve` with its offset in dB from the reference's "
                         "golden curve — the second half of BASELINE's metric ('none' = skip)")
    ap.add_argument("--ber-symbols", type=float, default=1e8, help="symbols per point of that sweep")
    ap.add_argument("--steady-steps", type=int, default=2000,
                    help="after the timed region (single GPU): this many more steps in one go, reported as `steady_state` — long "
                         "enough (>= 1 s) that clock ramp and the 13 ms timed window can be told apart (0 = skip)")
    args = ap.parse_args()

    from waveforms_amd.bert import init_ranks, spawn_ranks

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as child processes, before
        # anything here has touched the GPU; rank 0 prints the JSON line on the inherited stdout
        raise SystemExit(spawn_ranks(str(Path(__file__).resolve()), args.gpus, sys.argv[1:]))
    import torch

    if int(os.environ.get("WORLD_SIZE", 1)) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}")
    # Rehearsal switch for a 1-GPU box: WF_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses
    # gloo for the two collectives, so the N > 1 code path can be exercised without N GPUs.
    rank, world, dist, coll_dev = init_ranks(rehearsal=os.environ.get("WF_BENCH_REHEARSAL") == "1")

    from waveforms_amd import _hip
    from waveforms_amd.link import SOQPSKLink, operating_point_warmup, soqpsk_warmup_param

    _hip.apply_option_args(args.opt)
    nstreams = max(1, args.streams)
    cpm = args.waveform != "soqpsk"
    if cpm:
        from waveforms_amd.link import CPMLink
        from waveforms_amd.viterbi.cpm import detector_kernel_name

        # detector chunk warm-up rows: by operating point (waveforms_amd.link.operating_point_warmup), or --vit-warmup
        wide = args.waveform == "multih" and args.states != 16
        cwu = args.vit_warmup if args.vit_warmup >= 0 else (0 if wide else operating_point_warmup(args.waveform, args.ebn0))
        big_spec = None
        if wide:
            from waveforms_amd.viterbi import cpm as _cpm
            big_spec = _cpm.ARTM_64 if args.states == 64 else _cpm.ARTM_256
        links = [CPMLink(args.nsym, args.sps, waveform=args.waveform, spec=big_spec, private_ctx=nstreams > 1, fuse=args.fuse, warmup=cwu)      # (bits 1, 3, 5 apply)
                 for _ in range(nstreams)]
        bits_per_sym = links[0].spec.bits_per_symbol
    else:
        # detector chunk warm-up rows: by operating point (16 at Eb/N0 >= 6 dB, else the library default of 32), or
        # --vit-warmup: a matter of speed only (chunks that miss it are repaired on the device)
        wu_rows = args.vit_warmup if args.vit_warmup >= 0 else operating_point_warmup("soqpsk", args.ebn0)
        wu = soqpsk_warmup_param(wu_rows)
        links = [SOQPSKLink(args.nsym, args.sps, detector=args.detector, fuse=args.fuse, private_ctx=nstreams > 1, warmup=wu)
                 for _ in range(nstreams)]
        bits_per_sym = 1
    streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream()]
    link = links[0]
    slots = 64
    assert args.steps >= 1

    every = max(1, args.event_every)

    def instrumented(k: int) -> bool:
        return k % every == every - 1 or k == args.steps - 1

    def step(k: int, timed: bool) -> None:
        # every (rank, step) is its own trial block: distinct PRBS segment and Philox subsequence
        block = k * world + rank
        with torch.cuda.stream(streams[k % nstreams]):
            links[k % nstreams].run_block(args.ebn0, seed=1, stream_id=block & 0xFFFFFFFF,
                                          skip_bits=(block % 4096) * args.nsym * bits_per_sym,
                                          event_slot=(k % slots) if timed and instrumented(k) else -1)

    def fence() -> None:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(-1 - k, False)
    torch.cuda.synchronize()
    for l in links:
        l.reset_counts()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, True)
    fence()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        # every rank's own time (all_gather of one double), so that the first real N-GPU run shows imbalance
        mine = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in gathered]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    se = be = compared = 0
    for l in links:
        a_, b_, c_ = l.result()
        se, be, compared = se + a_, be + b_, compared + c_
    counts = torch.tensor([se, be, compared], dtype=torch.int64, device=coll_dev)
    collective = None
    if dist is not None:
        own = [int(v) for v in counts.cpu().tolist()]
        torch.cuda.synchronize()
        dist.barrier()
        tc = time.perf_counter()
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)     # the one collective of the job
        if coll_dev != "cpu":
            torch.cuda.synchronize()
        all_reduce_us = (time.perf_counter() - tc) * 1e6
        # what ran, as the ranks themselves saw it — so that the first real N-GPU run proves its own collective: the
        # backend, the group size, which physical device each rank drove, every rank's own count (their sum must be the
        # reduced total) and the wall time of the all-reduce
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        ident = {"rank": rank, "local_device": torch.cuda.current_device(), "name": props.name,
                 "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
                 "uuid": str(getattr(props, "uuid", "")), "compared": own[2], "bit_errors": own[1]}
        seen = [None] * world
        dist.all_gather_object(seen, ident)
        collective = {"backend": dist.get_backend(), "world_seen": dist.get_world_size(), "all_reduce_us": round(all_reduce_us, 1),
                      "devices": seen, "sum_of_rank_counts_equals_reduced": sum(d["bit_errors"] for d in seen) == int(counts[1].item()),
                      "distinct_devices": len({(d["pci"], d["uuid"]) for d in seen})}
    se, be, compared = (int(v) for v in counts.cpu().tolist())

    # per-stage kernel time from the HIP events recorded inside the timed region
    ev_steps = [k for k in range(max(0, args.steps - slots), args.steps) if instrumented(k)]
    acc = {}
    for k in ev_steps:
        for name, ms in links[k % nstreams].stage_ms(k % slots).items():
            acc[name] = acc.get(name, 0.0) + ms / len(ev_steps)
    bps = stage_bytes_per_symbol(args.sps)
    det_info = [0, 0, 0, 0]           # generic CPM detector: {form, ring slots, calls per chunk, warm-up calls} as the library will launch it
    if cpm:
        nf = links[0].spec.nfilt
        # bits -> symbols -> c128 samples -> noisy samples in place -> nf complex rows -> one decision byte
        bps = {"prbs": bits_per_sym, "map": bits_per_sym + 1, "modulate": 1 + 16 * args.sps, "awgn": 32 * args.sps,
               "mfbank": 16 * args.sps + 16 * nf, "viterbi": 16 * nf + 1, "count": 2}
        import ctypes as _ct
        link_form = (_ct.c_int * 4)()
        _hip.check(_hip.lib().wf_cpm_link_form(links[0]._ctx, _ct.byref(links[0].cfg), link_form))
        samples_form = link_form[0] == 2
        if samples_form:
            # fuse bit 7 in effect (round 6): the "modulate" slot timed modulator + channel with the noisy SAMPLES stored (1 B in,
            # 16 sps B out per symbol); the matched filters run inside the detector, which reads those samples (16 sps B) and
            # writes one decision byte — no rows exist
            acc["mod+awgn"] = acc.pop("modulate")
            acc.pop("awgn", None)
            acc.pop("mfbank", None)
            bps["mod+awgn"] = 1 + 16 * args.sps
            bps["viterbi"] = 16 * args.sps + 1
            STAGE_KERNEL["mod+awgn"] = "mod_chan_samples_kernel<4>"
        elif links[0].layout()["one_kernel_front_end"]:
            # fuse bit 3 in effect: the "modulate" slot timed modulator + channel + filters (one kernel)
            acc["mod+awgn+mfbank"] = acc.pop("modulate")
            acc.pop("awgn", None)
            acc.pop("mfbank", None)
            bps["mod+awgn+mfbank"] = 1 + 16 * nf
            # (conjugate-paired templates — wf_cpm_link_config.fuse bit 6 — run the four-real-sums form, template value 2 nf)
            STAGE_KERNEL["mod+awgn+mfbank"] = f"mod_chan_bank_kernel<4, {2 * nf if (links[0].cfg.fuse & 64) else nf}, 8>"
        STAGE_KERNEL.update({"mfbank": f"cpm_mf_rows_kernel<{nf}, {9 if args.sps == 8 else 0}, {'true' if links[0].cfg.fuse & 2 else 'false'}>",
                             # (the profiles time every kernel ALONE — link fuse 15 —, where the lane form launches the instantiation that
                             #  claims a SIMD's registers; the pipelined link runs the plain one: same code, same counters but for the claim)
                             "viterbi": detector_kernel_name(links[0].spec, links[0].layout()["calls"], links[0].cfg.warmup, ctx=links[0]._ctx, info4=det_info),
                             "map": "symbol_map_kernel",
                             "modulate": "mod_main_kernel<4, true, false>"})
        if samples_form and link_form[1] == 1:
            # (the profiles time every kernel ALONE, where the lane form launches the instantiation that claims a SIMD's registers)
            k1 = links[0].spec.K[1] if len(links[0].spec.K) > 1 else links[0].spec.K[0]
            STAGE_KERNEL["viterbi"] = (f"cpm_lane_kernel<lane_spec<{links[0].spec.M}, {links[0].spec.Lp}, {links[0].spec.NC}, {links[0].spec.p}, "
                                       f"{len(links[0].spec.K)}, {links[0].spec.K[0]}, {k1}>, 3, true, true, true>")
            det_info[:] = [1, 3, int(link_form[2]), int(link_form[3])]
    one_kernel = not cpm and bool(links[0].layout()["one_kernel_front_end"])     # asked from the library (wf_link_layout)
    if one_kernel:      # fuse bit 3: the "fir" slot times modulator + channel + bank; symbols in, packed rows out
        acc["mod+awgn+mfbank"] = acc.pop("fir")
        for k in ("phase", "awgn", "mfbank"):
            acc.pop(k, None)
        bps["mod+awgn+mfbank"] = 1 + 32
        bps["viterbi"] = 32 + 2
        jm = 4 if -(-links[0].cfg.ntaps // args.sps) <= 4 else 9
        # (a bank other than the sps + 1 pulse-truncation taps — the 73-tap PAM bank — is the kernel's matrix-core form)
        # (... its factored form -2 when the link handed the bank over as two real filters + a 3 x 2 combination: wf_link_config.d_mf_factor)
        form = 0 if links[0].cfg.mf_ntaps == args.sps + 1 else (-2 if links[0].cfg.d_mf_factor else -1)
        STAGE_KERNEL["mod+awgn+mfbank"] = f"mod_chan_bank_kernel<{jm}, {form}, {args.sps}>"
        STAGE_KERNEL["viterbi"] = "viterbi_batch_kernel<true>"
    elif not cpm and args.fuse & 1:   # the "fir" event slot times the fused modulator: symbols in, c128 out
        acc["modulate"] = acc.pop("fir")
        acc.pop("phase", None)
        bps["modulate"] = 1 + 16 * args.sps
    if not cpm and not one_kernel and args.fuse & 2:   # noisy samples never materialise: clean c128 in, 3 c128 per symbol out
        acc["awgn+mfbank"] = acc.pop("mfbank") + acc.pop("awgn")
        bps["awgn+mfbank"] = bps["mfbank"]
    packed = not cpm and not one_kernel and links[0].row_bytes == 32
    if packed:          # detector-packed rows: 4 doubles per symbol between the bank and the detector
        bps["awgn+mfbank"] = 16 * args.sps + 32
        bps["viterbi"] = 32 + 2
        STAGE_KERNEL["awgn+mfbank"] = ("mf_bank_kernel<3, true, 8, 9, true>" if links[0].cfg.mf_ntaps == 9
                                       else "mf_bank_kernel<3, true, 8, 0, true>")
        STAGE_KERNEL["viterbi"] = "viterbi_batch_kernel<true>"
    stages = {}
    for name, ms in acc.items():
        gb = bps[name] * args.nsym / 1e9
        stages[name] = {"ms": round(ms, 4), "algo_GB": round(gb, 4),
                        "GBps": round(gb / (ms / 1e3), 1) if ms > 0 else None}
    piped = bool(links[0].cfg.fuse & 32) and (bool(links[0].layout()["one_kernel_front_end"]) or (cpm and samples_form))
    alone = {}
    if piped:
        # fuse bit 5: a block's detector runs beside the next block's front end, so both live durations are those
        # of kernels SHARING the chip; the dominant kernel is then named by the committed profile's durations
        # (--fuse 15, one kernel at a time) where it has them
        for name in acc:
            r_ = profile_record(name, args.nsym, args.sps)
            if r_ and r_.get("avg_ns"):
                alone[name] = r_["avg_ns"] / 1e6
    dominant = max(alone, key=alone.get) if alone else max(acc, key=acc.get)
    d = stages[dominant]
    rec = profile_record(dominant, args.nsym, args.sps)
    hbm = {"achieved": d["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d["GBps"] / HBM_PEAK_GBS, 4)}
    roofline = {"bound": "hbm", "kernel": STAGE_KERNEL.get(dominant, dominant), "stage": dominant, **hbm,
                "traffic": rec["traffic"] if rec else None,
                "traffic_source": rec["source"] if rec else None,
                "traffic_profile_matches_build": rec["current"] if rec else None,
                "launch_ms": d["ms"], "algorithmic_bytes_per_launch": int(bps[dominant] * args.nsym),
                "hbm_frac": hbm["frac"], "hbm": hbm}
    if piped:
        roofline.update(launch_overlapped_with="the neighbouring block's kernels on the other stream (fuse bit 5)",
                        launch_ms_alone_in_profile=round(rec["avg_ns"] / 1e6, 4) if rec and rec.get("avg_ns") else None)
    vi = valu_issue(dominant, rec, d["ms"])
    if vi:
        # The roofline that BINDS is the one reported as bound / achieved / peak / frac: for the fused kernels HBM
        # traffic is already minimal and the vector pipe's issue slots are the scarce resource ("the
        # fully fused pipeline is compute-bound"); the HBM side stays beside it as hbm_frac / hbm.
        vfrac = vi.pop("valu_issue_frac")
        roofline.update(valu_issue_frac=vfrac, valu=vi)
        if vfrac > hbm["frac"]:
            roofline.update(bound="valu_issue", achieved=vi["issue_cycles_per_launch"],
                            peak=vi["simds"] * vi["shader_cycles_per_launch"], unit="SIMD issue cycles per launch", frac=vfrac)
            if vi.get("simd_busy_frac"):
                # fp64 matrix tiles in the kernel: vector issue + matrix-pipe busy cycles, which exclude each other on a SIMD
                roofline.update(bound="simd_issue(vector+matrix)", achieved=vi["issue_cycles_per_launch"] + vi["mfma_busy_cycles_per_launch"],
                                frac=vi["simd_busy_frac"])
        roofline.update(binding=roofline["bound"], binding_frac=roofline["frac"])
    # every stage with a profile record: its own HBM and issue fractions
    for name, st in stages.items():
        r_ = profile_record(name, args.nsym, args.sps)
        if r_:
            st["hbm_traffic_bytes"] = r_["traffic"]
            v_ = valu_issue(name, r_, st["ms"])
            if v_:
                st["valu_issue_frac"] = v_["valu_issue_frac"]

    steady = None
    if nstreams == 1 and args.steady_steps > 0:      # (N > 1: every rank runs its own, reported per rank below)
        from waveforms_amd import device as _dev0
        for l in links:
            l.reset_counts()
        torch.cuda.synchronize()
        # (the repair counters below are those of the steady loop alone: what the warm-up and the timed steps left is dropped here)
        _dev0.viterbi_repaired(reset=True, ctx=links[0]._ctx)
        _dev0.viterbi_cascaded(reset=True, ctx=links[0]._ctx)
        t2 = time.perf_counter()
        for k in range(args.steady_steps):
            links[0].run_block(args.ebn0, seed=1, stream_id=k & 0xFFFFFFFF, skip_bits=(k % 4096) * args.nsym * bits_per_sym)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        # (every launch proves its chunks on the device and repairs the ones that missed their warm-up: what is
        #  reported here is how often that ran over the 2000 blocks, and that nothing was left unproven)
        from waveforms_amd import device as _dev
        s2 = links[0].result()
        steady = {"steps": args.steady_steps, "seconds": round(dt2, 3), "ms_per_step": round(dt2 / args.steady_steps * 1e3, 4),
                  "value": round(args.steady_steps * args.nsym / dt2 / 1e6, 2), "unit": "Msym/s",
                  "bit_errors": int(s2[1]), "detector_chunks_unproven": 0,
                  "detector_chunk_repairs": int(_dev.viterbi_repaired(reset=True, ctx=links[0]._ctx)),
                  "detector_chunk_repairs_handed_on": int(_dev.viterbi_cascaded(reset=True, ctx=links[0]._ctx)),
                  "note": "same step, same single stream, outside the driver-timed K steps"}

    # The second half of BASELINE's metric: the BER curve's offset in dB from the reference's
    # (examples/soqpsk_detection.py:200-216 is the quantity; tests/golden/ber_golden*.csv the reference's own counts):
    # configs[3]'s sweep through the link the timed region just ran, one block after the other.
    ber_curve = None
    if rank == 0 and not cpm and args.sps == 8 and nstreams == 1 and args.ber_points not in ("", "none") and args.detector in ("PT", "PAM"):
        sys.path.insert(0, str(ROOT / "tools"))
        from ber_sweep import golden_curve
        from waveforms_amd.bert import ebn0_at_ber

        lo_, hi_ = (int(v) for v in args.ber_points.split(":"))
        pts = list(range(lo_, hi_ + 1))
        blocks = max(1, int(round(args.ber_symbols / args.nsym)))
        keep_wu = links[0].cfg.warmup
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        rows = []
        for pi, e in enumerate(pts):
            links[0].reset_counts()
            if args.vit_warmup < 0:
                links[0].cfg.warmup = soqpsk_warmup_param(operating_point_warmup("soqpsk", float(e)))
            for b in range(blocks):       # trial block (point, b) exactly as waveforms_amd.bert.SweepPlan deals them
                links[0].run_block(float(e), seed=1, stream_id=(pi << 32) | b, skip_bits=b * args.nsym)
            rows.append(links[0].result())
        dt4 = time.perf_counter() - t4
        links[0].cfg.warmup = keep_wu
        ber_pts = [r[1] / max(r[2], 1) for r in rows]
        ge, gb, _, gn = golden_curve(args.detector)
        ber_curve = {"ebn0_db": pts, "symbols_per_point": int(rows[0][2]), "bit_errors": [int(r[1]) for r in rows],
                     "seconds": round(dt4, 4), "Msym_per_s": round(sum(r[2] for r in rows) / dt4 / 1e6, 1),
                     "golden": "tests/golden/ber_golden*.csv (the reference's own run, PCG64 noise; "
                               f"{int(gn.min())} .. {int(gn.max())} symbols per point)", "tolerance_db": 0.05}
        for target in (1e-3, 1e-4):
            try:
                mine = ebn0_at_ber(pts, ber_pts, target)
                ber_curve[f"ebn0_at_{target:g}"] = round(mine, 4)
                ber_curve[f"delta_db_at_{target:g}"] = round(mine - ebn0_at_ber(ge, gb, target), 4)
            except ValueError:
                ber_curve[f"delta_db_at_{target:g}"] = None

    if rank == 0 and cpm and nstreams == 1 and args.ber_points not in ("", "none"):
        # The CPM links have no curve in the reference (it ships no detector for them: DESIGN.md section 2): the same sweep,
        # Eb/N0 at BER 1e-3 / 1e-4 read off it, no offset to quote.  (256 states: 12 ms per block — a shorter sweep.)
        from waveforms_amd.bert import ebn0_at_ber

        lo_, hi_ = (int(v) for v in args.ber_points.split(":"))
        pts = list(range(lo_, hi_ + 1))
        blocks = max(1, int(round(args.ber_symbols / args.nsym)))
        keep_wu = links[0].cfg.warmup
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        rows = []
        for pi, e in enumerate(pts):
            links[0].reset_counts()
            if args.vit_warmup < 0 and not wide:
                links[0].cfg.warmup = operating_point_warmup(args.waveform, float(e))
            for b in range(blocks):
                links[0].run_block(float(e), seed=1, stream_id=(pi << 32) | b, skip_bits=b * args.nsym * bits_per_sym)
            rows.append(links[0].result())
        dt4 = time.perf_counter() - t4
        links[0].cfg.warmup = keep_wu
        ber_pts = [r[1] / max(r[2] * bits_per_sym, 1) for r in rows]
        ber_curve = {"ebn0_db": pts, "symbols_per_point": int(rows[0][2]), "bit_errors": [int(r[1]) for r in rows],
                     "seconds": round(dt4, 4), "Msym_per_s": round(sum(r[2] for r in rows) / dt4 / 1e6, 1),
                     "golden": None, "note": "no curve of the reference for this waveform (it ships no detector for it); the detector is oracle/cpm_oracle.c's"}
        for target in (1e-3, 1e-4):
            try:
                ber_curve[f"ebn0_at_{target:g}"] = round(ebn0_at_ber(pts, ber_pts, target), 4)
            except ValueError:
                ber_curve[f"ebn0_at_{target:g}"] = None

    # The same K steps once more with several independent trial blocks in flight (own workspace, wf_ctx and
    # stream each), as the BER sweep runs them: the vector-pipe-bound front-end kernel of one block overlaps the
    # detector and the small integer kernels of its neighbours.  Reported beside `value`, never instead of it:
    # the per-kernel times above are only clean with one block at a time.
    overlapped = None
    if world == 1 and nstreams == 1 and args.overlap_streams > 1:
        n2 = args.overlap_streams
        extra = [type(links[0])(args.nsym, args.sps, **(dict(waveform=args.waveform, fuse=args.fuse, warmup=links[0].cfg.warmup)
                                                         if cpm else dict(detector=args.detector, fuse=args.fuse, warmup=links[0].cfg.warmup)),
                                private_ctx=True) for _ in range(n2)]
        lanes = [torch.cuda.Stream() for _ in range(n2)]

        def ostep(k: int) -> None:
            with torch.cuda.stream(lanes[k % n2]):
                extra[k % n2].run_block(args.ebn0, seed=1, stream_id=k & 0xFFFFFFFF, skip_bits=(k % 4096) * args.nsym * bits_per_sym)

        for k in range(2 * n2):
            ostep(-1 - k)
        torch.cuda.synchronize()
        for l in extra:
            l.reset_counts()
        t1 = time.perf_counter()
        for k in range(args.steps):
            ostep(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        tot = [sum(v) for v in zip(*(l.result() for l in extra))]
        overlapped = {"streams": n2, "value": round(args.steps * args.nsym / dt / 1e6, 2), "unit": "Msym/s",
                      "ms_per_step": round(dt / args.steps * 1e3, 4), "bit_errors": tot[1],
                      "same_blocks_same_counts": tot[1] == be and tot[0] == se}
        if args.steady_steps > 0:          # ... and, like `steady_state`, over a window long enough for the chip's clock to settle
            t3 = time.perf_counter()
            for k in range(args.steady_steps):
                ostep(k)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t3
            overlapped["steady_value"] = round(args.steady_steps * args.nsym / dt3 / 1e6, 2)
            overlapped["steady_steps"] = args.steady_steps
        del extra

    steady_all = None
    if dist is not None and steady is not None:
        steady_all = [None] * world
        dist.all_gather_object(steady_all, {"rank": rank, "ms_per_step": steady["ms_per_step"], "value": steady["value"],
                                            "detector_chunks_unproven": steady["detector_chunks_unproven"]})
    if rank == 0:
        total_sym = args.steps * args.nsym * world
        out = {
            "metric": (f"SOQPSK-TG Msym/s mod+Viterbi-detect @{args.sps}sps" if not cpm else    # BASELINE's metric at the default sps = 8
                       f"{'ARTM multi-h CPM' if args.waveform == 'multih' else 'PCM/FM'} Msym/s mod+{links[0].spec.nstates}-state-Viterbi-detect @{args.sps}sps"),
            "value": round(total_sym / elapsed / 1e6, 2), "unit": "Msym/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"SOQPSK-TG {args.nsym:.0e} symbols @{args.sps} sps, modulate + AWGN + "
                                    f"{args.detector} matched filter + 4-state Viterbi detect (BASELINE configs[1])" if not cpm else
                                    f"{'Multi-h ARTM CPM (IRIG106 Tier II)' if args.waveform == 'multih' else 'PCM/FM'} {args.nsym:.0e} symbols "
                                    f"@{args.sps} sps, modulate + AWGN + {links[0].spec.nfilt} matched filters ({links[0].spec.Lp}-symbol pulse "
                                    f"truncation) + {links[0].spec.nstates}-state trellis detect"
                                    + (" (BASELINE configs[2])" if args.waveform == "multih" and args.states == 16 else "")),
                       "waveform": args.waveform,
                       "symbols_per_step_per_gpu": args.nsym, "sps": args.sps, "ebn0_db": args.ebn0,
                       "prbs": "PN23", "noise": "Philox4x32-10 + Box-Muller (device)",
                       "fuse": args.fuse, "streams": nstreams,
                       "detector_warmup_rows": det_info[3] if cpm else (links[0].cfg.warmup + 1 if links[0].cfg.warmup else 32),
                       "detector_chunk_calls": det_info[2] if cpm else None,
                       "detector_form": {0: "rows", 1: "lanes", 2: "wide", 3: "quad"}[det_info[0]] if cpm else "lanes (one chunk per lane)",
                       "ctx_options": args.opt, "parallelism": f"independent trial blocks x{world}"},
            "ber": {"bit_errors": be, "symbol_errors": se, "symbols": compared,
                    "ber": be / max(compared * bits_per_sym, 1),
                    # every detector chunk started from bitwise the state of the sequential detector, or was run again
                    # from it (device-side proof + repair in every launch; link.result() raises otherwise)
                    "detector_chunks_unproven": 0},
            "roofline": roofline,
            "stages": stages,
        }
        if steady:
            # Two numbers a reader trips on (round-5 verdict): `value` is timed by the driver's contract over K steps that start on
            # an idle chip — the whole window sits inside the ~30 ms the clock needs under load to reach what it then holds —, the
            # steady figure is the same step over a window long enough for that; for N > 1 the per-rank steady figures are the ones
            # a scaling curve should be read from (the timed window is K steps + one barrier: ramp and launch skew dominate it)
            out["clock_ramp"] = {"timed_window_ms": round(elapsed * 1e3, 3), "timed_ms_per_step": round(elapsed / args.steps * 1e3, 4),
                                 "steady_ms_per_step": steady["ms_per_step"], "steady_over_timed": round(steady["ms_per_step"] / (elapsed / args.steps * 1e3), 4),
                                 "note": "the driver-timed K steps run inside the chip's clock ramp (~30 ms under load); steady_state is the same step over 2000 blocks"}
        if world > 1:
            out["scaling_figure"] = "steady_state_per_rank (each rank's own 2000-block window); `value` over K steps + one barrier is dominated by clock ramp and launch skew"
        if overlapped:
            out["overlapped"] = overlapped
        if steady:
            out["steady_state"] = steady
        if ber_curve:
            out["ber_curve"] = ber_curve
        if world > 1:
            out["per_rank_ms_per_step"] = [round(v, 4) for v in rank_ms]
            out["rank_time_max_over_min"] = round(max(rank_ms) / max(min(rank_ms), 1e-12), 4)
            out["collective"] = collective
            if steady_all:
                out["steady_state_per_rank"] = steady_all
        if not args.no_cpu_baseline:    # rank 0 only, after the timed region, at any N (the other ranks wait at the final barrier)
            out["cpu_baseline"] = cpu_baseline(args.sps, args.ebn0, args.cpu_sample, args.cpu_loop_sample,
                                                   args.waveform + (str(args.states) if args.waveform == "multih" and args.states != 16 else ""), gpus=world)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
