"""Host-side logic of the product package (tap design, trellis model, alias import,
sweep sharding) against the reference-generated goldens.  CPU only."""
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

TRELLIS_NAMES = ["SOQPSKTrellis8x1", "SOQPSKTrellis4x2", "SOQPSKTrellis4x2DiffEncoded",
                 "SimpleTrellis2", "SimpleTrellis4"]


def test_alias_serves_the_same_modules():
    import waveforms
    import waveforms.noise
    import waveforms_amd.noise
    from waveforms.cpm.trellis.model import SOQPSKTrellis4x2DiffEncoded as a
    from waveforms_amd.cpm.trellis.model import SOQPSKTrellis4x2DiffEncoded as b

    assert waveforms.__version__ == "0.1.0"
    assert waveforms.noise is waveforms_amd.noise and a is b
    # the names examples/soqpsk_detection.py:10-22 imports
    from waveforms.cpm.modulate import cpm_modulate, frequency_modulate, phase_modulate  # noqa: F401
    from waveforms.cpm.pamapprox import rho_pulses  # noqa: F401
    from waveforms.cpm.soqpsk import freq_pulse_soqpsk_mil, freq_pulse_soqpsk_tg  # noqa: F401
    from waveforms.cpm.trellis.encoder import TrellisEncoder  # noqa: F401
    from waveforms.glfsr import PNSequence  # noqa: F401
    from waveforms.noise import generate_complex_awgn  # noqa: F401
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector  # noqa: F401
    from waveforms.filters.lpf import kaiser_fir_lpf  # noqa: F401
    with pytest.raises(ImportError):
        import waveforms.does_not_exist  # noqa: F401


def test_glfsr_host_side(golden):
    from waveforms.glfsr import GLFSR, PNSequence
    from waveforms.glfsr.pn import GALOIS_LFSR_POLYS, generate_mask

    g = golden("glfsr")
    assert [generate_mask(k) for k in range(2, len(GALOIS_LFSR_POLYS))] == [int(m) for m in g["masks"]]
    for bad in (0, 1, 65, 100):
        with pytest.raises(KeyError):
            generate_mask(bad)
    p = PNSequence(23)
    assert (p.mask, p.state, p.degree) == (0x420000, 0x7FFFFF, 23)
    assert [p.next_bit() for _ in range(200)] == g["pn23_first200"].tolist()
    assert p.state == int(g["pn23_state200"][0])
    q = GLFSR(0x6000, 0x7FFF)
    bits = np.array([q.next_bit() for _ in range(32767)], dtype=np.uint8)
    assert np.array_equal(np.packbits(bits), g["pn15_packed"]) and q.state == 0x7FFF


@pytest.mark.parametrize("name", TRELLIS_NAMES)
def test_trellis_model(golden, name):
    from waveforms.cpm.trellis import model as tm

    g = golden("encode")
    tr = getattr(tm, name)
    flat = np.array([[b.inp, b.out, b.start, b.end] for col in tr.branches for b in col], dtype=np.int8)
    assert np.array_equal(flat, g[f"{name}__branches"])
    assert [tr.columns, tr.states, tr.input_cardinality, tr.output_cardinality,
            tr.branches_per_column] == [int(v) for v in g[f"{name}__dims"]]
    fsm = tm.FiniteStateMachine(tr)
    assert fsm.symbols == sorted({int(v) for v in flat[:, 1]})
    for c, col in enumerate(tr.branches):
        for b in col:
            assert fsm.forward_branch_mapping[c][b.start][b.inp] is b
            assert fsm.reverse_transitions[c][b.end][b.start] is b
            assert tm.forward_map(b.start, col)[b.inp] is b
            assert b in tm.reverse_branches(b.end, col) and b in tm.forward_branches(b.start, col)
    nxt, out = tr.dense_tables()
    assert nxt.shape == (tr.columns, tr.states, 1 << tr.input_cardinality)


def test_trellis_rejects_non_power_of_two_inputs():
    from waveforms.cpm.trellis.model import Branch, Trellis

    bad = Trellis(branches=[[Branch(0, 0, 0, 0), Branch(1, 0, 0, 0), Branch(2, 0, 0, 0)]])
    with pytest.raises(ValueError):
        _ = bad.input_cardinality


def test_pulse_design(golden):
    from waveforms.cpm.helpers import normalize_cpm_filter
    from waveforms.cpm.multih import MULTIH_IRIG_DENOM, MULTIH_IRIG_NUMER, freq_pulse_multih_irig
    from waveforms.cpm.pamapprox import pam_unit_pulse, pam_unit_pulse2, rho_pulses
    from waveforms.cpm.pcmfm import PCMFM_DENOM, PCMFM_NUMER, freq_pulse_pcmfm
    from waveforms.cpm.soqpsk import (SOQPSK_DENOM, SOQPSK_NUMER, freq_pulse_soqpsk_a, freq_pulse_soqpsk_b,
                                      freq_pulse_soqpsk_mil, freq_pulse_soqpsk_tg)
    from waveforms.filters.lpf import kaiser_fir_lpf

    g = golden("pulses")
    for sps in (4, 8, 10):
        np.testing.assert_array_equal(freq_pulse_soqpsk_tg(sps), g[f"tg_{sps}"])
        np.testing.assert_array_equal(freq_pulse_soqpsk_mil(sps), g[f"mil_{sps}"])
        np.testing.assert_array_equal(freq_pulse_soqpsk_a(sps), g[f"a_{sps}"])
        np.testing.assert_array_equal(freq_pulse_soqpsk_b(sps), g[f"b_{sps}"])
        np.testing.assert_array_equal(freq_pulse_multih_irig(sps), g[f"multih_{sps}"])
        for w, f in (("tg", freq_pulse_soqpsk_tg), ("mil", freq_pulse_soqpsk_mil)):
            for k, r in enumerate(rho_pulses(f(sps), 0.25, sps, 2)):
                np.testing.assert_array_equal(r, g[f"rho{k}_{w}_{sps}"])
    for sps, order in ((8, 4), (8, 6), (20, 4), (20, 8)):
        np.testing.assert_array_equal(freq_pulse_pcmfm(sps, order), g[f"pcmfm_{sps}_{order}"])
    np.testing.assert_array_equal(kaiser_fir_lpf(8, 0.5), g["kaiser_8_0p5"])
    np.testing.assert_array_equal(kaiser_fir_lpf(10, 0.7, 0.2, 60.0), g["kaiser_10_0p7_w0p2_r60"])
    q = np.cumsum(freq_pulse_soqpsk_tg(8)) / 8
    np.testing.assert_array_equal(pam_unit_pulse(q, 0.25), g["unit_pulse_tg_8"])
    np.testing.assert_array_equal(pam_unit_pulse2(q, 0.25), g["unit_pulse2_tg_8"])
    np.testing.assert_array_equal(normalize_cpm_filter(8, g["normalize_in"]), g["normalize_out"])
    assert [SOQPSK_NUMER, SOQPSK_DENOM, PCMFM_NUMER, PCMFM_DENOM, MULTIH_IRIG_DENOM, *MULTIH_IRIG_NUMER] == \
        [int(v) for v in g["consts"]]


def test_numpy_generator_noise_contract(golden):
    """generate_complex_awgn keeps honouring a caller-supplied numpy Generator and the
    module-level DEFAULT_RNG (reference waveforms/noise.py:5,23)."""
    import waveforms.noise as noise

    g = golden("awgn")
    rng = np.random.Generator(np.random.PCG64(seed=1))
    np.testing.assert_array_equal(noise.generate_complex_awgn(np.sqrt(2) / 2, 4104, rng),
                                  g["seed1_sigma_sqrt_half_4104"])
    saved = noise.DEFAULT_RNG
    try:
        noise.DEFAULT_RNG = np.random.Generator(np.random.PCG64(seed=1))
        np.testing.assert_array_equal(noise.generate_complex_awgn(np.sqrt(2) / 2, 100),
                                      g["seed1_sigma_sqrt_half_4104"][:100])
    finally:
        noise.DEFAULT_RNG = saved


def test_matched_filter_taps(oracle):
    from waveforms.cpm.soqpsk import freq_pulse_soqpsk_mil, freq_pulse_soqpsk_tg
    from waveforms.filters.matched import pam_matched_filter_taps, pt_matched_filter_taps

    for sps in (8, 10):
        for pulse in (freq_pulse_soqpsk_tg(sps), freq_pulse_soqpsk_mil(sps)):
            np.testing.assert_array_equal(pt_matched_filter_taps(pulse, 0.25, sps), oracle.pt_taps(pulse, 0.25, sps))
            # folding the pseudo-symbol weights into the taps == weighting the outputs
            taps = pam_matched_filter_taps(pulse, 0.25, sps)
            rng = np.random.Generator(np.random.PCG64(5))
            r = rng.normal(size=700) + 1j * rng.normal(size=700)
            want = oracle.pam_bank(r, pulse, 0.25, sps)
            got = np.array([np.convolve(r, t, mode="same") for t in taps])
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


def test_sweep_plan_and_interpolation():
    from waveforms.bert import SweepPlan, ber_table, ebn0_at_ber

    plan = SweepPlan(ebn0_db=list(range(13)), blocks_per_point=5, nsym=1 << 17)
    assert len(plan.jobs) == 65
    shards = [plan.shard(r, 8) for r in range(8)]
    assert sorted(j for s in shards for j in s) == sorted(plan.jobs)          # a partition
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1     # balanced
    assert plan.stream_id(3, 7) != plan.stream_id(7, 3) and plan.skip_bits(4) == 4 << 17
    tab = ber_table([0, 1], np.array([[10, 8, 100], [2, 1, 100]]))
    assert tab[0]["ber"] == 0.08 and tab[1]["ser"] == 0.02
    assert abs(ebn0_at_ber([8, 9, 10], [1e-2, 1e-3, 1e-4], 1e-3) - 9.0) < 1e-12
    assert abs(ebn0_at_ber([8, 10], [1e-2, 1e-4], 1e-3) - 9.0) < 1e-12
    with pytest.raises(ValueError):
        ebn0_at_ber([8, 9], [1e-2, 1e-3], 1e-6)


def _gloo_worker(rank, world, port, out):
    import os

    import torch.distributed as dist

    from waveforms.bert import SweepPlan, ber_sweep

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = SweepPlan(ebn0_db=[0.0, 4.0, 8.0], blocks_per_point=5, nsym=1000)

    def fake_runner(plan):
        acc = np.zeros((3, 3), dtype=np.int64)

        def run(point, block):          # deterministic stand-in for the GPU block
            acc[point] += (100 * point + block, 10 * point + block, plan.nsym)

        return run, lambda: acc

    res = ber_sweep(plan, runner=fake_runner)
    out.put((rank, res.tolist()))
    dist.destroy_process_group()


def test_sweep_sharding_and_reduce_gloo_world2():
    """N > 1 path: shards are disjoint, the single all-reduce gives every rank the totals."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blocks = np.arange(5).sum()
    want = [[100 * p * 5 + blocks, 10 * p * 5 + blocks, 5000] for p in range(3)]
    assert got[0] == want and got[1] == want


def test_spawn_ranks_starts_children_and_relays_rank0(tmp_path):
    """`python bench.py --gpus N` without torchrun: spawn_ranks starts the N ranks as child
    processes (torch.distributed.run, loopback rendezvous) and rank 0's line comes back on the
    inherited stdout.  Here the ranks are tests/_rank_probe.py with the stand-in block runner."""
    import json
    import subprocess
    import sys

    code = ("import sys; from waveforms_amd.bert import spawn_ranks; "
            f"sys.exit(spawn_ranks({str(ROOT / 'tests' / '_rank_probe.py')!r}, 2, ['--blocks', '5']))")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    blocks = int(np.arange(5).sum())
    assert got["world"] == 2
    assert got["counts"] == [[100 * p * 5 + blocks, 10 * p * 5 + blocks, 5000] for p in range(3)]
    # a failing rank is the caller's exit status
    code = ("import sys; from waveforms_amd.bert import spawn_ranks; "
            f"sys.exit(spawn_ranks({str(ROOT / 'tests' / '_rank_probe.py')!r}, 2, ['--no-such-flag']))")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_bench_gpus_n_self_launches_without_torchrun():
    """bench.py / tools/ber_sweep.py no longer exit with 'launch with torchrun': with --gpus N > 1
    and no WORLD_SIZE they go through spawn_ranks before importing torch."""
    for name in ("bench.py", "tools/ber_sweep.py"):
        src = (ROOT / name).read_text()
        assert "spawn_ranks(" in src and "launch with torchrun" not in src
        assert src.index("spawn_ranks(") < src.index("import torch\n"), name


def test_link_row_bytes_rule():
    """wf_link_layout's row rule (wf_pipeline.hip:link_packed_rows / link_one_kernel) — which link configurations
    carry detector-packed 32 B rows and which run modulator + channel + bank as one kernel (needs no GPU: it reads
    the configuration only)."""
    from waveforms_amd import _hip
    from waveforms_amd.link import SOQPSKLink

    def rb(fuse, sps=8, mf_ntaps=None, nfilt=3, ntaps=None, timing_offset=-1):
        link = SOQPSKLink.__new__(SOQPSKLink)
        cfg = _hip.LinkConfig()
        cfg.nsym, cfg.fuse, cfg.sps, cfg.mf_nfilt = 100_000, fuse, sps, nfilt
        cfg.mf_ntaps = sps + 1 if mf_ntaps is None else mf_ntaps
        cfg.ntaps = 8 * sps + 1 if ntaps is None else ntaps           # the SOQPSK-TG pulse
        cfg.timing_offset = timing_offset
        link.cfg = cfg
        lay = link.layout()
        assert lay["row_bytes"] == link.row_bytes
        return link.row_bytes, lay["one_kernel_front_end"]

    assert rb(7) == (32, 0) and rb(6) == (32, 0)
    assert rb(3) == (48, 0) and rb(0) == (48, 0) and rb(5) == (48, 0)      # bit 2 needs the fused channel (bit 1)
    assert rb(7, sps=4) == (48, 0) and rb(7, nfilt=1) == (16, 0)            # channel + bank PACK form: 3-filter banks at 8 samples per symbol only
    assert rb(7, mf_ntaps=73) == (32, 0)                                    # ... with any tap count (the PAM bank)
    # fuse 15: the one-kernel front end at 8, 10 and 20 samples per symbol with the pulse-truncation bank (sps + 1 taps),
    # and at 8 / 10 with any odd bank of up to 73 / 91 taps (the PAM banks: 73 taps for SOQPSK-TG, 17 for MIL; 91 at the
    # reference example's own 10 samples per symbol)
    assert rb(15) == (32, 1) and rb(15, sps=10) == (32, 1) and rb(15, sps=20) == (32, 1)
    assert rb(15, sps=10, timing_offset=-5) == (32, 1)
    assert rb(15, mf_ntaps=73) == (32, 1) and rb(15, mf_ntaps=17) == (32, 1)
    assert rb(15, mf_ntaps=75) == (32, 0) and rb(15, mf_ntaps=72) == (32, 0)    # longer / even banks: separate kernels, packed rows
    assert rb(15, sps=10, mf_ntaps=91) == (32, 1) and rb(15, sps=10, mf_ntaps=21) == (32, 1)
    assert rb(15, sps=10, mf_ntaps=93) == (48, 0) and rb(15, sps=20, mf_ntaps=181) == (48, 0)
    assert rb(15, sps=16) == (48, 0) and rb(15, sps=4) == (48, 0)
    assert rb(15, sps=10, ntaps=10 * 10 + 1) == (48, 0)                      # a pulse of more than 9 symbols


def test_design_status_block_is_generated_from_committed_profiles():
    """DESIGN.md's status tables are tools/gen_status.py's output for the committed profiles/r04_bench*.json and the built
    library's code-object notes: a figure there cannot drift from its source (round-3 verdict, weak #9)."""
    import subprocess
    import sys

    root = ROOT
    r = subprocess.run([sys.executable, str(root / "tools" / "gen_status.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (root / "DESIGN.md").stat().st_size <= 32 * 1024


def test_long_bank_factorisation():
    """The PAM bank's two forms on the host: ``pam_bank_factors`` (the reference's own: rho pulses and conj(pseudo symbols),
    examples/soqpsk_detection.py:158-173) reproduces ``pam_matched_filter_taps``; ``factor_long_bank`` finds a two-filter
    form from the taps alone and refuses a bank that has none; the packed buffer is b_0, b_1, then the weights."""
    from waveforms_amd.cpm.soqpsk import freq_pulse_soqpsk_mil, freq_pulse_soqpsk_tg
    from waveforms_amd.filters.matched import factor_long_bank, pack_bank_factors, pam_bank_factors, pam_matched_filter_taps

    for sps, pulse in ((8, freq_pulse_soqpsk_tg(8)), (10, freq_pulse_soqpsk_tg(10)), (8, freq_pulse_soqpsk_mil(8))):
        taps = pam_matched_filter_taps(pulse, 0.25, sps)
        basis, w = pam_bank_factors(pulse, 0.25, sps)
        assert basis.shape == (2, taps.shape[1]) and w.shape == (3, 2) and basis.dtype == np.float64
        assert np.abs(w @ basis - taps).max() < 1e-15
        b2, w2 = factor_long_bank(taps)
        assert np.abs(w2 @ b2 - taps).max() < 1e-13 and abs(np.dot(b2[0], b2[1])) < 1e-12
        buf = pack_bank_factors(basis, w)
        n = taps.shape[1]
        assert buf.size == 2 * n + 12 and np.array_equal(buf[:n], basis[0]) and np.array_equal(buf[n:2 * n], basis[1])
        assert buf[2 * n + 2 * (2 * 1 + 0)] == w[1, 0].real and buf[2 * n + 2 * (2 * 2 + 1) + 1] == w[2, 1].imag
    rng = np.random.default_rng(5)
    assert factor_long_bank(rng.standard_normal((3, 73)) + 1j * rng.standard_normal((3, 73))) is None
    assert factor_long_bank(np.ones((2, 9))) is None
    with pytest.raises(ValueError):
        pack_bank_factors(np.ones((3, 9)), np.ones((3, 2)))


def test_fresh_detector_carries_the_reference_state_arrays(golden):
    """waveforms/viterbi/algorithm.py:25-42: every detector instance has bi_history / metrics / path; before the first call
    they are the reference's zeros (no device needed), read-only snapshots here."""
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    g = golden("detector_state")
    for length in (1, 2, 5, 16):
        det = SOQPSKTrellisDetector(length, differantial_encoding=False)
        for name in ("bi_history", "metrics", "path"):
            got, want = getattr(det, name), g[f"L{length}_diff0_k0_{name}"]
            assert got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got, want)
        assert det.fsm.states == 4 and det.state_exp_term == [1j, -1, 1, -1j] and det.i == 0 and det.length == length
