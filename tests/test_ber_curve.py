"""BER-vs-Eb/N0: the oracle reproduces the reference's golden counts exactly (CPU), and the
GPU link's curve sits within +-0.05 dB of the reference's (metric: "BER-curve delta vs ref")."""
import csv
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
BLOCK = 1 << 17


def golden_rows():
    rows = {}
    for name in ("ber_golden.csv", "ber_golden_hi.csv", "ber_golden_hi2.csv"):
        path = ROOT / "tests" / "golden" / name
        if path.exists():
            for r in csv.DictReader(open(path)):
                rows[(int(r["ebn0_db"]), int(r["block"]))] = {k: int(v) for k, v in r.items()}
    return rows


@pytest.mark.parametrize("ebn0,block", [(3, 0), (8, 5), (11, 15)])
def test_oracle_reproduces_reference_ber_blocks(oracle, ebn0, block):
    """Same PN23 block, same PCG64 seed as tests/golden/make_ber_golden.py -> identical counts."""
    g = golden_rows()[(ebn0, block)]
    bits = oracle.glfsr_bits(oracle.lfsr_mask(23), (1 << 23) - 1, (block + 1) * BLOCK)[0][block * BLOCK:]
    pulse = oracle.freq_pulse_soqpsk_tg(8)
    for kind, off, key in (("PT", -1, "pt"), ("PAM", 0, "pam")):
        rng = np.random.Generator(np.random.PCG64(seed=1000 * ebn0 + block))
        res = oracle.detection_run(bits, pulse, 0.25, 8, oracle.sigma_for_ebn0(ebn0, 8), rng=rng,
                                   detector=kind, timing_offset=off)
        assert (res["compared"], res["sym_errors"], res["bit_errors"]) == \
            (g[f"{key}_compared"], g[f"{key}_sym_err"], g[f"{key}_bit_err"])


def test_golden_curve_is_monotone_and_crosses_targets():
    from ber_sweep import golden_curve

    from waveforms.bert import ebn0_at_ber

    for det in ("PT", "PAM"):
        e, ber, errs, n = golden_curve(det)
        assert e.tolist() == list(range(13)) and np.all(np.diff(ber) < 0)
        assert 8 < ebn0_at_ber(e, ber, 1e-3) < 11 and 9 < ebn0_at_ber(e, ber, 1e-4) < 12.5


@pytest.mark.gpu
@pytest.mark.parametrize("detector", ["PT", "PAM"])
def test_gpu_ber_curve_within_0p05_db_of_reference(detector):
    """Blocks of 2^17 symbols like the reference runs (same edge effects), 2^27 symbols per
    point on the GPU (its sampling noise is then negligible); the tolerance is the plain 0.05 dB
    BASELINE.json states."""
    from ber_sweep import golden_curve

    from waveforms.bert import SweepPlan, ber_sweep, ebn0_at_ber

    ebn0 = list(range(0, 13))           # all 13 points of BASELINE configs[3] (0 .. 12 dB)
    plan = SweepPlan(ebn0_db=ebn0, blocks_per_point=1024, nsym=BLOCK, detector=detector)
    counts = ber_sweep(plan, rank=0, world=1)
    ber = counts[:, 1] / counts[:, 2]
    ge, gb, gerr, _ = golden_curve(detector)
    # point-wise: consistent with the reference within 4 sigma of its Poisson noise
    for k, e in enumerate(ebn0):
        j = int(np.where(ge == e)[0][0])
        sigma = gb[j] / np.sqrt(max(gerr[j], 1))
        assert abs(ber[k] - gb[j]) < 4 * sigma + 0.01 * gb[j], (e, ber[k], gb[j], sigma)
    for target in (1e-3, 1e-4):
        mine, ref = ebn0_at_ber(ebn0, ber, target), ebn0_at_ber(ge, gb, target)
        # dB uncertainty of the reference crossing: d(dB) = d(log10 BER) / |slope|
        j = int(np.searchsorted(-np.log10(gb), -np.log10(target)))
        slope = abs(np.log10(gb[j]) - np.log10(gb[j - 1]))            # decades per dB
        sig_db = (1 / np.log(10)) / np.sqrt(min(gerr[j - 1], gerr[j])) / slope
        print(f"{detector} BER {target:g}: GPU {mine:.3f} dB, reference {ref:.3f} dB, delta {mine - ref:+.3f} dB "
              f"(reference 1-sigma {sig_db:.3f} dB)")
        # the stated bar, outright: both curves are deterministic (fixed seeds, committed reference
        # counts), so this is not a statistical acceptance test; the reference's own 1-sigma at the
        # crossing (0.007 - 0.015 dB with the 4.4e7 reference symbols per point committed for
        # 10 - 12 dB) is printed above for the reader, not added to the tolerance
        assert abs(mine - ref) <= 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("waveform,warmup,ebn0", [("soqpsk", 1, 0.0), ("multih", 16, 2.0), ("pcmfm", 8, 0.0)])
def test_gpu_sweep_counts_do_not_depend_on_the_warmup(waveform, warmup, ebn0):
    """A sweep whose detector warm-up is far too short (chunks do not merge: the launch's own proof fails) ends
    with exactly the counts of a sweep with the default warm-up: the chunks that missed are run again on the
    device from the true state, cascading into the following chunks where needed — nothing is repeated on the host,
    nothing raises."""
    from waveforms.bert import SweepPlan, gpu_block_runner

    def sweep(wu):
        plan = SweepPlan(ebn0_db=[ebn0, 6.0], blocks_per_point=3, nsym=1 << 19, waveform=waveform, warmup=wu)
        run, finish = gpu_block_runner(plan, streams=2)
        for point, block in plan.shard(0, 1):
            run(point, block)
        return finish(), run.stats

    good, st_good = sweep(0)
    short, st_short = sweep(warmup)
    assert st_short["repaired_chunks"] > st_good["repaired_chunks"], "the short warm-up was expected to send chunks to the repair"
    assert np.array_equal(good, short), (good, short)
