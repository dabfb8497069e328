"""The reference's own unit test (tests/test_glfsr.py:6-25 there), restated: for every LFSR
order 2..22 two consecutive full-length runs of PNSequence are identical (the sequence has
period 2**n - 1 and the register returns to its start).  Once on the CPU oracle, once through
the GPU kernel."""
import numpy as np
import pytest

ORDERS = list(range(2, 23))


@pytest.mark.parametrize("order", ORDERS)
def test_sequence_period_oracle(oracle, order):
    n = (1 << order) - 1
    both, state = oracle.glfsr_bits(oracle.lfsr_mask(order), n, 2 * n)
    assert np.array_equal(both[:n], both[n:]) and state == n
    assert int(both[:n].sum()) == 1 << (order - 1)      # maximal length: 2^(n-1) ones


@pytest.mark.gpu
@pytest.mark.parametrize("order", ORDERS)
def test_sequence_period_gpu(order):
    from waveforms.glfsr import PNSequence

    pn = PNSequence(order)
    first = pn.generate_sequence()
    second = pn.generate_sequence()
    assert len(first) == (1 << order) - 1 and first == second
    assert pn.state == (1 << order) - 1
