"""CPU: the packed factor image's format as restated in tests/q20_reference.py (the encoder of csrc/foldq.hip row by row,
and the decoder that defines the bits) keeps its contract — every row decodes to within its own error weight — on the
shapes the scoring pass uses, including the rows the format has to work for: zero rows, a bracket's largest entry in an
extra (digit-carried) column, one-column and full-width ranks."""
import numpy as np
import pytest

import q20_reference as q20


@pytest.mark.parametrize('K', [1, 10, 12, 13, 25, 50, 51, 100, 101, 200, 202])
def test_restated_encoder_keeps_the_error_weight_contract(K):
    rng = np.random.default_rng(K)
    n = 1500
    V = rng.standard_normal((n, K)) * ((np.arange(n) + 1.0) ** -0.5)[:, None]
    V[7] = 0.0
    V[100, K - 1] = np.abs(V[64:128]).max() * 1.5
    V[101, K - 1] = -np.abs(V[64:128]).max() * 1.4
    bits, tab = q20.encode(V)
    assert bits.shape == (n, q20.lanes(K) * 16) and bits.dtype == np.uint8
    dec = q20.decode(bits, tab, K)
    err = np.linalg.norm(dec[:, :K] - V, axis=1)
    assert (err <= dec[:, K] * 2.0 ** -24).all()
    vn = np.linalg.norm(V, axis=1)
    ok = vn > 0
    # about half a 20-bit step of the bracket's scale per entry: 2^24 * (4.3 / 2^19) / sqrt(12) ~ 40 times an fp32 rounding
    assert 10.0 < np.median(dec[ok, K] / vn[ok]) < 80.0
    assert (dec[7, :K] == 0).all() or np.abs(dec[7, :K]).max() <= tab[q20.bracket(7)] * 4096.0


def test_brackets_are_contiguous_index_ranges_four_per_octave():
    j = np.arange(1 << 16)
    b = q20.bracket(j)
    assert (np.diff(b) >= 0).all() and (np.diff(b) <= 1).all() and b[0] == 0 and b.max() < q20.TAB
    assert q20.bracket((1 << 24) - 1) < q20.TAB
    # four brackets per octave from 4 on: rows 2^e .. 2^(e+1) - 1 split into quarters
    for e in range(2, 16):
        assert len(np.unique(b[1 << e:1 << (e + 1)])) == 4
    assert [q20.lanes(k) for k in (1, 12, 13, 25, 26, 50, 51, 101, 102, 202, 203)] == [2, 2, 4, 4, 8, 8, 16, 16, 32, 32, 0]
