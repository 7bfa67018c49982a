"""Caller-level parity: the reference's only real caller counts
magic series with forward -> pointwise product -> inverse on a fixed 2^15-point
transform, for eight moduli (examples/magic-series/test-magic-series.cpp:22-39,299-333).
tests/magic_series.py does the same job through the engine; the known answers are the
exact counts in tests/golden/magic_series.json (which agree with the reference's).

CPU tier: the workload on the oracle -- validates the workload code and the fixture.
GPU tier: the workload on the engine, every modulus x every order of the reference's test.
"""
import json
import os

import numpy as np
import pytest

from tests import magic_series as ms

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "magic_series.json")) as f:
    KAT = json.load(f)
MODULI = [(m["name"], int(m["modulus"], 16), int(m["generator"], 16)) for m in KAT["moduli"]]
ORDERS = sorted(int(m) for m in KAT["counts"])
NTT_LEN = KAT["ntt_length"]


def test_fixture_is_the_references():
    assert KAT["agrees_with_reference_kats"] is True
    assert ORDERS == [10, 25, 35, 42, 100] and len(MODULI) == 9


@pytest.mark.parametrize("name,p,g", MODULI, ids=[m[0] for m in MODULI])
def test_qpochhammer_coefficients(name, p, g):
    for k, want in KAT["qpochhammer"].items():
        got = ms.one_minus_q_powers(range(1, int(k) + 1), len(want), p)
        assert [int(x) for x in got] == [w % p for w in want]


@pytest.mark.parametrize("name,p,g", MODULI, ids=[m[0] for m in MODULI])
def test_magic_series_on_oracle(port, name, p, g):
    be = ms.OracleBackend(port, p, g)
    orders = ORDERS if name == "baseline" else [10, 25, 42]
    for m in orders:
        assert ms.magic_series_count(be, m, p, NTT_LEN) == int(KAT["counts"][str(m)]) % p, (name, m)


@pytest.mark.parametrize("name,p,g", [MODULI[0], MODULI[2], MODULI[7], MODULI[8]],
                         ids=[MODULI[i][0] for i in (0, 2, 7, 8)])
def test_reciprocal_gives_restricted_partitions(port, name, p, g):
    """1 / prod (1 - q^j) by Newton's iteration on transforms == the partition numbers
    p(i, k) of the reference's RestrictedPartition test (test-magic-series.cpp:96-143)."""
    be = ms.OracleBackend(port, p, g)
    for k, want in KAT["restricted_partitions"].items():
        length = 128 if len(want) > 64 else 64
        den = ms.one_minus_q_powers(range(1, int(k) + 1), length, p)
        rec = ms.reciprocal(be, den, length, p)
        assert [int(x) for x in rec[:len(want)]] == [w % p for w in want[:length]], (name, k)


def test_reciprocal_is_a_reciprocal(port):
    p, g = MODULI[-1][1:]
    be = ms.OracleBackend(port, p, g)
    den = ms.one_minus_q_powers(range(1, 30), 512, p)
    rec = ms.reciprocal(be, den, 512, p)
    full = be.inverse(be.pointwise(be.forward(np.concatenate([den, np.zeros(512, np.uint64)])),
                                   be.forward(np.concatenate([rec, np.zeros(512, np.uint64)]))))
    assert int(full[0]) == 1 and not full[1:512].any()


@pytest.mark.gpu
@pytest.mark.parametrize("name,p,g", MODULI, ids=[m[0] for m in MODULI])
def test_restricted_partitions_on_engine(name, p, g):
    import sve_ntt_amd as eng
    be = ms.EngineBackend(eng, p, g)
    for k, want in KAT["restricted_partitions"].items():
        length = 128 if len(want) > 64 else 64
        den = ms.one_minus_q_powers(range(1, int(k) + 1), length, p)
        rec = ms.reciprocal(be, den, length, p)
        assert [int(x) for x in rec[:len(want)]] == [w % p for w in want[:length]], (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name,p,g", MODULI, ids=[m[0] for m in MODULI])
def test_magic_series_on_engine(name, p, g):
    import sve_ntt_amd as eng
    be = ms.EngineBackend(eng, p, g)
    for m in ORDERS:
        assert ms.magic_series_count(be, m, p, NTT_LEN) == int(KAT["counts"][str(m)]) % p, (name, m)
