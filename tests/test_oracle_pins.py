"""Pins the CPU oracle (oracle/swe_oracle.c) to the reference:
 (1) the Roe-flux known-answer vectors obtained from the reference arithmetic
     (tests/golden/roe_kat.json), bit for bit;
 (2) the reference's own accuracy gate for this path -- all nine MMS
     convergence rates of driver/tests/swe_roe/mms_conv_study.yaml:48-64 must
     exceed the thresholds the reference's CI enforces (src/rdymms.c:1004-1041).
"""
import json
import os

import numpy as np

from oracle import oracle as O
import mms

HERE = os.path.dirname(os.path.abspath(__file__))


def test_roe_flux_known_answers_bitwise():
    with open(os.path.join(HERE, "golden", "roe_kat.json")) as fh:
        kat = json.load(fh)
    for v in kat["vectors"]:
        f, amax = O.roe_flux(*v["left"], *v["right"], v["sn"], v["cn"])
        assert [float(x) for x in f] == v["flux"], (f, v)
        assert float(amax) == v["amax"]


def test_mms_convergence_rates_exceed_reference_thresholds():
    rates = mms.convergence_rates(mms.oracle_make_apply)
    for comp, expected in mms.EXPECTED.items():
        for got, thr, norm in zip(rates[comp], expected, ("L1", "L2", "Linf")):
            assert np.isfinite(got) and got > thr, f"{norm} rate for {comp}: {got} (expected > {thr})"
    # the thresholds are the reference's own rates cut to two digits; the oracle lands within 0.01 of each
    for comp, expected in mms.EXPECTED.items():
        for got, thr in zip(rates[comp], expected):
            assert got - thr < 0.01
