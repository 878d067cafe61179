"""Cross-examination of the C oracle (it cannot be pinned to reference outputs: the reference ships none and cannot be built here).

oracle/restatement_np.py restates the path a SECOND time, independently (dense numpy algebra written from the reference sources, inertial
table parsed from the text of the reference's robotParameters.cpp, generic KKT active-set solver).  Where /root/reference exists (the build
container) it must reproduce the committed oracle fixtures; everywhere, the anchors it wrote (tests/golden/survey_anchors.json) must agree
with the C oracle and with the numbers SURVEY.md 8c quotes."""
import json
import os

import numpy as np
import pytest

from oracle import restatement_np as R
from oracle.pyoracle import Oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
needs_ref = pytest.mark.skipif(not R.available(), reason="/root/reference is only present in the build container")


@needs_ref
def test_independent_restatement_reproduces_the_golden_eval_vectors():
    w = R.check_against_golden(verbose=False)
    for k in ("M", "C", "AG", "J", "CoM", "Jpqp", "u0", "Cg6", "AGpqp"):
        assert w[k] < 1e-12, (k, w[k])                             # model terms: two independent codings agree to round-off
    for k in ("tau", "f", "qpp", "a"):
        assert w[k] < 1e-8, (k, w[k])                              # through two different QP algorithms (KKT active set vs Goldfarb-Idnani)


@needs_ref
def test_inertial_table_parsed_from_the_reference_text_equals_the_oracle_table():
    from oracle.pyoracle import nao_raw_links
    raw = nao_raw_links()
    links = R.parse_links()
    for i in range(28):
        assert links[i]["mass"] == raw[i, 0] and np.array_equal(links[i]["com"], raw[i, 1:4]) and np.array_equal(links[i]["inertia"].ravel(), raw[i, 4:])


@needs_ref
def test_anchors_regenerate():
    a = R.survey_anchors(ticks=0)
    b = json.load(open(os.path.join(GOLD, "survey_anchors.json")))
    for k, v in a.items():
        assert np.allclose(v, b[k], rtol=1e-9, atol=1e-12), k


def test_committed_anchors_match_the_oracle_and_the_survey():
    a = json.load(open(os.path.join(GOLD, "survey_anchors.json")))
    # SURVEY.md 8c (scratch numpy transliteration of the survey session), to the digits quoted there
    assert abs(a["total_mass"] - 5.30539) < 1e-12
    assert np.allclose(a["com_initial_configuration"], [-8.706522035742e-3, 0.0, 0.2580191806516], atol=1e-12)
    assert abs(a["D"] - (-0.0265035677879715)) < 1e-15 and abs(a["K0"] - (-14.39564980491)) < 1e-10 and abs(a["K1"] - 1.134304133175) < 1e-11
    assert abs(a["K_sum"] - 20.85545314111) < 1e-10 and np.allclose(a["K_Px"], [20.85545314111, 7.706753397498], atol=1e-10)
    assert abs(a["tick0_u0x"] - 0.417109) < 1e-6
    assert np.allclose(a["tick0_f"], [-7.180185e-5, 0.7950184837, -2.398258e-4, 1.199450113, -3.7e-9, 25.92505445,
                                       -7.180185e-5, 0.7950184837, -2.398258e-4, 1.199474099, -3.7e-9, 25.92504727], atol=2e-9)
    assert abs(a["tick0_tau_rknee"] - (-1.014091330871)) < 1e-10 and abs(a["tick0_tau_lknee"] - (-1.014083780437)) < 1e-10
    assert abs(a["tick0_C5"] - 52.04299827) < 1e-8 and abs(a["tick0_AG44"] - 5.305096663) < 1e-9
    assert 0.203 < a["tick0_c_min"] and a["tick0_c_max"] < 3.04 and a["tick0_active"] == 0 and a["base_row_residual"] < 1e-12
    assert a["ticks"] == 500 and abs(a["com_x_after_ticks"] - (-1.34e-4)) < 1e-6 and abs(a["sum_fz_after_ticks"] - 52.04998) < 1e-5
    assert np.allclose(a["ik_posture_com"], [-0.02, 0.0, 0.26], atol=1e-10)
    assert np.allclose(a["ik_posture_right_sole"], [0, -0.05, 0], atol=1e-10) and np.allclose(a["ik_posture_left_sole"], [0, 0.05, 0], atol=1e-10)
    # the C oracle on the same inputs
    o = Oracle(sim_time=5.0, dt=0.01, horizon_time=0.5, do_ik=True)
    assert abs(o.mass - a["total_mass"]) < 1e-14
    r = o.robot()
    e = o.eval(r["q"], np.zeros(30), 0.0)
    assert np.allclose(e["f"], a["tick0_f"], rtol=0, atol=1e-8 * 26) and abs(e["tau"][3] - a["tick0_tau_rknee"]) < 1e-9
    assert abs(o.terms()["C"][5] - a["tick0_C5"]) < 1e-12 and abs(o.terms()["AG"][4, 4] - a["tick0_AG44"]) < 1e-13
    ro = o.rollout(np.concatenate([r["q"], np.zeros(30)]), 0.0, 500, log=True)
    assert abs(ro["comx"][-1] - a["com_x_after_ticks"]) < 1e-9
    assert abs(ro["log"][-1][24 + 5] + ro["log"][-1][24 + 11] - a["sum_fz_after_ticks"]) < 1e-7
