"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes -- through
size-independent properties of the problem.

Tolerances (north_star): 1e-6 relative on torques / forces (absolute floor 1e-9 * max|.|),
bit-exact on the preview index k.  fp64 throughout.
"""
import os

import numpy as np
import pytest
import torch

from helpers import TOL_REL, WEIGHT, close, close_on, dense_terms_from_debug, oracle_system, perturbed_velocities, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def make_controller(B, dt, th, zcom, **kw):
    from linearmpchumanoid_amd.controller import BatchedController, default_config
    ctl = BatchedController(B, default_config(dt=dt, time_horizon=th, z_com=zcom, **kw))
    return ctl


@pytest.fixture(scope="module")
def cfg2():
    """BASELINE config 2 constants: dt = 1 ms, N = 16."""
    o = oracle_system(1e-3, 0.016)
    return dict(dt=1e-3, th=0.016, zcom=o.zcom, q0=o.robot()["q"].copy())


def test_extension_loaded_is_the_hip_library():
    from linearmpchumanoid_amd import capi
    assert os.path.exists(capi.SO_PATH)
    assert capi.lib().lmh_device_count() >= 1


def test_model_setup_matches_oracle(cfg2):
    """Robot ctor re-expression + total mass (Robot.cpp:14-22) computed on the GPU."""
    ctl = make_controller(2, cfg2["dt"], cfg2["th"], cfg2["zcom"])
    o = oracle_system(cfg2["dt"], cfg2["th"])
    assert abs(ctl.mass()[0] - o.mass) < 1e-15
    assert rel_err(ctl.mpc_gain(), o.gain_row()) < 1e-12


def test_stage_parity_single_evaluation(cfg2):
    """Unit parity per stage: T, X, C, Cg, M, AG, AGpqp, Jpqp, J, CoM, MPC u0, PD refs, QP x, tau, f, qdd."""
    from linearmpchumanoid_amd.controller import unpack_debug
    B = 24
    v = perturbed_velocities(B); v[0] = 0
    vprev = perturbed_velocities(B, seed=555)
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
    ctl.set_refs_stance(2.0, 2)
    st = ctl.new_state(cfg2["q0"], v, t=0.0, v_prev=vprev)
    out, status, dbg = ctl.stand_step(st, debug=True)
    torch.cuda.synchronize()
    out, status, dbg, stn = out.cpu().numpy(), status.cpu().numpy(), dbg.cpu().numpy(), st.cpu().numpy()
    mask_mismatch = 0
    for i in range(B):
        o = oracle_system(cfg2["dt"], cfg2["th"])
        o.set_prev_velocity(vprev[i])
        e = o.eval(cfg2["q0"], v[i], 0.0)
        t, qp, rb = o.terms(), o.qp(), o.robot()
        d = unpack_debug(dbg[i]); dd = dense_terms_from_debug(d)
        tight = dict(T=(dd["T"], t["T"]), X=(dd["X"], t["X"]), C=(d["C"], t["C"]), M=(dd["M"], t["M"]), AG=(d["AG"], t["AG"]),
                     J=(dd["J"], t["J"]), CoM=(d["CoM"], rb["CoM"]), qref=(d["qppRef"], qp["qppRef"]), href=(d["hGpRef"], qp["hGpRef"]))
        for name, (a, b) in tight.items():
            assert rel_err(a, b) < 1e-11, (i, name, rel_err(a, b))
        # velocity-product terms are differences of O(50) quantities: compare on that scale
        assert np.abs(d["Cg6"] - t["Cg"][:6]).max() < 1e-11 * np.abs(t["C"]).max()
        assert np.abs(d["AGpqp"] - t["AGpqp"]).max() < 1e-11 * np.abs(t["C"]).max()
        assert close_on(d["Jpqp"], t["Jpqp"], 1e-11, np.abs(t["C"]).max())
        # PD references of a robot standing on its references: differences of kp * position (500 x 0.05 m) and kd * velocity terms
        assert close_on(d["footAccRef"], qp["footAccRef"], 1e-10, 500.0 * 0.05 + np.abs(qp["footAccRef"]).max())
        # u0 = -K (Px x - z): sum of K_i x_com terms, |K|_1 |x| ~ (g / z_c) |x_com|
        assert close_on(d["mpc"][:2], qp["u0"], 1e-11, 9.81 / 0.26 * 0.05 + np.abs(qp["u0"]).max())
        assert rel_err(d["a"], qp["x"][:30]) < 1e-8
        assert rel_err(out[i, :24], e["tau"]) < TOL_REL and rel_err(out[i, 24:36], e["f"]) < TOL_REL
        assert rel_err(out[i, 36:66], e["qpp"]) < TOL_REL
        assert status[i, 0] == e["k"] and status[i, 2] == 0
        mask_mismatch += int((int(status[i, 3]) & 0xFFFFFFFF) != e["active_mask"])
        assert np.array_equal(stn[i, 60:90], v[i])           # Robot::v_ <- dq
    assert mask_mismatch <= B // 6                           # degenerate (c_j == 0) ties may differ


def test_golden_eval_vectors_reference_literals():
    """Committed fixtures at the reference's own dt = 0.01 / N = 50 (apps/offline literals)."""
    g = np.load(os.path.join(GOLD, "eval_vectors.npz"))
    B = g["q"].shape[0]
    ctl = make_controller(B, float(g["dt"]), float(g["time_horizon"]), float(g["z_com"]), warm_start=0)
    ctl.set_refs_stance(5.0, 2)
    st = ctl.new_state(g["q"], g["v"], t=float(g["t"]), v_prev=g["v_prev"])
    out, status = ctl.stand_step(st)
    torch.cuda.synchronize()
    out, status = out.cpu().numpy(), status.cpu().numpy()
    for i in range(B):
        assert status[i, 0] == int(g["k"][i]) and status[i, 2] == 0
        assert rel_err(out[i, :24], g["tau"][i]) < TOL_REL
        assert rel_err(out[i, 24:36], g["f"][i]) < TOL_REL
        assert rel_err(out[i, 36:66], g["qpp"][i]) < TOL_REL


def test_rollout_parity_config2_fixture():
    """RK4 closed loop (stale Robot::v_, float-accumulated clock) vs the committed oracle rollout."""
    g = np.load(os.path.join(GOLD, "rollout_config2.npz"))
    B, nt = g["v"].shape[0], g["log"].shape[1]
    for warm in (0, 1):
        ctl = make_controller(B, float(g["dt"]), float(g["time_horizon"]), float(g["z_com"]), warm_start=warm)
        ctl.set_refs_stance(2.0, 2)
        st = ctl.new_state(g["q0"], g["v"], t=0.0)
        out, status, log = ctl.rollout(st, nt, log=True)
        torch.cuda.synchronize()
        stn, log, status = st.cpu().numpy(), log.cpu().numpy(), status.cpu().numpy()
        for i in range(B):
            assert status[i, 0] == int(g["k"][i][-1]) and status[i, 2] == 0
            assert close(stn[i, :60], g["state"][i], 1e-8)
            for tk in range(nt):
                assert rel_err(log[tk, i, :24], g["log"][i, tk, :24]) < TOL_REL, (warm, i, tk)
                assert rel_err(log[tk, i, 24:], g["log"][i, tk, 24:]) < TOL_REL, (warm, i, tk)
            assert abs(stn[i, 90] - nt * float(g["dt"])) < 1e-12


def test_offline_app_trace_parity():
    """apps/offline workload (dt = 0.01, N = 50, 120 of the 500 ticks): state, k, CoM-x trace."""
    g = np.load(os.path.join(GOLD, "offline_trace.npz"))
    import json
    ik = json.load(open(os.path.join(GOLD, "ik_posture.json")))
    ctl = make_controller(1, 0.01, 0.5, ik["z_com"], warm_start=0)
    ctl.set_refs_stance(5.0, 2)
    st = ctl.new_state(np.array(ik["q"]), np.zeros(30), t=0.0)
    o = oracle_system(0.01, 0.5, sim_time=5.0)
    r = o.rollout(np.concatenate([np.array(ik["q"]), np.zeros(30)]), 0.0, 120, log=True)
    out, status, log = ctl.rollout(st, 120, log=True)
    torch.cuda.synchronize()
    assert int(status.cpu().numpy()[0, 0]) == int(g["k"][119])
    assert np.abs(st.cpu().numpy()[0, :60] - r["state"]).max() < 1e-9
    assert rel_err(log.cpu().numpy()[-1, 0], r["log"][-1]) < TOL_REL


def test_eval_then_eval_keeps_reference_state_semantics(cfg2):
    """Two consecutive standStep calls: the second sees the first call's dq as Robot::v_."""
    v = perturbed_velocities(4, seed=77)
    ctl = make_controller(4, cfg2["dt"], cfg2["th"], cfg2["zcom"])
    ctl.set_refs_stance(2.0, 2)
    st = ctl.new_state(cfg2["q0"], v, t=0.0)
    ctl.stand_step(st)
    st[:, 30:60] *= 0.5
    out, status = ctl.stand_step(st)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i in range(4):
        o = oracle_system(cfg2["dt"], cfg2["th"])
        o.eval(cfg2["q0"], v[i], 0.0)
        e = o.eval(cfg2["q0"], 0.5 * v[i], 0.0)
        assert rel_err(out[i, :24], e["tau"]) < TOL_REL and rel_err(out[i, 24:36], e["f"]) < TOL_REL


def test_support_phases_and_flight(cfg2):
    """Edge cases: single support (one foot's wrench exactly zero), flight (no contact force at all)."""
    B = 6
    v = perturbed_velocities(B, seed=31) * 0.3
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
    n = 2500
    for ph, dead in ((1, slice(30, 36)), (2, slice(24, 30)), (3, slice(24, 36))):
        ctl.set_refs(np.zeros(n), np.zeros(n), np.full(n, ph, dtype=np.uint8))
        st = ctl.new_state(cfg2["q0"], v, t=0.0)
        out, status = ctl.stand_step(st)
        torch.cuda.synchronize()
        out, status = out.cpu().numpy(), status.cpu().numpy()
        assert (status[:, 2] == 0).all()
        assert np.abs(out[:, dead]).max() == 0.0
        for i in range(B):
            o = oracle_system(cfg2["dt"], cfg2["th"])
            zx, zy = o.zmp()
            o.set_refs(zx, zy, np.full(len(zx), ph, dtype=np.uint8))
            e = o.eval(cfg2["q0"], v[i], 0.0)
            assert rel_err(out[i, :24], e["tau"]) < TOL_REL
            assert close(out[i, 24:36], e["f"], TOL_REL, scale=WEIGHT)


def test_domain_randomised_models(cfg2):
    """Per-instance link parameters (mass x U(0.9,1.1), CoM +- 5 mm) through lmh_set_model."""
    from linearmpchumanoid_amd.controller import nominal_links
    B = 6
    raw = np.tile(nominal_links(), (B, 1, 1))
    rng = np.random.default_rng(20260004)
    raw[:, :, 0] *= rng.uniform(0.9, 1.1, (B, 28))
    raw[:, :, 1:4] += rng.uniform(-5e-3, 5e-3, (B, 28, 3)) * (raw[:, :, 0:1] > 0)
    v = perturbed_velocities(B, seed=41) * 0.2
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
    ctl.set_refs_stance(2.0, 2)
    ctl.set_model(raw)
    st = ctl.new_state(cfg2["q0"], v, t=0.0)
    out, status = ctl.stand_step(st)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    masses = ctl.mass()
    from oracle.pyoracle import Oracle
    for i in range(B):
        o = Oracle(sim_time=2.0, dt=cfg2["dt"], horizon_time=cfg2["th"], do_ik=False, raw_links=raw[i])
        assert abs(masses[i] - o.mass) < 1e-13
        o.set_zcom(cfg2["zcom"])          # same LIPM height as the GPU handle
        e = o.eval(cfg2["q0"], v[i], 0.0)
        assert rel_err(out[i, :24], e["tau"]) < TOL_REL and rel_err(out[i, 24:36], e["f"]) < TOL_REL


def test_ik_matches_oracle_posture():
    """Kinematics::compute on the GPU: same posture and the same 4 Newton steps as the oracle."""
    import json
    ik = json.load(open(os.path.join(GOLD, "ik_posture.json")))
    from oracle.pyoracle import Oracle
    q_init = Oracle(do_ik=False).robot()["q"]
    ctl = make_controller(3, 0.01, 0.5, 0.26)
    q = torch.as_tensor(np.tile(q_init, (3, 1))).to(ctl.device)
    q, iters = ctl.ik(q)
    torch.cuda.synchronize()
    qn = q.cpu().numpy()
    assert (iters.cpu().numpy() == 4).all()
    assert np.abs(qn - np.array(ik["q"])).max() < 1e-11


# ------------------------------------------------------------------------------- full size
def test_full_size_properties_config2(cfg2):
    """B = 1024 (BASELINE configs[1]): properties that hold for every instance without an oracle run:
    floating-base rows of the dynamics vanish at the QP solution, contact wrenches lie in the
    friction cones, instance order does not matter, warm start == cold start."""
    from linearmpchumanoid_amd.controller import unpack_debug
    B = 1024
    v = perturbed_velocities(B)
    res = {}
    for warm in (0, 1):
        ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=warm)
        ctl.set_refs_stance(2.0, 2)
        st = ctl.new_state(cfg2["q0"], v, t=0.0)
        out, status, _ = ctl.rollout(st, 5)
        out2, status2, dbg = ctl.stand_step(st, debug=True)
        torch.cuda.synchronize()
        res[warm] = (out.cpu().numpy(), st.cpu().numpy(), out2.cpu().numpy(), dbg.cpu().numpy(), status2.cpu().numpy())
    out_c, st_c, ev_c, dbg_c, status_c = res[0]
    out_w, st_w, ev_w, _, _ = res[1]
    assert (status_c[:, 2] == 0).all()
    # warm start reaches the same (unique) minimiser
    assert np.abs(out_w[:, :36] - out_c[:, :36]).max() < 1e-7 * np.abs(out_c[:, :36]).max()
    assert np.abs(st_w[:, :60] - st_c[:, :60]).max() < 1e-9 * np.abs(st_c[:, :60]).max()
    mu = 0.7
    for i in range(0, B, 7):
        d = unpack_debug(dbg_c[i]); dd = dense_terms_from_debug(d)
        x = d["a"]; f = ev_c[i, 24:36]
        resid = dd["M"][:6] @ x + d["C"][:6] - dd["J"].T[:6] @ f
        assert np.abs(resid).max() < 1e-8 * np.abs(d["C"][:6]).max()
        assert d["c"].min() > -1e-7 * max(1.0, d["c"].max())
        for ft in range(2):
            fx, fy, fz = f[6 * ft + 3: 6 * ft + 6]
            assert fz > -1e-7 and abs(fx) <= mu * fz + 1e-7 and abs(fy) <= mu * fz + 1e-7
    # permutation invariance: reversed instance order gives reversed results, bit for bit
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
    ctl.set_refs_stance(2.0, 2)
    st = ctl.new_state(cfg2["q0"], v[::-1].copy(), t=0.0)
    out_r, _, _ = ctl.rollout(st, 5)
    torch.cuda.synchronize()
    assert np.array_equal(out_r.cpu().numpy()[::-1], out_c)


def test_full_size_sample_against_oracle(cfg2):
    """B = 1024, 4 ticks: every 64th instance checked against its own oracle rollout."""
    B = 1024
    v = perturbed_velocities(B)
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=1)
    ctl.set_refs_stance(2.0, 2)
    st = ctl.new_state(cfg2["q0"], v, t=0.0)
    out, status, _ = ctl.rollout(st, 4)
    torch.cuda.synchronize()
    out, stn, status = out.cpu().numpy(), st.cpu().numpy(), status.cpu().numpy()
    for i in range(0, B, 64):
        o = oracle_system(cfg2["dt"], cfg2["th"])
        r = o.rollout(np.concatenate([cfg2["q0"], v[i]]), 0.0, 4, log=True)
        assert status[i, 0] == r["k"][-1]
        assert rel_err(out[i, :24], r["log"][-1][:24]) < TOL_REL and rel_err(out[i, 24:36], r["log"][-1][24:]) < TOL_REL
        assert np.abs(stn[i, :60] - r["state"]).max() < 1e-8


def test_walking_contact_switching_parity(cfg2):
    """BASELINE config-3 ingredients at small scale: piecewise ZMP, per-sample support phase (contact
    switching), swing-foot polynomial segments, per-instance step length.  4 instances x 700 ticks
    (settle, double support, one single-support swing) against the oracle; k and phase bit-exact."""
    from linearmpchumanoid_amd import trajectories
    from oracle.pyoracle import Oracle
    dt, N = 1e-3, 32
    th = N * dt
    nt = 700
    plan = trajectories.walk_plan(2.0, dt, num_steps=2, time_per_step=0.4, ds_time=0.1, step_height=0.02, settle_time=0.15)
    xs = np.array([0.02, 0.03, 0.04, 0.05])
    B = len(xs)
    ctl = make_controller(B, dt, th, cfg2["zcom"], warm_start=1)
    ctl.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
    ctl.set_segments(plan["segs"], plan["seg_of_sample"])
    ctl.set_xscale(xs)
    st = ctl.new_state(cfg2["q0"], np.zeros(30), t=0.0)
    out, status, log = ctl.rollout(st, nt, log=True)
    torch.cuda.synchronize()
    stn, log, status = st.cpu().numpy(), log.cpu().numpy(), status.cpu().numpy()
    assert (status[:, 2] == 0).all()
    saw_single_support = False
    for i in range(B):
        o = Oracle(sim_time=2.0, dt=dt, horizon_time=th, do_ik=True)
        o.set_zcom(cfg2["zcom"])
        o.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        o.set_segments(plan["segs"], plan["seg_of_sample"], xscale=float(xs[i]))
        r = o.rollout(np.concatenate([cfg2["q0"], np.zeros(30)]), 0.0, nt, log=True)
        assert status[i, 0] == r["k"][-1]
        assert close(stn[i, :60], r["state"], 1e-7)
        for tk in range(0, nt, 7):
            ref = r["log"][tk]
            assert close(log[tk, i, :24], ref[:24], TOL_REL), (i, tk)
            assert close(log[tk, i, 24:], ref[24:], TOL_REL, scale=WEIGHT), (i, tk)
            if np.abs(ref[24 + 6:24 + 12]).max() < 1e-9 and np.abs(ref[24:24 + 6]).max() > 1.0:
                saw_single_support = True
                assert np.abs(log[tk, i, 24 + 6:]).max() == 0.0     # swing foot carries exactly no force
    assert saw_single_support


def test_jump_schedule_parity_config5_ingredients(cfg2):
    """BASELINE config-5 ingredients at small scale: N = 48, double support -> flight -> double support.
    Rollout across both contact switches against the oracle; k bit-exact, contact wrenches exactly zero in
    flight, tau / f within 1e-6."""
    from linearmpchumanoid_amd import trajectories
    from oracle.pyoracle import Oracle
    dt, N = 1e-3, 48
    th = N * dt
    plan = trajectories.jump_plan(0.5, dt, stance_time=0.03, flight_time=0.04)
    nt = 110                                                       # 30 ticks stance, 40 flight, 40 stance again
    B = 4
    v = perturbed_velocities(B, seed=61) * 0.2
    ctl = make_controller(B, dt, th, cfg2["zcom"], warm_start=1)
    ctl.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
    st = ctl.new_state(cfg2["q0"], v, t=0.0)
    out, status, log = ctl.rollout(st, nt, log=True)
    torch.cuda.synchronize()
    stn, log, status = st.cpu().numpy(), log.cpu().numpy(), status.cpu().numpy()
    assert (status[:, 2] == 0).all()
    flight_ticks = 0
    for i in range(B):
        o = Oracle(sim_time=0.5, dt=dt, horizon_time=th, do_ik=True)
        o.set_zcom(cfg2["zcom"])
        o.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        r = o.rollout(np.concatenate([cfg2["q0"], v[i]]), 0.0, nt, log=True)
        assert status[i, 0] == r["k"][-1]
        assert close(stn[i, :60], r["state"], 1e-7)
        for tk in range(nt):
            ref = r["log"][tk]
            assert close(log[tk, i, :24], ref[:24], TOL_REL), (i, tk)
            assert close(log[tk, i, 24:], ref[24:], TOL_REL, scale=WEIGHT), (i, tk)
            if plan["phase"][r["k"][tk]] == 3:                      # k of the logged (stage-4) evaluation
                flight_ticks += 1
                assert np.abs(log[tk, i, 24:]).max() == 0.0        # no contact force at all, exactly
                assert np.abs(ref[24:]).max() < 1e-9
    assert flight_ticks >= B * 30


def _walk_setup(ctl, plan, xs):
    ctl.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
    ctl.set_segments(plan["segs"], plan["seg_of_sample"])
    ctl.set_xscale(xs)


def test_full_size_walking_config3(cfg2):
    """BASELINE configs[2] at full size: 4096 instances, N = 32, walking with contact switching and a
    per-instance step length U(0.02, 0.05) (seed 20260003 + i).  Size-independent properties for every
    instance (no flags, swing foot carries exactly no force, contact forces inside the friction cone, k
    bit-exact against the float-accumulated clock, instance order irrelevant) + every 512th instance against
    its own oracle rollout."""
    from linearmpchumanoid_amd import trajectories
    from oracle.pyoracle import Oracle
    dt, N, B, nt = 1e-3, 32, 4096, 460
    th = N * dt
    plan = trajectories.walk_plan(1.0, dt, num_steps=2, time_per_step=0.2, ds_time=0.05, step_height=0.02, settle_time=0.1)
    xs = np.array([np.random.default_rng(20260003 + i).uniform(0.02, 0.05) for i in range(B)])
    ctl = make_controller(B, dt, th, cfg2["zcom"], warm_start=1)
    _walk_setup(ctl, plan, xs)
    st = ctl.new_state(cfg2["q0"], np.zeros(30), t=0.0)
    out, status, log = ctl.rollout(st, nt, log=True)
    torch.cuda.synchronize()
    stn, log, status = st.cpu().numpy(), log.cpu().numpy(), status.cpu().numpy()
    assert (status[:, 2] == 0).all()
    # k of the last evaluation (stage 4 of the last tick): int((t + dt)/dt) on the float-accumulated clock
    t = 0.0
    for _ in range(nt - 1):
        t += dt
    assert (status[:, 0] == int((t + dt) / dt)).all()
    # support phase per tick from the same k sequence (stage-4 evaluation is what the log holds)
    t = 0.0
    mu = 0.7
    seen = set()
    for tk in range(nt):
        ph = int(plan["phase"][int((t + dt) / dt)])
        seen.add(ph)
        f = log[tk, :, 24:36]
        if ph == 1:
            assert np.abs(f[:, 6:]).max() == 0.0                   # right support: the left (swing) foot carries nothing
        if ph == 2:
            assert np.abs(f[:, :6]).max() == 0.0
        for ft in range(2):
            fx, fy, fz = f[:, 6 * ft + 3], f[:, 6 * ft + 4], f[:, 6 * ft + 5]
            assert (fz > -1e-7).all() and (np.abs(fx) <= mu * fz + 1e-7).all() and (np.abs(fy) <= mu * fz + 1e-7).all()
        t += dt
    assert seen == {0, 1, 2}
    # instance order does not matter (bit for bit)
    ctl2 = make_controller(B, dt, th, cfg2["zcom"], warm_start=1)
    _walk_setup(ctl2, plan, xs[::-1].copy())
    st2 = ctl2.new_state(cfg2["q0"], np.zeros(30), t=0.0)
    out2, _, _ = ctl2.rollout(st2, nt)
    torch.cuda.synchronize()
    assert np.array_equal(out2.cpu().numpy()[::-1], out.cpu().numpy())
    for i in range(0, B, 512):
        o = Oracle(sim_time=1.0, dt=dt, horizon_time=th, do_ik=True)
        o.set_zcom(cfg2["zcom"])
        o.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        o.set_segments(plan["segs"], plan["seg_of_sample"], xscale=float(xs[i]))
        r = o.rollout(np.concatenate([cfg2["q0"], np.zeros(30)]), 0.0, nt, log=True)
        assert close(stn[i, :60], r["state"], 1e-7)
        for tk in range(0, nt, 5):
            ref = r["log"][tk]
            assert close(log[tk, i, :24], ref[:24], TOL_REL), (i, tk)
            assert close(log[tk, i, 24:], ref[24:], TOL_REL, scale=WEIGHT), (i, tk)


def test_randomised_walking_config4_ingredients(cfg2):
    """BASELINE configs[3] on one GPU's share at small scale: per-link mass x U(0.9,1.1) and CoM +- 5 mm
    (seed 20260004 + i), per-instance start posture from the IK KERNEL on the randomised model, per-instance
    LIPM height = that posture's CoM height, then walking.  Each instance against an oracle built from the
    same raw link table (which runs its own IK)."""
    from linearmpchumanoid_amd import trajectories
    from linearmpchumanoid_amd.controller import nominal_links, initial_configuration
    from oracle.pyoracle import Oracle
    dt, N, B, nt = 1e-3, 32, 6, 160
    th = N * dt
    raw = np.tile(nominal_links(), (B, 1, 1))
    for i in range(B):
        rng = np.random.default_rng(20260004 + i)
        raw[i, :, 0] *= rng.uniform(0.9, 1.1, 28)
        raw[i, :, 1:4] += rng.uniform(-5e-3, 5e-3, (28, 3)) * (raw[i, :, 0:1] > 0)
    plan = trajectories.walk_plan(1.0, dt, num_steps=2, time_per_step=0.2, ds_time=0.05, step_height=0.02, settle_time=0.05)
    xs = np.linspace(0.02, 0.05, B)
    ctl = make_controller(B, dt, th, cfg2["zcom"], warm_start=1)
    ctl.set_model(raw)
    q = torch.as_tensor(np.tile(initial_configuration(), (B, 1))).to(ctl.device)
    q, iters = ctl.ik(q)                                           # Kinematics::compute per instance, randomised model
    com = torch.zeros((B, 3), dtype=torch.float64, device=ctl.device)
    from linearmpchumanoid_amd import capi
    import ctypes as C
    capi.check(capi.lib().lmh_robot_com(ctl._h, C.c_void_p(q.data_ptr()), C.c_void_p(com.data_ptr()), ctl._stream()))
    torch.cuda.synchronize()
    q0s, zc = q.cpu().numpy(), com.cpu().numpy()[:, 2].copy()
    assert (iters.cpu().numpy() <= 6).all()
    assert np.abs(com.cpu().numpy() - np.array([-0.02, 0.0, 0.26])).max() < 1e-9    # the IK target
    ctl.set_zcom(zc)
    _walk_setup(ctl, plan, xs)
    st = ctl.new_state(q0s, np.zeros(30), t=0.0)
    out, status, log = ctl.rollout(st, nt, log=True)
    torch.cuda.synchronize()
    stn, log, status = st.cpu().numpy(), log.cpu().numpy(), status.cpu().numpy()
    assert (status[:, 2] == 0).all()
    for i in range(B):
        o = Oracle(sim_time=1.0, dt=dt, horizon_time=th, do_ik=True, raw_links=raw[i])
        assert np.abs(o.robot()["q"] - q0s[i]).max() < 1e-10       # same IK posture
        assert abs(o.zcom - zc[i]) < 1e-12
        o.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        o.set_segments(plan["segs"], plan["seg_of_sample"], xscale=float(xs[i]))
        r = o.rollout(np.concatenate([o.robot()["q"], np.zeros(30)]), 0.0, nt, log=True)
        assert status[i, 0] == r["k"][-1]
        for tk in range(0, nt, 3):
            ref = r["log"][tk]
            assert close(log[tk, i, :24], ref[:24], TOL_REL), (i, tk)
            assert close(log[tk, i, 24:], ref[24:], TOL_REL, scale=WEIGHT), (i, tk)


def test_arbitrary_warm_start_sets_reach_the_same_minimiser(cfg2):
    """The warm start is only a starting point: whatever active set the status word carries (a foot with a single
    free vertex -> singular K_f, so the helper wave's K^-1 is refused and the general free-set solve with the lazily
    formed cone Hessian runs; nothing free; everything free; random sets), the evaluation must land on the unique
    minimiser, i.e. on the cold-start result and on the oracle."""
    B = 64
    v = perturbed_velocities(B, seed=909) * 1.5
    ctl_cold = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
    ctl_cold.set_refs_stance(2.0, 2)
    st = ctl_cold.new_state(cfg2["q0"], v, t=0.0)
    ref, sref = ctl_cold.stand_step(st)
    torch.cuda.synchronize()
    ref = ref.cpu().numpy()
    rng = np.random.default_rng(5)
    masks = [0x0000FFF0, 0xFFFFFFFF, 0x00000000, 0xFFF0FFF0, 0x0FFF0001] + [int(x) for x in rng.integers(0, 2 ** 32, 6, dtype=np.uint64)]
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=1)
    ctl.set_refs_stance(2.0, 2)
    for m in masks:                                               # status[3] holds the ACTIVE mask (1 = coefficient at its bound)
        status = ctl.new_status()
        status[:, 3] = int(np.array(m, dtype=np.uint32).astype(np.int32))
        for debug in (False, True):                               # two-wave plain kernel and single-wave debug kernel
            st = ctl.new_state(cfg2["q0"], v, t=0.0)
            res = ctl.stand_step(st, status=status.clone(), debug=debug)
            torch.cuda.synchronize()
            out, sts = res[0].cpu().numpy(), res[1].cpu().numpy()
            assert (sts[:, 2] == 0).all(), hex(m)
            assert rel_err(out[:, :36], ref[:, :36]) < 1e-7, (hex(m), debug, rel_err(out[:, :36], ref[:, :36]))
    for i in range(0, B, 16):
        o = oracle_system(cfg2["dt"], cfg2["th"])
        e = o.eval(cfg2["q0"], v[i], 0.0)
        assert rel_err(ref[i, :24], e["tau"]) < TOL_REL and rel_err(ref[i, 24:36], e["f"]) < TOL_REL


def test_cone_qp_kkt_at_scale(cfg2):
    """Optimality of the contact-force QP, independent of the oracle: for 1024 strongly perturbed states in each
    support phase the kernel's coefficients c must satisfy the KKT conditions of
    min 1/2 c'Pc - q'c, c >= 0 (forced coefficients 0) with P = G'WG + eps I and q = G'h taken from the debug record:
    c >= 0, (Pc - q)_j = 0 where c_j > 0, (Pc - q)_j >= 0 where c_j = 0.  P is strictly convex, so KKT <=> the
    unique minimiser.  This covers the push-through route, the general free-set route and the fall-backs."""
    from linearmpchumanoid_amd.controller import unpack_debug
    B = 1024
    v = perturbed_velocities(B, seed=777) * 2.0                   # 0.6 m/s pushes: most robots have active bounds
    worst = 0.0
    n_active = 0
    for ph in (0, 1, 2):
        ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
        n = 2500
        ctl.set_refs(np.zeros(n), np.zeros(n), np.full(n, ph, dtype=np.uint8))
        st = ctl.new_state(cfg2["q0"], v, t=0.0)
        out, status, dbg = ctl.stand_step(st, debug=True)
        torch.cuda.synchronize()
        dbg, status = dbg.cpu().numpy(), status.cpu().numpy()
        assert (status[:, 2] == 0).all()
        forced = np.zeros(32, dtype=bool)
        if ph == 2: forced[:16] = True                            # left support: the right foot carries nothing
        if ph == 1: forced[16:] = True
        for i in range(B):
            d = unpack_debug(dbg[i])
            Pm, q, c = d["P"], d["qv"], d["c"]
            lam = Pm @ c - q
            scale = 1.0 + np.abs(q).max()
            assert c.min() >= -1e-9 * (1.0 + c.max()), (ph, i, c.min())
            assert np.abs(c[forced]).max(initial=0.0) == 0.0
            free = (c > 1e-9 * (1.0 + c.max())) & ~forced
            at0 = ~free & ~forced
            assert np.abs(lam[free]).max(initial=0.0) < 1e-9 * scale, (ph, i)
            assert lam[at0].min(initial=0.0) > -1e-9 * scale, (ph, i, lam[at0].min(initial=0.0))
            worst = max(worst, np.abs(lam[free]).max(initial=0.0) / scale)
            n_active += int(at0.sum() > 0)
    assert n_active > B                                            # the bounds really are active in a large share of the cases


def test_two_wave_schedule_is_deterministic_and_equals_the_single_wave_schedule(cfg2):
    """The plain evaluation kernel and the rollout run two cooperating waves per robot (joined at workgroup
    barriers); the debug kernel runs the same arithmetic on one wave.  Every LDS entry is produced by the same
    expression in both schedules, so the results must agree BIT FOR BIT, and repeated launches must be identical
    (a missing barrier would show up as a difference or as run-to-run noise)."""
    B = 512
    v = perturbed_velocities(B, seed=404)
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
    ctl.set_refs_stance(2.0, 2)
    st1 = ctl.new_state(cfg2["q0"], v, t=0.0)
    st2 = st1.clone()
    o1, s1 = ctl.stand_step(st1)                                  # two waves
    o2, s2, _ = ctl.stand_step(st2, debug=True)                   # one wave
    torch.cuda.synchronize()
    assert torch.equal(o1[:, :78], o2[:, :78]) and torch.equal(s1, s2) and torch.equal(st1, st2)
    runs = []
    for _ in range(3):
        st = ctl.new_state(cfg2["q0"], v, t=0.0)
        out, status, log = ctl.rollout(st, 8, log=True)
        torch.cuda.synchronize()
        runs.append((st.clone(), out.clone(), status.clone(), log.clone()))
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(r, runs[0]))
    # rollout (two waves, state in registers) == 4 x 8 chained plain evaluations is covered against the oracle;
    # here: first tick's first stage equals the plain evaluation bit for bit through the log-free path
    assert (runs[0][2][:, 2] == 0).all()


def test_mixed_precision_mode(cfg2):
    """LMH_PRECISION_MIXED (BASELINE config 5's sweep): model terms in fp32 arithmetic, references + QP in fp64.
    The preview index k stays bit-exact; tau / f land within 1e-4 of the fp64 oracle (they do NOT meet the 1e-6
    bar: scripts/precision_sweep.py reports the distribution), flags stay clear, flight forces stay exactly zero."""
    from linearmpchumanoid_amd import capi
    B = 16
    v = perturbed_velocities(B, seed=23)
    outs = {}
    for prec in (capi.PRECISION_FP64, capi.PRECISION_MIXED):
        ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=1, precision=prec)
        ctl.set_refs_stance(2.0, 2)
        st = ctl.new_state(cfg2["q0"], v, t=0.0)
        out, status, log = ctl.rollout(st, 12, log=True)
        ev, evs = ctl.stand_step(st)                              # the plain evaluation kernel honours the mode too
        torch.cuda.synchronize()
        outs[prec] = (log.cpu().numpy(), status.cpu().numpy(), ev.cpu().numpy(), evs.cpu().numpy())
    l64, s64, e64, es64 = outs[capi.PRECISION_FP64]
    lmx, smx, emx, esmx = outs[capi.PRECISION_MIXED]
    assert np.array_equal(s64[:, 0], smx[:, 0]) and np.array_equal(es64[:, 0], esmx[:, 0])     # k bit-exact in every mode
    assert (smx[:, 2] == 0).all() and (esmx[:, 2] == 0).all()
    for i in range(0, B, 4):
        o = oracle_system(cfg2["dt"], cfg2["th"])
        r = o.rollout(np.concatenate([cfg2["q0"], v[i]]), 0.0, 12, log=True)
        for tk in range(12):
            assert rel_err(l64[tk, i, :24], r["log"][tk][:24]) < TOL_REL
            e_t, e_f = rel_err(lmx[tk, i, :24], r["log"][tk][:24]), rel_err(lmx[tk, i, 24:], r["log"][tk][24:])
            assert 1e-9 < e_t < 1e-4 and e_f < 1e-4, (i, tk, e_t, e_f)     # fp32-sized, not fp64-sized, errors
    assert rel_err(emx[:, :36], e64[:, :36]) < 1e-4
    with pytest.raises(capi.LmhError):
        make_controller(1, cfg2["dt"], cfg2["th"], cfg2["zcom"], precision=7)
    ctl = make_controller(4, cfg2["dt"], cfg2["th"], cfg2["zcom"], precision=capi.PRECISION_MIXED)
    ctl.set_refs(np.zeros(2500), np.zeros(2500), np.full(2500, 3, dtype=np.uint8))
    st = ctl.new_state(cfg2["q0"], v[:4], t=0.0)
    out, _, _ = ctl.rollout(st, 3)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy()[:, 24:36]).max() == 0.0      # flight: exactly no contact force in mixed mode too


# ------------------------------------------------------------------------------- edge cases
@pytest.mark.parametrize("N", [48, 64])
def test_long_horizons(cfg2, N):
    """BASELINE config 5 uses N = 48; 64 is the ABI maximum (LMH_MAX_HORIZON)."""
    from oracle.pyoracle import Oracle
    dt = 1e-3
    th = N * dt
    B = 3
    v = perturbed_velocities(B, seed=91) * 0.5
    ctl = make_controller(B, dt, th, cfg2["zcom"], warm_start=0)
    assert ctl.N == N
    ctl.set_refs_stance(1.0, 2)
    st = ctl.new_state(cfg2["q0"], v, t=0.0123)
    out, status = ctl.stand_step(st)
    torch.cuda.synchronize()
    out, status = out.cpu().numpy(), status.cpu().numpy()
    for i in range(B):
        o = Oracle(sim_time=1.0, dt=dt, horizon_time=th, do_ik=True)
        o.set_zcom(cfg2["zcom"])
        assert o.horizon == N
        assert rel_err(ctl.mpc_gain(), o.gain_row()) < 1e-10
        e = o.eval(cfg2["q0"], v[i], 0.0123)
        assert status[i, 0] == e["k"] and status[i, 2] == 0
        assert rel_err(out[i, :24], e["tau"]) < TOL_REL and rel_err(out[i, 24:36], e["f"]) < TOL_REL


def test_horizon_out_of_range_is_rejected(cfg2):
    from linearmpchumanoid_amd.capi import LmhError
    with pytest.raises(LmhError):
        make_controller(1, 1e-3, 0.065, cfg2["zcom"])       # N = 65 > LMH_MAX_HORIZON
    with pytest.raises(LmhError):
        make_controller(1, 1e-3, 0.0001, cfg2["zcom"])      # N = 0


def test_ragged_batch_sizes_and_single_instance(cfg2):
    """B = 1 and a B that is no multiple of anything: per-instance results do not depend on the batch."""
    v = perturbed_velocities(7, seed=12)
    ref = None
    for B in (7, 1):
        ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
        ctl.set_refs_stance(2.0, 2)
        st = ctl.new_state(cfg2["q0"], v[:B], t=0.0)
        out, status, _ = ctl.rollout(st, 3)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        if ref is None:
            ref = o
        else:
            assert np.array_equal(o[0], ref[0])


def test_preview_window_leaving_the_reference_arrays_is_flagged(cfg2):
    """The reference would read past the end of the ZMP vectors (mpcLinearPendulum.cpp:93-94); here the
    window is clamped and LMH_FLAG_ZMP_RANGE is raised."""
    from linearmpchumanoid_amd import capi
    ctl = make_controller(2, cfg2["dt"], cfg2["th"], cfg2["zcom"])
    ctl.set_refs_stance(0.1, 2)                              # (0.1 + 0.5)/1e-3 = 600 samples
    st = ctl.new_state(cfg2["q0"], np.zeros(30), t=0.590)    # k = 590, k + 16 > 599
    out, status = ctl.stand_step(st)
    torch.cuda.synchronize()
    status = status.cpu().numpy()
    assert (status[:, 2] & capi.FLAG_ZMP_RANGE).all()
    st2 = ctl.new_state(cfg2["q0"], np.zeros(30), t=0.5)
    _, status2 = ctl.stand_step(st2)
    torch.cuda.synchronize()
    assert (status2.cpu().numpy()[:, 2] == 0).all()


def test_non_finite_state_is_flagged_not_propagated_silently(cfg2):
    """The reference aborts on NaN/Inf (controller.cpp:448-466); the batched path flags the instance."""
    from linearmpchumanoid_amd import capi
    ctl = make_controller(3, cfg2["dt"], cfg2["th"], cfg2["zcom"])
    ctl.set_refs_stance(2.0, 2)
    v = np.zeros((3, 30)); v[1, 7] = np.nan
    st = ctl.new_state(cfg2["q0"], v, t=0.0)
    out, status = ctl.stand_step(st)
    torch.cuda.synchronize()
    status, out = status.cpu().numpy(), out.cpu().numpy()
    assert status[1, 2] & capi.FLAG_NONFINITE
    assert status[0, 2] == 0 and status[2, 2] == 0 and np.isfinite(out[0, :66]).all() and np.array_equal(out[0], out[2])


def test_per_instance_lipm_height(cfg2):
    """Domain randomisation of z_com -> per-instance gain rows."""
    from oracle.pyoracle import Oracle
    zs = np.array([0.24, 0.26, 0.275])
    v = perturbed_velocities(3, seed=5) * 0.3
    ctl = make_controller(3, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
    ctl.set_refs_stance(2.0, 2)
    ctl.set_zcom(zs)
    st = ctl.new_state(cfg2["q0"], v, t=0.0)
    out, status = ctl.stand_step(st)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i in range(3):
        o = Oracle(sim_time=2.0, dt=cfg2["dt"], horizon_time=cfg2["th"], do_ik=True)
        o.set_zcom(float(zs[i]))
        e = o.eval(cfg2["q0"], v[i], 0.0)
        assert rel_err(out[i, :24], e["tau"]) < TOL_REL and rel_err(out[i, 24:36], e["f"]) < TOL_REL
        assert np.abs(out[i, 72:75] - o.qp()["mpcRef"][:3]).max() < 1e-9      # Mpc3dLip::getXRef
