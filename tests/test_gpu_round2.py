"""GPU parity tests added in round 2 (VERDICT r01 "close the parity holes"): the benched regimes, non-default gains / weights
(qp_setup<18>), the Lawson-Hanson fall-back, the QP status flags, comVel / yRef, configs 4 and 5 at one GPU's share, the
record files and the N > 1 entry path of bench.py.  Same rules as test_gpu_parity.py: HIP path through the C ABI against the CPU
oracle; 1e-6 relative on tau / f, bit-exact k.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import TOL_REL, WEIGHT, close, close_on, oracle_system, perturbed_velocities, rel_err, vec_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_controller(B, dt, th, zcom, **kw):
    from linearmpchumanoid_amd.controller import BatchedController, default_config
    return BatchedController(B, default_config(dt=dt, time_horizon=th, z_com=zcom, **kw))


@pytest.fixture(scope="module")
def cfg2():
    o = oracle_system(1e-3, 0.016)
    return dict(dt=1e-3, th=0.016, zcom=o.zcom, q0=o.robot()["q"].copy())


# ------------------------------------------------------------------------------- (a) the regime bench.py --config 2 times
def test_config2_benched_regime_against_oracle(cfg2):
    """B = 1024, 250 ticks, warm start, velocity pushes (exactly `bench.py --config 2`'s rollouts: ticks 30..230 are what it
    times): no flag anywhere, every 64th robot against its own oracle rollout at ticks 100 / 200 / 250 (tau, f, k) and in the
    final state."""
    B, nt = 1024, 250
    v = perturbed_velocities(B)
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=1)
    ctl.set_refs_stance(nt * cfg2["dt"] + 1.0, 2)
    st = ctl.new_state(cfg2["q0"], v, t=0.0)
    out, status, log = ctl.rollout(st, nt, log=True)
    torch.cuda.synchronize()
    stn, log, status = st.cpu().numpy(), log.cpu().numpy(), status.cpu().numpy()
    assert (status[:, 2] == 0).all()
    n_active = 0
    for i in range(0, B, 64):
        o = oracle_system(cfg2["dt"], cfg2["th"], sim_time=nt * cfg2["dt"] + 1.0)
        r = o.rollout(np.concatenate([cfg2["q0"], v[i]]), 0.0, nt, log=True)
        assert status[i, 0] == r["k"][-1]
        for tk in (99, 199, 249):
            assert close(log[tk, i, :24], r["log"][tk][:24]) and close(log[tk, i, 24:], r["log"][tk][24:], scale=WEIGHT), (i, tk)
        assert close(stn[i, :60], r["state"], 1e-7)
        n_active += int(status[i, 3] != 0)
    assert n_active >= 4                                          # the sampled robots include ones with active friction bounds


# ------------------------------------------------------------------------------- (b) gains / weights / mu
@pytest.mark.parametrize("over", [
    dict(w_com_ang=50.0),                                          # qp_setup<18>: the angular-momentum rows carry weight
    dict(mu=0.4),
    dict(w_com_ang=20.0, mu=0.5, kp_joints=250.0, kd_joints=30.0, kp_mom=12.0, kd_mom=7.0, kp_feet=450.0, kd_feet=40.0,
         w_com_lin=3000.0, w_base_pos=8.0, w_base_ang=12.0, w_joints=2.0, w_force=1.5, w_foot=50000.0, eps_coeff=2e-8),
])
def test_non_default_gains_and_weights(cfg2, over):
    """Every gain / weight field of lmh_config against an oracle carrying the same values (controller.hpp:81,102-124), in
    double support, single support and over a short rollout."""
    B = 12
    v = perturbed_velocities(B, seed=321) * 1.5
    vprev = perturbed_velocities(B, seed=322)
    for ph in (0, 1):
        ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0, **over)
        n = 2500
        ctl.set_refs(np.zeros(n), np.zeros(n), np.full(n, ph, dtype=np.uint8))
        st = ctl.new_state(cfg2["q0"], v, t=0.0, v_prev=vprev)
        out, status = ctl.stand_step(st)
        o2, s2, _ = ctl.stand_step(ctl.new_state(cfg2["q0"], v, t=0.0, v_prev=vprev), debug=True)     # single-wave schedule
        torch.cuda.synchronize()
        out, status = out.cpu().numpy(), status.cpu().numpy()
        assert np.array_equal(out[:, :78], o2.cpu().numpy()[:, :78])
        assert (status[:, 2] == 0).all()
        for i in range(B):
            o = oracle_system(cfg2["dt"], cfg2["th"])
            o.set_gains(**over)
            zx, zy = o.zmp()
            o.set_refs(zx, zy, np.full(len(zx), ph, dtype=np.uint8))
            o.set_prev_velocity(vprev[i])
            e = o.eval(cfg2["q0"], v[i], 0.0)
            assert close(out[i, :24], e["tau"]) and close(out[i, 24:36], e["f"], scale=WEIGHT) and close(out[i, 36:66], e["qpp"]), (ph, i)
            if "mu" in over:
                fx, fy, fz = out[i, 24 + 3:24 + 6]
                assert abs(fx) <= over["mu"] * fz + 1e-7 and abs(fy) <= over["mu"] * fz + 1e-7
    ctl = make_controller(4, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=1, **over)
    ctl.set_refs_stance(2.0, 2)
    st = ctl.new_state(cfg2["q0"], v[:4], t=0.0)
    out, status, log = ctl.rollout(st, 25, log=True)
    torch.cuda.synchronize()
    log = log.cpu().numpy()
    assert (status.cpu().numpy()[:, 2] == 0).all()
    for i in range(4):
        o = oracle_system(cfg2["dt"], cfg2["th"])
        o.set_gains(**over)
        r = o.rollout(np.concatenate([cfg2["q0"], v[i]]), 0.0, 25, log=True)
        for tk in range(0, 25, 4):
            assert close(log[tk, i, :24], r["log"][tk][:24]) and close(log[tk, i, 24:], r["log"][tk][24:], scale=WEIGHT), (i, tk)


# ------------------------------------------------------------------------------- (c) Lawson-Hanson fall-back
def test_lawson_hanson_fallback_reaches_the_minimiser(cfg2):
    """lmh_config.bpp_rounds < 0 skips block pivoting: every cone solve runs the Lawson-Hanson pass from the empty set with the
    general (lazily formed G'WG + eps I) free-set factorisation.  Strong pushes, all three support phases: same minimiser as
    the default route, as the oracle, and KKT of the cone problem from the debug record.  bpp_rounds = 1 exercises the hand-over
    from block pivoting to Lawson-Hanson."""
    from linearmpchumanoid_amd.controller import unpack_debug
    B = 96
    v = perturbed_velocities(B, seed=777) * 2.0
    for ph in (0, 1, 2):
        n = 2500
        res = {}
        for rounds in (0, -1, 1):
            ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0, bpp_rounds=rounds, max_qp_iters=200)
            ctl.set_refs(np.zeros(n), np.zeros(n), np.full(n, ph, dtype=np.uint8))
            st = ctl.new_state(cfg2["q0"], v, t=0.0)
            out, status, dbg = ctl.stand_step(st, debug=True)
            out2, status2 = ctl.stand_step(ctl.new_state(cfg2["q0"], v, t=0.0))                  # two-wave kernel
            torch.cuda.synchronize()
            assert torch.equal(out[:, :78], out2[:, :78]) and torch.equal(status, status2)
            res[rounds] = (out.cpu().numpy(), status.cpu().numpy(), dbg.cpu().numpy())
            assert (res[rounds][1][:, 2] == 0).all(), (ph, rounds)
        ref = res[0][0]
        for rounds in (-1, 1):
            assert rel_err(res[rounds][0][:, :36], ref[:, :36]) < 1e-7, (ph, rounds)
        assert res[-1][1][:, 1].max() > 12                       # Lawson-Hanson adds one coefficient per solve: many more solves than block pivoting
        assert (res[-1][1][:, 1] >= res[0][1][:, 1]).all()
        forced = np.zeros(32, dtype=bool)
        if ph == 2: forced[:16] = True
        if ph == 1: forced[16:] = True
        for i in range(B):
            d = unpack_debug(res[-1][2][i])
            Pm, q, c = d["P"], d["qv"], d["c"]
            lam = Pm @ c - q
            scale = 1.0 + np.abs(q).max()
            assert c.min() >= 0.0 and np.abs(c[forced]).max(initial=0.0) == 0.0
            free = (c > 0) & ~forced
            assert np.abs(lam[free]).max(initial=0.0) < 1e-9 * scale and lam[~free & ~forced].min(initial=0.0) > -1e-9 * scale, (ph, i)
        for i in range(0, B, 12):
            o = oracle_system(cfg2["dt"], cfg2["th"])
            zx, zy = o.zmp()
            o.set_refs(zx, zy, np.full(len(zx), ph, dtype=np.uint8))
            e = o.eval(cfg2["q0"], v[i], 0.0)
            assert close(res[-1][0][i, :24], e["tau"]) and close(res[-1][0][i, 24:36], e["f"]), (ph, i)


# ------------------------------------------------------------------------------- (d) status flags
def test_qp_iteration_cap_flags_only_the_robots_that_hit_it(cfg2):
    """max_qp_iters = 1: robots whose cold-start cone solve needs a second round report LMH_FLAG_QP_MAXITER (the reference only
    logs "QP failed", controller.cpp:472-476); robots that finish in one round are bit-identical to the uncapped run."""
    from linearmpchumanoid_amd import capi
    B = 256
    v = perturbed_velocities(B, seed=777) * 2.0
    v[::2] *= 0.05                                                  # every second robot barely pushed: no active bound
    res = {}
    for cap in (64, 1):
        ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0, max_qp_iters=cap)
        ctl.set_refs_stance(2.0, 2)
        st = ctl.new_state(cfg2["q0"], v, t=0.0)
        out, status = ctl.stand_step(st)
        torch.cuda.synchronize()
        res[cap] = (out.cpu().numpy(), status.cpu().numpy())
    (o64, s64), (o1, s1) = res[64], res[1]
    assert (s64[:, 2] == 0).all()
    need_more = s64[:, 1] > 1
    assert need_more.any() and (~need_more).any()
    assert ((s1[:, 2] & capi.FLAG_QP_MAXITER) != 0).tolist() == need_more.tolist()
    assert (s1[need_more, 1] == 1).all()
    assert np.array_equal(o1[~need_more], o64[~need_more]) and np.array_equal(s1[~need_more], s64[~need_more])


def test_massless_model_raises_not_spd_for_that_robot_only(cfg2):
    """The QP's factorisations (Woodbury core, Schur complement S = Mb H^-1 Mb', W) are positive definite for ANY model as long as
    the floating-base rows Mb have full rank; a robot whose link table carries no mass at all has Mb = 0, so S pivots at 0 ->
    LMH_FLAG_NOT_SPD (the quotients that follow are not finite: LMH_FLAG_NONFINITE comes with it).  Its neighbours in the batch
    are bit-identical to a clean batch."""
    from linearmpchumanoid_amd import capi
    from linearmpchumanoid_amd.controller import nominal_links
    B, bad = 9, 4
    v = perturbed_velocities(B, seed=17) * 0.3
    raw = np.tile(nominal_links(), (B, 1, 1))
    outs = {}
    for poison in (False, True):
        r = raw.copy()
        if poison:
            r[bad, :, 0] = 0.0                                      # every link mass 0 (inertias kept)
            r[bad, :, 4:] = 0.0
        ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
        ctl.set_refs_stance(2.0, 2)
        ctl.set_model(r)
        st = ctl.new_state(cfg2["q0"], v, t=0.0)
        out, status, _ = ctl.rollout(st, 2)
        torch.cuda.synchronize()
        outs[poison] = (out.cpu().numpy(), status.cpu().numpy(), st.cpu().numpy())
    (oc, sc, stc), (op, sp, stp) = outs[False], outs[True]
    assert (sc[:, 2] == 0).all()
    assert sp[bad, 2] & capi.FLAG_NOT_SPD
    keep = np.arange(B) != bad
    assert (sp[keep, 2] == 0).all()
    assert np.array_equal(op[keep], oc[keep]) and np.array_equal(stp[keep], stc[keep])


# ------------------------------------------------------------------------------- (e) comVel, yRef, angular momentum
def test_com_velocity_momentum_and_mpc_references_directly(cfg2):
    """SURVEY row R4 (Robot::updateVelocityState -> comVel, angular momentum) and both Mpc3dLip reference triples, asserted
    directly against the oracle's Robot / Mpc3dLip state (out[66:78] and the debug record)."""
    from linearmpchumanoid_amd.controller import unpack_debug
    B = 16
    v = perturbed_velocities(B, seed=2024)
    vprev = perturbed_velocities(B, seed=2025)
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0)
    n = 2500
    zx = 0.01 * np.sin(np.arange(n) * 0.01); zy = 0.02 * np.cos(np.arange(n) * 0.02)      # non-trivial preview window on both axes
    ctl.set_refs(zx, zy)
    st = ctl.new_state(cfg2["q0"], v, t=0.1234, v_prev=vprev)
    out, status, dbg = ctl.stand_step(st, debug=True)
    torch.cuda.synchronize()
    out, dbg, status = out.cpu().numpy(), dbg.cpu().numpy(), status.cpu().numpy()
    for i in range(B):
        o = oracle_system(cfg2["dt"], cfg2["th"])
        o.set_refs(zx, zy)
        o.set_prev_velocity(vprev[i])
        e = o.eval(cfg2["q0"], v[i], 0.1234)
        rb, qp = o.robot(), o.qp()
        d = unpack_debug(dbg[i])
        assert status[i, 0] == e["k"] == 123
        assert np.abs(out[i, 66:69] - rb["CoM"]).max() < 1e-12
        assert close(out[i, 69:72], rb["comVel"], 1e-11)
        assert close(d["comVel"], rb["comVel"], 1e-11)
        assert close(d["angMom"], rb["angMom"], 1e-11)
        assert close_on(out[i, 72:75], qp["mpcRef"][:3], 1e-9, np.abs(qp["mpcRef"]).max())     # getXRef
        assert close_on(out[i, 75:78], qp["mpcRef"][3:], 1e-9, np.abs(qp["mpcRef"]).max())     # getYRef (y ~ 0: on the scale of the x record)
        assert close_on(d["mpc"][:2], qp["u0"], 1e-10, 9.81 / 0.26 * 0.05 + np.abs(qp["u0"]).max())


# ------------------------------------------------------------------------------- (f) config 4 at one GPU's share
def test_config4_one_gpu_share_randomised_ik_walking():
    """BASELINE configs[3], one GPU's share (what `bench.py --config 4` runs per rank): 4096 robots with randomised link
    masses / CoMs (seed 20260004 + i), start posture from the IK KERNEL on each robot's own model, per-instance LIPM height and
    step length, walking through DS -> SS.  Properties for all robots (IK hits its target, no flags, swing foot force exactly
    0, friction cones, bit-exact k); every 512th robot against an oracle built from the same raw table (own IK, own rollout)."""
    sys.path.insert(0, ROOT)
    import bench
    from oracle.pyoracle import Oracle
    args = bench.parse(["--config", "4", "--coupled"])            # mpc_dt = dt, 0.2 s steps (the decoupled form: test_gpu_round3)
    B, nt = args.instances, 480
    ctl = make_controller(B, args.dt, args.horizon * args.dt, 0.26, warm_start=1)
    state, host = bench.build_workload(args, ctl, 0, B, nt)
    assert np.abs(host["zcom"] - 0.26).max() < 1e-9               # the IK target CoM height, per instance
    st = state.clone()
    out, status, log = ctl.rollout(st, nt, log=True)
    torch.cuda.synchronize()
    stn, log, status = st.cpu().numpy(), log.cpu().numpy(), status.cpu().numpy()
    assert (status[:, 2] == 0).all(), (np.unique(status[:, 2]), np.nonzero(status[:, 2])[0][:8], status[status[:, 2] != 0][:4])
    t, seen = 0.0, set()
    for tk in range(nt):
        k = int((t + args.dt) / args.dt)
        ph = int(host["phase"][k]); seen.add(ph)
        f = log[tk, :, 24:36]
        if ph == 1: assert np.abs(f[:, 6:]).max() == 0.0
        if ph == 2: assert np.abs(f[:, :6]).max() == 0.0
        for ft in range(2):
            fx, fy, fz = f[:, 6 * ft + 3], f[:, 6 * ft + 4], f[:, 6 * ft + 5]
            assert (fz > -1e-7).all() and (np.abs(fx) <= 0.7 * fz + 1e-7).all() and (np.abs(fy) <= 0.7 * fz + 1e-7).all()
        t += args.dt
    assert (status[:, 0] == k).all() and {0, 1} <= seen
    for i in range(0, B, 512):
        o = Oracle(sim_time=nt * args.dt + 0.5, dt=args.dt, horizon_time=args.horizon * args.dt, do_ik=True, raw_links=host["raw"][i])
        assert np.abs(o.robot()["q"] - host["q0"][i]).max() < 1e-10 and abs(o.zcom - host["zcom"][i]) < 1e-12
        o.set_refs(host["zmp_x"], host["zmp_y"], host["phase"])
        o.set_segments(host["segs"], host["sos"], xscale=float(host["xscale"][i]))
        r = o.rollout(np.concatenate([o.robot()["q"], np.zeros(30)]), 0.0, nt, log=True)
        assert close(stn[i, :60], r["state"], 1e-7)
        for tk in range(0, nt, 6):
            assert close(log[tk, i, :24], r["log"][tk][:24]) and close(log[tk, i, 24:], r["log"][tk][24:], scale=WEIGHT), (i, tk)


# ------------------------------------------------------------------------------- (g) config 5, real schedule
def test_config5_jump_schedule_full_size():
    """BASELINE configs[4], one GPU's share, SURVEY 8d's schedule: 0.4 s double support -> 0.15 s flight -> double support,
    N = 48, 4096 robots with small velocity perturbations, 600 ticks (through the landing).  For all robots: k bit-exact, no
    contact force at all during flight, friction cones otherwise, flags clear while the closed loop is in range; every 512th
    robot against its oracle rollout for as long as the oracle itself stays finite."""
    from linearmpchumanoid_amd import trajectories
    from oracle.pyoracle import Oracle
    dt, N, B, nt = 1e-3, 48, 4096, 600
    th = N * dt
    o0 = oracle_system(dt, th)
    q0, zcom = o0.robot()["q"].copy(), o0.zcom
    plan = trajectories.jump_plan(nt * dt + 0.5, dt, stance_time=0.4, flight_time=0.15)
    v = perturbed_velocities(B, seed=20260005) * 0.1
    ctl = make_controller(B, dt, th, zcom, warm_start=1)
    ctl.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
    st = ctl.new_state(q0, v, t=0.0)
    out, status, log = ctl.rollout(st, nt, log=True)
    torch.cuda.synchronize()
    stn, log, status = st.cpu().numpy(), log.cpu().numpy(), status.cpu().numpy()
    t, n_flight = 0.0, 0
    for tk in range(nt):
        k = int((t + dt) / dt)
        f = log[tk, :, 24:36]
        if plan["phase"][k] == 3:
            n_flight += 1
            assert np.abs(f).max() == 0.0
        elif tk < 400:
            for ft in range(2):
                fx, fy, fz = f[:, 6 * ft + 3], f[:, 6 * ft + 4], f[:, 6 * ft + 5]
                assert (fz > -1e-7).all() and (np.abs(fx) <= 0.7 * fz + 1e-7).all() and (np.abs(fy) <= 0.7 * fz + 1e-7).all()
        t += dt
    assert n_flight == 150 and (status[:, 0] == k).all()
    checked = 0
    for i in range(0, B, 512):
        o = Oracle(sim_time=nt * dt + 0.5, dt=dt, horizon_time=th, do_ik=True)
        o.set_zcom(zcom)
        o.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        r = o.rollout(np.concatenate([q0, v[i]]), 0.0, nt, log=True)
        finite = np.isfinite(r["log"]).all(axis=1) & (np.abs(r["log"]).max(axis=1) < 1e6)
        last = nt if finite.all() else int(np.argmin(finite))
        assert last >= 550                                          # stance and the whole flight phase are always comparable
        for tk in list(range(0, last, 10)) + [399, 400, 549, min(550, last - 1)]:
            ref = r["log"][tk]
            assert close(log[tk, i, :24], ref[:24]) and close(log[tk, i, 24:], ref[24:], scale=WEIGHT), (i, tk)
        if last == nt:
            assert status[i, 2] == 0 and close(stn[i, :60], r["state"], 1e-6)
        checked += 1
    assert checked == 8


# ------------------------------------------------------------------------------- device-side reference generators
@pytest.mark.parametrize("gait", [
    dict(simulation_time=2.5, num_steps=4, time_per_step=0.5, ds_time=0.1, step_height=0.02, settle_time=0.3, first_support=1, foot_y=0.05),
    dict(simulation_time=1.0, num_steps=2, time_per_step=0.2, ds_time=0.05, step_height=0.02, settle_time=0.1, first_support=1, foot_y=0.05),
    dict(simulation_time=1.3, num_steps=7, time_per_step=0.17, ds_time=0.033, step_height=0.015, settle_time=0.0731, first_support=2, foot_y=0.045),
    dict(simulation_time=0.56, num_steps=5, time_per_step=0.2, ds_time=0.05, step_height=0.02, settle_time=0.1, first_support=1, foot_y=0.05),  # the last swing is cut by the end of the sample grid
])
def test_walk_generator_kernel_matches_the_host_plan(gait):
    """lmh_gen_walk (device kernel) against trajectories.walk_plan (host statement of the same plan): ZMP samples, support phase and
    seg_of_sample BIT-exact, segment start times exact, stance coefficients exact, swing polynomials (closed form on the device,
    findPolyCoeff's linear solve on the host) equal in position / velocity / acceleration to 1e-12 over the swing."""
    from linearmpchumanoid_amd import trajectories
    dt = 1e-3
    ctl = make_controller(2, dt, 0.032, 0.26)
    ctl.gen_walk(**gait)
    g = ctl.get_refs()
    p = trajectories.walk_plan(gait["simulation_time"], dt, **{k: v for k, v in gait.items() if k != "simulation_time"})
    assert np.array_equal(g["phase"], p["phase"]) and np.array_equal(g["seg_of_sample"], p["seg_of_sample"])
    assert np.array_equal(g["zmp_x"], p["zmp_x"]) and np.array_equal(g["zmp_y"], p["zmp_y"])
    assert g["segs"].shape == p["segs"].shape
    assert np.array_equal(g["segs"][:, 0], p["segs"][:, 0])
    for sgd, sgh in zip(g["segs"], p["segs"]):
        T = max(gait["time_per_step"] - gait["ds_time"], dt)
        t = np.linspace(0.0, T, 41)
        for ft in range(2):
            for ax in range(3):
                cd, ch = sgd[1 + 24 * ft + 8 * ax: 9 + 24 * ft + 8 * ax], sgh[1 + 24 * ft + 8 * ax: 9 + 24 * ft + 8 * ax]
                if not np.abs(ch[1:]).max() > 1e-9:                # a standing foot: constants, exactly
                    assert cd[0] == ch[0] and np.abs(cd[1:]).max() == 0.0
                    continue
                for der in range(3):
                    pd = np.polynomial.polynomial.Polynomial(cd).deriv(der)(t) if der else np.polynomial.polynomial.polyval(t, cd)
                    phh = np.polynomial.polynomial.Polynomial(ch).deriv(der)(t) if der else np.polynomial.polynomial.polyval(t, ch)
                    assert close(pd, phh, 1e-12), (ft, ax, der)
                assert np.abs(cd - ch).max() < 1e-11 * np.abs(ch).max()


def test_jump_generator_kernel_and_rollout_on_generated_plans(cfg2):
    """lmh_gen_jump against trajectories.jump_plan (exact), and a rollout on the device-generated walking plan against the oracle fed
    with the plan read back from the device."""
    from linearmpchumanoid_amd import trajectories
    from oracle.pyoracle import Oracle
    dt, th = 1e-3, 0.032
    ctl = make_controller(3, dt, th, cfg2["zcom"], warm_start=1)
    ctl.gen_jump(0.9, stance_time=0.4, flight_time=0.15)
    g, p = ctl.get_refs(), trajectories.jump_plan(0.9, dt, stance_time=0.4, flight_time=0.15)
    assert np.array_equal(g["phase"], p["phase"]) and np.array_equal(g["zmp_x"], p["zmp_x"]) and np.array_equal(g["zmp_y"], p["zmp_y"])
    assert len(g["segs"]) == 0
    ctl.gen_walk(0.6, num_steps=2, time_per_step=0.2, ds_time=0.05, step_height=0.02, settle_time=0.05)
    xs = np.array([0.02, 0.035, 0.05])
    ctl.set_xscale(xs)
    plan = ctl.get_refs()
    nt = 300
    st = ctl.new_state(cfg2["q0"], np.zeros(30), t=0.0)
    out, status, log = ctl.rollout(st, nt, log=True)
    torch.cuda.synchronize()
    log, status = log.cpu().numpy(), status.cpu().numpy()
    assert (status[:, 2] == 0).all()
    for i in range(3):
        o = Oracle(sim_time=0.6, dt=dt, horizon_time=th, do_ik=True)
        o.set_zcom(cfg2["zcom"])
        o.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        o.set_segments(plan["segs"], plan["seg_of_sample"], xscale=float(xs[i]))
        r = o.rollout(np.concatenate([cfg2["q0"], np.zeros(30)]), 0.0, nt, log=True)
        assert status[i, 0] == r["k"][-1]
        for tk in range(0, nt, 9):
            assert close(log[tk, i, :24], r["log"][tk][:24]) and close(log[tk, i, 24:], r["log"][tk][24:], scale=WEIGHT), (i, tk)


# ------------------------------------------------------------------------------- plant
def test_plant_rollout_parity_and_physics(cfg2):
    """lmh_config.plant = 1: the RK4 derivative is the forward dynamics M qdd = S'tau + J'w_contact - C(q, v) driven by the WBC torques,
    with the spring-damper contact at the sole vertices.  (i) against the oracle's plant (literal dense M solve, Dynamics::computeC
    re-run at the current velocity) over 200 ticks of a standing robot with small velocity perturbations, two contact parameter sets;
    (ii) physics: the ground carries the weight once the contact has settled."""
    from oracle.pyoracle import Oracle
    dt, th = 1e-3, 0.032
    B, nt = 6, 200
    v = perturbed_velocities(B, seed=31337) * 0.2
    v[0] = 0.0
    for contact in (dict(contact_k=2.0e4, contact_d=3.0, contact_dt=3.0, contact_mu=0.7), dict(contact_k=8.0e3, contact_d=2.0, contact_dt=4.0, contact_mu=0.5)):
        ctl = make_controller(B, dt, th, cfg2["zcom"], warm_start=1, plant=1, **contact)
        ctl.set_refs_stance(2.0, 2)
        st = ctl.new_state(cfg2["q0"], v, t=0.0)
        out, status, log = ctl.rollout(st, nt, log=True)
        torch.cuda.synchronize()
        stn, log, status = st.cpu().numpy(), log.cpu().numpy(), status.cpu().numpy()
        assert (status[:, 2] == 0).all()
        for i in range(B):
            o = Oracle(sim_time=2.0, dt=dt, horizon_time=th, do_ik=True)
            o.set_zcom(cfg2["zcom"])
            o.set_plant(True, k=contact["contact_k"], d=contact["contact_d"], dt=contact["contact_dt"], mu=contact["contact_mu"])
            r = o.rollout(np.concatenate([cfg2["q0"], v[i]]), 0.0, nt, log=True)
            assert status[i, 0] == r["k"][-1]
            assert close(stn[i, :60], r["state"], 1e-6), (i, np.abs(stn[i, :60] - r["state"]).max())
            for tk in range(0, nt, 7):
                assert close(log[tk, i, :24], r["log"][tk][:24]) and close(log[tk, i, 24:], r["log"][tk][24:], scale=WEIGHT), (i, tk)
            if i == 0:                                             # unperturbed robot: after 0.2 s the springs carry m g
                w, vf = o.contact()
                assert abs(w[5] + w[11] - o.mass * 9.81) < 0.05 * o.mass * 9.81 and (vf[:, 2] >= 0).all()
    # free fall and momentum conservation: see test_gpu_round3.test_plant_free_fall_and_momentum (the plant's velocity products are
    # evaluated at the current velocity since round 3)


# ------------------------------------------------------------------------------- summary kernel, record files, N > 1 entry
def test_summary_kernel_matches_the_host_form(cfg2):
    from linearmpchumanoid_amd import sharding
    B = 300
    v = perturbed_velocities(B, seed=5)
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"])
    ctl.set_refs_stance(2.0, 2)
    st = ctl.new_state(cfg2["q0"], v, t=0.0)
    out, status, _ = ctl.rollout(st, 6)
    s_dev = sharding.make_summary(st, out, status, ctl)
    torch.cuda.synchronize()
    s_host = sharding.make_summary(st.cpu(), out.cpu(), status.cpu())
    assert torch.equal(s_dev.cpu(), s_host)
    assert (s_host[:, 14] == torch.tensor([bin(int(x) & 0xFFFFFFFF).count("1") for x in status.cpu()[:, 3]], dtype=torch.float64)).all()


def _run_bench(argv, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_through_the_gpus_flag(tmp_path):
    """`bench.py --gpus 2` (the command the driver runs for the scaling table) starts two ranks by itself; here both share
    card 0 and talk over gloo (a one-GPU box).  n_gpus, the gathered summary (written through lmh_write_summary, read back
    through lmh_read_summary) and the per-rank workloads (instance ids continue across ranks) are checked."""
    from linearmpchumanoid_amd.controller import BatchedController
    p = tmp_path / "run.lmhsum"
    res = _run_bench(["--gpus", "2", "--backend", "gloo", "--instances", "256", "--steps", "2", "--warmup", "1", "--ticks", "8",
                      "--summary-out", str(p)], {"LMH_BENCH_DEVICE": "0"})
    assert res["n_gpus"] == 2 and res["summary_rows_gathered"] == 512 and res["instances_flagged"] == 0
    assert res["config"]["baseline_config"] == 4 and res["scaling"] == "weak"
    s, dt = BatchedController.read_summary(p)
    assert s.shape == (512, 16) and dt == 1e-3
    assert np.allclose(s[:, 6], 24 * 1e-3) and (s[:, 13] == 0).all() and (s[:, 8] > 30).all()     # t, flags, sum f_z ~ m g
    one = _run_bench(["--config", "4", "--instances", "512", "--steps", "2", "--warmup", "1", "--ticks", "8", "--no-cpu-baseline",
                      "--summary-out", str(tmp_path / "one.lmhsum")])
    s1, _ = BatchedController.read_summary(tmp_path / "one.lmhsum")
    assert np.array_equal(s1, s)                                    # sharding does not change any robot's result


def test_bench_under_torchrun_as_the_driver_launches_it():
    """The driver's N > 1 form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P
    bench.py --gpus 2 ...` -- bench.py is then ONE rank per process (no self-launch), reads RANK / LOCAL_RANK / WORLD_SIZE, and rank 0
    prints the one JSON line.  Both ranks share card 0 over gloo here (a one-GPU box)."""
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["LMH_BENCH_DEVICE"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--instances", "128",
           "--steps", "2", "--warmup", "1", "--ticks", "8"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["summary_rows_gathered"] == 256 and res["instances_flagged"] == 0 and res["value"] > 0
    # a --gpus that disagrees with the launcher's world size is refused, not silently run
    bad = subprocess.run(cmd[:cmd.index("--gpus") + 1] + ["4"] + cmd[cmd.index("--gpus") + 2:], env=env, capture_output=True, text=True, timeout=900)
    assert bad.returncode != 0


def test_bench_default_line_shape():
    """The JSON contract on a reduced workload: config 3 keys, binding roofline first, HBM / MFMA objects beside it, CPU
    baseline on the same workload with the last tick compared against the GPU."""
    res = _run_bench(["--instances", "128", "--steps", "2", "--warmup", "1", "--ticks", "10", "--cpu-seconds", "4"])
    assert res["n_gpus"] == 1 and res["config"]["baseline_config"] == 3 and "walking" in res["config"]["workload"]
    rf = res["roofline"]
    assert rf["bound"] == "fp64-valu" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["hbm"]["unit"] == "GB/s" and rf["kernel_ms"] > 0 and res["instances_flagged"] == 0
    assert res["config"]["rollout_restarts"] == 0 and res["config"]["mpc_dt"] == 1e-2 and res["timed_region_s"] > 0
    assert rf["nominal_flop_per_tick"] == 8.0e5 and "counted_flop_per_tick" in rf and "frac_counted" in rf
    cb = res["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["cpu_model"]
    assert cb["parity_vs_gpu_last_tick_max_rel"] < 1e-6
    assert cb["literal_2wbc_value"] and cb["literal_2wbc_value"] < cb["value"] * 1.05
    # the informational PCIe-inclusive mode runs the same rollouts (and keeps the CPU leg's inputs intact)
    hio = _run_bench(["--instances", "128", "--steps", "3", "--warmup", "1", "--ticks", "10", "--cpu-seconds", "3", "--host-io"])
    assert hio["config"]["host_io_over_pcie"] is True and hio["instances_flagged"] == 0 and hio["cpu_baseline"]["value"] > 0
    assert hio["cpu_baseline"]["parity_vs_gpu_last_tick_max_rel"] < 1e-6


# ------------------------------------------------------------------------------- BASELINE config 5: fp32-vs-fp64 tolerance sweep
def test_config5_precision_sweep_pass_rates():
    """north_star's "fp32 vs fp64 tolerance sweep" on SURVEY 8d's schedule (0.4 s stance, 0.15 s flight, landing; N = 48): the sweep of
    scripts/precision_sweep.py on 256 robots (profiles/r02_precision_sweep.json holds the 1024-robot table).  LMH_PRECISION_MIXED = fp32
    model terms; LMH_PRECISION_FP32 = the QP too (push-through contact solves with one fp64 residual refinement, fp64 fall-back route for
    rank-deficient contact sets).  What must hold in every mode: the preview index bit-identical, flight forces exactly zero, no failure
    flag.  The pass rates asserted are what the arithmetic delivers (measured, then fixed with a margin), not the 1e-6 fp64 parity bar:
    fp32 holds 1e-2 (tau) / 1e-3 (f) through the stance phase and does NOT hold 1e-2 for every robot at the landing -- the sweep records it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("precision_sweep", os.path.join(ROOT, "scripts", "precision_sweep.py"))
    ps = importlib.util.module_from_spec(spec); spec.loader.exec_module(ps)
    r = ps.sweep(B=256, nt=600)
    assert r["k_bit_identical"] and r["flight_forces_exactly_zero"]
    assert r["instances_flagged"] == {"fp64": 0, "mixed": 0, "fp32": 0}
    mx, f32 = r["mixed"], r["fp32"]
    assert mx["evaluation"]["tau"]["pass_rate"]["0.0001"] >= 0.99 and mx["evaluation"]["f"]["pass_rate"]["0.0001"] >= 0.99
    assert mx["closed_loop_stance_only"]["tau"]["pass_rate"]["0.0001"] >= 0.999
    assert f32["evaluation_stance_only"]["tau"]["pass_rate"]["0.01"] >= 0.99 and f32["evaluation_stance_only"]["f"]["pass_rate"]["0.001"] >= 0.99
    assert f32["closed_loop_stance_only"]["tau"]["pass_rate"]["0.01"] >= 0.99 and f32["closed_loop_stance_only"]["f"]["pass_rate"]["0.001"] >= 0.99
    assert f32["evaluation"]["tau"]["p50"] < 5e-3 and f32["evaluation"]["tau"]["p50"] > 1e-5      # it IS fp32 arithmetic, not the fp64 path
    # the stance phase never needs the fp64 route (all 32 generators free, or full-rank subsets); the landing does
    pc = r["fp32_chunks_with_fp64_route"]["per_chunk"]
    assert max(pc[:40]) == 0.0


@pytest.mark.parametrize("over", [dict(), dict(w_com_ang=50.0, mu=0.5)])
def test_fp32_mode_single_evaluations_against_fp64(cfg2, over):
    """LMH_PRECISION_FP32 on single evaluations in double support, both single supports and flight, default weights and the 18-row
    set-up (angular-momentum weight set): against the fp64 path on the same states -- torques within 5e-2 / forces within 5e-3 of
    the vector scale for >= 95 % of the robots (measured: tau p50 3e-3..5e-3, p95 1.6e-2; f p50 6e-4, p95 1.5e-3 -- the error is the
    fp32 Woodbury / Schur set-up's, the flight phase without any contact solve shows the same 2e-3), the active-set iteration takes the
    same number of rounds as in fp64, forces of a foot out of support exactly zero, friction cones hold to fp32 round-off, no failure
    flag; and the result differs from fp64 (it is not the fp64 path)."""
    from linearmpchumanoid_amd import capi
    B = 256
    v = perturbed_velocities(B, seed=77) * 0.2
    vprev = v - perturbed_velocities(B, seed=78) * 0.004
    mu = over.get("mu", 0.7)
    hard = capi.FLAG_QP_MAXITER | capi.FLAG_NONFINITE | capi.FLAG_ZMP_RANGE | capi.FLAG_NOT_SPD
    for ph in (0, 1, 2, 3):
        res = {}
        for prec in (capi.PRECISION_FP64, capi.PRECISION_FP32):
            ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=0, precision=prec, **over)
            n = 2500
            ctl.set_refs(np.zeros(n), np.zeros(n), np.full(n, ph, dtype=np.uint8))
            st = ctl.new_state(cfg2["q0"], v, t=0.0, v_prev=vprev)
            out, status = ctl.stand_step(st)[:2]
            torch.cuda.synchronize()
            res[prec] = (out.cpu().numpy(), status.cpu().numpy())
        o64, s64 = res[capi.PRECISION_FP64]
        o32, s32 = res[capi.PRECISION_FP32]
        assert ((s32[:, 2] & hard) == 0).all() and (s64[:, 2] == 0).all() and np.array_equal(s32[:, 0], s64[:, 0])
        e_tau = np.abs(o32[:, :24] - o64[:, :24]).max(axis=1) / np.abs(o64[:, :24]).max(axis=1)
        assert (e_tau <= 5e-2).mean() >= 0.95 and 1e-6 < np.median(e_tau) < 1e-2, (ph, np.median(e_tau), e_tau.max())
        assert np.abs(s32[:, 1] - s64[:, 1]).max() <= 2           # same active-set path (a round more or less at a degenerate vertex)
        f64_, f32_ = o64[:, 24:36], o32[:, 24:36]
        if ph == 3:
            assert np.abs(f32_).max() == 0.0
            continue
        e_f = np.abs(f32_ - f64_).max(axis=1) / np.abs(f64_).max(axis=1)
        assert (e_f <= 5e-3).mean() >= 0.95, (ph, np.median(e_f), e_f.max())
        if ph == 1: assert np.abs(f32_[:, 6:]).max() == 0.0        # left foot out of support
        if ph == 2: assert np.abs(f32_[:, :6]).max() == 0.0
        for ft in range(2):
            fx, fy, fz = f32_[:, 6 * ft + 3], f32_[:, 6 * ft + 4], f32_[:, 6 * ft + 5]
            sc = 1e-4 * (1.0 + np.abs(f32_).max(axis=1))
            assert (fz > -sc).all() and (np.abs(fx) <= mu * fz + sc).all() and (np.abs(fy) <= mu * fz + sc).all()


def test_rccl_collectives_of_the_bench_on_device_tensors(cfg2):
    """The collectives bench.py issues with --backend nccl (RCCL): barrier, all_gather of the [B,16] summaries (device tensors, straight
    from lmh_make_summary), MAX / SUM all_reduce -- on a one-rank RCCL group, which is all a one-GPU box can form (two ranks on one
    card are refused by RCCL; the two-rank path runs over gloo in test_bench_two_ranks_through_the_gpus_flag)."""
    import torch.distributed as dist
    from linearmpchumanoid_amd import sharding
    B = 64
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=1)
    ctl.set_refs_stance(1.0, 2)
    st = ctl.new_state(cfg2["q0"], perturbed_velocities(B), t=0.0)
    out, status, _ = ctl.rollout(st, 5)
    summary = sharding.make_summary(st, out, status, ctl)
    assert summary.is_cuda
    assert not dist.is_initialized()
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        dist.barrier(device_ids=[torch.cuda.current_device()])
        g = sharding.gather_summaries(summary, 1, 0)
        assert g.is_cuda and g.shape == (B, 16) and torch.equal(g, summary)
        tt = torch.tensor([1.25], dtype=torch.float64, device=summary.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ft = torch.tensor([3], dtype=torch.int64, device=summary.device)
        dist.all_reduce(ft, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert float(tt.item()) == 1.25 and int(ft.item()) == 3
    finally:
        dist.destroy_process_group()


def test_resident_workgroups_cover_every_robot_of_a_ragged_batch(cfg2):
    """lmh_rollout runs a grid of resident workgroups (4 per CU) that loop over their robots: a batch that is neither below the grid
    size nor a multiple of it (2500 robots = 2.44 rounds on 256 CUs) must give every robot exactly what it gets in a small batch --
    bit-identical state, outputs and status for robots of the first, the middle and the ragged last round."""
    B, nt = 2500, 6
    v = perturbed_velocities(B, seed=4242)
    ctl = make_controller(B, cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=1)
    ctl.set_refs_stance(1.0, 2)
    st = ctl.new_state(cfg2["q0"], v, t=0.0)
    out, status, log = ctl.rollout(st, nt, log=True)
    torch.cuda.synchronize()
    pick = np.array([0, 1, 511, 1023, 1024, 1025, 2047, 2048, 2300, 2498, 2499])
    small = make_controller(len(pick), cfg2["dt"], cfg2["th"], cfg2["zcom"], warm_start=1)
    small.set_refs_stance(1.0, 2)
    st2 = small.new_state(cfg2["q0"], v[pick], t=0.0)
    out2, status2, log2 = small.rollout(st2, nt, log=True)
    torch.cuda.synchronize()
    assert np.array_equal(st.cpu().numpy()[pick, :91], st2.cpu().numpy()[:, :91])
    assert np.array_equal(out.cpu().numpy()[pick, :78], out2.cpu().numpy()[:, :78])
    assert np.array_equal(status.cpu().numpy()[pick], status2.cpu().numpy())
    assert np.array_equal(log.cpu().numpy()[:, pick], log2.cpu().numpy())
    assert (status.cpu().numpy()[:, 2] == 0).all() and (status.cpu().numpy()[:, 0] == nt).all()
