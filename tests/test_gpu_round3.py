"""GPU parity tests added in round 3: the MPC sample time decoupled from the control step (lmh_config.mpc_dt), the long stable
closed loops of BASELINE configs 2 / 3 / 5 that this makes possible (2 000 / 4 000 ticks), the plant with its velocity
products at the current state, the dynamic robot -> workgroup assignment of the rollout kernel, and bench.py's flag accounting.
Same rules as the other GPU files: HIP path through the C ABI against the CPU oracle; relative 1e-6 on tau / f (helpers.close:
max|a - b| / max|b| of the same vector, no absolute floor), bit-exact k.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import TOL_REL, WEIGHT, close, close_on, perturbed_velocities, vec_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DT, MPC_DT = 1e-3, 1e-2


def horizon_time(N, mpc_dt):
    return N * mpc_dt + 1e-9                                       # int(th / mpc_dt) == N whatever the rounding of the quotient


def make_controller(B, N, zcom, mpc_dt=MPC_DT, **kw):
    from linearmpchumanoid_amd.controller import BatchedController, default_config
    return BatchedController(B, default_config(dt=DT, time_horizon=horizon_time(N, mpc_dt), z_com=zcom, mpc_dt=mpc_dt, **kw))


def make_oracle(N, sim_time, mpc_dt=MPC_DT):
    from oracle.pyoracle import Oracle
    return Oracle(sim_time=sim_time, dt=mpc_dt, horizon_time=horizon_time(N, mpc_dt), do_ik=True)


@pytest.fixture(scope="module")
def nao():
    o = make_oracle(32, 1.0)
    return dict(zcom=o.zcom, q0=o.robot()["q"].copy())


def k_sequence(nt, mpc_dt, t0=0.0):
    """k of the k4-stage evaluation of every tick: int((t + dt) / mpc_dt) on the float-accumulated clock (Clock.hpp:11, rk4.hpp:15)."""
    t, ks = t0, []
    for _ in range(nt):
        ks.append(int((t + DT) / mpc_dt))
        t += DT
    return ks


# ------------------------------------------------------------------------------- mpc_dt: parity
def test_mpc_dt_walking_parity_against_oracle(nao):
    """dt = 1 ms, mpc_dt = 10 ms, N = 32: walking with contact switching generated on the device on the
    10 ms sample grid, 900 ticks (settle, DS, SS-R, DS, into SS-L); every robot against an Oracle(dt = mpc_dt).rollout(dt = 1 ms):
    tau / f every 7th tick, the state, k of every launch boundary bit-exact; the gain row and the reference arrays are the MPC
    sample time's."""
    B, N, nt = 6, 32, 900
    sim = 2.0
    ctl = make_controller(B, N, nao["zcom"], warm_start=1)
    assert ctl.N == N
    ctl.gen_walk(sim, num_steps=3, time_per_step=0.5, ds_time=0.2, step_height=0.02, settle_time=0.1)
    plan = ctl.get_refs()
    assert len(plan["zmp_x"]) == int((sim + 0.5) / MPC_DT)         # ZMP::stanceZMP's count with timeStep = mpc_dt
    xs = np.linspace(0.02, 0.05, B)
    ctl.set_xscale(xs)
    o0 = make_oracle(N, sim)
    assert np.abs(ctl.mpc_gain() - o0.gain_row()).max() < 1e-12 * np.abs(o0.gain_row()).max()
    st = ctl.new_state(nao["q0"], np.zeros(30), t=0.0)
    logs, ks = [], []
    for c in range(3):                                             # three launches: the clock and v_prev carry over
        out, status, log = ctl.rollout(st, nt // 3, log=True)
        torch.cuda.synchronize()
        logs.append(log.cpu().numpy()); ks.append(status.cpu().numpy()[:, 0].copy())
        assert (status.cpu().numpy()[:, 2] == 0).all()
    log = np.concatenate(logs, axis=0)
    stn = st.cpu().numpy()
    kref = k_sequence(nt, MPC_DT)
    for c in range(3):
        assert (ks[c] == kref[(c + 1) * (nt // 3) - 1]).all()
    phases = set()
    for i in range(B):
        o = make_oracle(N, sim)
        o.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        o.set_segments(plan["segs"], plan["seg_of_sample"], xscale=float(xs[i]))
        r = o.rollout(np.concatenate([nao["q0"], np.zeros(30)]), 0.0, nt, dt=DT, log=True)
        assert list(r["k"]) == kref
        assert close(stn[i, :60], r["state"], 1e-7) and abs(stn[i, 90] - r["t"]) == 0.0
        for tk in range(0, nt, 7):
            ref = r["log"][tk]
            assert close(log[tk, i, :24], ref[:24], TOL_REL), (i, tk, vec_err(log[tk, i, :24], ref[:24]))
            assert close(log[tk, i, 24:], ref[24:], TOL_REL, scale=WEIGHT), (i, tk, vec_err(log[tk, i, 24:], ref[24:]))
            phases.add(int(plan["phase"][kref[tk]]))
    assert phases == {0, 1, 2}


def test_mpc_dt_single_evaluation_and_default(nao):
    """lmh_eval with mpc_dt: k = int(t / mpc_dt), xRef / yRef one MPC sample ahead (A, B of mpcLinearPendulum.cpp:45-47 with the MPC's
    dt); mpc_dt = 0 and mpc_dt = dt are the same controller bit for bit."""
    B, N = 8, 32
    v = perturbed_velocities(B, seed=77)
    ctl = make_controller(B, N, nao["zcom"], warm_start=0)
    ctl.set_refs_stance(2.0, 2)
    t_eval = 0.5371
    st = ctl.new_state(nao["q0"], v, t=t_eval)
    out, status = ctl.stand_step(st)
    torch.cuda.synchronize()
    out, status = out.cpu().numpy(), status.cpu().numpy()
    assert (status[:, 0] == int(t_eval / MPC_DT)).all() and (status[:, 2] == 0).all()
    for i in range(B):
        o = make_oracle(N, 2.0)
        e = o.eval(nao["q0"], v[i], t_eval)
        qp = o.qp()
        assert e["k"] == status[i, 0]
        assert close(out[i, :24], e["tau"]) and close(out[i, 24:36], e["f"], scale=WEIGHT) and close(out[i, 36:66], e["qpp"])
        sc = np.abs(qp["mpcRef"]).max()
        assert close_on(out[i, 72:75], qp["mpcRef"][:3], 1e-9, sc) and close_on(out[i, 75:78], qp["mpcRef"][3:], 1e-9, sc)
    from linearmpchumanoid_amd.controller import BatchedController, default_config
    outs = []
    for md in (0.0, DT):
        c = BatchedController(B, default_config(dt=DT, time_horizon=0.032, z_com=nao["zcom"], mpc_dt=md, warm_start=1))
        c.set_refs_stance(1.0, 2)
        s = c.new_state(nao["q0"], v, t=0.0)
        o_, st_, _ = c.rollout(s, 20)
        torch.cuda.synchronize()
        outs.append((o_.cpu().numpy().copy(), s.cpu().numpy().copy(), st_.cpu().numpy().copy()))
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))


# ------------------------------------------------------------------------------- the long closed loops of SURVEY 8d
def _rollout_chunks(ctl, st, nt, chunk, sample):
    """nt ticks in launches of `chunk`; returns (flags OR-ed over every launch [B], max QP rounds, tau|f log of the sampled robots
    [nt, len(sample), 36], k after every launch)."""
    B = st.shape[0]
    flags = torch.zeros((B,), dtype=torch.int32, device=st.device)
    itmax = torch.zeros((B,), dtype=torch.int32, device=st.device)
    out, status = ctl.new_out(), ctl.new_status()
    log = torch.zeros((chunk, B, 36), dtype=torch.float64, device=st.device)
    idx = torch.as_tensor(np.asarray(sample), device=st.device)
    keep, ks = [], []
    for c in range(nt // chunk):
        ctl.rollout(st, chunk, out, status, log)
        flags |= status[:, 2]
        itmax = torch.maximum(itmax, status[:, 1])
        keep.append(log[:, idx, :].cpu().numpy())
        ks.append(int(status[0, 0].item()))
    torch.cuda.synchronize()
    return flags.cpu().numpy(), itmax.cpu().numpy(), np.concatenate(keep, axis=0), ks, out.cpu().numpy()


def test_config3_walkers_complete_4000_ticks_without_a_flag(nao):
    """BASELINE configs[2] over SURVEY 8d's full length: 4096 walkers, per-instance step length U(0.02, 0.05) m
    (seed 20260003 + i), dt = 1 ms, N = 32 preview samples of mpc_dt = 10 ms, the reference's default timePerStep = 0.5 s
    (zmpGeneration.hpp:37-38) with 0.2 s of double support, 4 000 ticks = 7 steps: no status flag in any launch, every robot has
    walked (base x advanced by the planned distance), forces inside the cone, the swing foot carries exactly nothing; three robots
    against their oracle rollouts over the whole 4 000 ticks."""
    B, N, nt, chunk = 4096, 32, 4000, 500
    sim = nt * DT + 1.0
    xs = np.array([np.random.default_rng(20260003 + i).uniform(0.02, 0.05) for i in range(B)])
    ctl = make_controller(B, N, nao["zcom"], warm_start=1)
    n_steps = int((sim - 0.3) / 0.5)
    ctl.gen_walk(sim, num_steps=n_steps, time_per_step=0.5, ds_time=0.2, step_height=0.02, settle_time=0.3)
    plan = ctl.get_refs()
    ctl.set_xscale(xs)
    st = ctl.new_state(nao["q0"], np.zeros(30), t=0.0)
    sample = [0, 1777, 4095]
    flags, itmax, log, ks, out = _rollout_chunks(ctl, st, nt, chunk, sample)
    assert (flags == 0).all(), f"{int((flags != 0).sum())} walkers flagged: {np.unique(flags)}"
    kref = k_sequence(nt, MPC_DT)
    assert ks == [kref[(c + 1) * chunk - 1] for c in range(nt // chunk)]
    stn = st.cpu().numpy()
    assert np.isfinite(stn[:, :60]).all() and np.abs(stn[:, 30:60]).max() < 10.0
    # the robots have walked: the base has advanced with the mid-point of the planned footholds (it leads it by a few centimetres in
    # the middle of a step)
    seg_last = plan["segs"][plan["seg_of_sample"][kref[-1]]]
    mid = 0.5 * (seg_last[1] + seg_last[25]) * xs                  # rF x0, lF x0 of the current segment, in units of the step length
    assert np.abs((stn[:, 0] - nao["q0"][0]) - mid).max() < 0.06 and (stn[:, 0] - nao["q0"][0]).min() > 0.1
    mu = 0.7
    f = out[:, 24:36]
    for ft in range(2):
        fx, fy, fz = f[:, 6 * ft + 3], f[:, 6 * ft + 4], f[:, 6 * ft + 5]
        assert (fz > -1e-7).all() and (np.abs(fx) <= mu * fz + 1e-7).all() and (np.abs(fy) <= mu * fz + 1e-7).all()
    seen = set()
    for tk in range(0, nt, 10):
        ph = int(plan["phase"][kref[tk]])
        seen.add(ph)
        if ph == 1:
            assert np.abs(log[tk, :, 24 + 6:36]).max() == 0.0
        if ph == 2:
            assert np.abs(log[tk, :, 24:24 + 6]).max() == 0.0
    assert seen == {0, 1, 2}
    worst = 0.0
    for j, i in enumerate(sample):
        o = make_oracle(N, sim)
        o.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        o.set_segments(plan["segs"], plan["seg_of_sample"], xscale=float(xs[i]))
        r = o.rollout(np.concatenate([nao["q0"], np.zeros(30)]), 0.0, nt, dt=DT, log=True)
        assert list(r["k"]) == kref
        for tk in list(range(0, nt, 50)) + [nt - 1]:
            ref = r["log"][tk]
            et, ef = vec_err(log[tk, j, :24], ref[:24]), vec_err(log[tk, j, 24:], ref[24:])
            worst = max(worst, et, ef)
            assert et <= TOL_REL and ef <= TOL_REL, (i, tk, et, ef)
        assert close(stn[i, :60], r["state"], 1e-6)
    print(f"config 3, 4000 ticks: worst tau/f error of the sampled robots {worst:.2e}, max QP rounds {int(itmax.max())}")


def test_config2_balancers_complete_2000_ticks_without_a_flag(nao):
    """BASELINE configs[1] over SURVEY 8d's 2 000 ticks: 1024 robots, velocity pushes (seed 20260001 + i), N = 16 preview samples.
    16 x 10 ms = 0.16 s of preview is the LIPM's time constant and still diverges (oracle: NaN by tick 2000, see
    test_oracle.test_short_previews_diverge...), so the balance configuration samples its preview at mpc_dt = 20 ms (0.32 s).
    What can be balanced at all is bounded by the support polygon, not by the controller: the CoM starts 0.03 m in front of the heel
    edge (x = -0.05, Robot.cpp:38-42), so a backward push beyond 0.03 sqrt(g / z_c) = 0.18 m/s puts the capture point behind the
    heel and the robot tips over whatever the torques do.  (i) SURVEY's U(-0.3, 0.3) m/s pushes: the robots that end up flagged are
    exactly the ones pushed backwards beyond that limit -- every robot with v_x > -0.15 m/s completes the 2 000 ticks; (ii) the same
    seeds at half the amplitude (all inside the capture region; what bench.py --config 2 runs): no flag at all, the pushes are
    absorbed, two robots against the oracle over the whole range."""
    B, N, nt, chunk, md = 1024, 16, 2000, 500, 2e-2
    sim = nt * DT + 1.0
    v_full = perturbed_velocities(B)
    ctl = make_controller(B, N, nao["zcom"], mpc_dt=md, warm_start=1)
    ctl.set_refs_stance(sim, 2)
    st = ctl.new_state(nao["q0"], v_full, t=0.0)
    flags, _, _, _, _ = _rollout_chunks(ctl, st, nt, chunk, [0])
    fell = flags != 0
    lim = 0.03 * np.sqrt(9.81 / nao["zcom"])
    assert 0.17 < lim < 0.19
    assert fell.any() and (v_full[fell, 0] < -0.15).all(), (int(fell.sum()), v_full[fell, 0].max())
    assert not fell[v_full[:, 0] > -0.15].any() and fell[v_full[:, 0] < -0.22].all()
    v = 0.5 * v_full
    st = ctl.new_state(nao["q0"], v, t=0.0)
    sample = [3, 1000]
    flags, itmax, log, ks, out = _rollout_chunks(ctl, st, nt, chunk, sample)
    assert (flags == 0).all(), f"{int((flags != 0).sum())} flagged: {np.unique(flags)}"
    kref = k_sequence(nt, md)
    assert ks == [kref[(c + 1) * chunk - 1] for c in range(nt // chunk)]
    stn = st.cpu().numpy()
    assert np.abs(stn[:, 30:60]).max() < 0.1 and np.abs(stn[:, 30:32]).max() < 0.02       # pushes were up to 0.15 m/s
    for j, i in enumerate(sample):
        o = make_oracle(N, sim, mpc_dt=md)
        r = o.rollout(np.concatenate([nao["q0"], v[i]]), 0.0, nt, dt=DT, log=True)
        assert list(r["k"]) == kref
        for tk in list(range(0, nt, 40)) + [nt - 1]:
            ref = r["log"][tk]
            assert close(log[tk, j, :24], ref[:24]) and close(log[tk, j, 24:], ref[24:], scale=WEIGHT), (i, tk)
        assert close(stn[i, :60], r["state"], 1e-6)


def test_config5_jump_schedule_2000_ticks_n48(nao):
    """BASELINE configs[4] at one GPU's share, fp64: 4096 robots, N = 48 x 10 ms, stance 0.4 s / flight 0.15 s / double support
    (SURVEY 8d), 2 000 ticks through take-off, flight (forces exactly 0) and the landing transient: no flag, the robots settle back
    towards their stance height; two robots against the oracle (the whole range is finite now that the preview is 0.48 s)."""
    B, N, nt, chunk = 4096, 48, 2000, 500
    sim = nt * DT + 1.0
    v = perturbed_velocities(B, seed=20260005) * 0.2
    ctl = make_controller(B, N, nao["zcom"], warm_start=1)
    ctl.gen_jump(sim, 0.4, 0.15)
    plan = ctl.get_refs()
    st = ctl.new_state(nao["q0"], v, t=0.0)
    sample = [5, 4000]
    flags, itmax, log, ks, out = _rollout_chunks(ctl, st, nt, chunk, sample)
    assert (flags == 0).all(), f"{int((flags != 0).sum())} flagged: {np.unique(flags)}"
    kref = k_sequence(nt, MPC_DT)
    fl = [tk for tk in range(nt) if plan["phase"][kref[tk]] == 3]
    assert len(fl) == 150 and np.abs(log[fl][:, :, 24:]).max() == 0.0
    stn = st.cpu().numpy()
    assert np.isfinite(stn[:, :60]).all() and np.abs(stn[:, 2] - nao["q0"][2]).max() < 0.05
    for j, i in enumerate(sample):
        o = make_oracle(N, sim)
        o.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        r = o.rollout(np.concatenate([nao["q0"], v[i]]), 0.0, nt, dt=DT, log=True)
        for tk in list(range(0, nt, 25)) + [nt - 1]:
            ref = r["log"][tk]
            assert close(log[tk, j, :24], ref[:24]) and close(log[tk, j, 24:], ref[24:], scale=WEIGHT), (i, tk)
        assert close(stn[i, :60], r["state"], 1e-6)


# ------------------------------------------------------------------------------- plant: velocity products at the current state
def test_plant_free_fall_and_momentum(nao):
    """VERDICT r02 item 5 / SURVEY 8f #3: M qdd = S'tau + J'w_c - C(q, v) with C at the velocity of the state being integrated (a second
    Newton-Euler pass; the controller's own C keeps the reference's stale Robot::v_, controller.cpp:56 vs :59).  Robots released 0.2 m
    above the ground in the flight phase (no contact force asked for, none supplied; the foot references are raised with them), with
    random joint and base angular velocities: internal torques cannot move the CoM, so it follows c0 + v0 t - g t^2 / 2 and the
    angular momentum about it stays put.  Over 50 ms: |CoM - ballistic| <= 1e-6 m (measured 1.3e-7; the remainder is the model's own
    inconsistency -- the 0.7071 literals of Robot.cpp:93-103 make the leg rotations non-orthonormal by 1.9e-5, which shows as 5e-4 m/s^2
    in the CoM acceleration of M, C at rest), angular momentum within 1e-5 kg m^2/s.  And against the oracle's plant."""
    from linearmpchumanoid_amd.controller import BatchedController, default_config
    from oracle.pyoracle import Oracle
    B, nt, h, th = 4, 50, 0.2, 0.032
    ctl = BatchedController(B, default_config(dt=DT, time_horizon=th, z_com=nao["zcom"], warm_start=1, plant=1))
    n = 1500
    ctl.set_refs(np.zeros(n), np.zeros(n), np.full(n, 3, dtype=np.uint8))          # flight
    rF = np.zeros((3, 8)); lF = np.zeros((3, 8))
    rF[1, 0], lF[1, 0], rF[2, 0], lF[2, 0] = -0.05, 0.05, h, h
    ctl.set_foot_coeffs(rF, [6, 6, 8], lF, [6, 6, 8])
    q = nao["q0"].copy(); q[2] += h
    v = np.zeros((B, 30))
    for i in range(1, B):
        rng = np.random.default_rng(900 + i)
        v[i, 6:] = rng.normal(0.0, 0.3, 24); v[i, 3:6] = rng.uniform(-0.3, 0.3, 3)
    st = ctl.new_state(q, v, t=0.0)
    o0, _ = ctl.stand_step(st.clone())                             # CoM, comVel, (debug) angular momentum of the initial state
    _, _, d0 = ctl.stand_step(st.clone(), debug=True)
    out, status, log = ctl.rollout(st, nt, log=True)
    s1 = st.clone()
    o1, _ = ctl.stand_step(s1.clone())
    _, _, d1 = ctl.stand_step(s1.clone(), debug=True)
    torch.cuda.synchronize()
    from linearmpchumanoid_amd.controller import unpack_debug
    o0, o1, stn = o0.cpu().numpy(), o1.cpu().numpy(), st.cpu().numpy()
    assert (status.cpu().numpy()[:, 2] == 0).all() and np.abs(log.cpu().numpy()[:, :, 24:]).max() == 0.0
    T = nt * DT
    for i in range(B):
        c0, cv0, c1, cv1 = o0[i, 66:69], o0[i, 69:72], o1[i, 66:69], o1[i, 69:72]
        ball = c0 + cv0 * T + np.array([0.0, 0.0, -0.5 * 9.81 * T * T])
        assert np.abs(c1 - ball).max() <= 1e-6, (i, c1 - ball)
        # 5.4e-4 m/s^2 (the model's inconsistency at rest, test_oracle.test_plant_of_the_oracle_is_physical) x 50 ms = 2.7e-5 m/s
        assert np.abs(cv1 - (cv0 + np.array([0.0, 0.0, -9.81 * T]))).max() <= 5e-5, (i, cv1 - cv0)
        am0, am1 = unpack_debug(d0.cpu().numpy()[i])["angMom"], unpack_debug(d1.cpu().numpy()[i])["angMom"]
        assert np.abs(am1 - am0).max() <= 1e-5, (i, am1 - am0)
        o = Oracle(sim_time=1.0, dt=DT, horizon_time=th, do_ik=True)
        o.set_zcom(nao["zcom"])
        o.set_refs(np.zeros(n), np.zeros(n), np.full(n, 3, dtype=np.uint8))
        o.set_foot_coeffs(rF, [6, 6, 8], lF, [6, 6, 8])
        o.set_plant(True, k=ctl.cfg.contact_k, d=ctl.cfg.contact_d, dt=ctl.cfg.contact_dt, mu=ctl.cfg.contact_mu)
        r = o.rollout(np.concatenate([q, v[i]]), 0.0, nt, log=True)
        assert close(stn[i, :60], r["state"], 1e-7), (i, vec_err(stn[i, :60], r["state"]))


# ------------------------------------------------------------------------------- dynamic robot -> workgroup assignment
def test_ticket_scheduling_is_result_neutral_and_reusable(nao):
    """The rollout kernel's resident workgroups draw their robots from a global ticket (one atomicAdd per robot) instead of a static
    stride.  Which workgroup ran a robot must not matter: a batch far larger than the resident grid (1024 + ... robots with very
    different QP round counts: cold start, strong pushes) equals, robot by robot and bit for bit, the same robots run in small
    batches that fit one round; the ticket words are back at zero after every launch (launching again on the same handle, on two
    streams, gives the same bits)."""
    B, N, nt = 2600, 16, 12
    v = perturbed_velocities(B, seed=4242) * 2.0
    ctl = make_controller(B, N, nao["zcom"], mpc_dt=2e-2, warm_start=0)
    ctl.set_refs_stance(1.0, 2)
    st = ctl.new_state(nao["q0"], v, t=0.0)
    out, status, _ = ctl.rollout(st, nt)
    torch.cuda.synchronize()
    ref_out, ref_st, ref_status = out.cpu().numpy().copy(), st.cpu().numpy().copy(), status.cpu().numpy().copy()
    assert (ref_status[:, 2] == 0).all() and ref_status[:, 1].max() > ref_status[:, 1].min()      # uneven work per robot
    for lo in (0, 1111, 2400):
        small = make_controller(200, N, nao["zcom"], mpc_dt=2e-2, warm_start=0)
        small.set_refs_stance(1.0, 2)
        s2 = small.new_state(nao["q0"], v[lo:lo + 200], t=0.0)
        o2, t2, _ = small.rollout(s2, nt)
        torch.cuda.synchronize()
        assert np.array_equal(o2.cpu().numpy(), ref_out[lo:lo + 200]) and np.array_equal(s2.cpu().numpy(), ref_st[lo:lo + 200])
        assert np.array_equal(t2.cpu().numpy(), ref_status[lo:lo + 200])
    # the same handle again, twelve more launches than it has launch slots, alternating between two streams
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    results = []
    for rep in range(12):
        with torch.cuda.stream(streams[rep % 2]):
            s3 = ctl.new_state(nao["q0"], v, t=0.0)
            o3, t3, _ = ctl.rollout(s3, nt)
            results.append((o3, s3))
    torch.cuda.synchronize()
    for o3, s3 in results:
        assert np.array_equal(o3.cpu().numpy(), ref_out) and np.array_equal(s3.cpu().numpy(), ref_st)


def test_chunked_work_units_equal_a_sequence_of_short_launches(nao):
    """Inside one launch a robot is run in chunks of 250 ticks and goes back to a ring queue in between (another workgroup may take the
    next chunk).  A 620-tick launch of 1500 robots (more than the resident grid: fresh robots and ring entries both occur; pushes
    make the QP round counts uneven) must equal, bit for bit, three launches of 250 + 250 + 120 ticks (the kernel's own chunk boundaries) and two of 100 + 520: state, out record and log
    rows; status[0] (k) and [3] (active set) are those of the last tick, [1] the maximum and [2] the OR over the whole launch.
    Launching again on the same handle (queue words back at zero) reproduces it."""
    B, N = 1500, 16
    v = perturbed_velocities(B, seed=777)
    ctl = make_controller(B, N, nao["zcom"], mpc_dt=2e-2, warm_start=1)
    ctl.set_refs_stance(2.0, 2)
    for rep in range(2):
        st = ctl.new_state(nao["q0"], v, t=0.0)
        log = torch.zeros((620, B, 36), dtype=torch.float64, device=ctl.device)
        out, status, _ = ctl.rollout(st, 620, log=log)
        torch.cuda.synchronize()
        st2 = ctl.new_state(nao["q0"], v, t=0.0)
        out2, status2 = ctl.new_out(), ctl.new_status()
        itmax = np.zeros(B, dtype=np.int64); flags = np.zeros(B, dtype=np.int64)
        logs = []
        for n in ((250, 250, 120) if rep == 0 else (100, 520)):     # split at the kernel's own chunk boundaries, and elsewhere
            lg = torch.zeros((n, B, 36), dtype=torch.float64, device=ctl.device)
            ctl.rollout(st2, n, out2, status2, lg)
            torch.cuda.synchronize()
            s_ = status2.cpu().numpy()
            itmax = np.maximum(itmax, s_[:, 1]); flags |= s_[:, 2]
            logs.append(lg.cpu().numpy())
        a, b = status.cpu().numpy(), status2.cpu().numpy()
        assert np.array_equal(st.cpu().numpy(), st2.cpu().numpy()) and np.array_equal(out.cpu().numpy(), out2.cpu().numpy())
        assert np.array_equal(log.cpu().numpy(), np.concatenate(logs, axis=0))
        assert np.array_equal(a[:, 0], b[:, 0]) and np.array_equal(a[:, 3], b[:, 3])
        assert np.array_equal(a[:, 1], itmax) and np.array_equal(a[:, 2], flags)
        assert (a[:, 2] == 0).all() and a[:, 1].max() > 1
    # two multi-chunk launches sharing the chip (two streams, two launch slots of the handle): neither has all of its workgroups resident
    # from the start, and no workgroup owns a unit it has not claimed, so both drain; results as above
    ref_state, ref_out = st.cpu().numpy().copy(), out.cpu().numpy().copy()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    res = []
    for sm in streams:
        with torch.cuda.stream(sm):
            s3 = ctl.new_state(nao["q0"], v, t=0.0)
            o3, t3, _ = ctl.rollout(s3, 620)
            res.append((s3, o3))
    torch.cuda.synchronize()
    for s3, o3 in res:
        assert np.array_equal(s3.cpu().numpy(), ref_state) and np.array_equal(o3.cpu().numpy(), ref_out)


# ------------------------------------------------------------------------------- bench.py: flag accounting, precision 2, config 5
def _run_bench(argv, expect_rc=0, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == expect_rc, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_counts_flags_of_every_launch_not_only_the_last():
    """the rollout kernel overwrites status[:, 2] per launch, so a flag raised in a middle launch used to vanish from the
    bench line.  Jump schedule in five launches of 200 ticks with the QP round cap at 3: the landing (launch 3, ticks 400..600) needs
    more rounds -> LMH_FLAG_QP_MAXITER there, the settled last launch is clean: exit code 3 and instances_flagged > 0 although
    the last launch alone shows none.  The same run without the cap is clean (exit code 0)."""
    argv = ["--config", "5", "--instances", "64", "--ticks", "200", "--warmup", "0", "--steps", "8", "--reset-every", "0", "--no-cpu-baseline"]
    res = _run_bench(argv + ["--max-qp-iters", "3"], expect_rc=3)
    assert res["instances_flagged"] > 0 and res["instances_flagged_in_last_launch"] == 0
    ok = _run_bench(argv)
    assert ok["instances_flagged"] == 0 and ok["config"]["baseline_config"] == 5 and ok["config"]["rollout_restarts"] == 0


def test_bench_precision_2_and_config5_lines():
    """`--precision 2` (LMH_PRECISION_FP32) prints its line (the dtype table used to raise KeyError after the timed run) and does not
    count the informational LMH_FLAG_QP_FP64_ROUTE as a failure; `--config 5` is the entry point of BASELINE configs[4] (one step = one
    jump from the initial state, restarted per step and reported so)."""
    r2 = _run_bench(["--instances", "128", "--steps", "2", "--warmup", "1", "--ticks", "10", "--no-cpu-baseline", "--precision", "2"])
    assert r2["dtype"].startswith("f32") and r2["instances_flagged"] == 0 and r2["instances_fp64_route"] >= 0
    r5 = _run_bench(["--config", "5", "--instances", "256", "--steps", "2", "--warmup", "1", "--ticks", "700", "--cpu-seconds", "4"])
    assert r5["config"]["rollout_restarts"] == 2 and "jump" in r5["config"]["workload"] and r5["instances_flagged"] == 0
    assert r5["config"]["preview_s"] == pytest.approx(0.48) and r5["cpu_baseline"]["parity_vs_gpu_last_tick_max_rel"] < 1e-6


def test_bench_config5_on_two_ranks_over_gloo():
    """N > 1 readiness of the config-5 entry point (BASELINE configs[4] is quoted on 8 GPUs): `bench.py --gpus 2 --config 5` starts two
    ranks of itself (both on card 0 of a one-GPU box, gloo), each takes its shard of the jumpers, the summary rows of both come back in
    the one gather, no instance is flagged, and the line says what ran."""
    env_dev = os.environ.get("LMH_BENCH_DEVICE")
    os.environ["LMH_BENCH_DEVICE"] = "0"
    try:
        res = _run_bench(["--gpus", "2", "--backend", "gloo", "--config", "5", "--instances", "128", "--steps", "2", "--warmup", "1", "--ticks", "700",
                          "--no-cpu-baseline"])
    finally:
        if env_dev is None: os.environ.pop("LMH_BENCH_DEVICE", None)
        else: os.environ["LMH_BENCH_DEVICE"] = env_dev
    assert res["n_gpus"] == 2 and res["summary_rows_gathered"] == 256 and res["instances_flagged"] == 0
    assert res["config"]["baseline_config"] == 5 and res["scaling"] == "weak" and "jump" in res["config"]["workload"]
    assert res["config"]["parallelism"].endswith("x2") and res["value"] > 0


def test_config4_decoupled_randomised_walkers(nao):
    """BASELINE configs[3] at one GPU's share in the decoupled form bench.py --config 4 runs per rank: 4096 randomised models (IK
    kernel start postures, per-instance LIPM height and step length), dt = 1 ms, N = 32 x 10 ms, 0.5 s steps: 1 500 ticks
    (settle, three contact switches) without a flag, two robots against oracles built from the same raw link tables."""
    sys.path.insert(0, ROOT)
    import bench
    from oracle.pyoracle import Oracle
    args = bench.parse(["--config", "4"])
    B, nt, chunk = args.instances, 1500, 500
    ctl = make_controller(B, args.horizon, 0.26, mpc_dt=args.mpc_dt, warm_start=1)
    state, host = bench.build_workload(args, ctl, 0, B, nt)
    st = state.clone()
    sample = [7, 3333]
    flags, itmax, log, ks, out = _rollout_chunks(ctl, st, nt, chunk, sample)
    assert (flags == 0).all(), f"{int((flags != 0).sum())} flagged: {np.unique(flags)}"
    kref = k_sequence(nt, args.mpc_dt)
    assert ks == [kref[(c + 1) * chunk - 1] for c in range(nt // chunk)]
    assert {int(host["phase"][k]) for k in kref} == {0, 1, 2}
    stn = st.cpu().numpy()
    for j, i in enumerate(sample):
        o = Oracle(sim_time=nt * DT + 1.0, dt=args.mpc_dt, horizon_time=horizon_time(args.horizon, args.mpc_dt), do_ik=True, raw_links=host["raw"][i])
        assert np.abs(o.robot()["q"] - host["q0"][i]).max() < 1e-10 and abs(o.zcom - host["zcom"][i]) < 1e-12
        o.set_refs(host["zmp_x"], host["zmp_y"], host["phase"])
        o.set_segments(host["segs"], host["sos"], xscale=float(host["xscale"][i]))
        r = o.rollout(np.concatenate([o.robot()["q"], np.zeros(30)]), 0.0, nt, dt=DT, log=True)
        assert list(r["k"]) == kref and close(stn[i, :60], r["state"], 1e-6)
        for tk in list(range(0, nt, 20)) + [nt - 1]:
            assert close(log[tk, j, :24], r["log"][tk][:24]) and close(log[tk, j, 24:], r["log"][tk][24:], scale=WEIGHT), (i, tk)


# ------------------------------------------------------------------------------- no result depends on LDS nobody wrote
_POISON_PROBE = r"""
import hashlib, json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from linearmpchumanoid_amd.controller import BatchedController, default_config
from helpers import perturbed_velocities
ik = json.load(open("tests/golden/ik_posture.json"))
h = hashlib.sha256()
flags = 0
# walking (contact switching, several robots per resident workgroup is not needed: every robot is poisoned at its start), stance, jump
for kind in ("walk", "stance", "jump"):
    B = 96
    ctl = BatchedController(B, default_config(dt=1e-3, time_horizon=0.32 + 1e-9, z_com=ik["z_com"], mpc_dt=1e-2, warm_start=1))
    if kind == "walk":
        ctl.gen_walk(2.0, num_steps=3, time_per_step=0.3, ds_time=0.1, step_height=0.02, settle_time=0.05)
        ctl.set_xscale(np.linspace(0.02, 0.05, B))
    elif kind == "jump":
        ctl.gen_jump(2.0, 0.1, 0.1)
    else:
        ctl.set_refs_stance(2.0, 2)
    st = ctl.new_state(np.array(ik["q"]), perturbed_velocities(B) * 0.3, t=0.0)
    out, status, log = ctl.rollout(st, 420, log=True)
    o2, s2 = ctl.stand_step(st.clone())
    torch.cuda.synchronize()
    for a in (out, status, log, st, o2, s2):
        h.update(a.cpu().numpy().tobytes())
    flags += int((status.cpu().numpy()[:, 2] != 0).sum())
print(json.dumps({"sha": h.hexdigest(), "flags": flags}))
"""


def test_results_do_not_depend_on_uninitialised_lds():
    """The checker build liblmh_hip_var_poison.so (-DLMH_POISON, built by __graft_entry__.build()) fills the whole LDS image of every
    robot with NaNs before anything is loaded.  Walking, stance and jump rollouts (all three support phases, flight, 420 ticks, log,
    a single evaluation on top) must come out flag-free and BIT-IDENTICAL to the shipped library: no slot is read before it is written.
    (Round 3 found one that way: two entries of the joint-space inertia that the matrix-core tiles of the QP set-up over-read against
    zero padding.)"""
    from linearmpchumanoid_amd import build as hipbuild
    so = hipbuild.build_variant("poison", ["-DLMH_POISON"])
    assert os.path.exists(so)
    res = {}
    for variant in ("", "poison"):
        env = {k: v for k, v in os.environ.items() if k not in ("LMH_VARIANT", "LMH_DIAG")}
        if variant:
            env["LMH_VARIANT"] = variant
        r = subprocess.run([sys.executable, "-c", _POISON_PROBE], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[variant] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res[""]["flags"] == 0 and res["poison"]["flags"] == 0, res
    assert res[""]["sha"] == res["poison"]["sha"], res
