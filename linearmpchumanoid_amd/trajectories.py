"""Set-up time reference generators (inputs of the hot path), host side.

Mirrors the reference's trajectory producers: ZMP::stanceZMP (src/zmpGeneration.cpp:39-60),
footCoeffTrajectory (src/footRefTrajectory.cpp:4-47) and findPolyCoeff
(src/generalizedFunctions.cpp:103-163).  The reference declares a walking generator
(zmpGeneration.hpp:15-22 walkZMP) but never defines it; walk_refs() below is the build's own
definition on top of the reference's conventions (SupportFoot -> ZMP y of -/+0.05).
"""
import numpy as np

from .capi import PHASE_DOUBLE, PHASE_LEFT, PHASE_RIGHT, PHASE_FLIGHT


def stance_zmp(simulation_time, time_step, support_foot=2):
    """ZMP::stanceZMP; support_foot 0 Right, 1 Left, 2 Double (Task.hpp:9-13)."""
    samples = int((simulation_time + 0.5) / time_step)
    zx = np.zeros(samples)
    zy = np.full(samples, -0.05 if support_foot == 0 else (0.05 if support_foot == 1 else 0.0))
    return zx, zy


def find_poly_coeff(pos, vel, acc):
    """findPolyCoeff(Pos, Vel, Acc): rows are (t, value); ascending-power coefficients."""
    pos, vel, acc = (np.atleast_2d(np.asarray(a, dtype=np.float64)) for a in (pos, vel, acc))
    n = len(pos) + len(vel) + len(acc)
    A = np.zeros((n, n)); b = np.zeros(n)
    row = 0
    for t, val in pos:
        A[row] = t ** np.arange(n); b[row] = val; row += 1
    for t, val in vel:
        for j in range(1, n):
            A[row, j] = j * t ** (j - 1)
        b[row] = val; row += 1
    for t, val in acc:
        for j in range(2, n):
            A[row, j] = j * (j - 1) * t ** (j - 2)
        b[row] = val; row += 1
    return np.linalg.solve(A, b)


def foot_coeff_trajectory(current_pos, des_pos, step_height, T):
    """footCoeffTrajectory: x, y 5th order (6 coefficients), z 7th order (8). Returns ([3,8], [3])."""
    cur = np.asarray(current_pos, dtype=np.float64); des = np.asarray(des_pos, dtype=np.float64)
    co = np.zeros((3, 8)); n = np.array([6, 6, 8], dtype=np.int32)
    vel2 = [(0, 0), (T, 0)]; acc2 = [(0, 0), (T, 0)]
    for ax in range(2):
        co[ax, :6] = find_poly_coeff([(0, cur[ax]), (T, des[ax])], vel2, acc2)
    co[2, :8] = find_poly_coeff([(0, cur[2]), (T / 2, step_height), (T, des[2])],
                                [(0, 0), (T / 2, 0), (T, 0)], acc2)
    return co, n


def walk_plan(simulation_time, time_step, num_steps=4, time_per_step=0.5, ds_time=0.1, step_height=0.02,
              settle_time=0.3, first_support=PHASE_RIGHT, foot_y=0.05):
    """Build-defined walking references on the reference's conventions (no reference semantics: the
    reference declares ZMP(Task, numSteps, timePerStep, simulationTime) / walkZMP but defines neither).

    x quantities are in units of the step length: the per-instance scale (set_xscale) turns them into
    metres on the device.  Returns dict(zmp_x, zmp_y, phase, segs[n_seg,52], seg_of_sample):
      * phase[k]: PHASE_DOUBLE / PHASE_RIGHT (right foot supports, left swings) / PHASE_LEFT;
      * ZMP: support-foot position in single support, mid-point of the feet in double support;
      * segment g = t0 | rF[3][8] | lF[3][8]: constant polynomials while a foot stands, the
        footCoeffTrajectory() polynomials (5th order x/y, 7th order z through step_height) while it swings,
        evaluated at t - t0.
    Sample grid and length follow ZMP::stanceZMP: int((T + 0.5) / dt) samples, sample k <-> t = k dt."""
    n = int((simulation_time + 0.5) / time_step)
    zx = np.zeros(n); zy = np.zeros(n); ph = np.full(n, PHASE_DOUBLE, dtype=np.uint8)
    sos = np.zeros(n, dtype=np.uint16)
    segs = []

    def hold(t0, xr, xl):
        g = np.zeros(52); g[0] = t0
        g[1 + 0] = xr; g[1 + 8] = -foot_y            # rF x, y constants (z = 0)
        g[1 + 24 + 0] = xl; g[1 + 24 + 8] = foot_y   # lF
        return g

    def idx(t):
        return min(n, max(0, int(round(t / time_step))))

    xr = xl = 0.0
    segs.append(hold(0.0, xr, xl))
    cur = 0
    sup = first_support
    t = settle_time
    for s in range(num_steps):
        stride = 1.0 if (s == 0 or s == num_steps - 1) else 2.0
        t_ss0, t_ss1 = t + ds_time, t + time_per_step
        a, b, c = idx(t), idx(t_ss0), idx(t_ss1)
        # double support [a, b): hold segment, ZMP at the mid-point
        segs.append(hold(t, xr, xl)); cur = len(segs) - 1
        sos[a:b] = cur; zx[a:b] = 0.5 * (xr + xl); zy[a:b] = 0.0
        # single support [b, c)
        T = (c - b) * time_step
        g = hold(b * time_step, xr, xl)
        if sup == PHASE_RIGHT:                        # left foot swings
            co, _ = foot_coeff_trajectory([xl, foot_y, 0.0], [xl + stride, foot_y, 0.0], step_height, T)
            g[1 + 24:1 + 48] = co.reshape(-1); zx[b:c] = xr; zy[b:c] = -foot_y; xl += stride
        else:
            co, _ = foot_coeff_trajectory([xr, -foot_y, 0.0], [xr + stride, -foot_y, 0.0], step_height, T)
            g[1:1 + 24] = co.reshape(-1); zx[b:c] = xl; zy[b:c] = foot_y; xr += stride
        segs.append(g); cur = len(segs) - 1
        sos[b:c] = cur; ph[b:c] = sup
        sup = PHASE_LEFT if sup == PHASE_RIGHT else PHASE_RIGHT
        t += time_per_step
    a = idx(t)
    segs.append(hold(t, xr, xl)); cur = len(segs) - 1
    sos[a:] = cur; zx[a:] = 0.5 * (xr + xl); zy[a:] = 0.0
    return dict(zmp_x=zx, zmp_y=zy, phase=ph, segs=np.array(segs), seg_of_sample=sos)


def jump_plan(simulation_time, time_step, stance_time=0.4, flight_time=0.15):
    """Build-defined jumping contact schedule (BASELINE config 5; the reference only hints at it: "0 reaction
    variables" in the comment at controller.hpp:98): double support for stance_time, PHASE_FLIGHT for
    flight_time (both feet forced out of the QP), double support afterwards.  ZMP references stay at the
    stance values of ZMP::stanceZMP(Double); the feet keep their constant polynomials.
    Returns dict(zmp_x, zmp_y, phase) on the ZMP::stanceZMP sample grid."""
    zx, zy = stance_zmp(simulation_time, time_step, 2)
    n = len(zx)
    ph = np.full(n, PHASE_DOUBLE, dtype=np.uint8)
    a = min(n, int(round(stance_time / time_step)))
    b = min(n, int(round((stance_time + flight_time) / time_step)))
    ph[a:b] = PHASE_FLIGHT
    return dict(zmp_x=zx, zmp_y=zy, phase=ph)


__all__ = ["stance_zmp", "find_poly_coeff", "foot_coeff_trajectory", "walk_plan", "jump_plan",
           "PHASE_DOUBLE", "PHASE_RIGHT", "PHASE_LEFT", "PHASE_FLIGHT"]
