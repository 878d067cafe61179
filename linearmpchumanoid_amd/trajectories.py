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


def walk_refs(simulation_time, time_step, num_steps=4, time_per_step=0.5, step_length=0.03,
              ds_time=0.1, first_support=PHASE_RIGHT, settle_time=0.5):
    """Build-defined walking references (no reference semantics beyond the conventions).

    Piecewise-constant ZMP: y = -0.05 over the right foot, +0.05 over the left foot, 0 in
    double support; x advances by step_length per single-support phase.  Returns
    (zmp_x, zmp_y, phase) sampled like ZMP::stanceZMP (int((T+0.5)/dt) samples)."""
    samples = int((simulation_time + 0.5) / time_step)
    zx = np.zeros(samples); zy = np.zeros(samples); ph = np.full(samples, PHASE_DOUBLE, dtype=np.uint8)
    t0 = settle_time
    x = 0.0
    sup = first_support
    for s in range(num_steps):
        a = int(round((t0 + s * time_per_step + ds_time) / time_step))
        b = int(round((t0 + (s + 1) * time_per_step) / time_step))
        a, b = min(a, samples), min(b, samples)
        ph[a:b] = sup
        zy[a:b] = -0.05 if sup == PHASE_RIGHT else 0.05
        zx[a:b] = x
        d0 = int(round((t0 + s * time_per_step) / time_step))
        zx[min(d0, samples):a] = x
        x += step_length
        sup = PHASE_LEFT if sup == PHASE_RIGHT else PHASE_RIGHT
    end = int(round((t0 + num_steps * time_per_step) / time_step))
    zx[min(end, samples):] = x - step_length
    return zx, zy, ph


__all__ = ["stance_zmp", "find_poly_coeff", "foot_coeff_trajectory", "walk_refs",
           "PHASE_DOUBLE", "PHASE_RIGHT", "PHASE_LEFT", "PHASE_FLIGHT"]
