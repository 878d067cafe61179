"""On-disk record formats and the MuJoCo-side adapter (host logic, no GPU)."""
import numpy as np
import pytest

from linearmpchumanoid_amd import wire


def test_summary_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    s = rng.normal(size=(37, 16))
    p = tmp_path / "run.lmhsum"
    wire.write_summary(p, s, dt=1e-3)
    back, dt = wire.read_summary(p)
    assert np.array_equal(back, s) and dt == 1e-3
    assert p.stat().st_size == 64 + 37 * 16 * 8
    assert len(wire.SUMMARY_FIELDS) == wire.SUMMARY_WIDTH


def test_log_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(4)
    lg = rng.normal(size=(5, 3, 36))
    p = tmp_path / "run.lmhlog"
    wire.write_log(p, lg, dt=1e-3, t0=0.25)
    back, dt, t0 = wire.read_log(p)
    assert np.array_equal(back, lg) and dt == 1e-3 and t0 == 0.25
    raw = np.frombuffer(p.read_bytes()[64:], dtype="<f8")        # a C reader sees [tick][instance][36]
    assert raw[(2 * 3 + 1) * 36 + 7] == lg[2, 1, 7]


def test_bad_files_are_rejected(tmp_path):
    p = tmp_path / "x.lmhsum"
    wire.write_summary(p, np.zeros((2, 16)))
    with pytest.raises(ValueError):
        wire.read_log(p)                                         # wrong magic
    p.write_bytes(p.read_bytes()[:-8])
    with pytest.raises(ValueError):
        wire.read_summary(p)                                     # truncated payload
    with pytest.raises(ValueError):
        wire.write_summary(tmp_path / "y", np.zeros((2, 15)))
    with pytest.raises(ValueError):
        wire.write_log(tmp_path / "z", np.zeros((2, 3, 35)), 1e-3)
    (tmp_path / "e").write_bytes(b"")
    with pytest.raises(ValueError):
        wire.read_summary(tmp_path / "e")


def test_relabel_matrix_follows_the_reference_blocks():
    """apps/mujoco/main.cpp:182-200: identity blocks head / L leg / R leg / L arm / R arm, L(14,17) = -1."""
    L = wire.mujoco_relabel_matrix()
    assert L.shape == (24, 24)
    assert np.array_equal(np.abs(L).sum(axis=0), np.ones(24)) and np.array_equal(np.abs(L).sum(axis=1), np.ones(24))   # signed permutation
    assert L[14, 17] == -1 and L.sum() == 22
    q_ctl = np.arange(24, dtype=float) + 1
    q_mj = wire.mujoco_joints_from_controller(q_ctl)
    assert list(q_mj[0:2]) == [23, 24]                           # head first in MuJoCo
    assert list(q_mj[2:8]) == [7, 8, 9, 10, 11, 12]              # then the left leg (controller joints 6..11)
    assert list(q_mj[8:14]) == [1, 2, 3, 4, 5, 6]                # right leg
    assert q_mj[14] == -18 and list(q_mj[15:19]) == [19, 20, 21, 22]
    assert list(q_mj[19:24]) == [13, 14, 15, 16, 17]
    assert np.array_equal(wire.controller_joints_from_mujoco(q_mj), q_ctl)


def test_controller_input_and_torques():
    """MujocoSim::getControllerInput / applyTorques (simulators/mujoco/MujocoSim.cpp:119-146)."""
    qpos = np.arange(31, dtype=float); qvel = np.arange(30, dtype=float) * 10
    q, dq = wire.controller_input_from_mujoco(qpos, qvel)
    assert q.shape == (24,) and dq.shape == (24,) and q[0] == 7 and dq[0] == 60
    qb, dqb = wire.controller_input_from_mujoco(np.tile(qpos, (5, 1)), np.tile(qvel, (5, 1)))      # batched
    assert qb.shape == (5, 24) and dqb.shape == (5, 24)
    with pytest.raises(ValueError):
        wire.controller_input_from_mujoco(qpos[:30], qvel)
    ctrl = np.zeros(24)
    wire.apply_torques(ctrl, np.ones(24))
    assert ctrl.sum() == 24
    with pytest.raises(RuntimeError):
        wire.apply_torques(ctrl, np.ones(23))
