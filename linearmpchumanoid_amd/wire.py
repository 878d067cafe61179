"""On-disk formats either side of the hot path, and the MuJoCo-side adapter.

The reference writes nothing but stdout (apps/offline/main.cpp:86 prints CoM x per tick), so the two
record formats below are the build's own; they are plain little-endian arrays behind a fixed 64-byte
header so that a C reader is a struct and an fread:

    offset  0  char[8]   magic   "LMHSUM1\0" (run summary) | "LMHLOG1\0" (trajectory log)
            8  uint32    version (1)
           12  uint32    dtype code (1 = float64)
           16  uint64    n_instances
           24  uint64    n_ticks          (log; 0 for a summary)
           32  uint32    record width in doubles (16 summary | 36 log: tau(24) | f(12))
           36  uint32    reserved
           40  float64   dt
           48  float64   t0  (clock at the first logged tick)
           56  uint64    reserved
    payload: summary [n_instances][16] f64 (sharding.make_summary layout);
             log     [n_ticks][n_instances][36] f64 (exactly the d_log buffer of lmh_rollout).

The MuJoCo-side adapter restates the three small conversions the reference does between MuJoCo's
generalized coordinates and the controller's (simulators/mujoco/MujocoSim.cpp:119-146,
apps/mujoco/main.cpp:182-200); it needs no MuJoCo: it works on the qpos / qvel / ctrl arrays.
"""
import struct

import numpy as np

HEADER = struct.Struct("<8sIIQQIIddQ")
assert HEADER.size == 64
MAGIC_SUMMARY = b"LMHSUM1\0"
MAGIC_LOG = b"LMHLOG1\0"
SUMMARY_WIDTH = 16
LOG_WIDTH = 36
SUMMARY_FIELDS = ("base_x", "base_y", "base_z", "roll", "pitch", "yaw", "t", "max_abs_tau", "sum_fz", "fz_right", "fz_left",
                  "k", "qp_iterations", "flags", "active_count", "state_checksum")


def _write(path, magic, arr, n_inst, n_ticks, width, dt, t0):
    a = np.ascontiguousarray(arr, dtype="<f8")
    with open(path, "wb") as f:
        f.write(HEADER.pack(magic, 1, 1, n_inst, n_ticks, width, 0, float(dt), float(t0), 0))
        f.write(a.tobytes())


def _read(path, magic):
    with open(path, "rb") as f:
        head = f.read(HEADER.size)
        if len(head) != HEADER.size:
            raise ValueError(f"{path}: truncated header")
        mg, ver, dtype, n_inst, n_ticks, width, _, dt, t0, _ = HEADER.unpack(head)
        if mg != magic:
            raise ValueError(f"{path}: bad magic {mg!r}")
        if ver != 1 or dtype != 1:
            raise ValueError(f"{path}: unsupported version/dtype {ver}/{dtype}")
        count = n_inst * width * (n_ticks if magic == MAGIC_LOG else 1)
        data = np.frombuffer(f.read(), dtype="<f8")
        if data.size != count:
            raise ValueError(f"{path}: payload holds {data.size} doubles, header says {count}")
    return data, n_inst, n_ticks, width, dt, t0


def write_summary(path, summary, dt=0.0):
    """summary: [B,16] (sharding.make_summary / the gathered end-of-run table)."""
    s = np.asarray(summary, dtype=np.float64)
    if s.ndim != 2 or s.shape[1] != SUMMARY_WIDTH:
        raise ValueError("summary must be [B,16]")
    _write(path, MAGIC_SUMMARY, s, s.shape[0], 0, SUMMARY_WIDTH, dt, 0.0)


def read_summary(path):
    data, n, _, w, dt, _ = _read(path, MAGIC_SUMMARY)
    return data.reshape(n, w).copy(), dt


def write_log(path, log, dt, t0=0.0):
    """log: [n_ticks, B, 36] -- tau | f of the k4 stage of every tick (lmh_rollout's d_log)."""
    lg = np.asarray(log, dtype=np.float64)
    if lg.ndim != 3 or lg.shape[2] != LOG_WIDTH:
        raise ValueError("log must be [ticks,B,36]")
    _write(path, MAGIC_LOG, lg, lg.shape[1], lg.shape[0], LOG_WIDTH, dt, t0)


def read_log(path):
    data, n, nt, w, dt, t0 = _read(path, MAGIC_LOG)
    return data.reshape(nt, n, w).copy(), dt, t0


# ----------------------------------------------------------------------------- MuJoCo-side adapter
N_ACTUATED = 24


def mujoco_relabel_matrix():
    """relabelMujocoMatrix (apps/mujoco/main.cpp:182-200): L[mujoco joint, controller joint].

    MuJoCo (models/nao.xml) orders the actuated joints head(2), left leg(6), right leg(6), left arm(5),
    right arm(5); the controller orders them right leg, left leg, right arm, left arm, head
    (Robot.cpp:244-249).  The reference leaves the other entries uninitialised; they are
    zero here.  Entry (14, 17) is -1: LShoulderPitch has the opposite sign convention (main.cpp:196)."""
    L = np.zeros((N_ACTUATED, N_ACTUATED))
    L[0:2, 22:24] = np.eye(2)            # head
    L[2:8, 6:12] = np.eye(6)             # left leg
    L[8:14, 0:6] = np.eye(6)             # right leg
    L[14:19, 17:22] = np.eye(5)          # left arm
    L[14, 17] = -1.0
    L[19:24, 12:17] = np.eye(5)          # right arm
    return L


def controller_joints_from_mujoco(q_mj):
    """relabelJoints = L' * qpos[7:31] (apps/mujoco/main.cpp:64-65): MuJoCo joint order -> controller order."""
    q = np.asarray(q_mj, dtype=np.float64)
    return q @ mujoco_relabel_matrix()


def mujoco_joints_from_controller(q_ctl):
    q = np.asarray(q_ctl, dtype=np.float64)
    return q @ mujoco_relabel_matrix().T


def controller_input_from_mujoco(qpos, qvel):
    """MujocoSim::getControllerInput (simulators/mujoco/MujocoSim.cpp:119-137): q = qpos[7:], dq = qvel[6:]
    (floating base: 7 position + 6 velocity coordinates are skipped, joints are NOT relabelled there)."""
    qpos = np.asarray(qpos, dtype=np.float64); qvel = np.asarray(qvel, dtype=np.float64)
    if qpos.shape[-1] < 7 or qvel.shape[-1] < 6 or qpos.shape[-1] - 7 != qvel.shape[-1] - 6:
        raise ValueError("qpos must be [7 base | joints], qvel [6 base | joints]")
    return qpos[..., 7:].copy(), qvel[..., 6:].copy()


def apply_torques(ctrl, tau):
    """MujocoSim::applyTorques (MujocoSim.cpp:139-146): ctrl <- tau, size must equal the actuator count."""
    tau = np.asarray(tau, dtype=np.float64)
    if tau.shape[-1] != ctrl.shape[-1]:
        raise RuntimeError("applyTorques(): tau dimension mismatch")      # the reference throws std::runtime_error
    ctrl[...] = tau
    return ctrl
