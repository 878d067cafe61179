"""Host-side mirror of the reference's controller interface for B robots at once.

Names and argument meaning follow the reference classes (Controller::standStep / WBC,
Kinematics::compute, ZMP, footCoeffTrajectory, Mpc3dLip) so that tests read like calls into
the reference; every numeric result comes from the HIP kernels behind include/lmh.h.
torch is used only for device memory and streams.
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .capi import LmhConfig, check


def default_config(dt=0.01, time_horizon=0.5, z_com=0.26, **overrides):
    """lmh_config with the reference literals (controller.hpp:80-124, mpcLinearPendulum.hpp:43-49).

    dt is the control step (Clock); mpc_dt=... (an override) is the MPC sample time / reference sample period
    (Mpc3dLip and ZMP constructor arguments, apps/offline/main.cpp:21,39); 0 or absent = dt."""
    cfg = LmhConfig()
    capi.lib().lmh_config_default(C.byref(cfg))
    cfg.dt, cfg.time_horizon, cfg.z_com = dt, time_horizon, z_com
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(f"lmh_config has no field {k}")
        setattr(cfg, k, v)
    return cfg


def nominal_links():
    """createNaoParameters() table, [28][13] (mass | com | inertia), Aldebaran convention."""
    raw = np.zeros((28, 13), dtype=np.float64)
    capi.lib().lmh_nominal_links(raw.ctypes.data_as(C.c_void_p))
    return raw


def initial_configuration():
    """initialConfiguration() of the reference (src/Robot.cpp:242-251): posture the IK starts from."""
    return np.array([-0.0185, 0, 0.282, 0, 0, 0,
                     0, 0, -0.5, 0.8, -0.3, 0,
                     0, 0, -0.5, 0.8, -0.3, 0,
                     1.6, 0, 0, 0, 0,
                     -1.6, 0, 0, 0, 0,
                     0, 0], dtype=np.float64)


def ik_start_posture(device=0, com_target=(-0.02, 0.0, 0.26)):
    """apps/offline/main.cpp:24-39 on the GPU: IK to feet (0,-/+0.05,0) and the CoM target; returns (q[30], z_com)
    where z_com = Robot::getCoM()(2) after the IK (the Mpc3dLip constructor argument)."""
    ctl = BatchedController(1, default_config(), device=device)
    q = torch.as_tensor(initial_configuration()[None, :]).to(ctl.device)
    q, iters = ctl.ik(q, com_target=com_target)
    com = ctl.robot_com(q)
    torch.cuda.synchronize(ctl.device)
    out = q.cpu().numpy()[0].copy(), float(com.cpu().numpy()[0, 2])
    ctl.close()
    return out


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _dev_ptr(t):
    return C.c_void_p(t.data_ptr())


class BatchedController:
    """B independent Robot+Mpc3dLip+Controller triples resident on one GPU.

    reference: Controller(Robot&, Mpc3dLip&, ZMP&, rFCoeff, lFCoeff) (controller.hpp:52-57).
    """

    def __init__(self, n_instances, config=None, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedController needs a HIP device (no CPU fallback)")
        self.B = int(n_instances)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self.cfg = config if config is not None else default_config()
        self._h = C.c_void_p()
        check(capi.lib().lmh_create(C.byref(self.cfg), self.B, self.device_index, C.byref(self._h)))
        self.N = capi.lib().lmh_horizon(self._h)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            capi.lib().lmh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ set-up
    def set_model(self, raw_links=None):
        """createNaoParameters + Robot ctor re-expression; raw_links [28,13] or [B,28,13]."""
        if raw_links is None:
            check(capi.lib().lmh_set_model(self._h, None, 1))
            return
        raw = np.ascontiguousarray(raw_links, dtype=np.float64)
        n = 1 if raw.ndim == 2 else raw.shape[0]
        check(capi.lib().lmh_set_model(self._h, _np_ptr(raw), n))

    def mass(self):
        out = np.zeros(self.B, dtype=np.float64)
        check(capi.lib().lmh_get_mass(self._h, _np_ptr(out)))
        return out

    def set_refs(self, zmp_x, zmp_y, phase=None):
        """ZMP reference arrays (ZMP::getZmpXRef/YRef) + optional per-sample support phase."""
        zx = np.ascontiguousarray(zmp_x, dtype=np.float64)
        zy = np.ascontiguousarray(zmp_y, dtype=np.float64)
        ph = None if phase is None else np.ascontiguousarray(phase, dtype=np.uint8)
        check(capi.lib().lmh_set_refs(self._h, _np_ptr(zx), _np_ptr(zy), None if ph is None else _np_ptr(ph), len(zx)))

    def set_refs_stance(self, simulation_time, support_foot=2):
        """ZMP(Task::Stand, simulationTime, timeStep, supportFoot) (zmpGeneration.cpp:12-23,39-60)."""
        check(capi.lib().lmh_set_refs_stance(self._h, float(simulation_time), int(support_foot)))

    def set_foot_coeffs(self, r_coeff, r_n, l_coeff, l_n):
        r = np.zeros((3, 8)); l = np.zeros((3, 8))
        r[:, :np.shape(r_coeff)[1]] = r_coeff
        l[:, :np.shape(l_coeff)[1]] = l_coeff
        rn = np.ascontiguousarray(r_n, dtype=np.int32); ln = np.ascontiguousarray(l_n, dtype=np.int32)
        check(capi.lib().lmh_set_foot_coeffs(self._h, _np_ptr(r), _np_ptr(rn), _np_ptr(l), _np_ptr(ln)))

    def set_segments(self, segs, seg_of_sample):
        """Walking extension: piecewise foot polynomials [n_seg,52] selected by seg_of_sample[k] (uint16)."""
        if segs is None:
            check(capi.lib().lmh_set_segments(self._h, None, 0, None, 0))
            return
        sg = np.ascontiguousarray(segs, dtype=np.float64)
        so = np.ascontiguousarray(seg_of_sample, dtype=np.uint16)
        check(capi.lib().lmh_set_segments(self._h, _np_ptr(sg), sg.shape[0], _np_ptr(so), len(so)))

    def gen_walk(self, simulation_time, num_steps=4, time_per_step=0.5, ds_time=0.1, step_height=0.02, settle_time=0.3,
                 first_support=capi.PHASE_RIGHT, foot_y=0.05):
        """lmh_gen_walk: the walking plan of trajectories.walk_plan generated by a device kernel (ZMP samples, support phase, swing
        polynomial segments); nothing is uploaded."""
        check(capi.lib().lmh_gen_walk(self._h, float(simulation_time), int(num_steps), float(time_per_step), float(ds_time), float(step_height),
                                      float(settle_time), int(first_support), float(foot_y)))

    def gen_jump(self, simulation_time, stance_time=0.4, flight_time=0.15):
        """lmh_gen_jump: stance references with a flight phase (trajectories.jump_plan), generated on the device."""
        check(capi.lib().lmh_gen_jump(self._h, float(simulation_time), float(stance_time), float(flight_time)))

    def get_refs(self):
        """The reference set the handle currently holds, read back from the device: dict(zmp_x, zmp_y, phase, segs, seg_of_sample)."""
        n, ns = capi.lib().lmh_num_ref_samples(self._h), capi.lib().lmh_num_segments(self._h)
        zx, zy = np.zeros(n), np.zeros(n)
        ph = np.zeros(n, dtype=np.uint8)
        segs = np.zeros((ns, capi.SEG_STRIDE)); sos = np.zeros(n, dtype=np.uint16)
        check(capi.lib().lmh_get_refs(self._h, _np_ptr(zx), _np_ptr(zy), _np_ptr(ph), _np_ptr(segs) if ns else None, _np_ptr(sos) if ns else None))
        return dict(zmp_x=zx, zmp_y=zy, phase=ph, segs=segs, seg_of_sample=sos)

    def set_xscale(self, xscale):
        """Per-instance step-length scale of ZMP x and x-axis foot polynomials ([B] or None)."""
        if xscale is None:
            check(capi.lib().lmh_set_xscale(self._h, None, 0))
            return
        xs = np.ascontiguousarray(xscale, dtype=np.float64)
        check(capi.lib().lmh_set_xscale(self._h, _np_ptr(xs), len(xs)))

    def set_zcom(self, z_com):
        z = np.atleast_1d(np.ascontiguousarray(z_com, dtype=np.float64))
        check(capi.lib().lmh_set_zcom(self._h, _np_ptr(z), len(z)))

    def mpc_gain(self):
        K = np.zeros(self.N + 1, dtype=np.float64)
        check(capi.lib().lmh_get_mpc_gain(self._h, _np_ptr(K)))
        return K

    # ------------------------------------------------------------------ buffers
    def new_state(self, q, v, t=0.0, v_prev=None):
        """state records [B,96]: q | v | v_prev (Robot::v_, zeros after construction) | t."""
        st = torch.zeros((self.B, capi.STATE_STRIDE), dtype=torch.float64)
        st[:, 0:30] = torch.as_tensor(np.broadcast_to(np.asarray(q, dtype=np.float64), (self.B, 30)).copy())
        st[:, 30:60] = torch.as_tensor(np.broadcast_to(np.asarray(v, dtype=np.float64), (self.B, 30)).copy())
        if v_prev is not None:
            st[:, 60:90] = torch.as_tensor(np.broadcast_to(np.asarray(v_prev, dtype=np.float64), (self.B, 30)).copy())
        st[:, 90] = torch.as_tensor(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.B,)).copy())
        return st.to(self.device)

    def new_out(self):
        return torch.zeros((self.B, capi.OUT_STRIDE), dtype=torch.float64, device=self.device)

    def new_status(self):
        return torch.zeros((self.B, capi.STATUS_STRIDE), dtype=torch.int32, device=self.device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ hot path
    def stand_step(self, state, out=None, status=None, debug=False):
        """Controller::standStep(ControllerInput{q,dq,time}) for all instances (controller.cpp:48-79).

        Updates state[:, 60:90] (Robot::v_) in place like the reference mutates its Robot."""
        out = self.new_out() if out is None else out
        status = self.new_status() if status is None else status
        if debug:
            dbg = torch.zeros((self.B, capi.DEBUG_STRIDE), dtype=torch.float64, device=self.device)
            check(capi.lib().lmh_eval_debug(self._h, _dev_ptr(state), _dev_ptr(out), _dev_ptr(status), _dev_ptr(dbg), self._stream()))
            return out, status, dbg
        check(capi.lib().lmh_eval(self._h, _dev_ptr(state), _dev_ptr(out), _dev_ptr(status), self._stream()))
        return out, status

    def rollout(self, state, n_ticks, out=None, status=None, log=False):
        """n_ticks of rk4Step(dynamics) + Clock::step (apps/offline/main.cpp:66-122), fused on chip."""
        out = self.new_out() if out is None else out
        status = self.new_status() if status is None else status
        lg = torch.zeros((n_ticks, self.B, 36), dtype=torch.float64, device=self.device) if log is True else (log if log is not False else None)
        check(capi.lib().lmh_rollout(self._h, _dev_ptr(state), _dev_ptr(out), _dev_ptr(status),
                                     None if lg is None else _dev_ptr(lg), int(n_ticks), self._stream()))
        return out, status, lg

    def synchronize(self):
        """lmh_synchronize on the current stream: waits for it, and raises LmhError (code ERR_UNFINISHED) if a completed rollout of this
        handle left robots part-way (they carry FLAG_UNFINISHED in their status records)."""
        check(capi.lib().lmh_synchronize(self._h, self._stream()))

    def ik(self, q, com_target=(-0.02, 0.0, 0.26), rf=(0, -0.05, 0, 0, 0, 0), lf=(0, 0.05, 0, 0, 0, 0)):
        """Kinematics::desiredOperationalState + compute (invKinematics.cpp:11-52); q [B,30] device tensor, in place."""
        iters = torch.zeros((self.B,), dtype=torch.int32, device=self.device)
        ct = np.ascontiguousarray(com_target, dtype=np.float64)
        r6 = np.ascontiguousarray(rf, dtype=np.float64); l6 = np.ascontiguousarray(lf, dtype=np.float64)
        check(capi.lib().lmh_ik(self._h, _dev_ptr(q), _np_ptr(ct), _np_ptr(r6), _np_ptr(l6), _dev_ptr(iters), self._stream()))
        return q, iters

    def robot_com(self, q):
        """Robot::updateState + getCoM (Robot.cpp:264-269,225-238) for q [B,30] (device tensor) -> [B,3]."""
        com = torch.zeros((self.B, 3), dtype=torch.float64, device=self.device)
        check(capi.lib().lmh_robot_com(self._h, _dev_ptr(q), _dev_ptr(com), self._stream()))
        return com

    def make_summary(self, state, out, status):
        """End-of-run summary [B,16] (include/lmh.h lmh_make_summary): the record the RCCL gather moves."""
        s = torch.empty((self.B, capi.SUMMARY_WIDTH), dtype=torch.float64, device=self.device)
        check(capi.lib().lmh_make_summary(self._h, _dev_ptr(state), _dev_ptr(out), _dev_ptr(status), _dev_ptr(s), self._stream()))
        return s

    @staticmethod
    def write_summary(path, summary, dt=0.0):
        """lmh_write_summary: [n,16] host array -> 64-byte header + raw f64 file."""
        a = np.ascontiguousarray(summary, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != capi.SUMMARY_WIDTH:
            raise ValueError("summary must be [n,16]")
        check(capi.lib().lmh_write_summary(str(path).encode(), _np_ptr(a), a.shape[0], float(dt)))

    @staticmethod
    def read_summary(path):
        n, dt = C.c_uint64(0), C.c_double(0.0)
        check(capi.lib().lmh_read_summary(str(path).encode(), None, 0, C.byref(n), C.byref(dt)))      # header only
        a = np.zeros((n.value, capi.SUMMARY_WIDTH), dtype=np.float64)
        check(capi.lib().lmh_read_summary(str(path).encode(), _np_ptr(a), a.size, C.byref(n), C.byref(dt)))
        return a, dt.value

    @staticmethod
    def write_log(path, log, dt, t0=0.0):
        """lmh_write_log: [n_ticks,B,36] host array (lmh_rollout's d_log copied back)."""
        a = np.ascontiguousarray(log, dtype=np.float64)
        if a.ndim != 3 or a.shape[2] != 36:
            raise ValueError("log must be [ticks,B,36]")
        check(capi.lib().lmh_write_log(str(path).encode(), _np_ptr(a), a.shape[0], a.shape[1], float(dt), float(t0)))

    @staticmethod
    def read_log(path):
        nt, n, dt, t0 = C.c_uint64(0), C.c_uint64(0), C.c_double(0.0), C.c_double(0.0)
        check(capi.lib().lmh_read_log(str(path).encode(), None, 0, C.byref(nt), C.byref(n), C.byref(dt), C.byref(t0)))
        a = np.zeros((nt.value, n.value, 36), dtype=np.float64)
        check(capi.lib().lmh_read_log(str(path).encode(), _np_ptr(a), a.size, C.byref(nt), C.byref(n), C.byref(dt), C.byref(t0)))
        return a, dt.value, t0.value

    @staticmethod
    def split_out(out):
        """WBCOutput fields: tau [B,24], f [B,12] (n_R f_R n_L f_L), qpp [B,30]."""
        return out[:, 0:24], out[:, 24:36], out[:, 36:66]


# debug-dump layout (LMH_DEBUG_STRIDE record, see controller_eval in lmh_kernels.hip)
DEBUG_FIELDS = {
    "T": (0, (28, 3, 4)), "XE": (336, (28, 3, 3)), "Xp": (588, (28, 3)), "XB": (672, (28, 3, 3)),
    "C": (924, (30,)), "Cg6": (954, (6,)), "Mtop": (960, (6, 30)), "Hl": (1140, (24, 6)),
    "AG": (1284, (6, 30)), "AGpqp": (1464, (6,)), "Jpqp": (1470, (12,)), "Jc": (1482, (2, 6, 12)),
    "CoM": (1626, (3,)), "comVel": (1629, (3,)), "angMom": (1632, (3,)), "mpc": (1635, (8,)),
    "qppRef": (1643, (30,)), "hGpRef": (1673, (6,)), "footAccRef": (1679, (12,)),
    "Y": (1691, (30, 7)), "Si": (1901, (6, 6)), "W": (1937, (12, 12)), "h12": (2081, (12,)),
    "P": (2093, (32, 32)), "qv": (3117, (32,)), "c": (3149, (32,)), "a": (3181, (30,)),
}


def unpack_debug(dbg_row):
    """numpy views of one instance's debug record."""
    d = np.asarray(dbg_row)
    return {k: d[o:o + int(np.prod(s))].reshape(s) for k, (o, s) in DEBUG_FIELDS.items()}
