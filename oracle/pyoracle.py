"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE, not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/lmh_oracle.h).  It builds oracle/_build/liblmh_oracle.so on demand with gcc.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblmh_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liblmh_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_sys_create.restype = C.c_void_p
        _lib.orc_sys_create.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
        _lib.orc_sys_create_model.restype = C.c_void_p
        _lib.orc_sys_create_model.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
        _lib.orc_sys_mass.restype = C.c_double
        _lib.orc_sys_zcom.restype = C.c_double
        _lib.orc_batch_rollout.restype = C.c_double
        _lib.orc_sys_set_zcom.argtypes = [C.c_void_p, C.c_double]
        _lib.orc_sys_set_gains.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_sys_get_gains.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_batch_rollout_ex.restype = C.c_double
        for name in ("orc_sys_destroy", "orc_sys_mass", "orc_sys_horizon", "orc_sys_zcom", "orc_sys_nzmp"):
            getattr(_lib, name).argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(*shape):
    return np.zeros(shape, dtype=np.float64)


class Oracle:
    """One robot + controller, mirroring the object graph of apps/offline/main.cpp.

    `dt` is what main.cpp hands to ZMP (:21) and Mpc3dLip (:39): the MPC sample time.  The Clock's step (:18) only enters
    through rollout(..., dt=...), so a control step different from the MPC sample time is rollout(dt=control_dt) on an
    Oracle(dt=mpc_dt)."""

    def __init__(self, sim_time=5.0, dt=0.01, horizon_time=0.5, do_ik=True, raw_links=None):
        L = lib()
        raw = None if raw_links is None else np.ascontiguousarray(raw_links, dtype=np.float64)
        self._h = C.c_void_p(L.orc_sys_create_model(sim_time, dt, horizon_time, int(do_ik), _p(raw)))
        self.dt = dt
        self.sim_time = sim_time
        self.horizon_time = horizon_time

    def close(self):
        if self._h:
            lib().orc_sys_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scalars
    @property
    def mass(self):
        return lib().orc_sys_mass(self._h)

    @property
    def horizon(self):
        return lib().orc_sys_horizon(self._h)

    @property
    def zcom(self):
        return lib().orc_sys_zcom(self._h)

    @property
    def n_zmp(self):
        return lib().orc_sys_nzmp(self._h)

    # ---- state
    def robot(self):
        q, v, c, cv, am = _f64(30), _f64(30), _f64(3), _f64(3), _f64(3)
        lib().orc_sys_get_robot(self._h, _p(q), _p(v), _p(c), _p(cv), _p(am))
        return dict(q=q, v=v, CoM=c, comVel=cv, angMom=am)

    def links(self):
        out = _f64(28, 13)
        lib().orc_sys_get_links(self._h, _p(out))
        return out

    def set_prev_velocity(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        lib().orc_sys_set_prev_velocity(self._h, _p(v))

    def set_q(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        lib().orc_sys_set_q(self._h, _p(q))

    def set_zcom(self, z):
        lib().orc_sys_set_zcom(self._h, C.c_double(z))

    GAIN_FIELDS = ("mu", "kp_joints", "kd_joints", "kp_mom", "kd_mom", "kp_feet", "kd_feet",
                   "w_com_lin", "w_com_ang", "w_base_pos", "w_base_ang", "w_joints", "w_force", "w_foot", "eps_coeff")

    def gains(self):
        g = _f64(15)
        lib().orc_sys_get_gains(self._h, _p(g))
        return dict(zip(self.GAIN_FIELDS, g.tolist()))

    def set_gains(self, **kw):
        """controller.hpp:80-124 literals (lmh_config field names); unnamed ones keep their value."""
        cur = self.gains()
        for k, v in kw.items():
            if k not in cur:
                raise KeyError(k)
            cur[k] = float(v)
        g = np.array([cur[k] for k in self.GAIN_FIELDS], dtype=np.float64)
        lib().orc_sys_set_gains(self._h, _p(g))

    def set_plant(self, on=True, k=5.0e4, d=3.0e2, dt=3.0e2, mu=0.7):
        """Build-defined plant: forward dynamics driven by the WBC torques + spring-damper contact (lmh_oracle.h)."""
        lib().orc_sys_set_plant(self._h, C.c_int(int(on)), C.c_double(k), C.c_double(d), C.c_double(dt), C.c_double(mu))

    def contact(self):
        w, vf = _f64(12), _f64(8, 3)
        lib().orc_sys_contact(self._h, _p(w), _p(vf))
        return w, vf

    def set_segments(self, segs, seg_of_sample, xscale=1.0):
        segs = np.ascontiguousarray(segs, dtype=np.float64)
        sos = np.ascontiguousarray(seg_of_sample, dtype=np.uint16)
        assert len(sos) == self.n_zmp and segs.shape[1] == 52
        lib().orc_sys_set_segments(self._h, C.c_int(segs.shape[0]), _p(segs), _p(sos), C.c_double(xscale))

    def set_wbc_calls(self, n, faithful=False):
        lib().orc_sys_set_wbc_calls(self._h, int(n), int(faithful))

    def set_refs(self, zx, zy, phase=None):
        zx = np.ascontiguousarray(zx, dtype=np.float64)
        zy = np.ascontiguousarray(zy, dtype=np.float64)
        ph = None if phase is None else np.ascontiguousarray(phase, dtype=np.uint8)
        lib().orc_sys_set_refs(self._h, C.c_int(len(zx)), _p(zx), _p(zy), _p(ph))

    def zmp(self):
        n = self.n_zmp
        zx, zy = _f64(n), _f64(n)
        lib().orc_sys_get_zmp(self._h, _p(zx), _p(zy))
        return zx, zy

    def foot_coeffs(self):
        rF, lF = _f64(3, 8), _f64(3, 8)
        rn, ln = np.zeros(3, np.int32), np.zeros(3, np.int32)
        lib().orc_sys_get_foot_coeffs(self._h, _p(rF), _p(rn), _p(lF), _p(ln))
        return rF, rn, lF, ln

    def set_foot_coeffs(self, rF, rn, lF, ln):
        rF = np.ascontiguousarray(rF, dtype=np.float64)
        lF = np.ascontiguousarray(lF, dtype=np.float64)
        rn = np.ascontiguousarray(rn, dtype=np.int32)
        ln = np.ascontiguousarray(ln, dtype=np.int32)
        lib().orc_sys_set_foot_coeffs(self._h, _p(rF), _p(rn), _p(lF), _p(ln))

    def gain_row(self):
        K = _f64(self.horizon + 1)
        lib().orc_sys_gain_row(self._h, _p(K))
        return K

    def mpc_mats(self):
        n = self.horizon + 1
        Px, Pu = _f64(n, 2), _f64(n, n)
        lib().orc_sys_mpc_mats(self._h, _p(Px), _p(Pu))
        return Px, Pu

    # ---- evaluation
    def eval(self, q, dq, t):
        q = np.ascontiguousarray(q, dtype=np.float64)
        dq = np.ascontiguousarray(dq, dtype=np.float64)
        tau, f, qpp = _f64(24), _f64(12), _f64(30)
        info = np.zeros(8, np.int32)
        lib().orc_sys_eval(self._h, _p(q), _p(dq), C.c_double(t), _p(tau), _p(f), _p(qpp), _p(info))
        return dict(tau=tau, f=f, qpp=qpp, k=int(info[0]), phase=int(info[1]), qp_iters=int(info[2]),
                    qp_status=int(info[3]), active_mask=int(np.uint32(info[4])))

    def terms(self):
        T, X = _f64(28, 4, 4), _f64(28, 6, 6)
        Cv, Cg, M, AG = _f64(30), _f64(30), _f64(30, 30), _f64(6, 30)
        AGpqp, Jpqp, J = _f64(6), _f64(12), _f64(12, 30)
        lib().orc_sys_get_terms(self._h, _p(T), _p(X), _p(Cv), _p(Cg), _p(M), _p(AG), _p(AGpqp), _p(Jpqp), _p(J))
        return dict(T=T, X=X, C=Cv, Cg=Cg, M=M, AG=AG, AGpqp=AGpqp, Jpqp=Jpqp, J=J)

    def qp(self):
        H, g, A = _f64(74, 74), _f64(74), _f64(50, 74)
        lb, ub, x = _f64(50), _f64(50), _f64(74)
        qr, hr, fr, u0, mr = _f64(30), _f64(6), _f64(12), _f64(2), _f64(6)
        lib().orc_sys_get_qp(self._h, _p(H), _p(g), _p(A), _p(lb), _p(ub), _p(x), _p(qr), _p(hr), _p(fr), _p(u0), _p(mr))
        return dict(H=H, g=g, A=A, lbA=lb, ubA=ub, x=x, qppRef=qr, hGpRef=hr, footAccRef=fr, u0=u0, mpcRef=mr)

    def rollout(self, state, t, nticks, dt=None, log=False):
        dt = self.dt if dt is None else dt
        st = np.ascontiguousarray(state, dtype=np.float64).copy()
        tt = C.c_double(t)
        lg = _f64(nticks, 36) if log else None
        kl = np.zeros(nticks, np.int32)
        cx = _f64(nticks)
        info = np.zeros(8, np.int32)
        lib().orc_sys_rollout(self._h, _p(st), C.byref(tt), C.c_double(dt), C.c_int(nticks), _p(lg), _p(kl), _p(cx), _p(info))
        return dict(state=st, t=tt.value, log=lg, k=kl, comx=cx, info=info)


def nao_raw_links():
    raw = _f64(28, 13)
    lib().orc_sys_nao_raw(_p(raw))
    return raw


def batch_rollout(states, prev_v, t0, dt, nticks, sim_time, horizon_time, zcom, nthreads=1, wbc_calls=1):
    """CPU baseline: B independent closed loops, static partition over nthreads. Returns (seconds, out[B,36])."""
    st = np.ascontiguousarray(states, dtype=np.float64)
    B = st.shape[0]
    out = _f64(B, 36)
    pv = None if prev_v is None else np.ascontiguousarray(prev_v, dtype=np.float64)
    sec = lib().orc_batch_rollout(C.c_int(B), _p(st), _p(pv), _p(out), C.c_double(t0), C.c_double(dt), C.c_int(nticks),
                                  C.c_double(sim_time), C.c_double(horizon_time), C.c_double(zcom),
                                  C.c_int(nthreads), C.c_int(wbc_calls))
    return sec, st, out


def batch_rollout_ex(states, t0, dt, nticks, horizon_time, zmp_x, zmp_y, phase=None, segs=None, seg_of_sample=None,
                     xscale=None, zcom=0.26, raw_links=None, nthreads=1, wbc_calls=1, lib_override=None, mpc_dt=None):
    """CPU baseline, general form: B independent closed loops with caller-supplied references (walking / jumping
    plans), per-instance step length, LIPM height and (optionally) raw link tables.  dt is the Clock's step (RK4 step),
    mpc_dt the Mpc3dLip / ZMP sample time (apps/offline/main.cpp:18 vs :21,39; None = dt), horizon_time = N mpc_dt.
    Returns (seconds, states, out[B,36])."""
    L = lib_override if lib_override is not None else lib()
    st = np.ascontiguousarray(states, dtype=np.float64).copy()
    B = st.shape[0]
    out = _f64(B, 36)
    zx = np.ascontiguousarray(zmp_x, dtype=np.float64); zy = np.ascontiguousarray(zmp_y, dtype=np.float64)
    ph = None if phase is None else np.ascontiguousarray(phase, dtype=np.uint8)
    sg = None if segs is None else np.ascontiguousarray(segs, dtype=np.float64)
    so = None if seg_of_sample is None else np.ascontiguousarray(seg_of_sample, dtype=np.uint16)
    xs = None if xscale is None else np.ascontiguousarray(xscale, dtype=np.float64)
    zc = np.atleast_1d(np.ascontiguousarray(zcom, dtype=np.float64))
    raw = None if raw_links is None else np.ascontiguousarray(raw_links, dtype=np.float64)
    assert xs is None or len(xs) == B
    assert len(zc) in (1, B) and (raw is None or raw.shape[0] == B)
    fn = L.orc_batch_rollout_ex
    fn.restype = C.c_double
    sec = fn(C.c_int(B), _p(st), _p(out), C.c_double(t0), C.c_double(dt), C.c_int(nticks), C.c_double(horizon_time),
             C.c_int(len(zx)), _p(zx), _p(zy), _p(ph), C.c_int(0 if sg is None else sg.shape[0]), _p(sg), _p(so), _p(xs),
             C.c_int(len(zc)), _p(zc), _p(raw), C.c_int(nthreads), C.c_int(wbc_calls), C.c_double(0.0 if mpc_dt is None else mpc_dt))
    return sec, st, out
