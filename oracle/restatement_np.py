#!/usr/bin/env python3
"""SECOND, INDEPENDENT CPU restatement of the reference's controller evaluation (TEST INFRASTRUCTURE, not product code).

Why it exists: the reference ships no golden vectors and cannot be compiled here (Eigen3 / qpOASES absent), so the C oracle in this
directory can never be pinned to reference outputs ("parity unpinned").  What can be done is to cross-examine it: this file restates the
same path a second time, written from the reference SOURCES (file:line cited per function), not from oracle/*.c, in a different style --
dense 4x4 / 6x6 numpy algebra exactly as the Eigen code multiplies it, the literal 74x74 Hessian and 50x74 constraint matrix, and a
generic primal active-set KKT solver (numpy.linalg) instead of the oracle's Goldfarb-Idnani dual method.  The NAO inertial table is
PARSED FROM THE TEXT of /root/reference/src/robotParameters.cpp (read as data; nothing of the reference is executed), so it does not share
the oracle's hand-copied table either.  Agreement of two independent restatements shrinks the common-mode risk; it is not a pin.

Runs only where /root/reference exists (the build container).  `python oracle/restatement_np.py` checks itself against
tests/golden/eval_vectors.npz (tau, f, qdd, M, C, AG, J, ... at 1e-9) and regenerates the SURVEY 8c anchors into
tests/golden/survey_anchors.json.
"""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PI = 3.14159265358979323846                     # src/Robot.cpp:3
PARENT = [-1, 0, 1, 2, 3, 4, 5, 6, 0, 8, 9, 10, 11, 12, 13, 0, 15, 16, 17, 18, 0, 20, 21, 22, 23, 0, 25, 26]   # Robot.cpp:165
ACT = [0, 1, 2, 3, 4, 5, 6, 0, 7, 8, 9, 10, 11, 12, 0, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 0]      # Robot.cpp:172
NF, NQ = 28, 30


def available():
    return os.path.exists(os.path.join(REF, "src", "robotParameters.cpp"))


# ----------------------------------------------------------------------------- data: src/robotParameters.cpp, parsed as text
def parse_links():
    txt = open(os.path.join(REF, "src", "robotParameters.cpp")).read()
    num = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"
    links = [dict(mass=0.0, com=np.zeros(3), inertia=np.zeros((3, 3))) for _ in range(NF)]
    for m in re.finditer(r"links\[(\d+)\]\.mass\s*=\s*(" + num + r")\s*;", txt):
        links[int(m.group(1))]["mass"] = float(m.group(2))
    for m in re.finditer(r"links\[(\d+)\]\.com\s*<<\s*([^;]+);", txt):
        links[int(m.group(1))]["com"] = np.array([float(x) for x in re.findall(num, m.group(2))])
    for m in re.finditer(r"links\[(\d+)\]\.inertia\s*<<\s*([^;]+);", txt):
        links[int(m.group(1))]["inertia"] = np.array([float(x) for x in re.findall(num, m.group(2))]).reshape(3, 3)
    return links


# ----------------------------------------------------------------------------- src/generalizedFunctions.cpp
def cross_matrix(v):                              # :3-9
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=float)


def velocity_matrix(T):                           # :11-19
    X = np.zeros((6, 6))
    R = T[:3, :3].T
    p = T[:3, 3]
    X[:3, :3] = R
    X[3:, :3] = -R @ cross_matrix(p)
    X[3:, 3:] = R
    return X


def inverse_transformation(T):                    # :21-27 (R' as the inverse, also for the non-orthonormal leg frames)
    Ti = np.zeros((4, 4))
    Ti[:3, :3] = T[:3, :3].T
    Ti[:3, 3] = (-T[:3, :3].T) @ T[:3, 3]
    Ti[3, 3] = 1
    return Ti


def spatial_cross(v):                             # :29-35
    m = np.zeros((6, 6))
    m[:3, :3] = cross_matrix(v[:3]); m[3:, :3] = cross_matrix(v[3:]); m[3:, 3:] = cross_matrix(v[:3])
    return m


def spatial_cross_force(v):                       # :37-41
    return -spatial_cross(v).T


def euler_to_so3(rpy):                            # :52-72
    cr, sr, cp, sp, cy, sy = np.cos(rpy[0]), np.sin(rpy[0]), np.cos(rpy[1]), np.sin(rpy[1]), np.cos(rpy[2]), np.sin(rpy[2])
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def rot_to_axis_angle(R):                         # :74-101
    c = max(-1.0, min(1.0, (np.trace(R) - 1.0) / 2.0))
    phi = np.arccos(c)
    S = R - R.T
    v = np.array([S[2, 1], S[0, 2], S[1, 0]])
    return 0.5 * v if phi < 1e-6 else (phi / (2.0 * np.sin(phi))) * v


def polyval(c, x):                                # :165-176 (ascending powers)
    return sum(ci * x ** i for i, ci in enumerate(c))


def polyder(c):                                   # :178-193
    return np.zeros(1) if len(c) <= 1 else np.array([(i + 1) * c[i + 1] for i in range(len(c) - 1)])


def swap_base_velocity(X0, v):                    # :195-206
    v = v.copy()
    v[:3], v[3:6] = v[3:6].copy(), v[:3].copy()
    v[:6] = X0 @ v[:6]
    return v


# ----------------------------------------------------------------------------- src/Robot.cpp
def mat_trans(theta):                             # :176-223
    r = [-0.07071, 0, 0, 0, 0, 0, 0.07071, 0, 0, 0, 0, 0, 0, 0, 0.105, 0, 0.05595, 0, 0, 0.105, 0, 0.05595, 0, 0, 0]
    d = [0, 0, 0, -0.1, -0.1029, 0, 0, 0, 0, -0.1, -0.1029, 0, 0, 0, -0.015, 0, 0, 0, 0, -0.015, 0, 0, 0, 0, 0.030]
    h = PI / 2
    al = [0, h, h, 0, 0, -h, -h, -h, h, 0, 0, -h, -h, h, h, -h, h, h, h, h, -h, h, 0, -h, 0]
    out = []
    for i in range(25):
        ct, st, ca, sa = np.cos(theta[i]), np.sin(theta[i]), np.cos(al[i]), np.sin(al[i])
        out.append(np.array([[ct, -st, 0, d[i]], [ca * st, ca * ct, -sa, -r[i] * sa], [sa * st, sa * ct, ca, r[i] * ca], [0, 0, 0, 1]]))
    return out


def forward_kinematics(q):                        # :45-160
    T = [np.zeros((4, 4)) for _ in range(NF)]
    T[0][:3, 3] = q[:3]; T[0][:3, :3] = euler_to_so3(q[3:6]); T[0][3, 3] = 1
    off = {1: 0.75 * PI, 6: -0.5 * PI, 7: 0.25 * PI, 13: 0.5 * PI, 18: 0.5 * PI, 23: -0.5 * PI}
    theta = [q[6 + i] + off.get(i, 0.0) for i in range(24)] + [-PI / 2]
    Tm = mat_trans(theta)
    a01 = np.array([[0, -1, 0, 0], [0.7071, 0, 0.7071, 0], [-0.7071, 0, 0.7071, 0], [0, 0, 0, 1.0]])
    a09 = np.array([[1, 0, 0, 0], [0, 0.7071, 0.7071, 0], [0, -0.7071, 0.7071, 0], [0, 0, 0, 1.0]])
    foot = np.eye(4); foot[0, 3] = -0.0452
    T[1] = T[0] @ a01 @ Tm[0]
    for i in range(1, 6):
        T[i + 1] = T[i] @ Tm[i]
    T[7] = T[6] @ foot
    T[8] = T[0] @ a09 @ Tm[6]
    for i in range(8, 13):
        T[i + 1] = T[i] @ Tm[i - 1]
    T[14] = T[13] @ foot
    for first, tmi, offv in ((15, 12, [0, -0.098, 0.13591]), (20, 17, [0, 0.098, 0.13591]), (25, 22, [0, 0, 0.1615])):
        A = Tm[tmi].copy(); A[:3, 3] += offv
        T[first] = T[0] @ A
        for i in range(first, first + (4 if first < 25 else 2)):
            T[i + 1] = T[i] @ Tm[i - 2]
    return T


class Robot:                                      # Robot::Robot, :5-43
    def __init__(self):
        self.links = parse_links()
        self.update_state(np.zeros(NQ))
        self.mass = 0.0
        for i in range(NF):
            Rj = self.T[i][:3, :3]
            self.links[i]["com"] = Rj.T @ self.links[i]["com"]
            self.links[i]["inertia"] = Rj.T @ self.links[i]["inertia"] @ Rj
            self.mass += self.links[i]["mass"]
        self.update_state(initial_configuration())
        self.v = np.zeros(NQ)
        self.Rf_q0 = np.array([[0, 0, 1], [0, -1, 0], [1, 0, 0.0]])
        self.foot_vertices = [np.array(p) for p in ([0.1, 0.025, 0], [0.1, -0.025, 0], [-0.05, 0.025, 0], [-0.05, -0.025, 0])]

    def update_state(self, q):                    # :264-269
        self.q = np.array(q, dtype=float)
        self.T = forward_kinematics(self.q)
        if hasattr(self, "mass") and self.mass:
            com = np.zeros(3)
            for i in range(NF):                   # computeCoM, :225-238
                com = com + self.links[i]["mass"] * (self.T[i][:3, :] @ np.append(self.links[i]["com"], 1.0))
            self.CoM = com / self.mass
        piTi = [self.T[0]] + [inverse_transformation(self.T[PARENT[i]]) @ self.T[i] for i in range(1, NF)]     # :276-287
        self.X = [velocity_matrix(t) for t in piTi]                                                           # :289-298

    def update_velocity(self, v, AG):             # :271-274, 300-310
        self.v = np.array(v, dtype=float)
        h = AG @ swap_base_velocity(self.X[0], self.v)
        self.comVel = h[3:] / self.mass
        self.angMom = h[:3]


def initial_configuration():                      # Robot.cpp:242-251 (= desiredPosture, :253-262)
    return np.array([-0.0185, 0, 0.282, 0, 0, 0, 0, 0, -0.5, 0.8, -0.3, 0, 0, 0, -0.5, 0.8, -0.3, 0, 1.6, 0, 0, 0, 0, -1.6, 0, 0, 0, 0, 0, 0.0])


# ----------------------------------------------------------------------------- src/Dynamics.cpp
def spatial_inertia(link):                        # :4-13
    I = np.zeros((6, 6))
    cx = cross_matrix(link["com"])
    I[:3, :3] = link["inertia"] - link["mass"] * cx @ cx
    I[:3, 3:] = link["mass"] * cx
    I[3:, :3] = -link["mass"] * cx
    I[3:, 3:] = link["mass"] * np.eye(3)
    return I


class Dynamics:
    def compute_all(self, rb):                    # :202-216
        self.I = [spatial_inertia(rb.links[i]) if (i == 0 or ACT[i]) else np.zeros((6, 6)) for i in range(NF)]   # :15-27
        self.C = self.compute_c(rb, True)
        self.Cg = self.compute_c(rb, False)
        self.compute_m(rb)
        self.centroidal(rb)
        self.Jpqp = np.concatenate([self.jpqp_frame(rb, 7), self.jpqp_frame(rb, 14)])

    def forward_ne(self, rb, qD, vel, acc, f):    # :124-146
        S = np.array([0, 0, 1, 0, 0, 0.0])
        for i in range(1, NF):
            if ACT[i]:
                qd = qD[ACT[i] + 6 - 1]
                vel[i] = rb.X[i] @ vel[PARENT[i]] + S * qd
                acc[i] = rb.X[i] @ acc[PARENT[i]] + spatial_cross(vel[i]) @ S * qd
                f[i] = self.I[i] @ acc[i] + spatial_cross_force(vel[i]) @ self.I[i] @ vel[i]
            else:
                vel[i] = rb.X[i] @ vel[PARENT[i]]
                acc[i] = rb.X[i] @ acc[PARENT[i]]

    def compute_c(self, rb, gravity):             # :29-60
        C = np.zeros(NQ)
        g = np.zeros(6); g[5] = 9.81 if gravity else 0.0
        qD = swap_base_velocity(rb.X[0], rb.v)    # Robot::v_ is the PREVIOUS call's velocity here (controller.cpp:56 before :59)
        vel, acc, f = [None] * NF, [None] * NF, [None] * NF
        vel[0] = qD[:6]; acc[0] = rb.X[0] @ g
        f[0] = self.I[0] @ acc[0] + spatial_cross_force(vel[0]) @ self.I[0] @ vel[0]
        self.forward_ne(rb, qD, vel, acc, f)
        S = np.array([0, 0, 1, 0, 0, 0.0])
        for i in range(NF - 1, 0, -1):            # backwardNewtonEuler, :148-163
            if ACT[i]:
                C[ACT[i] + 6 - 1] = S @ f[i]
                f[PARENT[i]] = f[PARENT[i]] + rb.X[i].T @ f[i]
        C[:6] = f[0]
        return C

    def compute_m(self, rb):                      # :62-101
        M = np.zeros((NQ, NQ)); H = np.zeros((24, 24)); F2 = np.zeros((6, 24))
        S = np.array([0, 0, 1, 0, 0, 0.0])
        Ic = [m.copy() for m in self.I]
        for i in range(NF - 1, -1, -1):
            if ACT[i]:
                Ic[PARENT[i]] = Ic[PARENT[i]] + rb.X[i].T @ Ic[i] @ rb.X[i]
                fi = Ic[i] @ S
                H[ACT[i] - 1, ACT[i] - 1] = S @ fi
                j = i
                while PARENT[j] != 0:
                    fi = rb.X[j].T @ fi
                    j = PARENT[j]
                    H[ACT[j] - 1, ACT[i] - 1] = S @ fi
                    H[ACT[i] - 1, ACT[j] - 1] = H[ACT[j] - 1, ACT[i] - 1]
                F2[:, ACT[i] - 1] = rb.X[j].T @ fi
        M[:6, :6] = Ic[0]; M[6:, 6:] = H; M[:6, 6:] = F2; M[6:, :6] = F2.T
        self.M = M

    def centroidal(self, rb):                     # :103-121
        Ic1 = self.M[:6, :6]; F = self.M[:6, 6:]
        p1G = np.array([Ic1[2, 4], Ic1[0, 5], Ic1[1, 3]]) / rb.mass
        R = rb.T[0][:3, :3]
        X1G = np.zeros((6, 6)); X1G[:3, :3] = R; X1G[3:, 3:] = R; X1G[:3, 3:] = -R @ cross_matrix(p1G)
        self.AG = np.hstack([X1G @ Ic1, X1G @ F])
        self.AGpqp = X1G @ self.Cg[:6]

    def jpqp_frame(self, rb, frame):              # :165-200
        qD = swap_base_velocity(rb.X[0], rb.v)
        vel, acc, f = [None] * NF, [None] * NF, [None] * NF
        vel[0] = qD[:6]; acc[0] = np.zeros(6)
        self.forward_ne(rb, qD, vel, acc, f)
        R = rb.T[frame][:3, :3]
        R6 = np.zeros((6, 6)); R6[:3, :3] = R; R6[3:, 3:] = R
        return R6 @ acc[frame]


# ----------------------------------------------------------------------------- src/invKinematics.cpp:72-149
def feet_jacobian(rb):
    S = np.array([0, 0, 1, 0, 0, 0.0])
    J = np.zeros((12, NQ))
    for row, frame in ((0, 7), (6, 14)):
        Jf = np.zeros((6, NQ))
        Xn = rb.X[frame]
        for j in range(frame - 1, frame - 7, -1):          # the six leg joints, from the ankle up
            Jf[:, ACT[j] + 6 - 1] = Xn @ S
            Xn = Xn @ rb.X[j]
        Jf[:, :6] = Xn
        R = rb.T[frame][:3, :3]
        R6 = np.zeros((6, 6)); R6[:3, :3] = R; R6[3:, 3:] = R
        J[row:row + 6] = R6 @ Jf
    return J


# ----------------------------------------------------------------------------- src/mpcLinearPendulum.cpp
class Mpc:
    def __init__(self, dt, time_horizon, z_com, alpha=1e-3, beta=1.0, gravity=9.81):      # :41-68, hpp:43-49
        self.dt, self.z_com, self.alpha, self.beta = dt, z_com, alpha, beta
        N = self.N = int(time_horizon / dt)
        self.A = np.array([[1, dt], [0, 1.0]]); self.B = np.array([dt * dt / 2, dt]); Cm = np.array([1, 0.0])
        self.D = -z_com / gravity
        self.Px = np.zeros((N + 1, 2)); self.Pu = np.zeros((N + 1, N + 1))
        self.Px[0] = Cm; self.Pu[0, 0] = self.D
        Ap = np.eye(2)
        for i in range(1, N + 1):
            Ap = Ap @ self.A
            self.Px[i] = Cm @ Ap
            self.Pu[i, i - 1] = Cm @ self.B
            self.Pu[i, i] = self.D
            Aj = np.eye(2)
            for j in range(1, N - i + 1):
                Aj = Aj @ self.A
                self.Pu[i + j, i - 1] = Cm @ Aj @ self.B

    def compute(self, pos, vel, zx, zy, t):       # :78-109; the QP has no constraint at all: u = -H^-1 g
        H = self.alpha * np.eye(self.N + 1) + self.beta * (self.Pu.T @ self.Pu)
        k = int(t / self.dt)
        out = []
        for x0, v0, z in ((pos[0], vel[0], zx), (pos[1], vel[1], zy)):
            xk = np.array([x0, v0])
            g = self.beta * self.Pu.T @ (self.Px @ xk - z[k:k + self.N + 1])
            acc = np.linalg.solve(H, -g)[0]
            xk = self.A @ xk + self.B * acc
            out.append(np.array([xk[0], xk[1], acc]))
        self.xRef, self.yRef, self.k = out[0], out[1], k
        return k


# ----------------------------------------------------------------------------- generic QP: primal active set on the KKT system
def solve_qp(H, g, A, lbA, ubA, max_iter=400):
    """min 1/2 x'Hx + g'x  s.t. rows with lbA == ubA are equalities, the others lbA <= a'x (ubA = +inf there, controller.cpp:423-436)."""
    n = len(g)
    eq = [i for i in range(len(lbA)) if lbA[i] == ubA[i]]
    ineq = [i for i in range(len(lbA)) if lbA[i] != ubA[i]]
    W = set()
    x = None
    for _ in range(max_iter):
        rows = eq + sorted(W)
        Aw, bw = A[rows], lbA[rows]
        K = np.block([[H, Aw.T], [Aw, np.zeros((len(rows), len(rows)))]])
        rhs = np.concatenate([-g, bw])
        sol = np.linalg.solve(K, rhs)
        sol += np.linalg.solve(K, rhs - K @ sol)                      # one refinement step (H spans 13 orders of magnitude)
        x, lam = sol[:n], -sol[n:]                                    # H x + g = Aw' lam
        viol = [(A[i] @ x - lbA[i], i) for i in ineq if i not in W]
        worst = min(viol) if viol else (0.0, None)
        if worst[0] < -1e-11 * (1.0 + np.abs(x).max()):
            W.add(worst[1]); continue
        lam_in = [(lam[len(eq) + j], i) for j, i in enumerate(sorted(W))]
        neg = min(lam_in) if lam_in else (0.0, None)
        if neg[0] < -1e-9 * (1.0 + np.abs(lam).max()):
            W.discard(neg[1]); continue
        return x, sorted(W)
    raise RuntimeError("active-set iteration did not settle")


# ----------------------------------------------------------------------------- src/controller.cpp
class Controller:
    mu, KpJ, KdJ, KpM, KdM, KpF, KdF = 0.7, 300.0, 34.0, 10.0, 6.32, 500.0, 44.0        # controller.hpp:81,102-111
    wCoML, wCoMK, wBasePos, wBaseAng, wJoints, wForce, wFoot = 4000.0, 0.0, 10.0, 10.0, 1.0, 1.0, 100000.0   # :118-124

    def __init__(self, rb, mpc, zx, zy, rF, lF):
        self.rb, self.mpc, self.zx, self.zy, self.rF, self.lF = rb, mpc, zx, zy, rF, lF
        self.dyn = Dynamics()
        m = self.mu
        self.friction = np.array([[m, 0, -m, 0], [0, m, 0, -m], [1, 1, 1, 1.0]])           # controller.cpp:33-36

    def stand_step(self, q, dq, t):               # :48-79
        rb = self.rb
        rb.update_state(q)
        self.dyn.compute_all(rb)
        self.J = feet_jacobian(rb)
        rb.update_velocity(dq, self.dyn.AG)
        self.mpc.compute(rb.CoM[:2], rb.comVel[:2], self.zx, self.zy, t)
        return self.wbc(t)

    def pd_feet(self, t):                         # :327-386
        rb = self.rb
        v = swap_base_velocity(rb.X[0], rb.v)
        out = np.zeros(12)
        for k, (frame, co) in enumerate(((7, self.rF), (14, self.lF))):
            vel = self.J[6 * k:6 * k + 6] @ v
            err = rb.Rf_q0.T @ rb.T[frame][:3, :3]
            e = -rb.Rf_q0 @ rot_to_axis_angle(err)
            ref = np.array([polyval(c, t) for c in co])
            dref = np.concatenate([np.zeros(3), [polyval(polyder(c), t) for c in co]])
            ddref = np.concatenate([np.zeros(3), [polyval(polyder(polyder(c)), t) for c in co]])
            pos_err = np.concatenate([e, ref - rb.T[frame][:3, 3]])
            out[6 * k:6 * k + 6] = self.KpF * pos_err + self.KdF * (dref - vel) + ddref
        return out

    def wbc(self, t):                             # :81-154, 388-479
        rb, d, J = self.rb, self.dyn, self.J
        n = NQ
        qref = self.KpJ * (initial_configuration() - rb.q) + self.KdJ * (np.zeros(n) - rb.v)       # :296-308
        qref[:3], qref[3:6] = qref[3:6].copy(), qref[:3].copy()
        posRef = np.array([self.mpc.xRef[0], self.mpc.yRef[0], self.mpc.z_com])                    # :310-325
        velRef = np.array([self.mpc.xRef[1], self.mpc.yRef[1], 0.0])
        accRef = np.array([self.mpc.xRef[2], self.mpc.yRef[2], 0.0])
        href = np.zeros(6)
        href[3:] = rb.mass * (self.KpM * (posRef - rb.CoM) + self.KdM * (velRef - rb.comVel) + accRef)
        href[:3] = self.KdM * (np.zeros(3) - rb.angMom)
        fref = self.pd_feet(t)
        WJ = np.diag([self.wBasePos] * 3 + [self.wBaseAng] * 3 + [self.wJoints] * 24)
        WC = np.diag([self.wCoMK] * 3 + [self.wCoML] * 3)
        WF = self.wFoot * np.eye(12)
        H = np.zeros((74, 74))
        H[:n, :n] = d.AG.T @ WC @ d.AG + WJ + J.T @ WF @ J
        H[n:n + 12, n:n + 12] = self.wForce * np.eye(12)
        H[42:, 42:] = 1e-8 * np.eye(32)                                                           # :117
        g = np.zeros(74)
        g[:n] = d.AG.T @ WC @ d.AGpqp - d.AG.T @ WC @ href - WJ @ qref + J.T @ WF @ d.Jpqp - J.T @ WF @ fref
        H = 0.5 * (H + H.T)
        A = np.zeros((50, 74)); lb = np.zeros(50); ub = np.zeros(50)
        A[:6, :n] = d.M[:6]; A[:6, n:n + 12] = -(J.T)[:6]; lb[:6] = ub[:6] = -d.C[:6]
        iR_n, iR_f, iL_n, iL_f, iR_c, iL_c = 30, 33, 36, 39, 42, 58                                 # :171-176
        for (rowf, rown, idf, idn, idc) in ((6, 9, iR_f, iR_n, iR_c), (12, 15, iL_f, iL_n, iL_c)):   # rows fR 0..2, nR 3..5, fL 6..8, nL 9..11 (+6)
            for k in range(3):
                A[rowf + k, idf + k] = -1; A[rown + k, idn + k] = -1
            for vtx in range(4):
                A[rowf:rowf + 3, idc + 4 * vtx: idc + 4 * vtx + 4] = self.friction
                A[rown:rown + 3, idc + 4 * vtx: idc + 4 * vtx + 4] = cross_matrix(rb.foot_vertices[vtx]) @ self.friction
        A[18:, 42:] = np.eye(32); ub[18:] = 1e20
        x, active = solve_qp(H, g, A, lb, ub)
        gen = d.M @ x[:n] + d.C - J.T @ x[n:n + 12]
        acc = np.linalg.solve(rb.X[0], x[:6])                                                      # :143
        qdd = np.concatenate([acc[3:], acc[:3], x[6:n]])
        return dict(tau=gen[6:], f=x[n:n + 12], qpp=qdd, x=x, H=H, g=g, A=A, lbA=lb, active=active, qref=qref, href=href, fref=fref)


def offline_system(dt, time_horizon, z_com, sim_time=5.0):
    """apps/offline/main.cpp:12-58 without the IK (the posture is an input): stance ZMP (zmpGeneration.cpp:39-60), constant foot
    polynomials (footCoeffTrajectory with current == desired position gives constants)."""
    rb = Robot()
    mpc = Mpc(dt, time_horizon, z_com)
    n = int((sim_time + 0.5) / dt)
    rF = [np.array([0.0]), np.array([-0.05]), np.array([0.0])]
    lF = [np.array([0.0]), np.array([0.05]), np.array([0.0])]
    return Controller(rb, mpc, np.zeros(n), np.zeros(n), rF, lF)


def rk4_tick(ctl, x, t, dt):                      # rk4.hpp:5-18 over dynamics(), apps/offline/main.cpp:91-122
    def f(x, t):
        q, dq = x[:30], x[30:]
        out = ctl.stand_step(q, dq, t)
        xd = np.zeros(60)
        xd[:30] = dq
        xd[:3] = dq[:3] + cross_matrix(dq[3:6]) @ q[:3]
        e = q[3:6]
        Om = np.array([[np.cos(e[2]) / np.cos(e[1]), np.sin(e[2]) / np.cos(e[1]), 0], [-np.sin(e[2]), np.cos(e[2]), 0],
                       [np.cos(e[2]) * np.tan(e[1]), np.sin(e[2]) * np.tan(e[1]), 1]])
        xd[3:6] = Om @ dq[3:6]
        xd[30:] = out["qpp"]
        return xd, out
    k1, _ = f(x, t); k2, _ = f(x + 0.5 * dt * k1, t + 0.5 * dt); k3, _ = f(x + 0.5 * dt * k2, t + 0.5 * dt); k4, o4 = f(x + dt * k3, t + dt)
    return x + (dt / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4), o4


# ----------------------------------------------------------------------------- self-check + anchors
def check_against_golden(verbose=True):
    g = np.load(os.path.join(ROOT, "tests", "golden", "eval_vectors.npz"))
    worst = {}
    for i in range(g["q"].shape[0]):
        ctl = offline_system(float(g["dt"]), float(g["time_horizon"]), float(g["z_com"]))
        ctl.rb.v = g["v_prev"][i].copy()                               # Robot::v_ as the previous call left it
        out = ctl.stand_step(g["q"][i], g["v"][i], float(g["t"]))
        pairs = dict(tau=(out["tau"], g["tau"][i]), f=(out["f"], g["f"][i]), qpp=(out["qpp"], g["qpp"][i]), M=(ctl.dyn.M, g["M"][i]),
                     C=(ctl.dyn.C, g["C"][i]), AG=(ctl.dyn.AG, g["AG"][i]), J=(ctl.J, g["J"][i]), CoM=(ctl.rb.CoM, g["CoM"][i]),
                     Cg6=(ctl.dyn.Cg[:6], g["Cg6"][i]), AGpqp=(ctl.dyn.AGpqp, g["AGpqp"][i]), Jpqp=(ctl.dyn.Jpqp, g["Jpqp"][i]),
                     u0=(np.array([ctl.mpc.xRef[2], ctl.mpc.yRef[2]]), g["u0"][i]), a=(out["x"][:42], g["x"][i][:42]))
        assert ctl.mpc.k == int(g["k"][i])
        for k, (a, b) in pairs.items():
            scale = max(np.abs(b).max(), np.abs(g["C"][i]).max() if k in ("Cg6", "AGpqp") else 0.0, 1e-300)
            worst[k] = max(worst.get(k, 0.0), float(np.abs(np.asarray(a) - b).max() / scale))
    if verbose:
        print("independent restatement vs tests/golden/eval_vectors.npz (max error relative to the vector scale):")
        for k, v in worst.items():
            print("  %-6s %.2e" % (k, v))
    return worst


def survey_anchors(ticks=500):
    """The handful of scalars SURVEY.md 8c quotes from its (uncommitted) scratch transliteration, regenerated by this file."""
    ik = json.load(open(os.path.join(ROOT, "tests", "golden", "ik_posture.json")))
    rb0 = Robot()
    a = {"total_mass": rb0.mass, "com_initial_configuration": rb0.CoM.tolist()}
    q0 = np.array(ik["q"])
    rb0.update_state(q0)
    a["ik_posture_com"] = rb0.CoM.tolist()
    a["ik_posture_right_sole"] = rb0.T[7][:3, 3].tolist(); a["ik_posture_left_sole"] = rb0.T[14][:3, 3].tolist()
    ctl = offline_system(0.01, 0.5, rb0.CoM[2])
    mpc = ctl.mpc
    Hm = mpc.alpha * np.eye(mpc.N + 1) + mpc.beta * mpc.Pu.T @ mpc.Pu
    K = mpc.beta * np.linalg.solve(Hm, np.eye(mpc.N + 1)[0]) @ mpc.Pu.T
    a.update(D=mpc.D, K0=K[0], K1=K[1], K_sum=K.sum(), K_Px=(K @ mpc.Px).tolist())
    out = ctl.stand_step(q0, np.zeros(30), 0.0)
    a.update(tick0_u0x=ctl.mpc.xRef[2], tick0_f=out["f"].tolist(), tick0_tau_rknee=out["tau"][3], tick0_tau_lknee=out["tau"][9],
             tick0_C5=ctl.dyn.C[5], tick0_AG44=ctl.dyn.AG[4, 4], tick0_c_min=out["x"][42:].min(), tick0_c_max=out["x"][42:].max(),
             tick0_active=len(out["active"]), base_row_residual=float(np.abs((ctl.dyn.M @ out["x"][:30] + ctl.dyn.C - ctl.J.T @ out["x"][30:42])[:6]).max()))
    if ticks:
        ctl = offline_system(0.01, 0.5, rb0.CoM[2])
        x, t = np.concatenate([q0, np.zeros(30)]), 0.0
        for _ in range(ticks):
            x, o4 = rk4_tick(ctl, x, t, 0.01)
            t += 0.01
        a.update(ticks=ticks, com_x_after_ticks=ctl.rb.CoM[0], sum_fz_after_ticks=o4["f"][5] + o4["f"][11])
    return a


if __name__ == "__main__":
    if not available():
        sys.exit("needs /root/reference (build container only)")
    w = check_against_golden()
    assert max(w.values()) < 1e-8, w
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    anc = survey_anchors(nt)
    with open(os.path.join(ROOT, "tests", "golden", "survey_anchors.json"), "w") as f:
        json.dump(anc, f, indent=1)
    print(json.dumps(anc, indent=1))
