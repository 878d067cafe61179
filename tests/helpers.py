"""Shared helpers for the parity tests (oracle = checker; HIP path = thing under test)."""
import numpy as np

from oracle.pyoracle import Oracle

TOL_REL = 1e-6      # north_star: 1e-6 relative on torques / forces
TOL_FLOOR = 1e-9    # absolute floor as a fraction of max|.| (near-zero entries such as n_z)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    return float(np.abs(a - b).max() / scale)


def vec_err(a, b):
    """max|a - b| / max|b| of the SAME vector (north_star / SURVEY 8d); inf when the reference is all-zero and the vectors differ."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = float(np.abs(a - b).max()) if a.size else 0.0
    den = float(np.abs(b).max()) if b.size else 0.0
    if den == 0.0:
        return 0.0 if d == 0.0 else float("inf")
    return d / den


def close_on(a, b, tol, scale):
    """max|a - b| <= tol * scale with an EXPLICIT physical scale: for quantities that are small differences of larger ingredients
    (velocity products at rest, PD references of a robot standing on its references), where max|b| of the vector itself is round-off of
    those ingredients.  The caller names the ingredient scale; there is no implicit 1.0."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return bool(np.abs(a - b).max() <= tol * float(scale))


def close(a, b, tol=TOL_REL, scale=None):
    """The parity rule of north_star / SURVEY 8d: max|a - b| <= tol * max|b| of the SAME vector -- relative, with no absolute 1.0 in
    the denominator.  `scale` (optional) is the max-abs entry of the quantity SET the vector belongs to (e.g. the robot's weight for
    a contact-force vector that is identically zero in flight): it adds SURVEY's absolute floor TOL_FLOOR * scale = 1e-9 * scale to the
    allowed error.  Without it an all-zero reference demands identical vectors.  NaN anywhere fails."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b).max()
    allowed = tol * np.abs(b).max()
    if scale is not None:
        allowed = max(allowed, TOL_FLOOR * float(scale))
    return bool(d <= allowed)


WEIGHT = 5.305 * 9.81    # m g of the nominal NAO [N] (Robot::getMass 5.305 kg): the scale of the contact-force set


def perturbed_velocities(B, seed=20260001):
    """BASELINE config 2 perturbation: dq[0:2] ~ U(-0.3,0.3) m/s, dq[6:30] ~ N(0,0.05^2) rad/s."""
    v = np.zeros((B, 30))
    for i in range(B):
        rng = np.random.default_rng(seed + i)
        v[i, 0:2] = rng.uniform(-0.3, 0.3, 2)
        v[i, 6:] = rng.normal(0.0, 0.05, 24)
    return v


def oracle_system(dt, horizon_time, sim_time=2.0, raw_links=None):
    return Oracle(sim_time=sim_time, dt=dt, horizon_time=horizon_time, do_ik=True, raw_links=raw_links)


def dense_terms_from_debug(d):
    """Rebuild the reference-shaped matrices from the kernel's compact debug record."""
    parent = [-1, 0, 1, 2, 3, 4, 5, 6, 0, 8, 9, 10, 11, 12, 13, 0, 15, 16, 17, 18, 0, 20, 21, 22, 23, 0, 25, 26]
    T = np.zeros((28, 4, 4)); T[:, :3, :] = d["T"]; T[:, 3, 3] = 1
    X = np.zeros((28, 6, 6))
    for i in range(28):
        A = d["XE"][i].T
        X[i, :3, :3] = A; X[i, 3:, 3:] = A; X[i, 3:, :3] = d["XB"][i]
    M = np.zeros((30, 30))
    M[:6, :] = d["Mtop"]; M[6:, :6] = d["Mtop"][:, 6:].T
    start = [0] * 6 + [6] * 6 + [12] * 5 + [17] * 5 + [22] * 2
    nl = [6] * 12 + [5] * 10 + [2] * 2
    for a in range(24):
        for b in range(nl[a]):
            M[6 + a, 6 + start[a] + b] = d["Hl"][a, b]
    J = np.zeros((12, 30))
    for ft in range(2):
        J[6 * ft:6 * ft + 6, :6] = d["Jc"][ft][:, :6]
        J[6 * ft:6 * ft + 6, 6 + 6 * ft:12 + 6 * ft] = d["Jc"][ft][:, 6:]
    return dict(T=T, X=X, M=M, J=J, parent=parent)
