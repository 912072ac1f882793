#!/usr/bin/env python3
"""Generates tests/golden/traj_<model>.json: trajectory-level golden vectors - tau_t, the three
inverse-dynamics partials, the gradient and the Gauss-Newton Hessian bands - for the multi-DoF
contact models (hopper, mini_cheetah, allegro_hand) on a short horizon (N = 3: every branch of the
reference's block formulas occurs), by a route that shares no arithmetic with the oracle
(oracle/rigid_body.h, oracle/traj_opt.h) or the HIP kernels:

  * inverse dynamics by Kane's virtual power in 80-bit long double: position-level forward
    kinematics only; body-origin and angular Jacobians are NUMERICAL directional derivatives of
    poses (Richardson-extrapolated central differences); accelerations are numerical second
    derivatives along the path q (+) (s v + s^2 a / 2); Euler's equation per body, projected with
    the dense Jacobians (no recursion, no force propagation);
  * contact: reference optimizer/trajectory_optimizer.cc:247-386 restated in numpy;
  * the partials are DIRECTIONAL DERIVATIVES of ID(q, v, a) along the perturbation the reference
    applies (optimizer/trajectory_optimizer.cc:501-561: q_t[i] += e, v_t += e N+_t[:, i] / dt, ... with
    N+ not re-evaluated), taken by Richardson central differences with a step (1e-5 / 1e-6) far
    above the oracle's forward-difference step 1.5e-8 - i.e. the quantity the reference's finite
    difference approximates, not a re-run of its arithmetic;
  * gradient and Hessian bands from those partials by the block formulas of SURVEY.md A.1
    (optimizer/trajectory_optimizer.cc:1046-1080, 1103-1161) written with dense numpy blocks.

Model tables: idto_amd/models/*.model, whose conversion from the reference's URDF/SDF is
cross-checked by tools/make_model_fixture.py / tests/test_golden.py.  Conventions neither side can
see (Appendix D of SURVEY.md) are shared, so these vectors pin algebra, indexing, signs, the
contact law and the assembly - not Drake's unverifiable frame conventions.

Tolerances stored in the fixture, relative to the largest entry of the array: tau 1e-11 (observed
against the oracle: 1e-15 .. 8e-15), partials / gradient / Hessian 5e-6 (observed 2e-7 .. 1e-6: the
truncation error ~dq |tau''| of the reference's forward differences on the stiff contact models;
dtau_dqm, which is analytic in the reference, agrees to 1e-14).

Run from the repo root:  python tools/make_golden_traj.py   (a few minutes)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from idto_amd.model import load_model  # noqa: E402
from idto_amd.problem import load_config, make_problem, synthetic_trajectory  # noqa: E402

LD = np.longdouble
REV, PRI, PLA, FLO = 0, 1, 2, 3
SPHERE, BOX = 0, 1


def rot_axis(axis, ang):
    a = np.asarray(axis, LD)
    a = a / np.sqrt(a @ a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]], LD)
    return np.eye(3, dtype=LD) + np.sin(LD(ang)) * K + (1 - np.cos(LD(ang))) * (K @ K)


def quat_to_rot(qw):
    w, x, y, z = np.asarray(qw, LD) / np.sqrt(np.asarray(qw, LD) @ np.asarray(qw, LD))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], LD)


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], LD)


def hom(R, p):
    X = np.eye(4, dtype=LD)
    X[:3, :3] = R
    X[:3, 3] = p
    return X


class Mech:
    def __init__(self, model, contact):
        self.m = model
        self.cp = contact
        self.XPF = []
        for i in range(model.nbodies):
            x = np.asarray(model.X_PF[i], LD)
            self.XPF.append(hom(x[:9].reshape(3, 3), x[9:12]))
        eps = np.sqrt(LD(np.finfo(float).eps))
        s, k = LD(contact["smoothing_factor"]), LD(contact["contact_stiffness"])
        self.threshold = -s * np.log(np.exp(eps / (s * k)) - 1)   # TO.cc:266-269

    def fk(self, q):
        m, X = self.m, []
        for i in range(m.nbodies):
            par = int(m.parent[i])
            XW = (X[par] if par >= 0 else np.eye(4, dtype=LD)) @ self.XPF[i]
            jt, s = int(m.jtype[i]), int(m.qstart[i])
            if jt == REV:
                XM = hom(rot_axis(m.axis[i], q[s]), np.zeros(3, LD))
            elif jt == PRI:
                XM = hom(np.eye(3, dtype=LD), np.asarray(m.axis[i], LD) * q[s])
            elif jt == PLA:
                XM = hom(rot_axis([0, 0, 1], q[s + 2]), np.array([q[s], q[s + 1], 0], LD))
            else:
                XM = hom(quat_to_rot(q[s:s + 4]), np.asarray(q[s + 4:s + 7], LD))
            X.append(XW @ XM)
        return X

    def advance(self, q, dv):
        """q (+) dv: exponential map on the quaternion (angular velocity in the joint's F frame,
        which for the world-attached floating joints of these models is the world frame)"""
        m = self.m
        out = np.array(q, LD)
        for i in range(m.nbodies):
            jt, s, vs = int(m.jtype[i]), int(m.qstart[i]), int(m.vstart[i])
            if jt in (REV, PRI):
                out[s] += dv[vs]
            elif jt == PLA:
                out[s:s + 3] += dv[vs:vs + 3]
            else:
                w = np.asarray(dv[vs:vs + 3], LD)
                th = np.sqrt(w @ w)
                dq = np.array([1, 0, 0, 0], LD) if th == 0 else np.concatenate([[np.cos(th / 2)], np.sin(th / 2) * w / th])
                qq = np.asarray(q[s:s + 4], LD)
                out[s:s + 4] = quat_mul(dq, qq / np.sqrt(qq @ qq))
                out[s + 4:s + 7] += dv[vs + 3:vs + 6]
        return out

    @staticmethod
    def vee(S):
        return np.array([S[2, 1] - S[1, 2], S[0, 2] - S[2, 0], S[1, 0] - S[0, 1]], LD) / 2

    def jacobians(self, q, h=LD(1e-4)):
        """per body: Jo (3 x nv, velocity of the body origin) and Jw (3 x nv, angular velocity),
        numerically: d/ds of the pose along each unit generalised velocity"""
        m = self.m
        X0 = self.fk(q)
        Jo = [np.zeros((3, m.nv), LD) for _ in range(m.nbodies)]
        Jw = [np.zeros((3, m.nv), LD) for _ in range(m.nbodies)]
        for j in range(m.nv):
            e = np.zeros(m.nv, LD)
            e[j] = 1
            S = {s: self.fk(self.advance(q, s * e)) for s in (h, -h, h / 2, -h / 2)}
            for i in range(m.nbodies):
                def d(f):
                    d1 = (f(S[h][i]) - f(S[-h][i])) / (2 * h)
                    d2 = (f(S[h / 2][i]) - f(S[-h / 2][i])) / h
                    return (4 * d2 - d1) / 3
                Jo[i][:, j] = d(lambda X: X[:3, 3])
                R0 = X0[i][:3, :3]
                Jw[i][:, j] = d(lambda X: self.vee(X[:3, :3] @ R0.T - np.eye(3, dtype=LD)))
        return X0, Jo, Jw

    def bias(self, q, v, h=LD(1e-3)):
        """per body: the part of the origin acceleration / angular acceleration that is quadratic in v
        (second derivative of the pose along q (+) s v), by the five-point stencil"""
        pts = {s: self.fk(self.advance(q, s * v)) for s in (-2 * h, -h, LD(0), h, 2 * h)}
        out = []
        for i in range(self.m.nbodies):
            p = {s: pts[s][i][:3, 3] for s in pts}
            acc = (-p[2 * h] + 16 * p[h] - 30 * p[LD(0)] + 16 * p[-h] - p[-2 * h]) / (12 * h * h)
            R0 = pts[LD(0)][i][:3, :3]
            # rotation vector of R(s) R0^T: its second derivative at 0 is the angular acceleration
            # (d/ds of omega(s) at s = 0; the 1/2 w x w' correction vanishes because w(0) = 0)
            def rv(X):
                Rr = X[:3, :3] @ R0.T
                c = (np.trace(Rr) - 1) / 2
                ang = np.arccos(np.clip(c, -1, 1))
                ax = self.vee(Rr - Rr.T) / 2          # sin(ang) * axis
                return ax if ang < 1e-12 else ax * (ang / np.sin(ang))
            w = {s: rv(pts[s][i]) for s in pts}
            alp = (-w[2 * h] + 16 * w[h] - 30 * w[LD(0)] + 16 * w[-h] - w[-2 * h]) / (12 * h * h)
            out.append((acc, alp))
        return out

    def signed_distance(self, ga, gb, X):
        m = self.m

        def pose(g):
            b = int(m.geom_body[g])
            xg = np.asarray(m.geom_X[g], LD)
            return (X[b] if b >= 0 else np.eye(4, dtype=LD)) @ hom(xg[:9].reshape(3, 3), xg[9:12])

        XA, XB = pose(ga), pose(gb)
        tA, tB = int(m.geom_type[ga]), int(m.geom_type[gb])
        sA, sB = np.asarray(m.geom_size[ga], LD), np.asarray(m.geom_size[gb], LD)
        if tA == SPHERE and tB == SPHERE:
            d = XB[:3, 3] - XA[:3, 3]
            dist = np.sqrt(d @ d)
            n = d / dist
            return dist - sA[0] - sB[0], n, XA[:3, 3] + n * sA[0], XB[:3, 3] - n * sB[0]
        if {tA, tB} == {SPHERE, BOX}:
            sph_is_A = tA == SPHERE
            XS, XX = (XA, XB) if sph_is_A else (XB, XA)
            rad, hbox = (sA[0], sB) if sph_is_A else (sB[0], sA)
            c = XX[:3, :3].T @ (XS[:3, 3] - XX[:3, 3])
            pc = np.clip(c, -hbox, hbox)
            if np.any(pc != c):
                dd = np.sqrt((c - pc) @ (c - pc))
                g = (c - pc) / dd
                phi = dd - rad
            else:
                depth = hbox - np.abs(c)
                ax = int(np.argmin(depth))
                g = np.zeros(3, LD)
                g[ax] = 1 if c[ax] >= 0 else -1
                pc = c.copy()
                pc[ax] = g[ax] * hbox[ax]
                phi = -depth[ax] - rad
            gW = XX[:3, :3] @ g
            boxW = XX[:3, 3] + XX[:3, :3] @ pc
            sphW = XS[:3, 3] - gW * rad
            return (phi, -gW, sphW, boxW) if sph_is_A else (phi, gW, boxW, sphW)
        ztop = XB[2, 3] + sB[2]   # moving box against the top face of the world-fixed box (include/idto_model.h)
        corners = [XA[:3, 3] + XA[:3, :3] @ (np.array([sx, sy, sz], LD) * sA)
                   for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
        best = min(corners, key=lambda c: c[2])
        return best[2] - ztop, np.array([0, 0, -1], LD), best, np.array([best[0], best[1], ztop], LD)

    def contact_tau(self, q, v, X, Jo, Jw):
        m, cp = self.m, self.cp
        tau = np.zeros(m.nv, LD)
        for ga, gb in zip(m.pair_a, m.pair_b):
            ga, gb = int(ga), int(gb)
            phi, nhat, Ca, Cb = self.signed_distance(ga, gb, X)
            if phi > self.threshold:
                continue
            pC = (Ca + Cb) / 2
            bA, bB = int(m.geom_body[ga]), int(m.geom_body[gb])

            def JC(b):   # Jacobian of the body-fixed point that currently sits at pC
                if b < 0:
                    return np.zeros((3, m.nv), LD)
                r = pC - X[b][:3, 3]
                K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]], LD)
                return Jo[b] - K @ Jw[b]
            JA, JB = JC(bA), JC(bB)
            vrel = (JB - JA) @ v
            vn = nhat @ vrel
            vt = vrel - vn * nhat
            s = vn / LD(cp["dissipation_velocity"])
            d = (1 - s) if s < 0 else ((s - 2) ** 2 / 4 if s < 2 else LD(0))
            sig, k = LD(cp["smoothing_factor"]), LD(cp["contact_stiffness"])
            fn_c = -k * phi if -phi / sig >= 37 else sig * k * np.log1p(np.exp(-phi / sig))
            fn = fn_c * d
            ft = -LD(cp["friction_coefficient"]) * fn * vt / np.sqrt(LD(cp["stiction_velocity"]) ** 2 + vt @ vt)
            fB = fn * nhat + ft
            tau += (JB - JA).T @ fB
        return tau

    def prepare(self, q):
        return self.jacobians(np.asarray(q, LD))

    def inverse_dynamics(self, q, v, a, prep=None):
        """tau = sum_i J_i^T (inertial wrench - gravity) + damping v - contact  (Kane)"""
        m = self.m
        q, v, a = np.asarray(q, LD), np.asarray(v, LD), np.asarray(a, LD)
        X, Jo, Jw = prep if prep is not None else self.jacobians(q)
        b = self.bias(q, v)
        g = np.asarray(m.gravity, LD)
        tau = np.zeros(m.nv, LD)
        for i in range(m.nbodies):
            R = X[i][:3, :3]
            c = R @ np.asarray(m.com[i], LD)
            I = np.asarray(m.inertia[i], LD)
            IW = R @ np.array([[I[0], I[3], I[4]], [I[3], I[1], I[5]], [I[4], I[5], I[2]]], LD) @ R.T
            w = Jw[i] @ v
            alp = Jw[i] @ a + b[i][1]
            a_o = Jo[i] @ a + b[i][0]
            a_c = a_o + np.cross(alp, c) + np.cross(w, np.cross(w, c))
            K = np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0]], LD)
            Jc = Jo[i] - K @ Jw[i]
            tau += Jc.T @ (LD(m.mass[i]) * (a_c - g)) + Jw[i].T @ (IW @ alp + np.cross(w, IW @ w))
        tau += np.asarray(m.damping, LD) * v
        if m.npairs:
            tau -= self.contact_tau(q, v, X, Jo, Jw)
        return tau


def nplus(model, q):
    """v = N+(q) qdot, from (0, w) = 2 qdot (x) q^-1 for a unit quaternion (world-frame w):
    w = 2 (-u qw' + qw u' + u x u'),  q = (qw, u)"""
    N = np.zeros((model.nv, model.nq), LD)
    for i in range(model.nbodies):
        jt, s, vs = int(model.jtype[i]), int(model.qstart[i]), int(model.vstart[i])
        if jt in (REV, PRI):
            N[vs, s] = 1
        elif jt == PLA:
            N[vs:vs + 3, s:s + 3] = np.eye(3)
        else:
            w, x, y, z = np.asarray(q[s:s + 4], LD)
            N[vs:vs + 3, s:s + 4] = 2 * np.array([[-x, w, -z, y], [-y, z, w, -x], [-z, -y, x, w]], LD)
            N[vs + 3:vs + 6, s + 4:s + 7] = np.eye(3)
    return N


def richardson(f, h):
    d1 = (f(h) - f(-h)) / (2 * h)
    d2 = (f(h / 2) - f(-h / 2)) / h
    return (4 * d2 - d1) / 3


def trajectory_golden(name, N, seed, lower, step):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    contact = {k: float(getattr(sp, k)) for k in ("contact_stiffness", "dissipation_velocity", "stiction_velocity",
                                                  "friction_coefficient", "smoothing_factor")}
    mech = Mech(model, contact)
    q = np.asarray(synthetic_trajectory(cfg, model, N, seed=seed, lower=lower), LD)
    nq, nv, dt = model.nq, model.nv, LD(prob.time_step)
    Np = [nplus(model, q[t]) for t in range(N + 1)]
    v = [np.asarray(prob.v_init, LD)] + [Np[t] @ (q[t] - q[t - 1]) / dt for t in range(1, N + 1)]
    a = [(v[t + 1] - v[t]) / dt for t in range(N)]
    prep = [mech.prepare(q[t]) for t in range(N + 1)]
    tau = [mech.inverse_dynamics(q[k + 1], v[k + 1], a[k], prep[k + 1]) for k in range(N)]
    h = LD(step)
    P, T, M = [np.zeros((nv, nq), LD) for _ in range(N)], [np.zeros((nv, nq), LD) for _ in range(N)], \
        [np.zeros((nv, nq), LD) for _ in range(N)]
    for k in range(N):
        for i in range(nq):
            e = np.zeros(nq, LD)
            e[i] = 1
            n1, n0 = Np[k + 1][:, i], Np[k][:, i]
            # d tau_k / d q_{k+1}[i]  (TO.cc:514-531)
            P[k][:, i] = richardson(lambda s: mech.inverse_dynamics(q[k + 1] + s * e, v[k + 1] + s * n1 / dt,
                                                                    a[k] + s * n1 / dt / dt), h)
            if k >= 1:   # d tau_k / d q_k[i]  (:518-521, :534-540)
                T[k][:, i] = richardson(lambda s: mech.inverse_dynamics(q[k + 1], v[k + 1] - s * n1 / dt,
                                                                        a[k] - s * (n1 + n0) / dt / dt, prep[k + 1]), h)
            if k >= 2:   # d tau_k / d q_{k-1}[i] = M(q_{k+1}) N+_k[:, i] / dt^2  (:556-561)
                M[k][:, i] = richardson(lambda s: mech.inverse_dynamics(q[k + 1], v[k + 1], a[k] + s * n0 / dt / dt,
                                                                        prep[k + 1]), h)
        print(name, "partials of tau", k, flush=True)
    # ---- gradient and Hessian bands, SURVEY.md A.1 (TO.cc:1046-1080, 1103-1161)
    Qq, Qv, R = 2 * dt * np.asarray(prob.Qq, LD), 2 * dt * np.asarray(prob.Qv, LD), 2 * dt * np.asarray(prob.R, LD)
    Qfq, Qfv = 2 * np.asarray(prob.Qf_q, LD), 2 * np.asarray(prob.Qf_v, LD)
    V = [Np[t] / dt for t in range(N + 1)]
    W = [-Np[t] / dt for t in range(N + 1)]
    eq = [q[t] - np.asarray(prob.q_nom[t], LD) for t in range(N + 1)]
    ev = [v[t] - np.asarray(prob.v_nom[t], LD) for t in range(N + 1)]
    g = np.zeros((N + 1, nq), LD)
    A, B, C = np.zeros((N + 1, nq, nq), LD), np.zeros((N + 1, nq, nq), LD), np.zeros((N + 1, nq, nq), LD)
    C[0] = np.eye(nq)
    for t in range(1, N):
        last = t == N - 1
        g[t] = Qq @ eq[t] + V[t].T @ Qv @ ev[t] + W[t + 1].T @ (Qfv if last else Qv) @ ev[t + 1] \
            + P[t - 1].T @ R @ tau[t - 1] + T[t].T @ R @ tau[t]
        C[t] = Qq + V[t].T @ Qv @ V[t] + P[t - 1].T @ R @ P[t - 1] + T[t].T @ R @ T[t] \
            + W[t + 1].T @ (Qfv if last else Qv) @ W[t + 1]
        if not last:
            g[t] += M[t + 1].T @ R @ tau[t + 1]
            C[t] += M[t + 1].T @ R @ M[t + 1]
            B[t + 1] = P[t].T @ R @ T[t] + T[t + 1].T @ R @ M[t + 1] + V[t + 1].T @ Qv @ W[t + 1]
            A[t + 2] = P[t + 1].T @ R @ M[t + 1]
        else:
            B[N] = P[N - 1].T @ R @ T[N - 1] + V[N].T @ Qfv @ W[N]
    g[N] = P[N - 1].T @ R @ tau[N - 1] + Qfq @ eq[N] + V[N].T @ Qfv @ ev[N]
    C[N] = Qfq + V[N].T @ Qfv @ V[N] + P[N - 1].T @ R @ P[N - 1]
    f = lambda x: np.asarray(x, np.float64).tolist()
    return dict(config=name, num_steps=N, seed=seed, lower=lower, contact=contact, generator="tools/make_golden_traj.py",
                method="Kane virtual power in long double, numerical Jacobians; partials = directional derivatives "
                       "(Richardson central differences); g, H by the block formulas of TO.cc:1046-1161",
                tolerance_tau=1e-11, tolerance_derivatives=5e-6, q=f(q), tau=f(tau), dtau_dqp=f(P), dtau_dqt=f(T),
                dtau_dqm=f(M), gradient=f(g), H_A=f(A), H_B=f(B), H_C=f(C))


CASES = [("hopper", 3, 21, 0.02, 1e-5), ("mini_cheetah", 3, 22, 0.02, 1e-5), ("allegro_hand", 3, 23, 0.0, 1e-6)]


def main():
    outdir = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    only = sys.argv[1:]
    for name, N, seed, lower, step in CASES:
        if only and name not in only:
            continue
        fix = trajectory_golden(name, N, seed, lower, step)
        with open(os.path.join(outdir, f"traj_{name}.json"), "w") as f:
            json.dump(fix, f)
        print(name, "written: |tau|", np.abs(fix["tau"]).max(), "|P|", np.abs(fix["dtau_dqp"]).max())


if __name__ == "__main__":
    main()
