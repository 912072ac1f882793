"""Multibody model tables (`idto_model_t`) on the Python side.

The reference builds a Drake ``MultibodyPlant`` from URDF/SDF
(e.g. reference examples/mini_cheetah/mini_cheetah.cc:43-55); Drake is not part
of this build, so the plant is replaced by flat tables (include/idto_model.h)
stored as small text files under ``idto_amd/models/*.model`` (written by
tools/convert_models.py).  This module reads/writes that format and packs it
into the C struct for the C-ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

JOINT_TYPES = {"revolute": 0, "prismatic": 1, "planar": 2, "floating": 3}
JOINT_NQ = {0: 1, 1: 1, 2: 3, 3: 7}
JOINT_NV = {0: 1, 1: 1, 2: 3, 3: 6}
GEOM_TYPES = {"sphere": 0, "box": 1}
MAX_PATHS = 8
MAX_CHAIN = 8

MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")


class CModel(C.Structure):
    _fields_ = [
        ("nbodies", C.c_int), ("nq", C.c_int), ("nv", C.c_int),
        ("parent", C.POINTER(C.c_int)), ("jtype", C.POINTER(C.c_int)),
        ("qstart", C.POINTER(C.c_int)), ("vstart", C.POINTER(C.c_int)),
        ("X_PF", C.POINTER(C.c_double)), ("axis", C.POINTER(C.c_double)),
        ("mass", C.POINTER(C.c_double)), ("com", C.POINTER(C.c_double)),
        ("inertia", C.POINTER(C.c_double)), ("damping", C.POINTER(C.c_double)),
        ("actuated", C.POINTER(C.c_int)), ("gravity", C.c_double * 3),
        ("ngeoms", C.c_int), ("geom_body", C.POINTER(C.c_int)), ("geom_type", C.POINTER(C.c_int)),
        ("geom_X", C.POINTER(C.c_double)), ("geom_size", C.POINTER(C.c_double)),
        ("npairs", C.c_int), ("pair_a", C.POINTER(C.c_int)), ("pair_b", C.POINTER(C.c_int)),
        ("npaths", C.c_int), ("common_body", C.c_int),
        ("body_path", C.POINTER(C.c_int)), ("pair_path", C.POINTER(C.c_int)),
    ]


class CContactParams(C.Structure):
    _fields_ = [("contact_stiffness", C.c_double), ("dissipation_velocity", C.c_double),
                ("stiction_velocity", C.c_double), ("friction_coefficient", C.c_double),
                ("smoothing_factor", C.c_double)]


class CProblem(C.Structure):
    _fields_ = [("num_steps", C.c_int), ("time_step", C.c_double),
                ("q_init", C.POINTER(C.c_double)), ("v_init", C.POINTER(C.c_double)),
                ("Qq", C.POINTER(C.c_double)), ("Qv", C.POINTER(C.c_double)),
                ("Qf_q", C.POINTER(C.c_double)), ("Qf_v", C.POINTER(C.c_double)),
                ("R", C.POINTER(C.c_double)), ("q_nom", C.POINTER(C.c_double)),
                ("v_nom", C.POINTER(C.c_double))]


class CSolverParams(C.Structure):
    _fields_ = [("check_convergence", C.c_int),
                ("rel_cost_reduction", C.c_double), ("abs_cost_reduction", C.c_double),
                ("rel_gradient_along_dq", C.c_double), ("abs_gradient_along_dq", C.c_double),
                ("rel_state_change", C.c_double), ("abs_state_change", C.c_double),
                ("method", C.c_int), ("linesearch_method", C.c_int), ("max_iterations", C.c_int),
                ("max_linesearch_iterations", C.c_int), ("gradients_method", C.c_int),
                ("linear_solver", C.c_int), ("normalize_quaternions", C.c_int), ("verbose", C.c_int),
                ("scaling", C.c_int), ("scaling_method", C.c_int), ("equality_constraints", C.c_int),
                ("Delta0", C.c_double), ("Delta_max", C.c_double), ("num_threads", C.c_int),
                ("print_debug_data", C.c_int), ("debug_compare_against_dense", C.c_int), ("exact_hessian", C.c_int),
                ("plot_dumps", C.c_int)]


class CStats(C.Structure):
    _fields_ = [("capacity", C.c_int), ("count", C.c_int), ("solve_time", C.c_double),
                ("iteration_times", C.POINTER(C.c_double)), ("iteration_costs", C.POINTER(C.c_double)),
                ("linesearch_iterations", C.POINTER(C.c_int)), ("linesearch_alphas", C.POINTER(C.c_double)),
                ("trust_region_radii", C.POINTER(C.c_double)), ("q_norms", C.POINTER(C.c_double)),
                ("dq_norms", C.POINTER(C.c_double)), ("dqH_norms", C.POINTER(C.c_double)),
                ("trust_ratios", C.POINTER(C.c_double)), ("gradient_norms", C.POINTER(C.c_double)),
                ("dL_dqs", C.POINTER(C.c_double)), ("h_norms", C.POINTER(C.c_double)),
                ("merits", C.POINTER(C.c_double)), ("total", C.c_int)]


def dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def iptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _f(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a if shape is None else a.reshape(shape)


def _i(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


@dataclass
class Model:
    name: str = "model"
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.81]))
    body_names: list = field(default_factory=list)
    parent: np.ndarray = None      # [nb]
    jtype: np.ndarray = None       # [nb]
    X_PF: np.ndarray = None        # [nb, 12] (R row-major, p)
    axis: np.ndarray = None        # [nb, 3]
    mass: np.ndarray = None        # [nb]
    com: np.ndarray = None         # [nb, 3]
    inertia: np.ndarray = None     # [nb, 6] xx yy zz xy xz yz about the COM
    damping: np.ndarray = None     # [nv]
    actuated: np.ndarray = None    # [nv]
    geom_body: np.ndarray = None
    geom_type: np.ndarray = None
    geom_X: np.ndarray = None      # [ng, 12]
    geom_size: np.ndarray = None   # [ng, 3]
    pair_a: np.ndarray = None
    pair_b: np.ndarray = None
    npaths: int = 1
    common_body: int = -1
    body_path: np.ndarray = None
    pair_path: np.ndarray = None

    # ---- derived ------------------------------------------------------------
    @property
    def nbodies(self):
        return len(self.parent)

    @property
    def qstart(self):
        return _i(np.concatenate([[0], np.cumsum([JOINT_NQ[int(j)] for j in self.jtype])[:-1]]))

    @property
    def vstart(self):
        return _i(np.concatenate([[0], np.cumsum([JOINT_NV[int(j)] for j in self.jtype])[:-1]]))

    @property
    def nq(self):
        return int(sum(JOINT_NQ[int(j)] for j in self.jtype))

    @property
    def nv(self):
        return int(sum(JOINT_NV[int(j)] for j in self.jtype))

    @property
    def ngeoms(self):
        return 0 if self.geom_body is None else len(self.geom_body)

    @property
    def npairs(self):
        return 0 if self.pair_a is None else len(self.pair_a)

    @property
    def quaternion_starts(self):
        """q indices where a [qw qx qy qz] block starts (floating joints)."""
        return [int(qs) for qs, jt in zip(self.qstart, self.jtype) if int(jt) == 3]

    @property
    def unactuated_dofs(self):
        """reference optimizer/trajectory_optimizer.cc:63-72 (no actuators at all => none)."""
        if int(np.sum(self.actuated)) == 0:
            return []
        return [i for i in range(self.nv) if not self.actuated[i]]

    def normalize(self):
        nb = self.nbodies
        self.parent = _i(self.parent)
        self.jtype = _i(self.jtype)
        self.X_PF = _f(self.X_PF, (nb, 12))
        self.axis = _f(self.axis, (nb, 3))
        self.mass = _f(self.mass, (nb,))
        self.com = _f(self.com, (nb, 3))
        self.inertia = _f(self.inertia, (nb, 6))
        self.damping = _f(self.damping, (self.nv,))
        self.actuated = _i(self.actuated)
        self.gravity = _f(self.gravity, (3,))
        ng = self.ngeoms
        self.geom_body = _i(self.geom_body if ng else [])
        self.geom_type = _i(self.geom_type if ng else [])
        self.geom_X = _f(self.geom_X if ng else np.zeros((0, 12)), (ng, 12))
        self.geom_size = _f(self.geom_size if ng else np.zeros((0, 3)), (ng, 3))
        self.pair_a = _i(self.pair_a if self.pair_a is not None else [])
        self.pair_b = _i(self.pair_b if self.pair_b is not None else [])
        if self.body_path is None:
            self.body_path = np.zeros(nb, dtype=np.int32)
        self.body_path = _i(self.body_path)
        if self.pair_path is None:
            self.pair_path = np.zeros(self.npairs, dtype=np.int32)
        self.pair_path = _i(self.pair_path)
        self.validate()
        return self

    def validate(self):
        nb = self.nbodies
        assert self.npaths in (1, 2, 4, 8), "npaths must be a power of two <= 8"
        for i in range(nb):
            assert self.parent[i] < i, "bodies must be topologically ordered"
            if int(self.jtype[i]) == 3:
                assert self.parent[i] == -1, "floating joints must hang off the world"
                assert np.allclose(self.X_PF[i], [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]), "floating X_PF must be identity"
        # star decomposition: every non-common body follows its predecessor in the
        # same path, or hangs off the common body / the world
        last_in_path = {}
        count = {}
        for i in range(nb):
            p = int(self.body_path[i])
            if i == self.common_body:
                assert p == -1 and self.parent[i] == -1
                continue
            assert 0 <= p < self.npaths
            par = int(self.parent[i])
            ok = par == -1 or par == self.common_body or par == last_in_path.get(p, -2)
            assert ok, f"body {i}: parent {par} is not world / common / previous body of path {p}"
            if par == self.common_body and par != -1:
                assert p not in count or True
            last_in_path[p] = i
            count[p] = count.get(p, 0) + 1
            assert count[p] <= MAX_CHAIN
        # each path has at most one chain root hanging off the common body
        roots = {}
        for i in range(nb):
            if i != self.common_body and self.common_body >= 0 and int(self.parent[i]) == self.common_body:
                p = int(self.body_path[i])
                assert p not in roots, "two chain roots of one path hang off the common body"
                roots[p] = i
        # a child inside a chain must directly follow its parent (one child per chain body)
        nchild = {}
        for i in range(nb):
            par = int(self.parent[i])
            if par >= 0 and par != self.common_body:
                nchild[par] = nchild.get(par, 0) + 1
                assert nchild[par] <= 1, "chain bodies may have at most one child"
                assert self.body_path[par] == self.body_path[i]
        for k in range(self.npairs):
            p = int(self.pair_path[k])
            for g in (int(self.pair_a[k]), int(self.pair_b[k])):
                b = int(self.geom_body[g])
                assert b == -1 or b == self.common_body or int(self.body_path[b]) == p, \
                    f"pair {k} touches body {b} outside path {p}"

    # ---- C packing ----------------------------------------------------------
    def to_c(self):
        """Returns (CModel, keepalive list)."""
        self.normalize()
        keep = dict(parent=self.parent, jtype=self.jtype, qstart=self.qstart, vstart=self.vstart,
                    X_PF=self.X_PF, axis=self.axis, mass=self.mass, com=self.com, inertia=self.inertia,
                    damping=self.damping, actuated=self.actuated, geom_body=self.geom_body,
                    geom_type=self.geom_type, geom_X=self.geom_X, geom_size=self.geom_size,
                    pair_a=self.pair_a, pair_b=self.pair_b, body_path=self.body_path, pair_path=self.pair_path)
        m = CModel()
        m.nbodies, m.nq, m.nv = self.nbodies, self.nq, self.nv
        for k in ("parent", "jtype", "qstart", "vstart", "actuated", "geom_body", "geom_type", "pair_a", "pair_b",
                  "body_path", "pair_path"):
            setattr(m, k, iptr(keep[k]))
        for k in ("X_PF", "axis", "mass", "com", "inertia", "damping", "geom_X", "geom_size"):
            setattr(m, k, dptr(keep[k]))
        m.gravity = (C.c_double * 3)(*self.gravity)
        m.ngeoms, m.npairs = self.ngeoms, self.npairs
        m.npaths, m.common_body = self.npaths, self.common_body
        return m, keep

    # ---- text format ----------------------------------------------------------
    def save(self, path):
        self.normalize()
        inv_j = {v: k for k, v in JOINT_TYPES.items()}
        inv_g = {v: k for k, v in GEOM_TYPES.items()}
        fmt = lambda a: " ".join(repr(float(x)) for x in np.ravel(a))
        with open(path, "w") as f:
            f.write("idto_model 1\n")
            f.write(f"name {self.name}\n")
            f.write(f"gravity {fmt(self.gravity)}\n")
            f.write(f"nbodies {self.nbodies}\n")
            f.write(f"npaths {self.npaths}\ncommon_body {self.common_body}\n")
            for i in range(self.nbodies):
                nm = self.body_names[i] if i < len(self.body_names) else f"body{i}"
                f.write(f"body {i} {nm} parent {int(self.parent[i])} joint {inv_j[int(self.jtype[i])]} "
                        f"path {int(self.body_path[i])}\n")
                f.write(f"  X_PF {fmt(self.X_PF[i])}\n")
                f.write(f"  axis {fmt(self.axis[i])}\n")
                f.write(f"  mass {fmt(self.mass[i])}\n")
                f.write(f"  com {fmt(self.com[i])}\n")
                f.write(f"  inertia {fmt(self.inertia[i])}\n")
            f.write(f"damping {fmt(self.damping)}\n")
            f.write("actuated " + " ".join(str(int(x)) for x in self.actuated) + "\n")
            f.write(f"ngeoms {self.ngeoms}\n")
            for g in range(self.ngeoms):
                f.write(f"geom {g} body {int(self.geom_body[g])} type {inv_g[int(self.geom_type[g])]} "
                        f"size {fmt(self.geom_size[g])}\n")
                f.write(f"  X_BG {fmt(self.geom_X[g])}\n")
            f.write(f"npairs {self.npairs}\n")
            for k in range(self.npairs):
                f.write(f"pair {int(self.pair_a[k])} {int(self.pair_b[k])} path {int(self.pair_path[k])}\n")


def load_model(name_or_path: str) -> Model:
    path = name_or_path
    if not os.path.exists(path):
        path = os.path.join(MODEL_DIR, name_or_path + ".model")
    toks = open(path).read().split()
    pos = 0

    def nxt():
        nonlocal pos
        pos += 1
        return toks[pos - 1]

    def expect(s):
        t = nxt()
        assert t == s, f"{path}: expected {s}, got {t}"

    def floats(n):
        return [float(nxt()) for _ in range(n)]

    expect("idto_model"); assert nxt() == "1"
    m = Model()
    expect("name"); m.name = nxt()
    expect("gravity"); m.gravity = np.array(floats(3))
    expect("nbodies"); nb = int(nxt())
    expect("npaths"); m.npaths = int(nxt())
    expect("common_body"); m.common_body = int(nxt())
    parent, jtype, path_, X, ax, mass, com, inert, names = [], [], [], [], [], [], [], [], []
    for i in range(nb):
        expect("body"); assert int(nxt()) == i
        names.append(nxt())
        expect("parent"); parent.append(int(nxt()))
        expect("joint"); jtype.append(JOINT_TYPES[nxt()])
        expect("path"); path_.append(int(nxt()))
        expect("X_PF"); X.append(floats(12))
        expect("axis"); ax.append(floats(3))
        expect("mass"); mass.append(floats(1)[0])
        expect("com"); com.append(floats(3))
        expect("inertia"); inert.append(floats(6))
    m.body_names, m.parent, m.jtype, m.body_path = names, parent, jtype, path_
    m.X_PF, m.axis, m.mass, m.com, m.inertia = X, ax, mass, com, inert
    m.parent = _i(m.parent); m.jtype = _i(m.jtype)
    expect("damping"); m.damping = floats(m.nv)
    expect("actuated"); m.actuated = [int(nxt()) for _ in range(m.nv)]
    expect("ngeoms"); ng = int(nxt())
    gb, gt, gs, gx = [], [], [], []
    for g in range(ng):
        expect("geom"); assert int(nxt()) == g
        expect("body"); gb.append(int(nxt()))
        expect("type"); gt.append(GEOM_TYPES[nxt()])
        expect("size"); gs.append(floats(3))
        expect("X_BG"); gx.append(floats(12))
    m.geom_body, m.geom_type, m.geom_size, m.geom_X = gb, gt, gs, gx
    expect("npairs"); npair = int(nxt())
    pa, pb, pp = [], [], []
    for _ in range(npair):
        expect("pair"); pa.append(int(nxt())); pb.append(int(nxt()))
        expect("path"); pp.append(int(nxt()))
    m.pair_a, m.pair_b, m.pair_path = pa, pb, pp
    return m.normalize()
