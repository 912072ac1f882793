#!/usr/bin/env python3
"""Offline converter: the reference's URDF/SDF robot descriptions + example YAMLs
-> this repo's flat model tables (idto_amd/models/*.model) and problem configs
(idto_amd/configs/*.yaml).

Run in the authoring container only (needs /root/reference); the generated
files are committed, so nothing reads /root/reference at test/bench time.

What it re-states (reference file:line):
  * examples/<name>/<name>.cc `CreatePlantModel` — which URDF/SDF is loaded and
    what is added in code (ground box: examples/hopper/hopper.cc:44-49,
    examples/mini_cheetah/mini_cheetah.cc:50-55; welded hand + free ball:
    examples/allegro_hand/allegro_hand.cc:88-113);
  * Drake's parsing conventions the reference relies on:
    welded links are merged into their parent body, a root link without a joint
    gets a quaternion floating joint, DoFs are numbered depth-first in joint
    declaration order, collision pairs are filtered for same/adjacent bodies and
    the declared filter groups, geometry A of a pair is the one registered first;
  * examples/example_base.cc:377-426 `SetProblemDefinition` fields (kept in the
    config; the interpolation itself lives in idto_amd/problem.py).

The "star" decomposition (paths / common body, include/idto_model.h) is an
evaluation-order annotation of this repo and is specified per model below.
"""
import math
import os
import re
import sys
import xml.etree.ElementTree as ET

import numpy as np
import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from idto_amd.model import Model, JOINT_TYPES, GEOM_TYPES  # noqa: E402

REF = "/root/reference"
OUT_MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "idto_amd", "models")
OUT_CONFIGS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "idto_amd", "configs")


# ---------------------------------------------------------------- transforms
def rpy_to_R(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


class X:
    """Rigid transform."""

    def __init__(self, R=None, p=None):
        self.R = np.eye(3) if R is None else np.asarray(R, float)
        self.p = np.zeros(3) if p is None else np.asarray(p, float)

    def __matmul__(self, o):
        return X(self.R @ o.R, self.p + self.R @ o.p)

    def inv(self):
        return X(self.R.T, -self.R.T @ self.p)

    def flat(self):
        return list(self.R.ravel()) + list(self.p)


def parse_xyz_rpy(el):
    if el is None:
        return X()
    xyz = [float(v) for v in el.get("xyz", "0 0 0").split()]
    rpy = [float(v) for v in el.get("rpy", "0 0 0").split()]
    return X(rpy_to_R(*rpy), xyz)


def parse_pose(text):
    v = [float(t) for t in (text or "0 0 0 0 0 0").split()]
    return X(rpy_to_R(*v[3:6]), v[0:3])


def inertia_matrix(ixx, iyy, izz, ixy, ixz, iyz):
    return np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]], float)


class Link:
    def __init__(self, name):
        self.name = name
        self.mass = 0.0
        self.com = np.zeros(3)          # in link frame
        self.I = np.zeros((3, 3))       # about COM, link axes
        self.geoms = []                 # (type, size3, X_LG, reg_index)


class Joint:
    def __init__(self):
        self.name = self.type = self.parent = self.child = None
        self.X_PJ = X()                 # joint (= child link) frame in the parent LINK frame
        self.axis = np.array([0, 0, 1.0])
        self.damping = 0.0
        self.auto_floating = False


# ---------------------------------------------------------------- URDF / SDF
def _parse_xml(path):
    """ElementTree needs the `drake:` prefix bound; the reference's files do not declare it."""
    text = open(path).read()
    text = re.sub(r"<(/?)drake:", r"<\1drake_", text)
    return ET.fromstring(text)


def parse_urdf(path, reg):
    root = _parse_xml(path)
    links, joints, actuated, groups = {}, [], set(), []
    order = []
    for le in root.findall("link"):
        L = Link(le.get("name"))
        ine = le.find("inertial")
        if ine is not None:
            L.mass = float(ine.find("mass").get("value"))
            Xi = parse_xyz_rpy(ine.find("origin"))
            L.com = Xi.p
            ie = ine.find("inertia")
            I = inertia_matrix(*[float(ie.get(k, "0")) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")])
            L.I = Xi.R @ I @ Xi.R.T
        for ce in le.findall("collision"):
            Xg = parse_xyz_rpy(ce.find("origin"))
            g = ce.find("geometry")
            if g.find("sphere") is not None:
                L.geoms.append(("sphere", [float(g.find("sphere").get("radius")), 0, 0], Xg, reg[0]))
            elif g.find("box") is not None:
                sz = [float(v) / 2 for v in g.find("box").get("size").split()]
                L.geoms.append(("box", sz, Xg, reg[0]))
            else:
                raise ValueError(f"unsupported collision geometry in {path}:{L.name}")
            reg[0] += 1
        links[L.name] = L
        order.append(L.name)
    for je in root.findall("joint"):
        J = Joint()
        J.name, J.type = je.get("name"), je.get("type")
        J.parent, J.child = je.find("parent").get("link"), je.find("child").get("link")
        J.X_PJ = parse_xyz_rpy(je.find("origin"))
        ax = je.find("axis")
        if ax is not None:
            a = np.array([float(v) for v in ax.get("xyz").split()])
            J.axis = a / np.linalg.norm(a)
        dyn = je.find("dynamics")
        if dyn is not None:
            J.damping = float(dyn.get("damping", "0"))
        joints.append(J)
    for te in root.findall("transmission"):
        for j in te.findall("joint"):
            actuated.add(j.get("name"))
    for ge in root.iter():
        if ge.tag.endswith("collision_filter_group"):
            members = [m.get("link") for m in ge if m.tag.endswith("member")]
            ignored = [m.get("name") for m in ge if m.tag.endswith("ignored_collision_filter_group")]
            groups.append((ge.get("name"), members, ignored))
    return links, order, joints, actuated, groups


def parse_sdf(path, reg):
    model = _parse_xml(path).find("model")
    links, joints, groups, order = {}, [], [], []
    X_ML = {}
    for le in model.findall("link"):
        L = Link(le.get("name"))
        X_ML[L.name] = parse_pose(le.findtext("pose"))
        ine = le.find("inertial")
        if ine is not None:
            L.mass = float(ine.findtext("mass"))
            Xi = parse_pose(ine.findtext("pose"))
            L.com = Xi.p
            ie = ine.find("inertia")
            I = inertia_matrix(*[float(ie.findtext(k, "0")) for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz")])
            L.I = Xi.R @ I @ Xi.R.T
        for ce in le.findall("collision"):
            Xg = parse_pose(ce.findtext("pose"))
            g = ce.find("geometry")
            # libsdformat's Geometry::Load checks <box> before <sphere>: an element
            # listing both (the palm, reference models/allegro_hand.sdf:47-56) is a box.
            if g.find("box") is not None:
                sz = [float(v) / 2 for v in g.find("box").findtext("size").split()]
                L.geoms.append(("box", sz, Xg, reg[0]))
            elif g.find("sphere") is not None:
                L.geoms.append(("sphere", [float(g.find("sphere").findtext("radius")), 0, 0], Xg, reg[0]))
            else:
                raise ValueError("unsupported collision geometry")
            reg[0] += 1
        links[L.name] = L
        order.append(L.name)
    for je in model.findall("joint"):
        J = Joint()
        J.name, J.type = je.get("name"), je.get("type")
        J.parent, J.child = je.findtext("parent"), je.findtext("child")
        assert je.find("pose") is None, "joint <pose> not supported"
        J.X_PJ = X_ML[J.parent].inv() @ X_ML[J.child]
        ax = je.find("axis")
        if ax is not None:
            xe = ax.find("xyz")
            a = np.array([float(v) for v in xe.text.split()])
            if xe.get("expressed_in") == "__model__":
                a = X_ML[J.child].R.T @ a
            J.axis = a / np.linalg.norm(a)
            J.damping = float(ax.findtext("dynamics/damping", "0"))
        joints.append(J)
    for ge in model:
        if ge.tag.endswith("collision_filter_group"):
            members = [m.text for m in ge if m.tag.endswith("member")]
            ignored = [m.text for m in ge if m.tag.endswith("ignored_collision_filter_group")]
            groups.append((ge.get("name"), members, ignored))
    # every joint of these models is actuated by the example (allegro: all 16 finger joints)
    actuated = {j.name for j in joints}
    return links, order, joints, actuated, groups, X_ML


# ---------------------------------------------------------------- tree -> tables
def planar_frame(axis):
    """Frame I of a URDF planar joint expressed in the joint frame J: z_I = axis,
    x_I = J's x (projected), y_I = z_I x x_I.  For the hopper (axis 0 -1 0, origin
    rpy 0 -pi/2 0) this gives q = [height, horizontal, pitch], the order the
    reference's example config documents (examples/hopper/hopper.yaml:7); Drake's own
    choice of in-plane axes is not derivable from the tree."""
    z = axis / np.linalg.norm(axis)
    x = np.array([1.0, 0, 0]) - z * z[0]
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.column_stack([x, y, z])


def build_model(name, links, link_order, joints, actuated, groups, spec, extra_world_geoms=(), gravity=(0, 0, -9.81),
                world_weld=None):
    """spec: dict(common=<link name or None>, paths=[[link names]])"""
    children = {}
    for J in joints:
        children.setdefault(J.parent, []).append(J)
    child_links = {J.child for J in joints}
    roots = [ln for ln in link_order if ln not in child_links and ln != "world"]

    bodies = []          # dict(name, parent, jtype, X_PF, axis, links=[(link, X_BL)], damping list, act list)
    link_body = {}       # link name -> (body index or -1, X_BL)

    def add_body(link, parent_body, jtype, X_PF, axis, damping, act):
        bodies.append(dict(name=link, parent=parent_body, jtype=jtype, X_PF=X_PF, axis=axis,
                           links=[(link, X())], damping=damping, act=act))
        return len(bodies) - 1

    def visit(link, body, X_BL):
        link_body[link] = (body, X_BL)
        for J in children.get(link, []):
            if J.type == "fixed":
                if body >= 0:
                    bodies[body]["links"].append((J.child, X_BL @ J.X_PJ))
                visit(J.child, body, X_BL @ J.X_PJ)
            else:
                act = 1 if J.name in actuated else 0
                Xpf = X_BL @ J.X_PJ
                if J.type in ("revolute", "continuous"):
                    b = add_body(J.child, body, "revolute", Xpf, J.axis, [J.damping], [act])
                    visit(J.child, b, X())
                elif J.type == "prismatic":
                    b = add_body(J.child, body, "prismatic", Xpf, J.axis, [J.damping], [act])
                    visit(J.child, b, X())
                elif J.type == "planar":
                    R_JI = planar_frame(J.axis)
                    XJI = X(R_JI, np.zeros(3))
                    # body frame stays the child link frame L (= J at q = 0): the
                    # mobilised frame M = I, so the link sits at X_MI^-1 in M.
                    b = add_body(J.child, body, "planar", Xpf @ XJI, np.array([0, 0, 1.0]), [J.damping] * 3, [act] * 3)
                    bodies[b]["links"] = [(J.child, XJI.inv())]
                    visit(J.child, b, XJI.inv())
                else:
                    raise ValueError(J.type)

    # links welded to / hanging off the world
    if "world" in children:
        visit("world", -1, X())
    for r in roots:
        if world_weld and r == world_weld[0]:
            visit(r, -1, world_weld[1])
        else:
            b = add_body(r, -1, "floating", X(), np.array([0, 0, 1.0]), [0.0] * 6, [0] * 6)
            visit(r, b, X())

    # reorder bodies depth-first is already the visiting order (parents precede children)
    nb = len(bodies)
    m = Model(name=name, gravity=np.array(gravity, float))
    m.body_names = [b["name"] for b in bodies]
    m.parent = [b["parent"] for b in bodies]
    m.jtype = [JOINT_TYPES[b["jtype"]] for b in bodies]
    m.X_PF = [b["X_PF"].flat() for b in bodies]
    m.axis = [list(b["axis"]) for b in bodies]
    mass, com, inertia = [], [], []
    for b in bodies:
        M = sum(links[l].mass for l, _ in b["links"])
        c = np.zeros(3)
        for l, Xbl in b["links"]:
            c += links[l].mass * (Xbl.p + Xbl.R @ links[l].com)
        c = c / M if M > 0 else c
        I = np.zeros((3, 3))
        for l, Xbl in b["links"]:
            L = links[l]
            d = (Xbl.p + Xbl.R @ L.com) - c
            I += Xbl.R @ L.I @ Xbl.R.T + L.mass * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        mass.append(M); com.append(list(c))
        inertia.append([I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]])
    m.mass, m.com, m.inertia = mass, com, inertia
    m.damping = [d for b in bodies for d in b["damping"]]
    m.actuated = [a for b in bodies for a in b["act"]]

    # geometries in registration order
    geoms = []
    for ln in link_order:
        if ln not in link_body:
            continue
        body, X_BL = link_body[ln]
        for (gt, sz, Xlg, reg) in links[ln].geoms:
            geoms.append((reg, body, gt, sz, X_BL @ Xlg, ln))
    for (gt, sz, Xwg, reg) in extra_world_geoms:
        geoms.append((reg, -1, gt, sz, Xwg, "world"))
    geoms.sort(key=lambda g: g[0])
    m.geom_body = [g[1] for g in geoms]
    m.geom_type = [GEOM_TYPES[g[2]] for g in geoms]
    m.geom_size = [g[3] for g in geoms]
    m.geom_X = [g[4].flat() for g in geoms]

    # star decomposition
    bidx = {b["name"]: i for i, b in enumerate(bodies)}
    m.common_body = bidx[spec["common"]] if spec.get("common") else -1
    body_path = [-2] * nb
    if m.common_body >= 0:
        body_path[m.common_body] = -1
    for p, chain in enumerate(spec["paths"]):
        for ln in chain:
            body_path[bidx[ln]] = p
    assert -2 not in body_path, f"{name}: bodies without a path: {[bodies[i]['name'] for i in range(nb) if body_path[i] == -2]}"
    m.body_path = body_path
    npaths = 1
    while npaths < len(spec["paths"]):
        npaths *= 2
    m.npaths = npaths

    # candidate pairs after Drake's default filters + declared groups
    def excluded(la, lb):
        for (gname, members, ignored) in groups:
            if la in members and lb in members and gname in ignored:
                return True
        return False

    adjacent = set()
    for i, b in enumerate(bodies):
        if b["jtype"] != "floating":   # auto-added floating joints do not filter against the world
            adjacent.add((min(i, b["parent"]), max(i, b["parent"])))
    pa, pb, pp = [], [], []
    for i in range(len(geoms)):
        for j in range(i + 1, len(geoms)):
            bi, bj = geoms[i][1], geoms[j][1]
            if bi == bj or (min(bi, bj), max(bi, bj)) in adjacent or excluded(geoms[i][5], geoms[j][5]):
                continue
            paths = {body_path[b] for b in (bi, bj) if b >= 0 and b != m.common_body}
            assert len(paths) <= 1, f"{name}: pair {geoms[i][5]}-{geoms[j][5]} spans two paths"
            pa.append(i); pb.append(j); pp.append(paths.pop() if paths else 0)
    m.pair_a, m.pair_b, m.pair_path = pa, pb, pp
    return m.normalize()


def ground_box(reg):
    """Box(25, 25, 10) centred at z = -5 on the world body (hopper.cc:44-49)."""
    g = ("box", [12.5, 12.5, 5.0], X(np.eye(3), [0, 0, -5.0]), reg[0])
    reg[0] += 1
    return g


# ---------------------------------------------------------------- models
def make_pendulum():
    """Drake's examples/pendulum/Pendulum.urdf is not in the tree; the reference's
    tests state its parameters: m = 1, l = 0.5, b = 0.1, g = 9.81, point mass
    (optimizer/test/trajectory_optimizer_test.cc:935-937, 1109-1112, 984)."""
    m = Model(name="pendulum")
    m.body_names = ["arm"]
    m.parent, m.jtype = [-1], [0]
    m.X_PF = [X().flat()]
    m.axis = [[0, 1, 0]]
    m.mass, m.com, m.inertia = [1.0], [[0, 0, -0.5]], [[0, 0, 0, 0, 0, 0]]
    m.damping, m.actuated = [0.1], [1]
    m.npaths, m.common_body, m.body_path = 1, -1, [0]
    return m.normalize()


def make_free_body():
    """A single free body (quaternion DoFs), TO_test.cc:115-178 `QuaternionDofs`."""
    m = Model(name="free_body")
    m.body_names = ["body"]
    m.parent, m.jtype = [-1], [3]
    m.X_PF = [X().flat()]
    m.axis = [[0, 0, 1]]
    m.mass, m.com, m.inertia = [1.0], [[0, 0, 0]], [[0.1, 0.1, 0.1, 0, 0, 0]]
    m.damping, m.actuated = [0.0] * 6, [0] * 6
    m.npaths, m.common_body, m.body_path = 1, -1, [0]
    return m.normalize()


def convert_all():
    os.makedirs(OUT_MODELS, exist_ok=True)
    os.makedirs(OUT_CONFIGS, exist_ok=True)
    out = {}

    out["pendulum"] = make_pendulum()
    out["free_body"] = make_free_body()

    reg = [0]
    l, o, j, a, g = parse_urdf(f"{REF}/models/acrobot/acrobot.urdf", reg)
    out["acrobot"] = build_model("acrobot", l, o, j, a, g, dict(paths=[["Link1", "Link2"]]))

    for nm, fn in (("spinner", "spinner_friction.urdf"), ("spinner_sphere", "spinner_sphere.urdf")):
        reg = [0]
        l, o, j, a, g = parse_urdf(f"{REF}/models/{fn}", reg)
        out[nm] = build_model(nm, l, o, j, a, g, dict(paths=[["finger_one", "finger_two", "spinner"]]))

    reg = [0]
    l, o, j, a, g = parse_urdf(f"{REF}/models/hopper.urdf", reg)
    out["hopper"] = build_model("hopper", l, o, j, a, g, dict(paths=[["torso", "leg", "foot"]]),
                                extra_world_geoms=[ground_box(reg)])
    # the hopper without ground, as in TO_test.cc:1540-1634 (sizes / invariants only)
    reg = [0]
    l, o, j, a, g = parse_urdf(f"{REF}/models/hopper.urdf", reg)
    out["hopper_no_ground"] = build_model("hopper_no_ground", l, o, j, a, g, dict(paths=[["torso", "leg", "foot"]]))

    reg = [0]
    l, o, j, a, g = parse_urdf(f"{REF}/models/mini_cheetah_mesh.urdf", reg)
    legs = [[f"abduct_{s}", f"thigh_{s}", f"shank_{s}"] for s in ("fl", "fr", "hl", "hr")]
    out["mini_cheetah"] = build_model("mini_cheetah", l, o, j, a, g, dict(common="body", paths=legs),
                                      extra_world_geoms=[ground_box(reg)])

    reg = [0]
    l, o, j, a, g, _ = parse_sdf(f"{REF}/models/allegro_hand.sdf", reg)
    # free ball added in code (allegro_hand.cc:99-113): m = .05, r = .06, solid sphere
    ball = Link("ball")
    ball.mass = 0.05
    ball.I = np.eye(3) * (0.4 * 0.05 * 0.06 ** 2)
    ball.geoms.append(("sphere", [0.06, 0, 0], X(), reg[0]))
    reg[0] += 1
    l["ball"] = ball
    o.append("ball")
    fingers = [[f"link_{k}" for k in range(s, s + 4)] for s in (8, 12, 4, 0)]
    X_hand = X(rpy_to_R(0, -math.pi / 2, 0), [0, 0, 0])  # allegro_hand.cc:91-93
    out["allegro_hand"] = build_model("allegro_hand", l, o, j, a, g, dict(common="ball", paths=fingers),
                                      world_weld=("hand_root", X_hand))

    for name, m in out.items():
        m.save(os.path.join(OUT_MODELS, f"{name}.model"))
        print(f"{name}: nb={m.nbodies} nq={m.nq} nv={m.nv} geoms={m.ngeoms} pairs={m.npairs} "
              f"paths={m.npaths} common={m.common_body} unactuated={m.unactuated_dofs}")

    # problem configs: the fields of the reference's example YAMLs that define the problem
    keep = ["q_init", "v_init", "q_nom_start", "q_nom_end", "q_nom_relative_to_q_init", "q_guess", "Qq", "Qv", "R",
            "Qfq", "Qfv", "time_step", "num_steps", "max_iters", "method", "linesearch", "scaling", "scaling_method",
            "equality_constraints", "normalize_quaternions", "linear_solver", "Delta0", "Delta_max", "num_threads",
            "gradients_method", "contact_stiffness", "dissipation_velocity", "smoothing_factor",
            "friction_coefficient", "stiction_velocity", "exact_hessian", "tolerances", "mpc_iters",
            "controller_frequency"]
    for name in ("acrobot", "spinner", "hopper", "mini_cheetah", "allegro_hand"):
        src = yaml.safe_load(open(f"{REF}/examples/{name}/{name}.yaml"))
        cfg = {"model": name, "source": f"reference examples/{name}/{name}.yaml"}
        for k in keep:
            if k in src:
                cfg[k] = src[k]
        with open(os.path.join(OUT_CONFIGS, f"{name}.yaml"), "w") as f:
            yaml.safe_dump(cfg, f, sort_keys=False, default_flow_style=None)
    return out


if __name__ == "__main__":
    convert_all()
