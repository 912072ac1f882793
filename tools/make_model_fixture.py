#!/usr/bin/env python3
"""Cross-check fixtures for tools/convert_models.py: tests/golden/world_<model>.json.

Reads the reference's robot descriptions DIRECTLY (xml.etree, no code shared with the
converter or with idto_amd.model) and records, at the neutral configuration (every joint at
zero, floating bodies at the identity pose), what the description itself says about every link:
world pose, mass, centre of mass in the world, rotational inertia about the centre of mass in
world axes, which links are welded together (fixed joints), the world axis and anchor of every
movable joint, and every collision primitive's world pose and size.  tests/test_golden.py
recomputes the same quantities from the converted model tables (idto_amd/models/*.model: joint
frames, merged bodies, parallel-axis composites) and compares - so the converter's frame
composition, inertia rotations and welded-link merging are pinned by a second implementation.

What this can NOT pin are Drake's conventions that neither implementation can see (SURVEY.md
Appendix D): the in-plane axes of a URDF planar joint, DoF ordering, the angular-velocity frame
of a floating joint.

Semantics restated here: URDF - a joint's <origin> is the child link frame in the parent link
frame, <axis> is in the child frame, <inertial><origin> places the COM frame in the link frame
and the inertia tensor is given in that COM frame (models/hopper.urdf:14-19).  SDF 1.7 - a
link's <pose> is in the model frame, <inertial><pose> in the link frame, an axis with
expressed_in="__model__" is in the model frame (models/allegro_hand.sdf:82-137).  In-code
additions restated from the examples: ground box 25 x 25 x 10 at z = -5
(examples/hopper/hopper.cc:44-49, examples/mini_cheetah/mini_cheetah.cc:50-55); hand welded at
`hand_root` with RPY(0, -pi/2, 0) and a free ball m = 0.05, r = 0.06
(examples/allegro_hand/allegro_hand.cc:88-113).

Run in the authoring container (needs /root/reference):  python tools/make_model_fixture.py
"""
import json
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def T(R=None, p=None):
    M = np.eye(4)
    if R is not None:
        M[:3, :3] = R
    if p is not None:
        M[:3, 3] = p
    return M


def floats(text, n, default=0.0):
    v = [float(t) for t in (text or "").split()]
    return v + [default] * (n - len(v))


def urdf_origin(el):
    if el is None:
        return T()
    return T(rpy(*floats(el.get("rpy"), 3)), floats(el.get("xyz"), 3))


def sdf_pose(el):
    v = floats(el.text if el is not None else "", 6)
    return T(rpy(*v[3:6]), v[0:3])


def world_inertia(X_WC, I_C):
    R = X_WC[:3, :3]
    return R @ I_C @ R.T


def record_links(links):
    out = {}
    for name, L in links.items():
        X_WC = L["X_WL"] @ L["X_LC"]
        out[name] = dict(X_WL=L["X_WL"].tolist(), mass=L["mass"], com_W=X_WC[:3, 3].tolist(),
                         I_W=world_inertia(X_WC, L["I_C"]).tolist(), welded_to=L.get("welded_to"),
                         geoms=[dict(type=g["type"], size=g["size"], X_WG=(L["X_WL"] @ g["X_LG"]).tolist())
                                for g in L["geoms"]])
    return out


def parse_xml(path):
    # (the files use `drake:` element prefixes without declaring the namespace)
    return ET.fromstring(open(path).read().replace("<drake:", "<drake_").replace("</drake:", "</drake_"))


def read_urdf(path):
    root = parse_xml(path)
    links, joints = {}, []
    for le in root.findall("link"):
        ine = le.find("inertial")
        mass, I, X_LC = 0.0, np.zeros((3, 3)), T()
        if ine is not None:
            mass = float(ine.find("mass").get("value"))
            X_LC = urdf_origin(ine.find("origin"))
            i = ine.find("inertia")
            g = lambda k: float(i.get(k, "0"))
            I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        geoms = []
        for ce in le.findall("collision"):
            ge = ce.find("geometry")
            if ge.find("sphere") is not None:
                geoms.append(dict(type="sphere", size=[float(ge.find("sphere").get("radius"))], X_LG=urdf_origin(ce.find("origin"))))
            elif ge.find("box") is not None:
                geoms.append(dict(type="box", size=floats(ge.find("box").get("size"), 3), X_LG=urdf_origin(ce.find("origin"))))
        links[le.get("name")] = dict(mass=mass, I_C=I, X_LC=X_LC, geoms=geoms, X_WL=None)
    for je in root.findall("joint"):
        ax = je.find("axis")
        joints.append(dict(name=je.get("name"), type=je.get("type"), parent=je.find("parent").get("link"),
                           child=je.find("child").get("link"), X_PJ=urdf_origin(je.find("origin")),
                           axis=floats(ax.get("xyz"), 3) if ax is not None else [1.0, 0, 0]))
    # neutral configuration: every joint transform is the identity, the child frame is the joint frame
    children = {j["child"] for j in joints}
    for name in links:
        if name not in children:
            links[name]["X_WL"] = T()   # root link (welded to the world by a fixed joint or floating at the identity)
    links.setdefault("world", dict(mass=0.0, I_C=np.zeros((3, 3)), X_LC=T(), geoms=[], X_WL=T()))
    pending = list(joints)
    while pending:
        rest = []
        for j in pending:
            P = links[j["parent"]]
            if P["X_WL"] is None:
                rest.append(j)
                continue
            links[j["child"]]["X_WL"] = P["X_WL"] @ j["X_PJ"]
            if j["type"] == "fixed":
                links[j["child"]]["welded_to"] = j["parent"]
        assert len(rest) < len(pending), "kinematic loop"
        pending = rest
    jout = []
    for j in joints:
        if j["type"] in ("fixed",):
            continue
        X = links[j["child"]]["X_WL"]
        jout.append(dict(name=j["name"], type=j["type"], child=j["child"], parent=j["parent"],
                         axis_W=(X[:3, :3] @ np.array(j["axis"])).tolist(), anchor_W=X[:3, 3].tolist()))
    links.pop("world")
    return record_links(links), jout


def read_sdf(path, weld_link, X_W_weld):
    root = parse_xml(path)
    model = root.find("model")
    raw = {}
    for le in model.findall("link"):
        ine = le.find("inertial")
        mass, I, X_LC = 0.0, np.zeros((3, 3)), T()
        if ine is not None:
            mass = float(ine.find("mass").text)
            X_LC = sdf_pose(ine.find("pose"))
            i = ine.find("inertia")
            g = lambda k: float(i.find(k).text) if i is not None and i.find(k) is not None else 0.0
            I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        geoms = []
        for ce in le.findall("collision"):
            ge = ce.find("geometry")
            # (the palm's <geometry> lists a sphere AND a box, models/allegro_hand.sdf:49-56; libsdformat's
            # Geometry::Load tests <box> first, so the box is what the plant sees)
            if ge.find("box") is not None:
                geoms.append(dict(type="box", size=floats(ge.find("box").find("size").text, 3), X_LG=sdf_pose(ce.find("pose"))))
            elif ge.find("sphere") is not None:
                geoms.append(dict(type="sphere", size=[float(ge.find("sphere").find("radius").text)], X_LG=sdf_pose(ce.find("pose"))))
        raw[le.get("name")] = dict(mass=mass, I_C=I, X_LC=X_LC, geoms=geoms, X_ML=sdf_pose(le.find("pose")))
    X_WM = X_W_weld @ np.linalg.inv(raw[weld_link]["X_ML"])
    for L in raw.values():
        L["X_WL"] = X_WM @ L["X_ML"]
    raw[weld_link]["welded_to"] = "world"
    jout = []
    for je in model.findall("joint"):
        if je.get("type") == "fixed":
            raw[je.find("child").text]["welded_to"] = je.find("parent").text
            continue
        xe = je.find("axis").find("xyz")
        a = np.array(floats(xe.text, 3))
        child = je.find("child").text
        if xe.get("expressed_in") == "__model__":
            aW = X_WM[:3, :3] @ a
        else:
            aW = raw[child]["X_WL"][:3, :3] @ a
        X_WJ = raw[child]["X_WL"] @ sdf_pose(je.find("pose"))
        jout.append(dict(name=je.get("name"), type=je.get("type"), child=child, parent=je.find("parent").text,
                         axis_W=aW.tolist(), anchor_W=X_WJ[:3, 3].tolist()))
    return record_links(raw), jout


def ground():
    return dict(type="box", size=[25.0, 25.0, 10.0], X_WG=T(p=[0, 0, -5.0]).tolist())


def main():
    os.makedirs(OUT, exist_ok=True)
    out = {}
    for name, fn, world_geoms in (("acrobot", "acrobot/acrobot.urdf", []), ("spinner", "spinner_friction.urdf", []),
                                  ("hopper", "hopper.urdf", [ground()]), ("mini_cheetah", "mini_cheetah_mesh.urdf", [ground()])):
        links, joints = read_urdf(f"{REF}/models/{fn}")
        out[name] = dict(source=f"reference models/{fn}", links=links, joints=joints, world_geoms=world_geoms)
    links, joints = read_sdf(f"{REF}/models/allegro_hand.sdf", "hand_root", T(rpy(0, -math.pi / 2, 0)))
    r, m = 0.06, 0.05
    links["ball"] = dict(X_WL=T().tolist(), mass=m, com_W=[0, 0, 0], I_W=(np.eye(3) * 0.4 * m * r * r).tolist(),
                         welded_to=None, geoms=[dict(type="sphere", size=[r], X_WG=T().tolist())])
    out["allegro_hand"] = dict(source="reference models/allegro_hand.sdf + examples/allegro_hand/allegro_hand.cc:88-113",
                               links=links, joints=joints, world_geoms=[])
    for name, d in out.items():
        d["generator"] = "tools/make_model_fixture.py"
        with open(os.path.join(OUT, f"world_{name}.json"), "w") as f:
            json.dump(d, f, indent=1)
        print(name, len(d["links"]), "links,", len(d["joints"]), "movable joints, total mass",
              sum(L["mass"] for L in d["links"].values()))


if __name__ == "__main__":
    main()
