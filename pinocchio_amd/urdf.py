"""Minimal URDF -> Model converter.

Honours the rules of pinocchio's URDF parser that decide the constant tables
(reference ``src/parsers/urdf/model.cpp`` and
``include/pinocchio/parsers/urdf/model.hxx``), without urdfdom:

* optional free-flyer root joint at identity carrying the root link's inertia
  (``parsers/urdf/model.hxx:113-146``);
* depth-first recursion over child links (``src/parsers/urdf/model.cpp:288-291``);
  sibling order is urdfdom's: ``ModelInterface::joints_`` is a ``std::map`` keyed by
  joint name, so children are visited in alphabetical order of the JOINT name
  (urdfdom is third-party and absent from /root/reference; any sibling order
  gives a valid model, only the dof numbering changes);
* REVOLUTE with axis exactly +X/+Y/+Z -> RX/RY/RZ, anything else ->
  RevoluteUnaligned(axis.normalized()) (``model.hxx:493-527,619-626``);
  CONTINUOUS -> RUBX/RUBY/RUBZ, RevoluteUnboundedUnaligned otherwise (``src/parsers/urdf/model.cpp:164-192``);
  PRISMATIC likewise; FIXED -> no joint, the child's inertia is merged into the
  parent joint's body through the frame placement (``model.hxx:331-346``);
* URDF inertial -> Inertia(mass, com, R I R^T) (``src/parsers/urdf/model.cpp:30-42``).
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np

from . import model as M


def _floats(s: Optional[str], n: int, default: float = 0.0) -> np.ndarray:
    if s is None:
        return np.full(n, default)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    assert v.size == n, f"expected {n} numbers, got {s!r}"
    return v


def _rpy_to_R(rpy: np.ndarray) -> np.ndarray:
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _origin(el) -> np.ndarray:
    T = np.eye(4)
    if el is None:
        return T
    o = el.find("origin")
    if o is None:
        return T
    T[:3, :3] = _rpy_to_R(_floats(o.get("rpy"), 3))
    T[:3, 3] = _floats(o.get("xyz"), 3)
    return T


def _link_inertia(link) -> M.Inertia:
    inert = link.find("inertial")
    if inert is None:
        return M.Inertia.Zero()
    T = _origin(inert)
    mass = float(inert.find("mass").get("value"))
    i = inert.find("inertia")
    g = lambda k: float(i.get(k, "0"))
    I = np.array([[g("ixx"), g("ixy"), g("ixz")],
                  [g("ixy"), g("iyy"), g("iyz")],
                  [g("ixz"), g("iyz"), g("izz")]])
    R = T[:3, :3]
    return M.Inertia(mass, T[:3, 3], R @ I @ R.T)


_AXES = {(1.0, 0.0, 0.0): 0, (0.0, 1.0, 0.0): 1, (0.0, 0.0, 1.0): 2}


def build_model_from_urdf(text: str, free_flyer: bool = True, name: Optional[str] = None) -> M.Model:
    root = ET.fromstring(text)
    links = {l.get("name"): l for l in root.findall("link")}
    joints = {j.get("name"): j for j in root.findall("joint")}
    children: Dict[str, List[str]] = {k: [] for k in links}
    has_parent = set()
    for jn in sorted(joints):  # std::map order of urdfdom's joints_
        j = joints[jn]
        children[j.find("parent").get("link")].append(jn)
        has_parent.add(j.find("child").get("link"))
    roots = [k for k in links if k not in has_parent]
    if len(roots) != 1:
        raise ValueError(f"URDF must have exactly one root link, found {roots}")
    model = M.Model(name or root.get("name", "urdf"))

    root_link = roots[0]
    if free_flyer:
        jid = model.addJoint(0, M.JOINT_FREEFLYER, np.eye(4), "root_joint")
        model.appendBodyToJoint(jid, _link_inertia(links[root_link]), np.eye(4))
        start = jid
    else:
        start = 0  # root link welded to the universe: its inertia is dropped like pinocchio's universe body

    def visit(link_name: str, parent_joint: int, link_in_joint: np.ndarray):
        for jn in children[link_name]:
            j = joints[jn]
            child = j.find("child").get("link")
            T = link_in_joint @ _origin(j)
            jt = j.get("type")
            Y = _link_inertia(links[child])
            if jt == "fixed":
                if parent_joint > 0:
                    model.appendBodyToJoint(parent_joint, Y, T)
                visit(child, parent_joint, T)
                continue
            ax_el = j.find("axis")
            axis = _floats(ax_el.get("xyz") if ax_el is not None else "1 0 0", 3)
            k = _AXES.get(tuple(float(x) for x in axis))
            if jt == "revolute":
                jtype = M.JOINT_RX + k if k is not None else M.JOINT_REVOLUTE_UNALIGNED
            elif jt == "continuous":
                jtype = M.JOINT_RUBX + k if k is not None else M.JOINT_REVOLUTE_UNBOUNDED_UNALIGNED
            elif jt == "prismatic":
                jtype = M.JOINT_PX + k if k is not None else M.JOINT_PRISMATIC_UNALIGNED
            elif jt == "floating":
                jtype = M.JOINT_FREEFLYER
            else:
                raise ValueError(f"joint {jn}: URDF joint type {jt!r} is not supported")
            jid = model.addJoint(parent_joint, jtype, T, jn, axis=axis)
            model.appendBodyToJoint(jid, Y, np.eye(4))
            visit(child, jid, np.eye(4))

    visit(root_link, start, np.eye(4))
    return model
