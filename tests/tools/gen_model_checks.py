#!/usr/bin/env python3
"""Generate tests/golden/model_checks.npz: leg forward kinematics, translation Jacobian and gravity torque evaluated
straight from the reference's URDF by a small numpy program that shares no code with oracle/oracle_model.c or with
include/qlamd_robot_constants.h.

The reference gets these quantities from orocos-KDL (quadrupedkinematics.cpp:143-278,485-552: ChainFkSolverPos_recursive,
ChainJntToJacSolver, ChainDynParam::JntToGravity on the chains base_link -> *_foot_Link, :67-70), which is absent here.
This script is a second, independent reading of the same model:
  * the URDF (quadruped_model/urdf/quadruped_model.urdf) is parsed with xml.etree, chains are followed by link / joint
    names, every joint is a 4x4 homogeneous transform  T(xyz) Rz(yaw) Ry(pitch) Rx(roll) Rot(axis, q)  (URDF rpy = fixed
    axis XYZ, the literal 1.5708 / 3.1416 kept as written);
  * foot position = translation of the product; Jacobian column i = a_i x (p_foot - p_i) from the same frames;
  * gravity torque from the potential energy  U(q) = - sum_links m_l g . c_l(q)  (fixed foot link included, base link not):
    tau_i = dU/dq_i, differentiated with complex-step arithmetic (exact to rounding: no subtraction of nearby numbers).
Runs only in the build container (reads /root/reference); only the .npz travels.
"""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
URDF = "/root/reference/quadruped_model/urdf/quadruped_model.urdf"
FEET = ("lf_foot_Link", "rf_foot_Link", "rh_foot_Link", "lh_foot_Link")  # limb ids LF, RF, RH, LH; quadrupedkinematics.cpp:67-70


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    x, y, z = axis
    K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=complex)
    return np.eye(3, dtype=complex) * c + s * K + (1 - c) * np.outer(axis, axis)


def origin_T(elem):
    xyz = [float(v) for v in elem.attrib.get("xyz", "0 0 0").split()] if elem is not None else [0, 0, 0]
    r, p, y = [float(v) for v in elem.attrib.get("rpy", "0 0 0").split()] if elem is not None else [0, 0, 0]
    T = np.eye(4, dtype=complex)
    T[:3, :3] = rot((0, 0, 1), y) @ rot((0, 1, 0), p) @ rot((1, 0, 0), r)
    T[:3, 3] = xyz
    return T


def load_chains():
    root = ET.parse(URDF).getroot()
    links = {l.attrib["name"]: l for l in root.findall("link")}
    by_child = {j.find("child").attrib["link"]: j for j in root.findall("joint")}
    chains = []
    for foot in FEET:
        segs, link = [], foot
        while link != "base_link":
            j = by_child[link]
            inert = links[link].find("inertial")
            mass = float(inert.find("mass").attrib["value"])
            com = np.array([float(v) for v in inert.find("origin").attrib["xyz"].split()])
            axis = np.array([float(v) for v in j.find("axis").attrib["xyz"].split()]) if j.find("axis") is not None else None
            segs.append(dict(T0=origin_T(j.find("origin")), revolute=j.attrib["type"] in ("revolute", "continuous"), axis=axis,
                             mass=mass, com=com))
            link = j.find("parent").attrib["link"]
        chains.append(segs[::-1])
    return chains


def frames(chain, q):
    """Frames after each segment (joint rotation included) and the joint frames (origin, axis in base coordinates)."""
    T = np.eye(4, dtype=complex)
    out, joints, k = [], [], 0
    for s in chain:
        T = T @ s["T0"]
        if s["revolute"]:
            joints.append((T[:3, 3].copy(), T[:3, :3] @ s["axis"]))
            R = np.eye(4, dtype=complex)
            R[:3, :3] = rot(s["axis"], q[k])
            T = T @ R
            k += 1
        out.append(T.copy())
    return out, joints


def potential(chain, q, g):
    fr, _ = frames(chain, q)
    return -sum(s["mass"] * (g @ (T[:3, :3] @ s["com"] + T[:3, 3])) for s, T in zip(chain, fr))


def main():
    chains = load_chains()
    rng = np.random.default_rng(20261002)
    N = 96
    q = rng.uniform(-1.6, 1.6, (N, 12))
    q[:32] = np.tile([0.0, 0.75, -1.5], 4) + rng.uniform(-0.15, 0.15, (32, 12))   # around the bench posture
    q[32] = 0.0                                                                     # kinematicsTest.cpp's q = 0
    gdir = rng.normal(size=(N, 3))
    g = 9.8 * gdir / np.linalg.norm(gdir, axis=1, keepdims=True)
    g[32] = (0.0, 0.0, -9.8)
    foot, jac, grav = np.zeros((N, 4, 3)), np.zeros((N, 4, 3, 3)), np.zeros((N, 4, 3))
    h = 1e-30
    for n in range(N):
        for l, chain in enumerate(chains):
            ql = q[n, 3 * l:3 * l + 3].astype(complex)
            fr, joints = frames(chain, ql)
            p = fr[-1][:3, 3]
            foot[n, l] = p.real
            for i, (pi, ai) in enumerate(joints):
                jac[n, l, :, i] = np.cross(ai, p - pi).real
            for i in range(3):
                qc = ql.copy()
                qc[i] += 1j * h
                grav[n, l, i] = potential(chain, qc, g[n]).imag / h
    dst = os.path.join(ROOT, "tests", "golden", "model_checks.npz")
    np.savez_compressed(dst, q=q, gravity_in_base=g, foot=foot, jacobian=jac, gravity_torque=grav)
    print("wrote", dst, "; q = 0: LF foot", foot[32, 0], " gravity torque", grav[32, 0])


if __name__ == "__main__":
    main()
