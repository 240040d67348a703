#!/usr/bin/env python3
"""Build tests/golden/rbdl_leg_id.npz from the reference's own RBDL vectors (build container only: reads
/root/reference; the fixture it writes is data, no source text).

  single_leg_test/DataFloder/PlannedData.txt          10 001 rows  q(3) qd(3) qdd(3)   -- the INPUT of the reference run,
                                                      read back by MyRobotSolver::GetLengthofPlannedData
                                                      (single_leg_test/lib/model_test_header.cpp:246-275)
  single_leg_test/DataFloder/TauofInversedynamics.txt 10 001 rows  tau(3)              -- RBDL InverseDynamics on every row,
                                                      written by MyRobotSolver::IDynamicsCalculation (:277-301)
(The folder's PositionofForward / VelocityofForward / AccelerationofForward.txt, written by FDynamicsCalculation
 (:303-345), hold 10 001 identical rows -- q = (-1.2, -1, -0.4), velocity and acceleration 0 at a configuration that is
 not an equilibrium -- so they pin nothing and are not used.)

The model both were computed on is the 3-link chain of MyRobotSolver::model_initialization (:183-222), restated here as the
numbers a qlamd_robot_model leg takes (URDF conventions: joint origin xyz / rpy about fixed axes, joints about local z):
  RBDL SpatialTransform(E, r) carries the coordinate transform E parent->child; its roty(a) / rotx(a) are the transposes
  of the active rotations, so the child frame seen from the parent is Ry(a) / Rx(a): rpy = (0, pi/2, 0), (-pi/2, 0, 0), 0.
"""
import math
import os
import sys

import numpy as np

REF = "/root/reference/single_leg_test/DataFloder"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "rbdl_leg_id.npz")


def main():
    if not os.path.isdir(REF):
        sys.exit("needs the reference tree at /root/reference")
    planned = np.loadtxt(os.path.join(REF, "PlannedData.txt"))
    tau = np.loadtxt(os.path.join(REF, "TauofInversedynamics.txt"))
    assert planned.shape == (10001, 9) and tau.shape == (10001, 3)
    model = dict(
        # model_test_header.cpp:193-216: body_a on the root, body_b at (0,0,0.1) of a, body_c at (0.25,0,0.1) of b;
        # a fourth, massless segment stands for the fixed end link a qlamd_robot_model leg has
        joint_xyz=np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.1], [0.25, 0.0, 0.1], [0.0, 0.0, 0.0]]),
        joint_rpy=np.array([[0.0, math.pi / 2, 0.0], [-math.pi / 2, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]),
        link_mass=np.array([1.17, 3.39, 1.41, 0.0]),
        link_com=np.array([[0.0, 0.0128, 0.0], [0.114, 0.0, 0.0594], [0.0949, 0.0, -0.00166], [0.0, 0.0, 0.0]]),
        # ixx ixy ixz iyy iyz izz about the centre of mass (Body(mass, com, inertia_C), :191,199,208)
        link_inertia=np.array([[0.00172, 0.0, 0.0, 0.00132, 0.0, 0.00215], [0.00302, 0.0, 0.0, 0.0269, 0.0, 0.0285],
                               [0.000547, 0.0, 0.000222, 0.0109, 0.0, 0.0111], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]]),
        gravity=np.array([0.0, 0.0, -9.81]),        # :189
    )
    np.savez_compressed(OUT, q=planned[:, 0:3], qd=planned[:, 3:6], qdd=planned[:, 6:9], tau=tau, **model)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
