#!/usr/bin/env python3
"""GPU probe: run every auxiliary entry (rows a2, a10-a18, f1-f4) a few times at batch 4096 so that
`rocprofv3 --kernel-trace --stats` reports their kernel durations (profiles/r1/aux_kernels_*)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # before the C-ABI library: both bring a HIP runtime, torch's must be the one that initialises

from quadruped_locomotion_amd import capi, synth

if "--lib" in sys.argv:  # a variant build (tools/experiments/variants.py)
    capi.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])

B, REPS = 4096, 5
ctx = capi.Context()
rng = np.random.default_rng(0)

pb = synth.make_pose_problems(B)
prm = capi.default_pose_params()
for it in (1, 5):
    prm.tolerance, prm.max_iterations = 0.0, it
    for _ in range(REPS):
        capi.pose_sqp(ctx, pb, prm)
for _ in range(REPS):
    capi.pose_qp(ctx, pb)
    capi.pose_check(ctx, pb, np.full((B, 4), 0.1), 0.0)
    capi.pose_geometric(ctx, pb)
    capi.base_auto_optimize_pose(ctx, pb, None, np.full((B, 4), 0.1), 0.0)

sw = synth.make_swing_inputs(B)
st = synth.make_states(B, "trot")
tpos = rng.uniform(-0.3, 0.3, (B, 12))
eff = np.zeros((B, 12)); el = np.zeros((B, 12)); ei = np.zeros((B, 12))
mode = rng.integers(0, 5, (B, 4)).astype(np.uint8)
for _ in range(REPS):
    capi.swing_leg_torque(ctx, sw["q"], sw["qd"], sw["qd_old"], tpos, sw["tvel"], sw["support"])
    capi.swing_branch(ctx, eff, sw["q"], sw["qd"], sw["qd_old"], tpos, sw["tvel"], sw["support"], st["base_quat"], sw["q"], mode,
                      el, ei, 0.0025)
    capi.leg_inverse_kinematics(ctx, tpos, sw["q"])

from test_leg_state import make_io
io = make_io(B)
for _ in range(REPS):
    capi.leg_state_machine(ctx, io)

from test_wire_format import batch_of_messages
blob, off, _ = batch_of_messages(B, 1)
print("wire bytes per message: %.0f" % (len(blob) / B))
for _ in range(REPS):
    capi.robot_state_unpack(ctx, blob, off)
# a stream of one layout (one publisher): after the first launch every message hits the layout template
from test_wire_format import random_message
one, _ = random_message(np.random.default_rng(3), ragged=True)
blob1 = one * B
off1 = np.arange(B + 1, dtype=np.int64) * len(one)
print("uniform stream: %d bytes per message" % len(one))
ctx_u = capi.Context()
for _ in range(REPS + 1):
    capi.robot_state_unpack(ctx_u, blob1, off1)

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "qp_goldens.npz"))
for tag in ("n12", "n6"):                      # the golden force QPs (n = 12, m = 20 and n = 6, m = 10), tiled to 4096
    Gq, g0q, CIq, ci0q = (np.tile(g[f"{tag}_{k}"], (B // 128,) + (1,) * (g[f"{tag}_{k}"].ndim - 1)) for k in ("G", "g0", "CI", "ci0"))
    for _ in range(REPS):
        capi.qp_solve(ctx, Gq, g0q, None, None, CIq, ci0q)

# the same 4096 force problems in the (A, S, b, W, D, d, f) form of the ooqpei seam: any A with A'A = G - W and
# A'b = -g0 states the same problem (S = 1, W = 1e-4 I, the reference's regulariser): take it from the eigenvectors of G - W
lam, V = np.linalg.eigh(g["n12_G"] - 1e-4 * np.eye(12))
A128 = np.sqrt(np.clip(lam, 0.0, None))[:, :, None] * np.transpose(V, (0, 2, 1))          # [128][12][12], six rows ~ 0
b128 = np.stack([-np.linalg.lstsq(A128[i].T, g["n12_g0"][i], rcond=None)[0] for i in range(128)])
Aq, bq = np.tile(A128, (B // 128, 1, 1)), np.tile(b128, (B // 128, 1))
Dq = np.tile(np.transpose(g["n12_CI"], (0, 2, 1)), (B // 128, 1, 1))
dq = np.tile(-g["n12_ci0"], (B // 128, 1))
for _ in range(REPS):
    xw, stw = capi.weighted_lsq_qp(ctx, Aq, np.ones((B, 12)), bq, np.full((B, 12), 1e-4), None, None, Dq, dq, np.full(dq.shape, capi.NO_BOUND))
print("weighted lsq vs golden x: %.2e" % np.abs(xw[:128] - g["n12_x"]).max())

for _ in range(REPS):
    ctx.balance_solve_host(st)
    ctx.balance_solve_host(synth.make_states(B, "static"))

# whole-body row (f4): dynamics kernel and the fused step, trot and static states, device-resident inputs
for gait in ("trot", "static"):
    d = capi.to_device(synth.make_wholebody_states(B, gait))
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0"); grf = torch.zeros_like(tau)
    stt = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    M = torch.zeros(B, 18, 18, dtype=torch.float64, device="cuda:0"); h = torch.zeros(B, 18, dtype=torch.float64, device="cuda:0")
    Jc = torch.zeros(B, 12, 18, dtype=torch.float64, device="cuda:0")
    for _ in range(REPS):
        capi.wholebody_solve_device(ctx, d, tau, grf, stt)
        capi.wholebody_dynamics_device(ctx, d, M, h, Jc)
    torch.cuda.synchronize()
print("done")
