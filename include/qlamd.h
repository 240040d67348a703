/* qlamd -- C-ABI of the MI355X-native batched balance-controller solve.
 *
 * Plain C types only, caller-owned buffers, integer return codes, no
 * exceptions across the boundary, no hidden globals.
 * Every entry point names the reference interface it replaces; paths are
 * relative to the reference tree (ShunyaoWang/quadruped_locomotion).
 *
 * The library needs a HIP device (gfx950).  There is NO CPU fallback: without
 * a usable device qlamd_context_create() fails with QLAMD_ERR_NO_DEVICE.
 *
 * Threads and streams.  A context is a single-threaded object, like the reference's controller whose
 * update() and callbacks share one mutex (ros_balance_controller.hpp:142): it owns scratch memory (staging
 * slab, the intermediates of the whole tick, the layout template of the message parser) that every call
 * reuses.  Two calls on one context must not overlap: a second thread entering while a call is in progress
 * gets QLAMD_ERR_BUSY.  QLAMD_MEM_DEVICE calls are asynchronous on `stream`; work queued on one context must
 * stay ordered (calls share its scratch memory).  Calls on one stream are ordered by the stream.  The first call that
 * arrives on another stream drains the device once; from then on every call records an event of the context's own
 * behind its work and a call on another stream makes that stream wait for the event -- on the device, the host does not
 * block, and no stream handle is kept once its call has returned.  While a stream is being captured into a hipGraph
 * none of this happens: the capturing caller orders the graph itself.  Use one context per thread / per concurrent
 * stream.
 */
#ifndef QLAMD_H
#define QLAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QLAMD_VERSION_MAJOR 0
#define QLAMD_VERSION_MINOR 6

/* ---- return codes of the API calls ------------------------------------- */
#define QLAMD_OK 0
#define QLAMD_ERR_INVALID_ARGUMENT (-1)
#define QLAMD_ERR_NO_DEVICE (-2)      /* no HIP device / wrong architecture   */
#define QLAMD_ERR_HIP (-3)            /* a HIP runtime call failed            */
#define QLAMD_ERR_NOT_LOADED (-4)     /* parameters not loaded: the reference's
                                         checkIfParametersLoaded() == false,
                                         ContactForceDistribution.cpp:103     */
#define QLAMD_ERR_OUT_OF_MEMORY (-5)
#define QLAMD_ERR_BUSY (-6)           /* another thread is inside a call on this context */
#define QLAMD_ERR_NEEDS_RESERVE (-7)  /* the call would have to grow the context's device scratch while its stream is
                                         being captured into a hipGraph (an allocation would break the capture): call
                                         qlamd_reserve(ctx, max_batch) before the capture starts */

/* ---- per-robot status words (int32) ------------------------------------ */
#define QLAMD_STATUS_OK 0
#define QLAMD_STATUS_INFEASIBLE 1     /* QuadProg++ returns +inf, QuadProg++.cc:339-344 */
#define QLAMD_STATUS_NOT_PD 2         /* QuadProg++ throws logic_error, QuadProg++.cc:692-699 */
#define QLAMD_STATUS_MAX_ITER 3       /* iteration guard hit (no reference counterpart) */
#define QLAMD_STATUS_NO_COMMAND 4     /* whole tick only: no well-formed command message has reached this robot yet;
                                         nothing of the robot was read or written (the reference's update() runs on
                                         the last command baseCommandCallback stored, ros_balance_controller.cpp:761) */
#define QLAMD_STATUS_WARM_REJECTED 6     /* warm start with QLAMD_OPT_WARM_FALLBACK 0 only (qlamd_placement::prev_working_set): the set
                                         handed in was so far from this robot's state that the answer reached from it failed the
                                         final check (every row holds, every multiplier non-negative, to 1e-6); outputs as for
                                         any failed solve, the robot's working_set comes back 0, so its next step starts cold.
                                         By default (QLAMD_OPT_WARM_FALLBACK 1) such a robot is solved again from the empty
                                         working set inside the same launch and this status is never returned: a warm start
                                         costs time at worst, never an answer (the reference ends every update() with efforts
                                         for a solvable QP, ros_balance_controller.cpp:418-454).  Never seen with a robot's own
                                         previous set (9 M soaked control steps); QLAMD_COUNTER_WARM_RETRIES counts them */
#define QLAMD_STATUS_DEPENDENT_EQUALITY 5 /* qlamd_qp_solve_batch with p = 2 only: the second equality column is all-zero
                                         or linearly dependent on the first and was left out; x is the solution of the
                                         problem without it.  (solve_quadprog ignores add_constraint's result there,
                                         QuadProg++.cc:203-209, and carries on with R_qq ~ 0: that artefact is not
                                         reproduced, it is reported.) */

/* ---- where caller buffers live ------------------------------------------ */
#define QLAMD_MEM_DEVICE 0            /* device pointers, used in place, async on `stream` */
#define QLAMD_MEM_HOST 1              /* host pointers, staged through the context's device
                                         workspace; the call returns after the copy-back    */

typedef struct qlamd_context qlamd_context;

/* Controller parameters.  Replaces the ROS parameter loads of
 *   VirtualModelController::loadParameters   balance_controller/src/motion_control/VirtualModelController.cpp:429-548
 *   ContactForceDistribution::loadParameters balance_controller/src/contact_force_distribution/ContactForceDistribution.cpp:818-886
 * plus the constants hard-wired in quadruped_model/src/quadruped_state.cpp:26-45,83-97
 * and the +-300 clamp of ros_balance_controller.cpp:451-454. */
typedef struct qlamd_balance_params {
  double kp_trans[3], kd_trans[3], kff_trans[3]; /* heading, lateral, vertical */
  double kp_rot[3], kd_rot[3], kff_rot[3];       /* roll, pitch, yaw           */
  double force_weights[6];                       /* S = diag(force xyz, torque xyz) */
  double regularizer;                            /* W = regularizer * I        */
  double friction;                               /* mu                          */
  double min_normal_force;                       /* f_min                       */
  double torque_limit;                           /* 300                         */
  double torso_mass;                             /* 27.0                        */
  double leg_mass[4];                            /* 6.0 each, LF RF RH LH       */
  double gravity;                                /* 9.8; g_W = (0,0,-gravity)   */
  double grav_comp_percentage;                   /* 1.0                         */
  double com_in_base[3];                         /* (0,0,0)                     */
  double hip_in_base[4][3];                      /* (+-0.42, +-0.075, 0)        */
} qlamd_balance_params;

/* Fills the values of balance_controller/config/controller_gains.yaml:1-41 and
 * quadruped_state.cpp:26-45,83-97. */
void qlamd_balance_default_params(qlamd_balance_params *p);

/* Leg chains base_link -> *_foot_Link (quadrupedkinematics.cpp:67-70): per
 * [leg][segment] joint origin, fixed rpy, link mass and centre of mass.
 * Segments 0..2 are revolute about local z, segment 3 is the fixed foot.
 * Replaces kdl_parser::treeFromFile + tree_.getChain
 * (quadrupedkinematics.cpp:54-74). */
typedef struct qlamd_robot_model {
  double joint_xyz[4][4][3];
  double joint_rpy[4][4][3];
  double link_mass[4][4];
  double link_com[4][4][3];
  double link_inertia[4][4][6]; /* about the com, link frame: ixx ixy ixz iyy iyz izz (swing-leg dynamics) */
  /* base_link inertial (urdf/quadruped_model.urdf:7-23): only the whole-body (floating-base) entries read it */
  double base_mass;
  double base_com[3];
  double base_inertia[6];       /* about the com, base frame: ixx ixy ixz iyy iyz izz */
} qlamd_robot_model;

/* The reference robot, quadruped_model/urdf/quadruped_model.urdf. */
void qlamd_default_robot_model(qlamd_robot_model *m);

/* One batch of robot states, every field laid out [batch][k], k fastest.
 * Field meaning and order mirror hardware_interface::RobotStateHandle::Data
 * (balance_controller/include/balance_controller/ros_controler/robot_state_interface.hpp:28-65)
 * and the desired base state that baseCommandCallback stores
 * (ros_balance_controller.cpp:761-1083, written into State at :384-387).
 * Joint order LF(0-2) RF(3-5) RH(6-8) LH(9-11) (quadruped_state.cpp:340-343);
 * quaternions are (w,x,y,z), base -> world. */
typedef struct qlamd_state_batch {
  const double *joint_position;       /* [B][12] */
  const double *base_position;        /* [B][3]  world                        */
  const double *base_orientation;     /* [B][4]                               */
  const double *base_linear_velocity; /* [B][3]  world                        */
  const double *base_angular_velocity;/* [B][3]  base                         */
  const double *desired_position;     /* [B][3]                               */
  const double *desired_orientation;  /* [B][4]                               */
  const double *desired_linear_velocity;  /* [B][3]                           */
  const double *desired_angular_velocity; /* [B][3]                           */
  const uint8_t *support_leg;         /* [B][4]  State::isSupportLeg, != 0 = stance */
  const double *surface_normal;       /* [B][4][3] world, or NULL: the reference's
                                         update() override n_W = q.rotate(z),
                                         ros_balance_controller.cpp:378        */
} qlamd_state_batch;

/* The record of QLAMD_STATE_RECORDS (QLAMD_OPT_STATE_LAYOUT): the double fields of one robot, quaternions on 16-byte boundaries,
 * 48 doubles = three 128-byte lines. */
#define QLAMD_STATE_RECORD_DOUBLES 48
typedef struct qlamd_state_record {
  double joint_position[12];
  double base_position[3], pad0;
  double base_orientation[4];
  double base_linear_velocity[3];
  double base_angular_velocity[3];
  double desired_position[3], pad1;
  double desired_orientation[4];
  double desired_linear_velocity[3];
  double desired_angular_velocity[3];
  double pad2[8];
} qlamd_state_record;

/* Create / destroy.  `device` is a HIP device ordinal.  `model` may be NULL
 * (reference robot).  Replaces RosBalanceController::init's construction of
 * ContactForceDistribution + VirtualModelController + State
 * (ros_balance_controller.cpp:68-95). */
int qlamd_context_create(const qlamd_balance_params *params, const qlamd_robot_model *model,
                         int device, qlamd_context **out);
void qlamd_context_destroy(qlamd_context *ctx);

/* Device scratch of the context, allocated up front.  qlamd_full_tick_batch and qlamd_robot_state_unpack_batch keep
 * per-robot intermediates (leg state codes, the command in force when the caller keeps none, the message-layout template)
 * in device memory owned by the context and grow it on demand -- hipFree + hipMalloc inside the call, which synchronises
 * the device and is illegal while the call's stream is being captured.  qlamd_reserve sizes that scratch for batches up
 * to max_batch once (synchronous; call it before the first capture or before a latency-critical loop); a call that
 * would still have to grow it inside a capture returns QLAMD_ERR_NEEDS_RESERVE instead of breaking the capture.
 * QLAMD_MEM_HOST calls stage through a slab of their own that also grows on demand: host-buffer calls synchronise
 * anyway and cannot be captured.  qlamd_reserve itself must not be called while a stream is being captured (it would
 * have to synchronise the device): it then returns QLAMD_ERR_NEEDS_RESERVE and leaves the scratch as it was.
 * Capturing the whole tick: run ONE eager tick first.  The message parser keeps the field layout of the last message it
 * walked as a template for the next launch; a captured launch reads, on every replay, the template that was
 * in force when it was captured, i.e. the one the last eager launch left -- without one the replays walk every message
 * (about 10 us slower per 4096-robot tick, results identical). */
int qlamd_reserve(qlamd_context *ctx, int64_t max_batch);

/* Which implementation of the balance step runs: robots per wavefront.  0 (default) and 4 = the lane-cooperative kernel
 * (16 lanes per robot), the fast path at every batch size measured (1 K ... 1 M robots).  16 / 64 = one lane per robot:
 * an independent second implementation (range-space Goldfarb-Idnani, Cholesky in registers) kept as a CROSS-CHECK of the
 * first, not as a fast path -- it is slower everywhere, and with per-leg surface normals it needs more than the 512
 * registers a lane can have and spills 384 / 480 bytes per lane to scratch memory. */
int qlamd_set_robots_per_wave(qlamd_context *ctx, int robots_per_wave);

/* Context options (set once after creation or between calls; never read from the process environment).
 *   QLAMD_OPT_ON_FAILURE   what a robot whose QP solve fails (status != QLAMD_STATUS_OK) gets in joint_effort /
 *                          contact_force: QLAMD_ON_FAILURE_ZERO (default) writes 0 for all 12 entries -- a caller
 *                          that ignores status[] then commands a limp leg, so check status[]; QLAMD_ON_FAILURE_KEEP
 *                          leaves the robot's entries untouched: with one effort array reused tick after tick that is
 *                          the reference's behaviour, whose update() logs "VMC compute failed" and commands the
 *                          efforts still held in State from the previous tick (ros_balance_controller.cpp:418-424,441-454).
 *                          Applies to qlamd_balance_solve_batch, qlamd_force_distribution_batch,
 *                          qlamd_wholebody_solve_batch and qlamd_full_tick_batch, with QLAMD_MEM_HOST (the caller's
 *                          arrays are read as well as written then) and QLAMD_MEM_DEVICE alike.
 *   QLAMD_OPT_REFINE_PASSES  refinement passes of the lane-cooperative force QP on its final working set (default 1)
 *   QLAMD_OPT_DYNAMICS_FORM  layout of qlamd_wholebody_dynamics_batch's kernel: QLAMD_DYNAMICS_AUTO (default) picks by batch
 *                          size; _LEG = one lane per leg, 16 robots per wavefront (throughput: large batches); _ROW = 16
 *                          lanes per robot (latency: small batches).  Same results to rounding.
 *   QLAMD_OPT_PLACEMENT_WAIT  how long the wavefronts that make qlamd_placement::next_robot_order inside a placed launch wait
 *                          for each other before they give up, in polls of about 0.1 us (default 2^13: about a millisecond,
 *                          less than half a control period at 400 Hz, balance_controller_manager.cpp:48; 0: not at all).
 *                          They normally meet within microseconds -- the barrier relies on the leading workgroups of a grid
 *                          being dispatched together, in order, which holds on a device the launch has to itself; on a device
 *                          shared with other work (RCCL kernels of another stream, another process) it may not.  A launch that
 *                          gives up leaves the identity order in next_robot_order -- a valid placement: the next launch is
 *                          slower, no result changes -- and is counted (QLAMD_COUNTER_PLACEMENT_GIVE_UPS).
 *   QLAMD_OPT_WARM_FALLBACK  1 (default): a robot whose warm start is rejected (QLAMD_STATUS_WARM_REJECTED) is solved again from
 *                          the empty working set by its own wavefront before the launch ends -- balance / force-distribution,
 *                          whole-body and whole-tick entries alike; costs one scalar branch per wavefront when nobody is
 *                          rejected.  0: the rejection is reported instead (the behaviour of version 0.5).  2: diagnostics --
 *                          every robot that ends a warm-started solve with a non-empty working set is treated as rejected and
 *                          solved a second time (how the tests reach the second attempt at will; rejections of their own are
 *                          rare: none in 9 M soaked control steps).
 *   QLAMD_OPT_STATE_LAYOUT   how the nine double fields of qlamd_state_batch lie in DEVICE memory for qlamd_balance_solve_batch /
 *                          _placed_batch (lane-cooperative kernels only; the force-distribution entries, which read two of the
 *                          fields, refuse the option with QLAMD_ERR_INVALID_ARGUMENT):
 *                          QLAMD_STATE_FIELDS (default): one array [B][k] per field, as the reference's hardware interface
 *                          hands them out (robot_state_interface.hpp:28-65).  QLAMD_STATE_RECORDS: one record of
 *                          QLAMD_STATE_RECORD_DOUBLES doubles per robot (qlamd_state_record below) -- every field pointer of
 *                          qlamd_state_batch then addresses the same [B] array of records at its field's offset
 *                          (joint_position = &rec[0].joint_position[0], ...), field k of robot i at pointer + 48 i + k.  A
 *                          launch in a placement gathers robots from all over the batch: three full 128-byte lines per robot
 *                          instead of nine partially used ones (fetched bytes 2.44 -> 1.27 MB per 4096-robot launch, 38 -> 17
 *                          MB at 65 536; 1 % of a launch at 4096 robots, 2.7 % at 65 536: profiles/r6/packed_record.txt).
 *                          support_leg, surface_normal and virtual_wrench stay arrays of their own; host-buffer calls take
 *                          per-field arrays whatever the option says. */
#define QLAMD_OPT_ON_FAILURE 1
#define QLAMD_OPT_REFINE_PASSES 2
#define QLAMD_OPT_DYNAMICS_FORM 5
#define QLAMD_OPT_PLACEMENT_WAIT 6
#define QLAMD_OPT_WARM_FALLBACK 7
#define QLAMD_OPT_STATE_LAYOUT 8
#define QLAMD_STATE_FIELDS 0
#define QLAMD_STATE_RECORDS 1
#define QLAMD_ON_FAILURE_ZERO 0
#define QLAMD_ON_FAILURE_KEEP 1
#define QLAMD_DYNAMICS_AUTO 0
#define QLAMD_DYNAMICS_LEG 1
#define QLAMD_DYNAMICS_ROW 2
int qlamd_set_option(qlamd_context *ctx, int option, int value);

/* Event counters of the context, kept on the device and read back here (the call waits for the device): how often something
 * that costs time but never a result has happened since the context was created.
 *   QLAMD_COUNTER_PLACEMENT_GIVE_UPS  placed launches whose shadow wavefronts gave up waiting for each other and left the
 *                                     identity order in next_robot_order (QLAMD_OPT_PLACEMENT_WAIT) -- on a shared device the
 *                                     placement gain can be lost silently otherwise
 *   QLAMD_COUNTER_WARM_RETRIES        robots whose warm start was rejected and that were solved again cold by the same launch
 *                                     (QLAMD_OPT_WARM_FALLBACK 1), or reported as QLAMD_STATUS_WARM_REJECTED (0) */
#define QLAMD_COUNTER_PLACEMENT_GIVE_UPS 0
#define QLAMD_COUNTER_WARM_RETRIES 1
int qlamd_get_counter(qlamd_context *ctx, int counter, int64_t *value);

/* One control step for `batch` robots: virtual-model wrench -> leg FK ->
 * contact-force-distribution QP -> joint torques, clamped.
 * Replaces VirtualModelController::compute() (VirtualModelController.cpp:89-102)
 * -> ContactForceDistribution::computeForceDistribution() (ContactForceDistribution.cpp:99-136)
 * and the effort read-out + clamp of RosBalanceController::update
 * (ros_balance_controller.cpp:441-454).
 *   joint_effort   [B][12]  out, clamped to +-torque_limit, 0 for non-support legs; for a robot whose status is
 *                           not QLAMD_STATUS_OK: all 12 entries 0, or untouched with QLAMD_ON_FAILURE_KEEP
 *   contact_force  [B][12]  out or NULL: QP solution x = ground reaction forces in
 *                           the base frame (desiredContactForce_ = -x,
 *                           ContactForceDistribution.cpp:502-503), 0 for non-support legs
 *   status         [B]      out, QLAMD_STATUS_*
 *   memory         QLAMD_MEM_DEVICE or QLAMD_MEM_HOST for ALL pointers above
 *   stream         hipStream_t (NULL = default stream)
 */
int qlamd_balance_solve_batch(qlamd_context *ctx, const qlamd_state_batch *in, int64_t batch,
                              double *joint_effort, double *contact_force, int32_t *status,
                              int memory, void *stream);

/* Force distribution only, with the virtual wrench supplied by the caller:
 * replaces bool ContactForceDistributionBase::computeForceDistribution(const Force& F_B, const Torque& T_B)
 * (balance_controller/include/balance_controller/contact_force_distribution/ContactForceDistributionBase.hpp:97-98,
 *  ContactForceDistribution.cpp:99-136) including computeJointTorques (:516-578) and the +-limit clamp.
 *   virtual_wrench [B][6] = (F_B, T_B) in the base frame; other arguments as qlamd_balance_solve_batch. */
int qlamd_force_distribution_batch(qlamd_context *ctx, const double *joint_position, const double *base_orientation,
                                   const uint8_t *support_leg, const double *surface_normal,
                                   const double *virtual_wrench, int64_t batch, double *joint_effort,
                                   double *contact_force, int32_t *status, int memory, void *stream);

/* ---- placement: who shares a wavefront ---------------------------------------------------------
 * The hot kernel runs four robots per wavefront in lockstep; each robot's active-set loop (ContactForceDistribution.cpp:385-514
 * in the reference, QuadProg++.cc:216-445 for the method) takes a data-dependent number of add / drop passes, a wavefront
 * lasts as long as the union of its four robots' passes, and a launch of a few thousand robots as long as its slowest
 * wavefront.  By default slot s of a launch (row s % 4 of wavefront s / 4) takes robot s.  The placed entries take the
 * assignment from the caller, report how much work each robot was, and can leave the placement for the next control step:
 *   robot_order       [B] in, or NULL (identity): robot_order[s] = index of the robot that sits in slot s.  A permutation
 *                     of 0..B-1.  Results are written at the robot's own index and are bit for bit those of the plain entry
 *                     whatever the permutation (a robot's arithmetic never leaves its 16 lanes).  QLAMD_MEM_HOST:
 *                     checked (QLAMD_ERR_INVALID_ARGUMENT unless every robot appears exactly once); QLAMD_MEM_DEVICE: not
 *                     checked -- an entry outside [0, B) leaves its slot empty, a robot that is missing is not solved (its
 *                     outputs stay untouched), one that appears twice is solved twice with the same result.
 *   iterations        [B] out, or NULL: outer iterations of each robot's QP (QuadProg++'s `iter`: one per constraint
 *                     selected; 0 for a robot without a stance leg or with status QLAMD_STATUS_NOT_PD).
 *   prev_iterations   [B] in and
 *   next_robot_order  [B] out, both or neither: the placement (by `policy`) that follows from the counts in prev_iterations
 *                     -- the `iterations` of the caller's previous call -- for the caller's next call.  Up to 1 M robots
 *                     it is made by extra wavefronts inside the solve's own launch, in its shadow (one per 1024 robots,
 *                     per 4096 from 16 384 robots up; several of them meet at a barrier in global memory): no launch, no
 *                     time between two control steps; beyond that by qlamd_placement_from_iterations' launches behind the
 *                     solve.  Must not alias robot_order / iterations (two buffers of each, used in turn).
 * The hint is free for a caller that runs at 400 Hz (balance_controller_manager.cpp:48): the iteration counts of the last
 * control steps predict this one's (same count for 94 % of the robots of the bench batches one period later, rank
 * correlation 0.99: tools/experiments/placement_model.py).  The loop of such a caller, with two buffers of each kind:
 *     tick k:  placement = { order[k & 1], iters[k & 1], iters[(k - 1) & 1], order[(k + 1) & 1], QLAMD_PLACEMENT_AUTO };
 *              qlamd_balance_solve_placed_batch(ctx, &state, B, &placement, effort, NULL, status, QLAMD_MEM_DEVICE, stream);
 * (tick k runs in the placement made during tick k - 1 from the counts of tick k - 2; start with identity orders and zero
 * counts).  Policies:
 *   QLAMD_PLACEMENT_LATENCY     the hardest quarter of the robots one per wavefront, each joined by the three easiest left:
 *                               for batches of about one or two wavefronts per SIMD (4096 - 8192 robots on an MI355X),
 *                               where the launch lasts as long as its slowest wavefront
 *   QLAMD_PLACEMENT_THROUGHPUT  robots sorted by iteration count, hardest first, four neighbours per wavefront: the union
 *                               of four similar sequences is the shortest there is; for batches that fill every SIMD
 *                               several times over.  When the placement is made by a placed balance / force-distribution
 *                               call (next_robot_order: the call knows the support flags; qlamd_placement_from_iterations
 *                               does not) the robots on more than two support legs come first and the others after them,
 *                               each group sorted by count: a wavefront whose four robots all stand on at most two legs
 *                               solves 6-variable QPs (the support legs sit in front of the row: every product of a pass
 *                               and the inversion of G half as long; 9-11 % on batches of such robots), and a sorted
 *                               placement that ignores the class would mix them
 *   QLAMD_PLACEMENT_NONE        the batch order: next_robot_order is the identity (written by the launch itself, no extra
 *                               wavefronts) -- for a caller's loop that keeps its shape when no placement pays
 *   QLAMD_PLACEMENT_AUTO        without a warm start: latency below 16 384 robots, throughput from there.  With one
 *                               (prev_working_set / working_set below): NONE up to 4096 robots, THROUGHPUT above -- a
 *                               warm-started launch of a few thousand robots lasts as long as its slowest ROBOT (its three
 *                               neighbours in the wavefront cost it 0.3 us), so a placement buys nothing while every wavefront
 *                               has a SIMD to itself, and the latency policy, which starts the hard robots together, puts two
 *                               slow wavefronts on one SIMD once there are two a SIMD (the loop on trajectories, us per step,
 *                               latency / throughput / none: 8192 trot robots 25.5 / 23.3 / 23.8, 12 288: 31.5 / 26.5 / 29.1,
 *                               4096 static 14.7 / 14.7 / 14.4: profiles/r6/ab_warm_policies.txt)
 * A placement computed from stale or wrong hints costs time, never correctness.  Not available with
 * qlamd_set_robots_per_wave(16 | 64) (QLAMD_ERR_INVALID_ARGUMENT): one lane is one robot there and nothing is shared.
 * Other arguments as qlamd_balance_solve_batch / qlamd_force_distribution_batch; the arrays of `placement` live in the
 * memory space of the call; placement = NULL is the plain entry. */
#define QLAMD_PLACEMENT_AUTO 0
#define QLAMD_PLACEMENT_LATENCY 1
#define QLAMD_PLACEMENT_THROUGHPUT 2
#define QLAMD_PLACEMENT_NONE 3
typedef struct qlamd_placement {
  const int32_t *robot_order;
  int32_t *iterations;
  const int32_t *prev_iterations;
  int32_t *next_robot_order;
  int policy;                       /* QLAMD_PLACEMENT_*, for next_robot_order */
  /* The other half of the hint, for the balance / force-distribution entries with QLAMD_MEM_DEVICE: WARM START.  Each robot's
   * active-set loop starts from the working set in prev_working_set [B] (bit 5 leg + kind of row: kind 0 the minimal normal
   * force, 1..4 the friction pyramid's +t1, -t1, +t2, -t2; bits 20..23: the support legs the set was reached with, written by
   * the library, 0 = not recorded) instead of the empty one, and its final working set goes to working_set [B] (0 for a robot
   * whose status is not QLAMD_STATUS_OK); either may be NULL; they may be ONE array, updated in place (a robot's set is read
   * and written by its own lanes only).  A robot whose support legs are not the recorded ones any more -- a trot entering or
   * leaving double support -- and a robot without a record (a word of zeros: its first step) has no set to start from: a stale
   * one costs such a robot more than it saves.  With prev_working_set given it BUILDS one instead of adding a row a pass: by
   * rounds, every leg's most violated row installed as an equality together with the other legs', up to three rounds, then on as
   * after any warm start (four rows a round for the price of a pass and a half: the robots a launch of a trot waits for).
   * With working_set alone (prev_working_set NULL) every robot starts from the empty set by the reference's method, bit for bit
   * the plain entry's answer.  Hand a robot the set it ended with on its previous
   * control step (zeros to start with: a cold start).  At 400 Hz that is this
   * step's final set for 96 % of the robots of the bench batches: the rows are installed as equalities, slots whose
   * multiplier comes out negative are dropped, and the method of the reference continues from there -- a set that no longer
   * fits (other stance legs, other loads) costs passes, not the answer: the minimiser is unique, efforts and forces agree
   * with the cold start's to the solver's accuracy (1e-8), statuses are the same.  A mask that is no working set at all (more
   * than three rows of one leg) is ignored; an answer that fails the final check (possible only from a set unrelated to the
   * robot's state) is reported as QLAMD_STATUS_WARM_REJECTED, never returned.  What does change: `iterations` counts the
   * rows installed and dropped by the warm start plus the passes still needed, no longer QuadProg++'s `iter`
   * (solve_quadprog has no warm start, QuadProg++.cc:216-233) -- still what the robot cost, so still the placement hint. */
  const uint32_t *prev_working_set;
  uint32_t *working_set;
} qlamd_placement;
int qlamd_balance_solve_placed_batch(qlamd_context *ctx, const qlamd_state_batch *in, int64_t batch,
                                     const qlamd_placement *placement, double *joint_effort, double *contact_force,
                                     int32_t *status, int memory, void *stream);
int qlamd_force_distribution_placed_batch(qlamd_context *ctx, const double *joint_position, const double *base_orientation,
                                          const uint8_t *support_leg, const double *surface_normal,
                                          const double *virtual_wrench, int64_t batch, const qlamd_placement *placement,
                                          double *joint_effort, double *contact_force, int32_t *status, int memory,
                                          void *stream);
/* The same for the other entries that run a lane-cooperative active-set QP -- qlamd_qp_solve_batch,
 * qlamd_weighted_lsq_qp_batch, qlamd_wholebody_solve_batch -- without a second set of signatures: the placement is handed to
 * the context and taken (and cleared) by the NEXT call of one of these three on it, whichever it is.  QLAMD_MEM_DEVICE
 * calls only (a host-memory call that finds a placement pending returns QLAMD_ERR_INVALID_ARGUMENT and clears it).
 * robot_order / iterations index the call's problems; with prev_iterations / next_robot_order the placement for the
 * caller's next call is made by qlamd_placement_from_iterations' launches behind the solve, on its stream.
 * placement = NULL withdraws a pending one.  (qlamd_balance_solve_batch and the other entries ignore it.)
 * Warm start: qlamd_wholebody_solve_batch also takes prev_working_set / working_set this way, with TWO words per robot
 * ([B][2] uint32 = 64 bits, low word first: bit 11 leg + kind, kinds 0..4 as for the balance step, 5 + 2k / 6 + 2k the upper /
 * lower torque bound of the leg's joint k; bits 44..47: the support legs the set was reached with); the two dense entries start
 * cold and refuse them (QLAMD_ERR_INVALID_ARGUMENT).  QLAMD_PLACEMENT_AUTO means with a warm start what it means for the balance
 * step: no placement up to 4096 robots -- the solving launch writes the identity into next_robot_order itself and nothing is
 * launched behind it (the placement's two launches were 5 of the 23 us of a warm-started step of 4096 robots: 23.1 -> 16.2 us) --
 * and the throughput policy above. */
int qlamd_place_next_call(qlamd_context *ctx, const qlamd_placement *placement);

/* The placement on its own: iterations [B] in (any counts: only their order matters; negative counts count as 0, counts
 * above 23 as 23), robot_order [B] out.  A stable counting sort on the device (ties by robot index: the result is a
 * function of the counts alone); 6 us at 4096 robots, 9 us from 8192 robots up (two launches). */
int qlamd_placement_from_iterations(qlamd_context *ctx, const int32_t *iterations, int64_t batch, int policy,
                                    int32_t *robot_order, int memory, void *stream);

/* ---- swing-leg torque ----------------------------------------------------
 * Replaces MyRobotSolver::update (single_leg_test/lib/model_test_header.cpp:412-503) for the legs
 * that are NOT support legs:
 *   qdd = accel_scale * (qd - qd_oldest) / (accel_window * period)
 *   tau = InverseDynamics(leg, q_id, qd, qdd; gravity (0,0,-gravity) in the base frame)
 *       + J(q)' (kp o (p_des - FK(q)) + kd o (v_des - J(q) qd))
 * with the gains of balance_controller/config/controller_gains.yaml:42-51. */
typedef struct qlamd_swing_params {
  double kp[3], kd[3];      /* 300 / 20 per axis                                               */
  double period;            /* control period, 0.0025 s (balance_controller_manager.cpp:48)    */
  double accel_window;      /* 10  (Time_derta = period * 10, model_test_header.cpp:418)       */
  double accel_scale;       /* 0.5 (model_test_header.cpp:460)                                 */
  double gravity;           /* 9.81 (RBDL leg models)                                          */
} qlamd_swing_params;
void qlamd_swing_default_params(qlamd_swing_params *p);

typedef struct qlamd_swing_batch {
  const double *joint_position;        /* [B][12] current joints (FK, Jacobian)                 */
  const double *joint_velocity;        /* [B][12] newest entry of the velocity queue            */
  const double *joint_velocity_oldest; /* [B][12] oldest entry of the 11-deep velocity queue    */
  const double *target_foot_position;  /* [B][4][3] base frame (State::getTargetFootPositionInBaseForLimb) */
  const double *target_foot_velocity;  /* [B][4][3] base frame                                  */
  const uint8_t *support_leg;          /* [B][4]; torques are produced for legs with 0 here     */
  const double *id_joint_position;     /* [B][12] joints fed to the inverse dynamics, or NULL = joint_position.
                                          The reference feeds every limb the LAST stored limb's joints
                                          (model_test_header.cpp:106-112; ros_balance_controller.cpp:239);
                                          pass that here to reproduce it. */
} qlamd_swing_batch;

/* joint_effort [B][12]: swing legs written, support legs set to 0. */
int qlamd_swing_leg_torque_batch(qlamd_context *ctx, const qlamd_swing_params *params, const qlamd_swing_batch *in,
                                 int64_t batch, double *joint_effort, int memory, void *stream);

/* Virtual wrench only: (F_B, T_B) [B][6].  Replaces
 * VirtualModelController::computeError/GravityCompensation/VirtualForce/VirtualTorque
 * (VirtualModelController.cpp:104-268).  Reads the base and desired-base fields of `in` only
 * (joint_position / support_leg may be NULL). */
int qlamd_virtual_wrench_batch(qlamd_context *ctx, const qlamd_state_batch *in, int64_t batch,
                               double *wrench, int memory, void *stream);

/* Leg kinematics for all four legs: foot position [B][4][3], translation
 * Jacobian [B][4][9] (row-major d p / d q), gravity torque [B][4][3] for the
 * gravity field g_B = q^-1.rotate((0,0,-gravity)).  Any output may be NULL.
 * Replaces QuadrupedKinematics::FowardKinematicsSolve / AnalysticJacobian /
 * getGravityCompensationForLimb (quadrupedkinematics.cpp:143-278,485-552). */
int qlamd_leg_kinematics_batch(qlamd_context *ctx, const double *joint_position,
                               const double *base_orientation, int64_t batch,
                               double *foot_position, double *jacobian, double *gravity_torque,
                               int memory, void *stream);

/* ---- dense QP batch -------------------------------------------------
 * min 1/2 x'Gx + g0'x  s.t.  CE'x + ce0 = 0,  CI'x + ci0 >= 0, one problem per batch entry, all
 * of the same shape (n <= 12, p <= 2, m <= 48), row-major, one constraint per COLUMN of CE / CI as
 * in quadprogpp::solve_quadprog(G, g0, CE, ce0, CI, ci0, x) (qp_solver/include/qp_solver/QuadProg++.h:69-72),
 * which this replaces together with qp_solver::QuadraticProblemSolver::minimize
 * (qp_solver/src/quadraticproblemsolver.cpp:65-97; its wrapper passes CI = -A', ci0 = b for A x <= b,
 * :164).  The Goldfarb-Idnani iteration is the reference's, including its treatment of an all-zero
 * equality column when p = 1 (with p = 2 both columns have to be genuine, linearly independent normals: an
 * all-zero or dependent second column is left out and the problem's status is QLAMD_STATUS_DEPENDENT_EQUALITY).
 * G is not modified.
 *   G [B][n][n], g0 [B][n], CE [B][n][p] (NULL if p = 0), ce0 [B][p], CI [B][n][m], ci0 [B][m]
 *   x [B][n] out, objective [B] out or NULL (+inf when infeasible), status [B] out (QLAMD_STATUS_*)
 */
int qlamd_qp_solve_batch(qlamd_context *ctx, int n, int p, int m, const double *G, const double *g0,
                         const double *CE, const double *ce0, const double *CI, const double *ci0,
                         int64_t batch, double *x, double *objective, int32_t *status, int memory,
                         void *stream);

/* ---- weighted least squares with equality and two-sided inequality rows (the OOQP seam of row a8) -------------
 * min (Ax - b)'S(Ax - b) + x'Wx   s.t.  Cx = c,  d <= Dx <= f,   one problem per batch entry, all of one shape.
 * Replaces, argument for argument,
 *     static bool ooqpei::QuadraticProblemFormulation::solve(A, S, b, W, C, c, D, d, f, x)
 * (third-party ooqp_eigen_interface, un-vendored; call sites balance_controller/src/contact_force_distribution/
 * ContactForceDistribution.cpp:367 and :490, matrices declared ContactForceDistribution.hpp:172-188, problem stated
 * at .cpp:388): the narrowest seam at which the reference keeps its own assembly (prepareOptimization,
 * addMinimalForceConstraints, addFrictionConstraints, addDesiredLegLoadConstraints) and swaps only the solver.
 *   A [B][k][n] row-major (k <= 12, n <= 12), S [B][k] and W [B][n]: the DIAGONALS (the reference holds
 *   Eigen::DiagonalMatrix for both), b [B][k];
 *   C [B][p][n], c [B][p] (p <= 12; NULL when p = 0).  A row that is all-zero or in the span of the rows before it is
 *   skipped when consistent (the reference's first pass hands over 3 nS zero rows with c = 0, .cpp:364-366), and makes
 *   the problem QLAMD_STATUS_INFEASIBLE when not;
 *   D [B][m][n], d [B][m], f [B][m] (m <= 24; NULL when m = 0).  A bound of +-DBL_MAX (what the reference writes for
 *   "none", .cpp:246,329: std::numeric_limits<double>::max()) or +-infinity is no bound and never enters arithmetic.
 *   x [B][n] out; status [B] out: QLAMD_STATUS_OK / _INFEASIBLE / _NOT_PD (A'SA + W not positive definite) / _MAX_ITER.
 * ooqpei returns false when OOQP fails; here that is status != QLAMD_STATUS_OK (host/ooqp_eigen_interface/
 * QuadraticProblemFormulation.hpp maps it back to the bool).  OOQP is an interior-point method that stops at its
 * complementarity tolerance; this entry returns the exact minimiser of the same strictly convex problem (active-set
 * method of qlamd_qp_solve_batch, equality rows projected out one by one). */
int qlamd_weighted_lsq_qp_batch(qlamd_context *ctx, int n, int k, int p, int m, const double *A, const double *S,
                                const double *b, const double *W, const double *C, const double *c, const double *D,
                                const double *d, const double *f, int64_t batch, double *x, int32_t *status, int memory,
                                void *stream);

/* ---- pose optimisation batch (BASELINE config 5) ----------------------
 * Replaces free_gait::PoseOptimizationSQP::optimize(Pose&) with its setters
 * (free_gait_core/include/free_gait_core/pose_optimization/PoseOptimizationSQP.hpp:36-54,
 *  free_gait_core/src/pose_optimization/PoseOptimizationSQP.cpp:58-111,
 *  free_gait_core/src/pose_optimization/PoseOptimizationBase.cpp:21-50). */
typedef struct qlamd_pose_params {
  double hip_in_base[4][3];   /* adapter.getPositionBaseToHipInBaseFrame, limb order LF RF RH LH    */
  double com_weight;          /* 2.0, PoseOptimizationObjectiveFunction.cpp:17                      */
  double tolerance;           /* 0.05, PoseOptimizationSQP.cpp:99 (stop when |dp| < tolerance)      */
  int max_iterations;         /* 30                                                                  */
  int dummy_equality;         /* 1 = pass the all-zero equality column like the reference (Q1)      */
  int leg_order[4];           /* iteration order of the reference's unordered_map Stance (Q6);
                                 limbs missing from a problem's stance_mask are skipped             */
} qlamd_pose_params;

/* Reference defaults: hips (+-0.42, +-0.075, 0) (free_gait_core/test/AdapterDummy.cpp:111-125),
 * leg_order = libstdc++ order of a Stance filled LF, RF, LH, RH = {RH, LH, RF, LF}. */
void qlamd_pose_default_params(qlamd_pose_params *p);

typedef struct qlamd_pose_batch {
  const double *stance;          /* [B][4][3] foot positions, world           (setStance)               */
  const uint8_t *stance_mask;    /* [B][4] or NULL (= all four limbs)                                   */
  const double *nominal_stance;  /* [B][4][3] base frame                      (setNominalStance)        */
  const double *support_polygon; /* [B][4][2] counter-clockwise vertices      (setSupportRegion)        */
  const int32_t *n_vertices;     /* [B] or NULL (= 4); 3 or 4                                           */
  const double *center_of_mass;  /* [B][3] in base or NULL (= 0, every shipped adapter)                 */
  const double *max_limb_length; /* [B][4]                                    (setLimbLengthConstraints) */
  const double *pose;            /* [B][7] initial pose (x,y,z,qw,qx,qy,qz)   (optimize's in/out Pose)   */
} qlamd_pose_batch;

/* pose_out [B][7]; iterations [B] (SQP iterations taken) or NULL; status [B] QLAMD_STATUS_*
 * (a failed inner QP stops that problem, like the uncaught exception would in the reference). */
int qlamd_pose_sqp_batch(qlamd_context *ctx, const qlamd_pose_params *params, const qlamd_pose_batch *in,
                         int64_t batch, double *pose_out, int32_t *iterations, int32_t *status, int memory,
                         void *stream);

/* PoseOptimizationQP::optimize (free_gait_core/src/pose_optimization/PoseOptimizationQP.cpp:42-140): the step before
 * the SQP in BaseAuto::optimizePose (BaseAuto.cpp:394-400).  Position only (orientation of `pose` kept):
 * min sum |x + R d_i - f_i|^2  s.t. the support-region half-spaces on (x + R r_com)_xy, solved with the same dense
 * QP (n = 3) and the same all-zero equality column as the reference. */
int qlamd_pose_qp_batch(qlamd_context *ctx, const qlamd_pose_params *params, const qlamd_pose_batch *in, int64_t batch,
                        double *pose_out, int32_t *status, int memory, void *stream);

/* PoseConstraintsChecker::check (free_gait_core/src/pose_optimization/PoseConstraintsChecker.cpp:29-64) for the
 * poses in in->pose: ok[b] = 1 when the centre of mass lies inside the support region and every limb length is
 * within [min - tol, max + tol].  min_limb_length [B][4] or NULL (= 0). */
int qlamd_pose_check_batch(qlamd_context *ctx, const qlamd_pose_params *params, const qlamd_pose_batch *in,
                           const double *min_limb_length, double leg_length_tolerance, int64_t batch, uint8_t *ok,
                           int memory, void *stream);

/* PoseOptimizationGeometric::optimize (free_gait_core/src/pose_optimization/PoseOptimizationGeometric.cpp:34-105):
 * position = centroid of the support region + mean height offset; orientation = the closed-form least-squares
 * rotation between nominal and actual stance (4x4 eigen-problem), re-headed along the fore/hind mid-point line
 * with 70 % of its roll/pitch.  stance_for_orientation [B][12] by limb id (setStanceForOrientation) or NULL
 * (= in->stance; all four limbs are read).  in->pose is not read. */
int qlamd_pose_geometric_batch(qlamd_context *ctx, const qlamd_pose_params *params, const qlamd_pose_batch *in,
                               const double *stance_for_orientation, int64_t batch, double *pose_out, int memory,
                               void *stream);

/* BaseAuto::optimizePose (free_gait_core/src/base_motion/BaseAuto.cpp:394-400) in one launch: geometric -> QP ->
 * constraints check -> SQP for the problems the check rejects.  stage [B] (nullable): 2 = QP result accepted,
 * 3 = SQP ran; iterations [B] (nullable): SQP iterations (0 when it did not run); status [B] as the failing
 * solver reported it (the reference returns false).  in->pose is not read. */
int qlamd_base_auto_optimize_pose_batch(qlamd_context *ctx, const qlamd_pose_params *params, const qlamd_pose_batch *in,
                                        const double *stance_for_orientation, const double *min_limb_length,
                                        double leg_length_tolerance, int64_t batch, double *pose_out, int32_t *stage,
                                        int32_t *iterations, int32_t *status, int memory, void *stream);

/* ---- leg state machine of the controller plugin -------------------------------
 * One call = one control tick for every robot: the command side of baseCommandCallback
 * (ros_balance_controller.cpp:966-1078), footContactsCallback (:1084-1135) and the switch at the top of update()
 * (:234-380).  State ids as StateSwitcher::States (state_switcher/StateSwitcher.hpp:62-72).  All arrays [B][k] by
 * limb id (LF, RF, RH, LH); the 1-byte-per-leg arrays must be 4-byte aligned per robot (they are, at [B][4]). */
#define QLAMD_LEG_INIT 0
#define QLAMD_LEG_STANCE_NORMAL 1
#define QLAMD_LEG_STANCE_SLIPPING 2
#define QLAMD_LEG_STANCE_LOST_CONTACT 3
#define QLAMD_LEG_SWING_NORMAL 4
#define QLAMD_LEG_SWING_LATE_LIFT_OFF 5
#define QLAMD_LEG_SWING_EARLY_TOUCH_DOWN 6
#define QLAMD_LEG_SWING_BUMPED_INTO_OBSTACLE 7
#define QLAMD_LEG_SWING_LATELY_TOUCH_DOWN 8

typedef struct qlamd_leg_state_batch {
  /* in: desired_robot_state.{lf,rf,rh,lh}_leg_mode and the contact sensors */
  const uint8_t *support_leg;    /* [B][4]  leg_mode.support_leg                                   */
  const double *phase;           /* [B][4]  leg_mode.phase (stance phase if support_leg, else swing) */
  const uint8_t *is_footstep;    /* [B][4]  leg_mode.name == "footstep" (:951-964)                 */
  const uint8_t *contact;        /* [B][4]  foot_contacts[i].is_contact                            */
  const double *joint_position;  /* [B][12] measured joint positions                               */
  /* in/out: what the plugin keeps between ticks */
  int8_t *limb_state;            /* [B][4]  limbs_state                                            */
  uint8_t *store_flag;           /* [B][4]  store_current_joint_state_flag_                        */
  double *stored_joint_position; /* [B][12] stored_limb_joint_position_                            */
  double *joint_command;         /* [B][12] commands (held at the stored position after a lost contact) */
  double *foot_target;           /* [B][12] foot_positions (nudged back/up when bumped, down when late)  */
  uint8_t *support;              /* [B][4]  robot_state->isSupportLeg: feeds qlamd_state_batch.stance   */
  /* out */
  int8_t *leg_state_code;        /* [B][4]  the leg_state message (:222-224): 0, 2, 1, 3 or -1     */
} qlamd_leg_state_batch;

/* index_quirk != 0 reproduces footContactsCallback's `continue` that skips `i++` (:1100,1122): the next contact is
 * then applied to the same limb, as in the reference.  0 gives every limb its own contact. */
int qlamd_leg_state_machine_batch(qlamd_context *ctx, const qlamd_leg_state_batch *io, int index_quirk, int64_t batch,
                                  int memory, void *stream);

/* ---- free_gait_msgs/RobotState wire format -> structure-of-arrays -------------
 * What RosBalanceController::baseCommandCallback reads from a /desired_robot_state message
 * (ros_balance_controller.cpp:761-1083), for a batch of serialised messages (ROS 1 wire format, e.g. the record
 * payloads of a bag): messages = the concatenated bytes, message b = [offsets[b], offsets[b+1]).
 * The outputs line up with the des_* fields of qlamd_state_batch, with support_leg / phase / joint_command /
 * foot_target (= foot_position) / is_footstep (= leg_mode == 4) of qlamd_leg_state_batch and with the foot targets
 * of qlamd_swing_batch.  Any output pointer may be NULL. */
#define QLAMD_LEG_MODE_OTHER 0
#define QLAMD_LEG_MODE_JOINT 1
#define QLAMD_LEG_MODE_LEG_MODE 2
#define QLAMD_LEG_MODE_CARTESIAN 3
#define QLAMD_LEG_MODE_FOOTSTEP 4
#define QLAMD_WIRE_OK 0
#define QLAMD_WIRE_TRUNCATED 1      /* the message ends before its own length fields say it should            */
#define QLAMD_WIRE_MISSING_FIELD 2  /* joints.position has < 3 entries or a target_{position,velocity,acceleration}
                                       array is empty: the reference indexes them unchecked (:802-861)          */

typedef struct qlamd_robot_state_fields {
  double *des_pos;           /* [B][3]  base_pose.pose.pose.position                                 */
  double *des_quat;          /* [B][4]  base_pose.pose.pose.orientation as (w, x, y, z)              */
  double *des_linvel;        /* [B][3]  base_pose.twist.twist.linear                                 */
  double *des_angvel;        /* [B][3]  base_pose.twist.twist.angular                                */
  double *joint_command;     /* [B][12] {lf,rf,rh,lh}_leg_joints.position[0..2]                      */
  double *foot_position;     /* [B][12] *_target.target_position[0].point                            */
  double *foot_velocity;     /* [B][12] *_target.target_velocity[0].vector                           */
  double *foot_acceleration; /* [B][12] *_target.target_acceleration[0].vector                       */
  double *surface_normal;    /* [B][12] *_leg_mode.surface_normal.vector                             */
  double *phase;             /* [B][4]  *_leg_mode.phase                                             */
  uint8_t *support_leg;      /* [B][4]  *_leg_mode.support_leg                                       */
  uint8_t *leg_mode;         /* [B][4]  *_leg_mode.name as QLAMD_LEG_MODE_*                          */
} qlamd_robot_state_fields;

int qlamd_robot_state_unpack_batch(qlamd_context *ctx, const uint8_t *messages, const int64_t *offsets, int64_t batch,
                                   const qlamd_robot_state_fields *out, int32_t *status, int memory, void *stream);

/* ---- analytic leg inverse kinematics ------------------------------------------
 * QuadrupedKinematics::InverseKinematicsSolve (quadruped_model/src/quadrupedkinematics.cpp:377-483) for the four
 * feet of every robot, as QuadrupedState::getLimbJointPositionsFromPositionBaseToFootInBaseFrame calls it
 * (quadruped_state.cpp:283-300): on failure (a NaN angle) the leg keeps joint_position_last and ok = 0. */
#define QLAMD_IK_OUT_LEFT 0   /* the row of `results` the reference returns for each LimbType string (:466-473) */
#define QLAMD_IK_IN_RIGHT 1
#define QLAMD_IK_IN_LEFT 2
#define QLAMD_IK_OUT_RIGHT 3

typedef struct qlamd_ik_params {
  double d, l1, l2;         /* hip offset and link lengths; the reference hard-codes 0.1, 0.25, 0.25 (:383-385), which
                               is not the geometry of its own URDF (0.23 / 0.22053, 0.308, 0.308)               */
  uint8_t limb_config[4];   /* QLAMD_IK_* per limb; default "><" (quadruped_state.cpp:61,385-390)               */
} qlamd_ik_params;

void qlamd_ik_default_params(qlamd_ik_params *p);

/* foot_position [B][12] in the base frame, joint_position_last [B][12] or NULL, joint_position [B][12] out,
 * ok [B][4] out or NULL. */
int qlamd_leg_inverse_kinematics_batch(qlamd_context *ctx, const qlamd_ik_params *params, const double *foot_position,
                                       const double *joint_position_last, int64_t batch, double *joint_position,
                                       uint8_t *ok, int memory, void *stream);

/* ---- swing branch of RosBalanceController::update -----------------------------
 * ros_balance_controller.cpp:467-603: for every NON-support leg the joint effort command is
 *   the swing-leg torque (qlamd_swing_leg_torque_batch)       when the leg mode is "cartesian" or "footstep",
 *   joint PID on the position command + gravity compensation   when it is "joint" (or was never set),
 *   gravity compensation alone                                 when it is "leg_mode";
 * computeTorqueFromPositionCommand (:720-756): command clamped to the joint limits, control_toolbox PID on
 * (command - position) with the gains of balance_controller/config/control.yaml:18-29.  Support legs are left
 * untouched in joint_effort, so that running qlamd_balance_solve_batch first and this entry second on the same
 * array yields all 12 commands of a tick. */
typedef struct qlamd_joint_pid_params {
  double p[12], i[12], d[12], i_max[12], i_min[12];
  double lower[12], upper[12];   /* joint limits (enforceJointLimits, :1152-1166) */
  int antiwindup;
} qlamd_joint_pid_params;

typedef struct qlamd_swing_branch_extra {
  const double *base_orientation;   /* [B][4]  measured base orientation (w,x,y,z): gravity_in_base (:471)      */
  const double *joint_command;      /* [B][12] position commands (RobotState.*_leg_joints.position)             */
  const uint8_t *leg_mode;          /* [B][4]  QLAMD_LEG_MODE_* in force, or NULL (= never set)                */
  double *pid_error_last;           /* [B][12] in/out: Pid::p_error_last_                                      */
  double *pid_error_integral;       /* [B][12] in/out: Pid::i_error_                                           */
} qlamd_swing_branch_extra;

void qlamd_joint_pid_default_params(qlamd_joint_pid_params *p);

/* joint_effort [B][12] in/out.  period: the control period handed to update(). */
int qlamd_swing_branch_batch(qlamd_context *ctx, const qlamd_swing_params *params, const qlamd_joint_pid_params *pid,
                             const qlamd_swing_batch *in, const qlamd_swing_branch_extra *extra, double period,
                             int64_t batch, double *joint_effort, int memory, void *stream);

/* ---- whole-body (floating-base) dynamics and QP ---------------------------------
 * What BASELINE.json's north_star describes and the reference does NOT contain (no mass
 * matrix / NonlinearEffects / CalcPointJacobian call anywhere; RBDL only drives one swing leg,
 * single_leg_test/lib/model_test_header.cpp:229-244,460).  There is therefore no reference interface to cite; the
 * entries follow the conventions of the rest of this header.  Model: the whole urdf/quadruped_model.urdf tree --
 * base_link + 4 x (3 revolute links + fixed foot link), 18 degrees of freedom.
 *
 * Generalised velocity nu = [ v (3) ; w (3) ; qd (12) ]: velocity of the base origin and angular velocity of the
 * base, both in BASE coordinates, then the joint rates (LF, RF, RH, LH).  nu' = d nu / dt componentwise.
 *     M(q) nu' + h(q, nu) = [0 ; tau] + Jc' f
 * f [12]: forces the ground applies on the four feet (base coordinates); Jc [12][18]: rows 3l..3l+2 =
 * [ 1 , -[r_l]x , 0 .. J_leg(l) .. 0 ].  Gravity g_W = (0, 0, -gravity). */
typedef struct qlamd_wholebody_batch {
  const double *joint_position;        /* [B][12] */
  const double *joint_velocity;        /* [B][12] */
  const double *base_orientation;      /* [B][4]  (w,x,y,z) base -> world */
  const double *base_linear_velocity;  /* [B][3]  world, like the control-step state batch */
  const double *base_angular_velocity; /* [B][3]  base */
  const double *desired_base_acceleration;  /* [B][6]  [v' ; w'] in base coordinates (solve entry only) */
  const double *desired_joint_acceleration; /* [B][12] or NULL = 0                  (solve entry only) */
  const uint8_t *support_leg;          /* [B][4]                                    (solve entry only) */
  const double *surface_normal;        /* [B][4][3] world or NULL, like the control-step state batch (solve entry only) */
} qlamd_wholebody_batch;

typedef struct qlamd_wholebody_params {
  double torque_weight;   /* w_tau */
  double torque_limit;    /* tau_max, a constraint of the QP here (the reference clamps afterwards, :451-454) */
  double gravity;
} qlamd_wholebody_params;

void qlamd_wholebody_default_params(qlamd_wholebody_params *p);

/* Composite-rigid-body mass matrix, bias forces (Coriolis, centrifugal, gravity) and the contact Jacobian of the
 * four feet.  mass_matrix [B][18][18], nonlinear_effects [B][18], contact_jacobian [B][12][18]: each may be NULL. */
int qlamd_wholebody_dynamics_batch(qlamd_context *ctx, const qlamd_wholebody_batch *in, double gravity, int64_t batch,
                                   double *mass_matrix, double *nonlinear_effects, double *contact_jacobian,
                                   int memory, void *stream);

/* One whole-body control step per robot.  With [b ; tau0] = M nu'_des + h (inverse dynamics for the desired
 * accelerations), A = the reference's force-to-wrench map (ContactForceDistribution.cpp:168-206) and S, w_reg, mu,
 * f_min the context's force-distribution parameters:
 *     min (A f - b)' S (A f - b) + w_reg |f|^2 + w_tau |tau|^2
 *     s.t. tau = tau0 - J_leg' f (stance legs),  friction pyramid and minimal normal force on f
 *          (ContactForceDistribution.cpp:210-336),  |tau| <= tau_max
 * i.e. 6 nS variables, 3 nS equalities and 11 nS inequalities; the torques are eliminated on the device.
 * joint_effort [B][12] (swing legs: tau0), contact_force [B][12] or NULL, status [B] (QLAMD_STATUS_*; on failure the
 * robot's efforts and forces are 0, or untouched with QLAMD_ON_FAILURE_KEEP). */
int qlamd_wholebody_solve_batch(qlamd_context *ctx, const qlamd_wholebody_params *params,
                                const qlamd_wholebody_batch *in, int64_t batch, double *joint_effort,
                                double *contact_force, int32_t *status, int memory, void *stream);

/* ---- the whole control tick in one call (SURVEY.md section 8 row a1 with rows f1 and f2) -----------------------
 * What the plugin does between one /desired_robot_state message and 12 effort commands:
 *   baseCommandCallback (ros_balance_controller.cpp:761-1083)   message -> desired base state, leg modes, targets
 *   footContactsCallback + the switch of update() (:1084-1135, :234-380)   leg state machine -> support legs
 *   update() :384-454   balance solve for the support legs (virtual-model wrench -> force QP -> torques, clamped)
 *   update() :467-603   swing branch for the others (joint PID / gravity compensation / swing-leg inverse dynamics)
 * i.e. qlamd_robot_state_unpack_batch -> qlamd_leg_state_machine_batch -> qlamd_balance_solve_batch ->
 * qlamd_swing_branch_batch with the intermediate arrays kept in device memory by the context.  A leg mode name the
 * plugin does not know leaves that leg's mode as it was (:876-964).
 * A message that cannot be parsed (message_status != QLAMD_WIRE_OK) never reaches baseCommandCallback in the
 * reference: update() keeps running on the last command stored.  Here the command in force lives in `command`
 * (caller-owned, opaque, qlamd_tick_command_bytes(batch) bytes, zero-filled before the first tick, in/out): a
 * well-formed message replaces the robot's stored command, a malformed one leaves it and the tick runs on the stored
 * one.  A robot that has no stored command yet (or every robot with a malformed message when `command` is NULL) is
 * skipped: status = QLAMD_STATUS_NO_COMMAND, its joint_effort and all its persistent arrays stay untouched.
 * All arrays [B][k]; persistent arrays are caller-owned so that a controller can be checkpointed. */
typedef struct qlamd_tick_batch {
  /* in: this tick */
  const uint8_t *messages;              /* serialised free_gait_msgs/RobotState, concatenated */
  const int64_t *offsets;               /* [B+1] */
  const double *joint_position;         /* [B][12] measured */
  const double *joint_velocity;         /* [B][12] measured */
  const double *joint_velocity_oldest;  /* [B][12] ten ticks ago (MyRobotSolver's queue, model_test_header.cpp:417-431) */
  const double *base_position;          /* [B][3]  measured */
  const double *base_orientation;       /* [B][4]  */
  const double *base_linear_velocity;   /* [B][3]  */
  const double *base_angular_velocity;  /* [B][3]  */
  const uint8_t *contact;               /* [B][4]  foot contact sensors */
  /* in/out: what the plugin keeps between ticks */
  int8_t *limb_state;                   /* [B][4]  */
  uint8_t *store_flag;                  /* [B][4]  */
  double *stored_joint_position;        /* [B][12] */
  uint8_t *leg_mode;                    /* [B][4]  QLAMD_LEG_MODE_* in force */
  uint8_t *support;                     /* [B][4]  State::isSupportLeg (two leg states have no case in update() and leave it) */
  double *pid_error_last;               /* [B][12] */
  double *pid_error_integral;           /* [B][12] */
  /* out */
  double *joint_effort;                 /* [B][12] */
  int8_t *leg_state_code;               /* [B][4]  or NULL */
  int32_t *status;                      /* [B]     QLAMD_STATUS_* of the balance solve */
  int32_t *message_status;              /* [B]     QLAMD_WIRE_* */
  /* in/out, optional */
  void *command;                        /* qlamd_tick_command_bytes(B) bytes or NULL: the command in force per robot */
  uint32_t *working_set;                /* [B] or NULL: warm start of the balance solve (qlamd_placement::working_set): each
                                           robot's final working set of its previous tick in, this tick's out; zeros to start
                                           with.  One more piece of state the controller keeps between ticks -- the reference
                                           keeps none because OOQP starts every solve from scratch; efforts agree with the
                                           cold start's to 1e-7 */
  int32_t *placement_state;             /* [4][B] or NULL, QLAMD_MEM_DEVICE and more than 16 384 robots only (ignored otherwise): the
                                           state of the placed loop of qlamd_balance_solve_placed_batch for the tick's balance
                                           solve -- two placements and two arrays of iteration counts, used in turn; need not be
                                           initialised.  From 16 384 robots up every SIMD holds several wavefronts and sorting the
                                           robots by class and count is worth 30-40 % of the solve; below, on
                                           states that move, a placement buys a tick nothing (profiles/r6/ab_retry_forms.txt).
                                           The context remembers which tick of the loop comes next for the array it saw last: a
                                           context that alternates between batches starts the loop over each time (slower, never
                                           wrong).  Efforts are those of the unplaced tick bit for bit (cold) */
} qlamd_tick_batch;

size_t qlamd_tick_command_bytes(int64_t batch);

int qlamd_full_tick_batch(qlamd_context *ctx, const qlamd_swing_params *swing, const qlamd_joint_pid_params *pid,
                          const qlamd_tick_batch *io, double period, int index_quirk, int64_t batch, int memory,
                          void *stream);

const char *qlamd_strerror(int code);
int qlamd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QLAMD_H */
