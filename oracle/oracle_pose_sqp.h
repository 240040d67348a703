/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_pose_sqp.c for the reference citations). */
#ifndef ORACLE_POSE_SQP_H
#define ORACLE_POSE_SQP_H

#ifdef __cplusplus
extern "C" {
#endif

/* One pose-optimisation problem.  Arrays are indexed by limb id (LF, RF, RH, LH = 0..3);
 * leg_order[0..n_legs) lists the limbs of the stance in the order the reference iterates its
 * `Stance` (an unordered_map): this fixes summation and constraint-row order. */
typedef struct {
  int n_legs;
  int leg_order[4];
  double stance[4][3];    /* foot positions, world                      (setStance)            */
  double nominal[4][3];   /* nominal stance, base frame                 (setNominalStance)     */
  double hips[4][3];      /* base -> hip, base frame                    (setPositionsBaseToHip) */
  double max_len[4];      /* maximal limb length                        (setLimbLengthConstraints) */
  int n_vertices;
  double polygon[4][2];   /* support region vertices, counter-clockwise (setSupportRegion)     */
  double r_com[3];        /* centre of mass in base                     (setCenterOfMass)      */
  double com_weight;      /* 2.0, PoseOptimizationObjectiveFunction.cpp:17 */
} oracle_pose_problem;

void oracle_polygon_centroid(int nv, const double *v, double c[2]);
int oracle_polygon_halfspaces(int nv, const double *v, double *A, double *b);
void oracle_quat_box_plus(const double q[4], const double d[3], double out[4]);
double oracle_pose_cost(const oracle_pose_problem *pb, const double pose[7]);
void oracle_pose_grad_hess(const oracle_pose_problem *pb, const double pose[7], double g[6], double H[36]);
int oracle_pose_constraints(const oracle_pose_problem *pb, const double pose[7], double *val, double *vmax, double *A);

/* pose = (x, y, z, qw, qx, qy, qz).  tol / max_iter as PoseOptimizationSQP.cpp:99 (0.05, 30).
 * dummy_equality = 1 reproduces the all-zero equality column.  dp_hist (may be NULL)
 * receives the step of every iteration ([max_iter][6]).  Returns the last QP status. */
int oracle_pose_sqp(const oracle_pose_problem *pb, const double pose_in[7], double tol, int max_iter,
                    int dummy_equality, double pose_out[7], int *iters_out, double *cost_out, double *dp_hist);

/* A batch of problems (array of structs), OpenMP over problems: the CPU baseline of bench.py --workload pose_sqp. */
void oracle_pose_sqp_batch(const oracle_pose_problem *pbs, const double *pose_in, long long B, double tol, int max_iter,
                           int dummy_equality, double *pose_out, int *iters, int *status, int nthreads);

/* PoseOptimizationQP::optimize (free_gait_core/src/pose_optimization/PoseOptimizationQP.cpp:42-140): position only,
 * orientation kept; min sum |x + R d_i - f_i|^2  s.t.  G (x + R r_com)_xy <= h, plus the dummy equality. */
int oracle_pose_qp(const oracle_pose_problem *pb, const double pose_in[7], int dummy_equality, double pose_out[7]);

/* PoseConstraintsChecker::check (PoseConstraintsChecker.cpp:29-64): centre of mass inside the support region
 * (grid_map::Polygon::isInside; the inward-offset copy is built but not used there) and every limb length within
 * [min - tol, max + tol].  Returns 1 when the pose passes. */
int oracle_pose_check(const oracle_pose_problem *pb, const double pose[7], const double min_len[4], double leg_tol);
int oracle_polygon_is_inside(int nv, const double *v, const double pt[2]);

/* PoseOptimizationGeometric::optimize and BaseAuto::optimizePose (oracle_pose_geometric.c) */
void oracle_sym4_eigen(const double C[16], double w[4], double V[16]);
int oracle_pose_geometric(const oracle_pose_problem *pb, const double stance_for_orientation[4][3], double pose_out[7],
                          double q_procrustes[4]);
int oracle_base_auto_optimize_pose(const oracle_pose_problem *pb, const double stance_for_orientation[4][3],
                                   const double min_len[4], double leg_tol, double sqp_tol, int sqp_max_iter,
                                   int dummy_equality, double pose_out[7], int *stage_out);

#ifdef __cplusplus
}
#endif
#endif
