/* gear_host.h — C entry points of the C++ host layer (gear_amd/host/gear_host.hpp), the caller
 * side of the hot path: gear's JointPathPlanner with its feasibility closure evaluated in batches
 * through gear_hip.h.  The reference is a Rust crate and exposes no FFI; each entry point names the
 * reference item it stands for (paths relative to /root/reference).  These exist so that the host
 * layer can be driven from C and from the Python test-suite; a Rust caller would bind gear_hip.h
 * directly.
 *
 * Conventions: 0 (GP_OK) or a positive count on success, a negative GP_ERR_* otherwise with the
 * message in gp_last_error().  Joint vectors cross this boundary as double and are converted to the
 * planner's scalar type (float | double = the reference's type parameter N).  Returned strings are
 * owned by the planner handle and stay valid until its next call. */
#ifndef GEAR_HOST_H
#define GEAR_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "gear_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
  GP_OK = 0,
  GP_ERR_OTHER = -1,
  GP_ERR_COLLISION_START = -10,      /* Error::Collision { part: Start }      src/errors.rs:33-37 */
  GP_ERR_COLLISION_END = -11,        /* Error::Collision { part: End }                            */
  GP_ERR_SELF_COLLISION_START = -12, /* Error::SelfCollision { part: Start }  src/errors.rs:38-42 */
  GP_ERR_SELF_COLLISION_END = -13,
  GP_ERR_PATH_PLAN_FAIL = -14,       /* Error::PathPlanFail                   src/errors.rs:31-32 */
  GP_ERR_DOF_MISMATCH = -15,         /* Error::DofMismatch                    src/errors.rs:43-44 */
  GP_ERR_NOT_FOUND = -16,            /* Error::NotFound                                           */
  GP_ERR_PARSE = -17,                /* Error::ParseError                                         */
  GP_ERR_IO = -18,
  GP_ERR_GPU = -19                   /* a gear_hip.h call failed                                  */
};

typedef struct gp_planner gp_planner;   /* JointPathPlanner<N>                                    */
typedef struct gp_compound gp_compound; /* ncollide3d Compound<f64> as gear builds it             */

const char* gp_last_error(void);

/* JointPathPlannerBuilder::from_urdf_file(..).collision_check_margin(..).step_length(..)
 * .max_try(..).num_smoothing(..).self_collision_pairs(parse_colon_separated_pairs(..)).finalize()
 * (src/planner/joint_path_planner.rs:273-348, src/collision/collision_checker.rs:314-329).
 * precision 32|64 = N; seed: the seedable stand-in for rand::thread_rng. */
int gp_planner_create(const char* urdf_path, double margin, int precision, int device, double step_length, int max_try,
                      int num_smoothing, const char* const* self_pairs, int n_self_pairs, uint64_t seed, gp_planner** out);
void gp_planner_destroy(gp_planner* p);

/* No counterpart in the reference: speculate != 0 hands whole RRT connect lines and
 * `smoothing_lookahead` smoothing trials to the GPU per call; 0 evaluates sample by sample like the
 * rrt crate.  Both return the same path for the same seed.  Also re-seeds the random source. */
int gp_planner_set_speculation(gp_planner* p, int speculate, int smoothing_lookahead, uint64_t seed);

/* Compound::from_urdf_file / from_urdf_robot (src/collision/collision_checker.rs:349-369) */
int gp_compound_from_urdf_file(const char* path, gp_compound** out);
/* Compound::new(shapes) from posed primitives (box / ball / cylinder descriptors of gear_hip.h) */
int gp_compound_from_shapes(const gh_shape_desc* shapes, int n, gp_compound** out);
void gp_compound_destroy(gp_compound* c);
int gp_compound_size(const gp_compound* c);

/* using_joints: node indices of the movable joints in iter_joints() order.  end_joint NULL or "":
 * the whole collision_check_robot; else k::SerialChain::from_end(find(end_joint))
 * (src/planner/ik_planner.rs:113-118).  Returns the count. */
int gp_using_joints(gp_planner* p, const char* end_joint, int32_t* out, int cap);

/* JSON description of the chain (k::Chain order, limits, mimic) and of the CollisionChecker's map
 * (collision_checker.rs:234-262) — what gh_scene_create is fed.  For host-logic tests. */
const char* gp_describe(gp_planner* p);

/* The six predicates, src/planner/joint_path_planner.rs:71-134.  1 = true, 0 = false. */
int gp_is_feasible(gp_planner* p, const int32_t* using_joints, int n_using, const double* q, const gp_compound* objects);
int gp_is_feasible_with_self(gp_planner* p, const int32_t* using_joints, int n_using, const double* q);
int gp_has_any_colliding(gp_planner* p, const gp_compound* objects);
int gp_has_any_colliding_with_self(gp_planner* p);
/* names: '\n'-separated, obstacle-major with duplicates as in the reference; returns the count */
int gp_colliding_link_names(gp_planner* p, const gp_compound* objects, const char** names);
/* names: "joint1:joint2" per line */
int gp_colliding_link_names_with_self(gp_planner* p, const char** names);

/* is_feasible / is_feasible_with_self for n configurations at once (q: [n][n_using]); the chain's
 * state is not touched.  ok[i] = 1 when feasible. */
int gp_is_feasible_batch(gp_planner* p, const int32_t* using_joints, int n_using, const double* q, size_t n,
                         const gp_compound* objects, int with_self, uint8_t* ok);

/* JointPathPlanner::plan (joint_path_planner.rs:144-190) and plan_avoid_self_collision (:198-243):
 * start/goal validation, rrt::dual_rrt_connect, rrt::smooth_path.  path: [cap_rows][n_using], *rows
 * receives the number of way points.  stats (may be NULL): [0] GPU calls, [1] samples sent,
 * [2] samples the sequential algorithm consumes, [3] RRT iterations, [4] shortcuts applied,
 * [5] nanoseconds spent inside the GPU evaluator, [6] drawing RRT iterations ahead (random samples, nearest-vertex
 * lookups, stepping), [7] consuming them (8 values).
 * On GP_ERR_(SELF_)COLLISION_* gp_last_collision_names() holds the error's collision_link_names. */
int gp_plan(gp_planner* p, const int32_t* using_joints, int n_using, const double* start, const double* goal,
            const gp_compound* objects, double* path, int cap_rows, int* rows, uint64_t* stats);
int gp_plan_avoid_self_collision(gp_planner* p, const int32_t* using_joints, int n_using, const double* start,
                                 const double* goal, double* path, int cap_rows, int* rows, uint64_t* stats);
const char* gp_last_collision_names(gp_planner* p);

/* The seeds RandomInitializeIKSolver draws (src/ik.rs:77-78: generate_random_joint_positions_from_limits
 * + modify_to_nearest_angle), k of them, with is_feasible evaluated for all in one batch.
 * seeds: [k][n_using]; ok[i] = 1 when seed i is feasible.  A batch use the reference does not have. */
int gp_feasible_ik_seeds(gp_planner* p, const int32_t* using_joints, int n_using, const double* initial, size_t k,
                         const gp_compound* objects, double* seeds, uint8_t* ok);

/* using_joints.joint_positions() and joint_names() (joint_path_planner.rs:154,250-255) */
int gp_joint_positions(gp_planner* p, const int32_t* using_joints, int n_using, double* q);
const char* gp_joint_names(gp_planner* p);

/* rrt 0.4.0's stepping as the host layer restates it (gear_host.hpp connect_line / walk_line): the samples one
 * Tree::connect line (mode GH_EDGE_CONNECT) or one smooth_path walk (GH_EDGE_WALK) evaluates between `from` and
 * `to`, in precision 32|64, assuming all are free.  out: [cap_rows][dim]; returns the number of samples (which may
 * exceed cap_rows).  Test support for gh_check_edges_from_endpoints. */
int gp_edge_line(const double* from, const double* to, int dim, double step, int precision, int mode, double* out, int cap_rows);

/* Bulk validation of straight edges given by their end points (from, to: [n_edges][n_using]) with the planner's step
 * length: the first infeasible sample of every edge (or -1) and its sample count, the samples generated on the device
 * (gh_check_edges_from_endpoints).  The batched form of the closure calls at joint_path_planner.rs:168-188. */
int gp_first_infeasible_from_endpoints(gp_planner* p, const int32_t* using_joints, int n_using, const double* from,
                                       const double* to, size_t n_edges, int mode, const gp_compound* objects,
                                       int with_self, int32_t* first_bad, uint32_t* n_samples);

/* src/funcs.rs helpers, pure host arithmetic (no planner, no GPU).  Limits cross as three arrays:
 * has_limit[i] != 0 <=> Some(Range::new(lo[i], hi[i])), else None.  precision 32|64 = the type parameter T
 * the arithmetic runs in; values cross as double.
 * generate_clamped_joint_positions_from_limits (funcs.rs:25-54; test funcs.rs:203-211): out[n_limits];
 *   n_angles != n_limits -> GP_ERR_DOF_MISMATCH, as the reference's Err(DofMismatch).
 * generate_random_joint_positions_from_limits (funcs.rs:76-87; test funcs.rs:196-202): k draws, out[k][n_limits],
 *   from the seedable stand-in for rand::random.
 * modify_to_nearest_angle (funcs.rs:98-119; doctest funcs.rs:91-97): vec2 is modified in place. */
int gp_generate_clamped_joint_positions_from_limits(const double* angles, int n_angles, const double* lo, const double* hi,
                                                    const uint8_t* has_limit, int n_limits, int precision, double* out);
int gp_generate_random_joint_positions_from_limits(const double* lo, const double* hi, const uint8_t* has_limit, int n_limits,
                                                   int precision, uint64_t seed, size_t k, double* out);
int gp_modify_to_nearest_angle(const double* vec1, double* vec2, int n, const double* lo, const double* hi,
                               const uint8_t* has_limit, int precision);

#ifdef __cplusplus
}
#endif
#endif /* GEAR_HOST_H */
