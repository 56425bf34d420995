/*
 * gear_hip.h — C ABI of the MI355X batched collision-validation engine.
 *
 * This is the drop-in boundary for gear's planning hot path.  The reference
 * (openrr/gear) has no FFI of its own: the path is the Rust closure
 * `|q| is_feasible(using_joints, q, objects)` handed to rrt::dual_rrt_connect /
 * rrt::smooth_path (src/planner/joint_path_planner.rs:168-188, 221-241) plus
 * the six public predicates at src/planner/joint_path_planner.rs:71-134 and
 * CollisionChecker::check_env / check_self
 * (src/collision/collision_checker.rs:268-287).  Each entry point below names
 * the reference item it replaces.  A Rust `-sys` crate binds exactly these
 * symbols (tests/test_abi.py checks that the library exports every one of them).
 *
 * Conventions
 *  - every function returns GH_OK (0) or a negative gh_status; nothing throws
 *    across this boundary; gh_last_error() gives the text for the calling
 *    thread's last failure;
 *  - all buffers are caller-owned; `_dev` variants take device pointers and an
 *    optional hipStream_t (as void*), the others take host pointers and are
 *    synchronous;
 *  - a gh_scene is immutable after creation and may be shared by host threads
 *    (every entry point holds the scene's lock while it enqueues; host-buffer
 *    calls hold it until their results are back).  The large-batch kernels'
 *    workspace belongs to the DEVICE, not to a scene: a pool of up to four sets
 *    shared by every scene on that device, each set ordered by an event of its
 *    last use, so asynchronous `_dev` calls may be enqueued on any streams, from
 *    any scenes, in any order — a call that needs a set still in use makes its
 *    stream wait for it.  A scene's own staging buffers serve one call at a
 *    time (the lock);
 *  - scalars in descriptors are f64; `precision` (32|64) selects the arithmetic
 *    type T of the whole path, converting exactly where the reference converts
 *    (na::convert of URDF f64 values, src/collision/urdf.rs:18-27).
 */
#ifndef GEAR_HIP_H
#define GEAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums ------------------------------------------------------------- */

typedef enum gh_status {
  GH_OK = 0,
  GH_ERR_INVALID = -1,      /* malformed descriptor / null pointer            */
  GH_ERR_DOF_MISMATCH = -2, /* cf. gear::Error::DofMismatch (errors.rs:48)    */
  GH_ERR_HIP = -3,          /* a HIP runtime call failed                      */
  GH_ERR_TOO_LARGE = -4,    /* scene exceeds a documented capacity            */
  GH_ERR_NONCONVEX = -5,    /* (not returned any more: see GH_SHAPE_MESH)     */
  GH_ERR_NO_DEVICE = -6     /* HIP device missing: the product never falls
                               back to a CPU path                             */
} gh_status;

enum { GH_JOINT_FIXED = 0, GH_JOINT_ROTATIONAL = 1, GH_JOINT_LINEAR = 2 };

/* urdf_geometry_to_shape_handle, src/collision/urdf.rs:10-73 */
enum {
  GH_SHAPE_BOX = 0,      /* Cuboid(half extents)            urdf.rs:18-25 */
  GH_SHAPE_BALL = 1,     /* Ball(radius)                    urdf.rs:51-53 */
  GH_SHAPE_CYLINDER = 2, /* 30-gon prism TriMesh, Z axis    urdf.rs:26-50 */
  GH_SHAPE_MESH = 3      /* TriMesh, a triangle soup        urdf.rs:54-71.
                          * Distance to a TriMesh is the minimum over its triangles (a surface: a shape
                          * enclosed by it without touching is at a positive distance).  A closed convex
                          * mesh of <= 64 vertices is held as ONE shape (hull + that surface rule); any
                          * other mesh becomes one shape per triangle, and those count towards the
                          * capacities: 1023 robot shapes, 32767 obstacle shapes per scene. */
};

/* per-configuration result byte */
enum {
  GH_FLAG_ENV = 1,  /* has_any_colliding            joint_path_planner.rs:86-98   */
  GH_FLAG_SELF = 2, /* has_any_colliding_with_self  joint_path_planner.rs:123-128 */
  GH_FLAG_LIMIT = 4 /* set_joint_positions Err(_)   joint_path_planner.rs:77-82   */
};

/* `what` argument: which predicates to evaluate.
 * GH_CHECK_NO_LIMITS: skip the joint-limit gate (GH_FLAG_LIMIT is never set and collisions are evaluated
 * for every configuration): has_any_colliding, has_any_colliding_with_self and colliding_link_names*
 * (joint_path_planner.rs:86-110,123-134) never look at the limits, they read whatever state the chain
 * holds; only is_feasible* goes through set_joint_positions (:77,:114).
 * Every entry point ignores bits of `what` that are not declared in this header. */
enum { GH_CHECK_ENV = 1, GH_CHECK_SELF = 2, GH_CHECK_NO_LIMITS = 0x10 };
/* ---- diagnostics: nothing below is needed to use the library ------------------------------------------------
 * Optional `what` bits that only choose HOW the batch is evaluated; the result
 * bytes are identical either way (tests/test_gpu_pipeline.py, tests/test_gpu_small.py).
 * Without them a call of up to 16384 configurations (6144 in f64) takes the single-launch
 * small-batch kernel (a configuration's work spread over many lanes: the RRT
 * lines of joint_path_planner.rs:168-188), larger ones the multi-kernel
 * pipeline; the fused kernel serves scenes whose tables do not fit either.
 * No counterpart in the reference. */
enum { GH_PATH_SMALL = 0x20, GH_PATH_FUSED = 0x1000, GH_PATH_PIPELINE = 0x2000,
       GH_PATH_NO_GRID = 0x4000, /* sweep every obstacle instead of the broadphase grid's cell lists */
       GH_COUNT_CANDIDATES = 0x8000 /* count the broadphase candidates (gh_pipeline_last_stats); implies the pipeline */ };

/* ---- descriptors ------------------------------------------------------- */

/* One k::Node.  Nodes are listed in k::Chain iteration order (depth-first,
 * children visited in reverse insertion order; node 0 is the synthetic fixed
 * "root" joint), which is the order EnvCollisionNames walks
 * (collision_checker.rs:55,75). */
typedef struct gh_node_desc {
  int32_t parent;       /* index into nodes[], -1 for node 0                  */
  int32_t joint_type;   /* GH_JOINT_*                                         */
  double axis[3];       /* unit joint axis                                    */
  double origin_t[3];   /* joint origin translation                           */
  double origin_q[4];   /* joint origin rotation, unit quaternion (i,j,k,w)   */
  int32_t has_limit;    /* 0: continuous or upper==lower                      */
  int32_t mimic_parent; /* node index this joint mimics, -1 if none           */
  double limit_lo, limit_hi;
  double mimic_mult, mimic_offset;
  double position;      /* current joint position (joints not in the batch)   */
} gh_node_desc;

typedef struct gh_shape_desc {
  int32_t kind;     /* GH_SHAPE_*                                             */
  int32_t node;     /* robot: owning node index; obstacle: ignored            */
  double params[3]; /* box: half extents; ball: r; cylinder: r, length        */
  double t[3];      /* pose translation: node-local (robot) / world (obstacle)*/
  double q[4];      /* pose rotation (i,j,k,w)                                */
  int32_t mesh;     /* GH_SHAPE_MESH: index into meshes[]; else -1            */
  int32_t reserved;
} gh_shape_desc;

typedef struct gh_mesh_desc {
  int32_t n_vertices;
  int32_t n_triangles;
  const double* vertices;   /* [n_vertices][3], shape-local                   */
  const int32_t* triangles; /* [n_triangles][3]                               */
} gh_mesh_desc;

/* CollisionChecker + k::Chain flattened (collision_checker.rs:234-262). */
typedef struct gh_robot_desc {
  int32_t n_nodes;
  int32_t n_cols;   /* D: dof of `using_joints`                               */
  int32_t n_shapes; /* L                                                      */
  int32_t n_pairs;  /* P                                                      */
  int32_t n_meshes;
  int32_t reserved;
  const gh_node_desc* nodes;
  const int32_t* col_node;     /* [n_cols] batch column -> node index, in
                                  using_joints.iter_joints() order            */
  const gh_shape_desc* shapes; /* ordered by node, then <collision> order     */
  const int32_t* pairs;        /* [n_pairs][2] node indices (self pairs)      */
  const gh_mesh_desc* meshes;
} gh_robot_desc;

/* Compound<f64>::from_urdf_robot flattened (collision_checker.rs:349-369);
 * order = objects.shapes() order. */
typedef struct gh_obstacle_desc {
  int32_t n_shapes; /* O */
  int32_t n_meshes;
  const gh_shape_desc* shapes;
  const gh_mesh_desc* meshes;
} gh_obstacle_desc;

typedef struct gh_scene gh_scene;

/* instrumentation counters returned by gh_scene_counters (per scene, summed
 * over every instrumented call since creation) */
typedef struct gh_counters {
  uint64_t configs;      /* configurations processed                          */
  uint64_t pairs;        /* shape pairs considered (env + self)               */
  uint64_t cull_pass;    /* pairs surviving the bounding-sphere cull          */
  uint64_t sat_tests;    /* box-box SAT evaluations                           */
  uint64_t exact_tests;  /* GJK margin queries                                */
  uint64_t exact_iters;  /* GJK iterations                                    */
  uint64_t reserved[2];
} gh_counters;

/* ---- entry points ------------------------------------------------------ */

/* Replaces JointPathPlannerBuilder::finalize + CollisionChecker::from_urdf_robot
 * + Compound::from_urdf_robot (joint_path_planner.rs:314-328,
 * collision_checker.rs:234-262,349-369): flattens the robot and the obstacle
 * compound into SoA tables resident in HBM on `device`. */
int gh_scene_create(const gh_robot_desc* robot, const gh_obstacle_desc* obstacles,
                    double margin, int precision, int device, gh_scene** out);
void gh_scene_destroy(gh_scene* scene);

/* The same scene on several devices of one node (no counterpart in the reference, which is single-threaded:
 * this is the sharding SURVEY.md 8(e) describes, inside the library).  The scene tables are replicated on
 * every listed device; gh_check_batch, gh_check_edges and gh_fk_batch then give each device a contiguous
 * share of the batch (whole edges; balanced by samples), run the shares concurrently — one host thread,
 * stream, staging area and workspace per device — and write the results into the caller's buffers.  No
 * collective is involved.  The diagnostics (gh_link_hits, gh_self_hits, gh_pipeline_last_stats) are
 * served by the first device; the `_dev` entry points and gh_check_batch_counted take device pointers and
 * therefore a single-device scene.  A device may be listed more than once (each entry is a share). */
int gh_scene_create_multi(const gh_robot_desc* robot, const gh_obstacle_desc* obstacles, double margin,
                          int precision, const int* devices, int n_devices, gh_scene** out);
int gh_scene_device_count(const gh_scene* scene); /* 1 for gh_scene_create */

int gh_scene_dof(const gh_scene* scene);       /* D  */
int gh_scene_nodes(const gh_scene* scene);     /* N  */
int gh_scene_precision(const gh_scene* scene); /* 32|64 */
int gh_scene_device(const gh_scene* scene);

/* Replaces the closure |q| is_feasible(..) / is_feasible_with_self(..) applied
 * to n configurations (joint_path_planner.rs:71-84,113-121).
 * q: [n][D] row-major, float or double per scene precision.
 * flags[i]: GH_FLAG_* bits; a limit violation yields GH_FLAG_LIMIT alone
 * (the reference does not evaluate collisions then).
 *   is_feasible           == !(flags & (GH_FLAG_ENV  | GH_FLAG_LIMIT))
 *   is_feasible_with_self == !(flags & (GH_FLAG_SELF | GH_FLAG_LIMIT))
 * gh_check_batch takes host arrays, pageable or page-locked (large batches move about twice as fast from
 * page-locked ones) and returns when flags[] is written; gh_check_batch_dev takes device arrays and a stream. */
int gh_check_batch(gh_scene* scene, const void* q, size_t n, uint32_t what, uint8_t* flags);
int gh_check_batch_dev(gh_scene* scene, const void* q_dev, size_t n, uint32_t what,
                       uint8_t* flags_dev, void* hip_stream);

/* gh_check_batch_dev for callers that must not allocate.  It has no counterpart in the reference; it is
 * for code that cannot call hipMalloc while it runs: under HIP graph capture, in a tight serving loop, in a
 * harness that forbids it.  Same arguments and the same bytes in flags_dev as gh_check_batch_dev, but the
 * call makes no hipMalloc / hipFree (nor any other allocating call): the workspace of its kernels (hit
 * words, the exact-stage entries of every sweep wavefront, counters, pose columns) is cut from one 64 MiB
 * array that is part of the library's code object on every GPU, from the moment the code is loaded there
 * until the process ends; its address on a device is looked up when a scene is created on that device.
 * Limits: n <= 65536, and the workspace this scene needs for n configurations must fit the 64 MiB
 * (GH_ERR_TOO_LARGE otherwise, with the number of configurations that do fit).  Every call on a GPU, from
 * any scene, uses that GPU's one array.  The library orders them: an event per device is recorded behind each
 * call and the next call's stream waits for it, so calls made one after the other — from one scene or several,
 * on the scenes' own streams (hip_stream = NULL) or the caller's — never overlap on the array.  Calls enqueued
 * at the same time from several host threads run in the order they took the library's lock.  The exact-stage
 * capacity is the one the scene was created with (it does not grow here).  gh_pipeline_last_stats does not
 * see these calls. */
int gh_check_batch_noalloc(gh_scene* scene, const void* q_dev, size_t n, uint32_t what,
                           uint8_t* flags_dev, void* hip_stream);

/* Same, plus the first infeasible sample of each edge: the batched form of
 * rrt's stop-at-first-failure stepping (joint_path_planner.rs:168-188).
 * edge_off: [E+1] offsets into the n = edge_off[E] samples.
 * first_bad[e]: index within edge e of the first sample with
 * (flags & bad_mask) != 0, or -1. */
int gh_check_edges(gh_scene* scene, const void* q, const uint32_t* edge_off, size_t n_edges,
                   uint32_t what, uint8_t bad_mask, uint8_t* flags, int32_t* first_bad);
int gh_check_edges_dev(gh_scene* scene, const void* q_dev, const uint32_t* edge_off_dev,
                       size_t n_edges, size_t n_samples, uint32_t what, uint8_t bad_mask,
                       uint8_t* flags_dev, int32_t* first_bad_dev, void* hip_stream);

/* Edge batches from their END POINTS.  No counterpart in the reference as a function: it is the stepping of
 * rrt 0.4.0 that the reference's closure is called from (joint_path_planner.rs:168-188; restated in
 * gear_amd/host/gear_host.hpp connect_line / smooth_path), replayed on the device so that a large batch of edges
 * crosses PCIe as 2 x D values per edge instead of sample by sample.  One lane per edge runs the ITERATED recurrence
 *     d = sqrt(sum_i (cur_i - to_i)^2);   next_i = cur_i + (to_i - cur_i) * step / d
 * in the scene's precision with IEEE division and square root and no contraction: the samples are bit-identical to
 * the host stepping's (tests/test_gpu_edges_endpoints.py).  mode:
 *   GH_EDGE_CONNECT  Tree::connect: extend() repeatedly until a sample lies within `step` of `to`; a step that
 *                    would overshoot lands on `to` itself;
 *   GH_EDGE_WALK     smooth_path's walker: advance by `step` while at least `step` away from `to`; `to` itself is
 *                    not a sample (an edge shorter than `step` has none).
 * from, to: [n_edges][D].  first_bad[e]: as gh_check_edges.  n_samples[e]: samples of edge e; bit 31
 * (GH_EDGE_TRUNCATED) is set when the edge had not ended after max_len samples — its first max_len samples were
 * evaluated.  Meant for LARGE batches: for an RRT line or two the host computes the samples faster than one lane
 * does (the recurrence is serial), which is why the planner of gear_host.hpp keeps gh_check_edges for its calls.
 * The _dev form takes device arrays; the samples are written to q_dev (capacity cap_samples rows, as are flags_dev's
 * bytes), their offsets to edge_off_dev[n_edges + 1]; it synchronises `hip_stream` ONCE (the sample count decides
 * what is launched) and reports the count in *total_samples (GH_ERR_TOO_LARGE if it exceeds cap_samples). */
enum { GH_EDGE_CONNECT = 0, GH_EDGE_WALK = 1 };
#define GH_EDGE_TRUNCATED 0x80000000u
int gh_check_edges_from_endpoints(gh_scene* scene, const void* from, const void* to, size_t n_edges, double step,
                                  uint32_t max_len, uint32_t mode, uint32_t what, uint8_t bad_mask,
                                  int32_t* first_bad, uint32_t* n_samples);
int gh_check_edges_from_endpoints_dev(gh_scene* scene, const void* from_dev, const void* to_dev, size_t n_edges,
                                      double step, uint32_t max_len, uint32_t mode, uint32_t what, uint8_t bad_mask,
                                      void* q_dev, size_t cap_samples, uint32_t* edge_off_dev, uint32_t* n_samples_dev,
                                      uint8_t* flags_dev, int32_t* first_bad_dev, size_t* total_samples, void* hip_stream);

/* Replaces k::Chain::set_joint_positions + update_transforms as used at
 * collision_checker.rs:54,124 and joint_path_planner.rs:245-247.
 * poses: [n][N][7] = tx,ty,tz,qi,qj,qk,qw of every node's world transform.
 * Limits are not checked here (set_joint_positions_unchecked semantics). */
int gh_fk_batch(gh_scene* scene, const void* q, size_t n, void* poses);
int gh_fk_batch_dev(gh_scene* scene, const void* q_dev, size_t n, void* poses_dev, void* hip_stream);

/* Replaces colliding_link_names (joint_path_planner.rs:100-110), the
 * non-early-exit diagnostic.  link_mask: [n][O][W] with W = gh_link_mask_words()
 * 64-bit words; bit k of the mask for (config i, obstacle o) is set when node k
 * would be yielded by check_env for obstacle o.  Decoding obstacle-major, then
 * ascending node index, reproduces the reference's Vec<String> with its
 * duplicates.  what: 0, or GH_CHECK_NO_LIMITS for the reference's own behaviour
 * (no limit check at all); without it a limit violation yields all-zero masks,
 * the batched counterpart of is_feasible's early `false`. */
int gh_link_hits(gh_scene* scene, const void* q, size_t n, uint32_t what, uint64_t* link_mask);
int gh_link_mask_words(const gh_scene* scene);

/* Replaces colliding_link_names_with_self (joint_path_planner.rs:130-134).
 * pair_mask: [n][ceil(P/64)] words; bit p set when self pair p is yielded. */
int gh_self_hits(gh_scene* scene, const void* q, size_t n, uint32_t what, uint64_t* pair_mask);

/* Instrumented run (pairs culled / SAT'd / exact-tested; SURVEY.md 8d's s,g). */
int gh_check_batch_counted(gh_scene* scene, const void* q_dev, size_t n, uint32_t what,
                           uint8_t* flags_dev, gh_counters* out);

/* Diagnostic, no counterpart in the reference: counters of the last call on this scene that carried
 * GH_COUNT_CANDIDATES in `what` (the counters live in the device's shared workspace: read them before
 * another scene of the same device runs a large batch), summed over the last chunk of each workspace set it used (synchronises
 * the device).  out[0] broadphase candidates tested, out[1] items handed to the exact stage, out[2]
 * sweep wavefronts whose exact region ran full (their configurations were re-checked by the fused
 * kernel), out[3] exact entries a sweep wavefront may write. */
int gh_pipeline_last_stats(gh_scene* scene, uint64_t out[4]);

const char* gh_last_error(void);
const char* gh_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GEAR_HIP_H */
