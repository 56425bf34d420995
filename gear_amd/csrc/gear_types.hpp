// gear_types.hpp — plain tables shared by the host-side scene builder and the
// gfx950 kernels.  A gh_scene is these tables, flattened once at
// gh_scene_create and resident in HBM for the life of the scene.
#pragma once
#include <stdint.h>

namespace gear_hip {

enum : int { SK_BOX = 0, SK_BALL = 1, SK_MESH = 2 };
enum : int { JT_FIXED = 0, JT_ROT = 1, JT_LIN = 2 };

// Where a node's parent world pose comes from during the depth-first FK walk.
enum : int { PARENT_PREV = -1, PARENT_IDENTITY = -2 };

// One k::Node (reference: k::Chain::update_transforms, called at
// src/collision/collision_checker.rs:54,124).
template <class T>
struct NodeDev {
  T ot[3];          // joint origin translation
  T oq[4];          // joint origin rotation (i,j,k,w)
  T axis[3];        // unit joint axis
  T mult, off;      // mimic: position = off + mult * q[col]
  T const_pos;      // position when no batch column drives the node
  int32_t joint_type;
  int32_t col;         // driving batch column, -1: constant
  int32_t direct;      // 1: position = q[col] exactly (not through a mimic)
  int32_t parent_slot; // PARENT_PREV, PARENT_IDENTITY or LDS slot holding the parent's world pose
  int32_t save_slot;   // LDS slot to keep this node's world pose in, -1: none
  int32_t shape_begin, shape_end;  // robot shapes owned by this node
  int32_t pad;
};

// One k::joint::Range::is_valid check (k::Chain::set_joint_positions, called at
// src/planner/joint_path_planner.rs:77,114).
template <class T>
struct LimitDev {
  T mult, off, lo, hi;
  int32_t col;
  int32_t direct;
};

// One (ShapeHandle, Isometry3) — robot shapes carry a node-local pose, obstacle
// shapes a world pose (src/collision/collision_checker.rs:30, 349-369).
template <class T>
struct ShapeDev {
  T lt[3];    // pose translation
  T lq[4];    // pose rotation (i,j,k,w)
  T half[3];  // box half extents; ball: half[0] = radius
  T bc[3];    // bounding-sphere centre, shape-local
  T br;       // bounding-sphere radius
  int32_t kind;
  int32_t node;
  int32_t v_begin, v_count;  // mesh vertices  (T[3] each) in the scene's vertex pool
  int32_t f_begin, f_count;  // mesh face planes (nx,ny,nz,offset) in the plane pool
  int32_t prism_n;  // > 0: the mesh is the regular n-gon prism of urdf.rs:26-50 (vertices: bottom ring, then top ring)
  int32_t pad;
};

struct PairDev {
  int32_t s1_begin, s1_end;  // shapes of joint 1
  int32_t s2_begin, s2_end;  // shapes of joint 2
};

// One uniform grid over the obstacle field for a class of robot-shape radii (broadphase of the
// large-batch sweep).  Cell (ix,iy,iz) lists every obstacle whose bounding sphere, inflated by the
// margin and by the class radius, reaches the cell's box: a robot shape of that class whose centre
// falls in the cell can only be within reach of listed obstacles.
template <class T>
struct GridDev {
  T org[3];        // world position of the grid's low corner
  T inv_h;         // 1 / cell size
  int32_t nx, ny, nz;
  int32_t off_base;  // block index of this grid's cell (0,0,0) in cell_blk[] (nx*ny*nz head blocks)
};

// One self shape pair as the sweep meets it (at its later shape): the pair's index, the LDS slot of its earlier
// shape (-1: both are the same shape) and the inflated radius sum — one 16-byte scalar load per pair test.
template <class T> struct SpLate;
template <> struct alignas(16) SpLate<float> { int32_t p, fs; float rr, pad; };
template <> struct alignas(16) SpLate<double> { int32_t p, fs; double rr; };

struct CountersDev {
  unsigned long long configs, pairs, cull_pass, sat_tests, exact_tests, exact_iters, r0, r1;
};

// Everything a kernel needs, passed by value.  Table pointers are re-qualified
// into the constant address space inside the kernels so that wave-uniform
// reads go through the scalar cache.
template <class T>
struct SceneDev {
  int32_t n_nodes, n_cols, n_shapes, n_obs, n_pairs, n_limits, n_slots, pad;
  T margin;
  const NodeDev<T>* nodes;
  const LimitDev<T>* limits;
  const ShapeDev<T>* rshapes;
  const ShapeDev<T>* oshapes;
  const T* ocull;  // [n_obs + 1][4]: world bounding sphere cx,cy,cz and radius + margin (pre-inflated); last = far dummy
  // cull records for the batched kernels in quads of 4 obstacles, component-major inside a quad:
  // [n_obs_pad / 4][5][4] = -2cx, -2cy, -2cz, -2R, k  with R = obstacle radius + margin (inflated) and
  // k = |c|^2 - R^2 - tau; n_obs_pad is a multiple of 32 (tail = far dummies).  Then
  //   d^2 - (rho + R)^2 = (|p|^2 - rho^2) + k + p . (-2c) + rho * (-2R)      (five VALU ops per pair)
  // and a quad is 20 consecutive dwords: wide scalar loads, one quad ahead of the arithmetic.
  const T* ocullq;
  int32_t n_obs_pad, pad2;
  const PairDev* pairs;
  const int32_t* spairs;  // [n_spairs][4]: s1, s2, node-pair index, LDS pose slot of the earlier shape (-1: s1 == s2)
  const T* spair_r;       // [n_spairs]: inflated bounding-radius sum + margin
  int32_t n_spairs, n_first;
  // self shape pairs grouped by the shape the FK walk reaches LATER (the sweep tests a pair when that
  // shape's pose is live; the earlier shape's pose waits in an LDS slot)
  const int32_t* sp_first_slot;  // [n_shapes] LDS pose slot of a shape that is the earlier one of some pair, else -1
  const int32_t* sp_late_off;    // [n_shapes + 1] CSR offsets into sp_late
  const int32_t* sp_late;        // [n_spairs] self shape pair indices
  const SpLate<T>* sp_late_rec;  // [n_spairs] the same entries with what the sweep needs of each pair
  // what stage 1 of the narrow phase needs of a shape, compact: robot shapes {half[3], kind} (4 T each),
  // then obstacle shapes {t[3], kind | q[4] | half[3], 0} (12 T each); kinds stored as T(kind)
  const T* hot;                  // [n_shapes * 4 + n_obs * 12]
  const T* verts;   // [..][3]
  const T* planes;  // [..][4]
  // broadphase grids (n_grids == 0: none built)
  const GridDev<T>* grids;      // [n_grids]
  const int32_t* shape_grid;    // [n_shapes] grid of each robot shape
  // Cell lists as chains of 16-byte blocks of eight 16-bit obstacle indices (unused slots hold the
  // sentinel n_obs, whose ocull record is a far dummy).  Bit 31 of a block's last dword set: the block
  // holds six indices and the low 31 bits are the index of the next block.  Block 0 is empty (lanes
  // outside a grid read it); a grid's head blocks are contiguous from off_base, one per cell.
  const uint32_t* cell_blk;     // [n_blocks][4]
  int32_t n_grids;
  int32_t all_blk;  // head block of the chain that lists every obstacle (GH_PATH_NO_GRID), -1: none
  // the mesh pools once more as 4-wide records, vertices {x,y,z,0} then face planes {nx,ny,nz,offset}:
  // the exact stage (k_narrow2) copies this table to LDS when it fits, and reads a vertex or a plane
  // with one 16-byte access
  const T* mesh4;               // [(n_verts + n_planes)][4]
  int32_t n_verts, n_planes;
  // obstacle shapes as the CALLER counts them: a mesh that is not a closed convex polytope is one device shape
  // per triangle (gear_scene.hpp flatten_shape); obs_owner[o] is the caller's index of device obstacle o
  const int32_t* obs_owner;     // [n_obs]
  int32_t n_obs_user, pad3;
};

}  // namespace gear_hip
