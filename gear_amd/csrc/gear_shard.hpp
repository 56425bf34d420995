// gear_shard.hpp — how a device group (gh_scene_create_multi) divides a batch: plain arithmetic, no HIP, so that it
// can be unit-tested without a device (tests/host_harness.cpp, tests/test_narrowphase_host.py ... test_host_logic).
// The batch shards naturally: configurations are independent and an edge needs only its own samples;
// shares are contiguous, cover the batch exactly once, and never split an edge.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace gear_hip {

// share k of G of n configurations: [lo, hi)
inline void shard_range(size_t n, size_t k, size_t G, size_t* lo, size_t* hi) {
  *lo = n * k / G;
  *hi = n * (k + 1) / G;
}

// Contiguous shares of whole edges, balanced by samples: share k starts at the first edge whose first sample is at or
// beyond n * k / G.  Returns G + 1 cut points into the edges, cut[0] = 0, cut[G] = n_edges; a share may be empty.
inline std::vector<size_t> shard_edges(const uint32_t* edge_off, size_t n_edges, size_t G) {
  std::vector<size_t> cut(G + 1, n_edges);
  cut[0] = 0;
  const size_t n = n_edges ? edge_off[n_edges] : 0;
  for (size_t k = 1; k < G; k++)
    cut[k] = std::max<size_t>(cut[k - 1], (size_t)(std::lower_bound(edge_off, edge_off + n_edges, (uint32_t)(n * k / G)) - edge_off));
  return cut;
}

}  // namespace gear_hip
