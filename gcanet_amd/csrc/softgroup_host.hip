// softgroup_host.hip -- the SoftGroup routines that the reference itself runs on the HOST
// with CPU tensors (softgroup/ops/src/voxelize/voxelize.cpp:11-165,
// bfs_cluster/bfs_cluster.cpp:48-143, hierarchical_aggregation/hierarchical_aggregation.cpp).
// They are part of the drop-in boundary (CPU tensors in, CPU tensors out).  GPU versions
// (sort-based voxel dedup, label-propagation connected components) are the "next" rows
// of SURVEY.md section 8(f).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "common.h"

namespace {

struct VoxKey {
  int32_t k[4];
  bool operator<(const VoxKey &o) const { return std::lexicographical_compare(k, k + 4, o.k, o.k + 4); }
  bool operator==(const VoxKey &o) const { return std::memcmp(k, o.k, sizeof(k)) == 0; }
};

// breadth-first component from `seed`; appends members to `members` in visit order
template <class Accept>
static void bfs_component(int seed, const int32_t *nbr, const int32_t *start_len, std::vector<char> &visited,
                          std::vector<int32_t> &members, Accept accept) {
  size_t head = members.size();
  members.push_back(seed);
  visited[seed] = 1;
  while (head < members.size()) {
    const int cur = members[head++];
    const int s = start_len[cur * 2], len = start_len[cur * 2 + 1];
    for (int t = s; t < s + len; ++t) {
      const int j = nbr[t];
      if (!accept(cur, j) || visited[j]) continue;
      visited[j] = 1;
      members.push_back(j);
    }
  }
}

struct Cluster {
  size_t begin, end;  // range in the flat member array
  float cx, cy, cz;
  int cls, batch;
};

}  // namespace

GCN_EXPORT int gcn_voxelize_idx_host(const int64_t *coords, int N, int ncol, int mode, int32_t *input_map,
                                     int *M, int *maxActive, int64_t *output_coords, int32_t *output_map) {
  GCN_REQUIRE(N >= 0 && (ncol == 3 || ncol == 4), "gcn_voxelize_idx_host: coords must be (N,3) or (N,4), got ncol=%d", ncol);
  GCN_REQUIRE(mode >= 0 && mode <= 4, "gcn_voxelize_idx_host: mode must be 0..4");
  GCN_REQUIRE(M && maxActive && (N == 0 || (coords && input_map)), "gcn_voxelize_idx_host: null pointer");
  GCN_REQUIRE((output_coords == nullptr) == (output_map == nullptr), "gcn_voxelize_idx_host: pass both output buffers or neither");
  // sort point ids by (voxel key, id); the voxel id is the rank of a group's first point
  std::vector<VoxKey> key(N);
  for (int i = 0; i < N; ++i) {
    const int64_t *c = coords + (size_t)i * ncol;
    if (ncol == 3) key[i] = {{0, (int32_t)c[0], (int32_t)c[1], (int32_t)c[2]}};
    else key[i] = {{(int32_t)c[0], (int32_t)c[1], (int32_t)c[2], (int32_t)c[3]}};
  }
  std::vector<int32_t> order(N);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
  std::vector<int32_t> group_first;             // first (lowest) point id of each group, in key order
  std::vector<int32_t> group_of(N);             // group (key order) of each sorted position
  for (int s = 0; s < N; ++s) {
    if (s == 0 || !(key[order[s]] == key[order[s - 1]])) group_first.push_back(order[s]);
    group_of[s] = (int32_t)group_first.size() - 1;
  }
  const int nActive = (int)group_first.size();
  std::vector<int32_t> by_first(nActive);
  std::iota(by_first.begin(), by_first.end(), 0);
  std::sort(by_first.begin(), by_first.end(), [&](int32_t a, int32_t b) { return group_first[a] < group_first[b]; });
  std::vector<int32_t> voxel_id(nActive);
  for (int v = 0; v < nActive; ++v) voxel_id[by_first[v]] = v;
  std::vector<int32_t> count(nActive, 0);
  for (int s = 0; s < N; ++s) {
    const int v = voxel_id[group_of[s]];
    input_map[order[s]] = v;
    count[v]++;
  }
  int maxA = 1;
  if (mode == 3 || mode == 4)
    for (int v = 0; v < nActive; ++v) maxA = std::max(maxA, count[v]);
  *M = nActive;
  *maxActive = maxA;
  if (!output_map) return GCN_OK;
  const int W = maxA + 1;
  std::fill(output_map, output_map + (size_t)nActive * W, 0);
  if (mode == 3 || mode == 4) {
    // sorted order lists a group's points in ascending id == the reference's push_back order
    for (int s = 0; s < N; ++s) {
      const int v = voxel_id[group_of[s]];
      int32_t *row = output_map + (size_t)v * W;
      row[1 + row[0]] = order[s];
      row[0]++;
    }
  } else {
    // mode 0 unique / mode 1 front() / mode 2 back()   (voxelize.cpp:131-151)
    for (int s = 0; s < N; ++s) {
      const int v = voxel_id[group_of[s]];
      int32_t *row = output_map + (size_t)v * W;
      if (row[0] == 0 || mode == 2) row[1] = order[s];
      row[0] = 1;
    }
  }
  for (int v = 0; v < nActive; ++v) {
    const int src = output_map[(size_t)v * W + 1];
    for (int j = 0; j < ncol; ++j) output_coords[(size_t)v * ncol + j] = coords[(size_t)src * ncol + j];
  }
  return GCN_OK;
}

GCN_EXPORT int gcn_bfs_cluster_host(const float *class_numpoint_mean, const int32_t *ball_query_idxs,
                                    const int32_t *start_len, int N, float threshold, int class_id,
                                    int *sumNPoint, int *nCluster, int32_t *cluster_idxs, int32_t *cluster_offsets) {
  GCN_REQUIRE(N >= 0 && sumNPoint && nCluster && class_numpoint_mean, "gcn_bfs_cluster_host: bad argument");
  GCN_REQUIRE(N == 0 || start_len, "gcn_bfs_cluster_host: null pointer");
  GCN_REQUIRE((cluster_idxs == nullptr) == (cluster_offsets == nullptr), "gcn_bfs_cluster_host: pass both output buffers or neither");
  std::vector<char> visited(N, 0);
  std::vector<int32_t> members;
  const float mean = class_numpoint_mean[class_id];
  const float thr = (mean == -1) ? threshold : threshold * mean;  // bfs_cluster.cpp:86-91
  int sum = 0, ncl = 0;
  if (cluster_offsets) cluster_offsets[0] = 0;
  for (int i = 0; i < N; ++i) {
    if (visited[i]) continue;
    members.clear();
    bfs_component(i, ball_query_idxs, start_len, visited, members, [](int, int) { return true; });
    const int sz = (int)members.size();
    if (sz >= thr) {
      if (cluster_idxs) {
        for (int t = 0; t < sz; ++t) {
          cluster_idxs[(size_t)(sum + t) * 2] = ncl;
          cluster_idxs[(size_t)(sum + t) * 2 + 1] = members[t];
        }
        cluster_offsets[ncl + 1] = sum + sz;
      }
      sum += sz;
      ++ncl;
    }
  }
  *sumNPoint = sum;
  *nCluster = ncl;
  return GCN_OK;
}

GCN_EXPORT int gcn_hierarchical_aggregation_host(const int32_t *semantic_label, const float *coord_shift,
                                                 const int32_t *batch_idxs, const int32_t *ball_query_idxs,
                                                 const int32_t *start_len, int N, int using_set_aggr,
                                                 int32_t *cluster_idxs, int32_t *cluster_offsets,
                                                 int *sumNPoint, int *nCluster) {
  GCN_REQUIRE(N >= 0 && cluster_idxs && cluster_offsets && sumNPoint && nCluster, "gcn_hierarchical_aggregation_host: bad argument");
  GCN_REQUIRE(N == 0 || (semantic_label && coord_shift && batch_idxs && start_len), "gcn_hierarchical_aggregation_host: null pointer");
  // hierarchical_aggregation.cpp:7-8
  static const float class_mean[10] = {-1.f, -1.f, 3917.f, 12056.f, 2303.f, 8331.f, 3948.f, 3166.f, 5629.f, 11719.f};
  std::vector<char> visited(N, 0);
  std::vector<int32_t> members;  // all components, flat
  std::vector<Cluster> fragment, kept, primary;
  for (int i = 0; i < N; ++i) {
    if (visited[i]) continue;
    const size_t begin = members.size();
    bfs_component(i, ball_query_idxs, start_len, visited, members,
                  [&](int cur, int j) { return semantic_label[j] == semantic_label[cur]; });
    Cluster c{begin, members.size(), 0.f, 0.f, 0.f, semantic_label[i], batch_idxs[i]};
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (size_t t = begin; t < c.end; ++t) {  // visit order == accumulation order of the reference
      const int p = members[t];
      ax += coord_shift[p * 3]; ay += coord_shift[p * 3 + 1]; az += coord_shift[p * 3 + 2];
    }
    const int sz = (int)(c.end - c.begin);
    GCN_REQUIRE(c.cls >= 0 && c.cls < 10, "gcn_hierarchical_aggregation_host: semantic label %d outside [0,10)", c.cls);
    c.cx = ax / (float)sz; c.cy = ay / (float)sz; c.cz = az / (float)sz;
    const float mean = class_mean[c.cls];
    const float low = (float)(0.05 * mean), high = (float)(0.3 * mean);  // hierarchical_aggregation.cpp:60-61
    if (sz < high) {
      fragment.push_back(c);
      if (sz >= low) kept.push_back(c);
    } else {
      primary.push_back(c);
    }
  }
  // merged output of HierarchicalAggregation.forward (functions.py:52-72): kept, then primary
  int sum = 0, ncl = 0;
  cluster_offsets[0] = 0;
  auto emit = [&](size_t begin, size_t end, int id) {
    for (size_t t = begin; t < end; ++t) {
      cluster_idxs[(size_t)sum * 2] = id;
      cluster_idxs[(size_t)sum * 2 + 1] = members[t];
      ++sum;
    }
  };
  for (const Cluster &c : kept) {
    emit(c.begin, c.end, ncl);
    cluster_offsets[++ncl] = sum;
  }
  std::vector<int> owner;
  if (using_set_aggr && !primary.empty()) {
    // hierarchical_aggregation.cu:22-75: nearest same-class same-batch primary centroid;
    // absorbed if d2 < (0.01*sqrt(npts))^2.  Absorption order here: fragment index order.
    owner.assign(fragment.size(), -1);
    for (size_t f = 0; f < fragment.size(); ++f) {
      float nearest = 10000.f;
      int ni = -1;
      for (size_t i = 0; i < primary.size(); ++i) {
        if (primary[i].cls != fragment[f].cls || primary[i].batch != fragment[f].batch) continue;
        const float dx = primary[i].cx - fragment[f].cx, dy = primary[i].cy - fragment[f].cy,
                    dz = primary[i].cz - fragment[f].cz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < nearest) { nearest = d; ni = (int)i; }
      }
      if (ni < 0) continue;
      const float r = (float)(0.01 * std::sqrt((float)(primary[ni].end - primary[ni].begin)));
      if (nearest < r * r) owner[f] = ni;
    }
  }
  for (size_t i = 0; i < primary.size(); ++i) {
    emit(primary[i].begin, primary[i].end, ncl);
    if (!owner.empty()) {
      int nfrag = 0, npts = 0;  // caps: hierarchical_aggregation.cu:9-12
      for (size_t f = 0; f < fragment.size() && nfrag < 1000; ++f) {
        if (owner[f] != (int)i) continue;
        ++nfrag;
        for (size_t t = fragment[f].begin; t < fragment[f].end && npts < 3000; ++t, ++npts) emit(t, t + 1, ncl);
      }
    }
    cluster_offsets[++ncl] = sum;
  }
  *sumNPoint = sum;
  *nCluster = ncl;
  return GCN_OK;
}
