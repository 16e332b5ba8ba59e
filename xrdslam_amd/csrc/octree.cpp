// Sparse voxel octree for Vox-Fusion's map (host side).
//
// Behaviour restated from the reference's TorchScript class
// third_party/sparse_octree/src/octree.cpp:16-393 (+ include/octree.h,
// include/utils.h Morton helpers): every voxel inserts its 8 corner lattice
// points down to the leaf level; a leaf is SURFACE when it was reached as
// corner 0 of some voxel, FEATURE otherwise; NODE IDS ARE THE CREATION ORDER
// (a process-global counter, octree.cpp:9) because the embedding table is
// indexed by them (slam/models/sparse_voxel.py:309-315) — ids must therefore
// be bit-exact.  Not a copy: nodes live in one flat array (id-indexed, 8
// int32 children each) instead of heap-allocated pointer nodes, and voxels
// that were already inserted are skipped through a hash set (the reference
// re-walks 8 corners x 8 levels for every one of the 307 200 pixels of every
// frame: 60-80 ms; identical result because re-inserting a voxel creates no
// node and changes no type).
#include <stdint.h>

#include <cmath>
#include <queue>
#include <unordered_set>
#include <vector>

#include "xrdslam_hip.h"

namespace {

constexpr int kMaxBits = 21;
enum : int8_t { NONLEAF = -1, SURFACE = 0, FEATURE = 1 };

// corner order of the reference (octree.cpp:12-14): x slowest, z fastest
constexpr int kIncX[8] = {0, 0, 0, 0, 1, 1, 1, 1};
constexpr int kIncY[8] = {0, 0, 1, 1, 0, 0, 1, 1};
constexpr int kIncZ[8] = {0, 1, 0, 1, 0, 1, 0, 1};

inline uint64_t expand21(uint64_t v) {
  uint64_t x = v & 0x1fffff;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
inline uint64_t compact21(uint64_t v) {
  uint64_t x = v & 0x1249249249249249ull;
  x = (x | x >> 2) & 0x10c30c30c30c30c3ull;
  x = (x | x >> 4) & 0x100f00f00f00f00full;
  x = (x | x >> 8) & 0x1f0000ff0000ffull;
  x = (x | x >> 16) & 0x1f00000000ffffull;
  x = (x | x >> 32) & 0x1fffff;
  return x;
}
// MASK[i] of include/utils.h:41-62: top 3*(i+1) bits below the sign bit
inline uint64_t level_mask(int i) {
  uint64_t m = 0;
  for (int k = 0; k <= i; ++k) m |= 0x7000000000000000ull >> (3 * k);
  return m;
}
inline uint64_t encode(int x, int y, int z) {
  const uint64_t code = expand21((uint64_t)(int64_t)x) |
                        (expand21((uint64_t)(int64_t)y) << 1) |
                        (expand21((uint64_t)(int64_t)z) << 2);
  return code & level_mask(kMaxBits - 1);
}

int g_next_index = 0;  // shared by all trees, like Octant::next_index_

struct Node {
  uint64_t code;
  int32_t child[8];
  uint32_t side;
  int32_t index;
  int8_t type;
  bool is_leaf;
};

struct Tree {
  int size = 0, feat_dim = 0, max_level = 0;
  double voxel_size = 0;
  std::vector<Node> nodes;                 // local slot order = creation order
  std::unordered_set<uint64_t> all_keys;   // corner lattice keys
  std::unordered_set<uint64_t> seen_voxel; // voxels inserted before

  int new_node() {
    Node n;
    n.code = 0;
    for (int i = 0; i < 8; ++i) n.child[i] = -1;
    n.side = 0;
    n.index = g_next_index++;
    n.type = NONLEAF;
    n.is_leaf = false;
    nodes.push_back(n);
    return (int)nodes.size() - 1;
  }
  // slot of the leaf containing integer coordinate (x,y,z), or -1
  int find(int x, int y, int z) const {
    int n = 0;
    unsigned edge = size / 2;
    for (int d = 1; d <= max_level; edge /= 2, ++d) {
      const int cid = ((x & edge) > 0) + 2 * ((y & edge) > 0) + 4 * ((z & edge) > 0);
      const int c = nodes[n].child[cid];
      if (c < 0) return -1;
      n = c;
    }
    return n;
  }
};

}  // namespace

extern "C" {

void* xrd_octree_create(int grid_dim, int feat_dim, double voxel_size) {
  if (grid_dim < 2) return nullptr;
  Tree* t = new Tree();
  t->size = grid_dim;
  t->feat_dim = feat_dim;
  t->voxel_size = voxel_size;
  t->max_level = (int)std::log2((double)grid_dim);
  const int r = t->new_node();
  t->nodes[r].side = (uint32_t)grid_dim;
  return t;
}

void xrd_octree_destroy(void* h) { delete static_cast<Tree*>(h); }

void xrd_octree_reset_id_counter(void) { g_next_index = 0; }

int xrd_octree_insert(void* h, const int32_t* vox, int64_t n,
                      int* created_any) {
  Tree* t = static_cast<Tree*>(h);
  if (!t || (n > 0 && !vox) || n < 0) return XRD_ERR_ARG;
  bool created = false;
  const int shift = kMaxBits - t->max_level - 1;
  for (int64_t i = 0; i < n; ++i) {
    const int vx = vox[3 * i], vy = vox[3 * i + 1], vz = vox[3 * i + 2];
    if ((unsigned)vx < 0x1ffffeu && (unsigned)vy < 0x1ffffeu &&
        (unsigned)vz < 0x1ffffeu) {
      const uint64_t vkey = (uint64_t)vx | ((uint64_t)vy << 21) |
                            ((uint64_t)vz << 42);
      // seen before: all 8 corners exist and corner 0 is already SURFACE
      if (!t->seen_voxel.insert(vkey).second) continue;
    }
    for (int j = 0; j < 8; ++j) {
      const int x = vx + kIncX[j], y = vy + kIncY[j], z = vz + kIncZ[j];
      const uint64_t key = encode(x, y, z);
      t->all_keys.insert(key);
      int cur = 0;
      unsigned edge = t->size / 2;
      for (int d = 1; d <= t->max_level; edge /= 2, ++d) {
        const int cid = ((x & edge) > 0) + 2 * ((y & edge) > 0) + 4 * ((z & edge) > 0);
        int c = t->nodes[cur].child[cid];
        if (c < 0) {
          c = t->new_node();
          Node& nn = t->nodes[c];
          nn.code = key & level_mask(d + shift);
          nn.side = edge;
          nn.is_leaf = (d == t->max_level);
          nn.type = nn.is_leaf ? (j == 0 ? SURFACE : FEATURE) : NONLEAF;
          t->nodes[cur].child[cid] = c;
          created = true;
        } else if (t->nodes[c].type == FEATURE && j == 0) {
          t->nodes[c].type = SURFACE;
        }
        cur = c;
      }
    }
  }
  if (created_any) *created_any = created ? 1 : 0;
  return XRD_OK;
}

double xrd_octree_try_insert(void* h, const int32_t* vox, int64_t n) {
  Tree* t = static_cast<Tree*>(h);
  if (!t || (n > 0 && !vox) || n <= 0) return -1.0;
  std::unordered_set<uint64_t> tmp;
  for (int64_t i = 0; i < n; ++i)
    for (int j = 0; j < 8; ++j)
      tmp.insert(encode(vox[3 * i] + kIncX[j], vox[3 * i + 1] + kIncY[j],
                        vox[3 * i + 2] + kIncZ[j]));
  size_t both = 0;
  for (uint64_t k : tmp) both += t->all_keys.count(k);
  return 1.0 * (double)both / (double)tmp.size();
}

int xrd_octree_has_voxel(void* h, const int32_t* xyz) {
  Tree* t = static_cast<Tree*>(h);
  if (!t || !xyz) return 0;
  return t->find(xyz[0], xyz[1], xyz[2]) >= 0 ? 1 : 0;
}

int64_t xrd_octree_count_nodes(void* h) {
  Tree* t = static_cast<Tree*>(h);
  return t ? (int64_t)t->nodes.size() : -1;
}

int64_t xrd_octree_count_leaf_nodes(void* h) {
  // octree.cpp:372-393: SURFACE nodes
  Tree* t = static_cast<Tree*>(h);
  if (!t) return -1;
  int64_t c = 0;
  for (const Node& n : t->nodes) c += (n.type == SURFACE);
  return c;
}

// get_centres_and_children (octree.cpp:297-346).  Arrays have
// xrd_octree_count_nodes rows, indexed by node id; rows of FEATURE leaves stay
// (0,0,0,0) / -1 / -1 because the BFS never visits them.
int xrd_octree_export(void* h, float* voxels, float* children,
                      int32_t* features) {
  Tree* t = static_cast<Tree*>(h);
  if (!t || !voxels || !children || !features) return XRD_ERR_ARG;
  const int64_t T = (int64_t)t->nodes.size();
  for (int64_t i = 0; i < T * 4; ++i) voxels[i] = 0.f;
  for (int64_t i = 0; i < T * 8; ++i) {
    children[i] = -1.f;
    features[i] = -1;
  }
  std::queue<int> q;
  q.push(0);
  while (!q.empty()) {
    const int s = q.front();
    q.pop();
    const Node& nd = t->nodes[s];
    const int64_t row = nd.index;
    if (row < 0 || row >= T) return XRD_ERR_UNSUPPORTED;  // foreign id counter
    const float cx = (float)compact21(nd.code >> 0),
                cy = (float)compact21(nd.code >> 1),
                cz = (float)compact21(nd.code >> 2);
    voxels[row * 4 + 0] = cx;
    voxels[row * 4 + 1] = cy;
    voxels[row * 4 + 2] = cz;
    voxels[row * 4 + 3] = (float)nd.side;
    if (nd.type == SURFACE) {
      for (int i = 0; i < 8; ++i) {
        const int f = t->find((int)(cx + kIncX[i]), (int)(cy + kIncY[i]),
                              (int)(cz + kIncZ[i]));
        if (f >= 0) features[row * 8 + i] = t->nodes[f].index;
      }
    }
    for (int i = 0; i < 8; ++i) {
      const int c = nd.child[i];
      if (c >= 0 && t->nodes[c].type != FEATURE) {
        q.push(c);
        children[row * 8 + i] = (float)t->nodes[c].index;
      }
    }
  }
  return XRD_OK;
}

// get_voxels (octree.cpp:232-255): DFS pre-order, (x,y,z,side) of every node
int64_t xrd_octree_get_voxels(void* h, float* out, int64_t cap_rows) {
  Tree* t = static_cast<Tree*>(h);
  if (!t) return -1;
  std::vector<int> st{0};
  int64_t r = 0;
  while (!st.empty()) {
    const int s = st.back();
    st.pop_back();
    const Node& nd = t->nodes[s];
    if (out && r < cap_rows) {
      out[r * 4 + 0] = (float)compact21(nd.code >> 0);
      out[r * 4 + 1] = (float)compact21(nd.code >> 1);
      out[r * 4 + 2] = (float)compact21(nd.code >> 2);
      out[r * 4 + 3] = (float)nd.side;
    }
    ++r;
    for (int i = 7; i >= 0; --i)
      if (nd.child[i] >= 0) st.push_back(nd.child[i]);
  }
  return r;
}

// get_leaf_voxels (octree.cpp:203-230): DFS order, SURFACE leaves, (x,y,z)
int64_t xrd_octree_get_leaf_voxels(void* h, float* out, int64_t cap_rows) {
  Tree* t = static_cast<Tree*>(h);
  if (!t) return -1;
  std::vector<int> st{0};
  int64_t r = 0;
  while (!st.empty()) {
    const int s = st.back();
    st.pop_back();
    const Node& nd = t->nodes[s];
    if (nd.is_leaf && nd.type == SURFACE) {
      if (out && r < cap_rows) {
        out[r * 3 + 0] = (float)compact21(nd.code >> 0);
        out[r * 3 + 1] = (float)compact21(nd.code >> 1);
        out[r * 3 + 2] = (float)compact21(nd.code >> 2);
      }
      ++r;
      continue;
    }
    for (int i = 7; i >= 0; --i)
      if (nd.child[i] >= 0) st.push_back(nd.child[i]);
  }
  return r;
}

}  // extern "C"
