// TorchScript custom class `torch.classes.svo.Octree` on top of the C-ABI
// octree (xrd_octree_*, csrc/octree.cpp) — the seam the reference uses:
//   slam/models/sparse_voxel.py:22-35  torch.classes.load_library(<*.so under
//                                      third_party/sparse_octree/build/>)
//   slam/models/sparse_voxel.py:307    torch.classes.svo.Octree()
// with the method surface of third_party/sparse_octree/src/bindings.cpp:4-31
// (init / insert / try_insert / get_voxels / get_leaf_voxels / get_features /
// count_nodes / count_leaf_nodes / has_voxel / get_centres_and_children and
// the (size, feat_dim, voxel_size, all_pts) pickle state).  The shared object
// either goes where the reference looks for it or is loaded by
// xrdslam_amd.compat.install().  Host code only; built with
// torch.utils.cpp_extension against libxrdslam_hip.so.
#include <torch/custom_class.h>
#include <torch/script.h>

#include <tuple>
#include <vector>

#include "xrdslam_hip.h"

namespace {

torch::Tensor as_int_cpu(const torch::Tensor& t) {
  // the reference reads accessor<int, 2>(): int32 only
  TORCH_CHECK(t.scalar_type() == torch::kInt,
              "expected scalar type Int but found ", t.scalar_type());
  return t.cpu().contiguous();
}

class Octree : public torch::CustomClassHolder {
 public:
  Octree() = default;
  // unpickle: like the reference's "temporal solution" constructor
  // (octree.cpp:20-28) the id counter restarts and the batches are replayed
  Octree(int64_t grid_dim, int64_t feat_dim, double voxel_size,
         std::vector<torch::Tensor> all_pts) {
    xrd_octree_reset_id_counter();
    init(grid_dim, feat_dim, voxel_size);
    for (auto& pt : all_pts) insert(pt);
  }
  ~Octree() override {
    if (h_) xrd_octree_destroy(h_);
  }
  void init(int64_t grid_dim, int64_t feat_dim, double voxel_size) {
    if (h_) xrd_octree_destroy(h_);
    size_ = grid_dim;
    feat_dim_ = feat_dim;
    voxel_size_ = voxel_size;
    h_ = xrd_octree_create((int)grid_dim, (int)feat_dim, voxel_size);
    TORCH_CHECK(h_ != nullptr, "xrd_octree_create failed");
  }
  void insert(torch::Tensor pts) {
    if (!h_) {
      std::cout << "Octree not initialized!" << std::endl;
      return;
    }
    if (pts.dim() != 2 || pts.size(1) != 3) {
      std::cout << "Point dimensions mismatch: inputs are "
                << (pts.dim() ? pts.size(-1) : 0) << " expect 3" << std::endl;
      return;
    }
    auto v = as_int_cpu(pts);
    int created = 0;
    TORCH_CHECK(xrd_octree_insert(h_, v.data_ptr<int>(), v.size(0),
                                  &created) == XRD_OK,
                "xrd_octree_insert: ", xrd_last_error());
    if (created) all_pts.push_back(v);
  }
  double try_insert(torch::Tensor pts) {
    if (!h_ || pts.dim() != 2 || pts.size(1) != 3) return -1.0;
    auto v = as_int_cpu(pts);
    return xrd_octree_try_insert(h_, v.data_ptr<int>(), v.size(0));
  }
  torch::Tensor get_voxels() {
    const int64_t n = xrd_octree_get_voxels(h_, nullptr, 0);
    auto out = torch::zeros({n, 4}, torch::kFloat);
    xrd_octree_get_voxels(h_, out.data_ptr<float>(), n);
    return out;
  }
  torch::Tensor get_leaf_voxels() {
    const int64_t n = xrd_octree_get_leaf_voxels(h_, nullptr, 0);
    auto out = torch::zeros({n, 3}, torch::kFloat);
    xrd_octree_get_leaf_voxels(h_, out.data_ptr<float>(), n);
    return out;
  }
  // the reference declares it and leaves the body empty (octree.cpp:212-214)
  torch::Tensor get_features(torch::Tensor) { return torch::Tensor(); }
  int64_t count_nodes() { return h_ ? xrd_octree_count_nodes(h_) : 0; }
  int64_t count_leaf_nodes() { return h_ ? xrd_octree_count_leaf_nodes(h_) : 0; }
  bool has_voxel(torch::Tensor pt) {
    auto v = as_int_cpu(pt).reshape({-1});
    if (!h_ || v.size(0) != 3) return false;
    return xrd_octree_has_voxel(h_, v.data_ptr<int>()) != 0;
  }
  std::tuple<torch::Tensor, torch::Tensor, torch::Tensor>
  get_centres_and_children() {
    const int64_t n = count_nodes();
    auto vox = torch::empty({n, 4}, torch::kFloat);
    auto ch = torch::empty({n, 8}, torch::kFloat);
    auto ft = torch::empty({n, 8}, torch::kInt);
    TORCH_CHECK(xrd_octree_export(h_, vox.data_ptr<float>(),
                                  ch.data_ptr<float>(),
                                  ft.data_ptr<int>()) == XRD_OK,
                "xrd_octree_export: ", xrd_last_error());
    return std::make_tuple(vox, ch, ft);
  }

  int64_t size_ = 0, feat_dim_ = 0;
  double voxel_size_ = 0.0;
  std::vector<torch::Tensor> all_pts;

 private:
  void* h_ = nullptr;
};

}  // namespace

TORCH_LIBRARY(svo, m) {
  m.class_<Octree>("Octree")
      .def(torch::init<>())
      .def("init", &Octree::init)
      .def("insert", &Octree::insert)
      .def("try_insert", &Octree::try_insert)
      .def("get_voxels", &Octree::get_voxels)
      .def("get_leaf_voxels", &Octree::get_leaf_voxels)
      .def("get_features", &Octree::get_features)
      .def("count_nodes", &Octree::count_nodes)
      .def("count_leaf_nodes", &Octree::count_leaf_nodes)
      .def("has_voxel", &Octree::has_voxel)
      .def("get_centres_and_children", &Octree::get_centres_and_children)
      .def_pickle(
          [](const c10::intrusive_ptr<Octree>& self)
              -> std::tuple<int64_t, int64_t, double,
                            std::vector<torch::Tensor>> {
            return std::make_tuple(self->size_, self->feat_dim_,
                                   self->voxel_size_, self->all_pts);
          },
          [](std::tuple<int64_t, int64_t, double, std::vector<torch::Tensor>>
                 state) {
            return c10::make_intrusive<Octree>(
                std::get<0>(state), std::get<1>(state), std::get<2>(state),
                std::get<3>(state));
          });
}
