// torch.classes.pyg.CUDAHashMap on the HIP device (pyg_lib/csrc/classes/cuda/hash_map.cu:103-203: same
// class name, constructor, methods and pickle contract; kernels csrc/hip/hash_map.hip through the
// C-ABI).  "CUDA" is how PyTorch-ROCm names the HIP device.
#include <torch/custom_class.h>
#include <torch/library.h>

#include "binding_common.h"

namespace pyg_amd {
namespace {

struct CUDAHashMap : torch::CustomClassHolder {
  CUDAHashMap(const Tensor& key, double load_factor = 0.5) {
    at::TensorArg key_arg{key, "key", 0};
    at::CheckedFrom c{"CUDAHashMap.init"};
    at::checkDeviceType(c, key, at::DeviceType::CUDA);
    at::checkDim(c, key_arg, 1);
    at::checkContiguous(c, key_arg);
    TORCH_CHECK(key.scalar_type() == at::kShort || key.scalar_type() == at::kInt || key.scalar_type() == at::kLong,
                "\"cuda_hash_map_init\" not implemented for '", key.scalar_type(), "'");
    DeviceGuard guard(key.device());
    key_ = key.clone();
    slots_ = pyg_hip_hash_map_slots(key.numel(), load_factor);
    const auto opts = key.options().dtype(at::kLong);
    table_keys_ = at::empty({slots_}, opts);
    table_vals_ = at::empty({slots_}, opts);
    auto distinct = at::empty({1}, opts);
    check_status(pyg_hip_hash_map_build(dtype_code(key.scalar_type()), key_.data_ptr(), key_.numel(),
                                        reinterpret_cast<uint64_t*>(table_keys_.data_ptr<int64_t>()),
                                        table_vals_.data_ptr<int64_t>(), slots_, distinct.data_ptr<int64_t>(),
                                        current_stream(key_)));
    size_ = distinct.item<int64_t>();
  }

  Tensor get(const Tensor& query) {
    at::TensorArg query_arg{query, "query", 0};
    at::CheckedFrom c{"CUDAHashMap.get"};
    at::checkDeviceType(c, query, at::DeviceType::CUDA);
    at::checkDim(c, query_arg, 1);
    at::checkContiguous(c, query_arg);
    TORCH_CHECK(query.scalar_type() == key_.scalar_type(), "CUDAHashMap.get: query must have the dtype of the keys");
    DeviceGuard guard(query.device());
    auto out = at::empty({query.numel()}, query.options().dtype(at::kLong));
    check_status(pyg_hip_hash_map_get(dtype_code(query.scalar_type()), query.data_ptr(), query.numel(),
                                      reinterpret_cast<const uint64_t*>(table_keys_.data_ptr<int64_t>()),
                                      table_vals_.data_ptr<int64_t>(), slots_, out.data_ptr<int64_t>(),
                                      current_stream(query)));
    return out;
  }

  // distinct keys in insertion order (== the key tensor when it holds no duplicates)
  Tensor keys() {
    if (size_ == key_.numel()) return key_;
    DeviceGuard guard(key_.device());
    auto first = get(key_) == at::arange(key_.numel(), key_.options().dtype(at::kLong));
    return key_.masked_select(first);
  }
  int64_t size() { return size_; }
  at::ScalarType dtype() { return key_.scalar_type(); }
  at::Device device() { return key_.device(); }

 private:
  Tensor key_, table_keys_, table_vals_;
  int64_t slots_ = 0, size_ = 0;
};

}  // namespace

TORCH_LIBRARY_FRAGMENT(pyg, m) {
  m.class_<CUDAHashMap>("CUDAHashMap")
      .def(torch::init<at::Tensor&, double>())
      .def("get", &CUDAHashMap::get)
      .def("keys", &CUDAHashMap::keys)
      .def("size", &CUDAHashMap::size)
      .def("dtype", &CUDAHashMap::dtype)
      .def("device", &CUDAHashMap::device)
      .def_pickle(
          [](const c10::intrusive_ptr<CUDAHashMap>& self) -> at::Tensor { return self->keys(); },
          [](const at::Tensor& state) -> c10::intrusive_ptr<CUDAHashMap> {
            return c10::make_intrusive<CUDAHashMap>(state);
          });
}

}  // namespace pyg_amd
