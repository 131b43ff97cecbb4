// Host-side checkpoint store: {diffusers key -> fp32 tensor}.  Replaces torch's
// load_state_dict + the reference's two load hooks (unet.py:121-146): we keep checkpoint
// tensors as-is and re-lay them out for the kernels when a model handle is built.
#pragma once
#include "sd_common.h"

namespace sd {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  size_t numel() const {
    size_t n = 1;
    for (auto s : shape) n *= (size_t)s;
    return n;
  }
};

class WeightStore {
 public:
  void add(const std::string& name, const void* data, int dtype /*0 f16, 1 f32, 2 bf16*/,
           const int64_t* shape, int ndim);
  void load_safetensors(const std::string& path, const std::string& prefix);
  const HostTensor& get(const std::string& name) const;   // throws kNotFound
  const HostTensor* find(const std::string& name) const;  // nullptr when absent (deprecated VAE attention names accepted)
  bool has(const std::string& name) const { return find(name) != nullptr; }
  size_t size() const { return map_.size(); }

 private:
  std::map<std::string, HostTensor> map_;
};

}  // namespace sd

struct sd_weights {
  sd::WeightStore store;
};
