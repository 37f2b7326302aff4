// oracle/ref_mish_binding.cpp -- CPU ORACLE support (test infrastructure, not the product).
//
// Own pybind entry points over the REFERENCE's CPU Mish kernels, compiled from where they
// lie (/root/reference/mmdet/ops/mish_cuda/src/kernel/mish_cpu.cc, namespace
// mish_cpu_kernel).  The reference's own dispatcher (src/mish.cc) cannot be linked without
// its CUDA object, so this file plays its CPU branch: allocate empty_like and
// call the kernel (mish.cc:14-33).
#include <torch/extension.h>

namespace mish_cpu_kernel {
void mish(at::Tensor inp, at::Tensor out);
void mish_backward(at::Tensor grad_out, at::Tensor inp, at::Tensor grad_inp);
}  // namespace mish_cpu_kernel

torch::Tensor mish_forward(const torch::Tensor& input) {
  auto output = torch::empty_like(input);
  mish_cpu_kernel::mish(input, output);
  return output;
}

torch::Tensor mish_backward(const torch::Tensor& grad_out, const torch::Tensor& input) {
  auto grad_inp = torch::empty_like(input);
  mish_cpu_kernel::mish_backward(grad_out, input, grad_inp);
  return grad_inp;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("mish_forward", &mish_forward);
  m.def("mish_backward", &mish_backward);
}
