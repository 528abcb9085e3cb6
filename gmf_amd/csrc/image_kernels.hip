// Image encoder (ResNet-34 truncated after layer2, GMF_PointDSC/models/resnet.py:195-216).
// The convolutions stay with MIOpen (PyTorch-ROCm, NHWC fp32); what the framework leaves un-fused around them - the folded
// BatchNorm bias, the residual add and the ReLU, three element-wise passes per convolution - is one pass here:
//   y = max(y + bias[c] (+ residual), 0)   in place on an NHWC tensor (resnet.py:59-75, BasicBlock.forward).
#include <hip/hip_runtime.h>

#include "launchers.hpp"

namespace gmf {

// total4 = n_pixels * C / 4 float4s; C % 4 == 0, so a float4 never straddles two pixels
__global__ void __launch_bounds__(256)
k_bias_relu_nhwc(float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ residual, long total4,
                 int c4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const float4 b = reinterpret_cast<const float4*>(bias)[idx % c4];
  float4 v = reinterpret_cast<float4*>(y)[idx];
  v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  if (residual) {
    const float4 r = reinterpret_cast<const float4*>(residual)[idx];
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  reinterpret_cast<float4*>(y)[idx] = v;
}

hipError_t launch_bias_relu_nhwc(float* y, const float* bias, const float* residual, long n_pixels, int C, hipStream_t s) {
  const long total4 = n_pixels * (C / 4);
  hipLaunchKernelGGL(k_bias_relu_nhwc, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, y, bias, residual, total4, C / 4);
  return hipGetLastError();
}

}  // namespace gmf
