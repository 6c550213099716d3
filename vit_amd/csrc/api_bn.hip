// extern "C" surface, part 2: training-mode BatchNorm and max-pool backward (bn.hip).
#include "../../include/rvit.h"
#include "common.h"
#include "rvit_internal.h"

extern "C" {

int rvit_bn_stats(const void* y, int64_t M, int C, float* sums, void* workspace, void* stream) {
  return rvit_bn_stats_impl(y, M, C, sums, (float*)workspace, (hipStream_t)stream);
}

int rvit_bn_apply(const void* y, const float* sums, const float* gamma, const float* beta, const void* residual,
                  void* z, int64_t M, int C, float eps, int relu, float momentum, float* running_mean,
                  float* running_var, void* relu_mask, void* stream) {
  return rvit_bn_apply_impl(y, sums, gamma, beta, residual, z, M, C, eps, relu, momentum, running_mean, running_var,
                            relu_mask, (hipStream_t)stream);
}

int rvit_bn_apply_pool(const void* y, const float* sums, const float* gamma, const float* beta, void* pooled,
                       void* argmax, void* relu_mask, int B, int H, int W, int C, float eps, float momentum,
                       float* running_mean, float* running_var, void* stream) {
  return rvit_bn_apply_pool_impl(y, sums, gamma, beta, pooled, argmax, relu_mask, B, H, W, C, eps, momentum, running_mean,
                                 running_var, (hipStream_t)stream);
}

int rvit_bn_bwd_reduce(const void* dz, const void* z, const void* y, const float* fwd_sums, int64_t M, int C,
                       float eps, int relu, float* bwd_sums, void* workspace, void* stream) {
  return rvit_bn_bwd_reduce_impl(dz, z, y, fwd_sums, M, C, eps, relu, bwd_sums, (float*)workspace, (hipStream_t)stream);
}

int rvit_bn_bwd_apply(const void* dz, const void* z, const void* y, const float* fwd_sums, const float* bwd_sums,
                      const float* gamma, void* dy, void* dz_masked, float* dgamma, float* dbeta, int64_t M, int C,
                      float eps, int relu, void* stream) {
  return rvit_bn_bwd_apply_impl(dz, z, y, fwd_sums, bwd_sums, gamma, dy, dz_masked, dgamma, dbeta, M, C, eps, relu,
                                (hipStream_t)stream);
}

int rvit_bn_bwd_reduce_pool(const void* dpool, const void* argmax, const void* relu_mask, const void* y,
                            const float* fwd_sums, int B, int H, int W, int C, float eps, float* bwd_sums, void* workspace,
                            void* stream) {
  return rvit_bn_bwd_reduce_pool_impl(dpool, argmax, relu_mask, y, fwd_sums, B, H, W, C, eps, bwd_sums, (float*)workspace,
                                      (hipStream_t)stream);
}

int rvit_bn_bwd_apply_pool(const void* dpool, const void* argmax, const void* relu_mask, const void* y,
                           const float* fwd_sums, const float* bwd_sums, const float* gamma, void* dy, float* dgamma,
                           float* dbeta, int B, int H, int W, int C, float eps, void* stream) {
  return rvit_bn_bwd_apply_pool_impl(dpool, argmax, relu_mask, y, fwd_sums, bwd_sums, gamma, dy, dgamma, dbeta, B, H, W, C,
                                     eps, (hipStream_t)stream);
}

int rvit_maxpool3x3s2_bwd_nhwc(const void* dpool, const void* argmax, void* dx, int B, int H, int W, int C,
                               void* stream) {
  return rvit_maxpool_bwd_impl(dpool, argmax, dx, B, H, W, C, (hipStream_t)stream);
}

}  // extern "C"
