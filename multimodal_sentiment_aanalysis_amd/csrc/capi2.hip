// extern "C" entry points for the ResNet-side row kernels (BatchNorm, pooling, stem im2col): unit-testable pieces of
// the engines (declared in include/mmsa.h).
#include "../../include/mmsa.h"
#include "ops.h"

extern "C" {

size_t mmsa_bn_ws_bytes(int32_t C) { return bn_ws_bytes(C); }
int mmsa_bn_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                float* mean, float* invstd, const void* res, void* y, float* ws, int32_t M, int32_t C, float eps,
                float momentum, int32_t act, int32_t training, void* stream) {
  if (!x || !gamma || !beta || !mean || !invstd || !y || !ws) return MMSA_ERR_ARG;
  return bn_forward(dtype, x, gamma, beta, running_mean, running_var, mean, invstd, res, y, ws, M, C, eps, momentum, act,
                    training, (hipStream_t)stream);
}
int mmsa_bn_bwd(int32_t dtype, const void* dy, const void* x, const void* y, const float* mean, const float* invstd,
                const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, int32_t accumulate,
                float* ws, int32_t M, int32_t C, int32_t act, int32_t training, void* stream) {
  if (!dy || !x || !mean || !invstd || !gamma || !beta || !dx || !ws) return MMSA_ERR_ARG;
  return bn_backward(dtype, dy, x, y, mean, invstd, gamma, beta, dx, dres, dgamma, dbeta, accumulate, ws, M, C, act,
                     training, (hipStream_t)stream);
}
int mmsa_maxpool_fwd(int32_t dtype, const void* x, void* y, uint8_t* idx, int32_t B, int32_t H, int32_t W, int32_t C,
                     void* stream) {
  if (!x || !y || !idx) return MMSA_ERR_ARG;
  return maxpool_fwd(dtype, x, y, idx, B, H, W, C, (hipStream_t)stream);
}
int mmsa_maxpool_bwd(int32_t dtype, const void* dy, const uint8_t* idx, void* dx, int32_t B, int32_t H, int32_t W, int32_t C,
                     void* stream) {
  if (!dy || !idx || !dx) return MMSA_ERR_ARG;
  return maxpool_bwd(dtype, dy, idx, dx, B, H, W, C, (hipStream_t)stream);
}
int mmsa_avgpool_fwd(int32_t dtype, const void* x, void* y, int32_t B, int32_t HW, int32_t C, void* stream) {
  if (!x || !y) return MMSA_ERR_ARG;
  return avgpool_fwd(dtype, x, y, B, HW, C, (hipStream_t)stream);
}
int mmsa_avgpool_bwd(int32_t dtype, const void* dy, void* dx, int32_t B, int32_t HW, int32_t C, void* stream) {
  if (!dy || !dx) return MMSA_ERR_ARG;
  return avgpool_bwd(dtype, dy, dx, B, HW, C, (hipStream_t)stream);
}
int mmsa_stem_im2col(int32_t dtype, const float* img, void* col, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t OH,
                     int32_t OW, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t Kpad, void* stream) {
  if (!img || !col) return MMSA_ERR_ARG;
  return stem_im2col(dtype, img, col, B, Cin, H, W, OH, OW, KH, KW, stride, pad, Kpad, (hipStream_t)stream);
}

/* N1: fused contrastive losses (contrastive.hip) */
size_t mmsa_contrastive_ws_bytes(int32_t B, int32_t D) { return (B > 0 && D > 0) ? contrastive_ws_bytes(B, D) : 0; }
int mmsa_infonce_fwd_bwd(const float* feat1, const float* feat2, const int64_t* labels, const float* temperature, float* loss,
                         float* dfeat1, float* dfeat2, float* dtemp, int32_t B, int32_t D, float grad_scale, void* ws,
                         void* stream) {
  if (!feat1 || !feat2 || !labels || !temperature || !loss || !dfeat1 || !dfeat2 || !ws || dfeat1 == dfeat2) return MMSA_ERR_ARG;
  return infonce_fwd_bwd(feat1, feat2, (const long long*)labels, temperature, loss, dfeat1, dfeat2, dtemp, B, D, grad_scale,
                         (float*)ws, (hipStream_t)stream);
}
int mmsa_supcon_fwd_bwd(const float* z1, const float* z2, const int64_t* labels, float temperature, float* loss, float* dz1,
                        float* dz2, int32_t B, int32_t D, float grad_scale, void* ws, void* stream) {
  if (!z1 || !z2 || !labels || !loss || !dz1 || !dz2 || !ws) return MMSA_ERR_ARG;
  return supcon_fwd_bwd(z1, z2, (const long long*)labels, temperature, loss, dz1, dz2, B, D, grad_scale, (float*)ws,
                        (hipStream_t)stream);
}

}  // extern "C"
