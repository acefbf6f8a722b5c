// Library introspection + error strings for the C ABI.
#include <string.h>

#include "mvp_common.h"

extern "C" int mvp_get_info(mvp_info_t* out) {
  if (!out) return MVP_EINVAL;
  memset(out, 0, sizeof(*out));
  out->abi_version = MVP_ABI_VERSION;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  out->device_count = n;
  if (n > 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) {
      strncpy(out->arch, prop.gcnArchName, sizeof(out->arch) - 1);
      out->cu_count = prop.multiProcessorCount;
      out->gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    }
  }
  return MVP_OK;
}

// sizeof() of an argument struct by its C name: lets a binding verify its mirror of the header (field order / padding
// drift between include/mvp_hip.h and a ctypes / JNI / cgo struct is otherwise silent).  -1 = unknown name.
extern "C" int mvp_sizeof(const char* name) {
  if (!name) return -1;
#define MVP_SZ(T) if (strcmp(name, #T) == 0) return (int)sizeof(T);
  MVP_SZ(mvp_info_t) MVP_SZ(mvp_split_bf16_args) MVP_SZ(mvp_patch_gather_args) MVP_SZ(mvp_gemm_args) MVP_SZ(mvp_layernorm_args)
  MVP_SZ(mvp_attention_args) MVP_SZ(mvp_cls_rows_args) MVP_SZ(mvp_bn_tokens_args) MVP_SZ(mvp_pack_nchw_args) MVP_SZ(mvp_resize_args)
  MVP_SZ(mvp_depth_predict_args) MVP_SZ(mvp_depth_loss_args) MVP_SZ(mvp_angular_loss_args) MVP_SZ(mvp_colsum_args) MVP_SZ(mvp_adamw_args)
  MVP_SZ(mvp_corr_argmax_args) MVP_SZ(mvp_conv_weight_pack_args) MVP_SZ(mvp_upsample_cl_args) MVP_SZ(mvp_upconv_boxsum_args) MVP_SZ(mvp_upconv_gather_args) MVP_SZ(mvp_gemm_tn_args)
  MVP_SZ(mvp_depth_metrics_args) MVP_SZ(mvp_snorm_metrics_args) MVP_SZ(mvp_linear_bins_args) MVP_SZ(mvp_im2col_args)
  MVP_SZ(mvp_maxpool_cl_args) MVP_SZ(mvp_mask_split_args) MVP_SZ(mvp_metrics_breakdown_args) MVP_SZ(mvp_argmax_2d_args) MVP_SZ(mvp_scale_shift_args) MVP_SZ(mvp_stem_args) MVP_SZ(mvp_bn_running_update_args)
#undef MVP_SZ
  return -1;
}

extern "C" const char* mvp_strerror(int code) {
  switch (code) {
    case MVP_OK: return "ok";
    case MVP_EINVAL: return "invalid argument (shape, alignment or NULL pointer)";
    case MVP_ELAUNCH: return "HIP kernel launch failed";
    case MVP_ENODEV: return "no gfx950 device";
    default: return "unknown error";
  }
}
