// ctx.hpp — the context behind the opaque tfhe_hip_ctx of include/tfhe_hip.h,
// shared by the kernel translation unit (tfhe_hip.hip: kernels, device-pointer
// entry points) and the host layer (host_api.cpp: staging of host buffers,
// sample handles, the key file format, multi-GPU contexts).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/tfhe_hip.h"

namespace wfft {
struct cplx;
}

struct tfhe_hip_ctx {
  int device = 0;
  int num_cus = 256;
  int br_mode = TFHE_HIP_BR_AUTO;
  hipStream_t stream = nullptr;
  bool owns_stream = false;        // created by the library (shards of a multi-GPU context)
  std::string err;
  bool have_keys = false;
  tfhe_hip_params P{};
  int row_stride = 0;
  wfft::cplx* tables = nullptr;  // device: tE[1024], twist[1024]
  wfft::cplx* tables64 = nullptr;  // device: the tables of wave_fft64.hpp
  wfft::cplx* bkfft = nullptr;   // device blob 1
  char* ksblob = nullptr;        // device blob 2: rows then variances
  size_t bk_bytes = 0, ks_bytes = 0, ks_rows_bytes = 0;
  // scratch, grown on demand
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // this device's copy of the library's fixed scratch array (entry points that must not allocate)
  void* fixed_scratch = nullptr;
  // workspace of the circuit executor, grown on demand
  void* work = nullptr;
  size_t work_bytes = 0;
  // staging of the host-pointer entry points, grown on demand and kept: a device
  // buffer and a pinned host buffer of the same layout
  void* stage_dev = nullptr;
  size_t stage_dev_bytes = 0;
  void* stage_pin = nullptr;
  size_t stage_pin_bytes = 0;
  // optional per-kernel timing (tfhe_hip_profile_*)
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[3];   // prologue, blind rotate, key switch
  // multi-GPU context (tfhe_hip_create_multi): one single-device context per
  // entry of the device list; empty for a single-device context
  std::vector<tfhe_hip_ctx*> shards;
  // how a multi-GPU context's keys reached its other devices (tfhe_hip_key_transport)
  const char* key_transport = "none";
  // sample handles (tfhe_hip_samples_alloc) that still own device memory of this context;
  // tfhe_hip_destroy releases their buffers and detaches them
  std::vector<tfhe_hip_samples*> live_handles;
};

namespace tfhe_detail {

inline int fail(tfhe_hip_ctx* ctx, int code, const std::string& msg) {
  if (ctx) {
    try {
      ctx->err = msg;
    } catch (...) {
    }
  }
  return code;
}

int check_params(tfhe_hip_ctx* ctx, const tfhe_hip_params* p);
// keys loaded and the context's device made current
int need_keys(tfhe_hip_ctx* ctx);
// tfhe_hip_load_keys on a multi-GPU context (host_api.cpp)
int load_keys_multi(tfhe_hip_ctx* ctx, const tfhe_hip_params* params, const int32_t* bk,
                    const int32_t* ks_a, const int32_t* ks_b, const double* ks_var);

// keys present on shards[0]: gives every other shard a copy (host_api.cpp)
int replicate_keys_multi(tfhe_hip_ctx* ctx, const tfhe_hip_params* params);
// (re)allocates the two device key blobs of a single-device context for `params`, contents
// undefined, have_keys false until the caller has filled them (tfhe_hip.hip)
int alloc_key_blobs(tfhe_hip_ctx* ctx, const tfhe_hip_params* params);
// frees the device memory of the context's live sample handles and detaches them (host_api.cpp)
void release_handles(tfhe_hip_ctx* ctx);

}  // namespace tfhe_detail

#define HIP_TRY(ctx, expr)                                                           \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess)                                                            \
      return tfhe_detail::fail(ctx, TFHE_HIP_ERR_RUNTIME,                            \
                               std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)
