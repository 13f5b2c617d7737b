// host_api.cpp — the host layer of libtfhe_hip.so above the device-pointer
// entry points of tfhe_hip.hip:
//   * the host-pointer entry points (copy in, run, copy out) through a staging
//     pool that the context keeps: one device buffer and one pinned host buffer,
//     grown on demand and never freed before tfhe_hip_destroy;
//   * device-resident sample handles (tfhe_hip_samples_*), so that chained gates
//     never leave the GPU;
//   * the key file: a flat little-endian image of the key material a CloudKey
//     carries (src/lwe.rs:247-251, 432-446, 495-514);
//   * multi-GPU contexts (tfhe_hip_create_multi): one single-device context per
//     listed GPU, keys replicated once, every batch split contiguously.
// No kernels here; compiled as plain host C++.

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>

#include "ctx.hpp"

using tfhe_detail::fail;

namespace {

constexpr size_t kN = 1024;

// ---------------------------------------------------------------------------
// Staging pool
// ---------------------------------------------------------------------------
// Batches whose buffers add up to at most this many bytes go through the pinned
// host buffer (one asynchronous copy each way); larger ones are copied straight
// from and to the caller's memory, which the runtime stages itself.
constexpr size_t kPinnedLimit = 16u << 20;

int ensure_stage(tfhe_hip_ctx* ctx, size_t bytes, bool pinned) {
  if (bytes > ctx->stage_dev_bytes) {
    if (ctx->stage_dev) {
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      HIP_TRY(ctx, hipFree(ctx->stage_dev));
    }
    ctx->stage_dev = nullptr;
    ctx->stage_dev_bytes = 0;
    const size_t want = std::max(bytes, (size_t)1 << 16);
    HIP_TRY(ctx, hipMalloc(&ctx->stage_dev, want));
    ctx->stage_dev_bytes = want;
  }
  if (pinned && bytes > ctx->stage_pin_bytes) {
    if (ctx->stage_pin) HIP_TRY(ctx, hipHostFree(ctx->stage_pin));
    ctx->stage_pin = nullptr;
    ctx->stage_pin_bytes = 0;
    const size_t want = std::max(bytes, (size_t)1 << 16);
    HIP_TRY(ctx, hipHostMalloc(&ctx->stage_pin, want, hipHostMallocDefault));
    ctx->stage_pin_bytes = want;
  }
  return 0;
}

// The host buffers of one call.  Buffers are laid out input-only first, then
// in/out, then output-only, so that each direction is one contiguous span.
class Staging {
 public:
  explicit Staging(tfhe_hip_ctx* c) : ctx_(c) {}

  // in: copied to the device by upload(); out: filled by download().  Either may
  // be NULL; a NULL/NULL buffer gets no device memory and dev() returns NULL.
  int add(const void* in, void* out, size_t bytes) {
    if (count_ == kMax) return -1;
    items_[count_] = {in, out, bytes, 0};
    return count_++;
  }

  int upload() {
    size_t off = 0;
    auto place = [&](int cls) {
      for (int i = 0; i < count_; i++) {
        Item& it = items_[i];
        const int c = !it.in && !it.out ? -1 : (it.in && !it.out ? 0 : (it.in ? 1 : 2));
        if (c != cls) continue;
        it.off = off;
        off += (it.bytes + 255) / 256 * 256;
      }
    };
    place(0);
    out_begin_ = off;
    place(1);
    in_end_ = off;
    place(2);
    total_ = off;
    pinned_ = total_ <= kPinnedLimit;
    if (total_ == 0) return 0;
    if (int rc = ensure_stage(ctx_, total_, pinned_)) return rc;
    char* dev = (char*)ctx_->stage_dev;
    if (pinned_) {
      char* pin = (char*)ctx_->stage_pin;
      for (int i = 0; i < count_; i++)
        if (items_[i].in) memcpy(pin + items_[i].off, items_[i].in, items_[i].bytes);
      if (in_end_)
        HIP_TRY(ctx_, hipMemcpyAsync(dev, pin, in_end_, hipMemcpyHostToDevice, ctx_->stream));
    } else {
      for (int i = 0; i < count_; i++)
        if (items_[i].in)
          HIP_TRY(ctx_, hipMemcpyAsync(dev + items_[i].off, items_[i].in, items_[i].bytes,
                                       hipMemcpyHostToDevice, ctx_->stream));
    }
    return 0;
  }

  template <class T>
  T* dev(int slot) const {
    const Item& it = items_[slot];
    if (!it.in && !it.out) return nullptr;
    return (T*)((char*)ctx_->stage_dev + it.off);
  }

  // Copies the outputs back and waits for the stream.
  int download() {
    if (total_ == 0) return 0;
    char* dev = (char*)ctx_->stage_dev;
    if (pinned_) {
      char* pin = (char*)ctx_->stage_pin;
      if (total_ > out_begin_)
        HIP_TRY(ctx_, hipMemcpyAsync(pin + out_begin_, dev + out_begin_, total_ - out_begin_,
                                     hipMemcpyDeviceToHost, ctx_->stream));
      HIP_TRY(ctx_, hipStreamSynchronize(ctx_->stream));
      for (int i = 0; i < count_; i++)
        if (items_[i].out) memcpy(items_[i].out, pin + items_[i].off, items_[i].bytes);
    } else {
      for (int i = 0; i < count_; i++)
        if (items_[i].out)
          HIP_TRY(ctx_, hipMemcpyAsync(items_[i].out, dev + items_[i].off, items_[i].bytes,
                                       hipMemcpyDeviceToHost, ctx_->stream));
      HIP_TRY(ctx_, hipStreamSynchronize(ctx_->stream));
    }
    return 0;
  }

 private:
  struct Item {
    const void* in;
    void* out;
    size_t bytes, off;
  };
  static constexpr int kMax = 16;
  tfhe_hip_ctx* ctx_;
  Item items_[kMax];
  int count_ = 0;
  size_t out_begin_ = 0, in_end_ = 0, total_ = 0;
  bool pinned_ = false;
};

int ready(tfhe_hip_ctx* ctx) {          // single-device context with keys, device current
  return tfhe_detail::need_keys(ctx);
}

// ---------------------------------------------------------------------------
// Multi-GPU contexts: contiguous shards, one host thread per shard
// ---------------------------------------------------------------------------
struct Span {
  size_t lo, hi;
};

Span shard_span(size_t total, size_t rank, size_t world) {
  const size_t base = total / world, rem = total % world;
  const size_t lo = rank * base + std::min(rank, rem);
  return {lo, lo + base + (rank < rem ? 1 : 0)};
}

// Runs fn(shard context, lo, hi) for every non-empty shard of [0, total), all at
// once; returns the first failure with the failing shard's message.
template <class F>
int run_sharded(tfhe_hip_ctx* ctx, size_t total, F&& fn) {
  const size_t world = ctx->shards.size();
  std::vector<int> rc(world, 0);
  std::vector<std::thread> threads;
  bool started = true;
  try {
    threads.reserve(world);
    for (size_t r = 0; r < world; r++) {
      const Span sp = shard_span(total, r, world);
      if (sp.lo == sp.hi) continue;
      threads.emplace_back([&, r, sp] { rc[r] = fn(ctx->shards[r], sp.lo, sp.hi); });
    }
  } catch (...) {
    started = false;
  }
  for (auto& t : threads)
    if (t.joinable()) t.join();
  if (!started) return fail(ctx, TFHE_HIP_ERR_RUNTIME, "could not start the shard threads");
  for (size_t r = 0; r < world; r++)
    if (rc[r]) return fail(ctx, rc[r], "device " + std::to_string(ctx->shards[r]->device) + ": " + ctx->shards[r]->err);
  return 0;
}

bool multi(const tfhe_hip_ctx* ctx) { return !ctx->shards.empty(); }

template <class T>
const T* at(const T* p, size_t off) { return p ? p + off : nullptr; }
template <class T>
T* at(T* p, size_t off) { return p ? p + off : nullptr; }

// ---------------------------------------------------------------------------
// single-device host-pointer calls
// ---------------------------------------------------------------------------
int bootstrap_1(tfhe_hip_ctx* ctx, int32_t mu, const int32_t* a, const int32_t* b, int32_t* out_a,
                int32_t* out_b, double* var_out, size_t B, bool keyswitch) {
  if (int rc = ready(ctx)) return rc;
  const size_t n = ctx->P.n, on = keyswitch ? n : kN;
  Staging st(ctx);
  const int ia = st.add(a, nullptr, B * n * 4), ib = st.add(b, nullptr, B * 4);
  const int oa = st.add(nullptr, out_a, B * on * 4), ob = st.add(nullptr, out_b, B * 4);
  const int ov = st.add(nullptr, var_out, B * 8);
  if (int rc = st.upload()) return rc;
  int rc;
  if (keyswitch)
    rc = tfhe_hip_bootstrap_dev(ctx, mu, st.dev<int32_t>(ia), st.dev<int32_t>(ib), st.dev<int32_t>(oa),
                                st.dev<int32_t>(ob), st.dev<double>(ov), B);
  else
    rc = tfhe_hip_bootstrap_noks_dev(ctx, mu, st.dev<int32_t>(ia), st.dev<int32_t>(ib),
                                     st.dev<int32_t>(oa), st.dev<int32_t>(ob), B);
  if (rc) return rc;
  return st.download();
}

int keyswitch_1(tfhe_hip_ctx* ctx, const int32_t* a, const int32_t* b, int32_t* out_a, int32_t* out_b,
                double* var_out, size_t B) {
  if (int rc = ready(ctx)) return rc;
  const size_t n = ctx->P.n;
  Staging st(ctx);
  const int ia = st.add(a, nullptr, B * kN * 4), ib = st.add(b, nullptr, B * 4);
  const int oa = st.add(nullptr, out_a, B * n * 4), ob = st.add(nullptr, out_b, B * 4);
  const int ov = st.add(nullptr, var_out, B * 8);
  if (int rc = st.upload()) return rc;
  if (int rc = tfhe_hip_keyswitch_dev(ctx, st.dev<int32_t>(ia), st.dev<int32_t>(ib), st.dev<int32_t>(oa),
                                      st.dev<int32_t>(ob), st.dev<double>(ov), B)) return rc;
  return st.download();
}

int gate_1(tfhe_hip_ctx* ctx, int gate, const int32_t* ca_a, const int32_t* ca_b, const int32_t* cb_a,
           const int32_t* cb_b, const int32_t* cc_a, const int32_t* cc_b, int32_t* out_a,
           int32_t* out_b, double* var_out, size_t B) {
  if (int rc = ready(ctx)) return rc;
  const size_t n = ctx->P.n;
  Staging st(ctx);
  const int aa = st.add(ca_a, nullptr, B * n * 4), ab = st.add(ca_b, nullptr, B * 4);
  const int ba = st.add(cb_a, nullptr, B * n * 4), bb = st.add(cb_b, nullptr, B * 4);
  const int ca = st.add(cc_a, nullptr, B * n * 4), cb = st.add(cc_b, nullptr, B * 4);
  const int oa = st.add(nullptr, out_a, B * n * 4), ob = st.add(nullptr, out_b, B * 4);
  const int ov = st.add(nullptr, var_out, B * 8);
  if (int rc = st.upload()) return rc;
  if (int rc = tfhe_hip_gate_dev(ctx, gate, st.dev<int32_t>(aa), st.dev<int32_t>(ab), st.dev<int32_t>(ba),
                                 st.dev<int32_t>(bb), st.dev<int32_t>(ca), st.dev<int32_t>(cb),
                                 st.dev<int32_t>(oa), st.dev<int32_t>(ob), st.dev<double>(ov), B)) return rc;
  return st.download();
}

// Operands and results of the caller are bit-sliced [bits][stride][...]; this
// call takes instances [0, count) of every slice.
int circuit_add_1(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b, const int32_t* b_a,
                  const int32_t* b_b, int32_t* sum_a, int32_t* sum_b, int32_t* carry_a,
                  int32_t* carry_b, double* sum_var, double* carry_var, size_t count, int bits,
                  size_t stride) {
  if (int rc = ready(ctx)) return rc;
  const size_t n = ctx->P.n, nb = (size_t)bits;
  const size_t ma = nb * count * n * 4, mb = nb * count * 4;
  // device-side slices are dense [bits][count]; the copies below gather and scatter
  // the caller's [bits][stride] slices.
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_aa = take(ma), o_ab = take(mb), o_ba = take(ma), o_bb = take(mb);
  const size_t o_sa = take(ma), o_sb = take(mb), o_sv = take(2 * mb);
  const size_t o_ca = take(count * n * 4), o_cb = take(count * 4), o_cv = take(count * 8);
  if (int rc = ensure_stage(ctx, off, false)) return rc;
  char* dev = (char*)ctx->stage_dev;
  auto copy2d = [&](void* dst, size_t dpitch, const void* src, size_t spitch, size_t width,
                    hipMemcpyKind kind) -> hipError_t {
    if (nb == 0 || width == 0) return hipSuccess;
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, nb, kind, ctx->stream);
  };
  HIP_TRY(ctx, copy2d(dev + o_aa, count * n * 4, a_a, stride * n * 4, count * n * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, copy2d(dev + o_ab, count * 4, a_b, stride * 4, count * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, copy2d(dev + o_ba, count * n * 4, b_a, stride * n * 4, count * n * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, copy2d(dev + o_bb, count * 4, b_b, stride * 4, count * 4, hipMemcpyHostToDevice));
  if (int rc = tfhe_hip_circuit_add_dev(
          ctx, (int32_t*)(dev + o_aa), (int32_t*)(dev + o_ab), (int32_t*)(dev + o_ba),
          (int32_t*)(dev + o_bb), (int32_t*)(dev + o_sa), (int32_t*)(dev + o_sb), (int32_t*)(dev + o_ca),
          (int32_t*)(dev + o_cb), sum_var ? (double*)(dev + o_sv) : nullptr,
          carry_var ? (double*)(dev + o_cv) : nullptr, count, bits)) return rc;
  HIP_TRY(ctx, copy2d(sum_a, stride * n * 4, dev + o_sa, count * n * 4, count * n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, copy2d(sum_b, stride * 4, dev + o_sb, count * 4, count * 4, hipMemcpyDeviceToHost));
  if (sum_var)
    HIP_TRY(ctx, copy2d(sum_var, stride * 8, dev + o_sv, count * 8, count * 8, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpyAsync(carry_a, dev + o_ca, count * n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(carry_b, dev + o_cb, count * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (carry_var)
    HIP_TRY(ctx, hipMemcpyAsync(carry_var, dev + o_cv, count * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

// circuits::compare (which = 0) / circuits::eq (which = 1): bit-sliced operands
// [bits][stride][...] of the caller, instances [0, count) of every slice; one sample per
// instance comes back.
int circuit_reduce_1(tfhe_hip_ctx* ctx, int which, const int32_t* a_a, const int32_t* a_b,
                     const int32_t* b_a, const int32_t* b_b, int32_t* out_a, int32_t* out_b,
                     double* out_var, size_t count, int bits, size_t stride) {
  if (int rc = ready(ctx)) return rc;
  const size_t n = ctx->P.n, nb = (size_t)(bits > 0 ? bits : 0);
  const size_t ma = nb * count * n * 4, mb = nb * count * 4;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_aa = take(ma), o_ab = take(mb), o_ba = take(ma), o_bb = take(mb);
  const size_t o_oa = take(count * n * 4), o_ob = take(count * 4), o_ov = take(count * 8);
  if (int rc = ensure_stage(ctx, off, false)) return rc;
  char* dev = (char*)ctx->stage_dev;
  auto copy2d = [&](void* dst, size_t dpitch, const void* src, size_t spitch, size_t width) -> hipError_t {
    if (nb == 0 || width == 0) return hipSuccess;
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, nb, hipMemcpyHostToDevice, ctx->stream);
  };
  HIP_TRY(ctx, copy2d(dev + o_aa, count * n * 4, a_a, stride * n * 4, count * n * 4));
  HIP_TRY(ctx, copy2d(dev + o_ab, count * 4, a_b, stride * 4, count * 4));
  HIP_TRY(ctx, copy2d(dev + o_ba, count * n * 4, b_a, stride * n * 4, count * n * 4));
  HIP_TRY(ctx, copy2d(dev + o_bb, count * 4, b_b, stride * 4, count * 4));
  auto fn = which == 0 ? tfhe_hip_circuit_compare_dev : tfhe_hip_circuit_eq_dev;
  if (int rc = fn(ctx, (int32_t*)(dev + o_aa), (int32_t*)(dev + o_ab), (int32_t*)(dev + o_ba),
                  (int32_t*)(dev + o_bb), (int32_t*)(dev + o_oa), (int32_t*)(dev + o_ob),
                  out_var ? (double*)(dev + o_ov) : nullptr, count, bits)) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(out_a, dev + o_oa, count * n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out_b, dev + o_ob, count * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (out_var)
    HIP_TRY(ctx, hipMemcpyAsync(out_var, dev + o_ov, count * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int circuit_reduce(tfhe_hip_ctx* ctx, int which, const int32_t* a_a, const int32_t* a_b,
                   const int32_t* b_a, const int32_t* b_b, int32_t* out_a, int32_t* out_b,
                   double* out_var, size_t count, int bits) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->have_keys) return fail(ctx, TFHE_HIP_ERR_STATE, "no keys loaded");
  if (which == 0 ? bits < 0 : bits < 1)
    return fail(ctx, TFHE_HIP_ERR_ARG, which == 0 ? "bits must be >= 0" : "eq needs at least one bit");
  if (count == 0) return 0;
  if (!out_a || !out_b || (bits > 0 && (!a_a || !a_b || !b_a || !b_b)))
    return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  if (!multi(ctx))
    return circuit_reduce_1(ctx, which, a_a, a_b, b_a, b_b, out_a, out_b, out_var, count, bits, count);
  const size_t n = ctx->P.n;      // shard by instance
  return run_sharded(ctx, count, [=](tfhe_hip_ctx* s, size_t lo, size_t hi) {
    return circuit_reduce_1(s, which, at(a_a, lo * n), at(a_b, lo), at(b_a, lo * n), at(b_b, lo),
                            out_a + lo * n, out_b + lo, at(out_var, lo), hi - lo, bits, count);
  });
}

int polymul_1(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s, int32_t* out, double* margin,
              int accumulate, size_t B) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Staging st(ctx);
  const int id = st.add(d, nullptr, B * kN * 4), is = st.add(s, nullptr, B * kN * 4);
  int rc;
  if (margin) {
    const int om = st.add(nullptr, margin, B * 2 * 8);
    if ((rc = st.upload())) return rc;
    rc = tfhe_hip_polymul_margin_dev(ctx, st.dev<int32_t>(id), st.dev<int32_t>(is), st.dev<double>(om), B);
  } else {
    const size_t out_bytes = B * (accumulate ? kN : 2 * kN - 1) * 4;
    const int io = st.add(accumulate ? out : nullptr, out, out_bytes);
    if ((rc = st.upload())) return rc;
    rc = tfhe_hip_polymul_dev(ctx, st.dev<int32_t>(id), st.dev<int32_t>(is), st.dev<int32_t>(io), accumulate, B);
  }
  if (rc) return rc;
  return st.download();
}

}  // namespace

// ===========================================================================
// C ABI: host-pointer entry points
// ===========================================================================
extern "C" {

int tfhe_hip_bootstrap(tfhe_hip_ctx* ctx, int32_t mu, const int32_t* a, const int32_t* b,
                       int32_t* out_a, int32_t* out_b, double* var_out, size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->have_keys) return fail(ctx, TFHE_HIP_ERR_STATE, "no keys loaded");
  if (B == 0) return 0;
  if (!a || !b || !out_a || !out_b) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  if (!multi(ctx)) return bootstrap_1(ctx, mu, a, b, out_a, out_b, var_out, B, true);
  const size_t n = ctx->P.n;
  return run_sharded(ctx, B, [=](tfhe_hip_ctx* s, size_t lo, size_t hi) {
    return bootstrap_1(s, mu, a + lo * n, b + lo, out_a + lo * n, out_b + lo, at(var_out, lo), hi - lo, true);
  });
}

int tfhe_hip_bootstrap_noks(tfhe_hip_ctx* ctx, int32_t mu, const int32_t* a, const int32_t* b,
                            int32_t* out_a, int32_t* out_b, size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->have_keys) return fail(ctx, TFHE_HIP_ERR_STATE, "no keys loaded");
  if (B == 0) return 0;
  if (!a || !b || !out_a || !out_b) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  if (!multi(ctx)) return bootstrap_1(ctx, mu, a, b, out_a, out_b, nullptr, B, false);
  const size_t n = ctx->P.n;
  return run_sharded(ctx, B, [=](tfhe_hip_ctx* s, size_t lo, size_t hi) {
    return bootstrap_1(s, mu, a + lo * n, b + lo, out_a + lo * kN, out_b + lo, nullptr, hi - lo, false);
  });
}

int tfhe_hip_keyswitch(tfhe_hip_ctx* ctx, const int32_t* a, const int32_t* b, int32_t* out_a,
                       int32_t* out_b, double* var_out, size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->have_keys) return fail(ctx, TFHE_HIP_ERR_STATE, "no keys loaded");
  if (B == 0) return 0;
  if (!a || !b || !out_a || !out_b) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  if (!multi(ctx)) return keyswitch_1(ctx, a, b, out_a, out_b, var_out, B);
  const size_t n = ctx->P.n;
  return run_sharded(ctx, B, [=](tfhe_hip_ctx* s, size_t lo, size_t hi) {
    return keyswitch_1(s, a + lo * kN, b + lo, out_a + lo * n, out_b + lo, at(var_out, lo), hi - lo);
  });
}

int tfhe_hip_gate(tfhe_hip_ctx* ctx, int gate, const int32_t* ca_a, const int32_t* ca_b,
                  const int32_t* cb_a, const int32_t* cb_b, const int32_t* cc_a, const int32_t* cc_b,
                  int32_t* out_a, int32_t* out_b, double* var_out, size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->have_keys) return fail(ctx, TFHE_HIP_ERR_STATE, "no keys loaded");
  if (B == 0) return 0;
  if (!ca_a || !ca_b || !cb_a || !cb_b || !out_a || !out_b)
    return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  if (gate < TFHE_HIP_NAND || gate > TFHE_HIP_MUX) return fail(ctx, TFHE_HIP_ERR_ARG, "unknown gate id");
  if (gate == TFHE_HIP_MUX && (!cc_a || !cc_b)) return fail(ctx, TFHE_HIP_ERR_ARG, "mux needs a third operand");
  if (!multi(ctx)) return gate_1(ctx, gate, ca_a, ca_b, cb_a, cb_b, cc_a, cc_b, out_a, out_b, var_out, B);
  const size_t n = ctx->P.n;
  return run_sharded(ctx, B, [=](tfhe_hip_ctx* s, size_t lo, size_t hi) {
    return gate_1(s, gate, ca_a + lo * n, ca_b + lo, cb_a + lo * n, cb_b + lo, at(cc_a, lo * n),
                  at(cc_b, lo), out_a + lo * n, out_b + lo, at(var_out, lo), hi - lo);
  });
}

int tfhe_hip_circuit_add(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b, const int32_t* b_a,
                         const int32_t* b_b, int32_t* sum_a, int32_t* sum_b, int32_t* carry_a,
                         int32_t* carry_b, double* sum_var, double* carry_var, size_t count, int bits) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->have_keys) return fail(ctx, TFHE_HIP_ERR_STATE, "no keys loaded");
  if (bits < 0) return fail(ctx, TFHE_HIP_ERR_ARG, "bits must be >= 0");
  if (count == 0) return 0;
  if (!carry_a || !carry_b || (bits > 0 && (!a_a || !a_b || !b_a || !b_b || !sum_a || !sum_b)))
    return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  if (!multi(ctx))
    return circuit_add_1(ctx, a_a, a_b, b_a, b_b, sum_a, sum_b, carry_a, carry_b, sum_var, carry_var,
                         count, bits, count);
  // shard by instance: every dependency level of an instance stays on one GPU
  const size_t n = ctx->P.n;
  return run_sharded(ctx, count, [=](tfhe_hip_ctx* s, size_t lo, size_t hi) {
    return circuit_add_1(s, at(a_a, lo * n), at(a_b, lo), at(b_a, lo * n), at(b_b, lo), at(sum_a, lo * n),
                         at(sum_b, lo), carry_a + lo * n, carry_b + lo, at(sum_var, lo),
                         at(carry_var, lo), hi - lo, bits, count);
  });
}

int tfhe_hip_circuit_compare(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b,
                             const int32_t* b_a, const int32_t* b_b, int32_t* out_a, int32_t* out_b,
                             double* out_var, size_t count, int bits) {
  return circuit_reduce(ctx, 0, a_a, a_b, b_a, b_b, out_a, out_b, out_var, count, bits);
}

int tfhe_hip_circuit_eq(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b, const int32_t* b_a,
                        const int32_t* b_b, int32_t* out_a, int32_t* out_b, double* out_var,
                        size_t count, int bits) {
  return circuit_reduce(ctx, 1, a_a, a_b, b_a, b_b, out_a, out_b, out_var, count, bits);
}

int tfhe_hip_polymul(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s, int32_t* out,
                     int accumulate, size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (B == 0) return 0;
  if (!d || !s || !out) return fail(ctx, TFHE_HIP_ERR_ARG, "polynomial pointer is NULL");
  if (!multi(ctx)) return polymul_1(ctx, d, s, out, nullptr, accumulate, B);
  const size_t on = accumulate ? kN : 2 * kN - 1;
  return run_sharded(ctx, B, [=](tfhe_hip_ctx* c, size_t lo, size_t hi) {
    return polymul_1(c, d + lo * kN, s + lo * kN, out + lo * on, nullptr, accumulate, hi - lo);
  });
}

int tfhe_hip_polymul_margin(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s, double* margin,
                            size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (B == 0) return 0;
  if (!d || !s || !margin) return fail(ctx, TFHE_HIP_ERR_ARG, "polynomial pointer is NULL");
  tfhe_hip_ctx* c = multi(ctx) ? ctx->shards[0] : ctx;
  const int rc = polymul_1(c, d, s, nullptr, margin, 0, B);
  return rc && c != ctx ? fail(ctx, rc, c->err) : rc;
}

// The two parity hooks run on the first device of a multi-GPU context.
int tfhe_hip_extprod_margin(tfhe_hip_ctx* ctx, const int32_t* t, int bk_index, int32_t* out,
                            double* margin, size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  tfhe_hip_ctx* c = multi(ctx) ? ctx->shards[0] : ctx;
  auto run = [&]() -> int {
    if (int rc = ready(c)) return rc;
    if (B == 0) return 0;
    if (!t || !out) return fail(c, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
    if (bk_index < 0 || bk_index >= c->P.n) return fail(c, TFHE_HIP_ERR_ARG, "bk_index out of range");
    Staging st(c);
    const int it = st.add(t, nullptr, B * 3 * kN * 4), io = st.add(nullptr, out, B * 3 * kN * 4);
    const int im = st.add(nullptr, margin, B * 2 * 8);
    if (int rc = st.upload()) return rc;
    if (int rc = tfhe_hip_extprod_dev(c, st.dev<int32_t>(it), bk_index, st.dev<int32_t>(io),
                                      st.dev<double>(im), B)) return rc;
    return st.download();
  };
  const int rc = run();
  return rc && c != ctx ? fail(ctx, rc, c->err) : rc;
}

int tfhe_hip_extprod(tfhe_hip_ctx* ctx, const int32_t* t, int bk_index, int32_t* out, size_t B) {
  return tfhe_hip_extprod_margin(ctx, t, bk_index, out, nullptr, B);
}

int tfhe_hip_blind_rotate(tfhe_hip_ctx* ctx, int32_t* acc, const int32_t* bara, size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  tfhe_hip_ctx* c = multi(ctx) ? ctx->shards[0] : ctx;
  auto run = [&]() -> int {
    if (int rc = ready(c)) return rc;
    if (B == 0) return 0;
    if (!acc || !bara) return fail(c, TFHE_HIP_ERR_ARG, "pointer is NULL");
    Staging st(c);
    const int ia = st.add(acc, acc, B * 3 * kN * 4), ib = st.add(bara, nullptr, B * c->P.n * 4);
    if (int rc = st.upload()) return rc;
    if (int rc = tfhe_hip_blind_rotate_dev(c, st.dev<int32_t>(ia), st.dev<int32_t>(ib), B)) return rc;
    return st.download();
  };
  const int rc = run();
  return rc && c != ctx ? fail(ctx, rc, c->err) : rc;
}

}  // extern "C"

// ===========================================================================
// Device-resident sample handles
// ===========================================================================
struct tfhe_hip_samples {
  tfhe_hip_ctx* ctx = nullptr;       // single-device context that owns the buffers (NULL for a multi handle)
  size_t B = 0;
  int n = 0;
  int32_t* a = nullptr;              // [B][n]
  int32_t* b = nullptr;              // [B]
  double* var = nullptr;             // [B]
  std::vector<tfhe_hip_samples*> parts;   // multi-GPU: one handle per shard, contiguous split of B
};

namespace {

int samples_alloc_1(tfhe_hip_ctx* ctx, size_t B, tfhe_hip_samples** out) {
  if (int rc = ready(ctx)) return rc;
  tfhe_hip_samples* s = new (std::nothrow) tfhe_hip_samples();
  if (!s) return fail(ctx, TFHE_HIP_ERR_RUNTIME, "out of host memory");
  s->ctx = ctx;
  s->B = B;
  s->n = ctx->P.n;
  // one allocation: masks, bodies, variances
  const size_t ab = (B * s->n * 4 + 255) / 256 * 256, bb = (B * 4 + 255) / 256 * 256;
  char* base = nullptr;
  if (hipMalloc((void**)&base, ab + bb + std::max<size_t>(B * 8, 8)) != hipSuccess) {
    delete s;
    return fail(ctx, TFHE_HIP_ERR_RUNTIME, "hipMalloc failed for a sample handle");
  }
  s->a = (int32_t*)base;
  s->b = (int32_t*)(base + ab);
  s->var = (double*)(base + ab + bb);
  try {
    ctx->live_handles.push_back(s);
  } catch (...) {
    (void)hipFree(base);
    delete s;
    return fail(ctx, TFHE_HIP_ERR_RUNTIME, "out of host memory");
  }
  *out = s;
  return 0;
}

// the handle belongs to ctx (or to its shards, in order) and matches the loaded keys
int check_handle(tfhe_hip_ctx* ctx, const tfhe_hip_samples* s, const char* what) {
  if (!s) return fail(ctx, TFHE_HIP_ERR_ARG, std::string(what) + " handle is NULL");
  if (multi(ctx)) {
    if (s->parts.size() != ctx->shards.size()) return fail(ctx, TFHE_HIP_ERR_ARG, std::string(what) + " handle belongs to another context");
    for (size_t r = 0; r < s->parts.size(); r++)
      if (s->parts[r] && s->parts[r]->ctx != ctx->shards[r])
        return fail(ctx, TFHE_HIP_ERR_ARG, std::string(what) + " handle belongs to another context");
  } else if (s->ctx != ctx) {
    return fail(ctx, TFHE_HIP_ERR_ARG, std::string(what) + " handle belongs to another context");
  }
  if (s->n != ctx->P.n) return fail(ctx, TFHE_HIP_ERR_ARG, std::string(what) + " handle was made for other parameters");
  return 0;
}

}  // namespace

extern "C" {

int tfhe_hip_samples_alloc(tfhe_hip_ctx* ctx, size_t B, tfhe_hip_samples** out) {
  if (!ctx || !out) return TFHE_HIP_ERR_ARG;
  *out = nullptr;
  if (!multi(ctx)) return samples_alloc_1(ctx, B, out);
  if (!ctx->have_keys) return fail(ctx, TFHE_HIP_ERR_STATE, "no keys loaded");
  tfhe_hip_samples* s = new (std::nothrow) tfhe_hip_samples();
  if (!s) return fail(ctx, TFHE_HIP_ERR_RUNTIME, "out of host memory");
  s->B = B;
  s->n = ctx->P.n;
  const size_t world = ctx->shards.size();
  s->parts.assign(world, nullptr);
  for (size_t r = 0; r < world; r++) {
    const Span sp = shard_span(B, r, world);
    if (int rc = samples_alloc_1(ctx->shards[r], sp.hi - sp.lo, &s->parts[r])) {
      fail(ctx, rc, ctx->shards[r]->err);
      tfhe_hip_samples_free(s);
      return rc;
    }
  }
  *out = s;
  return 0;
}

void tfhe_hip_samples_free(tfhe_hip_samples* s) {
  if (!s) return;
  for (tfhe_hip_samples* p : s->parts) tfhe_hip_samples_free(p);
  // s->ctx is NULL once the owning context has been destroyed (release_handles): the
  // device memory went with the context, only the handle itself is left
  if (s->ctx && s->a) {
    auto& live = s->ctx->live_handles;
    live.erase(std::remove(live.begin(), live.end(), s), live.end());
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    (void)hipFree(s->a);
  }
  delete s;
}

size_t tfhe_hip_samples_count(const tfhe_hip_samples* s) { return s ? s->B : 0; }

int tfhe_hip_samples_upload(tfhe_hip_ctx* ctx, tfhe_hip_samples* s, const int32_t* a, const int32_t* b,
                            const double* var) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (int rc = check_handle(ctx, s, "sample")) return rc;
  if (s->B == 0) return 0;
  if (!a || !b) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  if (multi(ctx)) {
    size_t lo = 0;
    for (size_t r = 0; r < s->parts.size(); r++) {
      tfhe_hip_samples* p = s->parts[r];
      if (int rc = tfhe_hip_samples_upload(ctx->shards[r], p, a + lo * s->n, b + lo, at(var, lo)))
        return fail(ctx, rc, ctx->shards[r]->err);
      lo += p->B;
    }
    return 0;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(s->a, a, s->B * s->n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(s->b, b, s->B * 4, hipMemcpyHostToDevice, ctx->stream));
  if (var) HIP_TRY(ctx, hipMemcpyAsync(s->var, var, s->B * 8, hipMemcpyHostToDevice, ctx->stream));
  else HIP_TRY(ctx, hipMemsetAsync(s->var, 0, s->B * 8, ctx->stream));
  // the caller may reuse its buffers as soon as this returns
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int tfhe_hip_samples_download(tfhe_hip_ctx* ctx, const tfhe_hip_samples* s, int32_t* a, int32_t* b,
                              double* var) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (int rc = check_handle(ctx, s, "sample")) return rc;
  if (s->B == 0) return 0;
  if (!a || !b) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  if (multi(ctx)) {
    size_t lo = 0;
    for (size_t r = 0; r < s->parts.size(); r++) {
      const tfhe_hip_samples* p = s->parts[r];
      if (int rc = tfhe_hip_samples_download(ctx->shards[r], p, a + lo * s->n, b + lo, at(var, lo)))
        return fail(ctx, rc, ctx->shards[r]->err);
      lo += p->B;
    }
    return 0;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(a, s->a, s->B * s->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(b, s->b, s->B * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (var) HIP_TRY(ctx, hipMemcpyAsync(var, s->var, s->B * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int tfhe_hip_samples_gate(tfhe_hip_ctx* ctx, int gate, const tfhe_hip_samples* ca,
                          const tfhe_hip_samples* cb, const tfhe_hip_samples* cc,
                          tfhe_hip_samples* out) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (int rc = check_handle(ctx, ca, "first operand")) return rc;
  if (int rc = check_handle(ctx, cb, "second operand")) return rc;
  if (cc) if (int rc = check_handle(ctx, cc, "third operand")) return rc;
  if (int rc = check_handle(ctx, out, "result")) return rc;
  if (ca->B != out->B || cb->B != out->B || (cc && cc->B != out->B))
    return fail(ctx, TFHE_HIP_ERR_ARG, "sample handles differ in size");
  if (multi(ctx)) {      // the device-pointer calls only enqueue: the shards run side by side
    for (size_t r = 0; r < ctx->shards.size(); r++)
      if (int rc = tfhe_hip_samples_gate(ctx->shards[r], gate, ca->parts[r], cb->parts[r],
                                         cc ? cc->parts[r] : nullptr, out->parts[r]))
        return fail(ctx, rc, ctx->shards[r]->err);
    return 0;
  }
  return tfhe_hip_gate_dev(ctx, gate, ca->a, ca->b, cb->a, cb->b, cc ? cc->a : nullptr,
                           cc ? cc->b : nullptr, out->a, out->b, out->var, out->B);
}

int tfhe_hip_samples_bootstrap(tfhe_hip_ctx* ctx, int32_t mu, const tfhe_hip_samples* x,
                               tfhe_hip_samples* out) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (int rc = check_handle(ctx, x, "operand")) return rc;
  if (int rc = check_handle(ctx, out, "result")) return rc;
  if (x->B != out->B) return fail(ctx, TFHE_HIP_ERR_ARG, "sample handles differ in size");
  if (multi(ctx)) {
    for (size_t r = 0; r < ctx->shards.size(); r++)
      if (int rc = tfhe_hip_samples_bootstrap(ctx->shards[r], mu, x->parts[r], out->parts[r]))
        return fail(ctx, rc, ctx->shards[r]->err);
    return 0;
  }
  return tfhe_hip_bootstrap_dev(ctx, mu, x->a, x->b, out->a, out->b, out->var, out->B);
}

int tfhe_hip_samples_not(tfhe_hip_ctx* ctx, const tfhe_hip_samples* x, tfhe_hip_samples* out) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (int rc = check_handle(ctx, x, "operand")) return rc;
  if (int rc = check_handle(ctx, out, "result")) return rc;
  if (x->B != out->B) return fail(ctx, TFHE_HIP_ERR_ARG, "sample handles differ in size");
  if (multi(ctx)) {
    for (size_t r = 0; r < ctx->shards.size(); r++)
      if (int rc = tfhe_hip_samples_not(ctx->shards[r], x->parts[r], out->parts[r]))
        return fail(ctx, rc, ctx->shards[r]->err);
    return 0;
  }
  return tfhe_hip_not_dev(ctx, x->a, x->b, x->var, out->a, out->b, out->var, out->B);
}

int tfhe_hip_samples_constant(tfhe_hip_ctx* ctx, int value, tfhe_hip_samples* out) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (int rc = check_handle(ctx, out, "result")) return rc;
  if (multi(ctx)) {
    for (size_t r = 0; r < ctx->shards.size(); r++)
      if (int rc = tfhe_hip_samples_constant(ctx->shards[r], value, out->parts[r]))
        return fail(ctx, rc, ctx->shards[r]->err);
    return 0;
  }
  return tfhe_hip_constant_dev(ctx, value, out->a, out->b, out->var, out->B);
}

}  // extern "C"

// ===========================================================================
// Key file
// ===========================================================================
namespace {

struct KeyFileHeader {          // 64 bytes, little-endian
  char magic[8];                // "TFHEHIPK"
  uint32_t version;             // 1
  uint32_t header_bytes;        // 64
  int32_t params[8];            // n, N, k, l, bg_bit, ks_t, ks_base_bit, n_ks
  uint32_t has_ks_var;
  uint32_t reserved;
  uint64_t payload_bytes;
};
static_assert(sizeof(KeyFileHeader) == 64, "key file header");

struct KeyFileLayout {
  size_t bk, ks_a, ks_b, ks_var, total;     // byte offsets of the sections, file size
};

KeyFileLayout key_file_layout(const tfhe_hip_params* p, bool has_var) {
  KeyFileLayout L;
  const size_t base = (size_t)1 << p->ks_base_bit;
  const size_t bk_count = (size_t)p->n * p->l * (p->k + 1) * (p->k + 1) * p->N;
  const size_t rows = (size_t)p->n_ks * p->ks_t * base;
  L.bk = sizeof(KeyFileHeader);
  L.ks_a = L.bk + bk_count * 4;
  L.ks_b = L.ks_a + rows * p->n * 4;
  L.ks_var = (L.ks_b + rows * 4 + 7) / 8 * 8;
  L.total = L.ks_var + (has_var ? rows * 8 : 0);
  return L;
}

}  // namespace

extern "C" {

int tfhe_hip_key_file_size(const tfhe_hip_params* params, int has_ks_var, size_t* bytes) {
  if (!bytes || tfhe_detail::check_params(nullptr, params)) return TFHE_HIP_ERR_ARG;
  *bytes = key_file_layout(params, has_ks_var != 0).total;
  return 0;
}

int tfhe_hip_key_file_pack(const tfhe_hip_params* params, const int32_t* bk, const int32_t* ks_a,
                           const int32_t* ks_b, const double* ks_var, void* out, size_t capacity) {
  if (tfhe_detail::check_params(nullptr, params) || !bk || !ks_a || !ks_b || !out) return TFHE_HIP_ERR_ARG;
  const KeyFileLayout L = key_file_layout(params, ks_var != nullptr);
  if (capacity < L.total) return TFHE_HIP_ERR_ARG;
  KeyFileHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "TFHEHIPK", 8);
  h.version = 1;
  h.header_bytes = sizeof h;
  const int32_t pv[8] = {params->n, params->N, params->k, params->l, params->bg_bit,
                         params->ks_t, params->ks_base_bit, params->n_ks};
  memcpy(h.params, pv, sizeof pv);
  h.has_ks_var = ks_var ? 1 : 0;
  h.payload_bytes = L.total - sizeof h;
  char* o = (char*)out;
  memcpy(o, &h, sizeof h);
  memcpy(o + L.bk, bk, L.ks_a - L.bk);
  memcpy(o + L.ks_a, ks_a, L.ks_b - L.ks_a);
  const size_t rows = (size_t)params->n_ks * params->ks_t * ((size_t)1 << params->ks_base_bit);
  memcpy(o + L.ks_b, ks_b, rows * 4);
  memset(o + L.ks_b + rows * 4, 0, L.ks_var - (L.ks_b + rows * 4));
  if (ks_var) memcpy(o + L.ks_var, ks_var, rows * 8);
  return 0;
}

int tfhe_hip_load_keys_blob(tfhe_hip_ctx* ctx, const void* blob, size_t bytes) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!blob || bytes < sizeof(KeyFileHeader)) return fail(ctx, TFHE_HIP_ERR_ARG, "key file is truncated");
  KeyFileHeader h;
  memcpy(&h, blob, sizeof h);
  if (memcmp(h.magic, "TFHEHIPK", 8) != 0) return fail(ctx, TFHE_HIP_ERR_ARG, "not a key file (bad magic)");
  if (h.version != 1 || h.header_bytes != sizeof h)
    return fail(ctx, TFHE_HIP_ERR_ARG, "unsupported key file version");
  tfhe_hip_params p = {h.params[0], h.params[1], h.params[2], h.params[3],
                       h.params[4], h.params[5], h.params[6], h.params[7]};
  if (int rc = tfhe_detail::check_params(ctx, &p)) return rc;
  const KeyFileLayout L = key_file_layout(&p, h.has_ks_var != 0);
  if (bytes < L.total || h.payload_bytes != L.total - sizeof h)
    return fail(ctx, TFHE_HIP_ERR_ARG, "key file size does not match its parameters");
  const char* b = (const char*)blob;
  // the sections are 4- and 8-byte aligned relative to the file start; copy them out if the
  // caller's buffer is not
  if (((uintptr_t)b & 7) == 0)
    return tfhe_hip_load_keys(ctx, &p, (const int32_t*)(b + L.bk), (const int32_t*)(b + L.ks_a),
                              (const int32_t*)(b + L.ks_b),
                              h.has_ks_var ? (const double*)(b + L.ks_var) : nullptr);
  void* copy = malloc(L.total);
  if (!copy) return fail(ctx, TFHE_HIP_ERR_RUNTIME, "out of host memory");
  memcpy(copy, blob, L.total);
  const int rc = tfhe_hip_load_keys_blob(ctx, copy, L.total);
  free(copy);
  return rc;
}

}  // extern "C"

// ===========================================================================
// Multi-GPU contexts
// ===========================================================================
namespace {

// RCCL is bound at run time, and only when a context spans several GPUs, so that
// the library has no load-time dependency on it.
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  bool load() {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast;
  }
};
constexpr int kNcclUint8 = 1;      // ncclUint8 of ncclDataType_t

// Broadcasts the two key blobs of leaders[0] to the other leaders (one context per
// distinct GPU, their blobs already allocated) over RCCL.  Returns false when RCCL
// is unavailable or fails; the caller then copies peer to peer.
bool rccl_broadcast_keys(const std::vector<tfhe_hip_ctx*>& leaders) {
  static Rccl rccl;
  static const bool ok = rccl.load();
  if (!ok) return false;
  const int world = (int)leaders.size();
  std::vector<int> devs(world);
  for (int r = 0; r < world; r++) devs[r] = leaders[r]->device;
  std::vector<void*> comms(world, nullptr);
  if (rccl.CommInitAll(comms.data(), world, devs.data()) != 0) return false;
  bool good = true;
  for (int blob = 0; blob < 2 && good; blob++) {
    good = rccl.GroupStart() == 0;
    for (int r = 0; r < world && good; r++) {
      tfhe_hip_ctx* c = leaders[r];
      void* buf = blob == 0 ? (void*)c->bkfft : (void*)c->ksblob;
      const size_t bytes = blob == 0 ? c->bk_bytes : c->ks_bytes;
      good = hipSetDevice(c->device) == hipSuccess &&
             rccl.Broadcast(buf, buf, bytes, kNcclUint8, 0, comms[r], c->stream) == 0;
    }
    good = rccl.GroupEnd() == 0 && good;
  }
  for (int r = 0; r < world; r++) {
    if (hipSetDevice(leaders[r]->device) != hipSuccess || hipStreamSynchronize(leaders[r]->stream) != hipSuccess)
      good = false;
  }
  for (void* c : comms)
    if (c) rccl.CommDestroy(c);
  return good;
}

// dst's blobs (allocated) <- src's, across devices: peer access is enabled where the
// hardware offers it (the copy then runs over xGMI), otherwise the runtime stages it.
int peer_copy_keys(tfhe_hip_ctx* dst, const tfhe_hip_ctx* src) {
  HIP_TRY(dst, hipSetDevice(dst->device));
  if (dst->device != src->device) {
    int can = 0;
    HIP_TRY(dst, hipDeviceCanAccessPeer(&can, dst->device, src->device));
    if (can) {
      const hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        return fail(dst, TFHE_HIP_ERR_RUNTIME, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
      (void)hipGetLastError();          // "already enabled" is not an error to keep
    }
  }
  HIP_TRY(dst, hipMemcpyPeerAsync(dst->bkfft, dst->device, src->bkfft, src->device, dst->bk_bytes, dst->stream));
  HIP_TRY(dst, hipMemcpyPeerAsync(dst->ksblob, dst->device, src->ksblob, src->device, dst->ks_bytes, dst->stream));
  HIP_TRY(dst, hipStreamSynchronize(dst->stream));
  return 0;
}

}  // namespace

void tfhe_detail::release_handles(tfhe_hip_ctx* ctx) {
  for (tfhe_hip_samples* s : ctx->live_handles) {
    if (s->a) (void)hipFree(s->a);
    s->a = nullptr;
    s->b = nullptr;
    s->var = nullptr;
    s->ctx = nullptr;
  }
  ctx->live_handles.clear();
}

int tfhe_detail::load_keys_multi(tfhe_hip_ctx* ctx, const tfhe_hip_params* params, const int32_t* bk,
                                 const int32_t* ks_a, const int32_t* ks_b, const double* ks_var) {
  ctx->have_keys = false;
  // transform once, on the first device
  if (int rc = tfhe_hip_load_keys(ctx->shards[0], params, bk, ks_a, ks_b, ks_var))
    return fail(ctx, rc, ctx->shards[0]->err);
  return replicate_keys_multi(ctx, params);
}

int tfhe_detail::replicate_keys_multi(tfhe_hip_ctx* ctx, const tfhe_hip_params* params) {
  ctx->have_keys = false;
  std::vector<tfhe_hip_ctx*>& sh = ctx->shards;
  // leaders: the first context of every distinct GPU
  std::vector<tfhe_hip_ctx*> leaders;
  std::vector<int> leader_of(sh.size(), 0);
  for (size_t r = 0; r < sh.size(); r++) {
    int found = -1;
    if (getenv("TFHE_HIP_MULTI_DISTINCT") == nullptr)
      for (size_t k = 0; k < leaders.size(); k++)
        if (leaders[k]->device == sh[r]->device) found = (int)k;
    if (found < 0) {
      found = (int)leaders.size();
      leaders.push_back(sh[r]);
    }
    leader_of[r] = found;
  }
  // Test switches: FORCE_RCCL runs the broadcast even on a one-GPU communicator, NO_RCCL
  // skips it, DISTINCT treats every worker as if it sat on a GPU of its own (so that the
  // cross-device legs run on a one-GPU box).
  const bool force_rccl = getenv("TFHE_HIP_MULTI_FORCE_RCCL") != nullptr;
  ctx->key_transport = "local";
  if (leaders.size() > 1 || force_rccl) {
    // the other leaders get blobs of the right size, contents arrive below: once
    for (size_t k = 1; k < leaders.size(); k++)
      if (int rc = alloc_key_blobs(leaders[k], params)) return fail(ctx, rc, leaders[k]->err);
    bool done = false;
    if (getenv("TFHE_HIP_MULTI_NO_RCCL") == nullptr) done = rccl_broadcast_keys(leaders);
    if (!done)
      for (size_t k = 1; k < leaders.size(); k++)
        if (int rc = peer_copy_keys(leaders[k], sh[0])) return fail(ctx, rc, leaders[k]->err);
    for (size_t k = 1; k < leaders.size(); k++) leaders[k]->have_keys = true;
    ctx->key_transport = done ? "rccl" : "peer";
  }
  // the remaining contexts copy from the leader on their own GPU
  for (size_t r = 0; r < sh.size(); r++) {
    tfhe_hip_ctx* lead = leaders[leader_of[r]];
    if (sh[r] == lead) continue;
    if (int rc = tfhe_hip_import_keys_dev(sh[r], params, lead->bkfft, lead->ksblob)) return fail(ctx, rc, sh[r]->err);
  }
  ctx->P = *params;
  ctx->have_keys = true;
  return 0;
}

extern "C" const char* tfhe_hip_key_transport(tfhe_hip_ctx* ctx) {
  return ctx && ctx->have_keys ? ctx->key_transport : "none";
}

extern "C" int tfhe_hip_create_multi(tfhe_hip_ctx** out, const int* devices, int ndev) {
  if (!out) return TFHE_HIP_ERR_ARG;
  *out = nullptr;
  if (!devices || ndev < 1 || ndev > 64) return TFHE_HIP_ERR_ARG;
  tfhe_hip_ctx* ctx = new (std::nothrow) tfhe_hip_ctx();
  if (!ctx) return TFHE_HIP_ERR_RUNTIME;
  ctx->device = devices[0];
  for (int i = 0; i < ndev; i++) {
    tfhe_hip_ctx* s = nullptr;
    int rc = tfhe_hip_create(&s, devices[i]);
    if (rc == 0) {
      // a stream of its own, so that the shards of one call overlap even on a shared GPU
      hipStream_t st = nullptr;
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
        s->stream = st;
        s->owns_stream = true;
      }
      ctx->shards.push_back(s);
    } else {
      tfhe_hip_destroy(s);
      // undo: a context with no shards is a single-device context for destroy
      for (tfhe_hip_ctx* d : ctx->shards) tfhe_hip_destroy(d);
      ctx->shards.clear();
      delete ctx;
      return rc;
    }
  }
  *out = ctx;
  return 0;
}
