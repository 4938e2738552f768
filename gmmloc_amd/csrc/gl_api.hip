// Context management, error strings, params, device-memory helpers.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gl_internal.hpp"

namespace gl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// option table: name as in gl_ctx_set_option; the environment variable is GMMLOC_<NAME IN CAPITALS>
#define GL_OPTION_LIST(X) \
  X(ba_shape) X(ba_step32) X(ba_persist) X(ba_two_frames) X(ba_slow) X(ba_fixed_pack) X(ba_rendezvous_us) X(ba_test_abort_seq) X(ba_same_xcd) X(pose_waves) X(pose_regs) X(pose_compact) X(pose_compact_cap) X(fuse_records) X(bagen_nb) X(bagen_mode) X(view_slot_lds) X(view_threads) \
  X(assoc_index_min) X(assoc_grid) X(assoc_coop) X(assoc_rec_pad) X(assoc_coop_long) X(assoc_coop_bal) X(assoc_pack_mb) X(assoc_cell8) X(assoc_cell) X(assoc_globcells) X(match_desc_lds) X(pipe_lanes) X(pipe_judge) X(pipe_fuse_asm) X(schur_kper)
double* option_slot(Options& o, const char* name) {
#define X(n) \
  if (strcmp(name, #n) == 0) return &o.n;
  GL_OPTION_LIST(X)
#undef X
  return nullptr;
}
static void options_from_env(Options& o) {
#define X(n)                                          \
  {                                                   \
    char var[64] = "GMMLOC_";                         \
    size_t k = strlen(var);                           \
    for (const char* p = #n; *p && k + 1 < sizeof(var); ++p) var[k++] = (char)((*p >= 'a' && *p <= 'z') ? *p - 32 : *p); \
    var[k] = 0;                                       \
    if (const char* e = getenv(var)) o.n = atof(e);   \
  }
  GL_OPTION_LIST(X)
#undef X
}

int ctx_scratch(Ctx* c, size_t bytes, void** out) {
  if (bytes > c->scratch_bytes) {
    if (c->scratch) {
      GL_HIP(hipStreamSynchronize(c->stream));
      GL_HIP(hipFree(c->scratch));
      c->scratch = nullptr;
      c->scratch_bytes = 0;
    }
    const size_t want = bytes + bytes / 2;
    if (hipMalloc(&c->scratch, want) != hipSuccess) {
      set_error("scratch hipMalloc(%zu) failed", want);
      return GL_ERR_NOMEM;
    }
    c->scratch_bytes = want;
  }
  *out = c->scratch;
  return GL_OK;
}

// second block, for entry points that are themselves called with their inputs in the first one (the matchers' candidate cache:
// gl_search_local_points keeps the projection loop's outputs in `scratch` and hands them to gl_search_by_projection)
int ctx_scratch_b(Ctx* c, size_t bytes, void** out) {
  if (bytes > c->scratch_b_bytes) {
    if (c->scratch_b) {
      GL_HIP(hipStreamSynchronize(c->stream));
      GL_HIP(hipFree(c->scratch_b));
      c->scratch_b = nullptr;
      c->scratch_b_bytes = 0;
    }
    const size_t want = bytes + bytes / 2;
    if (hipMalloc(&c->scratch_b, want) != hipSuccess) {
      set_error("scratch hipMalloc(%zu) failed", want);
      return GL_ERR_NOMEM;
    }
    c->scratch_b_bytes = want;
  }
  *out = c->scratch_b;
  return GL_OK;
}

// third block: gl_track_frame_chain's own intermediates (its stages use the first two)
int ctx_scratch_c(Ctx* c, size_t bytes, void** out) {
  if (bytes > c->scratch_c_bytes) {
    if (c->scratch_c) {
      GL_HIP(hipStreamSynchronize(c->stream));
      GL_HIP(hipFree(c->scratch_c));
      c->scratch_c = nullptr;
      c->scratch_c_bytes = 0;
    }
    const size_t want = bytes + bytes / 2;
    if (hipMalloc(&c->scratch_c, want) != hipSuccess) {
      set_error("scratch hipMalloc(%zu) failed", want);
      return GL_ERR_NOMEM;
    }
    c->scratch_c_bytes = want;
  }
  *out = c->scratch_c;
  return GL_OK;
}

TimerScope::TimerScope(Ctx* ctx, int timer) : c(ctx), id(timer) {
  if (!c->timing) return;
  if (!c->pool.empty()) {
    e0 = c->pool.back().first;
    e1 = c->pool.back().second;
    c->pool.pop_back();
  } else {
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
      e0 = e1 = nullptr;
      return;
    }
  }
  (void)hipEventRecord(e0, c->stream);
}
TimerScope::~TimerScope() {
  if (!c->timing || !e0) return;
  (void)hipEventRecord(e1, c->stream);
  c->pending.push_back({id, {e0, e1}});
}

static void drain_timers(Ctx* c) {
  for (auto& p : c->pending) {
    float ms = 0.f;
    (void)hipEventSynchronize(p.second.second);
    if (hipEventElapsedTime(&ms, p.second.first, p.second.second) == hipSuccess) {
      c->timer_ms[p.first] += ms;
      c->timer_n[p.first] += 1;
    }
    c->pool.push_back(p.second);
  }
  c->pending.clear();
}

}  // namespace gl

extern "C" {

const char* gl_last_error_string(void) { return gl::g_err; }

int gl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// frame::sigma2_inv: init_config.hpp:60-79 (float arithmetic throughout);
// scalar defaults: gmmloc_ros/cfg/v1.yaml:27,32,35,37,40
void gl_default_params(gl_params* p) {
  if (!p) return;
  p->neighbor_dist_thresh = 2.5;
  p->tri_lambda2 = 400.0f;
  p->tri_str_thresh = 0.0064f;
  p->ba_lambda2 = 400.0f;
  p->tri_check_str_chi2 = 1;
  p->ba_first_as_prior = 1;
  float sf = 1.0f;
  p->sigma2_inv[0] = 1.0f;
  const float scale_factor = 1.2f;
  for (int i = 1; i < 8; ++i) {
    sf = sf * scale_factor;
    const float s2 = sf * sf;
    p->sigma2_inv[i] = 1.0f / s2;
  }
}

int gl_ctx_create(int device, void* hip_stream, gl_ctx_t** out) {
  GL_REQUIRE(out, "null argument");
  int n = 0;
  GL_HIP(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) {
    gl::set_error("gl_ctx_create: device %d out of range (%d devices)", device, n);
    return GL_ERR_ARG;
  }
  GL_HIP(hipSetDevice(device));
  gl::Ctx* c = new gl::Ctx();
  c->device = device;
  if (hipDeviceGetAttribute(&c->ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->ncu <= 0) c->ncu = 256;
  if (hipDeviceGetAttribute(&c->lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || c->lds_max <= 0) c->lds_max = 64 * 1024;
  c->stream = (hipStream_t)hip_stream;  // NULL = the device's default (null) stream
  gl::options_from_env(c->opt);         // the only place the knobs are read from the environment
  c->xcc_ids_trusted = gl::probe_xcc_ids(c);
  if (hipMalloc((void**)&c->counters, GL_COUNTER_COUNT * sizeof(int32_t)) != hipSuccess ||
      hipMemsetAsync(c->counters, 0, GL_COUNTER_COUNT * sizeof(int32_t), c->stream) != hipSuccess) {
    gl::set_error("gl_ctx_create: counter allocation failed");
    delete c;
    return GL_ERR_NOMEM;
  }
  *out = (gl_ctx_t*)c;
  return GL_OK;
}

int gl_ctx_destroy(gl_ctx_t* ctx) {
  if (!ctx) return GL_OK;
  gl::Ctx* c = gl::C(ctx);
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  gl::drain_timers(c);
  for (auto& p : c->pool) {
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->scratch_b) (void)hipFree(c->scratch_b);
  if (c->scratch_c) (void)hipFree(c->scratch_c);
  if (c->counters) (void)hipFree(c->counters);
  if (c->host_word) (void)hipHostFree(c->host_word);
  for (int k = 0; k < 3; ++k) {
    if (c->lane_stream[k]) (void)hipStreamDestroy(c->lane_stream[k]);
    if (c->ev_join[k]) (void)hipEventDestroy(c->ev_join[k]);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->dev_stage) (void)hipFree(c->dev_stage);
  if (c->host_stage) (void)hipHostFree(c->host_stage);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return GL_OK;
}

// (the null stream belongs to the calling thread's CURRENT device: every entry point that touches the stream
// makes the context's device current first)
int gl_ctx_synchronize(gl_ctx_t* ctx) {
  GL_REQUIRE(ctx, "null argument");
  GL_HIP(hipSetDevice(gl::C(ctx)->device));
  GL_HIP(hipStreamSynchronize(gl::C(ctx)->stream));
  return GL_OK;
}

int gl_ctx_set_option(gl_ctx_t* ctx, const char* name, double value) {
  GL_REQUIRE(ctx && name, "null argument");
  double* slot = gl::option_slot(gl::C(ctx)->opt, name);
  if (!slot) {
    gl::set_error("gl_ctx_set_option: unknown option '%s'", name);
    return GL_ERR_ARG;
  }
  *slot = value;
  return GL_OK;
}
int gl_ctx_get_option(gl_ctx_t* ctx, const char* name, double* value) {
  GL_REQUIRE(ctx && name && value, "null argument");
  const double* slot = gl::option_slot(gl::C(ctx)->opt, name);
  if (!slot) {
    gl::set_error("gl_ctx_get_option: unknown option '%s'", name);
    return GL_ERR_ARG;
  }
  *value = *slot;
  return GL_OK;
}

void* gl_ctx_stream(gl_ctx_t* ctx) { return ctx ? (void*)gl::C(ctx)->stream : nullptr; }

int gl_ctx_timing_enable(gl_ctx_t* ctx, int on) {
  GL_REQUIRE(ctx, "null argument");
  gl::C(ctx)->timing = on != 0;
  return GL_OK;
}

int gl_ctx_timing_read(gl_ctx_t* ctx, int timer, double* total_ms, int64_t* launches, int reset) {
  GL_REQUIRE(ctx && timer >= 0 && timer < GL_TIMER_COUNT, "bad argument");
  gl::Ctx* c = gl::C(ctx);
  gl::drain_timers(c);
  if (total_ms) *total_ms = c->timer_ms[timer];
  if (launches) *launches = c->timer_n[timer];
  if (reset) {
    c->timer_ms[timer] = 0.0;
    c->timer_n[timer] = 0;
  }
  return GL_OK;
}

int gl_ctx_counter_read(gl_ctx_t* ctx, int counter, int64_t* value, int reset) {
  GL_REQUIRE(ctx && value && counter >= 0 && counter < GL_COUNTER_COUNT, "bad argument");
  gl::Ctx* c = gl::C(ctx);
  if (counter == GL_COUNTER_BA_COOP_FALLBACK) {  // kept on the host (the launcher counts it)
    *value = c->coop_fallbacks;
    if (reset) c->coop_fallbacks = 0;
    return GL_OK;
  }
  GL_HIP(hipSetDevice(c->device));
  int32_t v = 0;
  GL_HIP(hipMemcpyAsync(&v, c->counters + counter, sizeof(v), hipMemcpyDeviceToHost, c->stream));
  if (reset) GL_HIP(hipMemsetAsync(c->counters + counter, 0, sizeof(v), c->stream));
  GL_HIP(hipStreamSynchronize(c->stream));
  *value = v;
  return GL_OK;
}

int gl_ctx_set_stats_buffers(gl_ctx_t* ctx, int32_t* trials_dev, int32_t* iters_dev, int n) {
  GL_REQUIRE(ctx, "null context");
  gl::C(ctx)->stats = n > 0 ? trials_dev : nullptr;
  gl::C(ctx)->stats_iters = (n > 0 && trials_dev) ? iters_dev : nullptr;
  gl::C(ctx)->stats_n = trials_dev ? n : 0;
  return GL_OK;
}
int gl_ctx_set_stats_buffer(gl_ctx_t* ctx, int32_t* trials_dev, int n) { return gl_ctx_set_stats_buffers(ctx, trials_dev, nullptr, n); }
int gl_ctx_set_edge_stats_buffer(gl_ctx_t* ctx, int32_t* edges_dev, int n) {
  GL_REQUIRE(ctx, "null context");
  gl::C(ctx)->stats_edges = n > 0 ? edges_dev : nullptr;
  gl::C(ctx)->stats_edges_n = edges_dev ? n : 0;
  return GL_OK;
}

int gl_malloc(gl_ctx_t* ctx, size_t bytes, void** dev_out) {
  GL_REQUIRE(ctx && dev_out, "null argument");
  GL_HIP(hipSetDevice(gl::C(ctx)->device));
  if (hipMalloc(dev_out, bytes ? bytes : 1) != hipSuccess) {
    gl::set_error("gl_malloc(%zu) failed", bytes);
    return GL_ERR_NOMEM;
  }
  return GL_OK;
}
int gl_free(gl_ctx_t* ctx, void* dev) {
  GL_REQUIRE(ctx, "null argument");
  GL_HIP(hipSetDevice(gl::C(ctx)->device));
  if (dev) GL_HIP(hipFree(dev));
  return GL_OK;
}
int gl_memcpy_h2d(gl_ctx_t* ctx, void* dst_dev, const void* src, size_t bytes) {
  GL_REQUIRE(ctx && dst_dev && src, "null argument");
  GL_HIP(hipSetDevice(gl::C(ctx)->device));
  GL_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, gl::C(ctx)->stream));
  GL_HIP(hipStreamSynchronize(gl::C(ctx)->stream));
  return GL_OK;
}
int gl_memcpy_d2h(gl_ctx_t* ctx, void* dst, const void* src_dev, size_t bytes) {
  GL_REQUIRE(ctx && dst && src_dev, "null argument");
  GL_HIP(hipSetDevice(gl::C(ctx)->device));
  GL_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, gl::C(ctx)->stream));
  GL_HIP(hipStreamSynchronize(gl::C(ctx)->stream));
  return GL_OK;
}
int gl_malloc_host(gl_ctx_t* ctx, size_t bytes, void** host_out) {
  GL_REQUIRE(ctx && host_out, "null argument");
  GL_HIP(hipSetDevice(gl::C(ctx)->device));
  if (hipHostMalloc(host_out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    gl::set_error("gl_malloc_host(%zu) failed", bytes);
    return GL_ERR_NOMEM;
  }
  return GL_OK;
}
int gl_free_host(gl_ctx_t* ctx, void* host) {
  GL_REQUIRE(ctx, "null argument");
  GL_HIP(hipSetDevice(gl::C(ctx)->device));
  if (host) GL_HIP(hipHostFree(host));
  return GL_OK;
}
int gl_memcpy_h2d_async(gl_ctx_t* ctx, void* dst_dev, const void* src_pinned, size_t bytes) {
  GL_REQUIRE(ctx && dst_dev && src_pinned, "null argument");
  GL_HIP(hipSetDevice(gl::C(ctx)->device));
  GL_HIP(hipMemcpyAsync(dst_dev, src_pinned, bytes, hipMemcpyHostToDevice, gl::C(ctx)->stream));
  return GL_OK;
}
int gl_memcpy_d2h_async(gl_ctx_t* ctx, void* dst_pinned, const void* src_dev, size_t bytes) {
  GL_REQUIRE(ctx && dst_pinned && src_dev, "null argument");
  GL_HIP(hipSetDevice(gl::C(ctx)->device));
  GL_HIP(hipMemcpyAsync(dst_pinned, src_dev, bytes, hipMemcpyDeviceToHost, gl::C(ctx)->stream));
  return GL_OK;
}

}  // extern "C"
