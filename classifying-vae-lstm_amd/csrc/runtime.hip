// runtime.hip -- device query, Philox RNG, hipGraph wrappers, opt-in event profiler.
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "philox.h"

#include <mutex>
#include <set>
#include <utility>

namespace clv {

int allow_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> granted;      // largest size set so far per (device, kernel)
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  std::lock_guard<std::mutex> lock(mu);
  auto it = granted.find({dev, kernel});
  if (it != granted.end() && it->second >= bytes) return CLV_OK;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  granted[{dev, kernel}] = bytes;
  return CLV_OK;
}


// ---------------------------------------------------------------- profiler --
struct ProfState {
  bool on = false;
  std::mutex mu;
  struct Pending { std::string name; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<std::pair<std::string, std::pair<int, float>>> totals;   // insertion-ordered
  std::string cur_name;
  hipEvent_t cur_a = nullptr;
};
static ProfState g_prof;

bool prof_on() { return g_prof.on; }

void prof_begin(const char* name, hipStream_t s) {
  hipEvent_t a;
  if (hipEventCreate(&a) != hipSuccess) return;
  hipEventRecord(a, s);
  g_prof.cur_name = name;
  g_prof.cur_a = a;
}

void prof_end(hipStream_t s) {
  if (!g_prof.cur_a) return;
  hipEvent_t b;
  if (hipEventCreate(&b) != hipSuccess) return;
  hipEventRecord(b, s);
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.pending.push_back({g_prof.cur_name, g_prof.cur_a, b});
  g_prof.cur_a = nullptr;
}

// ------------------------------------------------------------------ Philox --
// (philox4x32_10, u01, philox_normal_at, philox_uniform_at: philox.h)

// element i uses counter (i>>2) and word/branch (i&3): a pure function of (seed, step, stream, index)
template <bool NORMAL>
__global__ void philox_kernel(float* out, int64_t n, uint32_t k0, uint32_t k1, uint32_t step, const int32_t* step_dev,
                              uint32_t stream_id, uint64_t first) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // quad index relative to first>>2
  const uint64_t base = first >> 2;
  const uint64_t ctr = base + (uint64_t)q;
  const uint32_t st = step + (step_dev ? (uint32_t)*step_dev : 0u);
  uint32_t r[4];
  philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), stream_id, st, k0, k1, r);
  float v[4];
  if (NORMAL) {
    const float r0 = sqrtf(-2.f * logf(u01(r[0]))), t0 = 6.283185307179586f * u01(r[1]);
    const float r1 = sqrtf(-2.f * logf(u01(r[2]))), t1 = 6.283185307179586f * u01(r[3]);
    v[0] = r0 * cosf(t0); v[1] = r0 * sinf(t0); v[2] = r1 * cosf(t1); v[3] = r1 * sinf(t1);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = u01(r[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t gi = ctr * 4 + j;
    if (gi >= first && gi < first + (uint64_t)n) out[gi - first] = v[j];
  }
}

// two independent draws (the label noise and the latent noise of a training step) in one launch: blocks
// [0, nb0) serve the first, the rest the second
struct PhiloxSeg { float* out; int64_t n; uint32_t stream_id; uint64_t first; };
__global__ void philox_normal2_kernel(PhiloxSeg a, PhiloxSeg b, unsigned nb0, uint32_t k0, uint32_t k1, uint32_t step,
                                      const int32_t* step_dev) {
  const bool second = blockIdx.x >= nb0;
  const PhiloxSeg sg = second ? b : a;
  const int64_t q = (int64_t)(blockIdx.x - (second ? nb0 : 0u)) * blockDim.x + threadIdx.x;
  const uint64_t ctr = (sg.first >> 2) + (uint64_t)q;
  const uint32_t st = step + (step_dev ? (uint32_t)*step_dev : 0u);
  uint32_t r[4];
  philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), sg.stream_id, st, k0, k1, r);
  const float r0 = sqrtf(-2.f * logf(u01(r[0]))), t0 = 6.283185307179586f * u01(r[1]);
  const float r1 = sqrtf(-2.f * logf(u01(r[2]))), t1 = 6.283185307179586f * u01(r[3]);
  const float v[4] = {r0 * cosf(t0), r0 * sinf(t0), r1 * cosf(t1), r1 * sinf(t1)};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t gi = ctr * 4 + j;
    if (gi >= sg.first && gi < sg.first + (uint64_t)sg.n) sg.out[gi - sg.first] = v[j];
  }
}

template <bool NORMAL>
static int launch_philox(float* out, int64_t n, uint64_t seed, uint32_t step, const int32_t* step_dev,
                         uint32_t stream_id, uint64_t first, hipStream_t s) {
  if (!out || n <= 0) return CLV_EINVAL;
  const uint64_t q0 = first >> 2, q1 = (first + (uint64_t)n + 3) >> 2;
  const uint64_t quads = q1 - q0;
  ProfScope p(NORMAL ? "philox_normal" : "philox_uniform", s);
  hipLaunchKernelGGL((philox_kernel<NORMAL>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, out, n,
                     (uint32_t)seed, (uint32_t)(seed >> 32), step, step_dev, stream_id, first);
  return launch_status();
}

}  // namespace clv

using namespace clv;

extern "C" int clv_version(void) { return CLV_ABI_VERSION; }

extern "C" int clv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strstr(p.gcnArchName, "gfx950")) ++ok;
  }
  return ok;
}

extern "C" const char* clv_error_string(int code) {
  switch (code) {
    case CLV_OK: return "ok";
    case CLV_EINVAL: return "invalid argument or unsupported shape";
    case CLV_EWORKSPACE: return "workspace missing or too small";
    case CLV_ENOGPU: return "no gfx950 device";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

extern "C" int clv_philox_normal(float* out, int64_t n, uint64_t seed, uint32_t step, const int32_t* step_dev,
                                 uint32_t stream_id, uint64_t first_index, void* stream) {
  return launch_philox<true>(out, n, seed, step, step_dev, stream_id, first_index, (hipStream_t)stream);
}
extern "C" int clv_philox_normal2(float* out0, int64_t n0, uint32_t stream_id0, uint64_t first_index0,
                                  float* out1, int64_t n1, uint32_t stream_id1, uint64_t first_index1,
                                  uint64_t seed, uint32_t step, const int32_t* step_dev, void* stream) {
  if (!out0 || !out1 || n0 <= 0 || n1 <= 0) return CLV_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  auto nblocks = [](uint64_t first, int64_t n) {
    const uint64_t quads = ((first + (uint64_t)n + 3) >> 2) - (first >> 2);
    return (unsigned)((quads + 255) / 256);
  };
  const unsigned nb0 = nblocks(first_index0, n0), nb1 = nblocks(first_index1, n1);
  PhiloxSeg a{out0, n0, stream_id0, first_index0}, b{out1, n1, stream_id1, first_index1};
  ProfScope p("philox_normal", s);
  hipLaunchKernelGGL(philox_normal2_kernel, dim3(nb0 + nb1), dim3(256), 0, s, a, b, nb0, (uint32_t)seed,
                     (uint32_t)(seed >> 32), step, step_dev);
  return launch_status();
}
extern "C" int clv_philox_uniform(float* out, int64_t n, uint64_t seed, uint32_t step, const int32_t* step_dev,
                                  uint32_t stream_id, uint64_t first_index, void* stream) {
  return launch_philox<false>(out, n, seed, step, step_dev, stream_id, first_index, (hipStream_t)stream);
}

__global__ void i32_add_kernel(int32_t* p, int32_t v) { *p += v; }

extern "C" int clv_i32_add(int32_t* counter, int32_t v, void* stream) {
  if (!counter) return CLV_EINVAL;
  hipLaunchKernelGGL(i32_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, v);
  return clv::launch_status();
}

// ------------------------------------------------------------------ graphs --
extern "C" int clv_graph_begin_capture(void* stream) {
  CLV_HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return CLV_OK;
}
extern "C" int clv_graph_end_capture(void* stream, void** graph_exec_out) {
  if (!graph_exec_out) return CLV_EINVAL;
  hipGraph_t g = nullptr;
  CLV_HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e != hipSuccess) return (int)e;
  *graph_exec_out = (void*)ge;
  return CLV_OK;
}
extern "C" int clv_graph_launch(void* graph_exec, void* stream) {
  CLV_HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return CLV_OK;
}
extern "C" int clv_graph_destroy(void* graph_exec) {
  if (graph_exec) CLV_HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return CLV_OK;
}

// ---------------------------------------------------------------- profiler --
extern "C" int clv_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.on = on != 0;
  if (on) {
    for (auto& p : g_prof.pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    g_prof.pending.clear();
    g_prof.totals.clear();
  }
  return CLV_OK;
}

// an EMPTY bracket on `stream` (two event records, nothing between them), recorded as "event_pair": what the profiler's
// events themselves add to every bracketed launch -- bench.py reports it next to the kernel durations it measures
extern "C" int clv_prof_empty_scope(void* stream) {
  ProfScope p("event_pair", (hipStream_t)stream);
  return CLV_OK;
}

extern "C" int clv_prof_collect(clv_prof_record* host_out, int cap) {
  CLV_HIP_TRY(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g_prof.mu);
  for (auto& p : g_prof.pending) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, p.a, p.b);
    hipEventDestroy(p.a); hipEventDestroy(p.b);
    bool found = false;
    for (auto& t : g_prof.totals)
      if (t.first == p.name) { t.second.first += 1; t.second.second += ms; found = true; break; }
    if (!found) g_prof.totals.push_back({p.name, {1, ms}});
  }
  g_prof.pending.clear();
  int n = 0;
  for (auto& t : g_prof.totals) {
    if (n >= cap || !host_out) break;
    memset(&host_out[n], 0, sizeof(clv_prof_record));
    strncpy(host_out[n].name, t.first.c_str(), sizeof(host_out[n].name) - 1);
    host_out[n].launches = t.second.first;
    host_out[n].total_ms = t.second.second;
    ++n;
  }
  return n;
}
