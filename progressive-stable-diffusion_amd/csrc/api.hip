// Library-level entry points: error text, device info, hipGraph capture, HIP-event profiling.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "dadd_common.h"

int dadd_init_igemm();
int dadd_init_attention();
int dadd_init_norm();
int dadd_init_attn2_fused();

namespace {
thread_local char g_err[512] = "";

struct ProfState {
  int kind = 0;
  std::vector<hipEvent_t> ev;  // pairs (start, stop)
  std::vector<double> flop;
  size_t used = 0;
} g_prof;
}  // namespace

void dadd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

bool dadd_prof_active(int kind) { return g_prof.kind == kind; }

void dadd_prof_pre(hipStream_t s) {
  if (g_prof.used + 2 > g_prof.ev.size()) {
    for (int i = 0; i < 2; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      g_prof.ev.push_back(e);
    }
  }
  (void)hipEventRecord(g_prof.ev[g_prof.used], s);
}

void dadd_prof_post(hipStream_t s, double flop) {
  if (g_prof.used + 2 > g_prof.ev.size()) return;
  (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s);
  g_prof.used += 2;
  g_prof.flop.push_back(flop);
}

extern "C" {

const char* dadd_last_error(void) { return g_err; }

int dadd_version(void) { return 100; }

int dadd_init(void) {
  int rc = dadd_init_igemm();
  if (rc == DADD_OK) rc = dadd_init_attention();
  if (rc == DADD_OK) rc = dadd_init_norm();
  return rc != DADD_OK ? rc : dadd_init_attn2_fused();
}

int dadd_device_info(int device, int64_t out[4]) {
  DADD_REQUIRE(out != nullptr, "device_info: null out");
  hipDeviceProp_t prop;
  DADD_HIP(hipGetDeviceProperties(&prop, device));
  out[0] = prop.multiProcessorCount;
  out[1] = (int64_t)prop.maxSharedMemoryPerMultiProcessor;
  out[2] = prop.clockRate;
  int arch = 0;
  const char* g = strstr(prop.gcnArchName, "gfx");
  if (g) arch = atoi(g + 3);
  out[3] = arch;
  return DADD_OK;
}

int dadd_graph_begin(void* stream) {
  DADD_REQUIRE(g_prof.kind == 0, "graph_begin: profiling is active");
  DADD_HIP(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
  return DADD_OK;
}

int dadd_graph_end(void* stream, void** graph_exec_out) {
  DADD_REQUIRE(graph_exec_out != nullptr, "graph_end: null out");
  hipGraph_t graph = nullptr;
  DADD_HIP(hipStreamEndCapture(static_cast<hipStream_t>(stream), &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    dadd_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return DADD_EHIP;
  }
  *graph_exec_out = exec;
  return DADD_OK;
}

int dadd_graph_launch(void* graph_exec, void* stream) {
  DADD_REQUIRE(graph_exec != nullptr, "graph_launch: null graph");
  DADD_HIP(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipStream_t>(stream)));
  return DADD_OK;
}

int dadd_graph_destroy(void* graph_exec) {
  if (graph_exec) DADD_HIP(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
  return DADD_OK;
}

int dadd_prof_begin(int kind) {
  DADD_REQUIRE(kind >= 1 && kind <= 3, "prof_begin: unknown kernel family %d", kind);
  g_prof.kind = kind;
  g_prof.used = 0;
  g_prof.flop.clear();
  return DADD_OK;
}

int dadd_prof_end(double out[3]) {
  DADD_REQUIRE(out != nullptr, "prof_end: null out");
  if (g_prof.kind == 0) {
    dadd_set_error("prof_end without prof_begin");
    return DADD_ESTATE;
  }
  g_prof.kind = 0;
  double ms = 0.0, flop = 0.0;
  const size_t n = g_prof.used / 2;
  if (n > 0) DADD_HIP(hipEventSynchronize(g_prof.ev[g_prof.used - 1]));
  for (size_t i = 0; i < n; ++i) {
    float t = 0.f;
    DADD_HIP(hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
    ms += t;
    flop += g_prof.flop[i];
  }
  out[0] = (double)n;
  out[1] = ms;
  out[2] = flop;
  return DADD_OK;
}

// Median interval of an EMPTY (start, stop) event pair on `stream`: what the two hipEventRecord barrier
// packets cost by themselves.  bench.py subtracts it from every bracketed launch so that the live
// per-launch time is comparable with rocprofv3's kernel-only duration.
int dadd_prof_event_overhead(void* stream, double* out_ms) {
  DADD_REQUIRE(out_ms != nullptr, "prof_event_overhead: null out");
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr int N = 33;
  hipEvent_t ev[2 * N];
  for (auto& e : ev) DADD_HIP(hipEventCreate(&e));
  for (int i = 0; i < N; ++i) {
    DADD_HIP(hipEventRecord(ev[2 * i], s));
    DADD_HIP(hipEventRecord(ev[2 * i + 1], s));
  }
  DADD_HIP(hipEventSynchronize(ev[2 * N - 1]));
  float t[N];
  for (int i = 0; i < N; ++i) DADD_HIP(hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]));
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(t, t + N);
  *out_ms = (double)t[N / 2];
  return DADD_OK;
}

}  // extern "C"
