// Error channel, version and device queries of the C ABI (include/dkd.h).
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "common.h"

static thread_local char g_err[512] = "";

void dkd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int dkd_version(void) { return 100; }
extern "C" const char* dkd_last_error(void) { return g_err; }

extern "C" int dkd_device_info(int device, int* cu_count, char* name, int name_len) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    dkd_set_error("device_info: %s", hipGetErrorString(e));
    return DKD_ERR_HIP;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (name && name_len > 0) {
    strncpy(name, prop.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  return DKD_OK;
}


// ---- launch probe (off by default; bench.py): see DkdProbeScope in common.h
namespace {
struct ProbeRec {
  int sym;
  double flops, bytes;
  hipEvent_t e0, e1;
};
std::mutex g_probe_mu;
bool g_probe_on = false;
std::vector<ProbeRec> g_probe;
}  // namespace

DkdProbeScope::DkdProbeScope(int sym_, double flops_, double bytes_, hipStream_t s) : on(false), sym(sym_), flops(flops_), bytes(bytes_), st(s) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  if (!g_probe_on) return;
  on = true;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, st);
}

DkdProbeScope::~DkdProbeScope() {
  if (!on) return;
  (void)hipEventRecord(e1, st);
  std::lock_guard<std::mutex> lk(g_probe_mu);
  g_probe.push_back(ProbeRec{sym, flops, bytes, e0, e1});
}

extern "C" int dkd_probe_begin(void) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  g_probe.clear();
  g_probe_on = true;
  return DKD_OK;
}

extern "C" int dkd_probe_end_ex(int32_t n, double* flops, double* bytes, double* ms, int32_t* launches) {
  DKD_CHECK_ARG(n > 0 && n <= DKD_PROBE_SYMS && flops && ms && launches, "probe_end: need 0 < n <= %d arrays", DKD_PROBE_SYMS);
  std::lock_guard<std::mutex> lk(g_probe_mu);
  g_probe_on = false;
  for (int i = 0; i < n; ++i) {
    flops[i] = 0.0;
    ms[i] = 0.0;
    launches[i] = 0;
    if (bytes) bytes[i] = 0.0;
  }
  for (auto& r : g_probe) {
    (void)hipEventSynchronize(r.e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.e0, r.e1);
    if (r.sym < n) {
      flops[r.sym] += r.flops;
      if (bytes) bytes[r.sym] += r.bytes;
      ms[r.sym] += t;
      launches[r.sym] += 1;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_probe.clear();
  return DKD_OK;
}

// sym 0 = gemm_nt_kernel<128>, 1 = gemm_nt_kernel<64>, 2 = gemm_nt256_kernel.  Arrays of 3.
extern "C" int dkd_probe_end(double* flops, double* ms, int32_t* launches) { return dkd_probe_end_ex(3, flops, nullptr, ms, launches); }
