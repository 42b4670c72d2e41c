// Error reporting + version for libfil_hip.so.
#include "common.h"
#include <cstdlib>
#include <cstring>

#include <string.h>
#include <dlfcn.h>

#include <mutex>
#include <string>
#include <vector>

namespace fil {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- opt-in kernel timing ------------------------------------------------------------------------
struct ProfRec {
  const char* name;
  double work, executed;
  hipEvent_t e0, e1;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_next = 0;

bool prof_enabled() { return g_prof_on; }

static hipEvent_t pool_event() {
  if (g_pool_next == g_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    g_pool.push_back(e);
  }
  return g_pool[g_pool_next++];
}

// fil_profile_begin(filter): filter = "<substr>[,<substr>...]" or NULL; only scopes whose name contains one of the
// substrings record events -- an event pair costs about 5 us of stream time per scope, so bench.py's timed region
// profiles the GEMM kernels only and takes the full per-kernel table from a separate pass.
static std::vector<std::string> g_prof_filter;
static void load_filter(const char* f) {
  g_prof_filter.clear();
  if (f == nullptr) return;
  std::string cur;
  for (const char* p = f;; ++p) {
    if (*p == ',' || *p == 0) {
      if (!cur.empty()) g_prof_filter.push_back(cur);
      cur.clear();
      if (*p == 0) break;
    } else {
      cur.push_back(*p);
    }
  }
}
static bool scope_selected(const char* name) {
  if (g_prof_filter.empty()) return true;
  for (const auto& f : g_prof_filter)
    if (strstr(name, f.c_str()) != nullptr) return true;
  return false;
}

bool prof_begin_scope(const char* name, hipStream_t st, double work, double executed) {
  if (!scope_selected(name)) return false;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{name, work, executed, pool_event(), pool_event()};
  (void)hipEventRecord(r.e0, st);
  g_prof.push_back(r);
  return true;
}

void prof_end_scope(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().e1, st);
}

// ---- roctx ranges (opt-in, FIL_ROCTX=1): the marker library is looked up at run time so that libfil_hip.so does not link it
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  bool on = false;
};
static const Roctx& roctx() {
  static const Roctx r = [] {
    Roctx x;
    const char* e = getenv("FIL_ROCTX");
    if (e == nullptr || atoi(e) == 0) return x;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      if (h == nullptr) continue;
      x.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (x.push != nullptr && x.pop != nullptr) {
        x.on = true;
        break;
      }
    }
    return x;
  }();
  return r;
}
bool roctx_enabled() { return roctx().on; }
void roctx_push(const char* name) { (void)roctx().push(name); }
void roctx_pop() { (void)roctx().pop(); }

}  // namespace fil

extern "C" int fil_profile_begin(const char* filter) {
  std::lock_guard<std::mutex> lk(fil::g_prof_mu);
  fil::g_prof.clear();
  fil::g_pool_next = 0;
  fil::load_filter(filter);
  fil::g_prof_on = true;
  return FIL_OK;
}

// Writes one line per kernel name: "name count total_ms work_per_launch executed_per_launch\n"; returns bytes needed (incl. NUL).
extern "C" size_t fil_profile_end(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lk(fil::g_prof_mu);
  fil::g_prof_on = false;
  struct Agg { std::string name; long count; double ms; double work; double executed; };
  std::vector<Agg> aggs;
  for (auto& r : fil::g_prof) {
    (void)hipEventSynchronize(r.e1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
    Agg* a = nullptr;
    for (auto& x : aggs) if (x.name == r.name) { a = &x; break; }
    if (!a) { aggs.push_back(Agg{r.name, 0, 0.0, r.work, r.executed}); a = &aggs.back(); }
    a->count += 1;
    a->ms += ms;
  }
  std::string out;
  char line[256];
  for (auto& a : aggs) {
    snprintf(line, sizeof(line), "%s %ld %.6f %.6e %.6e\n", a.name.c_str(), a.count, a.ms, a.work, a.executed);
    out += line;
  }
  fil::g_prof.clear();
  if (buf && cap > 0) {
    const size_t n = std::min(cap - 1, out.size());
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}

extern "C" int fil_version(void) { return FIL_ABI_VERSION; }
extern "C" const char* fil_last_error(void) { return fil::g_err; }
