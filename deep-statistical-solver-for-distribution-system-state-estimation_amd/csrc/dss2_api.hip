// Library-level entry points of libdss2_hip.so: error string, version, topology hash.
#include <stdarg.h>
#include <mutex>
#include <stdlib.h>

#include "dss2_common.hpp"

namespace dss2 {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- launch plans ------------------------------------------------------------------------------------------------------------
struct Plan { std::vector<std::function<int(void*)>> ops; };
static std::atomic<Plan*> g_recording{nullptr};      // process-wide: a step's backward launches come from autograd's own thread
static std::mutex g_plan_mutex;

bool plan_recording() { return g_recording.load(std::memory_order_acquire) != nullptr; }
void plan_record(std::function<int(void*)> op) {
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  Plan* pl = g_recording.load(std::memory_order_acquire);
  if (pl) pl->ops.emplace_back(std::move(op));
}

}  // namespace dss2

struct dss2_plan { dss2::Plan p; };

extern "C" int dss2_plan_begin(dss2_plan** out) {
  using namespace dss2;
  if (!out) { set_error("plan_begin: null argument"); return 2; }
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  if (g_recording.load() != nullptr) { set_error("plan_begin: another plan is recording"); return 2; }
  dss2_plan* pl = new dss2_plan();
  g_recording.store(&pl->p, std::memory_order_release);
  *out = pl;
  return 0;
}

extern "C" int dss2_plan_end(dss2_plan* plan) {
  using namespace dss2;
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  if (!plan || g_recording.load() != &plan->p) { set_error("plan_end: this plan is not recording"); return 2; }
  g_recording.store(nullptr, std::memory_order_release);
  return 0;
}

extern "C" int dss2_plan_size(const dss2_plan* plan) { return plan ? (int)plan->p.ops.size() : 0; }

extern "C" int dss2_plan_run(const dss2_plan* plan, void* stream) {
  using namespace dss2;
  if (!plan) { set_error("plan_run: null plan"); return 2; }
  if (plan_recording()) { set_error("plan_run: a plan is recording"); return 2; }
  static const int sync_each = [] { const char* e = getenv("DSS2_PLAN_SYNC"); return e ? atoi(e) : 0; }();      // diagnostic: find the launch that faults
  int i = 0;
  for (const auto& op : plan->p.ops) {
    const int rc = op(stream);
    if (rc) return rc;
    if (sync_each) {
      const hipError_t e = hipStreamSynchronize(as_stream(stream));
      if (e != hipSuccess) { set_error("plan_run: launch %d of %d failed: %s", i, (int)plan->p.ops.size(), hipGetErrorString(e)); return 1; }
      if (sync_each > 1) fprintf(stderr, "plan_run: launch %d ok\n", i);
    }
    ++i;
  }
  return 0;
}

extern "C" void dss2_plan_destroy(dss2_plan* plan) {
  if (!plan) return;
  { std::lock_guard<std::mutex> lock(dss2::g_plan_mutex); if (dss2::g_recording.load() == &plan->p) dss2::g_recording.store(nullptr); }
  delete plan;
}

extern "C" const char* dss2_last_error(void) { return dss2::g_err; }

extern "C" int dss2_version(void) { return 1; }

