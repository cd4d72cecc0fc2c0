// Error reporting and version of libdiga_hip.so.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace diga {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return DIGA_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

// ---- event-based kernel-family timing ---------------------------------------------------
struct ProfRec {
    int tag;
    hipEvent_t start, stop;
    double work;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_free;

static hipEvent_t prof_event() {
    if (!g_prof_free.empty()) {
        hipEvent_t e = g_prof_free.back();
        g_prof_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(int tag, hipStream_t st, double work) : st_(st) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t a = prof_event(), b = prof_event();
    if (!a || !b) return;
    (void)hipEventRecord(a, st);
    g_prof_recs.push_back({tag, a, b, work});
    stop_ = b;
}

ProfScope::~ProfScope() {
    if (stop_) (void)hipEventRecord(stop_, st_);
}

}  // namespace diga

extern "C" int diga_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(diga::g_prof_mu);
    diga::g_prof_on = on != 0;
    return DIGA_OK;
}

extern "C" int diga_prof_reset(void) {
    std::lock_guard<std::mutex> lk(diga::g_prof_mu);
    for (auto& r : diga::g_prof_recs) {
        (void)hipEventSynchronize(r.stop);
        diga::g_prof_free.push_back(r.start);
        diga::g_prof_free.push_back(r.stop);
    }
    diga::g_prof_recs.clear();
    return DIGA_OK;
}

extern "C" int diga_prof_query(int tag, int64_t* h_count, double* h_total_ms) {
    DIGA_REQUIRE(h_count && h_total_ms && tag >= 0 && tag < DIGA_PROF_NTAGS, DIGA_EINVAL, "prof_query: bad argument");
    std::lock_guard<std::mutex> lk(diga::g_prof_mu);
    int64_t n = 0;
    double ms = 0.0;
    for (auto& r : diga::g_prof_recs) {
        if (r.tag != tag) continue;
        hipError_t e = hipEventSynchronize(r.stop);
        DIGA_REQUIRE(e == hipSuccess, (int)e, "prof_query: %s", hipGetErrorString(e));
        float t = 0.f;
        e = hipEventElapsedTime(&t, r.start, r.stop);
        DIGA_REQUIRE(e == hipSuccess, (int)e, "prof_query: %s", hipGetErrorString(e));
        ms += t;
        ++n;
    }
    *h_count = n;
    *h_total_ms = ms;
    return DIGA_OK;
}

extern "C" int diga_prof_query_work(int tag, double* h_work) {
    DIGA_REQUIRE(h_work && tag >= 0 && tag < DIGA_PROF_NTAGS, DIGA_EINVAL, "prof_query_work: bad argument");
    std::lock_guard<std::mutex> lk(diga::g_prof_mu);
    double w = 0.0;
    for (auto& r : diga::g_prof_recs)
        if (r.tag == tag) w += r.work;
    *h_work = w;
    return DIGA_OK;
}

extern "C" int diga_version(void) { return DIGA_ABI_VERSION; }
extern "C" const char* diga_last_error_string(void) { return diga::g_err; }
