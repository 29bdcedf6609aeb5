// capi.hip — error plumbing and device query of the C ABI (include/setok_hip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

char* setok_err_buf() { return g_err; }

int setok_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int setok_abi_version(void) { return SETOK_ABI_VERSION; }
extern "C" const char* setok_last_error(void) { return g_err; }

extern "C" int setok_device_info(char* name_host, int name_cap, int* cu_count_host) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        if (name_host && name_cap > 0) name_host[0] = 0;
        if (cu_count_host) *cu_count_host = 0;
        return setok_fail(SETOK_EUNSUPPORTED, "no HIP device visible");
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return setok_fail(SETOK_ELAUNCH, "hipGetDeviceProperties failed");
    if (name_host && name_cap > 0) {
        snprintf(name_host, name_cap, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count_host) *cu_count_host = p.multiProcessorCount;
    return SETOK_OK;
}

// ---- launch profiler ---------------------------------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct ProfRec { int kind, cls; double work, bytes; hipEvent_t e0, e1; bool attach, started, stopped;
                 int32_t* rows_host; int rows_full; double bytes_fixed; };   // a launch with a DEVICE-side row count: the count is copied back beside the launch
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
volatile int g_prof_on = 0;
bool g_prof_open = false;                       // between setok_profile_start and setok_profile_stop (g_prof_on drops while paused)
thread_local std::vector<int> g_open;               // indices of this thread's open attach-scopes, innermost last
}  // namespace

bool setok_prof_on() { return g_prof_on != 0; }

int setok_prof_begin(hipStream_t s, int kind, int cls, double work, double bytes, bool attach) {
    {   // a launch that is being CAPTURED into a graph is not measured: an event recorded under capture belongs to the graph and has no time of its own
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return -1;
    }
    ProfRec r{kind, cls, work, bytes, nullptr, nullptr, attach, false, false, nullptr, 0, 0.0};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
    if (!attach) { (void)hipEventRecord(r.e0, s); r.started = true; }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
    const int idx = (int)g_prof.size() - 1;
    if (attach) g_open.push_back(idx);
    return idx;
}

hipEvent_t setok_prof_start_event() {
    if (g_open.empty()) return nullptr;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const int idx = g_open.back();
    if (idx >= (int)g_prof.size() || g_prof[idx].started) return nullptr;      // (the records were reset under an open scope: nothing to attach to)
    g_prof[idx].started = true;
    return g_prof[idx].e0;
}

hipEvent_t setok_prof_stop_event() {
    if (g_open.empty()) return nullptr;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const int idx = g_open.back();
    if (idx >= (int)g_prof.size() || g_prof[idx].stopped) return nullptr;
    g_prof[idx].stopped = true;
    return g_prof[idx].e1;
}

void setok_prof_end(hipStream_t s, int index) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_open.empty() && g_open.back() == index) g_open.pop_back();
    if (index < 0 || index >= (int)g_prof.size()) return;
    ProfRec& r = g_prof[index];
    if (!r.started) { r.kind = -1; (void)hipEventRecord(r.e0, s); r.started = true; }   // an attach-scope whose path launched the plain way: no start time, the record is dropped
    if (!r.stopped) { (void)hipEventRecord(r.e1, s); r.stopped = true; }
}

// The record's work / bytes were computed for `rows_full` rows; the launch processes *rows_dev (<= rows_full) of them.  The count is copied to a
// pinned word in stream order (it is final when the launch is enqueued), and setok_profile_stop scales work and the row-proportional bytes
// (everything but `bytes_fixed`, the weight) by it: a launch sized for the worst case is never credited with rows it skipped.
// The pinned words are slots of ONE ring allocated by setok_profile_start (a hipHostMalloc per launch is a synchronising allocation
// inside the measured region and illegal under stream capture).  A launch that cannot be priced — ring exhausted, or the stream is being captured
// into a graph (a device-to-host copy per replay is not what the caller asked to record) — has its record dropped rather than over-credited.
namespace {
constexpr int PROF_RING = 8192;
int32_t* g_prof_ring = nullptr;
int g_prof_ring_used = 0;
// Launches that read the SAME device word in CONSECUTIVE profiler records on one stream (the six GEMMs of the inter encoder and `out` all read sum(L_i))
// share one copy of it: the word cannot have changed unless something in between writes row counts — setok_cluster_sort, the only producer in this
// library, says so (setok_prof_rows_changed), and any other record in between (another GEMM, a clustering call) ends the sharing as well.  Round 5: the copies were seven blit
// dispatches per step, ~9 us each with the idle time either side, inside the timed region.
const int32_t* g_rows_last_dev = nullptr;
int32_t* g_rows_last_host = nullptr;
hipStream_t g_rows_last_stream = nullptr;
int g_rows_last_index = -2;
}
void setok_prof_rows_changed() {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_rows_last_dev = nullptr;
}
void setok_prof_rows(hipStream_t s, int index, const int32_t* rows_dev, int rows_full, double bytes_fixed) {
    if (index < 0 || !rows_dev || rows_full <= 0) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (index >= (int)g_prof.size()) return;
    if (capturing || !g_prof_ring || g_prof_ring_used >= PROF_RING) { g_prof[index].kind = -1; return; }
    int32_t* h;
    if (rows_dev == g_rows_last_dev && s == g_rows_last_stream && g_rows_last_host && index == g_rows_last_index + 1) h = g_rows_last_host;      // the same word, nothing in between that writes it
    else {
        h = g_prof_ring + g_prof_ring_used++;
        *h = rows_full;
        (void)hipMemcpyAsync(h, rows_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s);
        g_rows_last_dev = rows_dev; g_rows_last_host = h; g_rows_last_stream = s;
    }
    g_rows_last_index = index;
    g_prof[index].rows_host = h; g_prof[index].rows_full = rows_full; g_prof[index].bytes_fixed = bytes_fixed;
}

extern "C" int setok_profile_start(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
    if (!g_prof_ring && hipHostMalloc((void**)&g_prof_ring, PROF_RING * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
        g_prof_ring = nullptr;
        (void)hipGetLastError();
    }
    g_prof_ring_used = 0;
    g_rows_last_dev = nullptr; g_rows_last_host = nullptr; g_rows_last_index = -2;
    g_prof_on = 1;
    g_prof_open = true;
    return SETOK_OK;
}

// Between setok_profile_start and setok_profile_stop: pause != 0 stops attaching events to launches (what was recorded stays), 0 resumes — a caller that times
// many steps probes a few of them (an event pair per GEMM launch is not free: ~4 us of device time per launch, 0.43 ms of a 42 ms step) without losing the rest.
extern "C" int setok_profile_pause(int pause) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_open) return setok_fail(SETOK_EINVAL, "setok_profile_pause: no recording is open (setok_profile_start)");
    g_prof_on = pause ? 0 : 1;
    g_rows_last_dev = nullptr;                       // (records are no longer consecutive across a pause)
    return SETOK_OK;
}

// Stops recording, waits for the recorded launches and writes up to `cap` records; returns the number of records (or a negative code).
extern "C" int setok_profile_stop(int* kind, int* cls, double* work, double* bytes, float* ms, int cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = 0;
    g_prof_open = false;
    int n = 0, dropped = 0;
    for (auto& r : g_prof) {
        float t = 0.f;
        if (r.kind < 0) ++dropped;
        const bool ok = hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess;
        if (ok && r.rows_host) {                             // (e1 has completed, hence so has the copy enqueued before the launch)
            const double f = (double)(*r.rows_host < r.rows_full ? (*r.rows_host < 0 ? 0 : *r.rows_host) : r.rows_full) / (double)r.rows_full;
            r.work *= f;
            r.bytes = r.bytes_fixed + (r.bytes - r.bytes_fixed) * f;
        }
        if (ok && r.kind >= 0 && n < cap && kind && cls && work && bytes && ms) { kind[n] = r.kind; cls[n] = r.cls; work[n] = r.work; bytes[n] = r.bytes; ms[n] = t; ++n; }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    const int total = (int)g_prof.size() - dropped;
    g_prof.clear();
    return n < total && cap >= total ? -1 : n;
}
