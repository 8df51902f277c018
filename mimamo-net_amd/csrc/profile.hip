// Optional in-library launch timing (bench.py's roofline leg): while enabled, every kernel launch of a
// category is bracketed by hipEvents on the stream it is launched on.  Off by default; zero overhead then.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "mm_common.h"

namespace mm {

struct ProfRec {
    int cat;
    double work;
    char tag[64];
    hipEvent_t e0, e1;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::mutex g_prof_mu;

bool prof_enabled() { return g_prof_on; }

void prof_before(int cat, double work, hipStream_t s, const char* tag) {
    if (!g_prof_on) return;
    ProfRec r;
    r.cat = cat;
    r.work = work;
    snprintf(r.tag, sizeof(r.tag), "%s", tag ? tag : "");
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, s);
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_recs.push_back(r);
}

void prof_after(int cat, hipStream_t s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> g(g_prof_mu);
    for (size_t i = g_recs.size(); i-- > 0;)
        if (g_recs[i].cat == cat) {
            (void)hipEventRecord(g_recs[i].e1, s);
            break;
        }
}

}  // namespace mm

extern "C" {

int mm_profile_begin(void) {
    std::lock_guard<std::mutex> g(mm::g_prof_mu);
    mm::g_recs.clear();
    mm::g_prof_on = true;
    return MM_OK;
}

// ms[c], work[c], launches[c] for c in [0, MM_PROF_CATEGORIES): 0 = conv/GEMM engine (work = algorithmic
// FLOPs), 1 = pyramid, 2 = phase window (work = algorithmic HBM bytes), 3 = Winograd transforms (bytes moved).
int mm_profile_end(double* ms, double* work, int64_t* launches) {
    if (!ms || !work || !launches) return MM_ERR_INVALID_ARG;
    mm::g_prof_on = false;
    MM_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> g(mm::g_prof_mu);
    for (int c = 0; c < MM_PROF_CATEGORIES; ++c) { ms[c] = 0; work[c] = 0; launches[c] = 0; }
    const char* dump = getenv("MM_PROF_DUMP");  // optional per-launch CSV: cat,work,ms,tag
    FILE* f = dump ? fopen(dump, "w") : nullptr;
    for (auto& r : mm::g_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess && r.cat >= 0 && r.cat < MM_PROF_CATEGORIES) {
            ms[r.cat] += t;
            work[r.cat] += r.work;
            launches[r.cat] += 1;
            if (f) fprintf(f, "%d,%.0f,%.5f,%s\n", r.cat, r.work, t, r.tag);
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    if (f) fclose(f);
    mm::g_recs.clear();
    return MM_OK;
}
}
