// mi355_flow -- launch-schedule trace (test infrastructure inside the product library, OFF unless mi355_sched_trace(1) was called).
//
// The multi-stream forwards (engine.hip / qwen_engine.hip / flux_engine.hip `forward_core`: the text chain on a plan-owned side stream,
// forked and joined with events) are checked for data races on a model of the HIP stream semantics: every launch is (stream, byte regions
// read, byte regions written), launches of one stream are ordered, record -> wait adds an edge, and two launches that touch overlapping
// bytes (one of them writing) must be ordered.  Round 2 fed that checker a HAND-TRANSCRIBED launch list; this file lets the engines emit the
// list themselves: every launch_* entry point and every event call reports here when tracing is on, and tests/test_gpu_schedules.py feeds
// the dump to the same checker (tests/_sched_check.py).  Regions are (pointer, bytes[, stride, count]): the q / k / V^T scatter epilogues
// write the image rows and the text rows of one buffer from different streams, so a region may be a strided set of blocks.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "kernels.h"

namespace mi355 {

static bool g_trace_on = false;
static std::string g_trace_buf;

bool sched_trace_on() { return g_trace_on; }

static void put_regions(const char* tag, std::initializer_list<TraceRegion> rs) {
    char tmp[160];
    for (const TraceRegion& r : rs) {
        if (!r.p || r.len == 0 || r.count == 0) continue;
        snprintf(tmp, sizeof(tmp), " %s:%llx:%llu:%llu:%llu", tag, (unsigned long long)(uintptr_t)r.p, (unsigned long long)r.len,
                 (unsigned long long)(r.count > 1 ? r.stride : 0), (unsigned long long)r.count);
        g_trace_buf += tmp;
    }
}

void sched_trace_launch(const char* name, hipStream_t st, std::initializer_list<TraceRegion> reads, std::initializer_list<TraceRegion> writes) {
    if (!g_trace_on) return;
    char tmp[96];
    snprintf(tmp, sizeof(tmp), "L %llx %s", (unsigned long long)(uintptr_t)st, name);
    g_trace_buf += tmp;
    put_regions("R", reads);
    put_regions("W", writes);
    g_trace_buf += "\n";
}

void sched_trace_event(int kind, hipStream_t st, hipEvent_t ev) {
    if (!g_trace_on) return;
    char tmp[96];
    snprintf(tmp, sizeof(tmp), "%s %llx %llx\n", kind == 0 ? "E" : "T", (unsigned long long)(uintptr_t)st, (unsigned long long)(uintptr_t)ev);
    g_trace_buf += tmp;
}

}  // namespace mi355

extern "C" int mi355_sched_trace(int on) {
    mi355::g_trace_on = on != 0;
    if (on) mi355::g_trace_buf.clear();
    return 0;
}

// copies the trace text (NUL-terminated) into `out` if it fits `cap`; returns the number of bytes needed (including the NUL)
extern "C" long long mi355_sched_trace_read(char* out, long long cap) {
    const long long need = (long long)mi355::g_trace_buf.size() + 1;
    if (out && cap >= need) memcpy(out, mi355::g_trace_buf.c_str(), (size_t)need);
    return need;
}
