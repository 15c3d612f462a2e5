// Optional per-launch timing with HIP events on the launch stream (vl_profile_begin/_report).
// Off by default: a scope costs one pointer test.  Never active inside a graph capture.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

struct ProfRecord {
    std::string name;
    double flops, bytes;
    double exec_flops;      // FLOPs the matrix pipes are asked to execute (padded rows / tiles, the whole LoRA K tile); < 0: = flops
    hipEvent_t e0, e1;
};

struct Profiler {
    std::vector<ProfRecord> recs;
};

extern Profiler* g_prof;   // defined in vitlora.hip; non-null only between begin and report

// Test hook (vl_debug_set_option "poison_lds"): when non-zero, every launch that has a ProfScope is preceded by a kernel that
// fills the whole LDS of every CU with NaN patterns -- a kernel that reads LDS it did not write then shows as a changed result
// (tests/test_hip_engine.py).  LDS keeps what the previous kernel on the CU left there; round 3's 0 x NaN came from exactly that.
extern int g_poison_lds;
void vl_poison_lds(hipStream_t s);

struct ProfScope {
    hipStream_t s;
    hipEvent_t e1 = nullptr;
    ProfScope(const char* name, double flops, double bytes, hipStream_t stream, double exec_flops = -1.0) : s(stream) {
        if (g_poison_lds) vl_poison_lds(stream);
        if (!g_prof) return;
        ProfRecord r;
        r.name = name; r.flops = flops; r.bytes = bytes; r.exec_flops = exec_flops < 0.0 ? flops : exec_flops;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        (void)hipEventRecord(r.e0, s);
        e1 = r.e1;
        g_prof->recs.push_back(r);
    }
    ~ProfScope() {
        if (e1) (void)hipEventRecord(e1, s);
    }
};
