// Riders: host-side bookkeeping of launches that are recorded instead of issued and later carried as extra workgroups of
// another launch of the same kernel (common.h; include/grapes_hip.h: grapes_rider_*).  No device code here.
//
// A PROGRAM is the ordered list of launches one recording produced (the prelude of a training step: next batch, hop 0's
// expansion, compaction, graph build and gather-SpMM).  Attached, its records are consumed in order: a record of kind OTHER is
// issued on its own as soon as it is at the head of the list (at attach time, or in front of the next host launch); a pairable
// record waits for a host launch of its kind, variant and workgroup size, which then launches the two-problem form of the
// kernel over both argument sets; detach issues whatever is left, in order.  The order of the chain is therefore the recorded
// one whatever rides and whatever does not, and every record is issued exactly once per attach.
#include "common.h"
#include <vector>
#include <memory>
#include <stdlib.h>

namespace {
struct Program { std::vector<GrapesRiderRecord> recs; };
struct RiderState {
    std::vector<std::unique_ptr<Program>> programs;
    Program* recording = nullptr;
    int32_t recording_slot = -1;
    Program* attached = nullptr;
    size_t next = 0;
    bool hold = false;               // attached in the early phase: only a BEGIN record may be taken (grapes_rider_release lifts it)
    int paired = 0, alone = 0;       // statistics of the last attach
} g_rd;

void flush_others(hipStream_t s) {
    while (g_rd.attached && g_rd.next < g_rd.attached->recs.size() && g_rd.attached->recs[g_rd.next].kind == GRAPES_RK_OTHER) {
        g_rd.attached->recs[g_rd.next].single(s);
        ++g_rd.next; ++g_rd.alone;
    }
}
}  // namespace

bool grapes_rider_recording() { return g_rd.recording != nullptr; }
int grapes_rider_grid(int grid) {
    static int cap = -1;
    if (cap < 0) { const char* e = grapes_tune_env("GRAPES_RIDER_GRID"); cap = e ? atoi(e) : 256; if (cap < 0) cap = 0; }
    if (!g_rd.recording || cap == 0) return grid;
    return grid < cap ? grid : cap;
}
void grapes_rider_record(GrapesRiderRecord&& r) { if (g_rd.recording) g_rd.recording->recs.push_back(std::move(r)); }
const GrapesRiderRecord* grapes_rider_match(int kind, int variant, int block, hipStream_t s) {
    if (!g_rd.attached) return nullptr;
    if (g_rd.hold && kind != GRAPES_RK_BEGIN) return nullptr;
    if (!g_rd.hold) flush_others(s);
    if (g_rd.next >= g_rd.attached->recs.size()) return nullptr;
    const GrapesRiderRecord& r = g_rd.attached->recs[g_rd.next];
    if (r.kind != kind || (variant != GRAPES_RIDER_ANY_VARIANT && r.variant != variant) || (block != 0 && r.block != block)) return nullptr;     // (block 0: any — the body is block-size agnostic)
    ++g_rd.next; ++g_rd.paired;
    return &r;
}

extern "C" int grapes_rider_record_begin(void) {
    if (g_rd.recording || g_rd.attached) return GRAPES_EINVAL;
    size_t slot = 0;                                         // a freed program's slot is taken again (an eager loop records every step)
    while (slot < g_rd.programs.size() && g_rd.programs[slot]) ++slot;
    if (slot == g_rd.programs.size()) g_rd.programs.emplace_back(nullptr);
    g_rd.programs[slot].reset(new Program());
    g_rd.recording = g_rd.programs[slot].get();
    g_rd.recording_slot = (int32_t)slot;
    return 0;
}
extern "C" int32_t grapes_rider_record_end(void) {
    if (!g_rd.recording) return -1;
    g_rd.recording = nullptr;
    return g_rd.recording_slot;
}
extern "C" int32_t grapes_rider_count(int32_t program) {
    if (program < 0 || program >= (int32_t)g_rd.programs.size() || !g_rd.programs[program]) return -1;
    return (int32_t)g_rd.programs[program]->recs.size();
}
extern "C" int grapes_rider_attach(int32_t program, int32_t hold, grapes_stream_t stream) {
    if (g_rd.recording || g_rd.attached || program < 0 || program >= (int32_t)g_rd.programs.size() || !g_rd.programs[program])
        return GRAPES_EINVAL;
    g_rd.attached = g_rd.programs[program].get();
    g_rd.next = 0; g_rd.paired = 0; g_rd.alone = 0;
    g_rd.hold = hold != 0;
    if (!g_rd.hold) flush_others((hipStream_t)stream);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
// end of the early phase: a BEGIN record nobody carried is issued now, then the records that cannot ride
extern "C" int grapes_rider_release(grapes_stream_t stream) {
    if (!g_rd.attached) return GRAPES_EINVAL;
    g_rd.hold = false;
    if (g_rd.next < g_rd.attached->recs.size() && g_rd.attached->recs[g_rd.next].kind == GRAPES_RK_BEGIN) {
        g_rd.attached->recs[g_rd.next].single((hipStream_t)stream);
        ++g_rd.next; ++g_rd.alone;
    }
    flush_others((hipStream_t)stream);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
// -> number of records that were issued on their own (not carried by a host launch), or a negative error
extern "C" int grapes_rider_detach(grapes_stream_t stream, int32_t* paired) {
    if (!g_rd.attached) return GRAPES_EINVAL;
    while (g_rd.next < g_rd.attached->recs.size()) {
        g_rd.attached->recs[g_rd.next].single((hipStream_t)stream);
        ++g_rd.next; ++g_rd.alone;
    }
    g_rd.attached = nullptr;
    g_rd.hold = false;
    if (paired) *paired = g_rd.paired;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e;
    return g_rd.alone;
}
// the whole program on its own, in order (the first pipelined step has no predecessor to ride in)
extern "C" int grapes_rider_launch(int32_t program, grapes_stream_t stream) {
    if (g_rd.recording || program < 0 || program >= (int32_t)g_rd.programs.size() || !g_rd.programs[program]) return GRAPES_EINVAL;
    for (auto& r : g_rd.programs[program]->recs) r.single((hipStream_t)stream);
    GRAPES_LAUNCH_CHECK();
    return 0;
}
extern "C" int grapes_rider_free(int32_t program) {
    if (program < 0 || program >= (int32_t)g_rd.programs.size() || !g_rd.programs[program]) return GRAPES_EINVAL;
    if (g_rd.attached == g_rd.programs[program].get() || g_rd.recording == g_rd.programs[program].get()) return GRAPES_EINVAL;
    g_rd.programs[program].reset();
    return 0;
}
