// Launch plans: a recorded list of the library's own launches (plus the event / stream-ordering calls and host callbacks
// between them) replayed from ONE C call.
//
// The reference replays a traced @tf.function (speech_jobs/whisper_dist.py:818-819: the first call traces, later calls
// replay the graph with no Python in between).  Here a training step is ~255 (Whisper) / ~340 (Wav2Vec2) launches of
// 5-100 us kernels; issuing each from Python costs 9-25 us of host time per launch (ctypes + descriptor marshalling), so
// the decoder-sized chains wait for the host.  hipGraph is not the answer on ROCm 7.2 (a replay of the ~430-node graph
// costs the host as much as the eager launches, tools/graph_check.py), so the plan is the library's own: while a plan is
// recording, every launching entry point appends a closure of ITSELF with its arguments copied by value (descriptors
// included) and then runs as usual, so the recorded step is a real step; a replay walks the closures - the same entry
// points, the same argument checks, the same dispatch rules, one hipLaunchKernel each, no Python.
//
// What changes from step to step is patched at replay time, not re-recorded:
//   - dropout seeds: every site seed is base + step * K + site * K' (blocks.KernelBlocks._site_seed), so a replay adds
//     `seed_delta` = (step_now - step_recorded) * K to every recorded seed (tmi_plan_seed_delta(), read by the closures);
//   - the Adam step number: `step_delta` is added to the recorded `step` argument of tmi_adam_step*.
// Events and cross-stream waits of the step (the weight-gradient stream, the early / late Adam slices) are recorded
// through tmi_plan_note_event_record / tmi_plan_note_stream_wait (the host code's own event objects: the handles stay
// valid because the host keeps them alive with the plan).  Anything else the step does on the host between launches
// (an RCCL collective issued through torch.distributed) is a callback node: the recorded function pointer is called in
// its place in the sequence.
#include "tmi_common.h"
#include <functional>
#include <vector>

struct tmi_plan {
  struct Node {
    int kind;  // 0 launch closure, 1 event record, 2 stream wait event, 3 host callback
    std::function<int()> fn;
    void* a;
    void* b;
  };
  std::vector<Node> nodes;
  int launches = 0;
};

namespace {
thread_local tmi_plan* g_rec = nullptr;   // the plan this thread is recording into
thread_local int g_depth = 0;             // > 0 inside an entry point: nested entry points are part of their caller's closure
thread_local uint64_t g_seed_delta = 0;
thread_local int64_t g_step_delta = 0;
}  // namespace

bool tmi_plan_recording() { return g_rec != nullptr && g_depth == 0; }
void tmi_plan_push(std::function<int()> fn) {
  g_rec->nodes.push_back(tmi_plan::Node{0, std::move(fn), nullptr, nullptr});
  ++g_rec->launches;
}
void tmi_plan_enter() { ++g_depth; }
void tmi_plan_leave() { --g_depth; }
uint64_t tmi_plan_seed_delta() { return g_seed_delta; }
int64_t tmi_plan_step_delta() { return g_step_delta; }

extern "C" int tmi_plan_create(tmi_plan** out) {
  if (!out) return TMI_ERR_INVALID;
  *out = new tmi_plan();
  return TMI_OK;
}

extern "C" int tmi_plan_destroy(tmi_plan* p) {
  if (p && p == g_rec) g_rec = nullptr;
  delete p;
  return TMI_OK;
}

extern "C" int tmi_plan_begin(tmi_plan* p) {
  if (!p || g_rec) {
    tmi_set_error("tmi_plan_begin: null plan, or this thread is already recording");
    return TMI_ERR_INVALID;
  }
  p->nodes.clear();
  p->launches = 0;
  g_rec = p;
  return TMI_OK;
}

extern "C" int tmi_plan_end(tmi_plan* p) {
  if (!p || g_rec != p) {
    tmi_set_error("tmi_plan_end: this plan is not recording on this thread");
    return TMI_ERR_INVALID;
  }
  g_rec = nullptr;
  return TMI_OK;
}

extern "C" int64_t tmi_plan_size(const tmi_plan* p, int32_t what) {
  if (!p) return -1;
  return what == 0 ? (int64_t)p->nodes.size() : (int64_t)p->launches;
}

extern "C" int tmi_plan_note_event_record(void* event, void* stream) {
  if (g_rec && g_depth == 0) g_rec->nodes.push_back(tmi_plan::Node{1, nullptr, event, stream});
  return TMI_OK;
}

extern "C" int tmi_plan_note_stream_wait(void* stream, void* event) {
  if (g_rec && g_depth == 0) g_rec->nodes.push_back(tmi_plan::Node{2, nullptr, stream, event});
  return TMI_OK;
}

extern "C" int tmi_plan_note_callback(void (*fn)(void)) {
  if (g_rec && g_depth == 0) g_rec->nodes.push_back(tmi_plan::Node{3, nullptr, reinterpret_cast<void*>(fn), nullptr});
  return TMI_OK;
}

extern "C" int tmi_plan_replay(tmi_plan* p, uint64_t seed_delta, int64_t step_delta) {
  if (!p || g_rec) {
    tmi_set_error("tmi_plan_replay: null plan, or this thread is recording");
    return TMI_ERR_INVALID;
  }
  g_seed_delta = seed_delta;
  g_step_delta = step_delta;
  int rc = TMI_OK;
  for (auto& n : p->nodes) {
    if (n.kind == 0) {
      rc = n.fn();
    } else if (n.kind == 1) {
      if (hipEventRecord(reinterpret_cast<hipEvent_t>(n.a), reinterpret_cast<hipStream_t>(n.b)) != hipSuccess) rc = TMI_ERR_LAUNCH;
    } else if (n.kind == 2) {
      if (hipStreamWaitEvent(reinterpret_cast<hipStream_t>(n.a), reinterpret_cast<hipEvent_t>(n.b), 0) != hipSuccess) rc = TMI_ERR_LAUNCH;
    } else {
      reinterpret_cast<void (*)(void)>(n.a)();
    }
    if (rc != TMI_OK) break;
  }
  g_seed_delta = 0;
  g_step_delta = 0;
  if (rc == TMI_ERR_LAUNCH) tmi_set_error("tmi_plan_replay: a recorded event / stream call failed");
  return rc;
}

// Fill / copy as entry points of their own, so that the handful of memsets and copies a step makes between its kernels
// are part of the plan (torch's fill_ / copy_ would run while recording and silently be missing from every replay).
extern "C" int tmi_memset_async(void* dst, int32_t value, int64_t bytes, void* stream) {
  if (!dst || bytes < 0) {
    tmi_set_error("tmi_memset_async: bad argument");
    return TMI_ERR_INVALID;
  }
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_memset_async(dst, value, bytes, stream); });
  if (bytes == 0) return TMI_OK;
  return hipMemsetAsync(dst, value, (size_t)bytes, reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? TMI_OK : TMI_ERR_LAUNCH;
}

extern "C" int tmi_memset2d_async(void* dst, int64_t pitch_bytes, int32_t value, int64_t width_bytes, int64_t rows, void* stream) {
  if (!dst || width_bytes < 0 || rows < 0 || pitch_bytes < width_bytes) {
    tmi_set_error("tmi_memset2d_async: bad argument");
    return TMI_ERR_INVALID;
  }
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_memset2d_async(dst, pitch_bytes, value, width_bytes, rows, stream); });
  if (width_bytes == 0 || rows == 0) return TMI_OK;
  return hipMemset2DAsync(dst, (size_t)pitch_bytes, value, (size_t)width_bytes, (size_t)rows, reinterpret_cast<hipStream_t>(stream)) == hipSuccess
             ? TMI_OK : TMI_ERR_LAUNCH;
}

extern "C" int tmi_memcpy_async(void* dst, const void* src, int64_t bytes, void* stream) {
  if (!dst || !src || bytes < 0) {
    tmi_set_error("tmi_memcpy_async: bad argument");
    return TMI_ERR_INVALID;
  }
  if (tmi_plan_recording()) tmi_plan_push([=]() -> int { return tmi_memcpy_async(dst, src, bytes, stream); });
  if (bytes == 0) return TMI_OK;
  return hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? TMI_OK
                                                                                                                              : TMI_ERR_LAUNCH;
}
