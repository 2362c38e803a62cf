// frames.hip -- library-owned, reference-counted device frames: OWNERSHIP TRANSFER of input images (host code only).
//
// Why: colour deferral holds an integrateColor back until the next integrateDepth.  The reference's colour image is ONE node-owned
// device buffer, refilled by a conversion on the mapper's stream right before every integrateColor (nvblox_ros/src/lib/nvblox_node.cpp:1237-1263,
// buffers nvblox_node.hpp:484-488, converter conversions/image_conversions_thrust.cu:68-84) -- a held-back frame would be overwritten.  Round 4
// answered with a 1.8 MB copy per frame (k_stage_color, 18 % of the frame).  This file removes the copy: the image lives in a frame the LIBRARY owns;
// the mapper RETAINS it while the frame is held back and lets go of it once the launches that read it are enqueued; the writer (nvblox::Image<T> in
// include/nvblox/sensors/image.h, or any C caller) asks for "a frame nobody else holds" before it writes and gets a different one while the mapper
// still holds the last -- rotation instead of copying.
//
// Safety is by construction, not by contract: a frame a mapper has let go of carries a FENCE {progress word in pinned host memory, sequence number}.
// The mapper's next view-marking launch reports, as its first action, how many colour-reading launches were enqueued before it; a kernel starts only
// after everything enqueued before it on its stream has finished, so progress >= seq proves the readers are done.  nvbx_frame_acquire hands a
// fenced frame to a writer only if (a) its fences have been reached, or (b) the writer says it writes on the reader's own stream (stream order),
// else it takes another frame, grows the pool (up to NVBX_FRAME_POOL_MAX frames per size, default 8), or waits for the oldest fence
// (back-pressure on a host that runs more than a pool ahead of the GPU; no queue drain -- it polls the progress word).
//
// Frames let go of WITHOUT a mapper's fence (round 6): nvbx_frame_release -- an nvblox::Image<T> that is destroyed, resized or rotated --
// used to return the frame to the pool at once, where hipFree (what the reference's buffers end in) waits for the device.  Work the library itself
// enqueued on the image (a depth frame of integrateDepth, a mask, a slice image) or the caller's own asynchronous copies could still be reading or
// writing it when the next nvbx_frame_acquire handed the memory out.  Now the last reference let go of records an EVENT on every stream the library
// knows on that device (the streams of live mappers -- in nvblox_ros the node's one cuda_stream_, on which its conversions run too -- and the legacy
// default stream) -- or on the one stream nvbx_frame_release_on names -- and the frame is handed out again only once those events have been reached
// (or to a writer on the very stream an event was recorded on).  Nothing waits: a frame that is not cool yet is skipped like one a mapper's launches
// still read.  (Streams the holder merely NAMED earlier -- a copyFromAsync -- are not recorded on: they may be gone by the time the image dies.)
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>
#include "nvbx_mapper.h"

namespace nvbx {

// What a fence of a mapper points into (its pinned progress word, its enqueue counter, its stream) dies with the mapper.  The block below is shared by
// the mapper's registry entry and every fence of it: nvbx_mapper_destroy marks it dead under g_mu and then waits until nobody is inside a wait that
// dereferences those pointers (a back-pressure wait on another thread used to read them after the mapper was gone).
struct OwnerState { bool dead = false; int waiters = 0; };
struct FrameFence { const volatile int32_t* progress; int32_t seq; const volatile int32_t* reports_enqueued; hipStream_t reader; const void* owner; std::shared_ptr<OwnerState> st; };
struct EventFence { hipEvent_t ev; hipStream_t stream; };
struct PoolFrame {
  void* ptr = nullptr; size_t bytes = 0; int device = 0; int32_t refs = 0; uint64_t freed_at = 0;
  std::vector<FrameFence> fences;
  std::vector<EventFence> events;          // recorded when the last reference was let go of without a mapper's fence
};
struct KnownStream { hipStream_t stream; int device; int users; };
static std::mutex g_mu;
static std::condition_variable g_cv;
static std::vector<PoolFrame*> g_frames;
static std::unordered_map<const void*, PoolFrame*> g_by_ptr;
static std::unordered_map<const void*, std::shared_ptr<OwnerState>> g_owners;
static std::vector<KnownStream> g_streams;                                   // streams of live mappers
static std::unordered_map<int, std::vector<hipEvent_t>> g_event_pool;        // per device
static uint64_t g_tick = 0;
static int64_t g_stat_created = 0, g_stat_waits = 0, g_stat_syncs = 0;

// the calling thread's current device is the caller's business: whatever the pool does on another device is undone on the way out
struct DeviceGuard {
  int prev = -1; bool switched = false;
  explicit DeviceGuard(int device) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != device) { switched = hipSetDevice(device) == hipSuccess; } }
  ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
};

static int pool_max() { const char* e = getenv("NVBX_FRAME_POOL_MAX"); return e ? std::max(2, atoi(e)) : 8; }      // (read where a frame would be created: rare)
// caller holds g_mu (a dead owner's words are not read)
static bool fence_reached(const FrameFence& f) { return f.st->dead || (int32_t)(__atomic_load_n(f.progress, __ATOMIC_ACQUIRE) - f.seq) >= 0; }
static void recycle_event(int device, hipEvent_t ev) { g_event_pool[device].push_back(ev); }
// drop the fences and events that have been reached; true = none left (any writer may have the frame).  Caller holds g_mu.
static bool cooled(PoolFrame* f) {
  f->fences.erase(std::remove_if(f->fences.begin(), f->fences.end(), [](const FrameFence& x) { return fence_reached(x); }), f->fences.end());
  if (!f->events.empty()) {
    f->events.erase(std::remove_if(f->events.begin(), f->events.end(), [f](const EventFence& x) {
                      if (hipEventQuery(x.ev) == hipErrorNotReady) return false;        // (any other answer: reached, or an error we cannot wait out)
                      recycle_event(f->device, x.ev); return true; }), f->events.end());
  }
  return f->fences.empty() && f->events.empty();
}
static bool fits(const PoolFrame* f, int device, size_t bytes) { return f->device == device && f->bytes >= bytes && f->bytes <= std::max(bytes + (64u << 10), bytes + bytes / 4); }

// caller holds g_mu
static PoolFrame* find_frame(const void* p) { auto it = g_by_ptr.find(p); return it == g_by_ptr.end() ? nullptr : it->second; }
// the last reference goes without a mapper's fence: one event per stream that may carry work on the frame.  Caller holds g_mu.
static void fence_by_events(PoolFrame* f, void* last_stream) {
  std::vector<hipStream_t> on;
  if (last_stream != NVBX_STREAM_UNKNOWN) on.push_back((hipStream_t)last_stream);
  else {
    on.push_back(nullptr);                                                  // the legacy default stream (kernels of a host without streams)
    for (const KnownStream& k : g_streams) if (k.device == f->device && std::find(on.begin(), on.end(), k.stream) == on.end()) on.push_back(k.stream);
  }
  DeviceGuard dg(f->device);
  for (hipStream_t s : on) {
    hipEvent_t ev = nullptr;
    std::vector<hipEvent_t>& pool = g_event_pool[f->device];
    if (!pool.empty()) { ev = pool.back(); pool.pop_back(); }
    else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(s); continue; }      // (no event: wait now)
    if (hipEventRecord(ev, s) != hipSuccess) { (void)hipGetLastError(); recycle_event(f->device, ev); continue; }      // (a stream that is gone carries no work)
    f->events.push_back(EventFence{ev, s});
  }
}
static void drop_ref(PoolFrame* f, void* last_stream, bool event_fence) {
  if (--f->refs != 0) return;
  if (event_fence) fence_by_events(f, last_stream);
  f->freed_at = ++g_tick;
}
static void free_frame_memory(PoolFrame* f) {
  DeviceGuard dg(f->device);
  (void)hipFree(f->ptr);                    // (hipFree waits for the device: in-flight readers are safe)
  for (const EventFence& x : f->events) recycle_event(f->device, x.ev);
}

bool frame_retain_if_frame(const void* p, size_t need_bytes, bool* too_small) {
  std::lock_guard<std::mutex> lk(g_mu);
  PoolFrame* f = find_frame(p);
  if (too_small) *too_small = false;
  if (!f || f->refs <= 0) return false;
  if (f->bytes < need_bytes) { if (too_small) *too_small = true; return false; }
  f->refs++;
  return true;
}
// the mapper lets go of a frame whose readers are ENQUEUED on `reader` (not necessarily finished): the fence says when they are
void frame_release_fenced(void* p, const volatile int32_t* progress, int32_t seq, const volatile int32_t* reports_enqueued, hipStream_t reader, const void* owner) {
  std::lock_guard<std::mutex> lk(g_mu);
  PoolFrame* f = find_frame(p);
  if (!f || f->refs <= 0) return;
  (void)cooled(f);
  std::shared_ptr<OwnerState>& st = g_owners[owner];
  if (!st) st = std::make_shared<OwnerState>();
  f->fences.push_back(FrameFence{progress, seq, reports_enqueued, reader, owner, st});
  drop_ref(f, NVBX_STREAM_UNKNOWN, false);        // (the other holders' work is theirs to fence: an Image that lets go later records its events then)
}
// a mapper is born: frames let go of without a fence also wait for what is enqueued on its stream (two mappers may share one stream)
void frames_register_stream(int device, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (KnownStream& k : g_streams) if (k.stream == stream && k.device == device) { k.users++; return; }
  g_streams.push_back(KnownStream{stream, device, 1});
}
// a mapper goes away (its stream has been synchronised): its fences are reached by definition, and its progress words must not be read again --
// nor its stream waited on: a wait in progress on another thread is let out first
void frames_forget_owner(const void* owner, int device, hipStream_t stream) {
  std::unique_lock<std::mutex> lk(g_mu);
  auto it = g_owners.find(owner);
  if (it != g_owners.end()) {
    std::shared_ptr<OwnerState> st = it->second;
    st->dead = true;
    g_cv.wait(lk, [&] { return st->waiters == 0; });
    g_owners.erase(owner);
  }
  for (PoolFrame* f : g_frames)
    f->fences.erase(std::remove_if(f->fences.begin(), f->fences.end(), [owner](const FrameFence& x) { return x.owner == owner; }), f->fences.end());
  for (size_t i = 0; i < g_streams.size(); i++)
    if (g_streams[i].stream == stream && g_streams[i].device == device) {
      if (--g_streams[i].users <= 0) {
        // events recorded on the stream have been reached (it is idle); the frames' own lists drop them at their next look -- but a stream the caller
        // destroys must not be named by a later "same stream" shortcut: clear the name
        for (PoolFrame* f : g_frames) for (EventFence& x : f->events) if (x.stream == stream) x.stream = (hipStream_t)NVBX_STREAM_UNKNOWN;
        g_streams.erase(g_streams.begin() + (long)i);
      }
      break;
    }
}

}  // namespace nvbx
using namespace nvbx;

extern "C" int nvbx_frame_acquire(int device, size_t bytes, void* writer_stream, void** dev_ptr_out) {
  if (!dev_ptr_out || bytes == 0 || bytes > ((size_t)1 << 36)) { set_error("nvbx_frame_acquire: invalid argument"); return NVBX_E_INVALID; }
  *dev_ptr_out = nullptr;
  const bool writer_known = writer_stream != NVBX_STREAM_UNKNOWN;
  for (int attempt = 0; attempt < 8; attempt++) {
    FrameFence wait_for{}; bool have_wait = false; bool have_ev = false; PoolFrame* wait_frame = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_mu);
      PoolFrame* best = nullptr; int in_class = 0; PoolFrame* oldest = nullptr;
      for (PoolFrame* f : g_frames) {
        if (!fits(f, device, bytes)) continue;
        in_class++;
        if (f->refs != 0) continue;
        bool ok = cooled(f);
        if (!ok && writer_known) {                  // (stream order: everything that still uses the frame was enqueued on the writer's own stream)
          ok = true;
          for (const FrameFence& x : f->fences) ok = ok && x.reader == (hipStream_t)writer_stream;
          for (const EventFence& x : f->events) ok = ok && x.stream == (hipStream_t)writer_stream;
        }
        if (ok) { if (!best || f->freed_at < best->freed_at) best = f; }
        else if (!oldest || f->freed_at < oldest->freed_at) oldest = f;
      }
      if (best) { best->refs = 1; *dev_ptr_out = best->ptr; return NVBX_OK; }
      if (in_class < pool_max() || !oldest) {
        // a new frame; frames of OTHER sizes that nobody has asked for lately are given back first (a host whose image size changes must not pile frames up)
        int idle = 0; for (PoolFrame* f : g_frames) if (f->refs == 0 && !fits(f, device, bytes)) idle++;
        if (idle > 2 * pool_max()) {
          for (size_t i = 0; i < g_frames.size();) {
            PoolFrame* f = g_frames[i];
            if (f->refs == 0 && !fits(f, device, bytes) && cooled(f)) { free_frame_memory(f); g_by_ptr.erase(f->ptr); delete f; g_frames.erase(g_frames.begin() + (long)i); }
            else i++;
          }
        }
        DeviceGuard dg(device);
        void* p = nullptr;
        const size_t alloc = (bytes + 255) & ~(size_t)255;
        if (hipMalloc(&p, alloc) != hipSuccess) { (void)hipGetLastError(); set_error("nvbx_frame_acquire: hipMalloc"); return NVBX_E_DEVICE; }
        PoolFrame* f = new PoolFrame(); f->ptr = p; f->bytes = alloc; f->device = device; f->refs = 1;
        g_frames.push_back(f); g_by_ptr[p] = f; g_stat_created++;
        *dev_ptr_out = p; return NVBX_OK;
      }
      // the pool of this size is full and every free frame still has work in flight: wait for the one that was let go of first
      for (const FrameFence& x : oldest->fences) if (!fence_reached(x)) { wait_for = x; have_wait = true; wait_for.st->waiters++; break; }
      if (!have_wait && !oldest->events.empty()) { have_ev = true; wait_frame = oldest; }
      if (have_wait || have_ev) g_stat_waits++;
    }
    if (have_ev) {
      // events of the library's own: poll them under the lock (the frame keeps them until they are reached)
      for (;;) {
        { std::lock_guard<std::mutex> lk(g_mu);
          bool pending = false;
          for (PoolFrame* f : g_frames) if (f == wait_frame && f->refs == 0) for (const EventFence& x : f->events) pending = pending || hipEventQuery(x.ev) == hipErrorNotReady;
          if (!pending) break; }
        std::this_thread::yield();
      }
      continue;
    }
    if (!have_wait) continue;
    // a mapper's fence.  Its readers' completion will be reported only if a later view-marking launch is already enqueued; otherwise wait for the stream
    // itself.  The mapper cannot finish dying while `waiters` counts this thread (frames_forget_owner), so its words and its stream stay valid here; once it
    // is marked dead it has synchronised its stream itself and the wait is over.
    bool reached = false, dead = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      { std::lock_guard<std::mutex> lk(g_mu);
        dead = wait_for.st->dead;
        reached = dead || fence_reached(wait_for);
        if (reached || (int32_t)(__atomic_load_n(wait_for.reports_enqueued, __ATOMIC_ACQUIRE) - wait_for.seq) < 0) break; }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
      std::this_thread::yield();
    }
    hipError_t sync_rc = hipSuccess;
    if (!reached) sync_rc = hipStreamSynchronize(wait_for.reader);
    {
      std::lock_guard<std::mutex> lk(g_mu);
      if (!reached) {
        g_stat_syncs++;
        if (sync_rc == hipSuccess)         // everything enqueued on that stream has finished: its fences up to here are reached
          for (PoolFrame* f : g_frames)
            f->fences.erase(std::remove_if(f->fences.begin(), f->fences.end(), [&](const FrameFence& x) { return x.reader == wait_for.reader && x.owner == wait_for.owner && (int32_t)(wait_for.seq - x.seq) >= 0; }), f->fences.end());
      }
      wait_for.st->waiters--;
    }
    g_cv.notify_all();
    if (sync_rc != hipSuccess) { set_error("nvbx_frame_acquire: hipStreamSynchronize", sync_rc); return NVBX_E_DEVICE; }
  }
  set_error("nvbx_frame_acquire: no frame became free");
  return NVBX_E_DEVICE;
}

extern "C" int nvbx_frame_retain(void* dev_ptr) {
  std::lock_guard<std::mutex> lk(g_mu);
  PoolFrame* f = find_frame(dev_ptr);
  if (!f || f->refs <= 0) { set_error("nvbx_frame_retain: not a live frame of nvbx_frame_acquire"); return NVBX_E_INVALID; }
  f->refs++;
  return NVBX_OK;
}
extern "C" int nvbx_frame_release_on(void* dev_ptr, void* last_stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  PoolFrame* f = find_frame(dev_ptr);
  if (!f || f->refs <= 0) { set_error("nvbx_frame_release: not a live frame of nvbx_frame_acquire"); return NVBX_E_INVALID; }
  drop_ref(f, last_stream, true);
  return NVBX_OK;
}
extern "C" int nvbx_frame_release(void* dev_ptr) { return nvbx_frame_release_on(dev_ptr, NVBX_STREAM_UNKNOWN); }
extern "C" int32_t nvbx_frame_refcount(const void* dev_ptr) {
  std::lock_guard<std::mutex> lk(g_mu);
  PoolFrame* f = find_frame(dev_ptr);
  return f ? f->refs : -1;
}
extern "C" int32_t nvbx_frame_device(const void* dev_ptr) {
  std::lock_guard<std::mutex> lk(g_mu);
  PoolFrame* f = find_frame(dev_ptr);
  return f ? f->device : -1;
}
// may the holder write the frame NOW, on `writer_stream`?  1 = yes: it is the only holder and no launch of a mapper can still be reading it (a mapper
// that held the image back has let go of it AND those launches have finished, or were enqueued on the writer's own stream); 0 = no: continue in
// another frame (nvbx_frame_acquire) and let go of this one; -1 = not a live frame
extern "C" int32_t nvbx_frame_writable(const void* dev_ptr, void* writer_stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  PoolFrame* f = find_frame(dev_ptr);
  if (!f || f->refs <= 0) return -1;
  if (f->refs > 1) return 0;
  if (cooled(f)) return 1;
  if (writer_stream == NVBX_STREAM_UNKNOWN) return 0;
  for (const FrameFence& x : f->fences) if (x.reader != (hipStream_t)writer_stream) return 0;
  for (const EventFence& x : f->events) if (x.stream != (hipStream_t)writer_stream) return 0;
  return 1;
}
extern "C" int nvbx_frame_pool_trim(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  for (size_t i = 0; i < g_frames.size();) {
    PoolFrame* f = g_frames[i];
    if (f->refs == 0 && (device < 0 || f->device == device)) {
      free_frame_memory(f);
      g_by_ptr.erase(f->ptr); delete f; g_frames.erase(g_frames.begin() + (long)i); n++;
    } else i++;
  }
  return n;
}
extern "C" int nvbx_frame_pool_stats(int64_t out[6]) {
  if (!out) return NVBX_E_INVALID;
  std::lock_guard<std::mutex> lk(g_mu);
  int64_t live = 0, freec = 0, bytes = 0;
  for (PoolFrame* f : g_frames) { if (f->refs > 0) live++; else freec++; bytes += (int64_t)f->bytes; }
  out[0] = live; out[1] = freec; out[2] = bytes; out[3] = g_stat_created; out[4] = g_stat_waits; out[5] = g_stat_syncs;
  return NVBX_OK;
}
extern "C" int nvbx_frame_upload(void* dev_ptr, const void* src, size_t bytes, void* hip_stream) {
  if (!dev_ptr || !src) return NVBX_E_INVALID;
  int device = 0;
  { std::lock_guard<std::mutex> lk(g_mu); PoolFrame* f = find_frame(dev_ptr); if (!f || f->refs <= 0 || f->bytes < bytes) { set_error("nvbx_frame_upload: not a live frame / too small"); return NVBX_E_INVALID; } device = f->device; }
  DeviceGuard dg(device);
  if (hip_stream == NVBX_STREAM_UNKNOWN) NVBX_HIP(hipMemcpy(dev_ptr, src, bytes, hipMemcpyDefault));       // (blocking)
  else NVBX_HIP(hipMemcpyAsync(dev_ptr, src, bytes, hipMemcpyDefault, (hipStream_t)hip_stream));
  return NVBX_OK;
}

extern "C" int nvbx_color_image_acquire(nvbx_mapper* m, int32_t rows, int32_t cols, int32_t bytes_per_pixel, void** dev_ptr_out) {
  if (!m || rows < 1 || cols < 1 || (bytes_per_pixel != 3 && bytes_per_pixel != 4)) { set_error("nvbx_color_image_acquire: invalid argument (3 = rgb8, 4 = bgra8)"); return NVBX_E_INVALID; }
  return nvbx_frame_acquire(m->device, (size_t)rows * (size_t)cols * (size_t)bytes_per_pixel, m->stream, dev_ptr_out);
}
