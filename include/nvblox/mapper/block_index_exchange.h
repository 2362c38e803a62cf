// nvblox/mapper/block_index_exchange.h -- libnvblox_hip extension (no reference counterpart: nvblox_ros serialises <= 4 cameras through one queue on
// one GPU, nvblox_node.hpp:298-332): the multi-GPU step BASELINE.json's north_star names -- "an RCCL all-gather over xGMI of updated block indices
// before the ESDF sweep" -- as a C++ host object, one per rank (= one GPU = one camera = one MultiMapper).  The C++ twin of
// isaac_ros_nvblox_amd/dist.py PipelinedDirtyBlockExchange, call for call; examples/rccl_index_exchange.cpp drives it, MultiMapper::setBlockIndexExchange
// hooks it into integrateDepth / integrateColor / updateEsdf so that a node's tick() does not change.
//
// Per frame i and rank:
//   beforeDepth   nvbx_set_view_export(slot i % 3): integrateDepth(i) itself writes the message -- int32 [1 + max_blocks][3], row 0 = {count, 0, 0},
//                 rows 1.. the Index3D of the blocks it updated -- from inside its TSDF-update launch (no export launch, the count never leaves the device)
//   start         ONE fixed-size all-gather of that buffer into [world][1 + max_blocks][3] (<= 48 KiB per rank at 4096 blocks; latency-bound on xGMI)
//   finishPrevious  the lists of frame i - 1 (gathered during a whole frame of GPU work) are handed to the mapper:
//                 nvbx_mark_esdf_dirty_gathered_deferred -- the union step rides in a launch of the pipelined frame (no launch of its own) -- or
//                 nvbx_mark_esdf_dirty_gathered (deferred = false: one launch now)
//   drain         joins the lists still on their way (before results are read / at shutdown)
// THREE buffer sets rotate: the set whose collective is in flight is never written, and neither is the set a mapper with colour deferral still reads
// (the union step of frame i's lists rides in integrateDepth(i + 2)'s fused launch).
//
// The collective itself is the caller's: `AllGather` is any callable that all-gathers n int32 from `send` into `recv` ([world][n]) in stream order on
// `stream` -- ncclAllGather(send, recv, n, ncclInt32, comm, stream) for RCCL (one process per GPU, or ncclGroupStart / ncclGroupEnd around the ranks of
// one process), a device-to-device copy for a stand-in peer.  The facade stays free of an RCCL dependency.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <utility>
#include "nvblox_hip.h"

namespace nvblox {

class BlockIndexExchange {
 public:
  using AllGather = std::function<int(const int32_t* send, int32_t* recv, size_t n_int32, hipStream_t stream)>;
  static constexpr int kSlots = 3;

  // comm_stream = nullptr: the collective is enqueued on the mapper's own stream (stream order does everything; what bench.py does with its one stream).
  // A stream of its own lets the collective overlap the mapper's launches; events order the two (one record + one wait each way per frame).
  BlockIndexExchange(int world, int rank, int64_t max_blocks, AllGather all_gather, hipStream_t comm_stream = nullptr)
      : world_(world), rank_(rank), max_blocks_(max_blocks), all_gather_(std::move(all_gather)), comm_stream_(comm_stream) {
    const size_t n = (size_t)(max_blocks_ + 1) * 3;
    for (int s = 0; s < kSlots; s++) {
      (void)hipMalloc((void**)&buf_[s], n * sizeof(int32_t)); (void)hipMemset(buf_[s], 0, n * sizeof(int32_t));
      (void)hipMalloc((void**)&all_[s], n * sizeof(int32_t) * (size_t)world_); (void)hipMemset(all_[s], 0, n * sizeof(int32_t) * (size_t)world_);
      if (comm_stream_) { (void)hipEventCreateWithFlags(&ready_[s], hipEventDisableTiming); (void)hipEventCreateWithFlags(&done_[s], hipEventDisableTiming); }
    }
  }
  ~BlockIndexExchange() {
    for (int s = 0; s < kSlots; s++) { if (buf_[s]) (void)hipFree(buf_[s]); if (all_[s]) (void)hipFree(all_[s]); if (ready_[s]) (void)hipEventDestroy(ready_[s]); if (done_[s]) (void)hipEventDestroy(done_[s]); }
  }
  BlockIndexExchange(const BlockIndexExchange&) = delete;
  BlockIndexExchange& operator=(const BlockIndexExchange&) = delete;

  int world() const { return world_; } int rank() const { return rank_; } int64_t max_blocks() const { return max_blocks_; }
  int64_t frames_started() const { return frame_; }
  const int32_t* gathered(int slot) const { return all_[slot]; }       // (tests)
  int32_t* message(int slot) const { return buf_[slot]; }

  void beforeDepth(nvbx_mapper* m) { checkRc(nvbx_set_view_export(m, buf_[frame_ % kSlots], max_blocks_), "nvbx_set_view_export"); }
  // A host whose depth rate exceeds its colour / ESDF rate (the reference node: 40 Hz depth, 5 Hz colour, 10 Hz ESDF, nvblox_base.yaml:13-23) calls
  // start() several times per finishPrevious().  No frame's lists may be dropped and no buffer set refilled before it was applied, so start() itself
  // advances the rotation when the previous frame's collective is still waiting for its finishPrevious (round 6): the lists gathered one
  // frame earlier are handed over (in the form the host last used), the previous frame's become pending.  Buffer set s is refilled by the collective
  // of frame i + 3, which is enqueued behind integrateDepth(i + 3); the union step of set s rides at the latest in integrateDepth(i + 2)'s launches.
  void start(nvbx_mapper* m) {
    if (started_ >= 0) advance(m, last_deferred_);
    const int s = (int)(frame_ % kSlots);
    hipStream_t ms = mapperStream(m);
    hipStream_t cs = comm_stream_ ? comm_stream_ : ms;
    if (comm_stream_) { (void)hipEventRecord(ready_[s], ms); (void)hipStreamWaitEvent(comm_stream_, ready_[s], 0); }      // the collective reads the message after the depth pass wrote it
    const size_t n = (size_t)(max_blocks_ + 1) * 3;
    if (all_gather_(buf_[s], all_[s], n, cs) != 0) { std::fprintf(stderr, "[nvblox_hip] BlockIndexExchange: all-gather failed\n"); std::abort(); }
    if (comm_stream_) (void)hipEventRecord(done_[s], comm_stream_);
    started_ = s; slot_frame_[s] = frame_; frame_++; finished_this_frame_ = false;
  }
  // between integrateDepth and updateEsdf of the current frame: hand over the PREVIOUS frame's lists; the current frame's stay in flight
  void finishPrevious(nvbx_mapper* m, bool deferred) {
    last_deferred_ = deferred;
    advance(m, deferred);
    finished_this_frame_ = true;
  }
  bool finishedThisFrame() const { return finished_this_frame_; }
  void drain(nvbx_mapper* m) {
    if (started_ >= 0) advance(m, false);
    if (pending_ >= 0) { apply(m, pending_, false); pending_ = -1; }
    checkRc(nvbx_set_view_export(m, nullptr, 0), "nvbx_set_view_export");
  }
  // (tests) how many frames' gathered lists have been handed to the mapper, and the frame number of the last one: after drain() every started frame
  // has been applied exactly once, in order
  int64_t frames_applied() const { return frames_applied_; }
  int64_t last_applied_frame() const { return last_applied_frame_; }
  bool applied_in_order() const { return applied_in_order_; }

 private:
  static void checkRc(int rc, const char* what) { if (rc < 0) { std::fprintf(stderr, "[nvblox_hip] %s failed (%d): %s\n", what, rc, nvbx_last_error()); std::abort(); } }
  static hipStream_t mapperStream(nvbx_mapper* m) { void* s = nullptr; checkRc(nvbx_get_stream(m, &s), "nvbx_get_stream"); return (hipStream_t)s; }
  void advance(nvbx_mapper* m, bool deferred) {
    if (pending_ >= 0) apply(m, pending_, deferred);
    pending_ = started_; started_ = -1;
  }
  void apply(nvbx_mapper* m, int slot, bool deferred) {
    if (slot_frame_[slot] != last_applied_frame_ + 1) applied_in_order_ = false;       // (a dropped frame, or a set refilled before it was applied)
    last_applied_frame_ = slot_frame_[slot]; frames_applied_++;
    if (comm_stream_) (void)hipStreamWaitEvent(mapperStream(m), done_[slot], 0);       // the union step reads the gathered lists after they have landed
    if (world_ < 2) return;
    if (deferred) checkRc(nvbx_mark_esdf_dirty_gathered_deferred(m, all_[slot], world_, rank_, max_blocks_), "nvbx_mark_esdf_dirty_gathered_deferred");
    else checkRc(nvbx_mark_esdf_dirty_gathered(m, all_[slot], world_, rank_, max_blocks_), "nvbx_mark_esdf_dirty_gathered");
  }
  int world_, rank_; int64_t max_blocks_;
  AllGather all_gather_;
  hipStream_t comm_stream_;
  int32_t* buf_[kSlots] = {}; int32_t* all_[kSlots] = {};
  hipEvent_t ready_[kSlots] = {}, done_[kSlots] = {};
  int64_t frame_ = 0;
  int64_t slot_frame_[kSlots] = {-1, -1, -1}, frames_applied_ = 0, last_applied_frame_ = -1;
  int pending_ = -1, started_ = -1;
  bool finished_this_frame_ = true, last_deferred_ = true, applied_in_order_ = true;
};

}  // namespace nvblox
