// round6_checks.cpp -- two host-side contracts of the facade that round 5's review found open, through the facade only:
//
//   lifetime <frames.bin>   A temporary nvblox::DepthImage / ColorImage handed to integrateDepth / integrateColor and destroyed right after the call --
//                           what `Image<float> img(MemoryType::kDevice)` at nvblox_node.cpp:835 and the fuser's per-frame images do -- while the
//                           mapper's stream is still busy with work enqueued BEFORE the call.  The reference's buffers end in cudaFree, which waits
//                           for the device; here the frame goes back to the library's pool and must not be handed to the next image before the
//                           launches that read it have finished.  Checks: (1) while the stream is busy, the next image of the same size gets OTHER
//                           memory; (2) the map equals a mapper fed long-lived images, bit for bit; (3) once the stream is idle the frame is reused.
//
//   cadence <frames.bin>    nvblox::BlockIndexExchange under the reference node's call cadence (nvblox_base.yaml:13-23: 40 Hz depth, 5 Hz colour,
//                           10 Hz ESDF): several integrateDepth per integrateColor / updateEsdf, plain and dynamic mapping types.  Every frame's
//                           gathered lists must reach the mapper exactly once, in order, no buffer set refilled before it was applied; and rank 0's
//                           map equals a mapper without any exchange.
//
// frames.bin as for image_frames: int32 {n, rows, cols}, float {fu fv cu cv}, then per frame T[16] f32, depth f32, rgb u8.
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <thread>
#include <string>
#include <vector>
#include "nvblox/nvblox.h"

using namespace nvblox;

struct Fr { float T[16]; std::vector<float> d; std::vector<Color> c; };

static bool layers_equal(Mapper& a, Mapper& b) {
  const uint32_t layers[3] = {NVBX_LAYER_TSDF, NVBX_LAYER_COLOR, NVBX_LAYER_ESDF}; const size_t vox_bytes[3] = {8, 8, 20};
  for (int l = 0; l < 3; l++) {
    const int64_t na = nvbx_num_blocks(a.c_handle(), layers[l]), nb = nvbx_num_blocks(b.c_handle(), layers[l]);
    if (na != nb) { std::printf("layer %d: %lld vs %lld blocks\n", l, (long long)na, (long long)nb); return false; }
    std::vector<nvbx_index3d> ia((size_t)na), ib((size_t)na);
    nvbx_block_indices(a.c_handle(), layers[l], ia.data(), na); nvbx_block_indices(b.c_handle(), layers[l], ib.data(), na);
    if (na && std::memcmp(ia.data(), ib.data(), sizeof(nvbx_index3d) * (size_t)na)) { std::printf("layer %d: index sets differ\n", l); return false; }
    std::vector<uint8_t> va((size_t)na * 512 * vox_bytes[l]), vb(va.size());
    if (na) { nvbx_get_blocks(a.c_handle(), layers[l], ia.data(), na, va.data(), nullptr); nvbx_get_blocks(b.c_handle(), layers[l], ia.data(), na, vb.data(), nullptr); }
    if (va != vb) { std::printf("layer %d: voxels differ\n", l); return false; }
  }
  return true;
}

// keeps `stream` busy for a few milliseconds with work the library knows nothing about (device memsets of a large buffer)
struct Busy {
  void* big = nullptr; size_t bytes = (size_t)1 << 30;
  Busy() { if (hipMalloc(&big, bytes) != hipSuccess) { bytes = (size_t)1 << 28; (void)hipMalloc(&big, bytes); } }
  ~Busy() { if (big) (void)hipFree(big); }
  void enqueue(hipStream_t s, int reps) const { for (int i = 0; i < reps; i++) (void)hipMemsetAsync(big, i & 255, bytes, s); }
};

static int lifetime(const std::vector<Fr>& fr, int rows, int cols, const Camera& camera) {
  auto stream = std::make_shared<CudaStreamOwning>();
  auto stream_ref = std::make_shared<CudaStreamOwning>();
  Mapper temp(0.05f, MemoryType::kDevice, ProjectiveLayerType::kTsdf, stream, 1 << 13), ref(0.05f, MemoryType::kDevice, ProjectiveLayerType::kTsdf, stream_ref, 1 << 13);
  Busy busy;
  // the reference mapper: long-lived images, nothing destroyed while anything is in flight
  std::vector<DepthImage> keep_d; std::vector<ColorImage> keep_c;
  for (const auto& x : fr) {
    keep_d.emplace_back(MemoryType::kDevice); keep_d.back().copyFromAsync(rows, cols, x.d.data(), *stream_ref);
    keep_c.emplace_back(MemoryType::kDevice); keep_c.back().copyFromAsync(rows, cols, x.c.data(), *stream_ref);
  }
  (void)hipStreamSynchronize(*stream_ref);
  for (size_t i = 0; i < fr.size(); i++) { const Transform T = Transform::fromRowMajor(fr[i].T); ref.integrateDepth(keep_d[i], T, camera); ref.integrateColor(keep_c[i], T, camera); ref.updateEsdf(); }
  ref.synchronize();

  int handed_out_while_busy = 0, busy_frames = 0; const void* first_ptr = nullptr;
  std::vector<float> garbage((size_t)rows * cols, 0.123f);
  for (size_t i = 0; i < fr.size(); i++) {
    const Transform T = Transform::fromRowMajor(fr[i].T);
    const void* p_depth = nullptr;
    {
      DepthImage tmp(MemoryType::kDevice);                       // a temporary, as at nvblox_node.cpp:835
      tmp.copyFromAsync(rows, cols, fr[i].d.data(), *stream);
      (void)hipStreamSynchronize(*stream);                        // (pageable source: the bytes are there)
      busy.enqueue(*stream, 12);                                  // the stream is busy with earlier work ...
      temp.integrateDepth(tmp, T, camera);                        // ... the launches that read `tmp` queue up behind it
      p_depth = tmp.dataConstPtr();
      if (!first_ptr) first_ptr = p_depth;
    }                                                             // destroyed with its readers still waiting
    const bool still_busy = hipStreamQuery(*stream) == hipErrorNotReady;
    {
      // the next image of the same size, written by a BLOCKING host copy (a writer the mapper's stream order does not cover)
      DepthImage next(rows, cols, MemoryType::kDevice);
      if (still_busy) { busy_frames++; if (next.dataConstPtr() == p_depth) handed_out_while_busy++; }
      (void)hipMemcpy(next.dataPtr(), garbage.data(), garbage.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    {
      ColorImage tmpc(MemoryType::kDevice);
      tmpc.copyFromAsync(rows, cols, fr[i].c.data(), *stream);
      (void)hipStreamSynchronize(*stream);
      temp.integrateColor(tmpc, T, camera);
    }
    temp.updateEsdf();
  }
  temp.synchronize();
  const bool equal = layers_equal(temp, ref);
  // the stream is idle: the frames come back (at most the pool's worth of them were ever created for this size)
  bool reused = false;
  { std::vector<std::unique_ptr<DepthImage>> probe; for (int k = 0; k < 12 && !reused; k++) { probe.emplace_back(new DepthImage(rows, cols, MemoryType::kDevice)); reused = probe.back()->dataConstPtr() == first_ptr; } }
  int64_t st[6]; nvbx_frame_pool_stats(st);
  std::printf("{\"check\": \"lifetime\", \"frames\": %zu, \"busy_frames\": %d, \"handed_out_while_busy\": %d, \"equal\": %s, \"reused_when_idle\": %s, \"pool_frames\": %lld}\n",
              fr.size(), busy_frames, handed_out_while_busy, equal ? "true" : "false", reused ? "true" : "false", (long long)(st[0] + st[1]));
  return (equal && handed_out_while_busy == 0 && busy_frames > 0 && reused) ? 0 : 1;
}

static int cadence(const std::vector<Fr>& fr, int rows, int cols, const Camera& camera) {
  int failures = 0;
  for (int dynamic = 0; dynamic < 2; dynamic++) {
    auto stream = std::make_shared<CudaStreamOwning>();
    const MappingType mt = dynamic ? MappingType::kDynamic : MappingType::kStaticTsdf;
    MultiMapper with(0.05f, mt, EsdfMode::k2D, MemoryType::kDevice, stream, 1 << 13), without(0.05f, mt, EsdfMode::k2D, MemoryType::kDevice, std::make_shared<CudaStreamOwning>(), 1 << 13);
    // a two-rank world with a stand-in peer: the peer's message names the blocks of THIS rank's frame before (any valid list will do: a peer's block
    // re-marks a column from the local, unchanged TSDF and changes no voxel), so the collective is two device copies in stream order
    int32_t* peer_msg = nullptr; const int64_t max_blocks = 4096; const size_t n_msg = (size_t)(max_blocks + 1) * 3;
    (void)hipMalloc((void**)&peer_msg, n_msg * sizeof(int32_t)); (void)hipMemset(peer_msg, 0, n_msg * sizeof(int32_t));
    int gathers = 0;
    auto ag = [&](const int32_t* send, int32_t* recv, size_t n, hipStream_t st) {
      gathers++;
      (void)hipMemcpyAsync(recv, send, n * 4, hipMemcpyDeviceToDevice, st);                 // rank 0 = this rank
      (void)hipMemcpyAsync(recv + n, peer_msg, n * 4, hipMemcpyDeviceToDevice, st);         // rank 1 = the peer
      (void)hipMemcpyAsync(peer_msg, send, n * 4, hipMemcpyDeviceToDevice, st);             // (next frame's peer list)
      return 0;
    };
    auto ex = std::make_shared<BlockIndexExchange>(2, 0, max_blocks, ag);
    with.setBlockIndexExchange(ex);
    std::vector<DepthImage> d; std::vector<ColorImage> c;
    for (const auto& x : fr) {
      d.emplace_back(MemoryType::kDevice); d.back().copyFromAsync(rows, cols, x.d.data(), *stream);
      c.emplace_back(MemoryType::kDevice); c.back().copyFromAsync(rows, cols, x.c.data(), *stream);
    }
    (void)hipStreamSynchronize(*stream);
    const int ticks = 40;                     // one simulated second of the node: 40 depth frames, colour every 8th, ESDF every 4th
    for (int i = 0; i < ticks; i++) {
      const size_t u = (size_t)i % fr.size(); const Transform T = Transform::fromRowMajor(fr[u].T);
      for (MultiMapper* mm : {&with, &without}) {
        mm->integrateDepth(d[u], T, camera, dynamic ? std::optional<Time>(Time((int64_t)i * 25)) : std::nullopt);
        if (i % 8 == 7) mm->integrateColor(c[u], T, camera);
        if (i % 4 == 3) mm->updateEsdf();
      }
    }
    ex->drain(with.background_mapper()->c_handle());
    with.updateEsdf(); without.updateEsdf();
    with.background_mapper()->synchronize(); without.background_mapper()->synchronize();
    const bool counted = ex->frames_started() == ticks && ex->frames_applied() == ticks && ex->last_applied_frame() == ticks - 1 && ex->applied_in_order() && gathers == ticks;
    const bool equal = layers_equal(*with.background_mapper(), *without.background_mapper());
    std::printf("%s: started %lld, applied %lld, last applied %lld, in order %d, collectives %d, map equals a mapper without exchange %d\n", dynamic ? "dynamic" : "static",
                (long long)ex->frames_started(), (long long)ex->frames_applied(), (long long)ex->last_applied_frame(), (int)ex->applied_in_order(), gathers, (int)equal);
    if (!counted || !equal) failures++;
    with.setBlockIndexExchange(nullptr); ex.reset();
    (void)hipFree(peer_msg);
  }
  std::printf("{\"check\": \"cadence\", \"failures\": %d}\n", failures);
  return failures ? 1 : 0;
}

//   threads <frames.bin>    ADVICE r05 (low): a back-pressure wait inside nvbx_frame_acquire on one thread while another thread destroys the mapper whose
//                           fence it waits for.  Thread B creates a mapper, hands it a colour frame (held back -> retained -> let go of with a fence when the
//                           next depth frame carries it out) and destroys the mapper at once, over and over; thread A acquires and releases frames of the same
//                           size from a pool of two, so it keeps running into B's fences.  Nothing to compare: it must neither crash nor hang, and the pool
//                           must come out consistent (every frame free at the end).
static int threads(const std::vector<Fr>& fr, int rows, int cols, const Camera& camera) {
  setenv("NVBX_FRAME_POOL_MAX", "2", 1);
  (void)nvbx_frame_pool_trim(-1);
  const size_t bytes = (size_t)rows * cols * 3;
  std::atomic<bool> stop{false}; std::atomic<long> acquired{0}; std::atomic<int> errors{0};
  std::thread a([&] {
    (void)hipSetDevice(0);
    while (!stop.load()) {
      void* p = nullptr;
      if (nvbx_frame_acquire(0, bytes, NVBX_STREAM_UNKNOWN, &p) != 0 || !p) { errors++; continue; }
      acquired++;
      if (nvbx_frame_release(p) != 0) errors++;
    }
  });
  nvbx_mapper_params pp; nvbx_default_params(&pp);
  const nvbx_camera cam = camera.c_abi();
  int mappers = 0;
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<void*> dev_depth;
  for (const auto& x : fr) { void* d = nullptr; (void)hipMalloc(&d, x.d.size() * 4); (void)hipMemcpy(d, x.d.data(), x.d.size() * 4, hipMemcpyHostToDevice); dev_depth.push_back(d); }
  while (std::chrono::steady_clock::now() - t0 < std::chrono::seconds(4)) {
    nvbx_mapper* m = nullptr;
    if (nvbx_mapper_create(0, nullptr, &pp, 1 << 11, &m) != 0) { errors++; break; }
    for (int i = 0; i < 3 && !errors; i++) {
      const Fr& x = fr[(size_t)(mappers + i) % fr.size()];
      if (nvbx_integrate_depth(m, (const float*)dev_depth[(size_t)(mappers + i) % fr.size()], rows, cols, x.T, &cam) != 0) errors++;
      void* q = nullptr; void* st = nullptr; (void)nvbx_get_stream(m, &st);
      if (nvbx_color_image_acquire(m, rows, cols, 3, &q) != 0) { errors++; break; }
      if (nvbx_frame_upload(q, x.c.data(), bytes, st) != 0) errors++;
      if (nvbx_integrate_color_owned(m, q, 3, rows, cols, x.T, &cam) != 0) errors++;       // held back: the mapper retains the frame
    }
    (void)nvbx_mapper_destroy(m);          // fences of this mapper die here, possibly under thread A's wait
    mappers++;
  }
  stop = true; a.join();
  for (void* d : dev_depth) (void)hipFree(d);
  (void)hipDeviceSynchronize();
  int64_t st[6]; nvbx_frame_pool_stats(st);
  std::printf("{\"check\": \"threads\", \"mappers\": %d, \"acquired\": %ld, \"errors\": %d, \"held_at_end\": %lld, \"pool_waits\": %lld, \"pool_syncs\": %lld}\n", mappers, acquired.load(), errors.load(),
              (long long)st[0], (long long)st[4], (long long)st[5]);
  return (errors.load() == 0 && st[0] == 0 && mappers > 3 && acquired.load() > 100) ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[2], "rb"); if (!f) return 2;
  int32_t hdr[3]; float k[4];
  if (std::fread(hdr, 4, 3, f) != 3 || std::fread(k, 4, 4, f) != 4) return 2;
  const int n = hdr[0], rows = hdr[1], cols = hdr[2];
  std::vector<Fr> fr((size_t)n);
  for (auto& x : fr) {
    x.d.resize((size_t)rows * cols); x.c.resize((size_t)rows * cols);
    if (std::fread(x.T, 4, 16, f) != 16 || std::fread(x.d.data(), 4, x.d.size(), f) != x.d.size() || std::fread(x.c.data(), 3, x.c.size(), f) != x.c.size()) return 2;
  }
  std::fclose(f);
  const Camera camera(k[0], k[1], k[2], k[3], cols, rows);
  const std::string what = argv[1];
  if (what == "lifetime") return lifetime(fr, rows, cols, camera);
  if (what == "cadence") return cadence(fr, rows, cols, camera);
  if (what == "threads") return threads(fr, rows, cols, camera);
  return 2;
}
