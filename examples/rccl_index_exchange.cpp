// rccl_index_exchange.cpp -- BASELINE.json's north-star collective with a C++ host and no Python: one camera per GPU, one nvblox::MultiMapper per GPU,
// and before every ESDF sweep an RCCL all-gather over xGMI of the block indices each GPU's depth pass has just updated
// (nvblox::BlockIndexExchange, include/nvblox/mapper/block_index_exchange.h; SURVEY.md 8e option (A)).
// ONE process drives N GPUs here (ncclCommInitAll + ncclGroupStart / ncclGroupEnd around the ranks' calls); nvblox_ros would run one node process per
// GPU with ncclCommInitRank and the SAME per-rank calls -- MultiMapper::setBlockIndexExchange hooks the exchange into integrateDepth / integrateColor /
// updateEsdf, so NvbloxNode::processDepthImage / processColorImage / processEsdf (nvblox_node.cpp:1062, 1264, 781) do not change.
//
// Per frame and rank: integrateDepth (writes + starts this frame's message) -> integrateColor (hands the PREVIOUS frame's gathered lists to the mapper:
// the union step rides in a launch of the pipelined frame) -> updateEsdf.  Two launches per frame per GPU, as on one GPU.
//
//   rccl_index_exchange [ranks] [frames] [frames.bin]
//     ranks <= GPUs present: rank r on GPU r, RCCL.  ranks > GPUs present (a 1-GPU box): every rank on GPU r % n, one shared stream, and the
//     all-gather is a loop of device-to-device copies (a stand-in transport with the same group semantics) -- the protocol, the three rotating buffer
//     sets and the mappers' launches are the same.
//     frames.bin (tests/test_cpp_facade.py): int32 {ranks, n, rows, cols}, float {fu fv cu cv}, then rank-major per frame T[16] f32, depth f32, rgb u8;
//     without it: an analytic 6 x 5 x 3 m room, cameras 45 degrees apart, 640 x 480.
// Prints one JSON line: frames/s over all ranks, per-rank map checksums, and whether rank 0's map equals a mapper WITHOUT any exchange fed the same
// frames (it must, bit for bit: re-marking a column from an unchanged TSDF changes no voxel).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "nvblox/nvblox.h"

using namespace nvblox;

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLCHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)

namespace {
struct Frame { float T[16]; std::vector<float> depth; std::vector<Color> rgb; };

// depth + a procedural texture of the inside of a 6 x 5 x 3 m room seen from (0, 0, 1.5) with yaw `yaw` (camera z forward, x right, y down)
void render(float yaw, int rows, int cols, Frame* out) {
  const float f[3] = {std::cos(yaw), std::sin(yaw), 0.f}, r[3] = {std::sin(yaw), -std::cos(yaw), 0.f}, dn[3] = {0.f, 0.f, -1.f}, o[3] = {0.f, 0.f, 1.5f};
  const float Tm[16] = {r[0], dn[0], f[0], o[0], r[1], dn[1], f[1], o[1], r[2], dn[2], f[2], o[2], 0, 0, 0, 1};
  std::memcpy(out->T, Tm, sizeof(Tm));
  out->depth.resize((size_t)rows * cols); out->rgb.resize((size_t)rows * cols);
  const float lo[3] = {-3.f, -2.5f, 0.f}, hi[3] = {3.f, 2.5f, 3.f};
  const float fu = cols / 2.f, cu = cols / 2.f - 0.5f, cv = rows / 2.f - 0.5f;
  for (int v = 0; v < rows; v++) for (int u = 0; u < cols; u++) {
    const float cx = (u - cu) / fu, cy = (v - cv) / fu;
    float t = 1e30f; float d3[3];
    for (int a = 0; a < 3; a++) {
      d3[a] = r[a] * cx + dn[a] * cy + f[a];
      if (d3[a] > 1e-9f) t = std::fmin(t, (hi[a] - o[a]) / d3[a]); else if (d3[a] < -1e-9f) t = std::fmin(t, (lo[a] - o[a]) / d3[a]);
    }
    out->depth[(size_t)v * cols + u] = t;
    const float p[3] = {o[0] + t * d3[0], o[1] + t * d3[1], o[2] + t * d3[2]};
    auto c = [&](float x) { return (uint8_t)((((int)std::floor(8.f * x)) & 1) ? 192 : 64); };
    out->rgb[(size_t)v * cols + u] = Color(c(p[0]), c(p[1]), c(p[2]));
  }
}

struct Checksums { int64_t tsdf_blocks = 0, color_blocks = 0, esdf_blocks = 0, slice_known = 0; double tsdf_sum = 0, slice_sum = 0; int rows = 0, cols = 0; };
Checksums checksums(Mapper& m) {
  Checksums c;
  nvbx_mapper* h = m.c_handle();
  c.tsdf_blocks = nvbx_num_blocks(h, NVBX_LAYER_TSDF); c.color_blocks = nvbx_num_blocks(h, NVBX_LAYER_COLOR); c.esdf_blocks = nvbx_num_blocks(h, NVBX_LAYER_ESDF);
  std::vector<nvbx_index3d> idx((size_t)c.tsdf_blocks); nvbx_block_indices(h, NVBX_LAYER_TSDF, idx.data(), c.tsdf_blocks);
  std::vector<nvbx_tsdf_voxel> vox((size_t)c.tsdf_blocks * 512);
  if (c.tsdf_blocks) nvbx_get_blocks(h, NVBX_LAYER_TSDF, idx.data(), c.tsdf_blocks, vox.data(), nullptr);
  for (const auto& v : vox) if (v.weight > 0.f) c.tsdf_sum += (double)v.distance * (double)v.weight;
  int32_t rows = 0, cols = 0; float bb[6];
  if (nvbx_esdf_slice_size(h, &rows, &cols, bb) == 0 && rows > 0 && cols > 0) {
    std::vector<float> img((size_t)rows * cols);
    nvbx_esdf_slice_to_host(h, 1000.0f, img.data(), (int64_t)img.size(), &rows, &cols, bb);
    for (float v : img) if (v < 999.0f) { c.slice_known++; c.slice_sum += (double)v; }
  }
  c.rows = rows; c.cols = cols;
  return c;
}
bool layers_equal(Mapper& a, Mapper& b) {
  const uint32_t layers[3] = {NVBX_LAYER_TSDF, NVBX_LAYER_COLOR, NVBX_LAYER_ESDF}; const size_t vox_bytes[3] = {8, 8, 20};
  for (int l = 0; l < 3; l++) {
    const int64_t na = nvbx_num_blocks(a.c_handle(), layers[l]), nb = nvbx_num_blocks(b.c_handle(), layers[l]);
    if (na != nb) return false;
    std::vector<nvbx_index3d> ia((size_t)na), ib((size_t)na);
    nvbx_block_indices(a.c_handle(), layers[l], ia.data(), na); nvbx_block_indices(b.c_handle(), layers[l], ib.data(), na);
    if (na && std::memcmp(ia.data(), ib.data(), sizeof(nvbx_index3d) * (size_t)na)) return false;
    std::vector<uint8_t> va((size_t)na * 512 * vox_bytes[l]), vb(va.size());
    if (na) { nvbx_get_blocks(a.c_handle(), layers[l], ia.data(), na, va.data(), nullptr); nvbx_get_blocks(b.c_handle(), layers[l], ia.data(), na, vb.data(), nullptr); }
    if (va != vb) return false;
  }
  return true;
}
}  // namespace

int main(int argc, char** argv) {
  int ndev = 0; HIPCHECK(hipGetDeviceCount(&ndev)); if (ndev < 1) return 2;
  int ranks = argc > 1 ? std::atoi(argv[1]) : ndev; ranks = std::max(1, std::min(ranks, 8));
  int n_frames = argc > 2 ? std::atoi(argv[2]) : 200;
  const char* file = argc > 3 ? argv[3] : nullptr;
  int rows = 480, cols = 640; float k[4] = {320.f, 320.f, 319.5f, 239.5f};
  std::vector<std::vector<Frame>> fr((size_t)ranks);
  if (file) {
    FILE* f = std::fopen(file, "rb"); if (!f) return 2;
    int32_t hdr[4]; if (std::fread(hdr, 4, 4, f) != 4 || std::fread(k, 4, 4, f) != 4) return 2;
    ranks = hdr[0]; n_frames = hdr[1]; rows = hdr[2]; cols = hdr[3]; fr.assign((size_t)ranks, {});
    for (int r = 0; r < ranks; r++) for (int i = 0; i < n_frames; i++) {
      Frame x; x.depth.resize((size_t)rows * cols); x.rgb.resize((size_t)rows * cols);
      if (std::fread(x.T, 4, 16, f) != 16 || std::fread(x.depth.data(), 4, x.depth.size(), f) != x.depth.size() || std::fread(x.rgb.data(), 3, x.rgb.size(), f) != x.rgb.size()) return 2;
      fr[(size_t)r].push_back(std::move(x));
    }
    std::fclose(f);
  } else {
    const int unique = std::min(n_frames, 24);
    for (int r = 0; r < ranks; r++) for (int i = 0; i < unique; i++) { Frame x; render(0.26f * i + 0.7853982f * r, rows, cols, &x); fr[(size_t)r].push_back(std::move(x)); }
  }
  const int unique = (int)fr[0].size();
  const bool rccl = ranks <= ndev && ranks > 1;
  const Camera camera(k[0], k[1], k[2], k[3], cols, rows);

  // ---- per rank: device, stream, MultiMapper, resident frames, exchange
  std::vector<ncclComm_t> comms((size_t)ranks, nullptr);
  if (rccl) { std::vector<int> devs((size_t)ranks); for (int r = 0; r < ranks; r++) devs[(size_t)r] = r; NCCLCHECK(ncclCommInitAll(comms.data(), ranks, devs.data())); }
  std::shared_ptr<CudaStream> shared_stream;        // stand-in transport: every rank on one stream
  struct Pending { const int32_t* send; int32_t* recv; size_t n; hipStream_t stream; };
  std::vector<Pending> group((size_t)ranks, Pending{nullptr, nullptr, 0, nullptr});
  std::vector<std::shared_ptr<MultiMapper>> mm((size_t)ranks);
  std::vector<std::shared_ptr<BlockIndexExchange>> ex((size_t)ranks);
  std::vector<std::vector<DepthImage>> d_depth((size_t)ranks); std::vector<std::vector<ColorImage>> d_rgb((size_t)ranks);
  for (int r = 0; r < ranks; r++) {
    HIPCHECK(hipSetDevice(r % ndev));
    std::shared_ptr<CudaStream> s;
    if (rccl || ranks == 1) s = std::make_shared<CudaStreamOwning>();
    else { if (!shared_stream) shared_stream = std::make_shared<CudaStreamOwning>(); s = shared_stream; }
    mm[(size_t)r] = std::make_shared<MultiMapper>(0.05f, MappingType::kStaticTsdf, EsdfMode::k2D, MemoryType::kDevice, s, 1 << 14);
    for (int i = 0; i < unique; i++) {
      d_depth[(size_t)r].emplace_back(MemoryType::kDevice); d_depth[(size_t)r].back().copyFromAsync(rows, cols, fr[(size_t)r][(size_t)i].depth.data(), *s);
      d_rgb[(size_t)r].emplace_back(MemoryType::kDevice); d_rgb[(size_t)r].back().copyFromAsync(rows, cols, fr[(size_t)r][(size_t)i].rgb.data(), *s);
    }
    HIPCHECK(hipStreamSynchronize(*s));
    BlockIndexExchange::AllGather ag;
    if (rccl) { ncclComm_t c = comms[(size_t)r]; ag = [c](const int32_t* send, int32_t* recv, size_t n, hipStream_t st) { return ncclAllGather(send, recv, n, ncclInt32, c, st) == ncclSuccess ? 0 : 1; }; }
    else ag = [&group, r](const int32_t* send, int32_t* recv, size_t n, hipStream_t st) { group[(size_t)r] = Pending{send, recv, n, st}; return 0; };     // carried out at group_end
    // NVBX_EXCHANGE_COMM_STREAM=1: the collective on a stream of its own per rank (events order it with the mapper's stream both ways) instead of on the mapper's stream
    hipStream_t cs = nullptr;
    if (const char* e = std::getenv("NVBX_EXCHANGE_COMM_STREAM")) if (e[0] == '1') HIPCHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    ex[(size_t)r] = std::make_shared<BlockIndexExchange>(ranks, r, 4096, ag, cs);
  }
  auto group_begin = [&]() { if (rccl) ncclGroupStart(); };
  auto group_end = [&]() {
    if (rccl) { ncclGroupEnd(); return; }
    for (int r = 0; r < ranks; r++) { if (!group[(size_t)r].recv) continue;
      for (int q = 0; q < ranks; q++) (void)hipMemcpyAsync(group[(size_t)r].recv + (size_t)q * group[(size_t)r].n, group[(size_t)q].send, group[(size_t)r].n * 4, hipMemcpyDeviceToDevice, group[(size_t)r].stream); }
    for (auto& g : group) g = Pending{nullptr, nullptr, 0, nullptr};
  };

  // the comparison mapper: rank 0's frames, no exchange, classic launch order
  HIPCHECK(hipSetDevice(0));
  Mapper plain(0.05f, MemoryType::kDevice, ProjectiveLayerType::kTsdf, std::make_shared<CudaStreamOwning>(), 1 << 14);
  plain.setColorIntegrationDeferred(false);

  // ---- prelude: every rank (and the comparison mapper) knows the whole room -- the peers' blocks exist locally, as after a few seconds of mapping
  for (int r = 0; r < ranks; r++) {
    HIPCHECK(hipSetDevice(r % ndev));
    DepthImage tmp(MemoryType::kDevice);
    for (int q = 0; q < ranks; q++) for (int i = 0; i < unique; i++) {
      const Frame& x = fr[(size_t)q][(size_t)i];
      tmp.copyFromAsync(rows, cols, x.depth.data(), *mm[(size_t)r]->background_mapper()->cuda_stream());
      mm[(size_t)r]->integrateDepth(tmp, Transform::fromRowMajor(x.T), camera);
      if (r == 0) { HIPCHECK(hipStreamSynchronize(*mm[0]->background_mapper()->cuda_stream())); plain.integrateDepth(tmp, Transform::fromRowMajor(x.T), camera); plain.synchronize(); }
    }
    mm[(size_t)r]->updateEsdf(); mm[(size_t)r]->background_mapper()->synchronize();
  }
  plain.updateEsdf(); plain.synchronize();
  for (int r = 0; r < ranks; r++) mm[(size_t)r]->setBlockIndexExchange(ex[(size_t)r]);

  // ---- the loop
  auto frame = [&](int i) {
    const int u = i % unique;
    group_begin();
    for (int r = 0; r < ranks; r++) { (void)hipSetDevice(r % ndev); mm[(size_t)r]->integrateDepth(d_depth[(size_t)r][(size_t)u], Transform::fromRowMajor(fr[(size_t)r][(size_t)u].T), camera); }
    group_end();
    for (int r = 0; r < ranks; r++) {
      (void)hipSetDevice(r % ndev);
      mm[(size_t)r]->integrateColor(d_rgb[(size_t)r][(size_t)u], Transform::fromRowMajor(fr[(size_t)r][(size_t)u].T), camera);
      mm[(size_t)r]->updateEsdf();
    }
  };
  auto sync_all = [&]() { for (int r = 0; r < ranks; r++) { (void)hipSetDevice(r % ndev); mm[(size_t)r]->background_mapper()->synchronize(); } };
  const int warm = std::min(5, n_frames);
  for (int i = 0; i < warm; i++) frame(i);
  sync_all();
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = warm; i < warm + n_frames; i++) frame(i);
  for (int r = 0; r < ranks; r++) { (void)hipSetDevice(r % ndev); ex[(size_t)r]->drain(mm[(size_t)r]->background_mapper()->c_handle()); mm[(size_t)r]->updateEsdf(); }
  sync_all();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // ---- the comparison mapper runs rank 0's frames the classic way
  HIPCHECK(hipSetDevice(0));
  for (int i = 0; i < warm + n_frames; i++) {
    const int u = i % unique; const Transform T = Transform::fromRowMajor(fr[0][(size_t)u].T);
    plain.integrateDepth(d_depth[0][(size_t)u], T, camera); plain.integrateColor(d_rgb[0][(size_t)u], T, camera); plain.updateEsdf();
  }
  plain.updateEsdf(); plain.synchronize();
  const bool equal = layers_equal(*mm[0]->background_mapper(), plain);
  nvbx_counters c0{}, c1{}; nvbx_get_counters(plain.c_handle(), &c0); nvbx_get_counters(mm[0]->background_mapper()->c_handle(), &c1);

  std::string per_rank = "[";
  for (int r = 0; r < ranks; r++) {
    (void)hipSetDevice(r % ndev);
    const Checksums c = checksums(*mm[(size_t)r]->background_mapper());
    char b[512];
    std::snprintf(b, sizeof(b), "%s{\"tsdf_blocks\": %lld, \"color_blocks\": %lld, \"esdf_blocks\": %lld, \"tsdf_sum\": %.9g, \"slice_shape\": [%d, %d], \"slice_known\": %lld, \"slice_sum\": %.9g}",
                  r ? ", " : "", (long long)c.tsdf_blocks, (long long)c.color_blocks, (long long)c.esdf_blocks, c.tsdf_sum, c.rows, c.cols, (long long)c.slice_known, c.slice_sum);
    per_rank += b;
  }
  per_rank += "]";
  // the parameters the mappers ran with (the facade's defaults), for a caller that drives the same protocol through another binding
  nvbx_mapper_params pp{}; nvbx_mapper_get_params(mm[0]->background_mapper()->c_handle(), &pp);
  std::string params_hex; { const unsigned char* b = reinterpret_cast<const unsigned char*>(&pp); char h[3]; for (size_t i = 0; i < sizeof(pp); i++) { std::snprintf(h, sizeof(h), "%02x", b[i]); params_hex += h; } }
  std::printf("{\"params_hex\": \"%s\", ", params_hex.c_str());
  std::printf("\"ranks\": %d, \"gpus\": %d, \"transport\": \"%s\", \"frames_per_rank\": %d, \"image\": [%d, %d], \"frames_per_s\": %.1f, \"ms_per_frame_per_rank\": %.4f, "
              "\"rank0_equals_mapper_without_exchange\": %s, \"esdf_columns_marked_last_update\": [%lld, %lld], \"per_rank\": %s}\n",
              ranks, std::min(ranks, ndev), rccl ? "rccl" : (ranks > 1 ? "stand-in (device copies, one stream)" : "none"), n_frames, rows, cols, ranks * n_frames / dt, dt / n_frames * 1e3,
              equal ? "true" : "false", (long long)c0.esdf_columns_marked, (long long)c1.esdf_columns_marked, per_rank.c_str());
  for (int r = 0; r < ranks; r++) { (void)hipSetDevice(r % ndev); mm[(size_t)r]->setBlockIndexExchange(nullptr); ex[(size_t)r].reset(); mm[(size_t)r].reset(); if (comms[(size_t)r]) ncclCommDestroy(comms[(size_t)r]); }
  return equal ? 0 : 3;
}
