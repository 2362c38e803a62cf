// rccl_fusion.cpp -- the multi-GPU measurement exchange driven from C++ through the C-ABI and RCCL directly (no Python, no torch):
// ONE process, N GPUs (ncclCommInitAll), one nvbx_mapper + one camera per GPU.  Per frame
//     nvbx_measure_depth (each GPU: view calculation + projection of ITS camera)
//  -> ncclAllGather of the record COUNTS (4 B per rank), copied to the host
//  -> ncclAllGather of the first n records of every rank's buffer, n = max(count) rounded up to 64 records: only what is used goes
//     over xGMI (~1.3 MB per rank for ~300 blocks in view, not the 4.2 MB the buffers are sized for); grouped, on the mappers' streams
//  -> nvbx_apply_measurements with stride n (each GPU: every camera's measurements, in rank order)
// after which every GPU holds the same fused map (include/nvblox_hip.h "measurement exchange"; SURVEY.md 8e option B, made exact).
// The program checks that against ONE mapper on GPU 0 integrating the N cameras as a batch (nvbx_integrate_depth_batch): block sets and
// voxels must be identical.  nvblox_ros would drive the same three calls from NvbloxNode::processDepthImage (nvblox_node.cpp:1062) with
// one node process per GPU and ncclCommInitRank instead of ncclCommInitAll.
//   make -C tests/cpp rccl_fusion && tests/cpp/rccl_fusion [n_gpus] [n_frames]      (n_gpus is clamped to the devices present)
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "nvblox_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ < 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, nvbx_last_error()); return 1; } } while (0)
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLCHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)

namespace {
constexpr int ROWS = 120, COLS = 160, STRIDE = 1024;
// depth of the inside of a 6 x 5 x 3 m room seen from (0, 0, 1.5) with yaw `yaw` (camera z forward, x right, y down)
void render(float yaw, float* depth, float* T /* row-major T_L_C */) {
  const float f[3] = {std::cos(yaw), std::sin(yaw), 0.f}, r[3] = {std::sin(yaw), -std::cos(yaw), 0.f}, dn[3] = {0.f, 0.f, -1.f}, o[3] = {0.f, 0.f, 1.5f};
  const float Tm[16] = {r[0], dn[0], f[0], o[0], r[1], dn[1], f[1], o[1], r[2], dn[2], f[2], o[2], 0, 0, 0, 1};
  std::memcpy(T, Tm, sizeof(Tm));
  const float lo[3] = {-3.f, -2.5f, 0.f}, hi[3] = {3.f, 2.5f, 3.f};
  for (int v = 0; v < ROWS; v++) for (int u = 0; u < COLS; u++) {
    const float cx = (u + 0.5f - 79.5f) / 80.f, cy = (v + 0.5f - 59.5f) / 80.f;
    float t = 1e30f;
    for (int a = 0; a < 3; a++) {
      const float d = r[a] * cx + dn[a] * cy + f[a];
      if (d > 1e-9f) t = std::fmin(t, (hi[a] - o[a]) / d); else if (d < -1e-9f) t = std::fmin(t, (lo[a] - o[a]) / d);
    }
    depth[v * COLS + u] = t;          // camera-z depth = ray parameter (the ray is (cx, cy, 1) in the camera frame)
  }
}
}  // namespace

int main(int argc, char** argv) {
  int ndev = 0; HIPCHECK(hipGetDeviceCount(&ndev));
  int n = argc > 1 ? std::atoi(argv[1]) : ndev; if (n > ndev) n = ndev; if (n > NVBX_MAX_BATCH) n = NVBX_MAX_BATCH; if (n < 1) return 2;
  const int n_frames = argc > 2 ? std::atoi(argv[2]) : 3;
  std::vector<int> devs(n); for (int i = 0; i < n; i++) devs[i] = i;
  std::vector<ncclComm_t> comms(n);
  NCCLCHECK(ncclCommInitAll(comms.data(), n, devs.data()));
  nvbx_mapper_params p; nvbx_default_params(&p);
  const nvbx_camera cam = {80.f, 80.f, 79.5f, 59.5f, COLS, ROWS};
  std::vector<nvbx_mapper*> rank(n, nullptr);
  std::vector<hipStream_t> stream(n);
  std::vector<float*> d_depth(n); std::vector<nvbx_measurement_block*> buf(n), all_buf(n); std::vector<int32_t*> cnt(n), all_cnt(n);
  for (int i = 0; i < n; i++) {
    HIPCHECK(hipSetDevice(i));
    CHECK(nvbx_mapper_create(i, nullptr, &p, 1 << 13, &rank[i]));
    void* s = nullptr; CHECK(nvbx_get_stream(rank[i], &s)); stream[i] = (hipStream_t)s;       // collectives go on the mapper's own stream: no events needed
    HIPCHECK(hipMalloc((void**)&d_depth[i], sizeof(float) * ROWS * COLS));
    HIPCHECK(hipMalloc((void**)&buf[i], sizeof(nvbx_measurement_block) * STRIDE)); HIPCHECK(hipMalloc((void**)&all_buf[i], sizeof(nvbx_measurement_block) * STRIDE * n));
    HIPCHECK(hipMalloc((void**)&cnt[i], 4)); HIPCHECK(hipMalloc((void**)&all_cnt[i], 4 * n));
  }
  // the reference: one mapper on GPU 0, the n cameras as a batch
  HIPCHECK(hipSetDevice(0));
  nvbx_mapper* single = nullptr; CHECK(nvbx_mapper_create(0, nullptr, &p, 1 << 13, &single));
  std::vector<float*> d_batch(n); for (int i = 0; i < n; i++) HIPCHECK(hipMalloc((void**)&d_batch[i], sizeof(float) * ROWS * COLS));
  std::vector<float> depth(ROWS * COLS), T(16 * n);
  int64_t sent_bytes = 0, used_bytes = 0;          // per rank: bytes handed to the payload collective / bytes of used records (largest rank)
  for (int k = 0; k < n_frames; k++) {
    for (int i = 0; i < n; i++) {
      render(0.3f * k + 0.7853982f * i, depth.data(), &T[16 * i]);
      HIPCHECK(hipSetDevice(i)); HIPCHECK(hipMemcpy(d_depth[i], depth.data(), sizeof(float) * ROWS * COLS, hipMemcpyHostToDevice));
      HIPCHECK(hipSetDevice(0)); HIPCHECK(hipMemcpy(d_batch[i], depth.data(), sizeof(float) * ROWS * COLS, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < n; i++) CHECK(nvbx_measure_depth(rank[i], d_depth[i], ROWS, COLS, &T[16 * i], &cam, buf[i], cnt[i], STRIDE));
    // phase 1: the counts (every rank learns every count; the host sizes the payload from them)
    NCCLCHECK(ncclGroupStart());
    for (int i = 0; i < n; i++) NCCLCHECK(ncclAllGather(cnt[i], all_cnt[i], 1, ncclInt32, comms[i], stream[i]));
    NCCLCHECK(ncclGroupEnd());
    std::vector<int32_t> h_cnt(n);
    HIPCHECK(hipSetDevice(0)); HIPCHECK(hipMemcpyAsync(h_cnt.data(), all_cnt[0], 4 * n, hipMemcpyDeviceToHost, stream[0])); HIPCHECK(hipStreamSynchronize(stream[0]));
    int64_t used = 0; for (int i = 0; i < n; i++) used = std::max<int64_t>(used, std::min<int64_t>(h_cnt[i], STRIDE));
    const int64_t n_rec = std::min<int64_t>(STRIDE, std::max<int64_t>(64, (used + 63) / 64 * 64));
    sent_bytes += n_rec * (int64_t)sizeof(nvbx_measurement_block); used_bytes += used * (int64_t)sizeof(nvbx_measurement_block);
    // phase 2: the used prefix of every rank's records
    NCCLCHECK(ncclGroupStart());
    for (int i = 0; i < n; i++) NCCLCHECK(ncclAllGather(buf[i], all_buf[i], sizeof(nvbx_measurement_block) * (size_t)n_rec, ncclChar, comms[i], stream[i]));
    NCCLCHECK(ncclGroupEnd());
    for (int i = 0; i < n; i++) CHECK(nvbx_apply_measurements(rank[i], all_buf[i], all_cnt[i], n, n_rec, 0, 0));
    CHECK(nvbx_integrate_depth_batch(single, n, d_batch.data(), ROWS, COLS, T.data(), std::vector<nvbx_camera>(n, cam).data()));
  }
  // every rank's map == the single mapper's
  const int64_t nb = nvbx_num_blocks(single, NVBX_LAYER_TSDF);
  std::vector<nvbx_index3d> idx((size_t)nb), idx_r((size_t)nb);
  std::vector<nvbx_tsdf_voxel> ref((size_t)nb * 512), got((size_t)nb * 512);
  if (nvbx_block_indices(single, NVBX_LAYER_TSDF, idx.data(), nb) != nb) return 1;
  CHECK(nvbx_get_blocks(single, NVBX_LAYER_TSDF, idx.data(), nb, ref.data(), nullptr));
  int bad = 0;
  for (int i = 0; i < n; i++) {
    if (nvbx_num_blocks(rank[i], NVBX_LAYER_TSDF) != nb || nvbx_block_indices(rank[i], NVBX_LAYER_TSDF, idx_r.data(), nb) != nb ||
        std::memcmp(idx.data(), idx_r.data(), sizeof(nvbx_index3d) * (size_t)nb)) { bad++; continue; }
    CHECK(nvbx_get_blocks(rank[i], NVBX_LAYER_TSDF, idx.data(), nb, got.data(), nullptr));
    if (std::memcmp(ref.data(), got.data(), sizeof(nvbx_tsdf_voxel) * (size_t)nb * 512)) bad++;
  }
  std::printf("{\"gpus\": %d, \"frames\": %d, \"tsdf_blocks\": %lld, \"ranks_differing_from_single_mapper\": %d, \"payload_bytes_sent_per_rank\": %lld, \"payload_bytes_used\": %lld, \"buffer_bytes\": %lld}\n",
              n, n_frames, (long long)nb, bad, (long long)sent_bytes, (long long)used_bytes, (long long)((int64_t)n_frames * STRIDE * (int64_t)sizeof(nvbx_measurement_block)));
  for (int i = 0; i < n; i++) { nvbx_mapper_destroy(rank[i]); ncclCommDestroy(comms[i]); }
  nvbx_mapper_destroy(single);
  return (bad == 0 && nb > 100) ? 0 : 3;
}
