// image_frames.cpp -- the node's ONE re-used nvblox::ColorImage under colour deferral, through the facade only (no library-specific call in the loop).
// What nvblox_ros does every colour frame (nvblox_node.cpp:1237-1264): the converter writes the whole image through the NON-CONST dataPtr() on the node's
// stream (conversions/image_conversions_thrust.cu:75-80), then integrateColor(color_image_, ...).  Here the converter is a hipMemcpyAsync into dataPtr(),
// and right after integrateColor the host scribbles over "its" image -- on the node's stream, on a second stream, or with a blocking host copy.
// A mapper in its default setting (integrateColor held back, two launches per frame) must end with the colour layer of a classic-order mapper, bit for
// bit, WITHOUT a k_stage_color launch: the image's device memory is a library-owned frame the mapper retains, and the image rotates to another frame
// on its next write access (include/nvblox/sensors/image.h).
// usage: image_frames <frames.bin>   (the file tests/test_cpp_facade.py writes: n, rows, cols, fu fv cu cv, then per frame T[16], depth f32, rgb u8)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "nvblox/nvblox.h"

using namespace nvblox;

static std::vector<uint8_t> colour_layer(Mapper& m, std::vector<nvbx_index3d>* idx_out) {
  const int64_t n = nvbx_num_blocks(m.c_handle(), NVBX_LAYER_COLOR);
  std::vector<nvbx_index3d> idx((size_t)n);
  nvbx_block_indices(m.c_handle(), NVBX_LAYER_COLOR, idx.data(), n);
  std::vector<uint8_t> vox((size_t)n * 512 * 8);
  if (n) nvbx_get_blocks(m.c_handle(), NVBX_LAYER_COLOR, idx.data(), n, vox.data(), nullptr);
  *idx_out = idx;
  return vox;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb"); if (!f) return 2;
  int32_t hdr[3]; float k[4];
  if (std::fread(hdr, 4, 3, f) != 3 || std::fread(k, 4, 4, f) != 4) return 2;
  const int n = hdr[0], rows = hdr[1], cols = hdr[2];
  struct Fr { float T[16]; std::vector<float> d; std::vector<Color> c; };
  std::vector<Fr> fr((size_t)n);
  for (auto& x : fr) {
    x.d.resize((size_t)rows * cols); x.c.resize((size_t)rows * cols);
    if (std::fread(x.T, 4, 16, f) != 16 || std::fread(x.d.data(), 4, x.d.size(), f) != x.d.size() || std::fread(x.c.data(), 3, x.c.size(), f) != x.c.size()) return 2;
  }
  std::fclose(f);
  const Camera camera(k[0], k[1], k[2], k[3], cols, rows);
  std::vector<Color> noise((size_t)rows * cols); for (size_t i = 0; i < noise.size(); i++) noise[i] = Color((uint8_t)(i * 7), (uint8_t)(i * 13), (uint8_t)(i * 29));
  int64_t stats0[6]; nvbx_frame_pool_stats(stats0);

  int failures = 0;
  for (int mode = 0; mode < 3; mode++) {          // who scribbles: 0 = the node's stream, 1 = a second stream (ordered by the caller), 2 = a blocking host copy
    auto stream = CudaStream::createCudaStream(static_cast<CudaStreamType>(2));
    CudaStreamOwning second;
    Mapper classic(0.05f, MemoryType::kDevice, ProjectiveLayerType::kTsdf, stream, 1 << 13), piped(0.05f, MemoryType::kDevice, ProjectiveLayerType::kTsdf, stream, 1 << 13);
    classic.setColorIntegrationDeferred(false);
    nvbx_set_profiling(piped.c_handle(), 1);
    DepthImage depth(MemoryType::kDevice);
    ColorImage image_classic(MemoryType::kDevice), image(MemoryType::kDevice);      // `image`: the node's one colour image
    int rotations = 0; bool shared_seen = true;
    for (int rep = 0; rep < 3; rep++) for (const auto& x : fr) {
      const Transform T = Transform::fromRowMajor(x.T);
      depth.copyFromAsync(rows, cols, x.d.data(), *stream);
      classic.integrateDepth(depth, T, camera); piped.integrateDepth(depth, T, camera);
      image_classic.copyFromAsync(rows, cols, x.c.data(), *stream);
      classic.integrateColor(image_classic, T, camera);
      // the converter: maybeReallocateImage + a full overwrite through the non-const dataPtr() on the node's stream
      if (image.rows() != rows || image.cols() != cols) image = ColorImage(rows, cols, MemoryType::kDevice);
      Color* out = image.dataPtr();
      (void)hipMemcpyAsync(out, x.c.data(), (size_t)rows * cols * 3, hipMemcpyHostToDevice, *stream);
      (void)hipStreamSynchronize(*stream);        // (pageable source: make sure the bytes are there whatever the runtime does with it)
      piped.integrateColor(image, T, camera);
      shared_seen = shared_seen && image.sharedWithMapper();
      // ... and the scribble, right behind the call: the write access must land in ANOTHER frame
      const Color* held = image.dataConstPtr();
      Color* scr = image.dataPtr();
      if (scr != held) rotations++;
      if (mode == 0) (void)hipMemcpyAsync(scr, noise.data(), noise.size() * 3, hipMemcpyHostToDevice, *stream);
      else if (mode == 1) { (void)hipMemcpyAsync(scr, noise.data(), noise.size() * 3, hipMemcpyHostToDevice, second); (void)hipStreamSynchronize(second); }
      else (void)hipMemcpy(scr, noise.data(), noise.size() * 3, hipMemcpyHostToDevice);
      classic.updateEsdf(); piped.updateEsdf();
    }
    classic.synchronize(); piped.synchronize();
    std::vector<nvbx_index3d> ia, ib;
    const std::vector<uint8_t> a = colour_layer(classic, &ia), b = colour_layer(piped, &ib);
    const bool same = ia.size() == ib.size() && !ia.empty() && std::memcmp(ia.data(), ib.data(), ia.size() * sizeof(nvbx_index3d)) == 0 && a == b;
    std::vector<char> prof(1 << 16); nvbx_get_profile(piped.c_handle(), prof.data(), (int64_t)prof.size());
    const bool copied = std::strstr(prof.data(), "k_stage_color") != nullptr;
    const bool fused = std::strstr(prof.data(), "k_integrate_tsdf_color") != nullptr;
    std::printf("mode %d: colour blocks %zu, equal %d, shared after integrateColor %d, rotations %d, stage copy %d, fused launches %d\n", mode, ia.size(), (int)same, (int)shared_seen, rotations,
                (int)copied, (int)fused);
    if (!same || !shared_seen || rotations != 3 * n || copied || !fused) failures++;
  }
  int64_t stats[6]; nvbx_frame_pool_stats(stats);
  std::printf("{\"failures\": %d, \"frames_created\": %lld, \"pool_waits\": %lld, \"pool_syncs\": %lld}\n", failures, (long long)(stats[3] - stats0[3]), (long long)(stats[4] - stats0[4]),
              (long long)(stats[5] - stats0[5]));
  return failures ? 1 : 0;
}
