// fuser_loop.cpp -- the offline fuser loop in C++ over the nvblox:: facade: the shape of FuserNode::fuseNextFrame /
// Fuser::integrateFrame (nvblox_ros/src/lib/fuser_node.cpp:202-224, fuser_node_main.cpp:36-52) without ROS and without
// the dataset loaders (no Replica / Redwood data here): frames come from a frames.bin written by the Python generator
// (tests/test_cpp_facade.py writes the same format), are uploaded once, and are then replayed through
//   integrateDepth -> integrateColor -> updateEsdf (every esdf_every frames) -> updateColorMesh (every mesh_every frames)
// with the reference's timer tags; it prints timing::Timing::Print() like the node does on shutdown
// (nvblox_node.cpp:178-180) and one JSON line with ms/frame measured around the whole loop (stream synchronised).
//
// usage: fuser_loop frames.bin [n_frames_to_integrate] [esdf_every] [mesh_every] [deferred_colour 0|1]
// deferred_colour = 1: Mapper::setColorIntegrationDeferred(true) -- the libnvblox_hip extension that carries integrateColor(i) / updateEsdf(i) out
// inside integrateDepth(i + 1), two launches per frame; the colour images here are resident and unchanged, as its contract asks.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>
#include <nvblox/nvblox.h>

using namespace nvblox;

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s frames.bin [n_frames] [esdf_every] [mesh_every] [deferred_colour]\n", argv[0]); return 2; }
  const int number_of_frames_to_integrate = argc > 2 ? std::atoi(argv[2]) : -1;      // fuser_node.cpp:207-209
  const int esdf_every = argc > 3 ? std::atoi(argv[3]) : 1, mesh_every = argc > 4 ? std::atoi(argv[4]) : 0;
  const bool deferred = argc > 5 && std::atoi(argv[5]) != 0;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 2; }
  int32_t hdr[3]; float k[4];
  if (std::fread(hdr, 4, 3, f) != 3 || std::fread(k, 4, 4, f) != 4) return 2;
  const int n = hdr[0], rows = hdr[1], cols = hdr[2];

  warmupCuda();                                                                        // fuser_node_main.cpp:38
  auto cuda_stream = std::make_shared<CudaStreamOwning>();
  // fuser_node.cpp:85-94
  auto multi_mapper = std::make_shared<MultiMapper>(0.05f, MappingType::kStaticTsdf, EsdfMode::k2D, MemoryType::kDevice, cuda_stream);
  MapperParams p;                                                                      // fuser.yaml:24-42
  p.projective_integrator_params.projective_integrator_max_integration_distance_m = 8.0f;
  p.projective_integrator_params.projective_integrator_truncation_distance_vox = 4.0f;
  p.projective_integrator_params.projective_integrator_weighting_mode = WeightingFunctionType::kConstantWeight;
  p.projective_integrator_params.projective_integrator_max_weight = 5.0f;
  p.esdf_integrator_params.esdf_slice_height = 0.09f; p.esdf_integrator_params.esdf_slice_min_height = 0.09f;
  p.esdf_integrator_params.esdf_slice_max_height = 0.65f;
  multi_mapper->setMapperParams(p);
  if (deferred) multi_mapper->background_mapper()->setColorIntegrationDeferred(true);
  const Camera camera(k[0], k[1], k[2], k[3], cols, rows);

  // the data loader's job: frames resident on the device
  std::vector<std::unique_ptr<DepthImage>> depth; std::vector<std::unique_ptr<ColorImage>> color; std::vector<Transform> T_L_C;
  std::vector<float> d((size_t)rows * cols); std::vector<Color> c((size_t)rows * cols); float T[16];
  for (int i = 0; i < n; i++) {
    if (std::fread(T, 4, 16, f) != 16 || std::fread(d.data(), 4, d.size(), f) != d.size() || std::fread(c.data(), 3, c.size(), f) != c.size()) return 2;
    depth.emplace_back(new DepthImage(MemoryType::kDevice)); color.emplace_back(new ColorImage(MemoryType::kDevice));
    depth.back()->copyFromAsync(rows, cols, d.data(), *cuda_stream); color.back()->copyFromAsync(rows, cols, c.data(), *cuda_stream);
    cuda_stream->synchronize();
    T_L_C.push_back(Transform::fromRowMajor(T));
  }
  std::fclose(f);

  const int total = number_of_frames_to_integrate < 0 ? n : number_of_frames_to_integrate;
  auto integrateFrame = [&](int frame_number) {                                        // [U] Fuser::integrateFrame
    const int i = frame_number % n;
    multi_mapper->integrateDepth(*depth[i], T_L_C[i], camera);
    multi_mapper->integrateColor(*color[i], T_L_C[i], camera);
    if (esdf_every > 0 && frame_number % esdf_every == 0) multi_mapper->updateEsdf();
    if (mesh_every > 0 && frame_number % mesh_every == 0) multi_mapper->updateColorMesh();
  };
  for (int i = 0; i < std::min(total, 20); i++) integrateFrame(i);                     // warm-up, like warmupCuda()
  cuda_stream->synchronize();
  timing::Timing::Reset();
  const auto t0 = std::chrono::steady_clock::now();
  int current_frame_number_ = 0;
  while (current_frame_number_ < total) integrateFrame(current_frame_number_++);       // fuser_node_main.cpp:45-52
  cuda_stream->synchronize();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::fprintf(stderr, "%s", timing::Timing::Print().c_str());
  std::printf("{\"frames\": %d, \"ms_per_frame\": %.5f, \"tsdf_blocks\": %d, \"esdf_every\": %d, \"mesh_every\": %d, \"deferred_colour\": %d, \"host\": \"c++ facade\"}\n",
              total, ms / total, multi_mapper->background_mapper()->tsdf_layer().numAllocatedBlocks(), esdf_every, mesh_every, deferred ? 1 : 0);
  return 0;
}
