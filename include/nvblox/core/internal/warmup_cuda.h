// nvblox/core/internal/warmup_cuda.h -- include path used by nvblox_node_main.cpp:19 / fuser_node_main.cpp:19; warmupCuda()
// (fuser_node_main.cpp:38) lives beside the stream types.
#pragma once
#include "nvblox/core/cuda_stream.h"
