"""MI355X-native nvblox_core hot path (TSDF / Color / ESDF-2D / Mesh) behind the nvblox::Mapper API.

Python here is only the test/bench harness over the C-ABI in include/nvblox_hip.h; the product is
csrc/ -> libnvblox_hip.so plus the C++ facade headers in include/nvblox/.
"""
