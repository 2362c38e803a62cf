// nvblox/utils/delays.h -- timing::Delays: per-tag mean of (time the data is handled - time it was stamped), fed by every
// sensor callback with nanosecond nvblox::Time values (nvblox_node.cpp:474-477,496,520,543,562) and printed at shutdown (:180).
#pragma once
#include <cstdint>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include "nvblox/core/types.h"

namespace nvblox {
namespace timing {

class Delays {
 public:
  static void tick(const std::string& tag, const Time& stamp, const Time& now) {
    std::lock_guard<std::mutex> l(mutex());
    auto& a = table()[tag];
    const double d = (double)(static_cast<int64_t>(now) - static_cast<int64_t>(stamp));
    a.sum += d; a.n++; if (d > a.max) a.max = d; a.last = d;
  }
  static double getMeanDelaySeconds(const std::string& tag) {      // stamps are nanoseconds
    std::lock_guard<std::mutex> l(mutex());
    auto it = table().find(tag);
    return (it == table().end() || !it->second.n) ? 0.0 : it->second.sum / (double)it->second.n * 1e-9;
  }
  static std::string Print() {
    std::lock_guard<std::mutex> l(mutex());
    std::ostringstream o; o << "NVBlox Delays (in ms)\nnamespace/tag - NumSamples - Mean - Max\n-----------\n";
    for (auto& kv : table()) o << kv.first << "\t" << kv.second.n << "\t" << (kv.second.n ? kv.second.sum / (double)kv.second.n * 1e-6 : 0.0) << "\t" << kv.second.max * 1e-6 << "\n";
    return o.str();
  }
  static void Reset() { std::lock_guard<std::mutex> l(mutex()); table().clear(); }
 private:
  struct Acc { double sum = 0.0, max = 0.0, last = 0.0; int64_t n = 0; };
  static std::map<std::string, Acc>& table() { static std::map<std::string, Acc> t; return t; }
  static std::mutex& mutex() { static std::mutex m; return m; }
};

}  // namespace timing
}  // namespace nvblox
