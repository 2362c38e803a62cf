// nvblox/utils/timing.h -- scoped wall-clock timers with the reference's hierarchical string tags
// (nvblox_node.cpp:72-75,587-588,976-977; README.md:69,78,87,96 quotes the core tags tsdf/integrate, color/integrate,
// esdf/integrate, mesh/integrate -- Mapper uses exactly those).
#pragma once
#include <chrono>
#include <map>
#include <mutex>
#include <sstream>
#include <string>

namespace nvblox {
namespace timing {

class Timing {
 public:
  struct Acc { double total_s = 0.0; int64_t n = 0; };
  static std::map<std::string, Acc>& table() { static std::map<std::string, Acc> t; return t; }
  static std::mutex& mutex() { static std::mutex m; return m; }
  static void add(const std::string& tag, double s) { std::lock_guard<std::mutex> l(mutex()); auto& a = table()[tag]; a.total_s += s; a.n++; }
  static std::string Print() {
    std::lock_guard<std::mutex> l(mutex());
    std::ostringstream o; o << "NVBlox Timing\n-----------\n";
    for (auto& kv : table()) o << kv.first << "\t" << kv.second.n << "\t" << kv.second.total_s << " s\t(" << (kv.second.n ? kv.second.total_s / kv.second.n * 1e3 : 0.0) << " ms avg)\n";
    return o.str();
  }
  static void Reset() { std::lock_guard<std::mutex> l(mutex()); table().clear(); }
  static double GetMeanSeconds(const std::string& tag) { std::lock_guard<std::mutex> l(mutex()); auto it = table().find(tag); return (it == table().end() || !it->second.n) ? 0.0 : it->second.total_s / it->second.n; }
};

class Timer {
 public:
  explicit Timer(const std::string& tag, bool start_stopped = false) : tag_(tag) { if (!start_stopped) Start(); }
  ~Timer() { if (running_) Stop(); }
  void Start() { t0_ = std::chrono::steady_clock::now(); running_ = true; }
  void Stop() { if (!running_) return; Timing::add(tag_, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count()); running_ = false; }
 private:
  std::string tag_;
  std::chrono::steady_clock::time_point t0_;
  bool running_ = false;
};

}  // namespace timing
}  // namespace nvblox
