// nvblox/utils/rates.h -- timing::Rates: per-tag call rates over a sliding window of tick timestamps, on a clock the caller
// can replace (nvblox_node.cpp:72-75 hands in the ROS clock so that rates follow simulation time), ticked in every callback
// (:469,492,515,539,559,588) and printed at shutdown / by the save_rates service (:179,667).
#pragma once
#include <chrono>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <sstream>
#include <string>

namespace nvblox {
namespace timing {

class Rates {
 public:
  static constexpr size_t kWindow = 100;                 // ticks kept per tag
  using TimestampFunctor = std::function<uint64_t()>;     // nanoseconds
  static void setGetTimestampFunctor(TimestampFunctor f) { std::lock_guard<std::mutex> l(mutex()); clock() = std::move(f); }
  static void tick(const std::string& tag) {
    std::lock_guard<std::mutex> l(mutex());
    auto& q = table()[tag];
    q.push_back(now());
    if (q.size() > kWindow) q.pop_front();
  }
  // mean rate over the window; 0 until a tag has two ticks (or if the clock did not advance)
  static float getMeanRateHz(const std::string& tag) {
    std::lock_guard<std::mutex> l(mutex());
    auto it = table().find(tag);
    return it == table().end() ? 0.0f : rateHz(it->second);
  }
  static bool exists(const std::string& tag) { std::lock_guard<std::mutex> l(mutex()); return table().count(tag) != 0; }
  static std::string Print() {
    std::lock_guard<std::mutex> l(mutex());
    std::ostringstream o; o << "NVBlox Rates (in Hz)\nnamespace/tag - NumSamples (Window Length) - Mean\n-----------\n";
    for (auto& kv : table()) o << kv.first << "\t" << kv.second.size() << "\t" << rateHz(kv.second) << "\n";
    return o.str();
  }
  static void Reset() { std::lock_guard<std::mutex> l(mutex()); table().clear(); }
 private:
  static float rateHz(const std::deque<uint64_t>& q) {
    if (q.size() < 2 || q.back() <= q.front()) return 0.0f;
    return (float)((double)(q.size() - 1) * 1e9 / (double)(q.back() - q.front()));
  }
  static uint64_t now() {
    if (clock()) return clock()();
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  static std::map<std::string, std::deque<uint64_t>>& table() { static std::map<std::string, std::deque<uint64_t>> t; return t; }
  static TimestampFunctor& clock() { static TimestampFunctor f; return f; }
  static std::mutex& mutex() { static std::mutex m; return m; }
};

}  // namespace timing
}  // namespace nvblox
