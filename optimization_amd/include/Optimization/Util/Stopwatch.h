// Optimization/Util/Stopwatch.h -- drop-in for the reference's Util/Stopwatch.h:15-29.
// tick() takes a time stamp, tock(stamp) returns the seconds since then, QUANTISED TO WHOLE
// MILLISECONDS exactly like the reference (its result.time[] / elapsed_time fields inherit that).
#pragma once

#include <chrono>

namespace Stopwatch {

using clock_type = std::chrono::high_resolution_clock;

inline std::chrono::time_point<clock_type> tick() { return clock_type::now(); }

inline double tock(const std::chrono::time_point<clock_type> &since) {
  const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(clock_type::now() - since);
  return ms.count() / 1000.0;
}

}  // namespace Stopwatch
