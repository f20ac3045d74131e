// Optimization/Base/Concepts.h -- drop-in for the reference header of the same path
// (reference: include/Optimization/Base/Concepts.h:37-88): the objective alias and the
// parameter / result records shared by every optimizer in the library.
//
// Part of the MI355X-native re-implementation; written from scratch against the reference's
// public interface (same names, same defaults, same field order).
#pragma once

#include <cstddef>
#include <functional>
#include <limits>
#include <vector>

namespace Optimization {

// f : M x Args... -> Scalar                                    (reference Base/Concepts.h:37-38)
template <typename Variable, typename Scalar = double, typename... Args>
using Objective = std::function<Scalar(const Variable &X, Args &...args)>;

// Termination / logging knobs common to all iterative methods  (reference Base/Concepts.h:42-60)
struct OptimizerParams {
  size_t max_iterations = 100;
  double max_computation_time = std::numeric_limits<double>::max();  // seconds
  bool log_iterates = false;  // keep every iterate in OptimizerResult::iterates
  bool verbose = false;       // print one line per (outer) iteration
  size_t precision = 3;       // digits after the decimal point in verbose output
};

// What every optimizer hands back                               (reference Base/Concepts.h:64-88)
template <typename Variable, typename Scalar = double>
struct OptimizerResult {
  Variable x;                            // final estimate
  Scalar f;                              // objective at x
  double elapsed_time;                   // seconds (millisecond resolution, see Util/Stopwatch.h)
  std::vector<Scalar> objective_values;  // one per started iteration + the final value
  std::vector<double> time;              // elapsed time at the start of each iteration
  std::vector<Variable> iterates;        // only if log_iterates
};

}  // namespace Optimization
