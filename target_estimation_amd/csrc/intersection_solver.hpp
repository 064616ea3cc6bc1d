// intersection_solver.hpp -- the reference's IntersectionSolver as an object (include/target_estimation/
// intersection_solver.hpp:56-126, src/intersection_solver.cpp:19-124): a handle on a TargetManager plus ONE convergence
// gate -- two MovingAvgFilter (utils.hpp:206-265) over the position / angle distance between consecutive intersection
// poses and the previous pose -- shared by every id queried through the object, exactly as the reference keeps them as
// members.  The queries themselves run on the GPU through the manager (te_quartic.hpp, kf_aux.hpp); the gate is a dozen
// flops per call and lives on the host.  (The batched calls of target_batch_c.h keep one gate PER TARGET on the device
// instead -- a different, documented semantic for populations; this class is the drop-in for the reference's object.)
#pragma once
#include <vector>

namespace te {

class TargetManager;

class IntersectionSolver {
 public:
  IntersectionSolver(TargetManager* manager, unsigned filters_length = 250);   // intersection_solver.cpp:19-40
  // intersection_solver.cpp:42-89: time from t1 to the first crossing of the sphere, -1 if none / unknown id
  double getIntersectionTimeWithSphere(unsigned id, double t1, const double* origin, double radius);
  // intersection_solver.cpp:91-124: pose7 = pose at t1 + delta ([0 0 0 0 0 0 1] if none); returns CONVERGED
  bool getIntersectionPoseWithSphere(unsigned id, double t1, double pos_th, double ang_th, const double* origin, double radius,
                                     double* pose7);
  // filtered errors of the last call that found an intersection (for inspection; not in the reference's interface)
  double lastPositionErrorFiltered() const { return last_pf_; }
  double lastAngleErrorFiltered() const { return last_af_; }

 private:
  struct MovingAvg {   // MovingAvgFilter, utils.hpp:206-265
    explicit MovingAvg(unsigned n) : window(n ? n : 1, 0.0) {}
    double update(double value);
    std::vector<double> window;
    double sum = 0.0;
    unsigned idx = 0;
    bool complete = false;
  };
  TargetManager* m_;
  MovingAvg pos_, ang_;
  double prev_[7];
  double last_pf_ = 0.0, last_af_ = 0.0;
};

}  // namespace te
