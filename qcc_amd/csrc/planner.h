// planner.h -- host-side gate records, classification and the sweep planner.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/qcc_hip.h"

namespace qh {

// One submitted gate in PHYSICAL bit positions.
struct GateRec {
  uint64_t ctl_mask;  // all these index bits must be 1
  int tgt;            // target bit
  double g[8];        // row-major 2x2, (re,im) pairs
};

// The boundary only ever sees the four matrix entries: classify by
// exact zeros, which the reference's constructors produce (ops.py:110-207).
inline bool is_diag(const double g[8]) { return g[2] == 0.0 && g[3] == 0.0 && g[4] == 0.0 && g[5] == 0.0; }

}  // namespace qh
