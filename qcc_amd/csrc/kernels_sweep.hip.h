// kernels_sweep.hip.h -- fused register-tile sweeps (placeholder until the
// sweep kernel lands; QH_FUSE_SWEEP currently degrades to per-gate launches).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "planner.h"

namespace qh {
struct SweepBuffers {};
inline void free_sweep_buffers(SweepBuffers *) {}
inline bool sweep_supported(int) { return false; }
inline int run_fused(const std::vector<GateRec> &, int, uint64_t, int, void *, hipStream_t, bool,
                     SweepBuffers *, qh_stats *, std::string *) { return QH_ERR_ARG; }
inline std::string plan_to_json(const std::vector<GateRec> &, int, uint64_t) { return "{\"sweeps\":[]}"; }
}  // namespace qh
