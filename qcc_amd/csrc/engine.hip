// engine.hip -- C-ABI (include/qcc_hip.h) of the MI355X gate-application engine.
//
// Replaces the reference's native hot path: apply1<>/applyc<> and their CPython
// wrappers in /root/reference/src/lib/xgates.cc:23-145.  One handle owns (or
// borrows) 2^nbits amplitudes in HBM plus a HIP stream; gates are either
// launched one kernel each (QH_FUSE_OFF, kernels_gate.hip.h) or queued and
// executed as fused register-tile sweeps (QH_FUSE_SWEEP, planner.h +
// kernels_sweep.hip.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/qcc_hip.h"
#include "kernels_gate.hip.h"
#include "planner.h"
#include "kernels_sweep.hip.h"

namespace {

thread_local std::string g_err = "";

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(e_ == hipErrorOutOfMemory ? QH_ERR_NOMEM : QH_ERR_HIP, "%s: %s (%s:%d)", \
                  #expr, hipGetErrorString(e_), __FILE__, __LINE__);                   \
  } while (0)

constexpr int kRedBlocks = 1024;

}  // namespace

struct qh_state_s {
  int nloc = 0, nglob = 0, bw = 128, device = 0;
  uint64_t shard = 0;
  void *d_psi = nullptr;
  void *host_psi = nullptr;    // qh_create_host_mapped: the pinned, GPU-visible host allocation d_psi points into
  void *d_alt = nullptr;       // second buffer of the same size: target of relayout sweeps (lazily allocated)
  int relayout = -1;           // -1 undecided, 0 off (attached memory, no room, QH_RELAYOUT=0), 1 on
  bool owns_mem = false, owns_stream = false, dry = false;
  bool poisoned = false;       // a sweep of a flush failed after others had run: the amplitudes are undefined until re-initialised
  hipStream_t stream = nullptr;
  int fusion = QH_FUSE_OFF;
  int perm[64];  // physical bit of logical bit
  std::vector<qh::GateRec> queue;
  qh_stats stats{};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> laps;   // qh_timer_lap: events on the stream, read back by qh_timer_laps
  size_t laps_used = 0;
  double *d_red = nullptr;     // kRedBlocks doubles
  uint64_t *d_redi = nullptr;  // kRedBlocks u64
  qh::SweepBuffers sweep;      // device/pinned op buffers for fused sweeps
  uint64_t *d_tmax = nullptr;  // per-unit maxima the last sweep of a flush leaves for qh_argmax (SweepParams::tilemax)
  uint64_t tmax_cap = 0;       // entries
  qh::Comm *comm = nullptr;    // multi-GPU exchange (exchange.hip.h)
  uint64_t amp_bytes() const { return bw == 128 ? 16 : 8; }
  uint64_t local_mask() const { return nloc >= 64 ? ~0ull : ((1ull << nloc) - 1ull); }
};

namespace {

template <typename R> qh::Gate2<R> to_gate(const double g[8]) {
  qh::Gate2<R> o;
  o.g0r = (R)g[0]; o.g0i = (R)g[1]; o.g1r = (R)g[2]; o.g1i = (R)g[3];
  o.g2r = (R)g[4]; o.g2i = (R)g[5]; o.g3r = (R)g[6]; o.g3i = (R)g[7];
  return o;
}

qh::BitIns make_ins(uint64_t ones_mask, int zero_pos) {
  qh::BitIns ins{};
  ins.n = 0;
  ins.ones = ones_mask;
  for (int b = 0; b < 64; ++b) {
    if (((ones_mask >> b) & 1ull) || b == zero_pos) ins.pos[ins.n++] = b;
  }
  return ins;
}

unsigned pick_grid(uint64_t nwork, int per_block) {
  uint64_t blocks = (nwork + per_block - 1) / per_block;
  if (blocks < 1) blocks = 1;
  // HIP refuses launches of 2^32 threads or more (a per-gate kernel over a 2^33-amplitude shard
  // would be 2^24 blocks of 256): the kernels are grid-stride, so cap the grid
  if (blocks > (1ull << 23)) blocks = 1ull << 23;
  return (unsigned)blocks;
}

int env_int(const char *name, int dflt) {
  const char *s = getenv(name);
  return s ? atoi(s) : dflt;
}
// Launch shape of the per-gate kernels (tools/membench on MI355X): small blocks
// of work, one chunk per block (no grid-stride), non-temporal access.

template <typename R, int U, bool NT>
void launch_pair_u(qh_state_s *h, uint64_t nwork, int p, const qh::BitIns &ins, const double g[8],
                   uint32_t lowpred) {
  using A = typename qh::AmpT<R>::type;
  const unsigned grid = pick_grid(nwork, 256 * U);
  if (nwork % (256 * U) == 0)
    hipLaunchKernelGGL((qh::k_pair<R, U, false, NT>), dim3(grid), dim3(256), 0, h->stream,
                       (A *)h->d_psi, nwork, p, ins, to_gate<R>(g), lowpred);
  else
    hipLaunchKernelGGL((qh::k_pair<R, U, true, NT>), dim3(grid), dim3(256), 0, h->stream,
                       (A *)h->d_psi, nwork, p, ins, to_gate<R>(g), lowpred);
}
// Launch shapes of the per-gate kernels, chosen by measurement (tools/membench/pairbench.hip, profiles/r04/):
int log2_u64(uint64_t v) { int g = 0; while ((1ull << g) < v) ++g; return g; }
bool tile_fits(uint64_t nwork, int per_block) {     // full tiles only (DPP partner fetch and the block rotation need them)
  return nwork >= (uint64_t)per_block && nwork % per_block == 0 && ((nwork / per_block) & (nwork / per_block - 1)) == 0 &&
         nwork / per_block < (1ull << 31);
}

template <typename R, int U, int WPB>
void launch_pair_tile(qh_state_s *h, uint64_t nwork, int p, const qh::BitIns &ins, const double g[8], uint32_t lowpred) {
  using A = typename qh::AmpT<R>::type;
  const uint64_t blocks = nwork / (64 * U * WPB);
  hipLaunchKernelGGL((qh::k_pair_tile<R, U, WPB, 3>), dim3((unsigned)blocks), dim3(64 * WPB), 0, h->stream, (A *)h->d_psi, p, ins,
                     to_gate<R>(g), lowpred, log2_u64(blocks));
}
template <typename R, int P>
void launch_pair_line(qh_state_s *h, uint64_t namps, const qh::BitIns &ins1, const double g[8], uint32_t lowpred) {
  using A = typename qh::AmpT<R>::type;
  const uint64_t blocks = namps / (64 * 8 * 4);
  hipLaunchKernelGGL((qh::k_pair_line<R, P, 8, 4, 3>), dim3((unsigned)blocks), dim3(256), 0, h->stream, (A *)h->d_psi, ins1,
                     to_gate<R>(g), lowpred, log2_u64(blocks));
}

// `ins` enumerates pairs (a zero inserted at p); `ins1` the amplitudes of the same set (no zero at p)
template <typename R>
void launch_pair(qh_state_s *h, uint64_t nwork, int p, const qh::BitIns &ins, const qh::BitIns &ins1, const double g[8],
                 uint32_t lowpred) {
  if (p < 3 && tile_fits(2 * nwork, 64 * 8 * 4)) {
    if (p == 0) launch_pair_line<R, 0>(h, 2 * nwork, ins1, g, lowpred);
    else if (p == 1) launch_pair_line<R, 1>(h, 2 * nwork, ins1, g, lowpred);
    else launch_pair_line<R, 2>(h, 2 * nwork, ins1, g, lowpred);
    return;
  }
  if (p >= 3 && p <= 8 && tile_fits(nwork, 64 * 8 * 4)) return launch_pair_tile<R, 8, 4>(h, nwork, p, ins, g, lowpred);
  if (p >= 20 && p <= 25 && tile_fits(nwork, 64 * 16 * 2)) return launch_pair_tile<R, 16, 2>(h, nwork, p, ins, g, lowpred);
  launch_pair_u<R, 1, true>(h, nwork, p, ins, g, lowpred);      // one pair per thread, 256-thread blocks (small / ragged states too)
}

template <typename R, int U, bool NT>
void launch_diag_u(qh_state_s *h, uint64_t nwork, int sel, const qh::BitIns &ins, double f0r, double f0i,
                   double f1r, double f1i, uint32_t lowpred) {
  using A = typename qh::AmpT<R>::type;
  const unsigned grid = pick_grid(nwork, 256 * U);
  if (nwork % (256 * U) == 0)
    hipLaunchKernelGGL((qh::k_diag<R, U, false, NT>), dim3(grid), dim3(256), 0, h->stream,
                       (A *)h->d_psi, nwork, sel, ins, (R)f0r, (R)f0i, (R)f1r, (R)f1i, lowpred);
  else
    hipLaunchKernelGGL((qh::k_diag<R, U, true, NT>), dim3(grid), dim3(256), 0, h->stream,
                       (A *)h->d_psi, nwork, sel, ins, (R)f0r, (R)f0i, (R)f1r, (R)f1i, lowpred);
}
template <typename R>
void launch_diag(qh_state_s *h, uint64_t nwork, int sel, const qh::BitIns &ins, double f0r, double f0i,
                 double f1r, double f1i, uint32_t lowpred = 0) {
  using A = typename qh::AmpT<R>::type;
  if (tile_fits(nwork, 64 * 8 * 4)) {
    const uint64_t blocks = nwork / (64 * 8 * 4);
    hipLaunchKernelGGL((qh::k_diag_tile<R, 8, 4, 3>), dim3((unsigned)blocks), dim3(256), 0, h->stream, (A *)h->d_psi, sel, ins,
                       (R)f0r, (R)f0i, (R)f1r, (R)f1i, lowpred, log2_u64(blocks));
    return;
  }
  launch_diag_u<R, 2, true>(h, nwork, sel, ins, f0r, f0i, f1r, f1i, lowpred);   // (a diagonal work item is one amplitude, a pair item two)
}

// One gate, physical bit positions, one kernel.  Returns QH_* status.
int launch_single(qh_state_s *h, const qh::GateRec &r) {
  const uint64_t cm_hi = r.ctl_mask >> h->nloc;
  if ((h->shard & cm_hi) != cm_hi) {
    h->stats.gates_noop++;
    return QH_OK;
  }
  const uint64_t cm_all = r.ctl_mask & h->local_mask();
  // bits 0,1 are never skipped in the enumeration (same 64-byte half line): predicate
  const uint64_t kLow = (h->nloc > 2) ? 3ull : 0ull;
  uint32_t lowpred = (uint32_t)(cm_all & kLow);
  const uint64_t cm = cm_all & ~kLow;
  const int nc_all = __builtin_popcountll(cm_all);
  const int nc = __builtin_popcountll(cm);
  if (nc + 1 > qh::kMaxIns) return fail(QH_ERR_ARG, "too many local control bits (%d)", nc);
  const double *g = r.g;
  const bool diag = qh::is_diag(g);
  const uint64_t ab = h->amp_bytes();
  if (r.tgt >= h->nloc) {
    if (!diag)
      return fail(QH_ERR_NONLOCAL,
                  "non-diagonal gate targets physical bit %d held by the shard index "
                  "(local bits: %d); exchange first",
                  r.tgt, h->nloc);
    const bool set = (h->shard >> (r.tgt - h->nloc)) & 1ull;
    const double fr = set ? g[6] : g[0], fi = set ? g[7] : g[1];
    if (fr == 1.0 && fi == 0.0) {
      h->stats.gates_noop++;
      return QH_OK;
    }
    const uint64_t nwork = 1ull << (h->nloc - nc);
    if (!h->dry) {
      const qh::BitIns ins = make_ins(cm, -1);
      if (h->bw == 128) launch_diag<double>(h, nwork, -1, ins, 1, 0, fr, fi, lowpred);
      else launch_diag<float>(h, nwork, -1, ins, 1, 0, fr, fi, lowpred);
    }
    h->stats.kernels_launched++;
    h->stats.bytes_algorithmic += (1ull << (h->nloc - nc_all)) * ab * 2;
    h->stats.bytes_swept += nwork * ab * 2;
    return QH_OK;
  }
  if (diag) {
    const bool one_sided = (g[0] == 1.0 && g[1] == 0.0);
    if (one_sided && g[6] == 1.0 && g[7] == 0.0) {
      h->stats.gates_noop++;
      return QH_OK;  // identity
    }
    if (one_sided) {
      const bool tgt_low = ((kLow >> r.tgt) & 1ull) != 0;   // target itself inside the half line
      if (tgt_low) lowpred |= 1u << r.tgt;
      const uint64_t nwork = 1ull << (h->nloc - nc - (tgt_low ? 0 : 1));
      if (!h->dry) {
        const qh::BitIns ins = make_ins(tgt_low ? cm : (cm | (1ull << r.tgt)), -1);
        if (h->bw == 128) launch_diag<double>(h, nwork, -1, ins, 1, 0, g[6], g[7], lowpred);
        else launch_diag<float>(h, nwork, -1, ins, 1, 0, g[6], g[7], lowpred);
      }
      h->stats.bytes_algorithmic += (1ull << (h->nloc - nc_all - 1)) * ab * 2;
      h->stats.bytes_swept += nwork * ab * 2;
    } else {
      const uint64_t nwork = 1ull << (h->nloc - nc);
      if (!h->dry) {
        const qh::BitIns ins = make_ins(cm, -1);
        if (h->bw == 128) launch_diag<double>(h, nwork, r.tgt, ins, g[0], g[1], g[6], g[7], lowpred);
        else launch_diag<float>(h, nwork, r.tgt, ins, g[0], g[1], g[6], g[7], lowpred);
      }
      h->stats.bytes_algorithmic += (1ull << (h->nloc - nc_all)) * ab * 2;
      h->stats.bytes_swept += nwork * ab * 2;
    }
    h->stats.kernels_launched++;
    return QH_OK;
  }
  const uint64_t nwork = 1ull << (h->nloc - nc - 1);
  if (!h->dry) {
    const qh::BitIns ins = make_ins(cm, r.tgt), ins1 = make_ins(cm, -1);
    if (h->bw == 128) launch_pair<double>(h, nwork, r.tgt, ins, ins1, g, lowpred);
    else launch_pair<float>(h, nwork, r.tgt, ins, ins1, g, lowpred);
  }
  h->stats.kernels_launched++;
  h->stats.bytes_algorithmic += (1ull << (h->nloc - nc_all - 1)) * 2 * ab * 2;
  h->stats.bytes_swept += nwork * 2 * ab * 2;
  return QH_OK;
}

int check_launch(qh_state_s *h) {
  if (h->dry) return QH_OK;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QH_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return QH_OK;
}

// Kernels, op-buffer allocations and events of a handle belong to ITS device, whatever device
// the calling thread had current (a process may hold handles on several GPUs).
int use_device(qh_state_s *h) {
  if (h->dry) return QH_OK;
  HIP_TRY(hipSetDevice(h->device));
  return QH_OK;
}

// Host waits.  A handle whose communicator has several ranks may wait for work that depends on OTHER processes (a
// grouped send/recv completes only when every peer has posted its half): if a peer is missing, or the ranks disagree
// about a round, a plain hipStreamSynchronize never returns.  Such handles poll instead and give up after
// QH_COMM_TIMEOUT_MS (default 300 s): QH_ERR_COMM, the handle poisoned -- an error line instead of a hung job.
int comm_timeout_ms() { return env_int("QH_COMM_TIMEOUT_MS", 300000); }      // (read at every wait: tests shorten it)
// QH_COMM_WATCH_ALL=1 (tests): watch every handle that has a communicator, the one-rank loop-back included
bool watched(const qh_state_s *h) {
  if (!h->comm || h->comm->dry) return false;
  return (h->comm->nranks > 1 && !h->comm->custom) || env_int("QH_COMM_WATCH_ALL", 0) != 0;
}
// A watched wait that gives up must not leave work behind that still writes into the CALLER's memory (the readers' D2H
// copies of a few bytes, qh_download's buffer) once this call has returned: abort the RCCL communicator -- its kernels
// then end, whatever the peers do -- and let the handle's streams drain (bounded: 10 s) before the error goes up.  The
// communicator is gone afterwards (exchanges and all-reduces fail with QH_ERR_COMM); the process should tear the job down.
void abort_comm_and_drain(qh_state_s *h) {
  qh::Comm *c = h->comm;
  if (!c || c->dry) return;
  if (c->nccl) {
    (void)qh::rccl().CommAbort(c->nccl);
    c->nccl = nullptr;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (hipStream_t s : {c->xstream, c->cstream, c->pstream, h->stream}) {
    if (!s) continue;
    while (hipStreamQuery(s) == hipErrorNotReady) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) return;
      struct timespec ts = {0, 200000};
      nanosleep(&ts, nullptr);
    }
  }
  (void)hipGetLastError();
}
template <typename Query> int poll_until_done(qh_state_s *h, Query query, const char *what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    const hipError_t e = query();
    if (e == hipSuccess) return QH_OK;
    if (e != hipErrorNotReady) {
      (void)hipGetLastError();
      return fail(QH_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    }
    if (spins > 2000) {
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms > comm_timeout_ms()) {
        h->poisoned = true;
        abort_comm_and_drain(h);
        return fail(QH_ERR_COMM, "%s: rank %d of %d still waiting after %.0f s (QH_COMM_TIMEOUT_MS): a peer is missing or the "
                    "ranks disagree about an exchange; the communicator was aborted, the state of this handle is undefined", what,
                    h->comm->rank, h->comm->nranks, ms * 1e-3);
      }
      struct timespec ts = {0, 50000};
      nanosleep(&ts, nullptr);
    }
  }
}
int wait_stream(qh_state_s *h, hipStream_t s, const char *what) {
  if (!watched(h)) {
    HIP_TRY(hipStreamSynchronize(s));
    return QH_OK;
  }
  return poll_until_done(h, [&] { return hipStreamQuery(s); }, what);
}
int wait_event(qh_state_s *h, hipEvent_t e, const char *what) {
  if (!watched(h)) {
    HIP_TRY(hipEventSynchronize(e));
    return QH_OK;
  }
  return poll_until_done(h, [&] { return hipEventQuery(e); }, what);
}

// Sweeps may re-lay the state out into a second buffer (planner.h, Planner::relayout) when the handle owns its
// memory and a second buffer of the same size fits.  A handle with a communicator of more than one rank only
// does so after qh_set_relayout(h, 1) -- the ranks must agree (an exchange needs the same layout everywhere),
// and only the caller can ask them all.
bool relayout_eligible(const qh_state_s *h) {
  return env_int("QH_RELAYOUT", 1) != 0 && h->owns_mem && !h->host_psi && qh::sweep_supported(h->nloc, h->bw);
}
// what a flush would get, without allocating anything (plans shown by qh_plan_json / qh_plan_export)
bool relayout_wanted(const qh_state_s *h) {
  if (h->dry) return env_int("QH_RELAYOUT", 1) != 0;      // planner-only handles: plan what a real handle would
  if (h->relayout >= 0) return h->relayout == 1;
  return relayout_eligible(h) && !(h->comm && h->comm->nranks > 1);
}
// A state buffer.  Buffers of 4..32 GiB (QH_ALLOC_CONTIG=1: every buffer >= 64 MiB, =0: none) are asked for as
// PHYSICALLY contiguous VRAM first (hipDeviceMallocContiguous); if the driver has no contiguous range left it is plain
// hipMalloc.  Measured, not derived (DESIGN 7 "placement", profiles/r03/alloc_contiguous_*): where the driver puts a
// buffer decides +-2.5 % of a sweep's time; contiguous 8- and 16-GiB buffers land in the fast mode 11 times of 12
// (30-qubit QFT 16.54 ms mean vs 16.92 over 12 interleaved fresh processes), a contiguous 256-GiB state is 6 % slower.
// (round 5, profiles/r05/alloc_second_buffer_ab.txt: on that box plain buffers cost 1.3-1.5 ms of a 17-ms QFT, a plain SECOND
// buffer 0.5-1.2; and the seconds an allocation sometimes takes are not the contiguous search -- any allocation that follows the
// release of a 128- or 256-GiB state pays ~6.2 s of driver work, plain ones too.)
hipError_t alloc_state_buffer(void **p, size_t bytes) {
  static const int contig = env_int("QH_ALLOC_CONTIG", -1), debug = env_int("QH_ALLOC_DEBUG", 0);
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipErrorOutOfMemory;
  bool got_contig = false;
  const bool want_contig = contig < 0 ? (bytes >= (4ull << 30) && bytes <= (32ull << 30)) : (contig > 0 && bytes >= (64ull << 20));
  if (want_contig) {
    e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous);
    got_contig = e == hipSuccess;
    if (!got_contig) {
      (void)hipGetLastError();
      *p = nullptr;
    }
  }
  if (!got_contig) e = hipMalloc(p, bytes);
  if (debug)
    fprintf(stderr, "[qh alloc %zu MiB %s at %p in %.1f ms]\n", bytes >> 20, got_contig ? "contiguous" : "plain", *p,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return e;
}
bool second_buffer_fits(const qh_state_s *h, size_t bytes) {
  size_t fr = 0, total = 0;
  // (a communicator of several ranks will want its staging halves too: up to 4 x (P-1) x chunk amplitudes)
  const size_t reserve = (bytes >> 5) + (3ull << 29) + ((h->comm && h->comm->nranks > 1) ? (4ull << 30) : 0);
  return hipMemGetInfo(&fr, &total) == hipSuccess && fr > bytes + reserve;
}
bool alloc_second_buffer(qh_state_s *h) {
  if (h->d_alt) return true;
  const size_t bytes = (size_t)h->amp_bytes() << h->nloc;
  if (second_buffer_fits(h, bytes) && alloc_state_buffer(&h->d_alt, bytes) == hipSuccess)
    return true;
  (void)hipGetLastError();
  h->d_alt = nullptr;
  return false;
}
bool relayout_ready(qh_state_s *h) {
  if (h->dry) return relayout_wanted(h);
  if (h->relayout < 0) {
    h->relayout = 0;
    if (relayout_eligible(h) && !(h->comm && h->comm->nranks > 1) && alloc_second_buffer(h)) h->relayout = 1;
  }
  return h->relayout == 1;
}

bool layout_is_canonical(const qh_state_s *h) {
  int last = -1;
  for (int b = 0; b < h->nglob; ++b) {
    if (h->perm[b] >= h->nloc) continue;
    if (h->perm[b] < last) return false;
    last = h->perm[b];
  }
  return true;
}

// Brings the local index bits back into ascending logical order (one gather pass into the second
// buffer): for the entry points that hand out or take amplitudes in physical order.
int canonicalize(qh_state_s *h) {
  if (h->dry || layout_is_canonical(h)) return QH_OK;
  if (!h->d_alt) return fail(QH_ERR_ARG, "internal: permuted layout without a second buffer");
  std::vector<int> loc;                       // logical bits that live on local positions, ascending
  for (int b = 0; b < h->nglob; ++b) if (h->perm[b] < h->nloc) loc.push_back(b);
  std::vector<int> posn;                      // the local positions they occupy, ascending
  for (int b : loc) posn.push_back(h->perm[b]);
  std::sort(posn.begin(), posn.end());
  qh::BitPerm bp{};
  bp.n = h->nloc;
  for (size_t k = 0; k < loc.size(); ++k) bp.src_of_dst[posn[k]] = (uint8_t)h->perm[loc[k]];
  const uint64_t n = 1ull << h->nloc;
  const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 1ull << 22);
  if (h->bw == 128)
    hipLaunchKernelGGL(qh::k_permute_bits<double>, dim3(grid), dim3(256), 0, h->stream, (const double2 *)h->d_psi, (double2 *)h->d_alt, n, bp);
  else
    hipLaunchKernelGGL(qh::k_permute_bits<float>, dim3(grid), dim3(256), 0, h->stream, (const float2 *)h->d_psi, (float2 *)h->d_alt, n, bp);
  int rc = check_launch(h);
  if (rc) return rc;
  std::swap(h->d_psi, h->d_alt);
  for (size_t k = 0; k < loc.size(); ++k) h->perm[loc[k]] = posn[k];
  return QH_OK;
}

// Runs the queue.  On failure the gates that did NOT run stay queued (qh_pending_gates):
// a planning / allocation failure keeps the whole queue, a failed per-gate launch keeps the
// failing gate and everything after it.  Only a failed launch of a planned sweep leaves the
// state partially updated; the queue is then dropped and the error says so.
int flush_impl(qh_state_s *h, qh::SlabIO *split = nullptr, qh::TileMaxOut *tmax = nullptr) {
  int rc = use_device(h);
  if (rc) return rc;
  if (h->poisoned)
    return fail(QH_ERR_HIP, "the state of this handle is undefined: a sweep of an earlier flush failed after others had run; "
                            "re-initialise it (qh_init_basis / qh_init_product / qh_upload of the whole shard)");
  std::vector<qh::Arrival> *arr = (h->comm && !h->comm->arrivals.empty()) ? &h->comm->arrivals : nullptr;
  if (h->queue.empty()) {
    qh::wait_all_arrivals(arr, h->stream);   // whoever called touches the state next
    return QH_OK;
  }
  if (h->fusion == QH_FUSE_SWEEP && qh::sweep_supported(h->nloc, h->bw)) {
    const uint64_t launched0 = h->stats.kernels_launched;
    qh::SlabIO local;
    qh::SlabIO *io = split ? split : &local;
    io->arrivals = arr;
    io->comm = h->comm;
    const bool relay = relayout_ready(h);
    void *result = h->d_psi;
    uint8_t final_pos[64];
    rc = qh::run_fused(h->queue, h->nloc, h->shard, h->bw, h->d_psi, h->stream, h->dry,
                       &h->sweep, &h->stats, &g_err, h->comm ? io : nullptr, h->d_alt, relay, &result, final_pos,
                       h->nglob > h->nloc, tmax);
    if (rc == QH_OK && relay) {
      if (result != h->d_psi) std::swap(h->d_psi, h->d_alt);
      for (int b = 0; b < h->nglob; ++b)
        if (h->perm[b] < h->nloc) h->perm[b] = final_pos[h->perm[b]];
    }
    qh::wait_all_arrivals(arr, h->stream);   // (a flush that planned no sweep)
    if (h->comm) h->comm->stats.sweeps_overlapped += io->sweeps_overlapped;
    if (rc != QH_OK && h->stats.kernels_launched == launched0) return rc;   // nothing ran: queue kept
    if (rc == QH_OK) rc = check_launch(h);
    if (rc != QH_OK) {
      g_err += " [sweeps of this flush may have run partially: its gates were dropped and the state is undefined until re-initialised]";
      h->poisoned = true;
    }
    h->queue.clear();
    return rc;
  }
  qh::wait_all_arrivals(arr, h->stream);
  size_t done = 0;
  for (const auto &r : h->queue) {
    rc = launch_single(h, r);
    if (rc == QH_OK) rc = check_launch(h);
    if (rc) break;
    ++done;
  }
  h->queue.erase(h->queue.begin(), h->queue.begin() + done);
  return rc;
}

// on: allocate the second buffer now (false if it does not fit); off: queued gates run, the layout returns to
// canonical order, the second buffer is freed.  *actual (optional) = the resulting mode.
int set_relayout(qh_state_s *h, bool on, int *actual = nullptr) {
  if (h->dry) {
    if (actual) *actual = relayout_wanted(h) ? 1 : 0;
    return QH_OK;
  }
  int rc = use_device(h);
  if (rc) return rc;
  if (on) {
    if (h->relayout != 1) h->relayout = (relayout_eligible(h) && alloc_second_buffer(h)) ? 1 : 0;
  } else {
    rc = flush_impl(h);
    if (rc == QH_OK) rc = canonicalize(h);
    if (rc) return rc;
    if (h->d_alt) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      (void)hipFree(h->d_alt);
      h->d_alt = nullptr;
    }
    h->relayout = 0;
  }
  if (actual) *actual = h->relayout == 1 ? 1 : 0;
  return QH_OK;
}

int submit_phys(qh_state_s *h, uint64_t cmask, int tbit, const double g[8]) {
  qh::GateRec r;
  r.ctl_mask = cmask;
  r.tgt = tbit;
  memcpy(r.g, g, sizeof r.g);
  if (h->fusion != QH_FUSE_OFF) {
    // a non-diagonal gate on a shard bit can never be executed: report now.
    if (tbit >= h->nloc && !qh::is_diag(g))
      return fail(QH_ERR_NONLOCAL,
                  "non-diagonal gate targets physical bit %d held by the shard index", tbit);
    h->queue.push_back(r);
    h->stats.gates_submitted++;
    if (h->queue.size() >= 8192) return flush_impl(h);
    return QH_OK;
  }
  int rc = use_device(h);
  if (rc == QH_OK) rc = launch_single(h, r);
  if (rc == QH_OK) rc = check_launch(h);
  if (rc == QH_OK) h->stats.gates_submitted++;
  return rc;
}

int apply_logical(qh_state_s *h, uint64_t ctl_mask, int tgt_bit, const double g[8]) {
  if (!h) return fail(QH_ERR_ARG, "null handle");
  if (!g) return fail(QH_ERR_ARG, "null gate");
  if (tgt_bit < 0 || tgt_bit >= h->nglob)
    return fail(QH_ERR_BAD_QUBIT, "target bit %d out of range [0,%d)", tgt_bit, h->nglob);
  if (h->nglob < 64 && (ctl_mask >> h->nglob))
    return fail(QH_ERR_BAD_QUBIT, "control mask 0x%llx has bits >= %d",
                (unsigned long long)ctl_mask, h->nglob);
  if ((ctl_mask >> tgt_bit) & 1ull) return fail(QH_ERR_SAME_QUBIT, "control == target (bit %d)", tgt_bit);
  uint64_t pm = 0;
  for (int b = 0; b < h->nglob; ++b)
    if ((ctl_mask >> b) & 1ull) pm |= 1ull << h->perm[b];
  return submit_phys(h, pm, h->perm[tgt_bit], g);
}

// Reference control semantics incl. quirk Q7 (see oracle/xgates_oracle.c):
// maps a reference control qubit number to a logical control bit, -1 = no-op,
// -2 = error.
int ref_ctl_bit(int nbits, int ctl, int tgt_bit) {
  const long c = (long)nbits - (long)ctl - 1;
  if (c < 0) return -2;
  if (c < nbits) return (int)c;
  const long b = c - nbits;  // bit b of the block base g
  if (b > tgt_bit && b < nbits) return (int)b;
  return -1;
}

int make_events(qh_state_s *h) {
  if (!h->ev0) {
    HIP_TRY(hipEventCreate(&h->ev0));
    HIP_TRY(hipEventCreate(&h->ev1));
  }
  return QH_OK;
}

int common_init(qh_state_s *h) {
  for (int b = 0; b < 64; ++b) h->perm[b] = b;
  if (h->dry) return QH_OK;
  HIP_TRY(hipMalloc(&h->d_red, kRedBlocks * sizeof(double)));
  HIP_TRY(hipMalloc(&h->d_redi, kRedBlocks * sizeof(uint64_t)));
  return QH_OK;
}

int check_args(int nbits, int bit_width) {
  if (nbits < 1 || nbits > 40) return fail(QH_ERR_ARG, "nbits %d out of range [1,40]", nbits);
  if (bit_width != 64 && bit_width != 128)
    return fail(QH_ERR_BAD_DTYPE, "bit_width %d (want 64 or 128)", bit_width);
  return QH_OK;
}

int select_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(QH_ERR_NO_DEVICE, "no HIP device visible (%s); this engine has no CPU fallback",
                e == hipSuccess ? "count=0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(QH_ERR_ARG, "device %d of %d", device, n);
  HIP_TRY(hipSetDevice(device));
  return QH_OK;
}

}  // namespace

extern "C" {

const char *qh_last_error(void) { return g_err.c_str(); }
int qh_version(void) { return 105; }   // 100 + round: bumped whenever plans, exchange geometry or the C-ABI change

int qh_device_count(int *count) {
  if (!count) return fail(QH_ERR_ARG, "null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *count = (e == hipSuccess) ? n : 0;
  return QH_OK;
}

int qh_create(int nbits, int bit_width, int device, qh_handle *out) {
  if (!out) return fail(QH_ERR_ARG, "null out");
  int rc = check_args(nbits, bit_width);
  if (rc) return rc;
  rc = select_device(device);
  if (rc) return rc;
  auto *h = new qh_state_s;
  h->nloc = h->nglob = nbits;
  h->bw = bit_width;
  h->device = device;
  const uint64_t bytes = (1ull << nbits) * h->amp_bytes();
  hipError_t e = alloc_state_buffer(&h->d_psi, bytes);
  if (e != hipSuccess) {
    delete h;
    return fail(QH_ERR_NOMEM, "hipMalloc(%llu bytes) for %d qubits: %s", (unsigned long long)bytes,
                nbits, hipGetErrorString(e));
  }
  h->owns_mem = true;
  e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    (void)hipFree(h->d_psi);
    delete h;
    return fail(QH_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  h->owns_stream = true;
  rc = common_init(h);
  if (rc) {
    qh_destroy(h);
    return rc;
  }
  *out = h;
  return QH_OK;
}

int qh_create_host_mapped(int nbits, int bit_width, int device, qh_handle *out) {
  if (!out) return fail(QH_ERR_ARG, "null out");
  int rc = check_args(nbits, bit_width);
  if (rc) return rc;
  if (nbits > 28) return fail(QH_ERR_ARG, "host-mapped states are for small registers (<= 28 qubits): every gate crosses PCIe");
  rc = select_device(device);
  if (rc) return rc;
  void *host = nullptr, *dev = nullptr;
  const size_t bytes = (size_t)(bit_width == 128 ? 16 : 8) << nbits;
  hipError_t e = hipHostMalloc(&host, bytes, hipHostMallocMapped);
  if (e == hipSuccess) e = hipHostGetDevicePointer(&dev, host, 0);
  if (e != hipSuccess) {
    if (host) (void)hipHostFree(host);
    return fail(QH_ERR_NOMEM, "hipHostMalloc(%zu bytes, mapped): %s", bytes, hipGetErrorString(e));
  }
  memset(host, 0, bytes);
  rc = qh_attach(nbits, bit_width, device, dev, nullptr, out);
  if (rc) {
    (void)hipHostFree(host);
    return rc;
  }
  (*out)->host_psi = host;
  return QH_OK;
}

int qh_host_ptr(qh_handle h, void **host_ptr) {
  if (!h || !host_ptr) return fail(QH_ERR_ARG, "null");
  *host_ptr = h->host_psi;
  return QH_OK;
}

int qh_attach(int nbits, int bit_width, int device, void *device_ptr, void *hip_stream,
              qh_handle *out) {
  if (!out || !device_ptr) return fail(QH_ERR_ARG, "null pointer");
  int rc = check_args(nbits, bit_width);
  if (rc) return rc;
  rc = select_device(device);
  if (rc) return rc;
  auto *h = new qh_state_s;
  h->nloc = h->nglob = nbits;
  h->bw = bit_width;
  h->device = device;
  h->d_psi = device_ptr;
  if (hip_stream) {
    h->stream = (hipStream_t)hip_stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete h;
      return fail(QH_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    h->owns_stream = true;
  }
  rc = common_init(h);
  if (rc) {
    qh_destroy(h);
    return rc;
  }
  *out = h;
  return QH_OK;
}

int qh_create_dry(int nbits, int bit_width, qh_handle *out) {
  if (!out) return fail(QH_ERR_ARG, "null out");
  int rc = check_args(nbits, bit_width);
  if (rc) return rc;
  auto *h = new qh_state_s;
  h->nloc = h->nglob = nbits;
  h->bw = bit_width;
  h->dry = true;
  common_init(h);
  *out = h;
  return QH_OK;
}

int qh_destroy(qh_handle h) {
  if (!h) return QH_OK;
  if (h->dry) (void)qh_comm_destroy(h);
  if (!h->dry) {
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)qh_comm_destroy(h);
    qh::free_sweep_buffers(&h->sweep);
    if (h->d_tmax) (void)hipFree(h->d_tmax);
    if (h->d_red) (void)hipFree(h->d_red);
    if (h->d_redi) (void)hipFree(h->d_redi);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->laps) (void)hipEventDestroy(e);
    if (h->owns_mem && h->d_psi) (void)hipFree(h->d_psi);
    if (h->d_alt) (void)hipFree(h->d_alt);
    if (h->host_psi) (void)hipHostFree(h->host_psi);
    if (h->owns_stream && h->stream) (void)hipStreamDestroy(h->stream);
  }
  delete h;
  return QH_OK;
}

int qh_set_shard(qh_handle h, int nbits_global, uint64_t shard_index) {
  if (!h) return fail(QH_ERR_ARG, "null handle");
  if (nbits_global < h->nloc || nbits_global > 62)
    return fail(QH_ERR_ARG, "nbits_global %d < local %d", nbits_global, h->nloc);
  if (nbits_global - h->nloc < 64 && (shard_index >> (nbits_global - h->nloc)) != 0)
    return fail(QH_ERR_ARG, "shard index %llu does not fit %d shard bits",
                (unsigned long long)shard_index, nbits_global - h->nloc);
  h->nglob = nbits_global;
  h->shard = shard_index;
  h->sweep.plans.clear();   // cached plans were made for another shard geometry
  return QH_OK;
}

int qh_device_ptr(qh_handle h, void **ptr) {
  if (!h || !ptr) return fail(QH_ERR_ARG, "null");
  if (!h->dry) {
    // the caller will address amplitudes in physical order through this pointer for as long as it likes: run
    // what is queued, bring the layout back to canonical order, and stop re-laying the state out (a later
    // relayout sweep would swap the two buffers under the caller: ADVICE r2)
    int rc = set_relayout(h, false);
    if (rc) return rc;
  }
  *ptr = h->d_psi;
  return QH_OK;
}
int qh_stream(qh_handle h, void **stream) {
  if (!h || !stream) return fail(QH_ERR_ARG, "null");
  *stream = (void *)h->stream;
  return QH_OK;
}
int qh_nbits(qh_handle h, int *nl, int *ng) {
  if (!h) return fail(QH_ERR_ARG, "null");
  if (nl) *nl = h->nloc;
  if (ng) *ng = h->nglob;
  return QH_OK;
}

int qh_logical_to_phys(qh_handle h, uint64_t logical, uint64_t *phys) {
  if (!h || !phys) return fail(QH_ERR_ARG, "null");
  uint64_t p = 0;
  for (int b = 0; b < h->nglob; ++b)
    if ((logical >> b) & 1ull) p |= 1ull << h->perm[b];
  *phys = p;
  return QH_OK;
}
int qh_phys_to_logical(qh_handle h, uint64_t phys, uint64_t *logical) {
  if (!h || !logical) return fail(QH_ERR_ARG, "null");
  uint64_t l = 0;
  for (int b = 0; b < h->nglob; ++b)
    if ((phys >> h->perm[b]) & 1ull) l |= 1ull << b;
  *logical = l;
  return QH_OK;
}
int qh_get_bitmap(qh_handle h, int32_t *out) {
  if (!h || !out) return fail(QH_ERR_ARG, "null");
  for (int b = 0; b < h->nglob; ++b) out[b] = h->perm[b];
  return QH_OK;
}
int qh_remap_swap(qh_handle h, int a, int b) {
  if (!h) return fail(QH_ERR_ARG, "null");
  if (a < 0 || b < 0 || a >= h->nglob || b >= h->nglob)
    return fail(QH_ERR_BAD_QUBIT, "physical bits %d,%d out of range", a, b);
  int rc = flush_impl(h);
  if (rc) return rc;
  int la = -1, lb = -1;
  for (int l = 0; l < h->nglob; ++l) {
    if (h->perm[l] == a) la = l;
    if (h->perm[l] == b) lb = l;
  }
  std::swap(h->perm[la], h->perm[lb]);
  return QH_OK;
}

// An initialisation replaces every amplitude, so whatever layout relayout sweeps left behind can go: a plain handle (no
// shard bits, no communicator: nobody else has a say in its bit map) starts over in canonical order.  (A state parked by
// the API mirror's pool comes back in the layout its last circuit left; k_init_product then un-permuted every index bit by
// bit: 11.9 ms for a 30-qubit register instead of the 2.9 ms of writing it, tools/probes/single_shot_breakdown.py.)
static void reset_layout_for_init(qh_handle h) {
  if (h->nglob == h->nloc && !h->comm)
    for (int b = 0; b < 64; ++b) h->perm[b] = b;
}

int qh_init_basis(qh_handle h, uint64_t index) {
  if (!h) return fail(QH_ERR_ARG, "null handle");
  if (h->nglob < 64 && (index >> h->nglob)) return fail(QH_ERR_ARG, "basis index out of range");
  h->queue.clear();
  h->poisoned = false;
  reset_layout_for_init(h);
  if (h->dry) return QH_OK;
  HIP_TRY(hipSetDevice(h->device));
  if (h->comm) qh::wait_all_arrivals(&h->comm->arrivals, h->stream);
  uint64_t phys;
  qh_logical_to_phys(h, index, &phys);
  HIP_TRY(hipMemsetAsync(h->d_psi, 0, (1ull << h->nloc) * h->amp_bytes(), h->stream));
  if ((phys >> h->nloc) == h->shard) {
    const uint64_t li = phys & h->local_mask();
    if (h->bw == 128)
      hipLaunchKernelGGL(qh::k_set_one<double>, dim3(1), dim3(1), 0, h->stream, (double2 *)h->d_psi, li);
    else
      hipLaunchKernelGGL(qh::k_set_one<float>, dim3(1), dim3(1), 0, h->stream, (float2 *)h->d_psi, li);
  }
  return check_launch(h);
}

int qh_init_product(qh_handle h, int nfactors, const int *nq, const double *const *amps, const uint64_t *basis) {
  if (!h || !nq || nfactors < 1) return fail(QH_ERR_ARG, "null handle / no factors");
  if (nfactors > qh::kMaxFactors) return fail(QH_ERR_ARG, "too many factors (merge small ones on the host)");
  {   // every factor a basis state: the product is ONE basis state (a memset and one amplitude, not a pass of arithmetic)
    bool all_basis = basis != nullptr;
    int total = 0;
    uint64_t index = 0;
    for (int f = 0; all_basis && f < nfactors; ++f) {
      all_basis = (!amps || !amps[f]) && nq[f] >= 1 && nq[f] <= 63 && !(basis[f] >> nq[f]) && total + nq[f] <= 64;
      if (all_basis) { index = nq[f] >= 64 ? basis[f] : ((index << nq[f]) | basis[f]); total += nq[f]; }
    }
    if (all_basis && total == h->nglob) return qh_init_basis(h, index);
  }
  // (validation first: a call that fails must leave the handle -- layout, queue -- exactly as it found it)
  int total = 0;
  uint64_t entries = 0;
  for (int f = 0; f < nfactors; ++f) {
    if (nq[f] < 1 || nq[f] > 63) return fail(QH_ERR_ARG, "factor size out of range");
    total += nq[f];
    const bool is_basis = !amps || !amps[f];
    if (is_basis && !basis) return fail(QH_ERR_ARG, "basis factor without basis[]");
    if (is_basis && (basis[f] >> nq[f])) return fail(QH_ERR_ARG, "basis index out of range");
    if (!is_basis) {
      if (nq[f] > 24) return fail(QH_ERR_ARG, "table factor larger than 2^24 amplitudes");
      entries += 1ull << nq[f];
    }
  }
  if (total != h->nglob) return fail(QH_ERR_ARG, "factor sizes do not add up to the number of qubits");
  if (entries > (1ull << 25)) return fail(QH_ERR_ARG, "factor tables larger than 2^25 amplitudes");
  reset_layout_for_init(h);
  h->queue.clear();
  h->poisoned = false;
  qh::ProductSpec sp{};
  sp.nf = nfactors;
  sp.nglob = h->nglob;
  sp.identity = 1;
  for (int b = 0; b < h->nglob; ++b) {
    sp.perm[b] = (uint8_t)h->perm[b];
    if (h->perm[b] != b) sp.identity = 0;
  }
  if (h->dry) return QH_OK;
  if (h->comm) qh::wait_all_arrivals(&h->comm->arrivals, h->stream);
  std::vector<double> tab(2 * std::max<uint64_t>(entries, 1));
  uint64_t off = 0;
  int shift = h->nglob;
  for (int f = 0; f < nfactors; ++f) {   // f_0 holds the most significant qubits (np.kron order)
    shift -= nq[f];
    sp.shift[f] = (uint8_t)shift;
    sp.nq[f] = (uint8_t)nq[f];
    sp.is_basis[f] = (!amps || !amps[f]) ? 1 : 0;
    if (sp.is_basis[f]) { sp.basis[f] = basis[f]; continue; }
    sp.off[f] = (uint32_t)off;
    memcpy(&tab[2 * off], amps[f], (size_t)16 << nq[f]);
    off += 1ull << nq[f];
  }
  HIP_TRY(hipSetDevice(h->device));
  double2 *d_tab = nullptr;
  HIP_TRY(hipMalloc((void **)&d_tab, tab.size() * sizeof(double)));
  hipError_t e = hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    const uint64_t n = 1ull << h->nloc;
    const dim3 grid((unsigned)std::min<uint64_t>((n + 255) / 256, 1ull << 22)), block(256);
    const uint64_t idx_high = h->shard << h->nloc;
    if (h->bw == 128)
      hipLaunchKernelGGL(qh::k_init_product<double>, grid, block, 0, h->stream, (double2 *)h->d_psi, n, idx_high, sp, d_tab);
    else
      hipLaunchKernelGGL(qh::k_init_product<float>, grid, block, 0, h->stream, (float2 *)h->d_psi, n, idx_high, sp, d_tab);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);   // `tab` and d_tab are released below
  }
  (void)hipFree(d_tab);
  if (e != hipSuccess) return fail(QH_ERR_HIP, hipGetErrorString(e));
  return QH_OK;
}

int qh_upload(qh_handle h, const void *host, uint64_t offset, uint64_t count) {
  if (!h || !host || h->dry) return fail(QH_ERR_ARG, "null/dry");
  if (offset + count > (1ull << h->nloc)) return fail(QH_ERR_ARG, "upload range out of bounds");
  HIP_TRY(hipSetDevice(h->device));
  if (offset == 0 && count == (1ull << h->nloc)) {
    // the whole shard is replaced: a fresh start -- nothing queued is worth running, a failed flush is forgotten, and a
    // plain handle goes back to canonical order without moving a byte (like the initialisations)
    h->queue.clear();
    h->poisoned = false;
    reset_layout_for_init(h);
  }
  int rc = flush_impl(h);
  if (rc == QH_OK) rc = canonicalize(h);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync((char *)h->d_psi + offset * h->amp_bytes(), host, count * h->amp_bytes(),
                         hipMemcpyHostToDevice, h->stream));
  return wait_stream(h, h->stream, "qh_upload");
}

int qh_download(qh_handle h, void *host, uint64_t offset, uint64_t count) {
  if (!h || !host || h->dry) return fail(QH_ERR_ARG, "null/dry");
  if (offset + count > (1ull << h->nloc)) return fail(QH_ERR_ARG, "download range out of bounds");
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc == QH_OK) rc = canonicalize(h);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(host, (const char *)h->d_psi + offset * h->amp_bytes(),
                         count * h->amp_bytes(), hipMemcpyDeviceToHost, h->stream));
  return wait_stream(h, h->stream, "qh_download");
}

int qh_amplitude(qh_handle h, uint64_t logical_index, double out[2]) {
  if (!h || !out || h->dry) return fail(QH_ERR_ARG, "null/dry");
  if (h->nglob < 64 && (logical_index >> h->nglob)) return fail(QH_ERR_ARG, "index out of range");
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc) return rc;
  uint64_t phys;
  qh_logical_to_phys(h, logical_index, &phys);
  if ((phys >> h->nloc) != h->shard) return fail(QH_ERR_NONLOCAL, "amplitude %llu lives on shard %llu", (unsigned long long)logical_index, (unsigned long long)(phys >> h->nloc));
  const uint64_t li = phys & h->local_mask();
  if (h->bw == 128) {
    HIP_TRY(hipMemcpyAsync(out, (const char *)h->d_psi + li * 16, 16, hipMemcpyDeviceToHost, h->stream));
    if ((rc = wait_stream(h, h->stream, "qh_amplitude"))) return rc;
  } else {
    float f[2];
    HIP_TRY(hipMemcpyAsync(f, (const char *)h->d_psi + li * 8, 8, hipMemcpyDeviceToHost, h->stream));
    if ((rc = wait_stream(h, h->stream, "qh_amplitude"))) return rc;
    out[0] = f[0];
    out[1] = f[1];
  }
  return QH_OK;
}

int qh_apply_bits(qh_handle h, uint64_t ctl_mask, int tgt_bit, const double gate[8]) {
  return apply_logical(h, ctl_mask, tgt_bit, gate);
}

int qh_apply1(qh_handle h, int tgt, const double gate[8]) {
  if (!h) return fail(QH_ERR_ARG, "null handle");
  const int p = h->nglob - tgt - 1;  // xgates.cc:26
  if (p < 0 || p >= h->nglob)
    return fail(QH_ERR_BAD_QUBIT, "apply1: qubit %d out of range for %d qubits", tgt, h->nglob);
  return apply_logical(h, 0, p, gate);
}

int qh_applyc(qh_handle h, int ctl, int tgt, const double gate[8]) {
  if (!h) return fail(QH_ERR_ARG, "null handle");
  const int p = h->nglob - tgt - 1;  // xgates.cc:48
  if (p < 0 || p >= h->nglob)
    return fail(QH_ERR_BAD_QUBIT, "applyc: target qubit %d out of range for %d qubits", tgt, h->nglob);
  const int c = ref_ctl_bit(h->nglob, ctl, p);  // xgates.cc:49,58-59
  if (c == -2)
    return fail(QH_ERR_BAD_QUBIT, "applyc: control qubit %d out of range for %d qubits", ctl, h->nglob);
  if (c == -1) {  // out-of-range (negative) control whose predicate is never true
    h->stats.gates_submitted++;
    h->stats.gates_noop++;
    return QH_OK;
  }
  if (c == p) return fail(QH_ERR_SAME_QUBIT, "applyc: control == target (qubit %d)", tgt);
  return apply_logical(h, 1ull << c, p, gate);
}

int qh_set_fusion(qh_handle h, int level) {
  if (!h) return fail(QH_ERR_ARG, "null handle");
  if (level != QH_FUSE_OFF && level != QH_FUSE_SWEEP) return fail(QH_ERR_ARG, "fusion level %d", level);
  int rc = flush_impl(h);
  h->fusion = level;
  return rc;
}

int qh_flush(qh_handle h) {
  if (!h) return fail(QH_ERR_ARG, "null handle");
  return flush_impl(h);
}

int qh_pending_gates(qh_handle h, uint64_t *count) {
  if (!h || !count) return fail(QH_ERR_ARG, "null");
  *count = h->queue.size();
  return QH_OK;
}

int qh_discard_pending(qh_handle h) {
  if (!h) return fail(QH_ERR_ARG, "null handle");
  h->queue.clear();
  return QH_OK;
}

int qh_sync(qh_handle h) {
  if (!h) return fail(QH_ERR_ARG, "null handle");
  if (h->dry) {
    return flush_impl(h);
  }
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc) return rc;
  return wait_stream(h, h->stream, "qh_sync");
}

int qh_set_relayout(qh_handle h, int on, int *actual) {
  if (!h) return fail(QH_ERR_ARG, "null handle");
  return set_relayout(h, on != 0, actual);
}

int qh_apply_stream(qh_handle h, uint64_t count, const int32_t *ops, const double *gates) {
  if (!h || (count && (!ops || !gates))) return fail(QH_ERR_ARG, "null");
  for (uint64_t k = 0; k < count; ++k) {
    const int32_t c = ops[2 * k], t = ops[2 * k + 1];
    const int rc = c == INT32_MIN ? qh_apply1(h, t, gates + 8 * k) : qh_applyc(h, c, t, gates + 8 * k);
    if (rc) {
      g_err = "gate " + std::to_string(k) + " of the stream: " + g_err;
      return rc;
    }
  }
  return QH_OK;
}

int qh_norm2(qh_handle h, double *out) {
  if (!h || !out || h->dry) return fail(QH_ERR_ARG, "null/dry");
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(h->d_red, 0, sizeof(double), h->stream));
  const uint64_t n = 1ull << h->nloc;
  const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
  if (h->bw == 128)
    hipLaunchKernelGGL(qh::k_norm2<double>, dim3(grid), dim3(256), 0, h->stream,
                       (const double2 *)h->d_psi, n, 0ull, 0ull, h->d_red);
  else
    hipLaunchKernelGGL(qh::k_norm2<float>, dim3(grid), dim3(256), 0, h->stream,
                       (const float2 *)h->d_psi, n, 0ull, 0ull, h->d_red);
  HIP_TRY(hipMemcpyAsync(out, h->d_red, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if ((rc = wait_stream(h, h->stream, "reader"))) return rc;
  return QH_OK;
}

int qh_prob_bit_value(qh_handle h, int logical_bit, int value, double *p) {
  if (!h || !p || h->dry) return fail(QH_ERR_ARG, "null/dry");
  if (logical_bit < 0 || logical_bit >= h->nglob) return fail(QH_ERR_BAD_QUBIT, "bit %d", logical_bit);
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc) return rc;
  value = value ? 1 : 0;
  const int pb = h->perm[logical_bit];
  uint64_t mask = 0, want = 0;
  if (pb >= h->nloc) {
    if ((int)((h->shard >> (pb - h->nloc)) & 1ull) != value) {
      *p = 0.0;
      return QH_OK;
    }
  } else {
    mask = 1ull << pb;
    want = value ? mask : 0;
  }
  HIP_TRY(hipMemsetAsync(h->d_red, 0, sizeof(double), h->stream));
  const uint64_t n = 1ull << h->nloc;
  const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
  if (h->bw == 128)
    hipLaunchKernelGGL(qh::k_norm2<double>, dim3(grid), dim3(256), 0, h->stream,
                       (const double2 *)h->d_psi, n, mask, want, h->d_red);
  else
    hipLaunchKernelGGL(qh::k_norm2<float>, dim3(grid), dim3(256), 0, h->stream,
                       (const float2 *)h->d_psi, n, mask, want, h->d_red);
  HIP_TRY(hipMemcpyAsync(p, h->d_red, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if ((rc = wait_stream(h, h->stream, "reader"))) return rc;
  return QH_OK;
}

int qh_prob_bit(qh_handle h, int logical_bit, double *p1) { return qh_prob_bit_value(h, logical_bit, 1, p1); }

}  // extern "C"

// ---- argmax (state.py:60-78 maxprob: the FIRST basis state, in logical order, with the largest probability) ----------
namespace {
// logical index of a physical one: bit inv[p] of the result = bit p (inv = where each physical bit position lives logically)
struct BitMap { int n; uint8_t to[64]; };
__device__ __forceinline__ uint64_t map_bits(uint64_t v, const BitMap &m) {
  uint64_t o = 0;
  for (int p = 0; p < m.n; ++p) o |= ((v >> p) & 1ull) << m.to[p];
  return o;
}
__device__ __forceinline__ double prob_of(double2 a) { return __builtin_fma(a.y, a.y, a.x * a.x); }   // (as the sweep islands compute it)
__device__ __forceinline__ double prob_of(float2 a) { return __builtin_fma((double)a.y, (double)a.y, (double)a.x * (double)a.x); }

// per block: the largest probability and the smallest LOGICAL index that has it (identity map: the physical index)
template <typename A, bool MAPPED>
__global__ __launch_bounds__(256) void k_argmax_logical(const A *__restrict__ psi, uint64_t n, BitMap bm, double *best_p, uint64_t *best_i) {
  __shared__ double sp[256];
  __shared__ uint64_t si[256];
  // physical -> logical through five byte tables (a near-uniform state -- a QFT's output -- ties at almost every amplitude:
  // the bit-by-bit map there cost 10 ms of a 30-qubit pass)
  __shared__ uint64_t lut[MAPPED ? 5 * 256 : 1];
  if (MAPPED) {
    for (int t = 0; t < 5; ++t) lut[t * 256 + threadIdx.x] = map_bits((uint64_t)threadIdx.x << (8 * t), bm);
    __syncthreads();
  }
  auto logical = [&](uint64_t idx) -> uint64_t {
    if (!MAPPED) return idx;
    return lut[idx & 255] | lut[256 + ((idx >> 8) & 255)] | lut[512 + ((idx >> 16) & 255)] | lut[768 + ((idx >> 24) & 255)] |
           lut[1024 + ((idx >> 32) & 255)];
  };
  double bp = -1.0;
  uint64_t bi = ~0ull;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  auto take = [&](double p, uint64_t idx) {
    if (p > bp) { bp = p; bi = logical(idx); }
    else if (p == bp) { const uint64_t li = logical(idx); if (li < bi) bi = li; }
  };
  for (; i + 3 * stride < n; i += 4 * stride) {      // four loads in flight per thread
    A a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = qh::ld_amp<true>(psi + i + k * stride);
#pragma unroll
    for (int k = 0; k < 4; ++k) take(prob_of(a[k]), i + k * stride);
  }
  for (; i < n; i += stride) take(prob_of(qh::ld_amp<true>(psi + i)), i);
  sp[threadIdx.x] = bp;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (unsigned st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      const double op = sp[threadIdx.x + st];
      const uint64_t oi = si[threadIdx.x + st];
      if (op > sp[threadIdx.x] || (op == sp[threadIdx.x] && oi < si[threadIdx.x])) { sp[threadIdx.x] = op; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { best_p[blockIdx.x] = sp[0]; best_i[blockIdx.x] = si[0]; }
}

// the per-unit maxima of the last sweep (bit patterns of non-negative doubles: they order like the doubles)
__global__ __launch_bounds__(256) void k_tmax_reduce(const uint64_t *__restrict__ t, uint64_t n, uint64_t *out) {
  __shared__ uint64_t sm[256];
  uint64_t m = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) m = t[i] > m ? t[i] : m;
  sm[threadIdx.x] = m;
  __syncthreads();
  for (unsigned st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st && sm[threadIdx.x + st] > sm[threadIdx.x]) sm[threadIdx.x] = sm[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}
constexpr unsigned kTmaxIds = 64;
// units whose maximum is the global one: ids[0] = how many, ids[1..] the first kTmaxIds of them
__global__ __launch_bounds__(256) void k_tmax_collect(const uint64_t *__restrict__ t, uint64_t n, uint64_t want, unsigned long long *ids) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    if (t[i] == want) {
      const unsigned long long k = atomicAdd(ids, 1ull);
      if (k < kTmaxIds) ids[1 + k] = i;
    }
}
// one block per collected unit: the smallest logical index among its amplitudes with probability == want
template <typename A>
__global__ __launch_bounds__(256) void k_tmax_scan(const A *__restrict__ psi, const unsigned long long *__restrict__ ids, qh::BitIns ins,
                                                    uint64_t tile_mask, int tile_bits, double want, BitMap bm, unsigned long long *out) {
  const uint64_t base = qh::expand_index(ids[1 + blockIdx.x], ins);
  unsigned long long best = ~0ull;
  for (uint64_t j = threadIdx.x; j < (1ull << tile_bits); j += 256) {
    uint64_t off = 0, jj = j;
    for (uint64_t m = tile_mask; m; m &= m - 1) { off |= (jj & 1ull) << __builtin_ctzll(m); jj >>= 1; }
    const uint64_t idx = base | off;
    if (prob_of(qh::ld_amp<false>(psi + idx)) == want) {
      const unsigned long long li = map_bits(idx, bm);
      best = li < best ? li : best;
    }
  }
  if (best != ~0ull) atomicMin(out, best);
}

// the smallest logical index in [l0, l1) whose amplitude has probability == want (l2p: logical -> physical)
template <typename A>
__global__ __launch_bounds__(256) void k_prefix_scan(const A *__restrict__ psi, uint64_t l0, uint64_t l1, double want, BitMap l2p,
                                                      unsigned long long *out) {
  unsigned long long best = ~0ull;
  for (uint64_t l = l0 + (uint64_t)blockIdx.x * 256 + threadIdx.x; l < l1; l += (uint64_t)gridDim.x * 256)
    if (prob_of(qh::ld_amp<false>(psi + map_bits(l, l2p))) == want) { best = l; break; }     // (ascending per thread: the first hit is its smallest)
  if (best != ~0ull) atomicMin(out, best);
}

// The last sweep of the flush left per-unit maxima: the largest of them is the answer's probability; the answer's index is the
// smallest logical index that has it.  Few units hold it (a peaked state: Grover, most algorithms' outputs): those units are
// scanned.  Many do (a flat state: a QFT's output ties at a tenth of its amplitudes): the first of them in LOGICAL order is
// near the start -- growing prefixes of the logical index range are scanned until one has a hit.  Exact either way.
int argmax_from_tilemax(qh_state_s *h, const qh::TileMaxOut &tm, const BitMap &bm, uint64_t *logical, double *prob, bool *found) {
  *found = false;
  const unsigned grid = (unsigned)std::min<uint64_t>((tm.nunits + 255) / 256, kRedBlocks);
  hipLaunchKernelGGL(k_tmax_reduce, dim3(grid), dim3(256), 0, h->stream, (const uint64_t *)tm.buf, tm.nunits, h->d_redi);
  std::vector<uint64_t> part(grid);
  HIP_TRY(hipMemcpyAsync(part.data(), h->d_redi, grid * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
  int rc = wait_stream(h, h->stream, "reader");
  if (rc) return rc;
  uint64_t top = 0;
  for (uint64_t v : part) top = std::max(top, v);
  double want;
  memcpy(&want, &top, 8);
  unsigned long long *ids = (unsigned long long *)h->d_redi;      // (kRedBlocks u64: room for 1 + kTmaxIds + the result)
  static_assert(kRedBlocks >= 2 + kTmaxIds, "reduction scratch");
  HIP_TRY(hipMemsetAsync(ids, 0, 8, h->stream));
  HIP_TRY(hipMemsetAsync(ids + 1 + kTmaxIds, 0xff, 8, h->stream));
  hipLaunchKernelGGL(k_tmax_collect, dim3(grid), dim3(256), 0, h->stream, (const uint64_t *)tm.buf, tm.nunits, top, ids);
  unsigned long long cnt = 0;
  HIP_TRY(hipMemcpyAsync(&cnt, ids, 8, hipMemcpyDeviceToHost, h->stream));
  if ((rc = wait_stream(h, h->stream, "reader"))) return rc;
  if (cnt == 0) return QH_OK;
  if (cnt > kTmaxIds) {
    BitMap l2p{};
    l2p.n = h->nloc;
    for (int p = 0; p < h->nloc; ++p) l2p.to[bm.to[p]] = (uint8_t)p;       // (bm: physical -> logical, a permutation of the local bits)
    const uint64_t n = 1ull << h->nloc;
    uint64_t l0 = 0;
    for (uint64_t l1 = std::min<uint64_t>(n, 1ull << 16); l0 < n; l0 = l1, l1 = std::min<uint64_t>(n, l1 << 6)) {
      const unsigned g = (unsigned)std::min<uint64_t>((l1 - l0 + 255) / 256, 1u << 16);
      if (h->bw == 128)
        hipLaunchKernelGGL(k_prefix_scan<double2>, dim3(g), dim3(256), 0, h->stream, (const double2 *)h->d_psi, l0, l1, want, l2p, ids + 1 + kTmaxIds);
      else
        hipLaunchKernelGGL(k_prefix_scan<float2>, dim3(g), dim3(256), 0, h->stream, (const float2 *)h->d_psi, l0, l1, want, l2p, ids + 1 + kTmaxIds);
      unsigned long long hit = ~0ull;
      HIP_TRY(hipMemcpyAsync(&hit, ids + 1 + kTmaxIds, 8, hipMemcpyDeviceToHost, h->stream));
      if ((rc = wait_stream(h, h->stream, "reader"))) return rc;
      if (hit != ~0ull) {
        *logical = hit;
        *prob = want;
        *found = true;
        return QH_OK;
      }
    }
    return QH_OK;      // (cannot happen)
  }
  if (h->bw == 128)
    hipLaunchKernelGGL(k_tmax_scan<double2>, dim3((unsigned)cnt), dim3(256), 0, h->stream, (const double2 *)h->d_psi, ids, tm.ins, tm.tile_mask,
                       __builtin_popcountll(tm.tile_mask), want, bm, ids + 1 + kTmaxIds);
  else
    hipLaunchKernelGGL(k_tmax_scan<float2>, dim3((unsigned)cnt), dim3(256), 0, h->stream, (const float2 *)h->d_psi, ids, tm.ins, tm.tile_mask,
                       __builtin_popcountll(tm.tile_mask), want, bm, ids + 1 + kTmaxIds);
  unsigned long long li = ~0ull;
  HIP_TRY(hipMemcpyAsync(&li, ids + 1 + kTmaxIds, 8, hipMemcpyDeviceToHost, h->stream));
  if ((rc = wait_stream(h, h->stream, "reader"))) return rc;
  if (li == ~0ull) return QH_OK;       // (cannot happen: a unit's maximum is one of its amplitudes) -> the full pass decides
  *logical = li;
  *prob = want;
  *found = true;
  return QH_OK;
}
}  // namespace

extern "C" int qh_argmax(qh_handle h, uint64_t *phys_index, double *prob) {
  if (!h || !phys_index || !prob || h->dry) return fail(QH_ERR_ARG, "null/dry");
  HIP_TRY(hipSetDevice(h->device));
  // the flush in front of the reader: its last sweep may leave the maximum of every unit it stores (no communicator)
  qh::TileMaxOut tm;
  if (h->fusion == QH_FUSE_SWEEP && !h->comm && !h->queue.empty() && qh::sweep_supported(h->nloc, h->bw) &&
      h->nloc >= qh::kLaneBits + 3 &&      // (sweep_supported() admits nloc = kLaneBits + 2: no shift by a negative count below)
      env_int("QH_FUSED_ARGMAX", 1) != 0) {
    const uint64_t need = 1ull << (h->nloc - qh::kLaneBits - 3);       // units of a three-register-bit tile (a sweep without dense gates); plans
                                                                       // with dense gates use five: a flush that wants more entries takes the full pass
    if (h->tmax_cap < need) {
      if (h->d_tmax) (void)hipFree(h->d_tmax);
      h->d_tmax = nullptr;
      h->tmax_cap = 0;
      if (hipMalloc((void **)&h->d_tmax, need * 8) == hipSuccess && h->d_tmax) h->tmax_cap = need;
      else { h->d_tmax = nullptr; (void)hipGetLastError(); }
    }
    tm.buf = h->d_tmax;
    tm.cap = h->tmax_cap;
  }
  int rc = flush_impl(h, nullptr, tm.buf ? &tm : nullptr);
  if (rc) return rc;
  // ties go to the smallest LOGICAL index, whatever layout relayout sweeps have left the state in
  BitMap bm{};
  bm.n = h->nloc;
  bool mapped = false;
  for (int b = 0; b < h->nglob; ++b)
    if (h->perm[b] < h->nloc) { bm.to[h->perm[b]] = (uint8_t)b; mapped |= h->perm[b] != b; }
  uint64_t logical = 0;
  bool found = false;
  if (tm.valid) {
    rc = argmax_from_tilemax(h, tm, bm, &logical, prob, &found);
    if (rc) return rc;
  }
  if (!found) {
    const uint64_t n = 1ull << h->nloc;
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, kRedBlocks);
    if (h->bw == 128) {
      if (mapped) hipLaunchKernelGGL((k_argmax_logical<double2, true>), dim3(grid), dim3(256), 0, h->stream, (const double2 *)h->d_psi, n, bm, h->d_red, h->d_redi);
      else hipLaunchKernelGGL((k_argmax_logical<double2, false>), dim3(grid), dim3(256), 0, h->stream, (const double2 *)h->d_psi, n, bm, h->d_red, h->d_redi);
    } else {
      if (mapped) hipLaunchKernelGGL((k_argmax_logical<float2, true>), dim3(grid), dim3(256), 0, h->stream, (const float2 *)h->d_psi, n, bm, h->d_red, h->d_redi);
      else hipLaunchKernelGGL((k_argmax_logical<float2, false>), dim3(grid), dim3(256), 0, h->stream, (const float2 *)h->d_psi, n, bm, h->d_red, h->d_redi);
    }
    std::vector<double> bp(grid);
    std::vector<uint64_t> bi(grid);
    HIP_TRY(hipMemcpyAsync(bp.data(), h->d_red, grid * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(bi.data(), h->d_redi, grid * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    if ((rc = wait_stream(h, h->stream, "reader"))) return rc;
    double best = -1.0;
    uint64_t idx = 0;
    for (unsigned k = 0; k < grid; ++k)
      if (bp[k] > best || (bp[k] == best && bi[k] < idx)) { best = bp[k]; idx = bi[k]; }
    logical = idx;
    *prob = best;
  }
  // the caller gets the PHYSICAL index of the shard-local amplitude (qh_phys_to_logical turns it back)
  uint64_t phys = 0;
  for (int b = 0; b < h->nglob; ++b)
    if (h->perm[b] < h->nloc && ((logical >> b) & 1ull)) phys |= 1ull << h->perm[b];
  *phys_index = (h->shard << h->nloc) | phys;
  return QH_OK;
}

extern "C" {
int qh_scale(qh_handle h, double re, double im) {
  if (!h || h->dry) return fail(QH_ERR_ARG, "null/dry");
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc) return rc;
  qh::BitIns ins{};
  const uint64_t nwork = 1ull << h->nloc;
  if (h->bw == 128) launch_diag<double>(h, nwork, -1, ins, 1, 0, re, im);
  else launch_diag<float>(h, nwork, -1, ins, 1, 0, re, im);
  return check_launch(h);
}

int qh_project_bit(qh_handle h, int logical_bit, int value) {
  if (!h || h->dry) return fail(QH_ERR_ARG, "null/dry");
  if (logical_bit < 0 || logical_bit >= h->nglob) return fail(QH_ERR_BAD_QUBIT, "bit %d", logical_bit);
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc) return rc;
  const int pb = h->perm[logical_bit];
  value = value ? 1 : 0;
  if (pb >= h->nloc) {
    if ((int)((h->shard >> (pb - h->nloc)) & 1ull) != value)
      HIP_TRY(hipMemsetAsync(h->d_psi, 0, (1ull << h->nloc) * h->amp_bytes(), h->stream));
    return QH_OK;
  }
  qh::BitIns ins = make_ins(value ? 0ull : (1ull << pb), value ? pb : -1);
  const uint64_t nwork = 1ull << (h->nloc - 1);
  const unsigned grid = (unsigned)std::min<uint64_t>((nwork + 255) / 256, 1u << 20);
  if (h->bw == 128)
    hipLaunchKernelGGL(qh::k_project<double>, dim3(grid), dim3(256), 0, h->stream, (double2 *)h->d_psi, nwork, ins);
  else
    hipLaunchKernelGGL(qh::k_project<float>, dim3(grid), dim3(256), 0, h->stream, (float2 *)h->d_psi, nwork, ins);
  return check_launch(h);
}

int qh_get_stats(qh_handle h, qh_stats *out) {
  if (!h || !out) return fail(QH_ERR_ARG, "null");
  *out = h->stats;
  return QH_OK;
}
int qh_reset_stats(qh_handle h) {
  if (!h) return fail(QH_ERR_ARG, "null");
  h->stats = qh_stats{};
  return QH_OK;
}

int qh_timer_begin(qh_handle h) {
  if (!h || h->dry) return fail(QH_ERR_ARG, "null/dry");
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc) return rc;
  rc = make_events(h);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  return QH_OK;
}
int qh_timer_end(qh_handle h, float *ms) {
  if (!h || !ms || h->dry) return fail(QH_ERR_ARG, "null/dry");
  if (!h->ev0) return fail(QH_ERR_ARG, "qh_timer_end without qh_timer_begin");
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  if ((rc = wait_event(h, h->ev1, "qh_timer_end"))) return rc;
  HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return QH_OK;
}

int qh_timer_lap(qh_handle h) {
  if (!h || h->dry) return fail(QH_ERR_ARG, "null/dry");
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc) return rc;
  if (h->laps_used == h->laps.size()) {
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreate(&e));
    h->laps.push_back(e);
  }
  HIP_TRY(hipEventRecord(h->laps[h->laps_used++], h->stream));
  return QH_OK;
}
int qh_timer_laps(qh_handle h, float *ms, int cap, int *count) {
  if (!h || !count || h->dry) return fail(QH_ERR_ARG, "null/dry");
  HIP_TRY(hipSetDevice(h->device));
  const int n = h->laps_used > 0 ? (int)h->laps_used - 1 : 0;
  if (h->laps_used) {
    const int rc = wait_event(h, h->laps[h->laps_used - 1], "qh_timer_laps");
    if (rc) return rc;
  }
  for (int k = 0; k < n && ms && k < cap; ++k) HIP_TRY(hipEventElapsedTime(&ms[k], h->laps[k], h->laps[k + 1]));
  *count = n;
  h->laps_used = 0;
  return QH_OK;
}

int qh_plan_json(qh_handle h, char *buf, uint64_t cap, uint64_t *needed) {
  if (!h) return fail(QH_ERR_ARG, "null");
  std::string s = qh::plan_to_json(h->queue, h->nloc, h->shard, h->bw, qh::sweep_max_rb(),
                                   qh::sweep_split_lanes(), relayout_wanted(h), h->nglob > h->nloc);
  if (needed) *needed = s.size() + 1;
  if (buf && cap) {
    const uint64_t n = std::min<uint64_t>(cap - 1, s.size());
    memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return QH_OK;
}

int qh_plan_export(qh_handle h, void *buf, uint64_t cap, uint64_t *needed) {
  if (!h) return fail(QH_ERR_ARG, "null");
  if (!qh::sweep_supported(h->nloc, h->bw)) return fail(QH_ERR_ARG, "state too small for sweeps");
  qh::PlanResult pr = qh::plan_best(h->queue, h->nloc, h->shard, h->bw, qh::sweep_max_rb(),
                                    qh::sweep_split_lanes(), relayout_wanted(h), h->nglob > h->nloc);
  std::vector<uint64_t> out;
  auto put_bytes = [&](const void *p, size_t n) {
    const size_t w = (n + 7) / 8, at = out.size();
    out.resize(at + w, 0);
    if (n) memcpy(&out[at], p, n);
  };
  out.push_back(0x51485033ull);
  out.push_back(pr.sweeps.size());
  out.push_back(pr.noop_gates);
  put_bytes(pr.final_pos, 64);
  for (auto &sp : pr.sweeps) {
    int64_t hdr[28] = {0};
    int k = 0;
    hdr[k++] = sp.rb;
    for (int i = 0; i < 6; ++i) hdr[k++] = sp.regpos[i];
    for (int i = 0; i < 6; ++i) hdr[k++] = sp.regpos_store[i];
    for (int i = 0; i < 3; ++i) hdr[k++] = sp.lanehi[i];
    hdr[k++] = sp.nwave;
    for (int i = 0; i < 2; ++i) hdr[k++] = sp.wavepos[i];
    hdr[k++] = (int64_t)sp.fixed_ones;
    hdr[k++] = (int64_t)sp.ntiles;
    hdr[k++] = (int64_t)sp.ops.size();
    hdr[k++] = (int64_t)sp.groups.size();
    hdr[k++] = (int64_t)sp.oterms.size();
    hdr[k++] = (int64_t)sp.tables.size();
    hdr[k++] = sp.n_ltab;
    hdr[k++] = sp.lane_low;
    hdr[k++] = sp.relayout ? 1 : 0;
    put_bytes(hdr, sizeof hdr);
    put_bytes(sp.dest_pos, 64);
    {
      // lanes: the index bit on lane bit i at load time, at store time, and the position 0..5 a relayout store sends it to;
      // then the wave bits at store time
      int64_t st[20] = {0};
      for (int i = 0; i < 6; ++i) { st[i] = sp.seat[i]; st[6 + i] = sp.seat_store[i]; st[12 + i] = sp.seat_dest[i]; }
      st[18] = sp.wavepos_store[0];
      st[19] = sp.wavepos_store[1];
      put_bytes(st, sizeof st);
      // what the kernel is handed for a relayout store: register / wave bit destinations and the runs of
      // unit-index bits (count, then 8 x mask, 8 x shift)
      int64_t kd[8 + 1 + 2 * qh::kMaxUnitSegs] = {0};
      for (int i = 0; i < 6; ++i) kd[i] = sp.reg_dest[i];
      kd[6] = sp.wave_dest[0];
      kd[7] = sp.wave_dest[1];
      uint64_t masks[qh::kMaxUnitSegs] = {0};
      int shifts[qh::kMaxUnitSegs] = {0};
      kd[8] = sp.relayout ? qh::unit_segments(sp, h->nloc, masks, shifts) : 0;
      for (int i = 0; i < qh::kMaxUnitSegs; ++i) { kd[9 + i] = (int64_t)masks[i]; kd[9 + qh::kMaxUnitSegs + i] = shifts[i]; }
      put_bytes(kd, sizeof kd);
    }
    put_bytes(sp.ops.data(), sp.ops.size() * sizeof(qh::SweepOp));
    put_bytes(sp.groups.data(), sp.groups.size() * sizeof(qh::DGroup));
    put_bytes(sp.oterms.data(), sp.oterms.size() * sizeof(qh::OTerm));
    put_bytes(sp.tables.data(), sp.tables.size() * sizeof(double));
  }
  const uint64_t bytes = out.size() * 8;
  if (needed) *needed = bytes;
  if (buf && cap) {
    if (cap < bytes) return fail(QH_ERR_ARG, "buffer too small (call with NULL to get the size)");
    memcpy(buf, out.data(), bytes);
  }
  return QH_OK;
}

}  // extern "C"

// ---- multi-GPU exchange (exchange.hip.h) --------------------------------------------------
namespace {

#define NCCL_TRY(expr)                                                                     \
  do {                                                                                     \
    ncclResult_t r_ = (expr);                                                              \
    if (r_ != ncclSuccess)                                                                 \
      return fail(QH_ERR_COMM, "%s: %s (%s:%d)", #expr, qh::rccl().GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

int comm_common_init(qh_state_s *h, int nranks, int rank) {
  if (!h || h->dry) return fail(QH_ERR_ARG, "null/dry handle");
  if (h->comm) return fail(QH_ERR_ARG, "handle already has a communicator");
  if (nranks < 1 || (nranks & (nranks - 1)) || rank < 0 || rank >= nranks)
    return fail(QH_ERR_ARG, "nranks %d must be a power of two, rank %d inside it", nranks, rank);
  if (nranks > qh::kMaxXferMoves + 1) return fail(QH_ERR_ARG, "at most %d ranks", qh::kMaxXferMoves + 1);
  if (h->bw != 128 && h->bw != 64) return fail(QH_ERR_BAD_DTYPE, "bit width");
  HIP_TRY(hipSetDevice(h->device));
  int rc0 = flush_impl(h);
  if (rc0) return rc0;
  if (nranks > 1 && h->relayout != 0) {
    // The ranks of a sharded state must hold the same layout whenever they exchange.  They do if all of them
    // re-lay out (the planner keeps rank-dependent gates as ghosts: planner.h) or none does; a rank that could
    // not get its second buffer would break that, so with several ranks relayout starts OFF and the caller
    // turns it on after every rank has said it can (qh_set_relayout; qcc_amd/sharded.py does).
    rc0 = set_relayout(h, false);
    if (rc0) return rc0;
  }
  auto *c = new qh::Comm;
  c->nranks = nranks;
  c->rank = rank;
  hipError_t e = hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->pstream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&c->t0);
  if (e == hipSuccess) e = hipEventCreate(&c->t1);
  if (e != hipSuccess) {
    delete c;
    return fail(QH_ERR_HIP, "exchange streams/events: %s", hipGetErrorString(e));
  }
  h->comm = c;
  return QH_OK;
}

// closes the timing bracket of the previous exchange (waits for it)
void close_timing(qh::Comm *c) {
  if (!c->timing_open) return;
  float ms = 0;
  if (hipEventSynchronize(c->t1) == hipSuccess && hipEventElapsedTime(&ms, c->t0, c->t1) == hipSuccess)
    c->stats.span_ms += ms;
  c->timing_open = false;
}

template <typename A>
void launch_xfer(bool unpack, void *psi, void *stage, uint64_t n, uint64_t start, const qh::XferGeom &g, hipStream_t st) {
  const uint64_t total = n * (uint64_t)g.np;
  const unsigned grid = (unsigned)std::min<uint64_t>((total + 255) / 256, 1ull << 20);
  if (unpack) hipLaunchKernelGGL(qh::k_xunpack<A>, dim3(grid), dim3(256), 0, st, (A *)psi, (const A *)stage, n, start, g);
  else hipLaunchKernelGGL(qh::k_xpack<A>, dim3(grid), dim3(256), 0, st, (const A *)psi, (A *)stage, n, start, g);
}

// Every rank must cut an exchange the same way (rounds, chunk sizes, slabs, where each index bit lives): the
// geometry is a function of rank-invariant inputs only -- the planner keeps rank-dependent gates as ghosts, so
// plans, tiles and layouts agree by construction -- and this is the check of that claim: a 64-bit signature of
// the geometry is compared with every peer's (always on the host-staged transport; on RCCL with
// QH_EXCHANGE_VERIFY=1, a host wait).  A mismatch is an error here instead of a hang or misplaced data there.
int verify_geometry(qh_state_s *h, uint64_t sig) {
  qh::Comm *c = h->comm;
  if (c->nranks <= 1) return QH_OK;
  if (c->custom) {
    const int np = c->nranks - 1;
    std::vector<uint64_t> mine(np, sig), theirs(np, 0);
    std::vector<int> peers;
    std::vector<void *> sp, rp;
    for (int j = 0, k = 0; j < c->nranks; ++j) {
      if (j == c->rank) continue;
      peers.push_back(j);
      sp.push_back(&mine[k]);
      rp.push_back(&theirs[k]);
      ++k;
    }
    if (c->custom(c->custom_user, np, peers.data(), sp.data(), rp.data(), sizeof(uint64_t)) != 0)
      return fail(QH_ERR_COMM, "host-staged transport: the round callback failed (geometry check)");
    for (int k = 0; k < np; ++k)
      if (theirs[k] != sig)
        return fail(QH_ERR_COMM, "exchange geometry differs between rank %d (%016llx) and rank %d (%016llx): the ranks "
                    "planned different sweeps or layouts", c->rank, (unsigned long long)sig, peers[k], (unsigned long long)theirs[k]);
    c->stats.geometry_checks++;
    return QH_OK;
  }
  if (!c->nccl) return fail(QH_ERR_COMM, "no RCCL communicator on this handle (never set up, or aborted by the watchdog)");
  // RCCL: every geometry this communicator has not compared yet (QH_EXCHANGE_VERIFY=1: every exchange, =0: never).
  // The all-reduce runs on the EXCHANGE stream -- it does not wait for the sweeps queued on the compute stream --
  // and the host waits for it: once per distinct geometry (a loop over one circuit cycles through a few layouts).
  static const int mode = env_int("QH_EXCHANGE_VERIFY", -1);
  if (mode == 0) return QH_OK;
  if (mode < 0 && std::find(c->verified.begin(), c->verified.end(), sig) != c->verified.end()) return QH_OK;
  if (!c->d_sig) {
    HIP_TRY(hipMalloc((void **)&c->d_sig, 4 * sizeof(double)));
    HIP_TRY(hipHostMalloc((void **)&c->h_sig, 8 * sizeof(double), hipHostMallocDefault));
  }
  double *v = c->h_sig, *w = c->h_sig + 4;
  v[0] = (double)(sig >> 32); v[1] = (double)(sig & 0xffffffffu); v[2] = -v[0]; v[3] = -v[1];
  HIP_TRY(hipMemcpyAsync(c->d_sig, v, 4 * sizeof(double), hipMemcpyHostToDevice, c->xstream));
  NCCL_TRY(qh::rccl().AllReduce(c->d_sig, c->d_sig, 4, ncclDouble, ncclMax, c->nccl, c->xstream));
  HIP_TRY(hipMemcpyAsync(w, c->d_sig, 4 * sizeof(double), hipMemcpyDeviceToHost, c->xstream));
  const int rc = wait_stream(h, c->xstream, "exchange geometry check (all-reduce of the signature)");
  if (rc) return rc;
  if (w[0] != -w[2] || w[1] != -w[3])
    return fail(QH_ERR_COMM, "exchange geometry differs between ranks (signature of rank %d: %016llx; largest / smallest halves seen: "
                "%08x%08x / %08x%08x): the ranks planned different sweeps or layouts, or run with different planner switches "
                "(QH_* environment) or builds", c->rank, (unsigned long long)sig, (unsigned)w[0], (unsigned)w[1], (unsigned)-w[2], (unsigned)-w[3]);
  if (c->verified.size() >= 256) c->verified.erase(c->verified.begin());
  c->verified.push_back(sig);
  c->stats.geometry_checks++;
  return QH_OK;
}

// What must be equal on every rank besides the geometry itself: the planner's switches and the build.
uint64_t env_build_hash() {
  std::string s = qh::planner_env_signature();
  // (the EFFECTIVE values: an unset switch and one set to its default are the same plan)
  s += std::to_string(env_int("QH_EXCHANGE_SLAB_BITS", 3)) + ";" + std::to_string(env_int("QH_EXCHANGE_PACK", -1)) + ";" +
       std::to_string(env_int("QH_RELAYOUT", 1) != 0) + ";";
  // the build: the library's version and the layouts the planner and the kernels share -- NOT the time of compilation (two
  // nodes that build the same sources in-tree must be able to exchange)
  s += "v" + std::to_string(qh_version()) + ":" + std::to_string(sizeof(qh::SweepArgs)) + ":" + std::to_string(sizeof(qh::SweepOp)) +
       ":" + std::to_string(sizeof(qh::SweepPlan)) + ":" + std::to_string(sizeof(qh_xgeom));
  uint64_t hsh = 0xcbf29ce484222325ull;
  for (unsigned char ch : s) { hsh ^= ch; hsh *= 0x100000001b3ull; }
  return hsh;
}

// The exchange proper.  `moves`: block value blk of the g LOGICAL local bits [base, base+g) goes to `peer`,
// whose data lands in block value `land`.  Logical = the bit numbers the caller uses (canonical positions);
// where those bits live now is the handle's business (relayout sweeps move them).
int do_exchange(qh_state_s *h, const std::vector<qh::BlockMove> &moves, int base, int gbits, uint64_t chunk_amps) {
  qh::Comm *c = h->comm;
  const int nloc = h->nloc;
  const uint64_t ab = h->amp_bytes();
  if (base < 0 || base + gbits > h->nglob || gbits > 8) return fail(QH_ERR_BAD_QUBIT, "exchange bits [%d,%d)", base, base + gbits);
  for (int k = 0; k < gbits; ++k)
    if (h->perm[base + k] >= nloc) return fail(QH_ERR_BAD_QUBIT, "exchange bit %d is not a local bit of this shard", base + k);
  if (moves.size() > (size_t)qh::kMaxXferMoves) return fail(QH_ERR_ARG, "too many peers");
  if (h->poisoned) return fail(QH_ERR_HIP, "the state of this handle is undefined (an earlier sweep or exchange failed); re-initialise it");
  const bool dry = h->dry;
  if (!dry) {
    HIP_TRY(hipSetDevice(h->device));
    close_timing(c);
  }
  // 1. the queued gates, the last sweep cut into slabs
  qh::SlabIO io;
  io.split_last = true;
  for (int k = 0; k < gbits; ++k) io.avoid |= 1ull << h->perm[base + k];   // (layout before the flush; run_fused follows the moves)
  io.want_bits = std::max(0, std::min(env_int("QH_EXCHANGE_SLAB_BITS", qh::kMaxSlabBits), qh::kMaxSlabBits));   // (UnitPerm::slab_pos)
  c->pool_used = 0;   // (arrivals of the previous exchange are waited for by this flush)
  const uint64_t sweeps0 = h->stats.sweeps;
  int rc = flush_impl(h, &io);
  if (rc) return rc;
  // 2. where the blocks' bits live now
  int pos[8];
  uint64_t blockbits = 0;
  for (int k = 0; k < gbits; ++k) { pos[k] = h->perm[base + k]; blockbits |= 1ull << pos[k]; }
  auto blk_off = [&](int v) {
    uint64_t o = 0;
    for (int k = 0; k < gbits; ++k) if ((v >> k) & 1) o |= 1ull << pos[k];
    return o;
  };
  uint64_t slab_mask = io.slab_mask;
  std::vector<uint64_t> slab_vals = io.slab_vals;
  if (io.slab_done.empty()) {
    slab_mask = 0;
    if (io.want_bits > 0)   // nothing was queued (or the last sweep could not be cut): slabs still let the NEXT sweep start early
      slab_mask = qh::pick_slab_bits(nloc, blockbits | 7ull, std::min(io.want_bits, std::max(0, nloc - gbits - 10)), 6);
    slab_vals.clear();
    for (int k = 0; k < (1 << qh::popc(slab_mask)); ++k) slab_vals.push_back(qh::deposit_bits((uint64_t)k, slab_mask));
  }
  const int K = (int)slab_vals.size();
  hipEvent_t all_done = nullptr;
  if (dry) {
    io.slab_done.assign(K, nullptr);
  } else if (io.slab_done.empty() || std::find(io.slab_done.begin(), io.slab_done.end(), nullptr) != io.slab_done.end()) {
    all_done = c->event();
    if (!all_done) return fail(QH_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(all_done, h->stream));
    io.slab_done.assign(K, all_done);
  }
  // 3. geometry of the rounds.  Two ways to move a round: DIRECT -- the blocks are contiguous runs of the shard,
  // sent from where they lie and copied home from the staging area -- when the low free bits give runs of a
  // whole chunk (or 16 MiB); PACKED -- a gather kernel packs each peer's amplitudes of the round into the staging area, a
  // scatter kernel puts the received ones in place -- whatever the layout (after relayout sweeps the blocks'
  // bits may sit anywhere above the 128-byte line).
  const uint64_t free_mask = h->local_mask() & ~blockbits & ~slab_mask;
  const int nfree = qh::popc(free_mask);
  const int run_bits = (~free_mask) ? __builtin_ctzll(~free_mask) : 64;
  if (!chunk_amps) chunk_amps = 1ull << 22;
  int want_bits = 0;
  while ((2ull << want_bits) <= chunk_amps && want_bits + 1 <= nfree) want_bits++;
  const int force = env_int("QH_EXCHANGE_PACK", -1);      // tests: 1 = always packed, 0 = never
  const bool packed = force >= 0 ? force != 0 : run_bits < std::min(want_bits, 20);   // direct: whole chunks, or runs of >= 16 MiB
  const int chunk_bits = packed ? want_bits : std::min(want_bits, run_bits);
  const uint64_t n = 1ull << chunk_bits;                       // amplitudes per peer and round
  const uint64_t nchunks = 1ull << (nfree - chunk_bits);
  const size_t np = moves.size();
  if (np == 0) return QH_OK;
  {
    uint64_t sig = 0x9e3779b97f4a7c15ull;
    auto mix = [&](uint64_t v) { sig ^= v + 0x9e3779b97f4a7c15ull + (sig << 6) + (sig >> 2); };
    for (int b = 0; b < h->nglob; ++b) mix((uint64_t)h->perm[b]);
    mix(slab_mask); mix((uint64_t)chunk_bits); mix(nchunks); mix(packed); mix((uint64_t)K); mix(blockbits); mix((uint64_t)base); mix((uint64_t)gbits);
    for (uint64_t v : slab_vals) mix(v);
    mix((uint64_t)np); mix((uint64_t)h->bw); mix(env_build_hash());
    qh_xgeom &G = c->last_geom;
    G.signature = sig;
    G.slab_mask = slab_mask;
    G.block_bits = blockbits;
    G.rounds_per_slab = nchunks;
    G.staging_bytes = (uint64_t)(packed ? 4 : 2) * np * n * ab;
    G.slabs = (uint32_t)K;
    G.chunk_bits = (uint32_t)chunk_bits;
    G.packed = packed ? 1 : 0;
    G.peers = (uint32_t)np;
    G.sweeps_before = (uint32_t)(h->stats.sweeps - sweeps0);
    G.last_sweep_split = io.split_done ? 1 : 0;
    if (dry) {     // planner-only handle: the decisions are on record, the next flush sees the arrivals it would see
      for (int k = 0; k < K; ++k) c->arrivals.push_back(qh::Arrival{slab_mask, slab_vals[k], nullptr});
      c->stats.exchanges++;
      c->stats.slabs += K;
      c->stats.rounds += (uint64_t)K * nchunks;
      c->stats.rounds_packed += packed ? (uint64_t)K * nchunks : 0;
      c->stats.bytes_sent += (uint64_t)np * (1ull << (nloc - gbits)) * ab;
      return QH_OK;
    }
    rc = verify_geometry(h, sig);
    if (rc) return rc;
  }
  char *psi = (char *)h->d_psi;
  qh::XferGeom xg_send{}, xg_land{};
  if (packed) {
    qh::BitIns ins{};
    for (int b = 0; b < nloc; ++b) if (!((free_mask >> b) & 1ull)) {
      if (ins.n == qh::kMaxIns) return fail(QH_ERR_ARG, "exchange: more than %d block + slab bits", qh::kMaxIns);
      ins.pos[ins.n++] = b;
    }
    xg_send.ins = xg_land.ins = ins;
    xg_send.np = xg_land.np = (int)np;
    for (size_t m = 0; m < np; ++m) { xg_send.off[m] = blk_off(moves[m].blk); xg_land.off[m] = blk_off(moves[m].land); }
  }
  // staging: [2 receive halves][2 send halves (packed only)] of (peers x chunk) amplitudes
  const size_t half = np * n * ab;
  const size_t need_stage = (packed ? 4 : 2) * half;
  if (c->staging_bytes < need_stage && (packed || !c->custom)) {
    if (c->staging) {
      HIP_TRY(hipStreamSynchronize(c->cstream));
      HIP_TRY(hipStreamSynchronize(c->pstream));
      HIP_TRY(hipStreamSynchronize(c->xstream));
      (void)hipFree(c->staging);
      c->staging = nullptr;
      c->staging_bytes = 0;
    }
    HIP_TRY(hipMalloc(&c->staging, need_stage));
    c->staging_bytes = need_stage;
  }
  auto xfer = [&](bool unpack, void *stage, uint64_t start, uint64_t sv, hipStream_t st) {
    qh::XferGeom g = unpack ? xg_land : xg_send;
    for (size_t m = 0; m < np; ++m) g.off[m] |= sv;
    if (h->bw == 128) launch_xfer<double2>(unpack, psi, stage, n, start, g, st);
    else launch_xfer<float2>(unpack, psi, stage, n, start, g, st);
  };
  if (c->custom) {
    // host-staged transport: synchronous rounds
    const size_t need = np * n * ab;
    if (c->h_bytes < need) {
      if (c->h_send) (void)hipHostFree(c->h_send);
      if (c->h_recv) (void)hipHostFree(c->h_recv);
      c->h_send = c->h_recv = nullptr;
      c->h_bytes = 0;
      HIP_TRY(hipHostMalloc(&c->h_send, need, hipHostMallocDefault));
      HIP_TRY(hipHostMalloc(&c->h_recv, need, hipHostMallocDefault));
      c->h_bytes = need;
    }
    std::vector<int> peers(np);
    std::vector<void *> sp(np), rp(np);
    for (size_t m = 0; m < np; ++m) {
      peers[m] = moves[m].peer;
      sp[m] = (char *)c->h_send + m * n * ab;
      rp[m] = (char *)c->h_recv + m * n * ab;
    }
    for (int k = 0; k < K; ++k) {
      const uint64_t sv = slab_vals[k];
      HIP_TRY(hipEventSynchronize(io.slab_done[k]));
      if (k == 0) { HIP_TRY(hipEventRecord(c->t0, c->xstream)); }
      for (uint64_t ci = 0; ci < nchunks; ++ci) {
        if (packed) {
          xfer(false, c->staging, ci << chunk_bits, sv, c->xstream);
          HIP_TRY(hipMemcpyAsync(c->h_send, c->staging, need, hipMemcpyDeviceToHost, c->xstream));
        } else {
          const uint64_t off = qh::deposit_bits(ci << chunk_bits, free_mask) | sv;
          for (size_t m = 0; m < np; ++m)
            HIP_TRY(hipMemcpyAsync(sp[m], psi + (off | blk_off(moves[m].blk)) * ab, n * ab, hipMemcpyDeviceToHost, c->xstream));
        }
        HIP_TRY(hipStreamSynchronize(c->xstream));
        if (c->custom(c->custom_user, (int)np, peers.data(), sp.data(), rp.data(), n * ab) != 0)
          return fail(QH_ERR_COMM, "host-staged transport: the round callback failed");
        if (packed) {
          HIP_TRY(hipMemcpyAsync(c->staging, c->h_recv, need, hipMemcpyHostToDevice, c->xstream));
          xfer(true, c->staging, ci << chunk_bits, sv, c->xstream);
        } else {
          const uint64_t off = qh::deposit_bits(ci << chunk_bits, free_mask) | sv;
          for (size_t m = 0; m < np; ++m)
            HIP_TRY(hipMemcpyAsync(psi + (off | blk_off(moves[m].land)) * ab, rp[m], n * ab, hipMemcpyHostToDevice, c->xstream));
        }
        HIP_TRY(hipStreamSynchronize(c->xstream));
        c->stats.rounds++;
      }
      hipEvent_t ev = c->event();
      if (!ev) return fail(QH_ERR_HIP, "hipEventCreate failed");
      HIP_TRY(hipEventRecord(ev, c->xstream));
      c->arrivals.push_back(qh::Arrival{slab_mask, sv, ev});
    }
    HIP_TRY(hipEventRecord(c->t1, c->xstream));
  } else {
    // RCCL: grouped send/recv per round on xstream, landing (copies / scatter kernel) on cstream, packing on pstream
    if (!c->nccl) return fail(QH_ERR_COMM, "no communicator (qh_comm_init)");
    const ncclDataType_t dt = h->bw == 128 ? ncclDouble : ncclFloat;
    const size_t cnt = (size_t)n * 2;
    auto &R = qh::rccl();
    hipEvent_t copied[2] = {nullptr, nullptr};   // receive half free again
    hipEvent_t sent[2] = {nullptr, nullptr};     // send half free again (packed)
    uint64_t round = 0;
    for (int k = 0; k < K; ++k) {
      const uint64_t sv = slab_vals[k];
      HIP_TRY(hipStreamWaitEvent(packed ? c->pstream : c->xstream, io.slab_done[k], 0));
      if (k == 0) { HIP_TRY(hipEventRecord(c->t0, packed ? c->pstream : c->xstream)); }
      hipEvent_t last_copy = nullptr;
      for (uint64_t ci = 0; ci < nchunks; ++ci, ++round) {
        const int par = (int)(round & 1);
        const uint64_t off = qh::deposit_bits(ci << chunk_bits, free_mask) | sv;
        char *stage = (char *)c->staging + par * half;
        char *sstage = (char *)c->staging + (2 + par) * half;
        if (packed) {
          if (sent[par]) HIP_TRY(hipStreamWaitEvent(c->pstream, sent[par], 0));
          xfer(false, sstage, ci << chunk_bits, sv, c->pstream);
          hipEvent_t pk = c->event();
          if (!pk) return fail(QH_ERR_HIP, "hipEventCreate failed");
          HIP_TRY(hipEventRecord(pk, c->pstream));
          HIP_TRY(hipStreamWaitEvent(c->xstream, pk, 0));
        }
        if (copied[par]) HIP_TRY(hipStreamWaitEvent(c->xstream, copied[par], 0));   // this half is free again
        NCCL_TRY(R.GroupStart());
        for (size_t m = 0; m < np; ++m) {
          const void *src = packed ? (const void *)(sstage + m * n * ab) : (const void *)(psi + (off | blk_off(moves[m].blk)) * ab);
          NCCL_TRY(R.Send(src, cnt, dt, moves[m].peer, c->nccl, c->xstream));
          NCCL_TRY(R.Recv(stage + m * n * ab, cnt, dt, moves[m].peer, c->nccl, c->xstream));
        }
        NCCL_TRY(R.GroupEnd());
        hipEvent_t landed = c->event();
        if (!landed) return fail(QH_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(landed, c->xstream));
        sent[par] = landed;
        HIP_TRY(hipStreamWaitEvent(c->cstream, landed, 0));
        if (packed) {
          xfer(true, stage, ci << chunk_bits, sv, c->cstream);
        } else {
          for (size_t m = 0; m < np; ++m)
            HIP_TRY(hipMemcpyAsync(psi + (off | blk_off(moves[m].land)) * ab, stage + m * n * ab, n * ab,
                                   hipMemcpyDeviceToDevice, c->cstream));
        }
        hipEvent_t cp = c->event();
        if (!cp) return fail(QH_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(cp, c->cstream));
        copied[par] = cp;
        last_copy = cp;
        c->stats.rounds++;
      }
      c->arrivals.push_back(qh::Arrival{slab_mask, sv, last_copy});
    }
    HIP_TRY(hipEventRecord(c->t1, c->cstream));
  }
  rc = check_launch(h);
  if (rc) return rc;
  c->timing_open = true;
  c->stats.exchanges++;
  c->stats.slabs += K;
  c->stats.rounds_packed += packed ? (uint64_t)K * nchunks : 0;
  c->stats.bytes_sent += (uint64_t)np * (1ull << (nloc - gbits)) * ab;
  return QH_OK;
}

int log2_exact(int v) {
  int g = 0;
  while ((1 << g) < v) ++g;
  return g;
}

}  // namespace

extern "C" {

int qh_comm_unique_id(void *id) {
  if (!id) return fail(QH_ERR_ARG, "null");
  std::string err;
  if (!qh::rccl().load(&err)) return fail(QH_ERR_COMM, "%s", err.c_str());
  static_assert(sizeof(ncclUniqueId) == QH_COMM_ID_BYTES, "id size");
  NCCL_TRY(qh::rccl().GetUniqueId((ncclUniqueId *)id));
  return QH_OK;
}

int qh_comm_init(qh_handle h, int nranks, int rank, const void *id) {
  if (!id) return fail(QH_ERR_ARG, "null id");
  std::string err;
  if (!qh::rccl().load(&err)) return fail(QH_ERR_COMM, "%s", err.c_str());
  int rc = comm_common_init(h, nranks, rank);
  if (rc) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclResult_t r = qh::rccl().CommInitRank(&h->comm->nccl, nranks, uid, rank);
  if (r != ncclSuccess) {
    (void)qh_comm_destroy(h);
    return fail(QH_ERR_COMM, "ncclCommInitRank(%d of %d): %s", rank, nranks, qh::rccl().GetErrorString(r));
  }
  int cnt = 0, ur = -1;      // what the communicator itself says (reported in qh_xstats: bench.py prints it)
  if (qh::rccl().CommCount(h->comm->nccl, &cnt) == ncclSuccess) h->comm->stats.comm_ranks = (uint32_t)cnt;
  if (qh::rccl().CommUserRank(h->comm->nccl, &ur) == ncclSuccess) h->comm->stats.comm_rank = (uint32_t)ur;
  if (cnt != nranks || ur != rank) {
    (void)qh_comm_destroy(h);
    return fail(QH_ERR_COMM, "the RCCL communicator reports rank %d of %d, asked for %d of %d", ur, cnt, rank, nranks);
  }
  return QH_OK;
}

int qh_comm_init_custom(qh_handle h, int nranks, int rank, qh_round_fn fn, void *user) {
  if (!fn) return fail(QH_ERR_ARG, "null callback");
  int rc = comm_common_init(h, nranks, rank);
  if (rc) return rc;
  h->comm->custom = fn;
  h->comm->custom_user = user;
  h->comm->stats.comm_ranks = (uint32_t)nranks;
  h->comm->stats.comm_rank = (uint32_t)rank;
  return QH_OK;
}

int qh_comm_init_dry(qh_handle h, int nranks, int rank) {
  if (!h || !h->dry) return fail(QH_ERR_ARG, "qh_comm_init_dry is for planner-only handles (qh_create_dry)");
  if (h->comm) return fail(QH_ERR_ARG, "handle already has a communicator");
  if (nranks < 1 || (nranks & (nranks - 1)) || rank < 0 || rank >= nranks)
    return fail(QH_ERR_ARG, "nranks %d must be a power of two, rank %d inside it", nranks, rank);
  if (nranks > qh::kMaxXferMoves + 1) return fail(QH_ERR_ARG, "at most %d ranks", qh::kMaxXferMoves + 1);
  auto *c = new qh::Comm;
  c->nranks = nranks;
  c->rank = rank;
  c->dry = true;
  h->comm = c;
  return QH_OK;
}

int qh_exchange_geometry(qh_handle h, qh_xgeom *out) {
  if (!h || !out) return fail(QH_ERR_ARG, "null");
  if (!h->comm) return fail(QH_ERR_ARG, "no communicator on this handle");
  *out = h->comm->last_geom;
  return QH_OK;
}

int qh_comm_destroy(qh_handle h) {
  if (!h || !h->comm) return QH_OK;
  qh::Comm *c = h->comm;
  if (c->dry) {
    delete c;
    h->comm = nullptr;
    return QH_OK;
  }
  (void)hipSetDevice(h->device);
  if (c->pstream) (void)hipStreamSynchronize(c->pstream);
  if (c->xstream) (void)hipStreamSynchronize(c->xstream);
  if (c->cstream) (void)hipStreamSynchronize(c->cstream);
  if (c->nccl) (void)qh::rccl().CommDestroy(c->nccl);
  for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
  if (c->t0) (void)hipEventDestroy(c->t0);
  if (c->t1) (void)hipEventDestroy(c->t1);
  if (c->staging) (void)hipFree(c->staging);
  if (c->d_sig) (void)hipFree(c->d_sig);
  if (c->h_sig) (void)hipHostFree(c->h_sig);
  if (c->h_send) (void)hipHostFree(c->h_send);
  if (c->h_recv) (void)hipHostFree(c->h_recv);
  if (c->xstream) (void)hipStreamDestroy(c->xstream);
  if (c->cstream) (void)hipStreamDestroy(c->cstream);
  if (c->pstream) (void)hipStreamDestroy(c->pstream);
  delete c;
  h->comm = nullptr;
  return QH_OK;
}

int qh_exchange_alltoall(qh_handle h, int base_bit, uint64_t chunk_amps) {
  if (!h || !h->comm) return fail(QH_ERR_ARG, "no communicator on this handle (qh_comm_init)");
  const int g = log2_exact(h->comm->nranks);
  std::vector<qh::BlockMove> mv;
  for (int j = 0; j < h->comm->nranks; ++j)
    if (j != h->comm->rank) mv.push_back(qh::BlockMove{j, j, j});
  return do_exchange(h, mv, base_bit, g, chunk_amps);
}

int qh_exchange_pair(qh_handle h, int shard_bit, int local_bit, uint64_t chunk_amps) {
  if (!h || !h->comm) return fail(QH_ERR_ARG, "no communicator on this handle (qh_comm_init)");
  const int g = log2_exact(h->comm->nranks);
  if (shard_bit < 0 || shard_bit >= g) return fail(QH_ERR_BAD_QUBIT, "shard bit %d of %d", shard_bit, g);
  const int mybit = (h->comm->rank >> shard_bit) & 1;
  std::vector<qh::BlockMove> mv{qh::BlockMove{h->comm->rank ^ (1 << shard_bit), 1 - mybit, 1 - mybit}};
  return do_exchange(h, mv, local_bit, 1, chunk_amps);
}

int qh_exchange_loopback(qh_handle h, int local_bit, uint64_t chunk_amps) {
  if (!h || !h->comm) return fail(QH_ERR_ARG, "no communicator on this handle (qh_comm_init)");
  std::vector<qh::BlockMove> mv{qh::BlockMove{h->comm->rank, 0, 1}, qh::BlockMove{h->comm->rank, 1, 0}};
  return do_exchange(h, mv, local_bit, 1, chunk_amps);
}

int qh_exchange_wait(qh_handle h) {
  if (!h || !h->comm || h->comm->dry) return QH_OK;
  HIP_TRY(hipSetDevice(h->device));
  for (const qh::Arrival &a : h->comm->arrivals) {
    if (!a.ev) continue;
    const int rc = wait_event(h, a.ev, "qh_exchange_wait");
    if (rc) return rc;
  }
  close_timing(h->comm);
  return QH_OK;
}

int qh_exchange_stats(qh_handle h, qh_xstats *out) {
  if (!h || !out) return fail(QH_ERR_ARG, "null");
  if (!h->comm) {
    memset(out, 0, sizeof *out);
    return QH_OK;
  }
  if (!h->comm->dry) {
    HIP_TRY(hipSetDevice(h->device));
    close_timing(h->comm);
  }
  *out = h->comm->stats;
  return QH_OK;
}

int qh_comm_allreduce_sum(qh_handle h, double *inout, int count) {
  if (!h || !h->comm || !inout) return fail(QH_ERR_ARG, "null / no communicator");
  if (!h->comm->nccl) return fail(QH_ERR_COMM, "qh_comm_allreduce_sum needs the RCCL transport");
  if (count < 1 || count > kRedBlocks) return fail(QH_ERR_ARG, "count %d", count);
  HIP_TRY(hipSetDevice(h->device));
  int rc = flush_impl(h);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_red, inout, count * sizeof(double), hipMemcpyHostToDevice, h->stream));
  NCCL_TRY(qh::rccl().AllReduce(h->d_red, h->d_red, (size_t)count, ncclDouble, ncclSum, h->comm->nccl, h->stream));
  HIP_TRY(hipMemcpyAsync(inout, h->d_red, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if ((rc = wait_stream(h, h->stream, "reader"))) return rc;
  return QH_OK;
}

}  // extern "C"

extern "C" {

// ---- literal drop-in on host buffers ------------------------------------------
// The device-side scratch of qh_host_apply1/applyc: per calling THREAD (the boundary is "one host
// thread per handle"), the two most recently used (nbits, width) shapes are kept so that an
// algorithm alternating between two register sizes does not pay a hipMalloc/hipFree pair per gate.
// qh_host_release() frees the calling thread's scratch; threads that exit without calling it leak
// nothing but address space until process exit (the handles are owned by the thread_local object).
namespace {
struct HostScratch {
  qh_handle slot[2] = {nullptr, nullptr};
  ~HostScratch() {
    for (qh_handle &h : slot) {
      if (h) qh_destroy(h);
      h = nullptr;
    }
  }
};
thread_local HostScratch g_host;
}  // namespace

static int host_handle(int nbits, int bw, qh_handle *out) {
  for (int k = 0; k < 2; ++k)
    if (g_host.slot[k] && g_host.slot[k]->nloc == nbits && g_host.slot[k]->bw == bw) {
      if (k == 1) std::swap(g_host.slot[0], g_host.slot[1]);   // most recent first
      *out = g_host.slot[0];
      return QH_OK;
    }
  if (g_host.slot[1]) qh_destroy(g_host.slot[1]);
  g_host.slot[1] = g_host.slot[0];
  g_host.slot[0] = nullptr;
  int rc = qh_create(nbits, bw, 0, &g_host.slot[0]);
  if (rc) return rc;
  g_host.slot[0]->relayout = 0;   // one gate per call: nothing to gain from a second buffer
  *out = g_host.slot[0];
  return QH_OK;
}

int qh_host_release(void) {
  for (qh_handle &h : g_host.slot) {
    if (h) qh_destroy(h);
    h = nullptr;
  }
  return QH_OK;
}

int qh_host_apply1(void *psi, const double gate[8], int nbits, int tgt, int bit_width) {
  if (!psi || !gate) return fail(QH_ERR_ARG, "null pointer");
  int rc = check_args(nbits, bit_width);
  if (rc) return rc;
  if (tgt < 0 || tgt >= nbits) return fail(QH_ERR_BAD_QUBIT, "apply1: qubit %d out of range for %d qubits", tgt, nbits);
  qh_handle h;
  rc = host_handle(nbits, bit_width, &h);
  if (rc) return rc;
  if ((rc = qh_upload(h, psi, 0, 1ull << nbits))) return rc;
  if ((rc = qh_apply1(h, tgt, gate))) return rc;
  return qh_download(h, psi, 0, 1ull << nbits);
}

int qh_host_applyc(void *psi, const double gate[8], int nbits, int ctl, int tgt, int bit_width) {
  if (!psi || !gate) return fail(QH_ERR_ARG, "null pointer");
  int rc = check_args(nbits, bit_width);
  if (rc) return rc;
  if (tgt < 0 || tgt >= nbits) return fail(QH_ERR_BAD_QUBIT, "applyc: qubit %d out of range for %d qubits", tgt, nbits);
  const int c = ref_ctl_bit(nbits, ctl, nbits - tgt - 1);
  if (c == -2) return fail(QH_ERR_BAD_QUBIT, "applyc: control qubit %d out of range for %d qubits", ctl, nbits);
  if (c == -1) return QH_OK;  // predicate never true: the reference leaves psi untouched
  qh_handle h;
  rc = host_handle(nbits, bit_width, &h);
  if (rc) return rc;
  if ((rc = qh_upload(h, psi, 0, 1ull << nbits))) return rc;
  if ((rc = qh_applyc(h, ctl, tgt, gate))) return rc;
  return qh_download(h, psi, 0, 1ull << nbits);
}

}  // extern "C"
