// exchange.hip.h -- the multi-GPU exchange step behind the C-ABI (include/qcc_hip.h, qh_comm_* /
// qh_exchange_*).
//
// The reference has no distributed path ("Distributed communication backend: none");
// this is the piece north_star adds: the state shards by its top log2(P) index bits, and a dense
// gate whose target is a shard bit is preceded by an exchange that makes that qubit local
// (SURVEY 8e).  One process per GPU; this file is what one rank runs.
//
//   transport   RCCL (ncclSend/ncclRecv grouped per round, bound with dlopen so that the library
//               loads on boxes without RCCL and shares the copy torch already mapped), on its own
//               HIP stream; or a host-staged callback (tests: several ranks sharing one GPU, any
//               fabric without peer access).
//   shape       rank r sends BLOCK j of its shard (the amplitudes whose g local bits
//               [base, base+g) equal j) to rank j and receives rank j's block r into the same
//               place (all P-1 peers in every round: all xGMI links busy), or -- pairwise mode --
//               one block to one peer.  In place: a round lands in one of two staging halves and
//               is copied home by a third stream while the next round is on the links.
//   overlap     the exchange is cut into SLABS (fixed values of up to three high local bits that
//               are tile bits of neither neighbouring sweep).  The last sweep before the exchange
//               is launched slab by slab and slab k's rounds start when ITS sub-launch has
//               finished (HIP events, no host wait); the first sweep after the exchange is
//               launched slab by slab too, slab k waiting only for slab k's arrival.  So the
//               links run while (K-1)/K of both neighbouring sweeps compute on the handle's stream.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <string>
#include <vector>

#include "kernels_gate.hip.h"

namespace qh {

// PACKED rounds (engine.hip, do_exchange): when the blocks' index bits do not leave long contiguous runs --
// relayout sweeps move index bits around -- a gather kernel packs the amplitudes of one round into the staging
// area, peer after peer, and a scatter kernel puts the received ones in place.  Work item i = (peer slot m,
// amplitude j of the round): index = the round's counter with zeros inserted at the block and slab bits
// (`ins`), plus the peer's block value and the slab value (`off[m]`).  HBM-bound: n x 16 B read + written.
constexpr int kMaxXferMoves = 63;
struct XferGeom {
  BitIns ins;
  int np;
  uint64_t off[kMaxXferMoves];
};
template <typename A>
__global__ __launch_bounds__(256) void k_xpack(const A *__restrict__ psi, A *__restrict__ stage, uint64_t n, uint64_t start, XferGeom g) {
  const uint64_t total = n * (uint64_t)g.np;
  const int nb = 63 - __builtin_clzll(n);
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
    const uint64_t idx = expand_index(start + (i & (n - 1)), g.ins) | g.off[i >> nb];
    st_amp<true>(stage + i, ld_amp<true>(psi + idx));
  }
}
template <typename A>
__global__ __launch_bounds__(256) void k_xunpack(A *__restrict__ psi, const A *__restrict__ stage, uint64_t n, uint64_t start, XferGeom g) {
  const uint64_t total = n * (uint64_t)g.np;
  const int nb = 63 - __builtin_clzll(n);
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
    const uint64_t idx = expand_index(start + (i & (n - 1)), g.ins) | g.off[i >> nb];
    st_amp<true>(psi + idx, ld_amp<true>(stage + i));
  }
}

struct RcclApi {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  std::string where;

  bool load(std::string *err) {
    if (lib) return true;
    // the copy already mapped in this process first (torch bundles its own librccl.so.1 next to
    // its own HIP runtime: two RCCLs over two HIP runtimes in one process do not work)
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (lib) { where = std::string(n) + " (already mapped)"; break; }
    }
    for (int k = 0; !lib && k < 3; ++k) {
      lib = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
      if (lib) where = names[k];
    }
    if (!lib) {
      *err = std::string("RCCL not found (librccl.so.1): ") + (dlerror() ? dlerror() : "?");
      return false;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
      void *p = dlsym(lib, name);
      if (!p) { ok = false; *err = std::string("RCCL symbol missing: ") + name; }
      return p;
    };
    GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
    CommCount = (decltype(CommCount))sym("ncclCommCount");
    CommUserRank = (decltype(CommUserRank))sym("ncclCommUserRank");
    if (!ok) { lib = nullptr; return false; }
    return true;
  }
};

inline RcclApi &rccl() {
  static RcclApi api;
  return api;
}

// One pending arrival: the amplitudes of the shard with (index & slab_mask) == slab_val are
// complete once `ev` has fired.
struct Arrival {
  uint64_t slab_mask, slab_val;
  hipEvent_t ev;
};

struct Comm {
  int nranks = 0, rank = 0;
  bool dry = false;                  // planner-only handle (qh_comm_init_dry): geometry is decided and recorded, nothing moves
  ncclComm_t nccl = nullptr;
  qh_round_fn custom = nullptr;      // host-staged transport (tests / no peer access)
  void *custom_user = nullptr;
  hipStream_t xstream = nullptr;     // the links
  hipStream_t cstream = nullptr;     // staging -> home copies / scatter kernels
  hipStream_t pstream = nullptr;     // gather kernels of packed rounds
  void *staging = nullptr;           // two receive halves (+ two send halves, packed rounds) of (peers x chunk) amplitudes
  size_t staging_bytes = 0;
  void *h_send = nullptr, *h_recv = nullptr;   // pinned, custom transport only
  size_t h_bytes = 0;
  std::vector<hipEvent_t> pool;      // events of the current exchange (reused by the next)
  size_t pool_used = 0;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool timing_open = false;
  std::vector<Arrival> arrivals;     // consumed by the next flush
  qh_xstats stats{};
  qh_xgeom last_geom{};              // how the last exchange was cut (qh_exchange_geometry)
  std::vector<uint64_t> verified;    // geometry signatures the ranks have already compared (RCCL transport)
  double *d_sig = nullptr, *h_sig = nullptr;   // 4 doubles in HBM / 8 pinned: the signature all-reduce of verify_geometry

  hipEvent_t event() {
    if (pool_used == pool.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[pool_used++];
  }
};

// The blocks one exchange moves: block value `blk` of the g bits at `base` goes to `peer`, and
// that peer's data lands in block value `land` (== blk except in the loop-back self test).
struct BlockMove { int peer, blk, land; };

// Picks up to `want` slab bits: the highest local bits outside `avoid`.
inline uint64_t pick_slab_bits(int nloc, uint64_t avoid, int want, int min_bit) {
  uint64_t m = 0;
  for (int b = nloc - 1; b >= min_bit && want > 0; --b)
    if (!((avoid >> b) & 1ull)) { m |= 1ull << b; --want; }
  return m;
}

// deposit the low bits of v into the set bits of mask (ascending)
inline uint64_t deposit_bits(uint64_t v, uint64_t mask) {
  uint64_t out = 0;
  for (uint64_t m = mask; m; m &= m - 1) {
    const int b = __builtin_ctzll(m);
    out |= (v & 1ull) << b;
    v >>= 1;
  }
  return out;
}

}  // namespace qh
