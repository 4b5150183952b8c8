// planner.h -- host-side gate records, classification and the sweep planner.
//
// The engine queues the reference's native calls (apply1/applyc,
// /root/reference/src/lib/xgates.cc:23-67) and this planner groups them into
// SWEEPS: one read + one write of the state per sweep instead of per gate
// (the idea the reference sketched and abandoned in
// src/libq/gates_jit.cc:53-136).  A sweep fixes a REGISTER TILE: the 6 low index
// bits (one amplitude per lane of a 64-wide wavefront) plus up to 5 "register"
// bits anywhere above (2^5 amplitudes per lane).  Every queued gate whose target
// is a tile bit can run inside the sweep; diagonal gates can always run (any bits).
//
// Pure host C++ (no HIP types): unit-tested on CPU through qh_plan_json.
#pragma once
#include <sched.h>
#include <stdint.h>

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/qcc_hip.h"

namespace qh {

// One submitted gate in PHYSICAL bit positions.
struct GateRec {
  uint64_t ctl_mask;  // all these index bits must be 1
  int tgt;            // target bit
  double g[8];        // row-major 2x2, (re,im) pairs
  uint64_t neg_mask = 0;  // all these index bits must be 0 (planner-internal: see propagate_x)
  // planner-internal, sharded handles only (see Planner::plan): `variant` = the gate touches a shard bit, so what
  // it does differs from rank to rank; `ghost` = on THIS rank it does nothing (its shard-bit control is 0, or its
  // rank-dependent phase is 1).  Ghosts stay in the gate list so that every rank plans the same sweeps, tiles and
  // layouts (an exchange needs the ranks to agree on where every index bit lives); they emit no arithmetic.
  uint8_t ghost = 0, variant = 0;
  // planner-internal: a diagonal gate on a shard bit, re-expressed on one of its local control bits, whose factor is
  // zero on SOME rank (a projector fed through apply1): "phase factor or dense 2x2" would then be decided differently
  // from rank to rank, so every rank takes the dense path
  uint8_t force_dense = 0;
};

// The boundary only ever sees the four matrix entries: classify by
// exact zeros, which the reference's constructors produce (ops.py:110-207).
inline bool is_diag(const double g[8]) {
  return g[2] == 0.0 && g[3] == 0.0 && g[4] == 0.0 && g[5] == 0.0;
}
inline bool is_one(double re, double im) { return re == 1.0 && im == 0.0; }
// Diagonal gates the sweep kernel runs as phase factors.  A diagonal gate with
// d0 == 0 but a local target cannot be written as "d0 everywhere, d1/d0 where
// the bit is set", so the planner treats it as a dense gate.
inline bool plan_diag(const double g[8], int tgt) {
  return is_diag(g) && (tgt < 0 || !(g[0] == 0.0 && g[1] == 0.0));
}
inline bool plan_diag(const GateRec &r) { return !r.force_dense && plan_diag(r.g, r.tgt); }

constexpr int kLaneBits = 6;   // wavefront = 64 lanes
constexpr int kLaneLow = 3;    // complex128: lane bits 0..2 are ALWAYS index bits 0..2 (8 x 16 B = one 128-byte line);
                               // complex64 uses 4 (16 x 8 B): SweepPlan::lane_low
constexpr int kLaneHi = 3;     // lane bits 3..5 sit on index bits 3,4,5 -- or on any three bits <= kMaxLaneHiBit
constexpr int kMaxLaneHiBit = 27;  // per-lane byte offset must fit the 32-bit voffset of global_load
constexpr int kMaxRegBits = 6; // complex128: 5 (32 amplitudes = 128 VGPRs of data per lane); complex64: 6 (64 amplitudes, the same 128 VGPRs)
inline int max_reg_bits(int bw) { return bw == 128 ? 5 : 6; }
constexpr int kMaxWaveBits = 2; // index bits selected by the wave id inside a workgroup ("super-tile", see OP_WSWAP)
constexpr int kMaxSweepOps = 1024;
constexpr int kMaxInsertBits = 12;  // bits a PLAN folds into the tile enumeration (fixed ones + register + movable lane + wave bits)
constexpr int kMaxSlabBits = 3;     // a slab launch around an exchange fixes up to this many more: kMaxIns (kernels_gate.hip.h) = the sum

enum : uint32_t { OP_DENSE_REG = 0, OP_DENSE_LANE = 1, OP_DIAG = 2,
                  OP_LSWAP = 3,    // lane bit tb (4|5) <-> register bit cm_reg (v_permlane swaps)
                  OP_WSWAP = 4 };  // wave bit tb <-> register bit cm_reg (through LDS); cm_thread = index bits that flip
// A DIAG op directly followed by an uncontrolled dense op on a LANE bit does not
// apply its per-lane factor c to the 2^RB slots: the dense op folds it into its
// per-lane matrix coefficients (H.diag(c)), two complex products per lane.
enum : uint32_t { OPF_DEFER_C = 1, OPF_USE_C = 2, OPF_REAL = 4,  // REAL: all four entries real
                  OPF_BFLY = 8, OPF_BFLY_SHIFT = 4,                // unit-entry butterfly, variant in bits 4..6
                  OPF_LANE_DPP = 128, OPF_SWAP_RI = 256,           // lane butterfly by DPP moves; partner re/im exchanged
                  OPF_GHOST = 1024,     // planner-internal: the op of a ghost gate (counted by the cost model, removed before the plan leaves)
                  // register butterfly behind a phase c (1 + i) / c (1 - i) on the slots with its target bit set (a T or T^+
                  // waiting for it: fused, tools/gen_sweep_asm.py L_bfr*); the real scale c travels in g[0]
                  OPF_ROT_P = 2048, OPF_ROT_M = 4096 };

// Gates of the form c*M with every entry of M in {1,-1,i,-i} (h, yroot, v = sqrt-x and
// their adjoints: ops.py:130-132,152-162) cost additions only once the scalar c is moved
// elsewhere.  Returns the variant (see tools/gen_sweep_asm.py, L_bf) or -1; c = g[0..1].
inline bool env_flag(const char *name, bool dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) != 0 : dflt;
}

inline int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

inline int butterfly_variant(const double *g) {
  const double cr = g[0], ci = g[1];
  if (cr == 0.0 && ci == 0.0) return -1;
  auto eq = [&](int k, double r, double i) { return g[2 * k] == r && g[2 * k + 1] == i; };
  if (eq(1, cr, ci) && eq(2, cr, ci) && eq(3, -cr, -ci)) return 0;     // [[1,1],[1,-1]]
  if (eq(1, -cr, -ci) && eq(2, cr, ci) && eq(3, cr, ci)) return 1;     // [[1,-1],[1,1]]
  if (eq(1, cr, ci) && eq(2, -cr, -ci) && eq(3, cr, ci)) return 2;     // [[1,1],[-1,1]]
  if (eq(1, ci, -cr) && eq(2, ci, -cr) && eq(3, cr, ci)) return 3;     // [[1,-i],[-i,1]]
  if (eq(1, -ci, cr) && eq(2, -ci, cr) && eq(3, cr, ci)) return 4;     // [[1,i],[i,1]]
  return -1;
}

// ---- device-visible records (plain data, copied verbatim to HBM) -------------
// A factor that multiplies amplitudes whose index has all bits of `mask` set;
// mask only contains bits OUTSIDE the tile (uniform over a wavefront).
struct OTerm {
  uint64_t mask;
  double re, im;
};
// A diagonal group.  Wave-uniform part u = phi0 * prod(chunk-table lookups) *
// prod(oterms that apply).  Per-lane factor f = ltab[lane] * u when LTAB is set,
// else ((lane & lane_mask) == lane_mask ? u : 1).  f multiplies the register
// slots k with (k & reg_mask) == reg_mask.
//   chunk table t: 256 entries indexed by (tile_idx >> shift_t) & 255 -- the
//   product of the single-bit outside factors of 8 consecutive index bits;
//   lane table: 64 entries, product of all merged pure-lane factors.
constexpr uint32_t DG_LTAB = 1u;
constexpr uint32_t DG_LTAB_LDS = 2u;   // set at launch: this sweep's lane tables were copied into LDS
// (bit 2 and bits 8..15 are set at launch too: kernels_sweep.hip.h group_handler_bits)
// DG_BITFAC: reg_mask is ONE register bit j and tab_off[3] points at four more factors w_0..w_3, one per
// register bit of bitfac_others(rb, j): a slot with bit j set is multiplied by f times the w_k of its set bits.
// The kernel walks the 2^4 subsets as a tree (one complex product per slot to derive the factor, one to
// apply it), so the controlled phases between one target and the other register bits -- a QFT's ladder --
// cost ONE group instead of one per partner bit.
constexpr uint32_t DG_BITFAC = 8u;
constexpr int kBitFacs = 4;
inline int bitfac_others(int rb, int j, int out[kBitFacs]) {   // the first four register bits other than j
  int n = 0;
  for (int b = 0; b < rb && n < kBitFacs; ++b) if (b != j) out[n++] = b;
  return n;
}
struct DGroup {
  uint32_t lane_mask, reg_mask;
  uint32_t oterm_off, n_oterms;
  double re, im;
  uint32_t flags, ltab_off;   // ltab_off/tab_off: in 16-byte entries of SweepPlan::tables
  uint32_t ntab, tab_shift;   // tab_shift: 4 x u8
  uint32_t tab_off[4];
};
struct SweepOp {
  uint32_t kind;       // OP_*
  uint32_t tb;         // dense: register-bit index or lane-bit index
  uint32_t cm_reg;     // dense: control mask over register-bit indices
  uint32_t n_groups;   // diag
  uint64_t cm_thread;  // dense: control mask over (shard|outside|lane) index bits
  uint32_t group_off;  // diag
  uint32_t flags;      // OPF_*
  double g[8];         // dense: the 2x2
};

struct SweepPlan {
  int rb = 0;                      // register bits used by the kernel instance
  int regpos[kMaxRegBits] = {0};   // ascending physical positions
  int regpos_store[kMaxRegBits] = {0};  // register bits when the tile is stored (OP_WSWAPs are not undone)
  int lane_low = kLaneLow;         // lane bits below this are the index bits of the same number
  int lanehi[kLaneHi] = {3, 4, 5}; // ascending positions of the 6 - lane_low movable lane bits ({lane_low..5} = contiguous runs)
  int nlanehi() const { return kLaneBits - lane_low; }
  // SEATS: which index bit each of the six lane bits holds.  The lane-resident bits of a tile are the line bits
  // 0..lane_low-1 plus lanehi[]; by default lane bit i holds the i-th of them.  Which LANE holds which amplitude of a
  // 128-byte line is free for the memory system (profiles/r04/lanemap_membench.txt: < 1-2.5 %), but not for the op
  // stream: a butterfly on lane bit 4 / 5 runs as a v_permlane swap + register butterfly (72 + 64 VALU instructions, and
  // the bit then LIVES in a register), on lane bits 0, 1, 3 as 206 instructions of DPP moves, on lane bit 2 as 334.  So
  // op-heavy sweeps seat their busiest lane-resident bits on lane bits 5 and 4 and the idlest on lane bit 2 (build_sweep).
  int seat[kLaneBits] = {0, 1, 2, 3, 4, 5};         // index bit held by lane bit i when the tile is loaded
  int seat_store[kLaneBits] = {0, 1, 2, 3, 4, 5};   // ... when it is stored (lane <-> register exchanges may stay: relayout)
  int seat_dest[kLaneBits] = {0, 1, 2, 3, 4, 5};    // relayout store: position 0..5 the bit on lane bit i goes to
  bool contiguous() const {
    for (int k = 0; k < nlanehi(); ++k) if (lanehi[k] != lane_low + k) return false;
    return true;
  }
  int nwave = 0;                   // wave bits: the 2^nwave waves of a workgroup hold the tiles differing in
  int wavepos[kMaxWaveBits] = {0}; // these index bits; a gate on one runs after OP_WSWAP moved it into registers
  uint64_t fixed_ones = 0;         // local bits fixed to 1 in the tile enumeration
  uint64_t ntiles = 0;             // wavefront tiles to process
  // RELAYOUT (out-of-place "gather" sweep): the tile is loaded from wherever its bits are and
  // stored CONTIGUOUSLY into the handle's second buffer -- lane bits on positions 0..5, register
  // bit k on 6+k, wave bit j on 6+rb+j, every other index bit above, order kept.  dest_pos[p] =
  // new position of the index bit at position p (local bits only).  See Planner::relayout().
  bool relayout = false;
  uint8_t dest_pos[64] = {0};
  int wavepos_store[kMaxWaveBits] = {0};   // index bits the wave bits hold when the tile is stored (lanes: seat_store)
  int reg_dest[kMaxRegBits] = {6, 7, 8, 9, 10, 11};  // relayout: where register bit k / wave bit j go (positions 6.. of the
  int wave_dest[kMaxWaveBits] = {11, 12};        // contiguous block, in ascending order of the index bits they hold)
  std::vector<SweepOp> ops;
  std::vector<DGroup> groups;
  std::vector<OTerm> oterms;
  std::vector<double> tables;      // (re,im) pairs: lane tables FIRST (n_ltab x 64 entries), then chunk tables
  std::vector<double> ltabs;       // lane tables while the sweep is being emitted (moved to the front of `tables`)
  int n_ltab = 0;
  // accounting
  uint64_t gates = 0;              // reference gate applications executed by this sweep
  uint64_t alg_bytes = 0;          // their minimal-touch bytes
  uint64_t swept_bytes = 0;        // bytes this launch reads + writes
};

struct PlanResult {
  std::vector<SweepPlan> sweeps;
  uint64_t noop_gates = 0;  // dropped: shard-bit control unsatisfied / identity
  bool moved = false;       // some sweep re-laid the state out:
  uint8_t final_pos[64];    // position, after the flush, of the index bit that was at position p before it
  PlanResult() { for (int p = 0; p < 64; ++p) final_pos[p] = (uint8_t)p; }
};

inline int popc(uint64_t x) { return __builtin_popcountll(x); }

// minimal-touch bytes of one gate on a local shard of 2^nloc amplitudes
inline uint64_t gate_alg_bytes(const GateRec &r, int nloc, uint64_t amp_bytes) {
  const uint64_t lm = (nloc >= 64) ? ~0ull : ((1ull << nloc) - 1);
  const int nc = popc(r.ctl_mask & lm);
  const bool diag = is_diag(r.g);
  uint64_t touched = 1ull << (nloc - nc);
  if (diag && r.tgt < nloc && is_one(r.g[0], r.g[1])) touched >>= 1;
  return touched * amp_bytes * 2;
}

// A relayout sweep writes unit w (tile / super-tile number: the index bits outside the tile, in
// ascending order) to block P(w) of the second buffer, P = the bit permutation dest_pos induces on
// those bits.  P is handed to the kernel as runs of bits that move together: dst |= shift(w & mask).
constexpr int kMaxUnitSegs = 8;
inline int unit_segments(const SweepPlan &sp, int nloc, uint64_t *masks, int *shifts) {
  uint64_t tile = 0;
  for (int i = 0; i < kLaneBits; ++i) tile |= 1ull << sp.seat_store[i];
  for (int k = 0; k < sp.rb; ++k) tile |= 1ull << sp.regpos_store[k];
  for (int k = 0; k < sp.nwave; ++k) tile |= 1ull << sp.wavepos_store[k];
  const int tilebits = popc(tile);
  int nseg = 0, i = 0, last_shift = 1 << 20;
  for (int p = 0; p < nloc; ++p) {
    if ((tile >> p) & 1ull) continue;
    const int shift = ((int)sp.dest_pos[p] - tilebits) - i;   // unit bit i moves to unit bit i + shift
    if (shift != last_shift) {
      ++nseg;
      last_shift = shift;
      if (masks && nseg <= kMaxUnitSegs) { masks[nseg - 1] = 0; shifts[nseg - 1] = shift; }
    }
    if (masks && nseg <= kMaxUnitSegs) masks[nseg - 1] |= 1ull << i;
    ++i;
  }
  return nseg;
}

class Planner {
 public:
  // one gate as the tile selection sees it (pass_fast / search_run), and the state of a pass in front of a gate
  struct PassRec { uint64_t dense_bits, diag_bits; uint32_t score; int tgt; };
  struct PassState { size_t idx = 0; uint64_t blocked_all = 0, blocked_diag = 0; size_t count = 0, score = 0; bool valid = false; };
  Planner(int nloc, uint64_t shard, int bw, int max_rb, bool split_lanes = true, int wave_bits = -1,
          bool allow_relayout = false, bool keep_ghosts = false)
      : nloc_(nloc), shard_(shard), amp_bytes_(bw == 128 ? 16 : 8), split_lanes_(split_lanes),
        relayout_(allow_relayout), keep_ghosts_(keep_ghosts) {
    rb_cap_ = std::min({max_rb, max_reg_bits(bw), nloc - kLaneBits});
    lane_low_ = bw == 128 ? 3 : 4;   // 128-byte lines: 8 complex128 or 16 complex64 amplitudes
    lane_hi_ = kLaneBits - lane_low_;
    if (wave_bits >= 0) max_wave_ = std::min(wave_bits, kMaxWaveBits);
  }

  // The gates of the queue as the sweeps will see them: shard bits resolved, the reference's 5-gate Toffolis
  // fused, X gates turned into pending flips (fills weight_; *noops counts the gates that vanished).
  void prepare(const std::vector<GateRec> &queue, std::vector<GateRec> *pending_out, std::vector<uint64_t> *alg_out, uint64_t *noops) {
    struct { uint64_t noop_gates = 0; } out;
    std::vector<GateRec> pending;
    pending.reserve(queue.size());
    alg_override_.clear();
    const uint64_t lm = (1ull << nloc_) - 1;
    for (const auto &q : queue) {
      // resolve shard-index bits now: they are constants on this rank.  A gate that does nothing HERE (its
      // shard-bit control is 0, its rank-dependent phase is 1) is dropped on an unsharded handle and kept as
      // a ghost on a sharded one: the ranks of a sharded state must plan identical sweeps, tile bits and
      // relayouts whatever their rank index is (GateRec::ghost).
      const uint64_t hi = q.ctl_mask >> nloc_;
      GateRec r = q;
      r.ctl_mask &= lm;
      r.variant = (hi != 0 || q.tgt >= nloc_) ? 1 : 0;
      r.ghost = ((shard_ & hi) != hi) ? 1 : 0;
      if (r.ghost && !keep_ghosts_) { out.noop_gates++; continue; }
      if (r.tgt >= nloc_) {
        // diagonal gate on a shard bit: a scalar factor under the local controls
        const bool set = (shard_ >> (r.tgt - nloc_)) & 1ull;
        const double fr = set ? r.g[6] : r.g[0], fi = set ? r.g[7] : r.g[1];
        if (is_one(fr, fi)) r.ghost = 1;
        if (r.ghost && !keep_ghosts_) { out.noop_gates++; continue; }
        // re-express as a one-sided diagonal gate on one of its control bits, or
        // as a global factor (tgt = -1) when it has no local control
        if (r.ctl_mask) {
          const int c = __builtin_ctzll(r.ctl_mask);
          r.ctl_mask &= ~(1ull << c);
          r.tgt = c;
          r.force_dense = ((r.g[0] == 0.0 && r.g[1] == 0.0) || (r.g[6] == 0.0 && r.g[7] == 0.0)) ? 1 : 0;
          r.g[0] = 1; r.g[1] = 0; r.g[6] = fr; r.g[7] = fi;
        } else {
          r.tgt = -1;
          r.g[0] = fr; r.g[1] = fi; r.g[6] = fr; r.g[7] = fi;
        }
        if (r.ghost) { r.g[0] = r.g[6] = 1; r.g[1] = r.g[7] = 0; }   // (never read: a ghost emits nothing)
        r.g[2] = r.g[3] = r.g[4] = r.g[5] = 0;
        if (r.ghost) out.noop_gates++;
        pending.push_back(r);
        alg_override_.push_back(r.ghost ? 0 : gate_alg_bytes(q, nloc_, amp_bytes_));
        continue;
      }
      if (!r.variant && is_diag(r.g) && is_one(r.g[0], r.g[1]) && is_one(r.g[6], r.g[7])) { out.noop_gates++; continue; }
      if (r.ghost) out.noop_gates++;
      pending.push_back(r);
      alg_override_.push_back(r.ghost ? 0 : gate_alg_bytes(r, nloc_, amp_bytes_));
    }
    std::vector<uint64_t> alg = alg_override_;
    std::vector<uint32_t> weight(pending.size(), 1);
    for (size_t i = 0; i < pending.size(); ++i) if (pending[i].ghost) weight[i] = 0;
    fuse_sleator_weinfurter(&pending, &alg, &weight);
    if (propagate_x_) propagate_x(&pending, &alg, &weight, &out.noop_gates);
    weight_.swap(weight);
    pending_out->swap(pending);
    alg_out->swap(alg);
    *noops = out.noop_gates;
  }

  // How many sweeps plan() would launch, and whether one of them has a far tile (plan_has_far_tile), from the
  // tile selection alone -- no ops are emitted, no index bit is relabelled (with relayout only the first sweep
  // can then have a far tile: later ones gather from wherever and the check is skipped for them).  A tenth of a
  // full plan: what plan_best needs to pick the number of wave bits.
  void skeleton(const std::vector<GateRec> &queue, size_t *nsweeps, bool *far_tile, std::vector<uint64_t> *tiles_out = nullptr) {
    std::vector<GateRec> pending;
    std::vector<uint64_t> alg;
    uint64_t noops = 0;
    prepare(queue, &pending, &alg, &noops);
    *nsweeps = 0;
    *far_tile = false;
    const uint64_t always = (1ull << lane_low_) - 1;
    while (!pending.empty()) {
      std::vector<int> sel, lanehi, regs, waves;
      select_tile_bits(pending, &sel);
      assign_bits(sel, &lanehi, &regs, &waves);
      if (!relayout_ || *nsweeps == 0) {
        int far = 0;
        for (int b : lanehi) far += b >= 25;
        for (int b : regs) far += b >= 25;
        if (far >= 8) *far_tile = true;
      }
      std::vector<uint8_t> flags(pending.size(), 0);
      pass(pending, always | mask_of(lanehi) | mask_of(regs) | mask_of(waves), pending.size(), &flags);
      if (tiles_out) tiles_out->push_back(mask_of(sel));
      std::vector<GateRec> rest;
      for (size_t i = 0; i < pending.size(); ++i) if (!flags[i]) rest.push_back(pending[i]);
      if (rest.size() == pending.size()) rest.erase(rest.begin());     // (as build_sweep: never loop forever)
      pending.swap(rest);
      ++*nsweeps;
    }
  }

  // ---- level search ---------------------------------------------------------------------------------------
  // The greedy selection maximises the gates of ONE sweep; layered circuits (supremacy: every qubit couples to its
  // grid neighbours every few layers) can need a sweep less when an early sweep takes FEWER gates but leaves the
  // frontier where the next one runs long (BASELINE config 3, seed 0: greedy 85+88+79+84+6 gates in 5 sweeps;
  // 66+128+80+68 in 4 exists).  Rounds 3-6 looked for such tiles by local search over the TILES (the cap of a tile
  // hard, "every gate runs" soft); round 6 replaced it by the search over the dual object.  A plan of K sweeps is a
  // labelling L of the dense gates with levels 0..K-1 that never decreases along a dependency (K-1 nested cuts through
  // the circuit; diagonal gates ride along: they need no tile bit and go to any sweep between their neighbours), and a
  // sweep's tile is the set of qubits with a gate on its level: at most `cap` of them besides the line bits.  The
  // search keeps the labelling VALID and the caps soft: a move lifts a gate and everything that must follow it one
  // level up (or lowers it and what must precede it), the cost is the overflow sum(max(0, |tile_s| - cap)), ties by
  // sum |tile_s|^2; tabu search over the evictions from overfull tiles (first gate of a qubit on the level up, last one
  // down), short tenure, restarts from the depth-proportional labelling.  It finds 4-sweep tilings of supremacy-30
  // seeds 2 3 4 6 7 in 2-20 ms where the tile search had found none in 2 x 10^9 gate visits (an integer program says
  // they exist, and that seed 5 needs five: profiles/r06/level_search.txt).  Deterministic (generator seeded by
  // `stream`, budget counted in label changes, canonical bit numbers): the ranks of a sharded state find the same tiles.
  // Returns true and the tiles (bit numbering of the start of the flush).
  void canonical_records(const std::vector<GateRec> &pending, std::vector<PassRec> *rec0, int back[64], uint64_t *movable) {
    // CANONICAL bit numbers -- index bits above the 128-byte line renumbered in the order the queue first uses them -- so
    // that the same circuit gets the same walk, and the same answer, whatever layout earlier relayout sweeps have left
    // the state in (a loop over one circuit sees a different layout every step).
    const uint64_t always = (1ull << lane_low_) - 1;
    int canon[64], ncanon = lane_low_;
    for (int b = 0; b < 64; ++b) canon[b] = b < lane_low_ ? b : -1;
    auto canon_mask = [&](uint64_t m) {
      uint64_t o = 0;
      for (uint64_t t = m; t; t &= t - 1) {
        const int b = __builtin_ctzll(t);
        if (canon[b] < 0) canon[b] = ncanon++;
        o |= 1ull << canon[b];
      }
      return o;
    };
    rec0->resize(pending.size());
    uint64_t dense_used = 0;
    for (size_t i = 0; i < pending.size(); ++i) {
      const GateRec &r = pending[i];
      const bool diag = plan_diag(r);
      const uint64_t tb = canon_mask((r.tgt >= 0) ? (1ull << r.tgt) : 0);
      const uint64_t cb = canon_mask((r.ctl_mask | r.neg_mask) & ((1ull << nloc_) - 1));
      (*rec0)[i] = PassRec{diag ? 0 : tb, cb | (diag ? tb : 0), 1, tb ? __builtin_ctzll(tb) : -1};
      if (!diag) dense_used |= tb;
    }
    for (int b = 0; b < 64; ++b) back[b] = -1;
    for (int b = 0; b < 64; ++b) if (canon[b] >= 0) back[canon[b]] = b;
    *movable = dense_used & ~always;
  }
  // The dependency graph of the DENSE gates, as pass_fast / build_sweep order them: a dense gate follows the last dense
  // gate on its target, the last dense gate on each of its control bits, and -- through every diagonal gate (or control)
  // that touched its target since -- the last dense gates on the other bits of that gate; diagonal uses commute.
  static void build_dag(const std::vector<PassRec> &rec0, std::vector<int> *node_bit, std::vector<std::vector<int>> *succ,
                        std::vector<std::vector<int>> *pred) {
    int last_dense[64];
    std::vector<int> need[64];
    for (int b = 0; b < 64; ++b) last_dense[b] = -1;
    for (const PassRec &r : rec0) {
      if (!r.dense_bits) {
        for (uint64_t t = r.diag_bits; t; t &= t - 1) {
          const int b = __builtin_ctzll(t);
          for (uint64_t u = r.diag_bits & ~(1ull << b); u; u &= u - 1)
            if (last_dense[__builtin_ctzll(u)] >= 0) need[b].push_back(last_dense[__builtin_ctzll(u)]);
        }
        continue;
      }
      const int v = (int)node_bit->size(), tb = r.tgt;
      node_bit->push_back(tb);
      succ->emplace_back();
      pred->emplace_back();
      std::vector<int> ps;
      if (last_dense[tb] >= 0) ps.push_back(last_dense[tb]);
      for (int u : need[tb]) ps.push_back(u);
      for (uint64_t t = r.diag_bits; t; t &= t - 1) if (last_dense[__builtin_ctzll(t)] >= 0) ps.push_back(last_dense[__builtin_ctzll(t)]);
      std::sort(ps.begin(), ps.end());
      ps.erase(std::unique(ps.begin(), ps.end()), ps.end());
      for (int u : ps) { (*succ)[u].push_back(v); (*pred)[v].push_back(u); }
      need[tb].clear();
      last_dense[tb] = v;
      for (uint64_t t = r.diag_bits; t; t &= t - 1) need[__builtin_ctzll(t)].push_back(v);
    }
  }
  // The object the level search works on, for tools/tiling_milp.py (QH_PLAN_DAG=1 adds it to qh_plan_json): the target bit of every
  // dense gate (numbering of the flush's start), the dependency edges, the line bits every tile holds, the tile caps.
  std::string dag_json(const std::vector<GateRec> &queue) {
    std::vector<GateRec> pending;
    std::vector<uint64_t> alg;
    uint64_t noops = 0;
    prepare(queue, &pending, &alg, &noops);
    std::vector<PassRec> rec0;
    int back[64];
    uint64_t movable = 0;
    canonical_records(pending, &rec0, back, &movable);
    std::vector<int> node_bit;
    std::vector<std::vector<int>> succ, pred;
    build_dag(rec0, &node_bit, &succ, &pred);
    std::string o = "{\"line_bits\":" + std::to_string(lane_low_) + ",\"cap_per_wave_bits\":[" + std::to_string(lane_hi_ + rb_cap_) + "," +
                    std::to_string(lane_hi_ + rb_cap_ + 1) + "," + std::to_string(lane_hi_ + rb_cap_ + 2) + "],\"target_bit\":[";
    for (size_t g = 0; g < node_bit.size(); ++g) o += (g ? "," : "") + std::to_string(back[node_bit[g]]);
    o += "],\"edges\":[";
    bool first = true;
    for (size_t u = 0; u < succ.size(); ++u)
      for (int v : succ[u]) { o += (first ? "[" : ",[") + std::to_string(u) + "," + std::to_string(v) + "]"; first = false; }
    return o + "]}";
  }
  bool search_levels(const std::vector<GateRec> &queue, size_t K, uint64_t change_budget, uint64_t stream,
                     std::vector<std::vector<int>> *tiles_out, uint64_t *changes_used = nullptr) {
    if (changes_used) *changes_used = 0;
    if (K < 2 || K > 8) return false;
    std::vector<GateRec> pending;
    std::vector<uint64_t> alg;
    uint64_t noops = 0;
    prepare(queue, &pending, &alg, &noops);
    std::vector<PassRec> rec0;
    int back[64];
    uint64_t movable = 0;
    canonical_records(pending, &rec0, back, &movable);
    const uint64_t always = (1ull << lane_low_) - 1;
    const int cap = lane_hi_ + rb_cap_ + max_wave_;
    if ((size_t)popc(movable) > K * (size_t)cap) return false;       // not even room to visit every qubit once
    std::vector<int> node_bit;
    std::vector<std::vector<int>> succ, pred;
    build_dag(rec0, &node_bit, &succ, &pred);
    const int ng = (int)node_bit.size();
    if (!ng) return false;
    std::vector<int> depth(ng, 0);
    int maxd = 0;
    for (int u = 0; u < ng; ++u) for (int v : succ[u]) { depth[v] = std::max(depth[v], depth[u] + 1); maxd = std::max(maxd, depth[v]); }   // (nodes are in queue order: topological)
    // per qubit: its nodes in order (the chain), for "first / last gate of q on level s"
    std::vector<int> chain[64];
    for (int g = 0; g < ng; ++g) chain[node_bit[g]].push_back(g);
    std::vector<int8_t> L(ng);
    int occ[64][8], ntile[8];       // gates of qubit q on level s; movable qubits on level s
    uint64_t changes = 0;
    auto set_level = [&](int g, int to) {
      const int q = node_bit[g], from = L[g];
      ++changes;
      if (!((always >> q) & 1)) {
        if (--occ[q][from] == 0) --ntile[from];
        if (occ[q][to]++ == 0) ++ntile[to];
      }
      L[g] = (int8_t)to;
    };
    auto init = [&]() {
      memset(occ, 0, sizeof occ);
      memset(ntile, 0, sizeof ntile);
      for (int g = 0; g < ng; ++g) {
        L[g] = (int8_t)std::min<int>((int)K - 1, depth[g] * (int)K / (maxd + 1));
        const int q = node_bit[g];
        if (!((always >> q) & 1) && occ[q][L[g]]++ == 0) ++ntile[L[g]];
      }
    };
    auto cost = [&](int *over) {
      long sq = 0;
      int o = 0;
      for (size_t s = 0; s < K; ++s) { sq += (long)ntile[s] * ntile[s]; if (ntile[s] > cap) o += ntile[s] - cap; }
      *over = o;
      return (long)o * 100000 + sq;
    };
    std::vector<int> stk;
    std::vector<std::pair<int, int8_t>> undo;
    auto move = [&](int g, bool down) {       // lift g and what must follow it (lower g and what must precede it); false: out of range
      const int to = L[g] + (down ? -1 : 1);
      if (to < 0 || to >= (int)K) return false;
      stk.assign(1, g);
      undo.emplace_back(g, L[g]);
      set_level(g, to);
      while (!stk.empty()) {
        const int u = stk.back();
        stk.pop_back();
        for (int v : down ? pred[u] : succ[u])
          if (down ? L[v] > to : L[v] < to) { undo.emplace_back(v, L[v]); set_level(v, to); stk.push_back(v); }
      }
      return true;
    };
    auto revert = [&]() { for (size_t i = undo.size(); i-- > 0;) set_level(undo[i].first, undo[i].second); undo.clear(); };
    uint64_t rng = 0x9e3779b97f4a7c15ull * (2 * stream + 1);
    auto rnd = [&](uint32_t n) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)((rng >> 11) % n); };
    constexpr long kRestart = 1500;       // iterations per walk (supremacy-30: a walk that succeeds mostly does within 1 000)
    constexpr int kTenure = 7;            // a gate that moved stays put for 7-13 iterations (4-8 best of 2..50; 30: 20x slower)
    std::vector<long> tabu((size_t)ng * 2);
    std::vector<int> cands;
    bool found = false;
    for (long walk = 0; !found; ++walk) {
      init();
      std::fill(tabu.begin(), tabu.end(), -1);
      int over = 0, best_over = 0;
      long best_key = cost(&best_over);
      over = best_over;
      for (long it = 0; it < kRestart && over > 0; ++it) {
        if (++changes > change_budget) goto out;       // (an iteration counts as a change: the budget runs out whatever the moves do)
        // a circuit without a K-sweep tiling shows it early (supremacy-30 with K = 3: the overflow never drops below 5; every
        // instance that has one is at 1-4 after 400 iterations): the first walk ends the search then
        if (walk == 0 && it == 400 && best_over > 4) goto out;
        cands.clear();
        for (size_t s = 0; s < K; ++s) {
          if (ntile[s] <= cap) continue;
          for (uint64_t t = movable; t; t &= t - 1) {
            const int q = __builtin_ctzll(t);
            if (!occ[q][s]) continue;
            int first = -1, last = -1;
            for (int g : chain[q]) if (L[g] == (int8_t)s) { if (first < 0) first = g; last = g; }
            cands.push_back(first * 2);
            cands.push_back(last * 2 + 1);
          }
        }
        long bk = 1L << 60;
        int bm = -1, ties = 0;
        for (int m : cands) {
          undo.clear();
          if (!move(m >> 1, m & 1)) continue;
          int o = 0;
          const long k = cost(&o);
          revert();
          if (tabu[m] > it && !(o < best_over)) continue;
          if (k < bk) { bk = k; bm = m; ties = 1; }
          else if (k == bk && rnd((uint32_t)++ties) == 0) bm = m;
        }
        if (bm < 0) continue;
        undo.clear();
        move(bm >> 1, bm & 1);
        const long k = cost(&over);
        for (const auto &u : undo) tabu[(size_t)u.first * 2 + ((bm & 1) ? 0 : 1)] = it + kTenure + (long)rnd(kTenure);
        if (k < best_key) { best_key = k; best_over = over; }
      }
      found = over == 0;
    }
  out:
    if (changes_used) *changes_used = changes;
    if (!found) return false;
    tiles_out->clear();
    for (size_t s = 0; s < K; ++s) {
      std::vector<int> b;
      for (uint64_t t = movable; t; t &= t - 1) if (occ[__builtin_ctzll(t)][s]) b.push_back(back[__builtin_ctzll(t)]);
      std::sort(b.begin(), b.end());
      tiles_out->push_back(b);
    }
    return true;
  }
  void set_tiles(const std::vector<std::vector<int>> &t) { tiles_ = t; }

  PlanResult plan(const std::vector<GateRec> &queue) {
    PlanResult out;
    std::vector<GateRec> pending;
    std::vector<uint64_t> alg;
    prepare(queue, &pending, &alg, &out.noop_gates);
    const uint64_t lm = (1ull << nloc_) - 1;
    std::vector<int> first_tile;
    keep_line_bits_ = !tiles_.empty();
    while (!pending.empty()) {
      std::vector<GateRec> rest;
      std::vector<uint64_t> rest_alg;
      std::vector<uint32_t> rest_w;
      std::vector<int> forced;
      const std::vector<int> *fs = nullptr;
      if (out.sweeps.size() < tiles_.size()) {          // tiles chosen for the whole flush (the tile search),
        for (int b : tiles_[out.sweeps.size()])         // in the bit numbering the flush started with
          if (b >= lane_low_ && b < nloc_) forced.push_back(out.final_pos[b]);
        fs = &forced;
      }
      out.sweeps.push_back(build_sweep(pending, alg, &rest, &rest_alg, &rest_w, fs));
      if (out.sweeps.size() == 1) {     // where this flush began: see the last sweep below
        const SweepPlan &f = out.sweeps[0];
        for (int k = 0; k < f.nlanehi(); ++k) first_tile.push_back(f.lanehi[k]);
        for (int k = 0; k < f.rb; ++k) first_tile.push_back(f.regpos[k]);
        for (int k = 0; k < f.nwave; ++k) first_tile.push_back(f.wavepos[k]);
      }
      if (out.sweeps.back().relayout) {
        std::vector<int> ahead;
        if (!rest.empty() && out.sweeps.size() < tiles_.size()) {
          // the next sweep's tile is already chosen (the tile search): those are the bits it will gather
          for (int b : tiles_[out.sweeps.size()])
            if (b >= lane_low_ && b < nloc_) ahead.push_back(out.final_pos[b]);
        } else if (!rest.empty()) {
          select_tile_bits(rest, &ahead);
        } else if (out.sweeps.size() > 1) {
          // last sweep of the flush: nothing is known about what comes next; circuits run in loops
          // (Grover iterations, the QFT of every bench step), so bet on the bits this flush started with
          for (int p : first_tile) ahead.push_back(out.final_pos[p]);
        }
        std::sort(ahead.begin(), ahead.end());
        finish_relayout(&out.sweeps.back(), ahead);
        if (unit_segments(out.sweeps.back(), nloc_, nullptr, nullptr) > kMaxUnitSegs)
          finish_relayout(&out.sweeps.back(), std::vector<int>());   // (the kernel moves at most 8 runs of unit-index bits)
      }
      const SweepPlan &done = out.sweeps.back();
      if (done.relayout) {      // the gates still to come live in the new layout
        auto map_mask = [&](uint64_t m) {
          uint64_t o = m >> nloc_ << nloc_;
          for (uint64_t t = m & lm; t; t &= t - 1) o |= 1ull << done.dest_pos[__builtin_ctzll(t)];
          return o;
        };
        for (GateRec &r : rest) {
          r.ctl_mask = map_mask(r.ctl_mask);
          r.neg_mask = map_mask(r.neg_mask);
          if (r.tgt >= 0 && r.tgt < nloc_) r.tgt = done.dest_pos[r.tgt];
        }
        for (int p = 0; p < nloc_; ++p) out.final_pos[p] = done.dest_pos[out.final_pos[p]];
        out.moved = true;
      }
      pending.swap(rest);
      alg.swap(rest_alg);
      weight_.swap(rest_w);
    }
    return out;
  }

 private:
  int nloc_;
  uint64_t shard_;
  uint64_t amp_bytes_;
  int rb_cap_;
  int lane_low_ = kLaneLow, lane_hi_ = kLaneHi;
  bool split_lanes_;   // allow lane bits 3..5 to sit on arbitrary index bits (8 free tile bits)
  bool relayout_;      // sweeps may store into the second buffer with the tile bits moved to the low positions
  bool keep_ghosts_;   // sharded handle: gates that do nothing on this rank stay in the list (GateRec::ghost)
  bool butterflies_ = env_flag("QH_BFLY", true);        // unit-entry butterfly ops (emit_ops_with)
  static constexpr size_t dense_weight_ = 1;             // score of a dense gate when choosing tile bits (diagonal = 1)
  // wave bits per tile (see plan_best): QH_WAVE_BITS pins it
  int max_wave_ = std::max(0, std::min(kMaxWaveBits, env_int("QH_WAVE_BITS", 1)));
  bool propagate_x_ = env_flag("QH_PROPAGATE_X", true);   // see propagate_x
  // Settled by measurement in rounds 2-4 and no longer switchable (profiles/r02..r04): in-place sweeps of
  // contiguous tiles store their wave exchanges un-undone; the least-used tile bits take the lane roles in op-heavy sweeps;
  // lane bits above the 2-MiB page come from the top; a chunk table needs two terms; controlled-phase ladders are factor
  // trees (DG_BITFAC); the butterflies' scalars ride on a phase factor; a relayout store puts the next sweep's targets
  // above the tile; the gates of a sweep are reordered for fewer layout exchanges; a lane <-> register exchange precedes
  // the phases of its gate.
  static constexpr bool store_swapped_ = true, lanes_by_count_ = true, lanes_high_ = true, bitfac_ = true, fold_sink_ = true;
  static constexpr int min_table_terms_ = 2;
  bool fold_pending_ = false;                            // ... still to be placed in the sweep being emitted
  bool keep_line_bits_ = false;                          // the flush runs on pre-chosen tiles: see emit_ops_with (relayout stores)
  double fold_re_ = 1, fold_im_ = 0;
  int lane_valu_ = env_int("QH_LANE_VALU", 1);          // 0 never, 1 by cost model (choose_lane_paths), 2 always (tests)
  bool defer_diag_ = env_flag("QH_DEFER_DIAG", true);   // see build_sweep
  static constexpr bool lookahead_ = true, reorder_ = true, lswap_early_ = true;   // see finish_relayout, reorder_for_fewer_swaps, emit_ops_with
  bool rot_fuse_ = env_flag("QH_ROT_FUSE", true);                 // a pi/4-type phase on a butterfly's target rides in the butterfly (emit_ops_with)
  int seats_ = env_int("QH_SEATS", 1);                             // op-heavy sweeps seat their lane-resident bits by cost (choose_seats)
  std::vector<uint64_t> alg_override_;
  std::vector<uint32_t> weight_;  // reference gate applications each pending record stands for
  std::vector<std::vector<int>> tiles_;   // tile bits of sweep 0, sweep 1, ... chosen for the whole flush (the tile search: set_tiles)

  static bool near(double a, double b) { return std::fabs(a - b) <= 4e-15; }
  static bool same_gate(const double a[8], const double b[8]) {
    for (int i = 0; i < 8; ++i) if (!near(a[i], b[i])) return false;
    return true;
  }
  static bool is_x(const double g[8]) {
    const double x[8] = {0, 0, 1, 0, 1, 0, 0, 0};
    for (int i = 0; i < 8; ++i) if (g[i] != x[i]) return false;
    return true;
  }
  // Uncontrolled X gates are not executed where they stand: an X on bit c is remembered as a
  // pending flip of that bit and pushed through the following gates -- a control on c changes
  // polarity (the reference wraps "controlled by |0>" as X.gate.X, circuit.py:166-169,207-215;
  // multi_control and the Grover oracles do the same around whole ladders), a gate acting on c
  // is conjugated (M -> X M X: rows and columns exchanged) -- and is emitted only at the end
  // of the flush if it is still pending.  Fewer dense gates, and bits that only ever saw X
  // need no place in a tile.
  void propagate_x(std::vector<GateRec> *pending, std::vector<uint64_t> *alg, std::vector<uint32_t> *weight,
                   uint64_t *noops) const {
    std::vector<GateRec> out;
    std::vector<uint64_t> oalg;
    std::vector<uint32_t> ow;
    uint64_t flip = 0, carry_alg = 0;
    uint32_t carry_w = 0;
    auto emit_x = [&](int bit) {
      GateRec x{};
      x.ctl_mask = 0;
      x.tgt = bit;
      x.g[2] = 1.0; x.g[4] = 1.0;
      out.push_back(x);
      oalg.push_back(carry_alg);
      ow.push_back(carry_w);
      carry_alg = 0;
      carry_w = 0;
    };
    for (size_t i = 0; i < pending->size(); ++i) {
      GateRec r = (*pending)[i];
      if (r.tgt >= 0 && r.ctl_mask == 0 && r.neg_mask == 0 && is_x(r.g) && !r.variant) {   // (an X under a shard-bit control is a gate like any other)
        flip ^= 1ull << r.tgt;
        carry_alg += (*alg)[i];
        carry_w += (*weight)[i];
        continue;
      }
      if (r.tgt >= 0 && ((flip >> r.tgt) & 1ull)) {   // X M X
        for (int k = 0; k < 2; ++k) {
          std::swap(r.g[k], r.g[6 + k]);
          std::swap(r.g[2 + k], r.g[4 + k]);
        }
      }
      uint64_t c = (r.ctl_mask | r.neg_mask) & flip;
      // A diagonal gate under zero-controls expands into 2^k phase terms with inverse factors:
      // with a zero on its diagonal (projectors) or past three zero-controls, the flips of its
      // control bits are executed here instead.
      // (a diagonal gate that touches a shard bit has rank-dependent entries: every rank takes the cautious branch)
      const bool singular = r.variant || (r.g[0] == 0.0 && r.g[1] == 0.0) || (r.g[6] == 0.0 && r.g[7] == 0.0);
      if (plan_diag(r) && c && (singular || popc((r.neg_mask ^ c) & (r.ctl_mask | r.neg_mask)) > 3)) {
        for (uint64_t t = c; t; t &= t - 1) {
          const int b = __builtin_ctzll(t);
          emit_x(b);
          flip &= ~(1ull << b);
        }
        c = 0;
      }
      const uint64_t to_neg = r.ctl_mask & c, to_pos = r.neg_mask & c;
      r.ctl_mask = (r.ctl_mask & ~to_neg) | to_pos;
      r.neg_mask = (r.neg_mask & ~to_pos) | to_neg;
      out.push_back(r);
      oalg.push_back((*alg)[i] + carry_alg);
      ow.push_back((*weight)[i] + carry_w);
      carry_alg = 0;
      carry_w = 0;
    }
    for (uint64_t t = flip; t; t &= t - 1) emit_x(__builtin_ctzll(t));
    if (carry_w) {               // X pairs that cancelled with nothing after them
      if (!ow.empty()) { ow.back() += carry_w; oalg.back() += carry_alg; }
      else *noops += carry_w;
    }
    pending->swap(out);
    alg->swap(oalg);
    weight->swap(ow);
  }

  static void mat2(const double a[8], const double b[8], double out[8]) {  // out = a * b
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 2; ++c) {
        double re = 0, im = 0;
        for (int k = 0; k < 2; ++k) {
          const double ar = a[2 * (2 * r + k)], ai = a[2 * (2 * r + k) + 1];
          const double br = b[2 * (2 * k + c)], bi = b[2 * (2 * k + c) + 1];
          re += ar * br - ai * bi;
          im += ar * bi + ai * br;
        }
        out[2 * (2 * r + c)] = re;
        out[2 * (2 * r + c) + 1] = im;
      }
  }

  // Peephole: the Sleator-Weinfurter doubly-controlled-U the reference emits for
  // every Toffoli / ccu (src/lib/circuit.py:227-246)
  //     C_a(V) t ; CX a->b ; C_b(V^-1) t ; CX a->b ; C_b(V) t         (V*V = U)
  // is, exactly, ONE gate U on t controlled by a AND b.  Recognising it removes the
  // two CX (which make b a dense target) and four of the five sweeps' worth of
  // work; the record keeps weight 5 and the five gates' minimal-touch bytes.
  void fuse_sleator_weinfurter(std::vector<GateRec> *pending, std::vector<uint64_t> *alg,
                               std::vector<uint32_t> *weight) const {
    std::vector<GateRec> out;
    std::vector<uint64_t> oalg;
    std::vector<uint32_t> ow;
    const std::vector<GateRec> &q = *pending;
    size_t i = 0;
    while (i < q.size()) {
      bool fused = false;
      if (i + 4 < q.size()) {
        const GateRec &g1 = q[i], &g2 = q[i + 1], &g3 = q[i + 2], &g4 = q[i + 3], &g5 = q[i + 4];
        const int t = g1.tgt, b = g2.tgt;
        const bool any_variant = g1.variant || g2.variant || g3.variant || g4.variant || g5.variant;
        if (!any_variant && t >= 0 && b >= 0 && b != t && g3.tgt == t && g5.tgt == t && g4.tgt == b && is_x(g2.g) &&
            is_x(g4.g) && g2.ctl_mask == g4.ctl_mask && g1.ctl_mask == g2.ctl_mask &&
            g3.ctl_mask == g5.ctl_mask && !((g1.ctl_mask >> b) & 1ull) && ((g3.ctl_mask >> b) & 1ull)) {
          const uint64_t common = g3.ctl_mask & ~(1ull << b);
          const uint64_t a_mask = g1.ctl_mask & ~common;
          if ((g1.ctl_mask & common) == common && popc(a_mask) == 1 && same_gate(g1.g, g5.g)) {
            double prod[8];
            mat2(g3.g, g1.g, prod);  // V^-1 * V must be the identity
            const double id[8] = {1, 0, 0, 0, 0, 0, 1, 0};
            if (same_gate(prod, id)) {
              GateRec r = g1;
              r.ctl_mask = g1.ctl_mask | (1ull << b);
              mat2(g1.g, g1.g, r.g);
              // snap entries that are zero/one up to rounding, so X stays a permutation
              for (int k = 0; k < 8; ++k) {
                if (std::fabs(r.g[k]) < 4e-16) r.g[k] = 0.0;
                if (near(r.g[k], 1.0)) r.g[k] = 1.0;
                if (near(r.g[k], -1.0)) r.g[k] = -1.0;
              }
              out.push_back(r);
              oalg.push_back((*alg)[i] + (*alg)[i + 1] + (*alg)[i + 2] + (*alg)[i + 3] + (*alg)[i + 4]);
              ow.push_back(5);
              i += 5;
              fused = true;
            }
          }
        }
      }
      if (!fused) {
        out.push_back(q[i]);
        oalg.push_back((*alg)[i]);
        ow.push_back((*weight)[i]);
        ++i;
      }
    }
    pending->swap(out);
    alg->swap(oalg);
    weight->swap(ow);
  }

  // One order-preserving pass over `pending` with a FIXED tile: a gate is taken if
  // (a) it commutes with every earlier gate that was skipped and (b) its target is
  // one of the tile bits `tilemask` (diagonal gates always fit).  Returns the
  // number of gates taken; optionally the indices taken.
  size_t pass(const std::vector<GateRec> &pending, uint64_t tilemask, size_t window,
              std::vector<uint8_t> *taken_flags) const {
    uint64_t blocked_all = 0;    // bits a skipped gate acts densely on
    uint64_t blocked_diag = 0;   // bits a skipped gate acts diagonally on
    size_t count = 0, score = 0;
    const size_t n = std::min(window, pending.size());
    for (size_t i = 0; i < n; ++i) {
      const GateRec &r = pending[i];
      const bool diag = plan_diag(r);
      const uint64_t tb = (r.tgt >= 0) ? (1ull << r.tgt) : 0;
      const uint64_t dense_bits = diag ? 0 : tb;
      const uint64_t diag_bits = r.ctl_mask | r.neg_mask | (diag ? tb : 0);
      const bool can_pass = !(dense_bits & (blocked_all | blocked_diag)) && !(diag_bits & blocked_all);
      const bool fits = (count < (size_t)kMaxSweepOps) &&
                        (diag || ((tilemask >> r.tgt) & 1ull));
      if (can_pass && fits) {
        ++count;
        score += diag ? 1 : dense_weight_;
        if (taken_flags) (*taken_flags)[i] = 1;
      } else {
        blocked_all |= dense_bits;
        blocked_diag |= diag_bits;
        // everything that can still be taken must avoid the blocked bits: once all
        // local bits are blocked densely nothing further can pass
      }
    }
    return score;
  }

  // Tile = index bits {0,1,2} (always) + 3 "lane-high" bits + up to 5 register bits.
  // With lane-high = {3,4,5} every wave load is one contiguous 1-KiB run; with other
  // lane-high bits it is eight 128-byte lines (measured 5-10% slower,
  // tools/membench/splitlane.hip) but the sweep can then take EIGHT new target bits
  // instead of five.  Split a chosen bit set into (lane-high, register) bits; false
  // if it does not fit.
  // With wave bits the tile grows by up to two more bits (the HIGHEST chosen ones: a wave
  // bit costs nothing in the memory access pattern of a wave).  Order of preference for the
  // bits above bit 5: registers, wave bits, split lanes.
  bool assign_bits(const std::vector<int> &sel, std::vector<int> *lanehi, std::vector<int> *regs,
                   std::vector<int> *waves) const {
    std::vector<int> low, other;
    for (int b : sel) ((b >= lane_low_ && b < kLaneBits) ? low : other).push_back(b);
    std::sort(low.begin(), low.end());
    std::sort(other.begin(), other.end());
    const int extra = std::max<int>(0, (int)other.size() - rb_cap_);
    const int nw = std::min(extra, max_wave_);
    const int need_move = extra - nw;
    if (need_move > 0 && !split_lanes_) return false;
    if ((int)low.size() + need_move > lane_hi_) return false;
    if (need_move > 0) {
      // split-lane tile: the LOWEST bits go to the wave id, the highest to the registers
      // (tools/geom_scan_waves.py, targets 13..22: 7.9 ms vs 8.8 ms the other way round)
      waves->assign(other.begin(), other.begin() + nw);
      other.erase(other.begin(), other.begin() + nw);
    } else {
      // contiguous lanes: highest bits to the wave id (tools/geom_scan_waves7.py)
      waves->assign(other.end() - nw, other.end());
      other.resize(other.size() - nw);
    }
    *lanehi = low;
    if (need_move > 0 && other[0] >= 17 && lanes_high_) {
      // every candidate is above the 2-MiB page (each line of a load in another page anyway):
      // then the HIGHEST bits (up to kMaxLaneHiBit: lane offsets are 32-bit byte offsets) make
      // the better lane bits (tools/geom_scan_wave1.py, targets 21..29: 6.9 ms vs 7.25 ms; tried again in round 5 on
      // Grover-34's far tiles, targets 17..26: lowest bits as lanes 101 ms instead of 103 per sweep, highest bits as waves
      // 103-105: within 2 %, rule kept -- profiles/r05/grover34_per_sweep.txt)
      std::vector<int> rest;
      int moved = 0;
      for (size_t k = other.size(); k-- > 0;) {
        if (moved < need_move && other[k] <= kMaxLaneHiBit) { lanehi->push_back(other[k]); moved++; }
        else rest.push_back(other[k]);
      }
      if (moved < need_move) return false;
      std::sort(rest.begin(), rest.end());
      *regs = rest;
    } else {
      for (int k = 0; k < need_move; ++k)
        if (other[k] > kMaxLaneHiBit) return false;
      lanehi->insert(lanehi->end(), other.begin(), other.begin() + need_move);
      regs->assign(other.begin() + need_move, other.end());
    }
    for (int b = lane_low_; b < kLaneBits && (int)lanehi->size() < lane_hi_; ++b)   // spare slots: the contiguous positions
      if (std::find(lanehi->begin(), lanehi->end(), b) == lanehi->end()) lanehi->push_back(b);
    std::sort(lanehi->begin(), lanehi->end());
    return (int)lanehi->size() == lane_hi_;
  }

  static uint64_t mask_of(const std::vector<int> &bits) {
    uint64_t m = 0;
    for (int b : bits) m |= 1ull << b;
    return m;
  }

  // assign_bits() as a predicate on a bit mask (no vectors: called for every candidate of every round)
  bool tile_fits(uint64_t selmask) const {
    const uint64_t lowmask = ((1ull << kLaneBits) - 1) & ~((1ull << lane_low_) - 1);
    const int nlow = popc(selmask & lowmask);
    uint64_t other = selmask & ~lowmask;
    const int nother = popc(other);
    const int extra = std::max(0, nother - rb_cap_);
    const int nw = std::min(extra, max_wave_);
    const int need_move = extra - nw;
    if (need_move == 0) return nlow <= lane_hi_;
    if (!split_lanes_ || nlow + need_move > lane_hi_) return false;
    for (int k = 0; k < nw; ++k) other &= other - 1;          // split-lane tile: the lowest bits go to the wave id
    const uint64_t reach = (kMaxLaneHiBit >= 63) ? ~0ull : ((2ull << kMaxLaneHiBit) - 1);
    if (__builtin_ctzll(other) >= 17 && lanes_high_) return popc(other & reach) >= need_move;
    uint64_t t = other;
    for (int k = 0; k < need_move; ++k) {
      if (__builtin_ctzll(t) > kMaxLaneHiBit) return false;
      t &= t - 1;
    }
    return true;
  }

  // pass() on precomputed per-gate masks (the inner loop of the tile-bit selection).  A candidate bit c can only
  // change the outcome from the first dense gate on c that the current tile skips although nothing blocks it:
  // the baseline pass records the state in front of that gate per bit, and a candidate's pass resumes there.
  size_t pass_fast(const std::vector<PassRec> &rec, uint64_t tilemask, const PassState &st, PassState *snaps = nullptr) const {
    uint64_t blocked_all = st.blocked_all, blocked_diag = st.blocked_diag;
    size_t count = st.count, score = st.score;
    const size_t n = rec.size();
    for (size_t i = st.idx; i < n; ++i) {
      const PassRec &r = rec[i];
      const bool can_pass = !(r.dense_bits & (blocked_all | blocked_diag)) && !(r.diag_bits & blocked_all);
      const bool fits = count < (size_t)kMaxSweepOps && (!r.dense_bits || (tilemask & r.dense_bits));
      if (can_pass && fits) {
        ++count;
        score += r.score;
      } else {
        if (snaps && can_pass && r.dense_bits && !snaps[r.tgt].valid) {
          PassState &p = snaps[r.tgt];
          p.idx = i; p.blocked_all = blocked_all; p.blocked_diag = blocked_diag; p.count = count; p.score = score; p.valid = true;
        }
        blocked_all |= r.dense_bits;
        blocked_diag |= r.diag_bits;
      }
    }
    return score;
  }

  void select_tile_bits(const std::vector<GateRec> &pending, std::vector<int> *sel_out) const {
    const size_t window = std::min<size_t>(pending.size(), 4096);
    const uint64_t always = (1ull << lane_low_) - 1;
    std::vector<int> cand;       // dense target bits above bit 2, in order of first use
    std::vector<PassRec> rec(window);
    uint64_t seen = 0;
    for (size_t i = 0; i < window; ++i) {
      const GateRec &r = pending[i];
      const bool diag = plan_diag(r);
      const uint64_t tb = (r.tgt >= 0) ? (1ull << r.tgt) : 0;
      rec[i] = PassRec{diag ? 0 : tb, r.ctl_mask | r.neg_mask | (diag ? tb : 0), (uint32_t)(diag ? 1 : dense_weight_), r.tgt};
      if (!diag && r.tgt >= lane_low_ && !((seen >> r.tgt) & 1ull)) {
        cand.push_back(r.tgt);
        seen |= 1ull << r.tgt;
      }
    }
    std::vector<int> sel;
    uint64_t selmask = 0;
    PassState snaps[64];
    size_t best_total = pass_fast(rec, always, PassState(), snaps);
    while ((int)sel.size() < lane_hi_ + rb_cap_ + max_wave_) {
      int best_bit = -1;
      size_t best = best_total;
      for (int c : cand) {
        if ((selmask >> c) & 1ull) continue;
        if (!snaps[c].valid) continue;                       // no gate on c waits for the tile: the score cannot change
        if (!tile_fits(selmask | (1ull << c))) continue;
        const size_t sc = pass_fast(rec, always | selmask | (1ull << c), snaps[c]);
        if (sc > best) { best = sc; best_bit = c; }
      }
      if (best_bit < 0) break;
      sel.push_back(best_bit);
      selmask |= 1ull << best_bit;
      for (PassState &p : snaps) p.valid = false;
      best_total = pass_fast(rec, always | selmask, PassState(), snaps);
    }
    sel_out->swap(sel);
  }

  // Choose the tile bits of a sweep greedily by SIMULATION: add, one at a time, the
  // candidate bit that lets the most queued gates run in this sweep (first-come
  // order breaks ties).  For a QFT this reproduces "the next target bits in order";
  // for layered circuits (supremacy, Grover ladders) it picks qubits whose gates
  // unblock each other instead of the first ones that happen to come up.
  SweepPlan build_sweep(const std::vector<GateRec> &pending, const std::vector<uint64_t> &alg,
                        std::vector<GateRec> *rest, std::vector<uint64_t> *rest_alg,
                        std::vector<uint32_t> *rest_w, const std::vector<int> *forced_sel = nullptr) {
    SweepPlan sp;
    const uint64_t always = (1ull << lane_low_) - 1;
    std::vector<int> sel, lanehi, regs, waves;
    if (forced_sel) sel = *forced_sel;
    else select_tile_bits(pending, &sel);
    if (!assign_bits(sel, &lanehi, &regs, &waves) && forced_sel) {     // (a tile the search proposed does not fit after all)
      sel.clear(); lanehi.clear(); regs.clear(); waves.clear();
      select_tile_bits(pending, &sel);
      assign_bits(sel, &lanehi, &regs, &waves);
    }
    uint64_t regmask = mask_of(regs);
    uint64_t lanemask = always | mask_of(lanehi);
    std::vector<uint8_t> flags(pending.size(), 0);
    pass(pending, lanemask | regmask | mask_of(waves), pending.size(), &flags);
    // A diagonal gate that no dense op of THIS sweep waits for, but a dense gate of a later
    // sweep does, is left for that sweep: there its factor joins the group that is applied
    // in front of that dense op anyway (one more table factor), instead of costing a group
    // of its own here.  (Leaving it is legal: it commutes with everything taken after it.)
    if (defer_diag_) {
      uint64_t future = 0, later_targets = 0;
      for (size_t i = 0; i < pending.size(); ++i)
        if (!flags[i] && !plan_diag(pending[i]) && pending[i].tgt >= 0) future |= 1ull << pending[i].tgt;
      for (size_t i = pending.size(); i-- > 0;) {
        if (!flags[i]) continue;
        const GateRec &r = pending[i];
        if (!plan_diag(r)) { later_targets |= 1ull << r.tgt; continue; }
        const uint64_t bits = r.ctl_mask | r.neg_mask | (r.tgt >= 0 ? (1ull << r.tgt) : 0);
        if (!(bits & later_targets) && (bits & future)) flags[i] = 0;
      }
    }
    std::vector<const GateRec *> taken;
    bool any_dense = false;
    for (size_t i = 0; i < pending.size(); ++i) {
      if (flags[i]) {
        taken.push_back(&pending[i]);
        if (!plan_diag(pending[i])) any_dense = true;
        sp.gates += weight_[i];
        sp.alg_bytes += alg[i];
      } else {
        rest->push_back(pending[i]);
        rest_alg->push_back(alg[i]);
        rest_w->push_back(weight_[i]);
      }
    }
    if (taken.empty()) {  // cannot happen (the first pending gate always fits some tile), but never loop forever
      taken.push_back(&pending[0]);
      rest->erase(rest->begin());
      rest_alg->erase(rest_alg->begin());
      rest_w->erase(rest_w->begin());
      if (!plan_diag(pending[0]) && !((lanemask >> pending[0].tgt) & 1ull)) {
        regs.assign(1, pending[0].tgt);
        waves.clear();
        regmask = 1ull << pending[0].tgt;
      }
      sp.gates = weight_[0];
      sp.alg_bytes = alg[0];
    }
    // Op-heavy sweeps (many dense gates per tile: supremacy layers) are bound by instruction
    // issue, and a lane-bit gate costs 2-4x a register-bit gate: among the bits of a split-lane
    // tile the ones with the FEWEST dense gates take the lane roles.  (The tile mask, hence the
    // set of gates taken, does not change.)
    if (lanes_by_count_) {
      std::vector<int> cnt(64, 0);
      int ndense = 0;
      for (const GateRec *r : taken)
        if (!plan_diag(*r) && r->tgt >= 0) { cnt[r->tgt]++; ndense++; }
      if (ndense >= 16) {
        for (int iter = 0; iter < lane_hi_; ++iter) {
          int bl = -1, br = -1, gain = 0;
          for (size_t i = 0; i < lanehi.size(); ++i) {
            if (lanehi[i] < kLaneBits) continue;            // 3,4,5: fixed by contiguity
            for (size_t j = 0; j < regs.size(); ++j)
              if (regs[j] <= kMaxLaneHiBit && cnt[lanehi[i]] - cnt[regs[j]] > gain) {
                gain = cnt[lanehi[i]] - cnt[regs[j]];
                bl = (int)i;
                br = (int)j;
              }
          }
          if (gain < 2) break;
          std::swap(lanehi[bl], regs[br]);
        }
        std::sort(lanehi.begin(), lanehi.end());
        std::sort(regs.begin(), regs.end());
        regmask = mask_of(regs);
        lanemask = always | mask_of(lanehi);
      }
    }
    // drop register bits that ended up unused (a later candidate made them moot)
    {
      uint64_t used = 0;
      for (const GateRec *r : taken)
        if (!plan_diag(*r) && r->tgt >= 0) used |= 1ull << r->tgt;
      std::vector<int> keep;
      for (int p : regs) if ((used >> p) & 1ull) keep.push_back(p);
      regs.swap(keep);
      keep.clear();
      for (int p : waves) if ((used >> p) & 1ull) keep.push_back(p);
      waves.swap(keep);
      if (regs.empty() && !waves.empty()) {   // OP_WSWAP needs a register bit to exchange with
        regs.push_back(waves.back());
        waves.pop_back();
      }
    }
    const uint64_t wavemask = mask_of(waves);
    // Bits every taken gate requires to be 1 (only useful above the lane bits and
    // outside the register tile): fold them into the tile enumeration so that the
    // untouched part of the state is never read.
    uint64_t common = ~0ull;
    for (const GateRec *r : taken) {
      uint64_t req = r->ctl_mask;
      if (plan_diag(*r) && r->tgt >= 0 && is_one(r->g[0], r->g[1]) && !r->variant) req |= 1ull << r->tgt;   // (a variant gate's entries differ between ranks)
      common &= req;
    }
    regmask = 0;
    for (int p : regs) regmask |= 1ull << p;
    common &= ~lanemask & ~regmask & ~wavemask & ((1ull << nloc_) - 1);
    // keep enough free bits for the tile: need rb register bits among non-fixed bits
    int rb = std::max<int>((int)regs.size(), std::min(rb_cap_, any_dense ? rb_cap_ : 3));
    const int nwv = (int)waves.size();
    while (popc(common) > 0 && nloc_ - kLaneBits - popc(common) - nwv < rb) common &= common - 1;
    while (popc(common) + rb + lane_hi_ + nwv > kMaxInsertBits) common &= common - 1;  // the rest stay per-op controls
    rb = std::min(rb, nloc_ - kLaneBits - popc(common) - nwv);
    sp.fixed_ones = common;
    // pad the register tile with free bits: 10..13 first (a tile whose spare register
    // bits sit there streams ~10% faster than with bits 6..9 next to contiguous lanes
    // or with bits >= 18: tools/geom_scan.py), then the lowest free ones
    auto pad = [&](int p) {
      if ((int)regs.size() < rb && p >= lane_low_ && p < nloc_ && !((regmask | common | lanemask | wavemask) >> p & 1ull)) {
        regs.push_back(p);
        regmask |= 1ull << p;
      }
    };
    for (int p = 10; p < 14; ++p) pad(p);
    for (int p = lane_low_; p < nloc_; ++p) pad(p);
    std::sort(regs.begin(), regs.end());
    sp.rb = (int)regs.size();
    for (int k = 0; k < sp.rb; ++k) sp.regpos[k] = regs[k];
    sp.lane_low = lane_low_;
    for (int k = 0; k < kLaneHi; ++k) sp.lanehi[k] = k < lane_hi_ ? lanehi[k] : -1;
    for (int i = 0; i < kLaneBits; ++i) sp.seat[i] = i < lane_low_ ? i : lanehi[i - lane_low_];
    choose_seats(taken, &sp);
    sp.nwave = nwv;
    for (int k = 0; k < nwv; ++k) sp.wavepos[k] = waves[k];
    sp.ntiles = 1ull << (nloc_ - kLaneBits - sp.rb - popc(common));
    sp.swept_bytes = (sp.ntiles << (kLaneBits + sp.rb)) * amp_bytes_ * 2;
    emit_ops(taken, &sp);
    return sp;
  }

  // Seats of the six lane-resident bits (SweepPlan::seat) of an op-heavy sweep, by what a dense gate costs there: the two
  // bits with the most dense gates take lane bits 5 and 4 (v_permlane swap, then register butterflies for as long as the
  // bit stays in a register), the one with the fewest lane bit 2 (two DPP moves per dword), the rest lane bits 3, 0, 1.
  // Sweeps with few dense gates keep the plain map (index bit i on lane bit i: the best one for the memory system).
  // Ghost gates count like live ones (the seats shape the exchange geometry's layouts: rank-invariant).
  void choose_seats(const std::vector<const GateRec *> &taken, SweepPlan *sp) const {
    if (!seats_ || amp_bytes_ != 16) return;
    int cnt[64] = {0}, ndense = 0;
    for (const GateRec *r : taken)
      if (!plan_diag(*r) && r->tgt >= 0) { cnt[r->tgt]++; ndense++; }
    if (ndense < 16 && seats_ < 2) return;           // (QH_SEATS=2: every sweep, whatever it saves -- tests)
    static const int kSeatCost[kLaneBits] = {206, 206, 334, 206, 100, 100};   // (4, 5: exchange shared by the gates that follow)
    static const int kSeatOrder[kLaneBits] = {5, 4, 3, 0, 1, 2};              // busiest bit first
    int bits[kLaneBits];
    for (int i = 0; i < kLaneBits; ++i) bits[i] = sp->seat[i];
    std::stable_sort(bits, bits + kLaneBits, [&](int a, int b) { return cnt[a] > cnt[b]; });
    long before = 0, after = 0;
    for (int i = 0; i < kLaneBits; ++i) before += (long)cnt[sp->seat[i]] * kSeatCost[i];
    for (int i = 0; i < kLaneBits; ++i) after += (long)cnt[bits[i]] * kSeatCost[kSeatOrder[i]];
    if (before - after < 128 && seats_ < 2) return;
    for (int i = 0; i < kLaneBits; ++i) sp->seat[kSeatOrder[i]] = bits[i];
  }

  int reg_index(const SweepPlan &sp, int pos) const {
    for (int k = 0; k < sp.rb; ++k) if (sp.regpos[k] == pos) return k;
    return -1;
  }

  static int wave_index(const SweepPlan &sp, int pos) {
    for (int k = 0; k < sp.nwave; ++k) if (sp.wavepos[k] == pos) return k;
    return -1;
  }

  // lane-bit index (0..5) of a physical index bit, -1 if it is not a lane bit of this tile
  static int lane_index(const SweepPlan &sp, int pos) {
    for (int i = 0; i < kLaneBits; ++i) if (sp.seat[i] == pos) return i;
    return -1;
  }

  // Split an index-bit mask into (lane part as LANE-INDEX bits, the same lane part
  // as physical bits, register part, outside part); bits fixed to one by the
  // enumeration are dropped (always satisfied).
  void split_mask(const SweepPlan &sp, uint64_t m, uint32_t *lane, uint64_t *lane_phys, uint32_t *reg,
                  uint64_t *outside, bool drop_fixed = true) const {
    if (drop_fixed) m &= ~sp.fixed_ones;
    *lane = 0;
    *lane_phys = 0;
    *reg = 0;
    *outside = 0;
    for (uint64_t t = m; t; t &= t - 1) {
      const int b = __builtin_ctzll(t);
      const int li = lane_index(sp, b);
      if (li >= 0) { *lane |= 1u << li; *lane_phys |= 1ull << b; continue; }
      const int ri = reg_index(sp, b);
      if (ri >= 0) *reg |= 1u << ri; else *outside |= 1ull << b;
    }
  }

  struct PTerm { uint64_t mask; double re, im; };

  // How many lane butterflies leave the LDS (ds_bpermute) path, by lane-bit class: lane bits 0, 1 and 3
  // reach their partner with ONE DPP move per dword (quad_perm, row_ror:8: "01"), lane bit 2 needs two ("23").
  struct LaneChoice { int dpp01 = 0, dpp23 = 0, lswap = 0, real01 = 0, real23 = 0; };
  static bool one_step_dpp(uint32_t lane_bit) { return lane_bit < 2 || lane_bit == 3; }

  // ds_bpermute_b32 issues once per ~6.1 cycles per CU (tools/membench/bpermbench: the LDS pipe is shared
  // by the four SIMDs and their twelve waves).  Lane butterflies can instead fetch the partner by DPP moves
  // (lane bits 0..3) or exchange lane bit 4/5 with a register bit by v_permlane{16,32}_swap and run as
  // register butterflies: more VALU instructions, no LDS.  What decides is not the pipes' throughput but the
  // LATENCY of a wave's op stream: with three waves per SIMD a sweep takes (wave lifetime per tile) x 171
  // tiles, and under this load the clock drops to 1.7-2.1 GHz.  Measured inside the kernel (s_memtime per op,
  // tools/probes/prof_island.sh): a bpermute butterfly keeps its wave for 3 200-7 700 cycles when the other
  // eleven waves of the CU queue at the same pipe, a DPP one for 1 300-3 000, swap + register butterfly for
  // 1 800-2 500.  So a sweep whose LDS pipe would be busy for more than 2 500 cycles per tile
  // (default 2 500: a third of the HBM time of a tile -- three lane butterflies and a wave exchange) moves
  // ALL its lane butterflies to the VALU; lighter sweeps keep the LDS path (30-qubit QFT first sweep
  // 6.7 -> 6.35 ms, its other two 6.05 -> 5.87; supremacy 47.9 -> 47.1; QFT-33 159 -> 153).
  LaneChoice choose_lane_paths(const SweepPlan &sp) const {
    const double kLds = 6.1 * (amp_bytes_ == 16 ? 128 : 64);
    const double dw = amp_bytes_ == 16 ? 1.0 : 0.5;
    double lds = 0;
    for (const SweepOp &o : sp.ops) {
      if (o.kind == OP_WSWAP) lds += 256 * dw;      // 16 ds_write_b128 + 16 ds_read_b128, 8 cycles each
      else if (o.kind == OP_DENSE_LANE) lds += kLds;
    }
    const double floor_cycles = 2500;
    LaneChoice ch;
    // (Round 4 tried the middle ground -- leave butterflies on the otherwise idle LDS pipe up to a budget of pipe cycles
    // per tile, cheapest VALU alternative moved first -- on the premise that op-heavy sweeps are bound by VALU issue:
    // every butterfly left on ds_bpermute made every workload slower, supremacy-30 33.5 -> 33.6 / 34.0 / 34.4 / 40.8 ms
    // for 1600 / 2500 / 3500 / unlimited cycles, QFT-30 16.5 -> 17.2-17.6: profiles/r04/lane_lds_budget_ab.txt.  The
    // dependent latency of 128 shuffles behind eleven other waves is what costs, not the pipe's throughput.)
    if (lds > floor_cycles) ch.dpp01 = ch.dpp23 = ch.lswap = ch.real01 = ch.real23 = 1 << 20;
    return ch;
  }

  // Diagonal gates are placed LAZILY: a phase term stays pending until a dense
  // op targets one of its bits (or the sweep ends), so the terms of many
  // reference gates meet in few DIAG ops and merge into tables.
  // The gates of a sweep in an order that exchanges layouts less often.  Only five (six) of a tile's ten or more
  // bits sit in registers at a time; a butterfly on lane bit 4 / 5 or on a wave bit first trades places with a register
  // bit (LSWAP 72 VALU instructions, WSWAP 64 LDS accesses + a barrier).  A layered circuit visits its qubits round
  // robin, so in queue order almost every dense gate of a supremacy sweep outside the registers costs an exchange
  // (12 per sweep of ~37 dense gates).  Gates that commute may run in any order: list scheduling over the dependency
  // graph (two gates depend on each other when one acts densely on a bit the other touches) -- take, in queue order,
  // every ready gate whose target is in a register or on lane bits 0..3 (or that is diagonal); only when none is left
  // take the first ready gate that needs an exchange, with the Belady victim.  Ghosts and their bits take part like any
  // gate (rank-invariant).  The emission below then decides the real exchanges on the new order.
  std::vector<const GateRec *> reorder_for_fewer_swaps(const std::vector<const GateRec *> &taken, const SweepPlan &sp) const {
    const size_t n = taken.size();
    if (!reorder_ || n < 4 || n > 600 || (sp.nwave == 0 && sp.nlanehi() < 2)) return taken;
    std::vector<uint64_t> dense(n), all(n);
    for (size_t i = 0; i < n; ++i) {
      const GateRec &r = *taken[i];
      const uint64_t tb = r.tgt >= 0 ? (1ull << r.tgt) : 0;
      dense[i] = plan_diag(r) ? 0 : tb;
      all[i] = tb | r.ctl_mask | r.neg_mask;
    }
    std::vector<int> ndep(n, 0);
    std::vector<std::vector<uint32_t>> succ(n);
    for (size_t j = 0; j < n; ++j)
      for (size_t i = 0; i < j; ++i)
        if ((dense[i] & all[j]) || (dense[j] & all[i])) { succ[i].push_back((uint32_t)j); ndep[j]++; }
    uint64_t regs = 0, cheap = (1ull << sp.lane_low) - 1;        // where a dense gate needs no exchange
    for (int k = 0; k < sp.rb; ++k) regs |= 1ull << sp.regpos[k];
    cheap = 0;
    for (int i = 0; i < 4; ++i) cheap |= 1ull << sp.seat[i];                                          // lane bits 0..3: DPP, no exchange
    std::vector<uint8_t> done(n, 0);
    std::vector<const GateRec *> out;
    out.reserve(n);
    size_t scheduled = 0;
    auto take = [&](size_t i) {
      done[i] = 1;
      out.push_back(taken[i]);
      ++scheduled;
      for (uint32_t j : succ[i]) ndep[j]--;
    };
    while (scheduled < n) {
      bool progress = true;
      while (progress) {                       // everything that runs where the bits are now, in queue order
        progress = false;
        for (size_t i = 0; i < n; ++i)
          if (!done[i] && ndep[i] == 0 && (!dense[i] || (dense[i] & (regs | cheap)))) { take(i); progress = true; }
      }
      if (scheduled == n) break;
      size_t pick = n;
      for (size_t i = 0; i < n; ++i) if (!done[i] && ndep[i] == 0) { pick = i; break; }
      if (pick == n) return taken;             // (cannot happen: the dependency graph follows the queue order)
      // the register bit that gives way: the one used as a dense target LATEST among the gates still to come
      int victim = -1;
      size_t victim_next = 0;
      for (uint64_t t = regs; t; t &= t - 1) {
        const int b = __builtin_ctzll(t);
        size_t next = n + 1;
        for (size_t j = 0; j < n; ++j) if (!done[j] && j != pick && (dense[j] >> b & 1ull)) { next = j; break; }
        if (victim < 0 || next > victim_next) { victim = b; victim_next = next; }
      }
      if (victim >= 0) regs = (regs & ~(1ull << victim)) | dense[pick];
      take(pick);
    }
    return out;
  }

  void emit_ops(const std::vector<const GateRec *> &taken_in, SweepPlan *sp) {
    const std::vector<const GateRec *> taken = reorder_for_fewer_swaps(taken_in, *sp);
    emit_ops_with(taken, sp, LaneChoice{});
    if (lane_valu_) {
      LaneChoice ch = choose_lane_paths(*sp);     // (ghost ops count: the choice must not depend on the rank)
      if (lane_valu_ == 2) ch.dpp01 = ch.dpp23 = ch.lswap = ch.real01 = ch.real23 = 1 << 20;
      if (ch.dpp01 + ch.dpp23 + ch.lswap + ch.real01 + ch.real23 != 0) {
        sp->ops.clear(); sp->groups.clear(); sp->oterms.clear(); sp->tables.clear(); sp->ltabs.clear();
        emit_ops_with(taken, sp, ch);
      }
    }
    sp->ops.erase(std::remove_if(sp->ops.begin(), sp->ops.end(), [](const SweepOp &o) { return (o.flags & OPF_GHOST) != 0; }),
                  sp->ops.end());
  }

  void emit_ops_with(const std::vector<const GateRec *> &taken, SweepPlan *sp, LaneChoice ch) {
    fold_pending_ = false;
    // Butterflies (unit-entry gates c*M, see butterfly_variant): the scalars c are multiplied
    // into ONE uncontrolled dense gate of the sweep (a scalar commutes with everything) -- the
    // last one, which runs on the general path.  Without such a sink nothing is converted.
    std::vector<int8_t> role(taken.size(), 0);   // 1 = butterfly, 2 = sink
    std::vector<double> sink_re(taken.size(), 1.0), sink_im(taken.size(), 0.0);
    {
      int last = -1, nbf = 0;
      std::vector<int> cand;
      for (size_t i = 0; i < taken.size(); ++i) {
        const GateRec *r = taken[i];
        if (plan_diag(*r) || (r->ctl_mask & ~sp->fixed_ones) || r->neg_mask || r->variant) continue;
        last = (int)i;
        if (butterflies_ && butterfly_variant(r->g) >= 0) cand.push_back((int)i);
      }
      for (int i : cand) if (i != last) nbf++;
      if (nbf >= 1) {
        // Between two sinks the stored amplitudes are the true ones divided by the scalars
        // moved so far (sqrt 2 per h / v / yroot): a butterfly becomes an intermediate sink
        // before that factor leaves the comfortable range of the element type.
        const double limit = amp_bytes_ == 16 ? 0x1p100 : 0x1p24;
        double pr = 1, pi = 0, growth = 1;
        for (int i : cand) {
          if (i == last) continue;
          const double mag = std::hypot(taken[i]->g[0], taken[i]->g[1]);
          if (growth / mag > limit || growth / mag < 1.0 / limit) {
            role[i] = 2;
            sink_re[i] = pr; sink_im[i] = pi;
            pr = 1; pi = 0; growth = 1;
            continue;
          }
          role[i] = 1;
          cmul_acc(&pr, &pi, taken[i]->g[0], taken[i]->g[1]);
          growth /= mag;
        }
        role[last] = 2;
        sink_re[last] = pr; sink_im[last] = pi;
        // If the last gate is a butterfly itself, the whole product can ride on a phase factor that some DIAG
        // op of the sweep applies to every amplitude anyway (flush_diag: a group without register mask): every
        // such gate then costs additions only (supremacy: 400 FP64 instructions per tile less, QFT: 96).
        if (fold_sink_ && butterflies_ && butterfly_variant(taken[last]->g) >= 0) {
          role[last] = 1;
          cmul_acc(&pr, &pi, taken[last]->g[0], taken[last]->g[1]);
          fold_pending_ = true;
          fold_re_ = pr; fold_im_ = pi;
        }
      }
    }
    // current tile geometry: OP_LSWAP / OP_WSWAP exchange a lane / wave bit with a register bit on the fly
    SweepPlan geom;
    geom.rb = sp->rb;
    geom.fixed_ones = sp->fixed_ones;
    memcpy(geom.regpos, sp->regpos, sizeof geom.regpos);
    memcpy(geom.lanehi, sp->lanehi, sizeof geom.lanehi);
    memcpy(geom.seat, sp->seat, sizeof geom.seat);
    geom.lane_low = sp->lane_low;
    geom.nwave = sp->nwave;
    memcpy(geom.wavepos, sp->wavepos, sizeof geom.wavepos);
    // layout exchanges so far (undone in reverse): kind (0 lane / 1 wave), its bit index, register bit
    struct Swap { int wave, idx, r; };
    std::vector<Swap> swaps;
    auto lswap = [&](int li, int r) {
      SweepOp op{};
      op.kind = OP_LSWAP;
      op.tb = (uint32_t)li;
      op.cm_reg = (uint32_t)r;
      // the kernel moves the thread's own bit from the lane's index position to the register's (its thread
      // index keeps describing the amplitudes it holds: lane / outside controls and OP_WSWAP rely on it)
      op.n_groups = (uint32_t)geom.seat[li];
      op.cm_thread = (1ull << geom.seat[li]) | (1ull << geom.regpos[r]);
      sp->ops.push_back(op);
      std::swap(geom.seat[li], geom.regpos[r]);
    };
    auto wswap = [&](int wi, int r) {
      SweepOp op{};
      op.kind = OP_WSWAP;
      op.tb = (uint32_t)wi;
      op.cm_reg = (uint32_t)r;
      op.cm_thread = (1ull << geom.wavepos[wi]) | (1ull << geom.regpos[r]);
      sp->ops.push_back(op);
      std::swap(geom.wavepos[wi], geom.regpos[r]);
    };
    auto restore_layout = [&]() {
      while (!swaps.empty()) {
        const Swap w = swaps.back();
        if (w.wave) wswap(w.idx, w.r); else lswap(w.idx, w.r);
        swaps.pop_back();
      }
    };
    // the register bit to give up in an exchange: the one whose qubit is a dense target
    // again LATEST among the gates of this sweep (never, if possible) -- Belady
    size_t gi_now = 0;
    auto victim_reg = [&]() {
      int best = 0;
      size_t best_next = 0;
      for (int r = 0; r < geom.rb; ++r) {
        size_t next = taken.size() + 1;
        for (size_t j = gi_now + 1; j < taken.size(); ++j)
          if (!plan_diag(*taken[j]) && taken[j]->tgt == geom.regpos[r]) { next = j; break; }
        // ties: a register no exchange has touched yet (its exchange can then be undone, or left, on its own)
        bool used = false, best_used = false;
        for (const Swap &w : swaps) { used |= w.r == r; best_used |= w.r == best; }
        if (next > best_next || (next == best_next && best_used && !used)) { best_next = next; best = r; }
      }
      return best;
    };
    // Undo the lane exchanges only (in-place store with the wave bits left exchanged): possible when no
    // later wave exchange went through the same register bit; false = the caller restores everything.
    auto undo_lane_swaps = [&]() {
      for (size_t i = 0; i < swaps.size(); ++i)
        if (!swaps[i].wave)
          for (size_t j = i + 1; j < swaps.size(); ++j)
            if (swaps[j].wave && swaps[j].r == swaps[i].r) return false;
      for (size_t i = swaps.size(); i-- > 0;)
        if (!swaps[i].wave) {
          lswap(swaps[i].idx, swaps[i].r);
          swaps.erase(swaps.begin() + (long)i);
        }
      return true;
    };
    std::vector<PTerm> pending;
    auto add_pending = [&](uint64_t mask, double re, double im) {
      if (is_one(re, im)) return;
      for (auto &t : pending) if (t.mask == mask) {
        const double nr = t.re * re - t.im * im, ni = t.re * im + t.im * re;
        t.re = nr; t.im = ni;
        return;
      }
      pending.push_back(PTerm{mask, re, im});
    };
    for (size_t gi = 0; gi < taken.size(); ++gi) {
      const GateRec *r = taken[gi];
      const bool diag = plan_diag(*r);
      if (diag && r->ghost) continue;           // (no phase here; diagonal gates never move the layout)
      if (!diag) {
        // a target that lives in the wave id comes into a register bit first: the phases
        // waiting for this gate are then in-tile factors instead of one group per partner bit
        gi_now = gi;
        const int wi = wave_index(geom, r->tgt);
        if (wi >= 0) {
          const int vr = victim_reg();
          wswap(wi, vr);
          swaps.push_back(Swap{1, wi, vr});
        }
        // ... and so does a butterfly's target on lane bit 4 / 5 that is about to be exchanged with a register bit
        // anyway: the phases waiting for it then multiply the 2^(RB-1) slots of that register bit with every lane
        // at work (and can join a factor tree) instead of all 2^RB slots with half the lanes idle
        const int bv = role[gi] == 1 ? butterfly_variant(r->g) : -1;
        if (lswap_early_ && bv >= 0 && ch.lswap > 0 && lane_index(geom, r->tgt) >= 4) {
          const int l0 = lane_index(geom, r->tgt);
          ch.lswap--;
          const int vr = victim_reg();
          lswap(l0, vr);
          swaps.push_back(Swap{0, l0, vr});
        }
        // A phase waiting on the butterfly's own target bit alone -- diag(1, phi) with phi = c (1 +- i), c real: T, T^+
        // and their odd powers -- does not become a DIAG group (an op and a group dispatch, a complex product on half the
        // slots): the butterfly takes it along (OPF_ROT_*: an addition and an fma per slot, c folded into its own fma
        // constants).  Register butterflies of complex128 tiles only; diagonal terms commute, so taking this one out of
        // the pending set changes nothing else.
        uint32_t rot_flags = 0;
        double rot_c = 0.0;
        {
          const int lnow = lane_index(geom, r->tgt);
          const bool to_reg = lnow < 0 || (lnow >= 4 && ch.lswap > 0);
          if (rot_fuse_ && bv >= 0 && to_reg && amp_bytes_ == 16 && !r->ghost)
            for (size_t k = 0; k < pending.size(); ++k) {
              const PTerm &t = pending[k];
              if (t.mask != (1ull << r->tgt)) continue;
              const double ar = std::fabs(t.re), ai = std::fabs(t.im);
              if (ar > 0.0 && std::fabs(ar - ai) <= 4e-16 * ar) {
                rot_c = 0.5 * (ar + ai) * (t.re < 0 ? -1.0 : 1.0);
                rot_flags = ((t.re < 0) == (t.im < 0)) ? OPF_ROT_P : OPF_ROT_M;
                pending.erase(pending.begin() + (long)k);
              }
              break;
            }
        }
        const size_t n_ops_before = sp->ops.size();
        flush_diag(&pending, 1ull << r->tgt, sp, geom);
        // non-zero only when THIS flush emitted a DIAG op right in front of the dense op
        size_t n_ops_after_flush = sp->ops.size() > n_ops_before ? sp->ops.size() : 0;
        SweepOp op{};
        uint32_t lane, reg; uint64_t outside, lane_phys;
        uint32_t nlane = 0, nreg = 0; uint64_t noutside = 0, nlane_phys = 0;   // zero-controls
        split_mask(geom, r->ctl_mask, &lane, &lane_phys, &reg, &outside);
        if (r->neg_mask) split_mask(geom, r->neg_mask, &nlane, &nlane_phys, &nreg, &noutside, false);
        // thread controls: (index & cm_thread) == cm_thread & ~zero-controls; the zero-control
        // mask of a dense op travels in the two header words a DIAG op uses for its groups
        const uint64_t nthread = noutside | nlane_phys;
        op.cm_thread = outside | lane_phys | nthread;   // tested against the thread's physical index
        op.n_groups = (uint32_t)nthread;
        op.group_off = (uint32_t)(nthread >> 32);
        op.cm_reg = reg | (nreg << 8);                  // bits 0..4 must be one, bits 8..12 must be zero
        memcpy(op.g, r->g, sizeof op.g);
        if (role[gi] == 2) for (int k = 0; k < 4; ++k) cmul_acc(&op.g[2 * k], &op.g[2 * k + 1], sink_re[gi], sink_im[gi]);
        int li = lane_index(geom, r->tgt);
        if (bv >= 0 && li >= 4 && ch.lswap > 0) {   // lane bit 4/5 <-> a register bit, then a register butterfly
          ch.lswap--;
          const int vr = victim_reg();
          lswap(li, vr);
          swaps.push_back(Swap{0, li, vr});
          n_ops_after_flush = 0;
          li = -1;
        }
        if (li >= 0) { op.kind = OP_DENSE_LANE; op.tb = (uint32_t)li; }
        else { op.kind = OP_DENSE_REG; op.tb = reg_index(geom, r->tgt); }
        if (r->ghost) {
          // the layout exchanges above happened as on the ranks where the gate is live, and the cost model
          // (choose_lane_paths) sees an op of the same kind; emit_ops() removes it
          op.flags = OPF_GHOST;
          memset(op.g, 0, sizeof op.g);
          sp->ops.push_back(op);
          continue;
        }
        if (op.g[1] == 0.0 && op.g[3] == 0.0 && op.g[5] == 0.0 && op.g[7] == 0.0) op.flags |= OPF_REAL;
        if (bv >= 0) {
          op.flags = OPF_BFLY | ((uint32_t)bv << OPF_BFLY_SHIFT);
          memset(op.g, 0, sizeof op.g);
          if (rot_flags) {
            assert(op.kind == OP_DENSE_REG);     // (to_reg above: the target sits in a register by now)
            op.flags |= rot_flags;
            op.g[0] = rot_c;
            sp->ops.push_back(op);
            continue;
          }
          if (bv == 1) { op.g[0] = -1.0; op.g[1] = 1.0; }        // LDS lane form: new = own + beta*partner,
          else if (bv == 2) { op.g[0] = 1.0; op.g[1] = -1.0; }   // beta on the 0-lane / on the 1-lane
          int *budget = (li < 0 || li >= 4) ? nullptr : one_step_dpp((uint32_t)li) ? &ch.dpp01 : &ch.dpp23;
          if (budget && *budget > 0) {
            // DPP path: new.re = own.re + b_re*q.re, new.im = own.im + b_im*q.im with q the partner
            // (re/im exchanged for v, v^+); g = b_re(0-lane), b_re(1-lane), b_im(0-lane), b_im(1-lane).
            // h = [[1,1],[1,-1]] = Z * [[1,1],[-1,1]]: run variant 2, the Z joins the diagonal terms.
            --*budget;
            op.flags |= OPF_LANE_DPP;
            static const double kBeta[5][4] = {{1, -1, 1, -1}, {-1, 1, -1, 1}, {1, -1, 1, -1}, {1, 1, -1, -1}, {-1, -1, 1, 1}};
            memcpy(op.g, kBeta[bv], sizeof kBeta[bv]);
            if (bv >= 3) op.flags |= OPF_SWAP_RI;
            sp->ops.push_back(op);
            if (bv == 0) add_pending(1ull << r->tgt, -1.0, 0.0);
            continue;
          }
        }
        if (bv < 0 && (op.flags & OPF_REAL) && op.kind == OP_DENSE_LANE && op.tb < 4) {
          int *budget = one_step_dpp(op.tb) ? &ch.real01 : &ch.real23;   // real lane op: partner by DPP instead of LDS
          if (*budget > 0) {
            --*budget;
            op.flags |= OPF_LANE_DPP;
            sp->ops.push_back(op);
            continue;
          }
        }
        // (for a REAL gate folding would turn its 4-op real path into the 9-op complex
        // one: no gain over applying c to the slots, so only complex gates fold)
        if (!(op.flags & (OPF_REAL | OPF_BFLY)) && op.kind == OP_DENSE_LANE && op.cm_thread == 0 && op.cm_reg == 0 && !sp->ops.empty() &&
            sp->ops.back().kind == OP_DIAG && sp->ops.size() == n_ops_after_flush) {
          sp->ops.back().flags |= OPF_DEFER_C;
          op.flags |= OPF_USE_C;
        }
        sp->ops.push_back(op);
        continue;
      }
      // diagonal: amp *= d0 under controls (if d0 != 1), then amp *= d1/d0 where tgt set.
      // Zero-controls by inclusion-exclusion: [all of N are 0] = sum over S in N of (-1)^|S| [all of S are 1],
      // i.e. factor f on (mask, none of N) = product over S of f^((-1)^|S|) on mask | S.
      const uint64_t neg = r->neg_mask;
      auto add_signed = [&](uint64_t mask, double fr, double fi) {
        const double den = fr * fr + fi * fi;
        for (uint64_t sub = neg;; sub = (sub - 1) & neg) {
          if (popc(sub) & 1) add_pending(mask | sub, fr / den, -fi / den);
          else add_pending(mask | sub, fr, fi);
          if (!sub) break;
        }
      };
      const uint64_t bits = r->ctl_mask | (r->tgt >= 0 ? (1ull << r->tgt) : 0);
      const double d0r = r->g[0], d0i = r->g[1], d1r = r->g[6], d1i = r->g[7];
      if (r->tgt < 0) {
        add_signed(r->ctl_mask, d0r, d0i);
      } else if (is_one(d0r, d0i)) {
        add_signed(bits, d1r, d1i);
      } else {
        add_signed(r->ctl_mask, d0r, d0i);
        const double den = d0r * d0r + d0i * d0i;  // != 0: see plan_diag()
        add_signed(bits, (d1r * d0r + d1i * d0i) / den, (d1i * d0r - d1r * d0i) / den);
      }
    }
    flush_diag(&pending, ~0ull, sp, geom);
    if (fold_pending_) {      // no DIAG op of the sweep had a factor for every amplitude: one of its own
      SweepOp op{};
      op.kind = OP_DIAG;
      op.group_off = (uint32_t)sp->groups.size();
      op.n_groups = 1;
      DGroup g{};
      g.re = fold_re_; g.im = fold_im_;
      sp->groups.push_back(g);
      sp->ops.push_back(op);
      fold_pending_ = false;
    }
    // Lane exchanges are undone (the lane -> address map of the store is fixed); wave
    // exchanges are not: the tile is stored where its amplitudes now belong (own slot
    // offsets, base corrected by the moved index bits) -- one LDS exchange less per wave bit.
    // (only for tiles with contiguous lanes: with split lanes the exchanged layout is the
    // slower store geometry -- 7.1 vs 6.75 ms on sweep 2 of the QFT -- and undoing wins)
    // A RELAYOUT sweep stores the tile contiguously into the second buffer whatever its bits are
    // now: neither kind of exchange is undone (see relayout()).
    if (want_relayout(*sp, !swaps.empty())) {
      sp->relayout = true;      // (dest_pos: plan() -> finish_relayout, once the next sweep's targets are known)
      // Tiles chosen for the whole flush (the tile search) count on the line bits being part of every
      // tile: a line bit that an exchange has left in a register (or, through a register, in the wave id) comes back to
      // the lanes before the store, so that positions 0..lane_low-1 hold the same qubits in the next sweep.  Without such
      // tiles the next sweep simply selects its tile around whatever sits there.
      if (keep_line_bits_)
        for (int b = 0; b < geom.lane_low; ++b) {
          if (lane_index(geom, b) >= 0) continue;
          const int wi = wave_index(geom, b);
          if (wi >= 0) wswap(wi, victim_reg());
          const int r = reg_index(geom, b);
          int li = -1;
          for (int sidx = kLaneBits - 1; sidx >= 4; --sidx) if (geom.seat[sidx] >= geom.lane_low) li = sidx;
          if (r >= 0 && li >= 0) lswap(li, r);
        }
    } else if (store_swapped_ && sp->contiguous()) {
      if (!undo_lane_swaps()) restore_layout();
    } else {
      restore_layout();
    }
    memcpy(sp->regpos_store, geom.regpos, sizeof geom.regpos);
    memcpy(sp->seat_store, geom.seat, sizeof geom.seat);
    memcpy(sp->wavepos_store, geom.wavepos, sizeof geom.wavepos);
    // lane tables go to the front of `tables` (one contiguous block: the kernel copies it to LDS)
    const uint32_t nlt = (uint32_t)(sp->ltabs.size() / 2);
    if (nlt) {
      for (auto &g : sp->groups) {
        for (uint32_t t = 0; t < g.ntab; ++t) g.tab_off[t] += nlt;
        if (g.flags & DG_BITFAC) g.tab_off[3] += nlt;
      }
      sp->tables.insert(sp->tables.begin(), sp->ltabs.begin(), sp->ltabs.end());
      sp->ltabs.clear();
    }
    sp->n_ltab = (int)(nlt / 64);
  }

  // A sweep whose tile does not sit on the low index bits reads eight 128-byte lines per load
  // instruction (split lanes) and rewrites the same scattered lines: 6.2-7.1 ms per 2 x 16 GiB
  // against 5.2 ms for a contiguous tile (profiles/r02: the lines of a wave are 64 KiB - 32 MiB
  // apart; with tile bits >= 21 a third of the L1 translation requests miss).  Measured with the
  // same tiles (tools/membench/oopsweep): scattered load + CONTIGUOUS store into a second buffer
  // runs 15-20% faster than the in-place sweep, so a sweep that touches every amplitude anyway
  // may as well leave its tile bits on the low positions -- where they stay for the next flush.
  bool want_relayout(const SweepPlan &sp, bool has_swaps = false) const {
    if (!relayout_ || sp.fixed_ones) return false;      // (a sweep that skips amplitudes cannot move the rest)
    const int low = sp.lane_low;
    for (int k = 0; k < sp.nlanehi(); ++k) if (sp.lanehi[k] != low + k) return true;
    for (int k = 0; k < sp.rb; ++k) if (sp.regpos[k] != kLaneBits + k) return true;
    for (int k = 0; k < sp.nwave; ++k) if (sp.wavepos[k] != kLaneBits + sp.rb + k) return true;
    return false;                                       // already contiguous: in place
  }

  // The store map of a relayout sweep: its own tile bits (as they sit at store time) on the low
  // positions; directly above them the bits the NEXT sweep will want as dense targets (`ahead`,
  // N3 "qubit remapping to keep hot targets in low bits": a gather from lines 64 KiB - 16 MiB apart
  // runs at 5.8 ms per 2 x 16 GiB, from lines >= 32 MiB apart at 6.5-6.9); the rest above, order kept.
  void finish_relayout(SweepPlan *sp, const std::vector<int> &ahead) const {
    // the six bits on the lanes at store time take positions 0..5: a line bit keeps its own position, the others fill
    // the free ones in ascending order of the index bits they hold -- whatever their seats were (the layout a sweep
    // leaves, hence the next sweep's gather pattern, does not depend on how this one seated its lanes)
    uint64_t placed = 0, used = 0;
    int order[kLaneBits];
    for (int i = 0; i < kLaneBits; ++i) order[i] = sp->seat_store[i];
    std::sort(order, order + kLaneBits);
    // (tried: the bits the NEXT sweep's tile wants first among them, so that its lane bits sit lowest -- five supremacy-30
    //  instances +0.7 / -0.6 / +1.0 / -2.2 / +0.6 %: nothing, profiles/r05/seats_ab3_summary.txt)
    for (int i = 0; i < kLaneBits; ++i) {
      const int b = order[i];
      if (b < sp->lane_low) { sp->dest_pos[b] = (uint8_t)b; placed |= 1ull << b; used |= 1ull << b; }
    }
    for (int i = 0; i < kLaneBits; ++i) {
      const int b = order[i];
      if ((placed >> b) & 1ull) continue;
      const int d = __builtin_ctzll(~used);
      sp->dest_pos[b] = (uint8_t)d;
      used |= 1ull << d;
      placed |= 1ull << b;
    }
    for (int i = 0; i < kLaneBits; ++i) sp->seat_dest[i] = sp->dest_pos[sp->seat_store[i]];
    int next = kLaneBits;
    auto put = [&](int p) {
      if (p < 0 || p >= nloc_ || ((placed >> p) & 1ull)) return;
      sp->dest_pos[p] = (uint8_t)next++;
      placed |= 1ull << p;
    };
    // register and wave bits share the positions above the lanes, sorted by the index bits they hold:
    // the block keeps its bits in ascending order (but for the lanes), so the phase tables of later
    // sweeps keep finding their eight-bit windows filled
    std::vector<int> rw;
    for (int k = 0; k < sp->rb; ++k) rw.push_back(sp->regpos_store[k]);
    for (int k = 0; k < sp->nwave; ++k) rw.push_back(sp->wavepos_store[k]);
    std::sort(rw.begin(), rw.end());
    for (int p : rw) put(p);
    for (int k = 0; k < sp->rb; ++k) sp->reg_dest[k] = sp->dest_pos[sp->regpos_store[k]];
    for (int k = 0; k < sp->nwave; ++k) sp->wave_dest[k] = sp->dest_pos[sp->wavepos_store[k]];
    if (lookahead_) for (int p : ahead) put(p);
    for (int p = 0; p < nloc_; ++p) put(p);
    for (int p = nloc_; p < 64; ++p) sp->dest_pos[p] = (uint8_t)p;
  }

  static void cmul_acc(double *re, double *im, double fr, double fi) {
    const double nr = *re * fr - *im * fi, ni = *re * fi + *im * fr;
    *re = nr; *im = ni;
  }

  // Emit one DIAG op with every pending term touching `bits` (all terms when
  // bits == ~0, including bit-less global factors).
  void flush_diag(std::vector<PTerm> *pending, uint64_t bits, SweepPlan *sp, const SweepPlan &geom) {
    std::vector<PTerm> sel, keep;
    for (auto &t : *pending) ((bits == ~0ull || (t.mask & bits)) ? sel : keep).push_back(t);
    pending->swap(keep);
    if (sel.empty()) return;
    struct PGroup {
      uint32_t lane, reg; double re = 1, im = 0;
      std::vector<OTerm> single;  // single outside bit
      std::vector<OTerm> multi;   // several outside bits
    };
    std::vector<PGroup> groups;
    for (auto &t : sel) {
      uint32_t lane, reg; uint64_t outside, lane_phys;
      split_mask(geom, t.mask, &lane, &lane_phys, &reg, &outside);
      PGroup *g = nullptr;
      for (auto &pg : groups) if (pg.lane == lane && pg.reg == reg) { g = &pg; break; }
      if (!g) { groups.push_back(PGroup{lane, reg}); g = &groups.back(); }
      if (outside == 0) { cmul_acc(&g->re, &g->im, t.re, t.im); continue; }
      auto &vec = (popc(outside) == 1) ? g->single : g->multi;
      bool found = false;
      for (auto &o : vec) if (o.mask == outside) { cmul_acc(&o.re, &o.im, t.re, t.im); found = true; break; }
      if (!found) vec.push_back(OTerm{outside, t.re, t.im});
    }
    SweepOp op{};
    op.kind = OP_DIAG;
    op.group_off = (uint32_t)sp->groups.size();
    // single-bit outside factors of a group become 256-entry chunk tables (one scalar
    // load per 8 index bits at run time); the rest stay loop terms
    auto attach_outside = [&](DGroup &g, PGroup &pg) {
      std::vector<OTerm> loop_terms = pg.multi;
      // (a line bit that an exchange has carried into the wave id is an outside bit below the table windows: a loop term)
      for (size_t k = pg.single.size(); k-- > 0;)
        if (pg.single[k].mask < (1ull << sp->lane_low)) { loop_terms.push_back(pg.single[k]); pg.single.erase(pg.single.begin() + (long)k); }
      for (int shift = sp->lane_low; shift < 64 && !pg.single.empty(); shift += 8) {
        const uint64_t cmask = (shift + 8 >= 64) ? (~0ull << shift) : (((1ull << 8) - 1) << shift);
        std::vector<OTerm> in;
        for (auto &o : pg.single) if (o.mask & cmask) in.push_back(o);
        if (in.empty()) continue;
        if (g.ntab == 4 || (int)in.size() < min_table_terms_) {  // no table slot left / not worth a table
          for (auto &o : in) loop_terms.push_back(o);
          continue;
        }
        g.tab_shift |= (uint32_t)shift << (8 * g.ntab);
        g.tab_off[g.ntab] = (uint32_t)(sp->tables.size() / 2);
        g.ntab++;
        // entry v = product of the factors of v's set bits, built by doubling: entry(v) = entry(v without its
        // lowest set bit) x factor(that bit) -- the same products in the same order as a loop over the terms
        // (terms of one bit were merged when they were collected)
        double bf[8][2];
        for (int b = 0; b < 8; ++b) { bf[b][0] = 1; bf[b][1] = 0; }
        for (auto &o : in) {
          const int b = __builtin_ctzll(o.mask) - shift;
          cmul_acc(&bf[b][0], &bf[b][1], o.re, o.im);
        }
        const size_t t0 = sp->tables.size();
        sp->tables.resize(t0 + 512);
        double *tb = &sp->tables[t0];
        tb[0] = 1; tb[1] = 0;
        for (uint32_t v = 1; v < 256; ++v) {
          const int hb = 31 - __builtin_clz(v);            // highest set bit: entry(v) = entry(v - 2^hb) x factor(hb)
          double fr = tb[2 * (v ^ (1u << hb))], fi = tb[2 * (v ^ (1u << hb)) + 1];
          cmul_acc(&fr, &fi, bf[hb][0], bf[hb][1]);
          tb[2 * v] = fr; tb[2 * v + 1] = fi;
        }
      }
      g.oterm_off = (uint32_t)sp->oterms.size();
      g.n_oterms = (uint32_t)loop_terms.size();
      for (auto &o : loop_terms) sp->oterms.push_back(o);
    };
    // (1) groups with equal reg_mask whose factor depends on the lane only merge into ONE
    // lane table; the wave-uniform group of the same reg_mask (lane_mask 0: constant and
    // outside-bit factors) rides along, so the slots are multiplied once: f = ltab[lane]*u
    std::vector<bool> done(groups.size(), false);
    for (size_t i = 0; i < groups.size(); ++i) {
      if (done[i]) continue;
      std::vector<size_t> lane_only;
      int uniform = -1;
      for (size_t j = i; j < groups.size(); ++j) {
        if (done[j] || groups[j].reg != groups[i].reg) continue;
        const bool has_out = !groups[j].single.empty() || !groups[j].multi.empty();
        if (!has_out && groups[j].lane != 0) lane_only.push_back(j);
        else if (groups[j].lane == 0) uniform = (int)j;
      }
      if (lane_only.empty() || lane_only.size() + (uniform >= 0 ? 1 : 0) < 2) continue;
      DGroup g{};
      g.reg_mask = groups[i].reg;
      g.re = 1; g.im = 0;
      g.flags = DG_LTAB;
      g.ltab_off = (uint32_t)(sp->ltabs.size() / 2);
      for (uint32_t lane = 0; lane < 64; ++lane) {
        double fr = 1, fi = 0;
        for (size_t j : lane_only)
          if ((lane & groups[j].lane) == groups[j].lane) cmul_acc(&fr, &fi, groups[j].re, groups[j].im);
        sp->ltabs.push_back(fr);
        sp->ltabs.push_back(fi);
      }
      for (size_t j : lane_only) done[j] = true;
      if (uniform >= 0) {
        g.re = groups[uniform].re; g.im = groups[uniform].im;
        attach_outside(g, groups[uniform]);
        done[uniform] = true;
      }
      sp->groups.push_back(g);
    }
    // (2) the rest: one DGroup each
    for (size_t i = 0; i < groups.size(); ++i) {
      if (done[i]) continue;
      PGroup &pg = groups[i];
      DGroup g{};
      g.lane_mask = pg.lane; g.reg_mask = pg.reg; g.re = pg.re; g.im = pg.im;
      attach_outside(g, pg);
      sp->groups.push_back(g);
    }
    if (fold_pending_) {
      // the butterflies' scalars (emit_ops_with): into a factor that reaches every lane and slot, if this op has one
      DGroup *host = nullptr;
      bool c_part = false;
      for (size_t k = op.group_off; k < sp->groups.size(); ++k) {
        DGroup &g = sp->groups[k];
        if (g.reg_mask) continue;
        c_part = true;
        if (!host && ((g.flags & DG_LTAB) || g.lane_mask == 0)) host = &g;
      }
      if (host) {
        cmul_acc(&host->re, &host->im, fold_re_, fold_im_);
        fold_pending_ = false;
      } else if (c_part) {
        DGroup g{};
        g.re = fold_re_; g.im = fold_im_;
        sp->groups.push_back(g);
        fold_pending_ = false;
      }
    }
    fuse_bit_factors(sp, op.group_off, geom.rb);
    op.n_groups = (uint32_t)(sp->groups.size() - op.group_off);
    sp->ops.push_back(op);
  }

  // Scalar groups on two register bits {i, j} join the group on {j} as bit factors (DG_BITFAC).
  void fuse_bit_factors(SweepPlan *sp, uint32_t first, int rb) const {
    if (!bitfac_) return;
    auto scalar2 = [](const DGroup &g) {
      return g.lane_mask == 0 && g.flags == 0 && g.ntab == 0 && g.n_oterms == 0 && popc(g.reg_mask) == 2;
    };
    int best_j = -1, best_n = 0;
    for (int j = 0; j < rb; ++j) {
      int others[kBitFacs];
      const int no = bitfac_others(rb, j, others);
      int n = 0;
      for (size_t k = first; k < sp->groups.size(); ++k) {
        const DGroup &g = sp->groups[k];
        if (!scalar2(g) || !(g.reg_mask >> j & 1)) continue;
        const int i = __builtin_ctz(g.reg_mask & ~(1u << j));
        for (int t = 0; t < no; ++t) n += others[t] == i;
      }
      if (n > best_n) { best_n = n; best_j = j; }
    }
    if (best_j < 0) return;
    int base = -1;
    for (size_t k = first; k < sp->groups.size(); ++k)
      if (sp->groups[k].reg_mask == (1u << best_j) && sp->groups[k].ntab <= 3) { base = (int)k; break; }
    if (best_n < (base >= 0 ? 2 : 3)) return;
    int others[kBitFacs];
    const int no = bitfac_others(rb, best_j, others);
    double w[2 * kBitFacs] = {1, 0, 1, 0, 1, 0, 1, 0};
    std::vector<DGroup> kept;
    const int base_abs = base;
    for (size_t k = first; k < sp->groups.size(); ++k) {
      const DGroup &g = sp->groups[k];
      int slot = -1;
      if (scalar2(g) && (g.reg_mask >> best_j & 1)) {
        const int i = __builtin_ctz(g.reg_mask & ~(1u << best_j));
        for (int t = 0; t < no; ++t) if (others[t] == i) slot = t;
      }
      if (slot < 0) { kept.push_back(g); continue; }
      cmul_acc(&w[2 * slot], &w[2 * slot + 1], g.re, g.im);
      if ((int)k < base_abs) base--;           // (k == base_abs cannot happen: the base has a 1-bit mask)
    }
    if (base < 0) {
      DGroup g{};
      g.reg_mask = 1u << best_j;
      g.re = 1; g.im = 0;
      kept.push_back(g);
      base = (int)kept.size() - 1;
    } else {
      base -= (int)first;
    }
    kept[base].flags |= DG_BITFAC;
    kept[base].tab_off[3] = (uint32_t)(sp->tables.size() / 2);
    for (double x : w) sp->tables.push_back(x);
    sp->groups.resize(first);
    for (const DGroup &g : kept) sp->groups.push_back(g);
  }
};

// The plan with the fewest sweeps over the wave-bit choices {1, 2, 0}, the earlier choice
// winning ties.  Measured (ms): 30-qubit QFT 21.3 with one wave bit / 22.2 with two (same 3
// sweeps: the four-wave barrier waits for the slowest of four op streams) / 24.6 without
// (4 sweeps); supremacy-30 52 (6 sweeps) / 51 (5) / 54 (7); QFT-31 52 (4) / 44 (3); QFT-32
// 104 (4) / 86 (3).  A plan is set aside if one of its tiles spreads over eight or more index
// bits >= 25 -- every 128-byte line of a wave then sits in another 512-MiB region and the
// sweep runs at half speed (QFT-33, two wave bits: 113 ms for the last sweep instead of 55).
// Planning costs a few ms per attempt and runs while the previous flush is still executing.
inline bool plan_has_far_tile(const PlanResult &pr) {
  for (const SweepPlan &sp : pr.sweeps) {
    int far = 0;
    for (int k = 0; k < sp.nlanehi(); ++k) far += sp.lanehi[k] >= 25;
    for (int k = 0; k < sp.rb; ++k) far += sp.regpos[k] >= 25;
    if (far >= 8) return true;
  }
  return false;
}

// ---- what a planned sweep costs (round 6: the objective of plan_best) --------------------------------------------------------
// The op streams of these sweeps run against the socket's 1 400 W: time is ENERGY.  Socket energy of one
// tile's op stream in nJ, every op and DIAG group priced by the wave-instructions tools/gen_sweep_asm.py emits for it
// (complex128 counts; the same classes and figures as tools/plan_valu_cost.py, which SQ_INSTS_VALU confirms to 2 %) times the
// measured price per instruction kind (profiles/r04/valu_power_per_instruction.txt: v_fma_f64 2.0, v_mul_f64 1.9, v_add_f64
// 1.5, DPP move 0.75, v_permlane swap 1.2, v_xor_b32 1.0 nJ).
inline double sweep_op_energy_nj(const SweepPlan &sp) {
  const double nr = (double)(1u << sp.rb);
  double e = 0;
  for (const SweepOp &op : sp.ops) {
    const uint32_t f = op.flags;
    if (f & OPF_GHOST) continue;
    switch (op.kind) {
      case OP_DENSE_REG:
        if ((f & OPF_BFLY) && (f & (OPF_ROT_P | OPF_ROT_M))) e += 3 * nr * 1.58;
        else if (f & OPF_BFLY) e += 2 * nr * 1.5;
        else if (f & OPF_REAL) e += 5 * nr * 1.9;
        else e += (10 * nr + 10) * 1.95;
        break;
      case OP_DENSE_LANE:
        if ((f & OPF_LANE_DPP) && (f & OPF_BFLY)) e += (nr * (op.tb == 2 ? 10 : 6) + 14) * 1.0;
        else if (f & OPF_LANE_DPP) e += (nr * (op.tb == 2 ? 12 : 8) + 20) * 1.1;
        else if (f & OPF_BFLY) e += (2 * nr + 8) * 1.5 + 4 * nr * 1.0;         // + the LDS shuffles (ds_bpermute: ~1 nJ each, by time)
        else e += (((f & OPF_REAL) ? 4 : 10) * nr + 30) * 1.9 + 4 * nr * 1.0;
        break;
      case OP_LSWAP: e += (2 * nr + 8) * 1.2; break;
      case OP_WSWAP: e += 5 * 1.0 + 2 * nr * 1.0; break;                       // 16 + 16 ds_*_b128 per wave and the barrier's wait
      case OP_DIAG: {
        bool c_touched = false, c_sign = false;
        for (uint32_t gi = 0; gi < op.n_groups; ++gi) {
          const DGroup &g = sp.groups[op.group_off + gi];
          const double slots = nr / (double)(1u << popc((uint64_t)g.reg_mask));
          const bool general = (g.flags & DG_LTAB) || g.ntab || g.n_oterms;
          const bool sign = g.re == -1.0 && g.im == 0.0 && !(g.flags & DG_LTAB) && !g.ntab;
          if (g.flags & DG_BITFAC) { e += (124 + ((g.lane_mask || general) ? 11 : 0)) * 1.95; continue; }
          if (sign) {
            if (!g.reg_mask) { e += 6 * 1.0; c_touched = c_sign = true; }
            else e += (4 + 2 * slots) * 1.0;
            continue;
          }
          double pro = 0;
          if (general) pro = 4 + 4 * g.ntab + 4 * g.n_oterms + ((g.flags & DG_LTAB) ? 4 : 7);
          else if (g.lane_mask) pro = 11;
          if (!g.reg_mask) { e += (pro + 4) * 1.9; c_touched = true; c_sign = false; }
          else e += 4 * slots * 1.95 + pro * 1.0;
        }
        if (c_touched && !(f & OPF_DEFER_C)) e += (c_sign ? (2 * nr + 1) : 4 * nr) * 1.95;
        break;
      }
      default: break;
    }
  }
  return e;
}

// Predicted time of a plan in ms.  A sweep is a stream of 2 x (state bytes) through HBM beside its op stream; round 6 fitted
// the per-sweep times of 159 sweeps of 36 supremacy-30 plans (2^30 amplitudes, complex128; profiles/r06/level_search.txt) against
// the op energy of the price list above: flat at the stream's own time up to ~4 000 instructions (~3 J of ops) per tile, then
// +0.74 ms per 1 000 instructions = ~1.0 ms per Joule of ops -- steeper than energy / 1 400 W (0.71 ms per J): a heavy sweep
// is bound by VALU issue (a wave-instruction on FP64 keeps its SIMD 4 cycles, 512 tiles per SIMD: 8 000 instructions per tile
// = 8.2 ms at 2 GHz) before it is bound by the socket's power -- so the cost of a plan is CONVEX in how its ops are spread over
// its sweeps (a light sweep cannot go below the stream, a heavy one pays the steeper rate).  The stream's own time: 5.5 ms for
// tiles of up to one wave bit (the QFT's sweeps: 5.45-5.6), 6.0 ms for four-wave workgroups.  Both scale with the bytes swept
// (fixed bits fold the tile count; complex64 moves half).  Residual of the fit: 0.6 ms per sweep (placement, gather geometry).
inline double plan_predicted_ms(const PlanResult &pr, int nloc, int bw) {
  double ms = 0;
  for (const SweepPlan &sp : pr.sweeps) {
    const double scale = (double)sp.swept_bytes / (2.0 * 16.0 * (double)(1ull << 30));
    const double e_ops_j = sweep_op_energy_nj(sp) * (double)sp.ntiles * 1e-9;
    const double floor_ms = (sp.nwave >= 2 ? 6.0 : 5.5) * scale;
    // (four-wave workgroups: two LDS exchanges and their barriers -- candidates of one circuit forced one by one,
    // profiles/r06/level_search.txt: tilings with two wave bits run 0.2-0.4 ms per sweep above those with one at equal op energy)
    ms += std::max(floor_ms, 3.0 * scale + 0.99 * e_ops_j) + (sp.nwave >= 2 ? 0.3 * scale : 0.0);
  }
  (void)nloc; (void)bw;
  return ms;
}

// Host threads this process may really run side by side: the hardware's count, cut down by the scheduler affinity and by a
// cgroup CPU quota (a container limited to a few cores still reports every core of its host).
inline unsigned usable_cpus() {
  unsigned n = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (!n || (unsigned)c < n)) n = (unsigned)c; }
  for (const char *path : {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"}) {
    FILE *f = fopen(path, "r");
    if (!f) continue;
    long long quota = -1, period = 100000;
    char word[32] = {0};
    if (fscanf(f, "%31s %lld", word, &period) >= 1 && strcmp(word, "max") != 0) quota = atoll(word);
    fclose(f);
    if (strstr(path, "cfs_quota")) {      // cgroup v1: the period sits in its own file
      period = 100000;
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 100000; fclose(g); }
    }
    if (quota > 0 && period > 0) { const unsigned q = (unsigned)std::max<long long>(1, quota / period); if (!n || q < n) n = q; }
    break;
  }
  return n;
}

inline PlanResult plan_best(const std::vector<GateRec> &queue, int nloc, uint64_t shard, int bw, int max_rb,
                            bool split_lanes, bool allow_relayout = false, bool keep_ghosts = false) {
  if (getenv("QH_WAVE_BITS")) return Planner(nloc, shard, bw, max_rb, split_lanes, -1, allow_relayout, keep_ghosts).plan(queue);
  if (const char *e = getenv("QH_PLAN_TILES")) {      // (probe: "wb:b,b,..;b,b,..": tiles in the bit numbering of the flush's start)
    std::vector<std::vector<int>> tiles(1);
    const int wb = atoi(e);
    const char *p = strchr(e, ':');
    for (p = p ? p + 1 : e; *p; ) {
      if (*p == ';') { tiles.emplace_back(); ++p; continue; }
      if (*p == ',') { ++p; continue; }
      char *end = nullptr;
      tiles.back().push_back((int)strtol(p, &end, 10));
      if (end == p) break;
      p = end;
    }
    Planner forced(nloc, shard, bw, max_rb, split_lanes, wb, allow_relayout, keep_ghosts);
    forced.set_tiles(tiles);
    return forced.plan(queue);
  }
  // the number of wave bits from the tile selections alone (Planner::skeleton), then ONE full plan: a third of the
  // planning time of three full plans (30-qubit QFT: 1.9 -> 0.9 ms)
  int best_wb = 1;
  size_t best_n = 0, n_of[kMaxWaveBits + 1] = {0, 0, 0};
  bool have = false, best_far = false, far_of[kMaxWaveBits + 1] = {false, false, false};
  const int only_wb = env_int("QH_PLAN_ONLY_WB", -1);      // (probe: everything below with a pinned number of wave bits)
  for (int wb : {1, 2, 0}) {
    if (only_wb >= 0 && wb != only_wb) continue;
    size_t n = 0;
    bool far = false;
    Planner(nloc, shard, bw, max_rb, split_lanes, wb, allow_relayout, keep_ghosts).skeleton(queue, &n, &far);
    n_of[wb] = n;
    far_of[wb] = far;
    if (!have || (best_far && !far) || (best_far == far && n < best_n)) {
      best_wb = wb;
      best_n = n;
      best_far = far;
      have = true;
    }
    if (best_n <= 1 && !best_far) break;
  }
  Planner chosen(nloc, shard, bw, max_rb, split_lanes, best_wb, allow_relayout, keep_ghosts);
  // Fewer sweeps?  Worth a search only where a sweep costs about what the search does: budget = the label changes that
  // fit into ~3 sweep times of one host thread (2 x state bytes at 5.5 TB/s, ~5 ns per change on the GPU box's host: 4 M
  // changes = ~20 ms for a 16-GiB state), hidden behind the GPU whenever circuits are submitted back to back, paid once per
  // circuit with the plan cache on.  QH_PLAN_SEARCH=0 switches it off, QH_PLAN_SEARCH_STEPS pins the budget.
  // Round 6: the search is Planner::search_levels (nested cuts instead of tiles), and it runs as a PORTFOLIO: for the wave-bit
  // count the skeletons chose and for two wave bits (a tile of 13 bits instead of 12), every K from three below the greedy
  // count (never below what the qubit count allows) up to one below it, each on six generator streams -- one host thread
  // per task, so the wall time is ONE budget whatever fails (the search for the K that does not exist always does).  A task
  // that finds tiles builds its plan and prices it (plan_predicted_ms: stream energy + op energy under the socket's power
  // limit); the cheapest plan wins -- tilings of one circuit differ by 5-10 % in what their ops cost --, the greedy plan
  // included (a sharded handle: fewest sweeps, then task order -- see below).  Every task is deterministic by itself, so the
  // outcome does not depend on the threads' timing.  24 supremacy-30 instances (seeds 0-23): greedy 5-8 sweeps, the tile
  // search 4 4 5 5 5 6 5 5 4 5 5 5 ..., now 4 4 4 4 4 5 4 4 4 4 5 4 4 4 4 4 4 5 4 5 4 4 4 5 -- for every one the minimum an integer
  // program finds under 13-bit tiles (tools/tiling_milp.py); GPU time per circuit, old library against new on one box: 32.8 ->
  // 29.8 ms on average, -19 % at best (profiles/r06/level_search.txt).
  uint64_t dense_bits = 0;
  for (const GateRec &q : queue) if (q.tgt >= 0 && q.tgt < nloc && !plan_diag(q.g, q.tgt)) dense_bits |= 1ull << q.tgt;
  const int lane_low = bw == 128 ? 3 : 4;
  const int cap0 = (kLaneBits - lane_low) + std::min({max_rb, max_reg_bits(bw), nloc - kLaneBits});
  if (best_n >= 3 && env_flag("QH_PLAN_SEARCH", true)) {
    const double sweep_us = 2.0 * (double)(bw == 128 ? 16 : 8) * (double)(1ull << nloc) / 5.5e6;
    uint64_t budget = std::min<uint64_t>((uint64_t)(sweep_us * 640.0), 5000000);
    if (const char *e = getenv("QH_PLAN_SEARCH_STEPS")) budget = strtoull(e, nullptr, 10);
    struct Task {
      int wb;
      size_t K;
      uint64_t stream, used = 0;
      bool ok = false, planned = false;
      double ms = 0;
      PlanResult pr;
    };
    std::vector<Task> tasks;
    const size_t nmovable = (size_t)popc(dense_bits >> lane_low);
    std::vector<std::pair<int, size_t>> shapes;       // (wave bits, K)
    for (int wb : {1, 2, 0}) {
      // the wave-bit count the skeletons chose, and one and two wave bits (tiles of 12 and 13 bits) wherever they can save a sweep
      if (budget < 5000) break;
      if (wb != best_wb && (wb == 0 || far_of[wb] || n_of[wb] == 0 || best_n < 4 || (wb == 2 && !env_flag("QH_PLAN_SEARCH_WB2", true)))) continue;
      if (only_wb >= 0 && wb != only_wb) continue;
      const size_t cap = (size_t)(cap0 + wb), greedy = n_of[wb];
      const size_t kmin = std::max<size_t>({2, (nmovable + cap - 1) / cap, greedy > 3 ? greedy - 3 : 0});
      for (size_t K = kmin; K + 1 <= greedy && K <= best_n && K <= 8; ++K) shapes.emplace_back(wb, K);
    }
    // six generator streams per shape (one host thread each) -- fewer on a host with few cores (an unsharded handle only: the
    // ranks of a sharded state must find the same plan whatever their hosts are); QH_PLAN_SEARCH_STREAMS pins it
    int streams = 6;
    if (const unsigned hw = usable_cpus(); hw && !keep_ghosts && !shapes.empty())
      streams = std::max(1, std::min(6, (int)(hw / shapes.size())));
    streams = std::max(1, std::min(8, env_int("QH_PLAN_SEARCH_STREAMS", streams)));
    for (const auto &sh : shapes)
      for (int s = 0; s < streams; ++s) { tasks.emplace_back(); tasks.back().wb = sh.first; tasks.back().K = sh.second; tasks.back().stream = (uint64_t)(s + 1); }
    // (the planners are made HERE: their constructors read the switches, and getenv stays out of the worker threads)
    std::vector<Planner> searchers, planners;
    searchers.reserve(tasks.size());
    planners.reserve(tasks.size());
    for (const Task &t : tasks) {
      searchers.emplace_back(nloc, shard, bw, max_rb, split_lanes, t.wb, allow_relayout, keep_ghosts);
      planners.emplace_back(nloc, shard, bw, max_rb, split_lanes, t.wb, allow_relayout, keep_ghosts);
    }
    auto run = [&](Task &t) {
      const size_t i = (size_t)(&t - &tasks[0]);
      std::vector<std::vector<int>> tiles;
      t.ok = searchers[i].search_levels(queue, t.K, budget, t.stream, &tiles, &t.used);
      if (!t.ok) return;
      Planner &forced = planners[i];
      forced.set_tiles(tiles);
      t.pr = forced.plan(queue);
      t.planned = t.pr.sweeps.size() <= t.K && !plan_has_far_tile(t.pr);     // (the model ignores relabelling and tile positions: check)
      // A sharded handle must pick what every other rank picks, and the price is not rank-invariant (a gate that is a ghost on this
      // rank has left the op list it is computed from): there the fewest sweeps win, ties by the order of the task list.
      if (t.planned) t.ms = keep_ghosts ? 1000.0 * (double)t.pr.sweeps.size() + 1e-3 * (double)(i + 1) : plan_predicted_ms(t.pr, nloc, bw);
    };
    PlanResult greedy_plan;
    if (tasks.empty() || !env_flag("QH_PLAN_SEARCH_THREADS", true)) {
      for (Task &t : tasks) run(t);
      greedy_plan = chosen.plan(queue);
    } else {
      std::vector<std::thread> th;
      th.reserve(tasks.size());
      std::vector<Task *> inline_tasks;       // (a host that refuses another thread: the task runs here -- same answer, later)
      for (Task &t : tasks) {
        try { th.emplace_back(run, std::ref(t)); } catch (const std::system_error &) { inline_tasks.push_back(&t); }
      }
      greedy_plan = chosen.plan(queue);
      for (Task *t : inline_tasks) run(*t);
      for (std::thread &x : th) x.join();
    }
    // every task is deterministic by itself (its generator, its budget in label changes); the winner is the plan with the
    // smallest predicted time, ties by the order of the task list -- whatever the threads' timing was
    const Task *win = nullptr;
    double best_ms = keep_ghosts ? 1000.0 * (double)greedy_plan.sweeps.size() : plan_predicted_ms(greedy_plan, nloc, bw);
    for (const Task &t : tasks)
      if (t.planned && t.pr.sweeps.size() <= greedy_plan.sweeps.size() && t.ms < best_ms) { win = &t; best_ms = t.ms; }
    if (const int pick = env_int("QH_PLAN_SEARCH_PICK", -1); pick >= 0) {      // (probe: the pick-th task of the list, if it has a plan -- tools/probes/r06_candidates.sh)
      if ((size_t)pick < tasks.size() && tasks[pick].planned) win = &tasks[pick];
    }
    if (env_flag("QH_PLAN_SEARCH_LOG", false)) {
      for (const Task &t : tasks)
        fprintf(stderr, "[qh plan search]   task %d: wave bits %d K=%zu stream %llu: %s after %llu label changes%s\n", (int)(&t - &tasks[0]), t.wb, t.K, (unsigned long long)t.stream,
                t.ok ? "found" : "not found", (unsigned long long)t.used,
                t.planned ? (", " + std::to_string(t.pr.sweeps.size()) + " sweeps, predicted " + std::to_string(t.ms) + " ms").c_str() : "");
      fprintf(stderr, "[qh plan search] greedy: %d wave bit(s), %zu sweeps, predicted %.2f ms; chosen: %s (%zu tasks, budget %llu label changes each)\n", best_wb,
              greedy_plan.sweeps.size(), plan_predicted_ms(greedy_plan, nloc, bw),
              win ? (std::to_string(win->pr.sweeps.size()) + " sweeps with " + std::to_string(win->wb) + " wave bit(s), predicted " + std::to_string(win->ms) + " ms").c_str() : "greedy",
              tasks.size(), (unsigned long long)budget);
    }
    if (win) return win->pr;
    return greedy_plan;
  }
  return chosen.plan(queue);
}

inline std::string plan_to_json(const std::vector<GateRec> &queue, int nloc, uint64_t shard, int bw = 128,
                                int max_rb = kMaxRegBits, bool split_lanes = true, bool allow_relayout = false,
                                bool keep_ghosts = false) {
  if (nloc < kLaneBits + 2) return "{\"sweeps\":[],\"note\":\"state too small for sweeps\"}";
  PlanResult pr = plan_best(queue, nloc, shard, bw, max_rb, split_lanes, allow_relayout, keep_ghosts);
  std::string s = "{\"noop_gates\":" + std::to_string(pr.noop_gates);
  if (env_flag("QH_PLAN_DAG", false))         // (tools/tiling_milp.py)
    s += ",\"dag\":" + Planner(nloc, shard, bw, max_rb, split_lanes, 1, allow_relayout, keep_ghosts).dag_json(queue);
  s += ",\"sweeps\":[";
  char buf[384];
  for (size_t i = 0; i < pr.sweeps.size(); ++i) {
    const SweepPlan &sp = pr.sweeps[i];
    int nd = 0, ndiag = 0, nbf = 0;
    int nswap = 0, ndpp = 0;
    for (auto &o : sp.ops) {
      if (o.kind == OP_LSWAP || o.kind == OP_WSWAP) { nswap++; continue; }
      (o.kind == OP_DIAG ? ndiag : nd)++;
      if (o.flags & OPF_BFLY) nbf++;
      if (o.flags & OPF_LANE_DPP) ndpp++;
    }
    std::string rp = "[";
    for (int k = 0; k < sp.rb; ++k) rp += (k ? "," : "") + std::to_string(sp.regpos[k]);
    rp += "],\"lanehi\":[";
    for (int k = 0; k < sp.nlanehi(); ++k) rp += (k ? "," : "") + std::to_string(sp.lanehi[k]);
    rp += "],\"wavepos\":[";
    for (int k = 0; k < sp.nwave; ++k) rp += (k ? "," : "") + std::to_string(sp.wavepos[k]);
    rp += "]";
    rp += sp.relayout ? ",\"relayout\":1" : ",\"relayout\":0";
    snprintf(buf, sizeof buf,
             "%s{\"gates\":%llu,\"dense_ops\":%d,\"butterfly_ops\":%d,\"dpp_ops\":%d,\"lswap_ops\":%d,\"diag_ops\":%d,\"groups\":%zu,\"oterms\":%zu,\"table_entries\":%zu,"
             "\"regpos\":%s,\"fixed_ones\":%llu,\"ntiles\":%llu,\"alg_bytes\":%llu,\"swept_bytes\":%llu}",
             i ? "," : "", (unsigned long long)sp.gates, nd, nbf, ndpp, nswap, ndiag, sp.groups.size(), sp.oterms.size(),
             sp.tables.size() / 2, rp.c_str(), (unsigned long long)sp.fixed_ones, (unsigned long long)sp.ntiles,
             (unsigned long long)sp.alg_bytes, (unsigned long long)sp.swept_bytes);
    s += buf;
    if (env_flag("QH_PLAN_VERBOSE", false)) {   // op list for tools/plan_dump.py (debugging aid)
      s.pop_back();
      s += ",\"ops\":[";
      for (size_t k = 0; k < sp.ops.size(); ++k) {
        const SweepOp &o = sp.ops[k];
        snprintf(buf, sizeof buf, "%s{\"kind\":%u,\"tb\":%u,\"flags\":%u,\"cm_reg\":%u,\"cm_thread\":%llu,\"groups\":[",
                 k ? "," : "", o.kind, o.tb, o.flags, o.cm_reg, (unsigned long long)o.cm_thread);
        s += buf;
        for (uint32_t gi = 0; o.kind == OP_DIAG && gi < o.n_groups; ++gi) {
          const DGroup &g = sp.groups[o.group_off + gi];
          snprintf(buf, sizeof buf, "%s[%u,%u,%u,%u,%u,%.17g,%.17g]", gi ? "," : "", g.lane_mask, g.reg_mask, g.flags, g.ntab, g.n_oterms, g.re, g.im);
          s += buf;
        }
        s += "]}";
      }
      s += "]}";
    }
  }
  s += "]}";
  return s;
}

}  // namespace qh
