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
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
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

constexpr int kLaneBits = 6;   // wavefront = 64 lanes
constexpr int kMaxRegBits = 5; // 32 amplitudes (128 VGPRs of data) per lane
constexpr int kMaxSweepOps = 1024;

enum : uint32_t { OP_DENSE_REG = 0, OP_DENSE_LANE = 1, OP_DIAG = 2 };

// ---- device-visible records (plain data, copied verbatim to HBM) -------------
// A factor that multiplies amplitudes whose index has all bits of `mask` set;
// mask only contains bits OUTSIDE the tile (uniform over a wavefront).
struct OTerm {
  uint64_t mask;
  double re, im;
};
// A diagonal group: f = phi0 * prod(oterms that apply); applied to lanes with
// (lane & lane_mask) == lane_mask and register slots with (k & reg_mask) == reg_mask.
struct DGroup {
  uint32_t lane_mask, reg_mask;
  uint32_t oterm_off, n_oterms;
  double re, im;
};
struct SweepOp {
  uint32_t kind;       // OP_*
  uint32_t tb;         // dense: register-bit index or lane-bit index
  uint32_t cm_reg;     // dense: control mask over register-bit indices
  uint32_t n_groups;   // diag
  uint64_t cm_thread;  // dense: control mask over (shard|outside|lane) index bits
  uint32_t group_off;  // diag
  uint32_t pad;
  double g[8];         // dense: the 2x2
};

struct SweepPlan {
  int rb = 0;                      // register bits used by the kernel instance
  int regpos[kMaxRegBits] = {0};   // ascending physical positions (>= 6)
  uint64_t fixed_ones = 0;         // local bits fixed to 1 in the tile enumeration
  uint64_t ntiles = 0;             // wavefront tiles to process
  std::vector<SweepOp> ops;
  std::vector<DGroup> groups;
  std::vector<OTerm> oterms;
  // accounting
  uint64_t gates = 0;              // reference gate applications executed by this sweep
  uint64_t alg_bytes = 0;          // their minimal-touch bytes
  uint64_t swept_bytes = 0;        // bytes this launch reads + writes
};

struct PlanResult {
  std::vector<SweepPlan> sweeps;
  uint64_t noop_gates = 0;  // dropped: shard-bit control unsatisfied / identity
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

class Planner {
 public:
  Planner(int nloc, uint64_t shard, int bw, int max_rb)
      : nloc_(nloc), shard_(shard), amp_bytes_(bw == 128 ? 16 : 8) {
    rb_cap_ = std::min({max_rb, kMaxRegBits, nloc - kLaneBits});
  }

  PlanResult plan(const std::vector<GateRec> &queue) {
    PlanResult out;
    std::vector<GateRec> pending;
    pending.reserve(queue.size());
    const uint64_t lm = (1ull << nloc_) - 1;
    for (const auto &q : queue) {
      // resolve shard-index bits now: they are constants on this rank
      const uint64_t hi = q.ctl_mask >> nloc_;
      if ((shard_ & hi) != hi) { out.noop_gates++; continue; }
      GateRec r = q;
      r.ctl_mask &= lm;
      if (r.tgt >= nloc_) {
        // diagonal gate on a shard bit: a scalar factor under the local controls
        const bool set = (shard_ >> (r.tgt - nloc_)) & 1ull;
        const double fr = set ? r.g[6] : r.g[0], fi = set ? r.g[7] : r.g[1];
        if (is_one(fr, fi)) { out.noop_gates++; continue; }
        // re-express as a one-sided diagonal gate on one of its control bits, or
        // as a global factor (tgt = -1) when it has no local control
        if (r.ctl_mask) {
          const int c = __builtin_ctzll(r.ctl_mask);
          r.ctl_mask &= ~(1ull << c);
          r.tgt = c;
          r.g[0] = 1; r.g[1] = 0; r.g[6] = fr; r.g[7] = fi;
        } else {
          r.tgt = -1;
          r.g[0] = fr; r.g[1] = fi; r.g[6] = fr; r.g[7] = fi;
        }
        r.g[2] = r.g[3] = r.g[4] = r.g[5] = 0;
        pending.push_back(r);
        alg_override_.push_back(gate_alg_bytes(q, nloc_, amp_bytes_));
        continue;
      }
      if (is_diag(r.g) && is_one(r.g[0], r.g[1]) && is_one(r.g[6], r.g[7])) { out.noop_gates++; continue; }
      pending.push_back(r);
      alg_override_.push_back(gate_alg_bytes(r, nloc_, amp_bytes_));
    }
    std::vector<uint64_t> alg = alg_override_;
    while (!pending.empty()) {
      std::vector<GateRec> rest;
      std::vector<uint64_t> rest_alg;
      out.sweeps.push_back(build_sweep(pending, alg, &rest, &rest_alg));
      pending.swap(rest);
      alg.swap(rest_alg);
    }
    return out;
  }

 private:
  int nloc_;
  uint64_t shard_;
  uint64_t amp_bytes_;
  int rb_cap_;
  std::vector<uint64_t> alg_override_;

  // One greedy pass: take, in order, every gate that (a) commutes with all the
  // earlier gates we had to skip and (b) fits the register tile.
  SweepPlan build_sweep(const std::vector<GateRec> &pending, const std::vector<uint64_t> &alg,
                        std::vector<GateRec> *rest, std::vector<uint64_t> *rest_alg) {
    SweepPlan sp;
    std::vector<int> regs;       // register bit positions, in order of first use
    uint64_t blocked_all = 0;    // bits a skipped gate acts densely on
    uint64_t blocked_diag = 0;   // bits a skipped gate acts diagonally on
    std::vector<const GateRec *> taken;
    bool any_dense = false;
    for (size_t i = 0; i < pending.size(); ++i) {
      const GateRec &r = pending[i];
      const bool diag = plan_diag(r.g, r.tgt);
      const uint64_t tb = (r.tgt >= 0) ? (1ull << r.tgt) : 0;
      const uint64_t dense_bits = diag ? 0 : tb;
      const uint64_t diag_bits = r.ctl_mask | (diag ? tb : 0);
      const bool can_pass = !(dense_bits & (blocked_all | blocked_diag)) && !(diag_bits & blocked_all);
      bool fits = (int)taken.size() < kMaxSweepOps;
      bool need_reg = false;
      if (fits && !diag && r.tgt >= kLaneBits) {
        if (std::find(regs.begin(), regs.end(), r.tgt) == regs.end()) {
          if ((int)regs.size() < rb_cap_) need_reg = true;
          else fits = false;
        }
      }
      if (can_pass && fits) {
        if (need_reg) regs.push_back(r.tgt);
        if (!diag) any_dense = true;
        taken.push_back(&r);
        sp.gates++;
        sp.alg_bytes += alg[i];
      } else {
        blocked_all |= dense_bits;
        blocked_diag |= diag_bits;
        rest->push_back(r);
        rest_alg->push_back(alg[i]);
      }
    }
    // Bits every taken gate requires to be 1 (only useful above the lane bits and
    // outside the register tile): fold them into the tile enumeration so that the
    // untouched part of the state is never read.
    uint64_t common = ~0ull;
    for (const GateRec *r : taken) {
      uint64_t req = r->ctl_mask;
      if (plan_diag(r->g, r->tgt) && r->tgt >= 0 && is_one(r->g[0], r->g[1])) req |= 1ull << r->tgt;
      common &= req;
    }
    uint64_t regmask = 0;
    for (int p : regs) regmask |= 1ull << p;
    common &= ~((1ull << kLaneBits) - 1) & ~regmask & ((1ull << nloc_) - 1);
    // keep enough free bits for the tile: need rb register bits among non-fixed bits
    int rb = std::max<int>((int)regs.size(), std::min(rb_cap_, any_dense ? rb_cap_ : 3));
    while (popc(common) > 0 && nloc_ - kLaneBits - popc(common) < rb) common &= common - 1;
    rb = std::min(rb, nloc_ - kLaneBits - popc(common));
    sp.fixed_ones = common;
    // pad the register tile with the lowest free bits (cheap, keeps runs long)
    for (int p = kLaneBits; p < nloc_ && (int)regs.size() < rb; ++p)
      if (!((regmask | common) >> p & 1ull)) { regs.push_back(p); regmask |= 1ull << p; }
    std::sort(regs.begin(), regs.end());
    sp.rb = (int)regs.size();
    for (int k = 0; k < sp.rb; ++k) sp.regpos[k] = regs[k];
    sp.ntiles = 1ull << (nloc_ - kLaneBits - sp.rb - popc(common));
    sp.swept_bytes = (sp.ntiles << (kLaneBits + sp.rb)) * amp_bytes_ * 2;
    emit_ops(taken, &sp);
    return sp;
  }

  int reg_index(const SweepPlan &sp, int pos) const {
    for (int k = 0; k < sp.rb; ++k) if (sp.regpos[k] == pos) return k;
    return -1;
  }

  // Split an index-bit mask into (lane part, register part, outside part); bits
  // fixed to one by the enumeration are dropped (always satisfied).
  void split_mask(const SweepPlan &sp, uint64_t m, uint32_t *lane, uint32_t *reg, uint64_t *outside) const {
    m &= ~sp.fixed_ones;
    *lane = (uint32_t)(m & ((1ull << kLaneBits) - 1));
    *reg = 0;
    *outside = 0;
    for (uint64_t t = m >> kLaneBits << kLaneBits; t; t &= t - 1) {
      const int b = __builtin_ctzll(t);
      const int ri = reg_index(sp, b);
      if (ri >= 0) *reg |= 1u << ri; else *outside |= 1ull << b;
    }
  }

  struct OpenDiag {  // a DIAG op still accepting gates
    int op_index = -1;
  };

  void emit_ops(const std::vector<const GateRec *> &taken, SweepPlan *sp) {
    // per bit: index of the last op acting densely on it (-1 none)
    int last_dense[64];
    for (int &v : last_dense) v = -1;
    int open_diag = -1;  // index in sp->ops of the most recent DIAG op
    // groups of the open diag op are collected here and flushed at the end
    struct PGroup { uint32_t lane, reg; double re, im; std::vector<OTerm> ot; };
    std::vector<std::vector<PGroup>> diag_groups;  // per DIAG op
    std::vector<int> diag_op_ids;
    auto diag_slot = [&](int op_id) -> std::vector<PGroup> & {
      for (size_t k = 0; k < diag_op_ids.size(); ++k) if (diag_op_ids[k] == op_id) return diag_groups[k];
      diag_op_ids.push_back(op_id);
      diag_groups.emplace_back();
      return diag_groups.back();
    };
    for (const GateRec *r : taken) {
      const bool diag = plan_diag(r->g, r->tgt);
      if (!diag) {
        SweepOp op{};
        uint32_t lane, reg; uint64_t outside;
        split_mask(*sp, r->ctl_mask, &lane, &reg, &outside);
        op.cm_thread = outside | lane;
        op.cm_reg = reg;
        memcpy(op.g, r->g, sizeof op.g);
        if (r->tgt < kLaneBits) { op.kind = OP_DENSE_LANE; op.tb = r->tgt; }
        else { op.kind = OP_DENSE_REG; op.tb = reg_index(*sp, r->tgt); }
        last_dense[r->tgt] = (int)sp->ops.size();
        sp->ops.push_back(op);
        continue;
      }
      // diagonal: amp *= d0 under controls (if d0 != 1), then amp *= d1/d0 where tgt set
      uint64_t bits = r->ctl_mask | (r->tgt >= 0 ? (1ull << r->tgt) : 0);
      int newest_dense = -1;
      for (uint64_t t = bits; t; t &= t - 1) newest_dense = std::max(newest_dense, last_dense[__builtin_ctzll(t)]);
      if (open_diag < 0 || open_diag < newest_dense) {
        SweepOp op{};
        op.kind = OP_DIAG;
        open_diag = (int)sp->ops.size();
        sp->ops.push_back(op);
      }
      auto &groups = diag_slot(open_diag);
      auto add_term = [&](uint64_t mask, double re, double im) {
        if (is_one(re, im)) return;
        uint32_t lane, reg; uint64_t outside;
        split_mask(*sp, mask, &lane, &reg, &outside);
        PGroup *g = nullptr;
        for (auto &pg : groups) if (pg.lane == lane && pg.reg == reg) { g = &pg; break; }
        if (!g) { groups.push_back(PGroup{lane, reg, 1.0, 0.0, {}}); g = &groups.back(); }
        if (outside == 0) {
          const double nr = g->re * re - g->im * im, ni = g->re * im + g->im * re;
          g->re = nr; g->im = ni;
        } else {
          for (auto &t : g->ot) if (t.mask == outside) {
            const double nr = t.re * re - t.im * im, ni = t.re * im + t.im * re;
            t.re = nr; t.im = ni;
            return;
          }
          g->ot.push_back(OTerm{outside, re, im});
        }
      };
      const double d0r = r->g[0], d0i = r->g[1], d1r = r->g[6], d1i = r->g[7];
      if (r->tgt < 0) {
        add_term(r->ctl_mask, d0r, d0i);
      } else if (is_one(d0r, d0i)) {
        add_term(bits, d1r, d1i);
      } else {
        add_term(r->ctl_mask, d0r, d0i);
        const double den = d0r * d0r + d0i * d0i;  // != 0: see plan_diag()
        const double qr = (d1r * d0r + d1i * d0i) / den, qi = (d1i * d0r - d1r * d0i) / den;
        add_term(bits, qr, qi);
      }
    }
    // flush diag groups into the flat arrays
    for (size_t k = 0; k < diag_op_ids.size(); ++k) {
      SweepOp &op = sp->ops[diag_op_ids[k]];
      op.group_off = (uint32_t)sp->groups.size();
      for (auto &pg : diag_groups[k]) {
        DGroup g{};
        g.lane_mask = pg.lane; g.reg_mask = pg.reg; g.re = pg.re; g.im = pg.im;
        g.oterm_off = (uint32_t)sp->oterms.size();
        g.n_oterms = (uint32_t)pg.ot.size();
        for (auto &t : pg.ot) sp->oterms.push_back(t);
        sp->groups.push_back(g);
      }
      op.n_groups = (uint32_t)(sp->groups.size() - op.group_off);
    }
  }
};

inline std::string plan_to_json(const std::vector<GateRec> &queue, int nloc, uint64_t shard, int bw = 128,
                                int max_rb = kMaxRegBits) {
  if (nloc < kLaneBits + 2) return "{\"sweeps\":[],\"note\":\"state too small for sweeps\"}";
  Planner pl(nloc, shard, bw, max_rb);
  PlanResult pr = pl.plan(queue);
  std::string s = "{\"noop_gates\":" + std::to_string(pr.noop_gates) + ",\"sweeps\":[";
  char buf[256];
  for (size_t i = 0; i < pr.sweeps.size(); ++i) {
    const SweepPlan &sp = pr.sweeps[i];
    int nd = 0, ndiag = 0;
    for (auto &o : sp.ops) (o.kind == OP_DIAG ? ndiag : nd)++;
    std::string rp = "[";
    for (int k = 0; k < sp.rb; ++k) rp += (k ? "," : "") + std::to_string(sp.regpos[k]);
    rp += "]";
    snprintf(buf, sizeof buf,
             "%s{\"gates\":%llu,\"dense_ops\":%d,\"diag_ops\":%d,\"groups\":%zu,\"oterms\":%zu,"
             "\"regpos\":%s,\"fixed_ones\":%llu,\"ntiles\":%llu,\"alg_bytes\":%llu,\"swept_bytes\":%llu}",
             i ? "," : "", (unsigned long long)sp.gates, nd, ndiag, sp.groups.size(), sp.oterms.size(),
             rp.c_str(), (unsigned long long)sp.fixed_ones, (unsigned long long)sp.ntiles,
             (unsigned long long)sp.alg_bytes, (unsigned long long)sp.swept_bytes);
    s += buf;
  }
  s += "]}";
  return s;
}

}  // namespace qh
