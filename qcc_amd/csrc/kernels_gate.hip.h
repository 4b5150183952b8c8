// kernels_gate.hip.h -- one-gate-per-launch kernels for gfx950 (MI355X).
//
// These are the "unfused" kernels: each launch applies exactly one reference
// gate application (apply1<> / applyc<>, /root/reference/src/lib/xgates.cc:23-67)
// to the amplitudes resident in HBM.  They are pure streaming kernels:
// arithmetic intensity is 28 flop per 64 B, far below the FP64 vector roof, so
// the only roofline that matters is HBM bandwidth and the design rules are
//   * every lane moves one 16-byte amplitude per load/store instruction
//     (global_load_dwordx4 / global_store_dwordx4, 1 KiB per wave instruction);
//   * untouched amplitudes are never read: control bits and the "bit set" half
//     of a diagonal gate are folded into the index enumeration (bit insertion),
//     so a CU1 moves S/2 bytes, not 2S -- EXCEPT bits 0 and 1: amplitudes that
//     differ only there share a 64-byte half-line, skipping them saves no HBM
//     traffic and turns the stores into 16-of-32-byte partial writes (measured:
//     4.9 ms instead of 2.9 ms for a CU1 controlled by bit 0 at 30 qubits), so
//     those bits are a per-lane predicate (`lowpred`) and whole lines are
//     rewritten;
//   * U independent work items per thread are loaded before any is used, to
//     keep >= 8 KiB of loads in flight per CU;
//   * 64-bit index arithmetic throughout.
//
// Algorithmic (minimal-touch) bytes per launch, S = bytes of the local state:
//   k_pair  no control: 2S      with c control bits: 2S / 2^c
//   k_diag  one-sided (d0 == 1): S / 2^c   two-sided: 2S / 2^c
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qh {

constexpr int kMaxIns = 15;   // 12 bits a plan may fold into the tile enumeration + 3 slab bits of a launch around an exchange
                              // (planner.h kMaxInsertBits + kMaxSlabBits; static_assert in kernels_sweep.hip.h)

// Sorted (ascending) list of bit positions at which a bit is inserted into a
// dense work-item counter to form an amplitude index; `ones` has the inserted
// bits that are fixed to 1 (controls / "bit set" half), the rest are 0 (the
// low element of a pair).
struct BitIns {
  int n;
  int pos[kMaxIns];
  uint64_t ones;
};

__device__ __forceinline__ uint64_t expand_index(uint64_t j, const BitIns &ins) {
#pragma unroll
  for (int k = 0; k < kMaxIns; ++k) {
    if (k < ins.n) {
      const uint64_t low = (1ull << ins.pos[k]) - 1ull;
      j = ((j & ~low) << 1) | (j & low);
    }
  }
  return j | ins.ones;
}

template <typename R> struct AmpT;
template <> struct AmpT<double> { using type = double2; };
template <> struct AmpT<float> { using type = float2; };

template <typename R> struct Gate2 {
  R g0r, g0i, g1r, g1i, g2r, g2i, g3r, g3i;
};

// Streaming access: every amplitude is read once and written once per launch and
// the state (>= 16 GiB) dwarfs the 256 MiB Infinity Cache, so loads and stores
// carry the non-temporal hint (measured +8% on MI355X, tools/membench).
template <bool NT, typename A> __device__ __forceinline__ A ld_amp(const A *p) {
  if constexpr (NT) {
    A v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
  } else {
    return *p;
  }
}
template <bool NT, typename A> __device__ __forceinline__ void st_amp(A *p, const A &v) {
  if constexpr (NT) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
  } else {
    *p = v;
  }
}

template <typename R, typename A>
__device__ __forceinline__ void butterfly(const Gate2<R> &g, A &a, A &b) {
  const R ar = a.x, ai = a.y, br = b.x, bi = b.y;
  A t1, t2;
  t1.x = (g.g0r * ar - g.g0i * ai) + (g.g1r * br - g.g1i * bi);
  t1.y = (g.g0r * ai + g.g0i * ar) + (g.g1r * bi + g.g1i * br);
  t2.x = (g.g2r * ar - g.g2i * ai) + (g.g3r * br - g.g3i * bi);
  t2.y = (g.g2r * ai + g.g2i * ar) + (g.g3r * bi + g.g3i * br);
  a = t1;
  b = t2;
}

// Dense / anti-diagonal 2x2 on bit p.  Work item j (one per amplitude PAIR)
// expands to the index of the pair's low element: a zero inserted at p, ones
// inserted at the control bits.
template <typename R, int U, bool GUARD, bool NT>
__global__ __launch_bounds__(256) void k_pair(typename AmpT<R>::type *__restrict__ psi,
                                               uint64_t nwork, int p, BitIns ins,
                                               Gate2<R> g, uint32_t lowpred) {
  using A = typename AmpT<R>::type;
  const uint64_t stride = (uint64_t)gridDim.x * (256ull * U);
  const uint64_t q2 = 1ull << p;
  for (uint64_t base = (uint64_t)blockIdx.x * (256ull * U) + threadIdx.x; base < nwork;
       base += stride) {
    A a[U], b[U];
    uint64_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t j = base + 256ull * u;
      if (!GUARD || j < nwork) {
        idx[u] = expand_index(j, ins);
        a[u] = ld_amp<NT>(&psi[idx[u]]);
        b[u] = ld_amp<NT>(&psi[idx[u] | q2]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t j = base + 256ull * u;
      if (!GUARD || j < nwork) {
        if (((uint32_t)idx[u] & lowpred) == lowpred) butterfly<R, A>(g, a[u], b[u]);
        st_amp<NT>(&psi[idx[u]], a[u]);
        st_amp<NT>(&psi[idx[u] | q2], b[u]);
      }
    }
  }
}

// Diagonal gate: amp *= (bit sel of index set ? f1 : f0); sel < 0 means f1
// always (one-sided gate: the enumeration already fixed the target bit to 1).
template <typename R, int U, bool GUARD, bool NT>
__global__ __launch_bounds__(256) void k_diag(typename AmpT<R>::type *__restrict__ psi,
                                               uint64_t nwork, int sel, BitIns ins, R f0r,
                                               R f0i, R f1r, R f1i, uint32_t lowpred) {
  using A = typename AmpT<R>::type;
  const uint64_t stride = (uint64_t)gridDim.x * (256ull * U);
  for (uint64_t base = (uint64_t)blockIdx.x * (256ull * U) + threadIdx.x; base < nwork;
       base += stride) {
    A a[U];
    uint64_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t j = base + 256ull * u;
      if (!GUARD || j < nwork) {
        idx[u] = expand_index(j, ins);
        a[u] = ld_amp<NT>(&psi[idx[u]]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t j = base + 256ull * u;
      if (!GUARD || j < nwork) {
        const bool hi = sel < 0 || ((idx[u] >> sel) & 1ull);
        const bool on = ((uint32_t)idx[u] & lowpred) == lowpred;
        const R fr = on ? (hi ? f1r : f0r) : (R)1, fi = on ? (hi ? f1i : f0i) : (R)0;
        A t;
        t.x = fr * a[u].x - fi * a[u].y;
        t.y = fr * a[u].y + fi * a[u].x;
        st_amp<NT>(&psi[idx[u]], t);
      }
    }
  }
}

// ---- wave-tile shapes (round 4) -----------------------------------------------------------------
// Chosen by measurement (tools/membench/pairbench.hip, profiles/r04/pairbench.txt; in place, 2^30 complex128):
//   * a wave owns 64*U consecutive work items (U KiB contiguous per stream), all its loads are issued before the
//     first result is needed, blocks of WPB waves take consecutive tiles, and the block index is rotated by 3 bits
//     (consecutive workgroups stream from 8 far-apart regions of the state: as k_sweep does);
//   * one-stream kernels (diagonal gates: the touched quarter / half of the state): U = 8, four waves -- 5.6-6.6 TB/s
//     against 5.0-5.9 for one item per thread;
//   * pair kernels: the same shape wins for target bits 3..8 (6.0-6.2 vs 5.9 TB/s), U = 16 / two waves for bits
//     20..25 (5.7-6.0 vs 5.5); elsewhere the round-3 shape (one pair per thread, 256-thread blocks) stays;
//   * a target INSIDE the 128-byte line (bits 0..2 of complex128, 0..3 of complex64) pairs amplitudes of one wave
//     row: every thread then loads ONE amplitude (whole lines, each fetched once) and takes its partner from the
//     neighbouring lane by DPP moves -- the two-loads-per-thread enumeration fetched every line twice (5.4 TB/s).
// (the launcher passes blk_bits = 0 for grids too small to rotate)
template <int ROT> __device__ __forceinline__ uint64_t rot_block(uint64_t bi, int blk_bits) {
  if (ROT && blk_bits > ROT) bi = ((bi >> ROT) | (bi << (blk_bits - ROT))) & ((1ull << blk_bits) - 1);
  return bi;
}

template <typename R, int U, int WPB, int ROT>
__global__ __launch_bounds__(64 * WPB) void k_pair_tile(typename AmpT<R>::type *__restrict__ psi, int p, BitIns ins,
                                                         Gate2<R> g, uint32_t lowpred, int blk_bits) {
  using A = typename AmpT<R>::type;
  const uint64_t bi = rot_block<ROT>(blockIdx.x, blk_bits);
  const uint64_t base = ((bi * WPB + (threadIdx.x >> 6)) * U) * 64ull + (threadIdx.x & 63);
  const uint64_t q2 = 1ull << p;
  A a[U], b[U];
  uint64_t idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) idx[u] = expand_index(base + 64ull * u, ins);
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = ld_amp<true>(&psi[idx[u]]);
#pragma unroll
  for (int u = 0; u < U; ++u) b[u] = ld_amp<true>(&psi[idx[u] | q2]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (((uint32_t)idx[u] & lowpred) == lowpred) butterfly<R, A>(g, a[u], b[u]);
    st_amp<true>(&psi[idx[u]], a[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) st_amp<true>(&psi[idx[u] | q2], b[u]);
}

template <typename R, int U, int WPB, int ROT>
__global__ __launch_bounds__(64 * WPB) void k_diag_tile(typename AmpT<R>::type *__restrict__ psi, int sel, BitIns ins,
                                                         R f0r, R f0i, R f1r, R f1i, uint32_t lowpred, int blk_bits) {
  using A = typename AmpT<R>::type;
  const uint64_t bi = rot_block<ROT>(blockIdx.x, blk_bits);
  const uint64_t base = ((bi * WPB + (threadIdx.x >> 6)) * U) * 64ull + (threadIdx.x & 63);
  A a[U];
  uint64_t idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) idx[u] = expand_index(base + 64ull * u, ins);
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = ld_amp<true>(&psi[idx[u]]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool hi = sel < 0 || ((idx[u] >> sel) & 1ull);
    const bool on = ((uint32_t)idx[u] & lowpred) == lowpred;
    const R fr = on ? (hi ? f1r : f0r) : (R)1, fi = on ? (hi ? f1i : f0i) : (R)0;
    A t;
    t.x = fr * a[u].x - fi * a[u].y;
    t.y = fr * a[u].y + fi * a[u].x;
    st_amp<true>(&psi[idx[u]], t);
  }
}

// value of lane (l ^ (1 << P)), P = 0..3, by DPP moves (no LDS): quad_perm for bits 0 and 1, row_half_mirror +
// quad_perm for bit 2, row_ror:8 for bit 3
template <int P> __device__ __forceinline__ int lane_xor_i32(int v) {
  if constexpr (P == 0) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);        // quad_perm:[1,0,3,2]
  else if constexpr (P == 1) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
  else if constexpr (P == 2) {
    const int t = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);                    // row_half_mirror: l -> l ^ 7
    return __builtin_amdgcn_update_dpp(0, t, 0x1B, 0xf, 0xf, false);                            // quad_perm:[3,2,1,0]: ^ 3
  } else return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);                      // row_ror:8: l ^ 8 within 16
}
template <int P> __device__ __forceinline__ double lane_xor(double v) {
  const long long w = __double_as_longlong(v);
  const int lo = lane_xor_i32<P>((int)(w & 0xffffffffll)), hi = lane_xor_i32<P>((int)(w >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int P> __device__ __forceinline__ float lane_xor(float v) {
  return __int_as_float(lane_xor_i32<P>(__float_as_int(v)));
}

// Dense / anti-diagonal 2x2 on a target bit P INSIDE the line: work item = one AMPLITUDE (controls inserted as ones,
// no zero at P); lane l and lane l ^ 2^P hold the pair.
template <typename R, int P, int U, int WPB, int ROT>
__global__ __launch_bounds__(64 * WPB) void k_pair_line(typename AmpT<R>::type *__restrict__ psi, BitIns ins, Gate2<R> g,
                                                         uint32_t lowpred, int blk_bits) {
  using A = typename AmpT<R>::type;
  const uint64_t bi = rot_block<ROT>(blockIdx.x, blk_bits);
  const uint64_t base = ((bi * WPB + (threadIdx.x >> 6)) * U) * 64ull + (threadIdx.x & 63);
  const bool hi = (threadIdx.x >> P) & 1u;      // (index bit P == lane bit P: the 64 items of a wave row are consecutive indices)
  // new = ca * own + cb * partner: row 0 of the matrix on the pair's low element, row 1 on the high one
  const R car = hi ? g.g3r : g.g0r, cai = hi ? g.g3i : g.g0i, cbr = hi ? g.g2r : g.g1r, cbi = hi ? g.g2i : g.g1i;
  A a[U];
  uint64_t idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) idx[u] = expand_index(base + 64ull * u, ins);
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = ld_amp<true>(&psi[idx[u]]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    A q;
    q.x = lane_xor<P>(a[u].x);
    q.y = lane_xor<P>(a[u].y);
    A t = a[u];
    if (((uint32_t)idx[u] & lowpred) == lowpred) {
      t.x = (car * a[u].x - cai * a[u].y) + (cbr * q.x - cbi * q.y);
      t.y = (car * a[u].y + cai * a[u].x) + (cbr * q.y + cbi * q.x);
    }
    st_amp<true>(&psi[idx[u]], t);
  }
}

// ---- initialisation ----------------------------------------------------------
template <typename R>
__global__ void k_set_one(typename AmpT<R>::type *psi, uint64_t index) {
  typename AmpT<R>::type one;
  one.x = (R)1;
  one.y = (R)0;
  psi[index] = one;
}

// Product state f_0 (x) ... (x) f_{k-1} written in place (circuit.py:121-164
// builds it with np.kron on the host).  Every amplitude is the product of one entry per
// factor; a factor is a table of complex128 values or a basis state (entry 1 at `basis`).
constexpr int kMaxFactors = 32;
struct ProductSpec {
  int nf;
  int identity;               // physical bit == logical bit (no remap since creation)
  int nglob;
  uint8_t shift[kMaxFactors]; // lowest LOGICAL bit of the factor
  uint8_t nq[kMaxFactors];
  uint8_t is_basis[kMaxFactors];
  uint32_t off[kMaxFactors];  // first table entry (complex128 units)
  uint64_t basis[kMaxFactors];
  uint8_t perm[64];           // physical bit of logical bit
};

template <typename R>
__global__ __launch_bounds__(256) void k_init_product(typename AmpT<R>::type *__restrict__ psi, uint64_t n,
                                                      uint64_t idx_high, ProductSpec sp,
                                                      const double2 *__restrict__ tab) {
  // grid-stride: a launch may not exceed 2^32 threads, a 2^33-amplitude shard has more amplitudes
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
  uint64_t logical = idx_high | i;
  if (!sp.identity) {
    const uint64_t phys = logical;
    logical = 0;
    for (int b = 0; b < sp.nglob; ++b) logical |= ((phys >> sp.perm[b]) & 1ull) << b;
  }
  double re = 1.0, im = 0.0;
  for (int f = 0; f < sp.nf; ++f) {
    const uint64_t v = (logical >> sp.shift[f]) & ((sp.nq[f] >= 64) ? ~0ull : ((1ull << sp.nq[f]) - 1ull));
    if (sp.is_basis[f]) {
      if (v != sp.basis[f]) { re = 0.0; im = 0.0; break; }
    } else {
      const double2 t = tab[sp.off[f] + v];
      const double nr = re * t.x - im * t.y;
      im = re * t.y + im * t.x;
      re = nr;
    }
  }
  typename AmpT<R>::type a;
  a.x = (R)re;
  a.y = (R)im;
  st_amp<true>(psi + i, a);
  }
}

// Out-of-place permutation of the index bits: dst[i] = src[P(i)], bit p of i supplying bit
// src_of_dst[p] of P(i).  Relayout sweeps never move the low line bits, so neighbouring lanes
// still read whole 128-byte lines.  Used to bring a re-laid-out state back to canonical order
// before amplitudes are handed out in physical order (download, device pointer).
struct BitPerm {
  int n;
  uint8_t src_of_dst[64];
};
template <typename R>
__global__ __launch_bounds__(256) void k_permute_bits(const typename AmpT<R>::type *__restrict__ src,
                                                      typename AmpT<R>::type *__restrict__ dst, uint64_t n, BitPerm bp) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    uint64_t j = 0;
    for (int p = 0; p < bp.n; ++p) j |= ((i >> p) & 1ull) << bp.src_of_dst[p];
    st_amp<true>(dst + i, ld_amp<true>(src + j));
  }
}

// ---- readers ----------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// out[0] += sum |a|^2 over indices i with (i & mask) == want (mask==0: all).
template <typename R>
__global__ __launch_bounds__(256) void k_norm2(const typename AmpT<R>::type *__restrict__ psi,
                                                uint64_t n, uint64_t mask, uint64_t want, double *out) {
  __shared__ double part[4];
  double acc = 0.0;
  // four independent (non-temporal) loads in flight per thread: a read-only pass is bound by the bytes in flight
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    typename AmpT<R>::type a[4];
    bool on[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      on[k] = ((i + k * stride) & mask) == want;
      if (on[k]) a[k] = ld_amp<true>(psi + i + k * stride);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (on[k]) acc += (double)a[k].x * (double)a[k].x + (double)a[k].y * (double)a[k].y;
  }
  for (; i < n; i += stride) {
    if ((i & mask) == want) {
      const auto a = ld_amp<true>(psi + i);
      acc += (double)a.x * (double)a.x + (double)a.y * (double)a.y;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// zero every amplitude whose bit `bit` differs from `value`.
template <typename R>
__global__ __launch_bounds__(256) void k_project(typename AmpT<R>::type *__restrict__ psi,
                                                  uint64_t nwork, BitIns ins) {
  typename AmpT<R>::type z;
  z.x = 0;
  z.y = 0;
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < nwork;
       j += (uint64_t)gridDim.x * 256)
    psi[expand_index(j, ins)] = z;
}

}  // namespace qh
