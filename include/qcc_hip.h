/*
 * qcc_hip.h -- C-ABI of the MI355X (gfx950) state-vector gate-application engine.
 *
 * This is the drop-in boundary for the hot path of qcc4cp/qcc: the two native
 * entry points of the reference's `libxgates` CPython extension
 *
 *     apply1(psi, gate, nbits, tgt, bit_width)        src/lib/xgates.cc:89-107
 *     applyc(psi, gate, nbits, ctl, tgt, bit_width)   src/lib/xgates.cc:126-145
 *
 * (templates apply1<> :23-41 and applyc<> :45-67; only callers are
 * qc.apply1/qc.applyc, src/lib/circuit.py:180-215).  Everything here is plain
 * C: opaque handle, raw pointers, sizes and ints.  No torch / numpy / Python
 * types cross this boundary.  INTEGRATION.md shows the binding a maintainer of
 * the reference adds (a `libxgates` module over ctypes).
 *
 * Conventions kept from the reference
 *   - amplitudes are interleaved (re,im), C-contiguous, length 2^nbits,
 *     complex128 when bit_width==128, complex64 when bit_width==64
 *     (src/lib/tensor.py:42-46); gates are 4 complex numbers, row-major
 *     [a b c d] (xgates.cc:18-21), ALWAYS passed here as 8 doubles;
 *   - qubit numbers in qh_apply1/qh_applyc/qh_host_* are the reference's
 *     big-endian qubit indices: qubit q is index bit (nbits-1-q)
 *     (xgates.cc:26,48-49);
 *   - updates are in place.
 * Deliberate differences
 *   - 64-bit indices (reference: int, breaks at 31 qubits, xgates.cc:27,33);
 *   - errors are returned as status codes + qh_last_error(), never exit()
 *     (reference: exit(EXIT_FAILURE), xgates.cc:28-32);
 *   - the state may live in HBM behind a handle across calls (the reference
 *     borrows a host NumPy buffer per call); qh_host_apply1/applyc keep the
 *     borrow-a-host-buffer contract for literal drop-in use.
 *
 * Threading: one host thread per handle.  Work is asynchronous on the handle's
 * HIP stream; qh_sync() waits.  All readers synchronise internally.
 */
#ifndef QCC_HIP_H_
#define QCC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qh_state_s *qh_handle;

/* status codes */
#define QH_OK 0
#define QH_ERR_BAD_QUBIT 1      /* qubit / bit position out of range            */
#define QH_ERR_SAME_QUBIT 2     /* control == target                            */
#define QH_ERR_BAD_DTYPE 3      /* bit_width not 64 / 128                       */
#define QH_ERR_HIP 4            /* a HIP runtime call failed (see last_error)   */
#define QH_ERR_ARG 5            /* NULL pointer, bad size, bad handle           */
#define QH_ERR_NOMEM 6          /* device allocation failed                     */
#define QH_ERR_NO_DEVICE 7      /* no gfx950 device visible                     */
#define QH_ERR_NONLOCAL 8       /* non-diagonal gate targets a bit held by the
                                   shard index: exchange first (qh_exchange_*)  */
#define QH_ERR_COMM 9           /* RCCL / transport failure (see last_error)    */

/* fusion levels for qh_set_fusion */
#define QH_FUSE_OFF 0           /* one kernel per gate, launched immediately    */
#define QH_FUSE_SWEEP 1         /* queue gates, plan register-tile sweeps       */

const char *qh_last_error(void);
int qh_version(void);
int qh_device_count(int *count);

/* ---- state lifetime ----------------------------------------------------- */
/* Allocate 2^nbits amplitudes in HBM on `device` (own HIP stream).           */
int qh_create(int nbits, int bit_width, int device, qh_handle *out);
/* Use caller-owned device memory / stream (e.g. a torch allocation and
 * torch's current stream, passed as raw pointers).  stream may be NULL
 * (engine creates its own).                                                  */
int qh_attach(int nbits, int bit_width, int device, void *device_ptr,
              void *hip_stream, qh_handle *out);
/* The state in pinned host memory the GPU works on directly (zero copy): for SMALL registers whose
 * owner wants the reference's contract literally -- apply1/applyc mutate the caller-visible buffer in
 * place (xgates.cc:37-38) and every holder of that buffer sees it.  qh_host_ptr gives the host
 * address (valid until qh_destroy; read it after qh_sync).  Every gate crosses PCIe: <= 28 qubits.   */
int qh_create_host_mapped(int nbits, int bit_width, int device, qh_handle *out);
int qh_host_ptr(qh_handle h, void **host_ptr);   /* NULL for states that live in HBM */
/* Planner-only handle: no device, no memory.  Gates can be queued and the
 * plan inspected with qh_plan_json (used by CPU-side tests).                 */
int qh_create_dry(int nbits, int bit_width, qh_handle *out);
int qh_destroy(qh_handle h);

/* Multi-GPU sharding: this handle holds the 2^nbits amplitudes whose global
 * index has high bits == shard_index, out of a 2^nbits_global state.  Gates
 * are then addressed in GLOBAL qubit numbers / bit positions.               */
int qh_set_shard(qh_handle h, int nbits_global, uint64_t shard_index);
/* Device pointer of the shard, amplitudes in physical order.  A handle that owns its memory may
 * re-lay the state out between two buffers while sweeps run (planner.h, relayout): this call
 * brings it back to canonical order first and, from then on, keeps the state in that one buffer
 * (as qh_set_relayout(h, 0) does): the pointer stays valid for the life of the handle.        */
int qh_device_ptr(qh_handle h, void **ptr);
/* Relayout sweeps (a second buffer of the state's size; planner.h): on = 1 allocates it now, on = 0 runs what is
 * queued, returns the layout to canonical order and frees it.  *actual (may be NULL) = the resulting mode.
 * Default: decided at the first flush (on if the handle owns its memory and the buffer fits).  A handle with a
 * communicator of SEVERAL ranks starts with relayout off: every rank must hold the same layout when they exchange,
 * so the caller turns it on only after all ranks report that they can (qcc_amd/sharded.py).                  */
int qh_set_relayout(qh_handle h, int on, int *actual);
int qh_stream(qh_handle h, void **stream);
int qh_nbits(qh_handle h, int *nbits_local, int *nbits_global);

/* ---- initialisation and host <-> device --------------------------------- */
/* |index> in global logical index space (zero elsewhere).                    */
int qh_init_basis(qh_handle h, uint64_t index);
/* Product state f_0 (x) f_1 (x) ... (x) f_{k-1} (np.kron order: f_0 on the most significant
 * qubits), built on the device -- what qc.reg/qubit/bitstring/state() build with
 * np.kron on the host (src/lib/circuit.py:121-164, state.py:185-246).
 * Factor j spans nq[j] qubits (sum == nbits_global) and is a table of 2^nq[j]
 * complex128 amplitudes (amps[j], interleaved re,im; nq[j] <= 24) or, where amps is NULL
 * or amps[j] is NULL, the basis state |basis[j]>.  At most 32 factors.           */
int qh_init_product(qh_handle h, int nfactors, const int *nq, const double *const *amps,
                    const uint64_t *basis);
/* offset/count in amplitudes of the LOCAL shard, physical order.             */
int qh_upload(qh_handle h, const void *host, uint64_t offset, uint64_t count);
int qh_download(qh_handle h, void *host, uint64_t offset, uint64_t count);

/* ---- the hot path -------------------------------------------------------- */
/* Reference semantics (xgates.cc:23-41): 2x2 `gate` on qubit `tgt`.          */
int qh_apply1(qh_handle h, int tgt, const double gate[8]);
/* Reference semantics (xgates.cc:45-67): `gate` on `tgt` where qubit `ctl` is 1. */
int qh_applyc(qh_handle h, int ctl, int tgt, const double gate[8]);
/* Generalisation in LOGICAL bit positions (bit 0 = least significant index
 * bit = reference qubit nbits-1): gate on bit `tgt_bit` for indices having
 * all bits of `ctl_mask` set.  Used by the multi-GPU layer and by native
 * multi-controlled gates.                                                    */
int qh_apply_bits(qh_handle h, uint64_t ctl_mask, int tgt_bit,
                  const double gate[8]);

/* The whole stream in one call: ops[2k] = control qubit of gate k or QH_NO_CTL (then qh_apply1), ops[2k+1] =
 * target qubit, gates + 8k = its 8 doubles -- exactly `count` qh_apply1/qh_applyc calls (xgates.cc:89-145), without
 * the per-call cost of the host language's FFI (ctypes: ~3 us per gate).  Stops at the first error.          */
#define QH_NO_CTL INT32_MIN
int qh_apply_stream(qh_handle h, uint64_t count, const int32_t *ops, const double *gates);

int qh_set_fusion(qh_handle h, int level);
/* Launch everything queued (no-op with QH_FUSE_OFF).                         */
int qh_flush(qh_handle h);
/* qh_flush + wait for the stream.                                            */
int qh_sync(qh_handle h);
/* Gates accepted but not yet executed.  After a failed flush these are the gates
 * that did NOT run (a planning/allocation failure keeps the whole queue; a failed
 * per-gate launch keeps that gate and every later one): the caller may change the
 * fusion level and flush again, or drop them.  (The reference has no counterpart:
 * xgates.cc:23-67 runs every gate synchronously or exits.)                    */
int qh_pending_gates(qh_handle h, uint64_t *count);
int qh_discard_pending(qh_handle h);

/* ---- logical -> physical bit map (global<->local qubit swaps) ----------- */
/* Records that the DATA of physical bits a and b has been exchanged (by the
 * communication layer for a >= nbits_local).  Later
 * gates are routed through the map; readers report physical indices, convert
 * with qh_phys_to_logical.                                                   */
int qh_remap_swap(qh_handle h, int phys_bit_a, int phys_bit_b);
int qh_get_bitmap(qh_handle h, int32_t *phys_of_logical /* [nbits_global] */);
int qh_phys_to_logical(qh_handle h, uint64_t phys_index, uint64_t *logical);
int qh_logical_to_phys(qh_handle h, uint64_t logical_index, uint64_t *phys);

/* ---- multi-GPU exchange (the reference has no distributed path) --------------
 * One process per GPU, each with one handle on its shard (qh_set_shard).  A dense gate whose
 * target is a shard bit needs the exchange below first; gates on local bits, controls on shard
 * bits and diagonal gates on shard bits never communicate.  Transport: RCCL send/recv over xGMI
 * on a second HIP stream (qh_comm_init), or a host-staged callback (qh_comm_init_custom: tests
 * with several ranks on one GPU, fabrics without peer access).
 *
 * qh_exchange_* first run the queued gates, then enqueue the exchange and RETURN: the exchange
 * proceeds in slabs, slab k leaving as soon as the part of the last queued sweep that writes it
 * has finished, and the first sweep submitted afterwards starts on slab k as soon as slab k has
 * arrived (HIP events between the handle's stream and the exchange streams; no host wait).
 * Every other entry point that touches the state waits for the arrivals first.
 * The bit map is NOT changed here: the caller records the swap (qh_remap_swap, or its own map). */
#define QH_COMM_ID_BYTES 128
/* One round of the host-staged transport: send bytes from send_host[i] to rank peers[i] and
 * receive the same number of bytes from it into recv_host[i], for all i at once; 0 = ok.   */
typedef int (*qh_round_fn)(void *user, int npeers, const int *peers, void *const *send_host,
                           void *const *recv_host, uint64_t bytes);
typedef struct {
  uint64_t exchanges;        /* qh_exchange_* calls                                        */
  uint64_t rounds;           /* grouped send/recv rounds                                   */
  uint64_t bytes_sent;       /* bytes this rank sent (== bytes received)                   */
  uint64_t slabs;            /* slabs the exchanges were cut into                          */
  uint64_t sweeps_overlapped;/* sweeps launched slab-wise around exchanges                 */
  double span_ms;            /* HIP-event time from the first send to the last landing     */
  uint64_t rounds_packed;    /* rounds moved through the gather / scatter kernels (blocks
                                without long contiguous runs: after relayout sweeps)       */
  uint64_t geometry_checks;  /* exchange geometries whose signature was compared with every
                                peer's and found equal (0 with one rank)                   */
  uint32_t comm_ranks;       /* size and ...                                               */
  uint32_t comm_rank;        /* ... rank as the TRANSPORT reports them (ncclCommCount /
                                ncclCommUserRank; the caller's arguments on the host-staged
                                transport)                                                 */
} qh_xstats;
/* How the last qh_exchange_* call of a handle was cut.  Every field must be the same on every rank (the planner keeps
 * rank-dependent gates as ghosts so that it is; `signature` is what the ranks compare before data moves): tools and tests
 * read it, on real handles and -- without any device -- on planner-only ones (qh_create_dry + qh_comm_init_dry).        */
typedef struct {
  uint64_t signature;        /* 64-bit hash of everything below + the bit map + planner switches + build            */
  uint64_t slab_mask;        /* local index bits whose values number the slabs (layout after the queued sweeps)     */
  uint64_t block_bits;       /* local index bits that select the block a peer gets                                  */
  uint64_t rounds_per_slab;  /* grouped send/recv rounds per slab                                                   */
  uint64_t staging_bytes;    /* staging area this exchange needs (receive halves + send halves of packed rounds)    */
  uint32_t slabs;            /* 2^popcount(slab_mask)                                                               */
  uint32_t chunk_bits;       /* log2(amplitudes per peer and round)                                                 */
  uint32_t packed;           /* 1: rounds go through the gather / scatter kernels, 0: sent from where they lie      */
  uint32_t peers;            /* peers per round                                                                     */
  uint32_t sweeps_before;    /* sweeps the queued gates were planned into (the last one precedes the exchange)      */
  uint32_t last_sweep_split; /* 1: that last sweep was launched slab by slab (overlaps the exchange)                */
} qh_xgeom;
int qh_exchange_geometry(qh_handle h, qh_xgeom *out);
/* Planner-only counterpart of qh_comm_init for handles made by qh_create_dry: qh_exchange_* then plan the queued gates,
 * decide slabs, rounds, chunk size and path exactly as a real handle of that rank would, and move nothing.          */
int qh_comm_init_dry(qh_handle h, int nranks, int rank);
int qh_comm_unique_id(void *id /* QH_COMM_ID_BYTES, rank 0; broadcast by the caller */);
int qh_comm_init(qh_handle h, int nranks, int rank, const void *id);
int qh_comm_init_custom(qh_handle h, int nranks, int rank, qh_round_fn fn, void *user);
int qh_comm_destroy(qh_handle h);
/* All g = log2(nranks) shard bits <-> local bits [base_bit, base_bit+g): rank r sends block j of
 * its shard to rank j and receives rank j's block r in its place (all peers at once).
 * chunk_amps: amplitudes per peer and round (0 = default 2^22).                             */
int qh_exchange_alltoall(qh_handle h, int base_bit, uint64_t chunk_amps);
/* Shard bit k (0..g-1) <-> local_bit: the half of the shard whose local_bit differs from this
 * rank's shard bit k is swapped with rank (r ^ 2^k)'s (one peer, one link).                 */
int qh_exchange_pair(qh_handle h, int shard_bit, int local_bit, uint64_t chunk_amps);
/* Self test of the transport on one rank: the two halves of the shard selected by local_bit are
 * sent to this rank itself and land exchanged (== an X gate on that bit), through the same
 * rounds / staging / slabs as a real exchange.                                              */
int qh_exchange_loopback(qh_handle h, int local_bit, uint64_t chunk_amps);
int qh_exchange_wait(qh_handle h);                 /* host wait for all arrivals             */
/* Host waits of a handle whose communicator has several ranks are bounded: if the stream has not drained within
 * QH_COMM_TIMEOUT_MS (default 300000) -- a peer is missing, or the ranks disagree about a round -- the call returns
 * QH_ERR_COMM instead of hanging, and the handle refuses further work.  Before data moves the ranks compare a signature
 * of the exchange geometry (every geometry a communicator has not seen yet; QH_EXCHANGE_VERIFY=1 every exchange, =0
 * never): a disagreement is QH_ERR_COMM with both signatures in qh_last_error().                                    */
int qh_exchange_stats(qh_handle h, qh_xstats *out);
/* sum over ranks of `count` doubles, in place (RCCL transport only): norms, probabilities    */
int qh_comm_allreduce_sum(qh_handle h, double *inout, int count);

/* ---- device-side readers (state.py:24-78) ------------------ */
int qh_norm2(qh_handle h, double *out);                       /* sum |a|^2 of the shard */
/* One amplitude by LOGICAL index (state.py:31-40 ampl/prob), read where it lives now: no
 * re-layout, 16 bytes over PCIe.  QH_ERR_NONLOCAL if another shard holds it.                */
int qh_amplitude(qh_handle h, uint64_t logical_index, double out[2]);
int qh_argmax(qh_handle h, uint64_t *phys_index, double *prob); /* max |a|^2 of the shard */
int qh_prob_bit(qh_handle h, int logical_bit, double *p1);    /* sum |a|^2 with bit set (shard) */
int qh_prob_bit_value(qh_handle h, int logical_bit, int value, double *p); /* ... with bit == value */
int qh_scale(qh_handle h, double re, double im);              /* a *= (re + i im) */
/* Project on logical_bit == value (zero the rest); caller renormalises with qh_scale. */
int qh_project_bit(qh_handle h, int logical_bit, int value);

/* ---- measurement of the engine itself ----------------------------------- */
typedef struct {
  uint64_t gates_submitted;   /* qh_apply* calls accepted                      */
  uint64_t kernels_launched;  /* gate/sweep kernels launched                   */
  uint64_t sweeps;            /* fused sweep launches                          */
  uint64_t bytes_algorithmic; /* minimal-touch bytes of the gates  */
  uint64_t bytes_swept;       /* bytes the launched kernels had to move        */
  uint64_t gates_noop;        /* gates skipped by shard-bit predicate          */
} qh_stats;
int qh_get_stats(qh_handle h, qh_stats *out);
int qh_reset_stats(qh_handle h);
/* hipEvent pair on the handle's stream.                                      */
int qh_timer_begin(qh_handle h);
int qh_timer_end(qh_handle h, float *milliseconds);
/* Lap marks: qh_timer_lap flushes what is queued and records an event on the stream (no host wait);
 * qh_timer_laps waits for the last mark, writes the milliseconds between consecutive marks (at most cap),
 * sets *count to the number of intervals and forgets the marks.  Per-step times of a loop without stalling it. */
int qh_timer_lap(qh_handle h);
int qh_timer_laps(qh_handle h, float *milliseconds, int cap, int *count);
/* JSON text of the sweeps the planner would launch for the current queue
 * (does not launch or clear).  Returns bytes needed; writes at most cap.     */
int qh_plan_json(qh_handle h, char *buf, uint64_t cap, uint64_t *needed);

/* The same plan in binary form, complete (ops, phase groups, tables): 3 x u64 (magic
 * 0x51485033, sweeps, gates dropped as no-ops), 64 bytes final_pos (position after the flush of
 * the index bit at each position before it), then per sweep 28 x i64 (rb, regpos[6],
 * regpos_store[6], lanehi[3], nwave, wavepos[2], fixed_ones, ntiles, #ops, #groups, #oterms,
 * #table doubles, #lane tables, lane_low, relayout), 64 bytes dest_pos, 20 x i64 (seat[6]: the index
 * bit on each lane bit when the tile is loaded, seat_store[6]: when it is stored, seat_dest[6]: the
 * position 0..5 a relayout store sends it to, wavepos at store time [2]), 25 x i64 (relayout store as the kernel gets it: reg_dest[6], wave_dest[2], number of
 * unit-index runs, 8 masks, 8 shifts), followed by the SweepOp / DGroup / OTerm / table arrays of
 * qcc_amd/csrc/planner.h, each padded to 8 bytes.  For tools and tests that check a plan
 * without a GPU (tests/plan_interp.py executes it with NumPy).                 */
int qh_plan_export(qh_handle h, void *buf, uint64_t cap, uint64_t *needed);

/* ---- literal drop-in on host buffers (what `libxgates` binds) ----------- */
/* psi: host pointer to 2^nbits complex numbers of width bit_width, updated in
 * place (H2D, kernel, D2H -- PCIe inclusive).  gate: 8 doubles.              */
int qh_host_apply1(void *psi, const double gate[8], int nbits, int tgt,
                   int bit_width);
int qh_host_applyc(void *psi, const double gate[8], int nbits, int ctl, int tgt,
                   int bit_width);
/* The two calls above keep device scratch per calling thread (the two most recent register
 * shapes); this frees the calling thread's.  Safe to call at any time, from any thread.       */
int qh_host_release(void);

#ifdef __cplusplus
}
#endif
#endif /* QCC_HIP_H_ */
