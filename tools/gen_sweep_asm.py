#!/usr/bin/env python3
"""Generates qcc_amd/csrc/sweep_island_rb{2..5}.inc: the per-tile body of the
fused sweep kernel (kernels_sweep.hip.h) as gfx950 assembly with FIXED physical
registers.

Why assembly: the tile (2^RB complex128 amplitudes per lane = up to 128 VGPRs)
must stay in the same registers across an interpreter loop over queued gate ops.
hipcc's register allocator copies the whole tile at every op-kind branch (2x the
VGPRs, spills at RB=5, one v_mov per amplitude per op), so the loop is written
by hand: every op updates the tile in place.

Two element types: complex128 (sweep_island_rb*.inc, a real number = a VGPR
pair) and complex64 (sweep_island_f32_rb*.inc, one VGPR per real number; the op
stream -- matrices, phase factors, tables -- stays double precision in memory and
is converted with v_cvt_f32_f64 as it is read).

Register map (island-private, declared as clobbers to the compiler), complex128:
  v[T0+4k .. T0+4k+3]   tile slot k: x = v[+0:+1], y = v[+2:+3]; T0 = 40
  v16..v39              24 temporaries, reused per op kind (see V below)
                        => 168 VGPRs at RB=5: THREE waves per SIMD
  s36..s99              scalar state (op header, gate matrix / group header,
                        cursors, masks, table entries)
  s16..s23              header of the NEXT op (prefetched); s24,s25 = address of the handler table
                        during the op loop (ops and DIAG-group apply code are reached by s_setpc_b64
                        into a table of s_branch instructions; the host numbers the handlers:
                        kernels_sweep.hip.h op_handler_id / group_handler_bits), store base at the end
Operands supplied by the C++ kernel:
  %0,%1  tile base address lo,hi (SGPR)      %2  SweepParams* (SGPR pair)
  %3     tile index (idx_high|base) (SGPR pair, for outside-bit predicates)
  %4     this lane's byte offset inside a tile (VGPR)  %5 lane id (VGPR)
  %6,%7  thread index lo,hi (VGPR)
Data layouts must match planner.h (SweepOp 96 B, DGroup 64 B, OTerm 24 B) and
kernels_sweep.hip.h (SweepParams: slot byte offsets at +0x40).
"""
import os
import re
import sys

# streaming tile loads/stores are non-temporal (each byte is touched once per sweep)
NT = '' if os.environ.get('QH_ISLAND_NT', '1') == '0' else ' nt'
T0 = 40
TEMP_LO, TEMP_HI = 16, 39
# QH_ISLAND_PROF=1: a measurement build of the complex128 RB=5 island (sweep_island_prof_rb5.inc, compiled
# only with -DQH_PROF, tools/probes/prof_island.sh): sampled waves write s_memtime at the start, after the
# tile has arrived, at the head of every op, after the stores are issued and after they have completed.
PROF = os.environ.get('QH_ISLAND_PROF') == '1'


class DT:
  """Element type of the tile: wide=True complex128, wide=False complex64."""
  wide = True


def W():
  return 2 if DT.wide else 1          # VGPRs per real number


def T(k):
  return T0 + 2 * W() * k


def V2(i):
  """The real number held at temp index i (a VGPR pair for f64, one VGPR for f32)."""
  return f'v[{i}:{i + 1}]' if DT.wide else f'v{i}'


def X(k):
  return V2(T(k))


def Y(k):
  return V2(T(k) + W())


def MUL():
  return 'v_mul_f64' if DT.wide else 'v_mul_f32'


def FMA():
  return 'v_fma_f64' if DT.wide else 'v_fma_f32'


def MOV():
  return 'v_mov_b64' if DT.wide else 'v_mov_b32'


def ADDS(d, x, y, neg=False):
  """d = x + y (neg: d = x - y)."""
  if DT.wide:
    return f'v_add_f64 {d}, {x}, {"-" if neg else ""}{y}'
  return f'{"v_sub_f32" if neg else "v_add_f32"} {d}, {x}, {y}'


class Asm:
  def __init__(self):
    self.lines = []

  def __call__(self, s):
    self.lines.append(s)

  def label(self, name):
    self.lines.append(f'{name}_%=:')


def L(name):
  return f'{name}_%='


# --- VGPR temporaries by context -------------------------------------------------------
# common
V_A, V_B = 16, 17
# dense register op: 4 result temporaries
R_T = [30, 32, 34, 36]
# dense lane op
LN_ADDR = 16
LN_TMP = 17
LN_COEF = {'car': 18, 'cai': 20, 'cbr': 22, 'cbi': 24}
LN_BUF = [26, 30]          # two shuffle buffers of 4 dwords
LN_T = [34, 36]
# diagonal op
D_C = (18, 20)             # c = cr + i ci
D_U = (22, 24)             # wave-uniform u
D_F = (26, 28)             # per-lane factor f
D_TMP = [30, 32, 34, 36]   # cmul temporaries (4 slots interleaved)
D_LTAB = 34                # lane-table entry lands in v[34:37] (free until the apply phase)


def vpair(r):
  """'v26' -> 'v[26:27]'"""
  n = int(r[1:])
  return f'v[{n}:{n + 1}]'


def pk_cmul(a, dst, src, f, tmp):
  """complex64, packed: dst = src * f (register pairs (re, im); dst may be src; tmp another pair)."""
  a(f'v_pk_mul_f32 {tmp}, {src}, {f} op_sel_hi:[1,0]')                                    # (x fr, y fr)
  a(f'v_pk_fma_f32 {dst}, {src}, {f}, {tmp} op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]')   # (x fr - y fi, y fr + x fi)


def cmul_slots(a, slots, fr, fi):
  """slot *= (fr,fi) for 1..4 slots, interleaved to hide the FP64 latency."""
  assert len(slots) <= 4
  if not DT.wide and fr[0] == 'v' and fi == f'v{int(fr[1:]) + 1}':
    f = vpair(fr)
    tm = [f'v[{t}:{t + 1}]' for t in D_TMP]
    for t, k in zip(tm, slots):
      a(f'v_pk_mul_f32 {t}, v[{T(k)}:{T(k) + 1}], {f} op_sel_hi:[1,0]')
    for t, k in zip(tm, slots):
      a(f'v_pk_fma_f32 v[{T(k)}:{T(k) + 1}], v[{T(k)}:{T(k) + 1}], {f}, {t} op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]')
    return
  tm = [V2(t) for t in D_TMP]
  # t = y*fi ; y = y*fr ; y += x*fi ; x = x*fr - t   (4 FP64 ops, result in place)
  for t, k in zip(tm, slots):
    a(MUL() + f' {t}, {Y(k)}, {fi}')
  for t, k in zip(tm, slots):
    a(MUL() + f' {Y(k)}, {Y(k)}, {fr}')
  for t, k in zip(tm, slots):
    a(FMA() + f' {Y(k)}, {X(k)}, {fi}, {Y(k)}')
  for t, k in zip(tm, slots):
    a(FMA() + f' {X(k)}, {X(k)}, {fr}, -{t}')


def cmul_vv(a, xr, xi, fr, fi, tmp):
  """(xr,xi) *= (fr,fi), all 64-bit register pairs (VGPR or SGPR factor)."""
  a(MUL() + f' {tmp}, {xi}, {fi}')
  a(MUL() + f' {xi}, {xi}, {fr}')
  a(FMA() + f' {xi}, {xr}, {fi}, {xi}')
  a(FMA() + f' {xr}, {xr}, {fr}, -{tmp}')


def gen(rb, wide=True, prof=False):
  global D_C, D_U, D_F
  DT.wide = wide
  # (re, im) of a temporary complex number sit in ADJACENT registers: complex64 multiplies by them with
  # packed FP32 instructions (v_pk_mul_f32 / v_pk_fma_f32 take 64-bit register pairs)
  D_C, D_U, D_F = (18, 18 + W()), (22, 22 + W()), (26, 26 + W())
  nr = 1 << rb
  a = Asm()
  batch = min(8, nr)
  nrec = [0]

  def prof_rec(wait_stores=False, real_at=None):
    """s[30:31] = this wave's row of the profile buffer (0: not sampled), s101 = byte offset of the next record."""
    if not prof:
      return
    nrec[0] += 1
    skip = f'L_prof{nrec[0]}'
    a('s_cmp_eq_u64 s[30:31], 0')
    a(f's_cbranch_scc1 {L(skip)}')
    if wait_stores:
      a('s_waitcnt vmcnt(0)')
    a('s_memtime s[28:29]')
    a('s_waitcnt lgkmcnt(0)')
    a('v_mov_b32 v30, s28')
    a('v_mov_b32 v31, s29')
    a('v_mov_b32 v32, s101')
    a('global_store_dwordx2 v32, v[30:31], s[30:31]')
    a('s_add_u32 s101, s101, 8')
    a('s_and_b32 s101, s101, 0x3ff')              # 128 records per row
    if real_at is not None:                       # the 100 MHz counter beside it: records 126 / 127 of the row
      a('s_memrealtime s[28:29]')
      a('s_waitcnt lgkmcnt(0)')
      a('v_mov_b32 v30, s28')
      a('v_mov_b32 v31, s29')
      a(f'v_mov_b32 v32, {real_at * 8}')
      a('global_store_dwordx2 v32, v[30:31], s[30:31]')
    a.label(skip)

  def tile_io(store):
    # the store uses its own slot offsets (+0x240) and a base corrected by the index bits
    # OP_WSWAP moved between the wave id and the registers (they are not swapped back)
    blo, bhi, table = ('s24', 's25', 0x240) if store else ('%0', '%1', 0x40)
    if store:
      # in place: the load address corrected by the index bits OP_WSWAP moved (mask = ~0);
      # relayout sweep: the tile's own contiguous block of the second buffer (mask = 0, the
      # kernel passes that block's address and lane offsets as the store operands)
      a('s_load_dwordx2 s[74:75], %2, 0x28')      # SweepParams::store_delta_mask
      a('s_mov_b64 s[72:73], %3')
      a('s_sub_u32 s24, s72, s26')                # tile index now - tile index at load time
      a('s_subb_u32 s25, s73, s27')
      a('s_waitcnt lgkmcnt(0)')
      a('s_and_b64 s[24:25], s[24:25], s[74:75]')
      a(f's_lshl_b64 s[24:25], s[24:25], {2 + W()}')
      a('s_add_u32 s24, s24, %[sblo]')
      a('s_addc_u32 s25, s25, %[sbhi]')
    for j in range(nr // batch):
      a(f's_load_dwordx{2 * batch} s[52:{52 + 2 * batch - 1}], %2, {table + 8 * batch * j}')
      a('s_waitcnt lgkmcnt(0)')
      for i in range(batch):
        k = batch * j + i
        a(f's_add_u32 s98, {blo}, s{52 + 2 * i}')
        a(f's_addc_u32 s99, {bhi}, s{53 + 2 * i}')
        dw = 'dwordx4' if DT.wide else 'dwordx2'
        regs = f'v[{T(k)}:{T(k) + 2 * W() - 1}]'
        if store:
          a(f'global_store_{dw} %[svoff], {regs}, s[98:99]' + NT)
        else:
          a(f'global_load_{dw} {regs}, %4, s[98:99]' + NT)

  # ---- prologue: parameters, then one 1-KiB global_load_dwordx4 per slot ------------
  if prof:
    a('s_mov_b64 s[30:31], %[prow]')
    a('s_mov_b32 s101, 0')
    prof_rec(real_at=126)
  a('s_load_dwordx4 s[36:39], %2, 0x0')   # ops cursor, groups base
  a('s_load_dwordx2 s[40:41], %2, 0x10')  # oterms base
  a('s_load_dwordx2 s[42:43], %2, 0x18')  # s42 = number of ops (unused: a sentinel ends the list), s43 = tables - groups (bytes)
  tile_io(store=False)
  a('s_mov_b64 s[26:27], %3')                     # tile index at load time (see the store)
  a('s_load_dwordx8 s[16:23], s[36:37], 0x0')     # header of the first op
  a('s_waitcnt vmcnt(0)')
  prof_rec()

  # ---- op loop ----------------------------------------------------------------------
  # The 32-byte op header (kind tb cm_reg n_groups cm_thread(2) group_off flags) of op i+1 is
  # fetched into s[16:23] while op i runs: the scalar-load latency (hundreds of cycles behind
  # the tile stream) is off the critical path.  The 64-byte matrix g[8] is loaded only by the
  # op kinds that read it (most ops of a QFT or supremacy sweep do not).
  # Dispatch: the host puts a handler number into the upper half of `kind` (planner.h op_handler_id);
  # s[24:25] holds the address of a table of s_branch instructions, one per handler.  A taken branch
  # costs a wave ~50 cycles (the instruction buffer refills), and the compare-and-branch chains this
  # replaces took 3-5 of them per op plus 5-10 not-taken ones: ~400 cycles of the ~650 a 64-instruction
  # register butterfly held its wave (tools/probes/prof_island.sh).  Every handler ends with its own copy
  # of the loop head (next_op), so an op costs two jumps: s_setpc_b64 into the table, s_branch to the code.
  # The op list ends with a sentinel whose handler is the store (HID_DONE): no counter, no end test.
  HID_DIAG, HID_DENSE, HID_WSWAP, HID_BFL, HID_DPP, HID_DONE, HID_LSWAP, HID_BFREG, NHID = 0, 1, 2, 3, 4, 5, 8, 24, 64

  def op_head():
    prof_rec()
    a('s_waitcnt lgkmcnt(0)')
    a('s_mov_b64 s[44:45], s[16:17]')
    a('s_mov_b64 s[46:47], s[18:19]')
    a('s_mov_b64 s[48:49], s[20:21]')
    a('s_mov_b64 s[50:51], s[22:23]')
    a('s_load_dwordx8 s[16:23], s[36:37], 0x60')   # next op's header (the buffer is padded: reading one past the end is harmless)
    a('s_lshr_b32 s74, s44, 16')                   # handler number
    a('s_and_b32 s44, s44, 0xffff')                # kind
    a('s_lshl_b32 s74, s74, 2')
    a('s_add_u32 s72, s24, s74')
    a('s_addc_u32 s73, s25, 0')
    a('s_setpc_b64 s[72:73]')

  def next_op():
    a('s_mov_b64 exec, -1')                     # (waves are always full: 64 x k threads per block)
    a('s_add_u32 s36, s36, 96')
    a('s_addc_u32 s37, s37, 0')
    op_head()

  targets = ['L_next'] * NHID
  targets[HID_DIAG], targets[HID_DENSE], targets[HID_WSWAP] = 'L_diag', 'L_dense', 'L_wswap'
  targets[HID_BFL], targets[HID_DPP], targets[HID_DONE] = 'L_bfl_e', 'L_dpp', 'L_done'
  for r in range(rb):
    targets[HID_LSWAP + 2 * r], targets[HID_LSWAP + 2 * r + 1] = f'L_lswap16_r{r}', f'L_lswap32_r{r}'
  for v in range(5):
    for b in range(rb):
      targets[HID_BFREG + 8 * v + b] = f'L_bf{v}_{b}'
  a('s_getpc_b64 s[24:25]')                     # address of the next instruction; the table starts 12 bytes on
  a('s_add_u32 s24, s24, 12')
  a('s_addc_u32 s25, s25, 0')
  a(f's_branch {L("L_op")}')
  for t in targets:
    a(f's_branch {L(t)}')
  gmasks = [1 << b for b in range(rb)] + [(1 << b0) | (1 << b1) for b0 in range(rb) for b1 in range(b0 + 1, rb)]
  for t in ['L_gm0'] + [f'L_gm{m}' for m in gmasks] + ['L_gmx'] + [f'L_gbf{j}' for j in range(rb)]:   # apply handlers of DIAG groups (see L_diag)
    a(f's_branch {L(t)}')
  a.label('L_op')
  op_head()
  a.label('L_dense')
  a('s_load_dwordx16 s[52:67], s[36:37], 0x20')  # g[8]
  # control predicate of this thread: (it & cm_thread) == cm_thread  -> s[68:69]
  # (zero-controls: header words n_groups / group_off of a dense op hold the bits of cm_thread
  # that must be 0; bits 8..12 of cm_reg the register bits that must be 0)
  a('s_andn2_b32 s74, s48, s47')
  a('s_andn2_b32 s75, s49, s50')
  a(f's_and_b32 s76, s46, {(1 << rb) - 1:#x}')      # register bits that must be one
  a(f's_bfe_u32 s77, s46, {(rb << 16) | 8:#x}')      # ... that must be zero (bits 8.. of cm_reg)
  a(f'v_and_b32 v{V_A}, s48, %6')
  a(f'v_and_b32 v{V_B}, s49, %7')
  a(f'v_cmp_eq_u32 vcc, s74, v{V_A}')
  a(f'v_cmp_eq_u32_e64 s[72:73], s75, v{V_B}')
  a('s_nop 1')
  a('s_and_b64 s[68:69], vcc, s[72:73]')
  a('s_waitcnt lgkmcnt(0)')                   # g[8] (and the header prefetch)
  # REAL fast paths: all four matrix entries real (x, ry, cx, ccx ...) and no REGISTER-bit
  # control.  Lane / outside-bit controls just narrow EXEC: the real paths update in place,
  # so disabled lanes keep their amplitudes (a gate pair always shares its predicate).
  a('s_bitcmp1_b32 s51, 2')                   # OPF_REAL
  a(f's_cbranch_scc0 {L("L_generic")}')
  a('s_bitcmp1_b32 s51, 7')                   # OPF_LANE_DPP on a real lane op: partner by DPP, no LDS
  a(f's_cbranch_scc1 {L("L_lrd")}')
  a('s_and_b64 exec, exec, s[68:69]')
  a(f's_cbranch_execz {L("L_next")}')
  a('s_cmp_eq_u32 s46, 0')
  a(f's_cbranch_scc1 {L("L_real")}')
  # register-bit controls (ccx, ladders of multi_control): same in-place real arithmetic,
  # slots whose index misses a control bit are skipped one by one
  a('s_cmp_eq_u32 s44, 1')
  a(f's_cbranch_scc1 {L("L_lane_real_c")}')
  for b in range(rb):
    a(f's_cmp_eq_u32 s45, {b}')
    a(f's_cbranch_scc1 {L(f"L_rrc{b}")}')
  next_op()
  a.label('L_generic')
  a('s_cmp_eq_u32 s44, 1')
  a(f's_cbranch_scc1 {L("L_lane")}')
  for b in range(rb):
    a(f's_cmp_eq_u32 s45, {b}')
    a(f's_cbranch_scc1 {L(f"L_reg{b}")}')
  a.label('L_next')
  next_op()

  # ---- dense 2x2 on register bit b: in-place butterflies ----------------------------
  gnames = ['g0r', 'g0i', 'g1r', 'g1i', 'g2r', 'g2i', 'g3r', 'g3i']
  if DT.wide:
    g = {nm: f's[{52 + 2 * i}:{53 + 2 * i}]' for i, nm in enumerate(gnames)}
  else:
    g = {nm: f'v{16 + i}' for i, nm in enumerate(gnames)}

  def load_matrix_f32(first=16):
    """complex64 tile: the op's double-precision matrix -> 8 floats in v[first..first+7]."""
    if not DT.wide:
      for i in range(8):
        a(f'v_cvt_f32_f64 v{first + i}, s[{52 + 2 * i}:{53 + 2 * i}]')
  t0, t1, t2, t3 = (V2(t) for t in R_T)
  for b in range(rb):
    a.label(f'L_reg{b}')
    load_matrix_f32()
    for h in range(nr // 2):
      k0 = ((h >> b) << (b + 1)) | (h & ((1 << b) - 1))
      k1 = k0 | (1 << b)
      skip = f'L_r{b}_{h}'
      a(f's_andn2_b32 s74, s76, {k0}')      # control bits (register part) not set in k0
      a(f's_and_b32 s75, s77, {k0}')
      a('s_or_b32 s74, s74, s75')
      a('s_cmp_eq_u32 s74, 0')
      a(f's_cbranch_scc0 {L(skip)}')
      ar, ai, br, bi = X(k0), Y(k0), X(k1), Y(k1)
      a(MUL() + f' {t0}, {g["g0r"]}, {ar}')
      a(MUL() + f' {t1}, {g["g0r"]}, {ai}')
      a(MUL() + f' {t2}, {g["g2r"]}, {ar}')
      a(MUL() + f' {t3}, {g["g2r"]}, {ai}')
      a(FMA() + f' {t0}, -{g["g0i"]}, {ai}, {t0}')
      a(FMA() + f' {t1}, {g["g0i"]}, {ar}, {t1}')
      a(FMA() + f' {t2}, -{g["g2i"]}, {ai}, {t2}')
      a(FMA() + f' {t3}, {g["g2i"]}, {ar}, {t3}')
      a(FMA() + f' {t0}, {g["g1r"]}, {br}, {t0}')
      a(FMA() + f' {t1}, {g["g1r"]}, {bi}, {t1}')
      a(FMA() + f' {t2}, {g["g3r"]}, {br}, {t2}')
      a(FMA() + f' {t3}, {g["g3r"]}, {bi}, {t3}')
      a(FMA() + f' {t0}, -{g["g1i"]}, {bi}, {t0}')
      a(FMA() + f' {t1}, {g["g1i"]}, {br}, {t1}')
      a(FMA() + f' {t2}, -{g["g3i"]}, {bi}, {t2}')
      a(FMA() + f' {t3}, {g["g3i"]}, {br}, {t3}')
      a('s_and_saveexec_b64 s[70:71], s[68:69]')
      a(MOV() + f' {ar}, {t0}')
      a(MOV() + f' {ai}, {t1}')
      a(MOV() + f' {br}, {t2}')
      a(MOV() + f' {bi}, {t3}')
      a('s_mov_b64 exec, s[70:71]')
      a.label(skip)
    next_op()

  # ---- REAL uncontrolled dense ops: half the arithmetic, results in place ---------------
  a.label('L_real')
  a('s_cmp_eq_u32 s44, 1')
  a(f's_cbranch_scc1 {L("L_lane_real")}')
  for b in range(rb):
    a(f's_cmp_eq_u32 s45, {b}')
    a(f's_cbranch_scc1 {L(f"L_rr{b}")}')
  next_op()
  for b in range(rb):
    a.label(f'L_rr{b}')
    load_matrix_f32()
    pairs = []
    for h in range(nr // 2):
      k0 = ((h >> b) << (b + 1)) | (h & ((1 << b) - 1))
      pairs.append((k0, k0 | (1 << b)))
    for i in range(0, len(pairs), 2):           # two pairs interleaved (4 temporaries)
      grp = pairs[i:i + 2]
      tmps = [(V2(R_T[2 * j]), V2(R_T[2 * j + 1])) for j in range(len(grp))]
      for (k0, k1), (ta, tb) in zip(grp, tmps):
        a(MUL() + f' {ta}, {g["g0r"]}, {X(k0)}')
        a(MUL() + f' {tb}, {g["g0r"]}, {Y(k0)}')
      for (k0, k1), (ta, tb) in zip(grp, tmps):
        a(FMA() + f' {ta}, {g["g1r"]}, {X(k1)}, {ta}')
        a(FMA() + f' {tb}, {g["g1r"]}, {Y(k1)}, {tb}')
      for (k0, k1), (ta, tb) in zip(grp, tmps):
        a(MUL() + f' {X(k1)}, {g["g3r"]}, {X(k1)}')
        a(MUL() + f' {Y(k1)}, {g["g3r"]}, {Y(k1)}')
      for (k0, k1), (ta, tb) in zip(grp, tmps):
        a(FMA() + f' {X(k1)}, {g["g2r"]}, {X(k0)}, {X(k1)}')
        a(FMA() + f' {Y(k1)}, {g["g2r"]}, {Y(k0)}, {Y(k1)}')
      for (k0, k1), (ta, tb) in zip(grp, tmps):
        a(MOV() + f' {X(k0)}, {ta}')
        a(MOV() + f' {Y(k0)}, {tb}')
    next_op()
  def lane_pipeline(first_buf, depth, combine_slot):
    """Partner values of slot k arrive by ds_bpermute `depth` slots ahead of their use
    (the shuffle latency, not its issue rate, bounds a lane op: SQ_WAIT_INST_LDS was 42%
    of the wave time with one slot of lookahead)."""
    bufs = [first_buf + 2 * W() * j for j in range(depth)]
    assert bufs[-1] + 2 * W() - 1 <= TEMP_HI and (depth - 1) * 2 * W() <= 15

    def shuf_to(k, buf):
      for d in range(2 * W()):
        a(f'ds_bpermute_b32 v{buf + d}, v{LN_ADDR}, v{T(k) + d}')

    for k in range(min(depth, nr)):
      shuf_to(k, bufs[k % depth])
    for k in range(nr):
      ahead = min(k + depth - 1, nr - 1) - k
      a(f's_waitcnt lgkmcnt({ahead * 2 * W()})')
      buf = bufs[k % depth]
      combine_slot(k, V2(buf), V2(buf + W()))
      if k + depth < nr:
        shuf_to(k + depth, buf)


  for b in range(rb):                            # real gate on register bit b under register controls
    a.label(f'L_rrc{b}')
    load_matrix_f32()
    ta, tb_ = V2(R_T[0]), V2(R_T[1])
    for h in range(nr // 2):
      k0 = ((h >> b) << (b + 1)) | (h & ((1 << b) - 1))
      k1 = k0 | (1 << b)
      skip = f'L_rrc{b}_{h}'
      a(f's_andn2_b32 s74, s76, {k0}')
      a(f's_and_b32 s75, s77, {k0}')
      a('s_or_b32 s74, s74, s75')
      a('s_cmp_eq_u32 s74, 0')
      a(f's_cbranch_scc0 {L(skip)}')
      a(MUL() + f' {ta}, {g["g0r"]}, {X(k0)}')
      a(MUL() + f' {tb_}, {g["g0r"]}, {Y(k0)}')
      a(FMA() + f' {ta}, {g["g1r"]}, {X(k1)}, {ta}')
      a(FMA() + f' {tb_}, {g["g1r"]}, {Y(k1)}, {tb_}')
      a(MUL() + f' {X(k1)}, {g["g3r"]}, {X(k1)}')
      a(MUL() + f' {Y(k1)}, {g["g3r"]}, {Y(k1)}')
      a(FMA() + f' {X(k1)}, {g["g2r"]}, {X(k0)}, {X(k1)}')
      a(FMA() + f' {Y(k1)}, {g["g2r"]}, {Y(k0)}, {Y(k1)}')
      a(MOV() + f' {X(k0)}, {ta}')
      a(MOV() + f' {Y(k0)}, {tb_}')
      a.label(skip)
    next_op()

  # lane bit, real: new = ca*mine + cb*other with real per-lane ca, cb -- 4 FP64 ops per slot
  a.label('L_lane_real')
  a.label('L_lane_real_c')
  a('s_lshl_b32 s74, 1, s45')
  a(f'v_xor_b32 v{LN_ADDR}, s74, %5')
  a(f'v_lshlrev_b32 v{LN_ADDR}, 2, v{LN_ADDR}')
  a(f'v_and_b32 v{LN_TMP}, s74, %5')
  a(f'v_cmp_ne_u32 vcc, 0, v{LN_TMP}')
  a('s_bitcmp1_b32 s51, 1')                     # USE_C needs complex coefficients: generic path
  a(f's_cbranch_scc1 {L("L_lane_c1")}')
  if DT.wide:
    for v, (lo, hi) in ((LN_COEF['car'], (52, 64)), (LN_COEF['cbr'], (56, 60))):
      for d in range(2):
        a(f'v_mov_b32 v{v + d}, s{lo + d}')
        a(f'v_mov_b32 v{LN_TMP}, s{hi + d}')
        a(f'v_cndmask_b32 v{v + d}, v{v + d}, v{LN_TMP}, vcc')
  else:
    load_matrix_f32(26)                          # g0r g0i g1r g1i g2r g2i g3r g3i -> v26..v33
    a(f'v_cndmask_b32 v{LN_COEF["car"]}, v26, v32, vcc')   # hi ? g3r : g0r
    a(f'v_cndmask_b32 v{LN_COEF["cbr"]}, v28, v30, vcc')   # hi ? g2r : g1r
  rca, rcb = V2(LN_COEF['car']), V2(LN_COEF['cbr'])

  def shuf_r(k, buf):
    for d in range(2 * W()):
      a(f'ds_bpermute_b32 v{buf + d}, v{LN_ADDR}, v{T(k) + d}')

  def comb_real(k, pr, pi):
    a(MUL() + f' {X(k)}, {rca}, {X(k)}')
    a(MUL() + f' {Y(k)}, {rca}, {Y(k)}')
    a(FMA() + f' {X(k)}, {rcb}, {pr}, {X(k)}')
    a(FMA() + f' {Y(k)}, {rcb}, {pi}, {Y(k)}')

  def comb_real_c(k, pr, pi):                    # register-bit controls: combine only the selected slots
    skip = f'L_lrc_{k}'
    a(f's_andn2_b32 s74, s76, {k}')
    a(f's_and_b32 s75, s77, {k}')
    a('s_or_b32 s74, s74, s75')
    a('s_cmp_eq_u32 s74, 0')
    a(f's_cbranch_scc0 {L(skip)}')
    comb_real(k, pr, pi)
    a.label(skip)
  a('s_cmp_eq_u32 s46, 0')
  a(f's_cbranch_scc0 {L("L_lane_real_cc")}')
  lane_pipeline(24, 4, comb_real)
  next_op()
  a.label('L_lane_real_cc')
  lane_pipeline(24, 4, comb_real_c)              # (all slots are shuffled: the pipeline's wait counts stay static)
  next_op()

  # ---- unit-entry butterflies (OPF_BFLY): the gate is c*M with M's entries in {1,-1,i,-i};
  # the planner moved c into another op of the sweep, so M costs adds only, in place.
  #   variant (flags bits 4..6): 0  [[1, 1],[ 1,-1]]  (h)        1  [[1,-1],[1,1]]  (yroot)
  #   2  [[1,1],[-1,1]] (yroot^+)   3  [[1,-i],[-i,1]] (v, sqrt-x)   4  [[1,i],[i,1]] (v^+)
  a.label('L_bf')                               # (section marker; handlers are reached through the table)
  for v in range(5):
    for b in range(rb):
      a.label(f'L_bf{v}_{b}')
      pairs = []
      for h in range(nr // 2):
        k0 = ((h >> b) << (b + 1)) | (h & ((1 << b) - 1))
        pairs.append((k0, k0 | (1 << b)))
      if not DT.wide:
        # complex64: one packed add and one packed fma per pair ((re, im) of a slot = one 64-bit register pair);
        # K = (2, 2), signs and the re/im exchange of the v gates by neg_* / op_sel
        a('v_mov_b32 v16, 2.0')
        a('v_mov_b32 v17, 2.0')
        P = lambda k: f'v[{T(k)}:{T(k) + 1}]'
        first = {0: '', 1: ' neg_lo:[0,1] neg_hi:[0,1]', 2: '',
                 3: ' op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]',       # a' = (ar + bi, ai - br)
                 4: ' op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]'}[v]    # a' = (ar - bi, ai + br)
        second = {0: ' neg_lo:[0,1,0] neg_hi:[0,1,0]',                    # b' = a' - 2b
                  1: '',                                                  # b' = a' + 2b
                  2: ' neg_lo:[0,0,1] neg_hi:[0,0,1]',                    # b' = 2b - a'
                  3: ' op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]',  # b' = (2br + ai', 2bi - ar')
                  4: ' op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1]'}[v]   # b' = (2br - ai', 2bi + ar')
        for i in range(0, len(pairs), 8):
          grp = pairs[i:i + 8]
          for k0, k1 in grp:
            a(f'v_pk_add_f32 {P(k0)}, {P(k0)}, {P(k1)}{first}')
          for k0, k1 in grp:
            a(f'v_pk_fma_f32 {P(k1)}, {P(k1)}, v[16:17], {P(k0)}{second}')
        next_op()
        continue
      for i in range(0, len(pairs), 4):           # four pairs interleaved: 8 independent chains
        grp = pairs[i:i + 4]
        if v == 0:      # a' = a + b ; b' = a - b = a' - 2b
          for k0, k1 in grp:
            a(ADDS(X(k0), X(k0), X(k1)))
            a(ADDS(Y(k0), Y(k0), Y(k1)))
          for k0, k1 in grp:
            a(FMA() + f' {X(k1)}, -2.0, {X(k1)}, {X(k0)}')
            a(FMA() + f' {Y(k1)}, -2.0, {Y(k1)}, {Y(k0)}')
        elif v == 1:    # a' = a - b ; b' = a + b = a' + 2b
          for k0, k1 in grp:
            a(ADDS(X(k0), X(k0), X(k1), neg=True))
            a(ADDS(Y(k0), Y(k0), Y(k1), neg=True))
          for k0, k1 in grp:
            a(FMA() + f' {X(k1)}, 2.0, {X(k1)}, {X(k0)}')
            a(FMA() + f' {Y(k1)}, 2.0, {Y(k1)}, {Y(k0)}')
        elif v == 2:    # a' = a + b ; b' = b - a = 2b - a'
          for k0, k1 in grp:
            a(ADDS(X(k0), X(k0), X(k1)))
            a(ADDS(Y(k0), Y(k0), Y(k1)))
          for k0, k1 in grp:
            a(FMA() + f' {X(k1)}, 2.0, {X(k1)}, -{X(k0)}')
            a(FMA() + f' {Y(k1)}, 2.0, {Y(k1)}, -{Y(k0)}')
        elif v == 3:    # a' = a - i b ; b' = b - i a
          for k0, k1 in grp:
            a(ADDS(X(k0), X(k0), Y(k1)))              # ar' = ar + bi
            a(ADDS(Y(k0), Y(k0), X(k1), neg=True))    # ai' = ai - br
          for k0, k1 in grp:
            a(FMA() + f' {Y(k1)}, 2.0, {Y(k1)}, -{X(k0)}')   # bi' = bi - ar = 2bi - ar'
            a(FMA() + f' {X(k1)}, 2.0, {X(k1)}, {Y(k0)}')    # br' = br + ai = 2br + ai'
        else:           # a' = a + i b ; b' = b + i a
          for k0, k1 in grp:
            a(ADDS(X(k0), X(k0), Y(k1), neg=True))    # ar' = ar - bi
            a(ADDS(Y(k0), Y(k0), X(k1)))              # ai' = ai + br
          for k0, k1 in grp:
            a(FMA() + f' {Y(k1)}, 2.0, {Y(k1)}, {X(k0)}')    # bi' = bi + ar = 2bi + ar'
            a(FMA() + f' {X(k1)}, 2.0, {X(k1)}, -{Y(k0)}')   # br' = br - ai = 2br - ai'
      next_op()
  # lane bit: partner p via ds_bpermute, own value o
  a.label('L_bfl_e')
  a('s_bfe_u32 s74, s51, 0x30004')            # butterfly variant
  a.label('L_bfl')
  a('s_bitcmp1_b32 s51, 7')                   # OPF_LANE_DPP: partner values by DPP moves (VALU), not LDS
  a(f's_cbranch_scc1 {L("L_dpp")}')
  a('s_lshl_b32 s75, 1, s45')
  a(f'v_xor_b32 v{LN_ADDR}, s75, %5')
  a(f'v_lshlrev_b32 v{LN_ADDR}, 2, v{LN_ADDR}')
  a(f'v_and_b32 v{LN_TMP}, s75, %5')
  a(f'v_cmp_ne_u32 vcc, 0, v{LN_TMP}')           # this lane holds the "1" element of the pair
  coef = V2(LN_COEF['car'])
  for v, lab in ((0, 'L_bfl_a'), (1, 'L_bfl_b'), (2, 'L_bfl_b'), (3, 'L_bfl_v'), (4, 'L_bfl_w')):
    a(f's_cmp_eq_u32 s74, {v}')
    a(f's_cbranch_scc1 {L(lab)}')
  next_op()

  def bf_lane(form):
    def comb(k, pr, pi):
      if form == 'a':      # new = alpha*o + p
        a(FMA() + f' {X(k)}, {coef}, {X(k)}, {pr}')
        a(FMA() + f' {Y(k)}, {coef}, {Y(k)}, {pi}')
      elif form == 'b':    # new = o + beta*p
        a(FMA() + f' {X(k)}, {coef}, {pr}, {X(k)}')
        a(FMA() + f' {Y(k)}, {coef}, {pi}, {Y(k)}')
      elif form == 'v':    # new = o - i p
        a(ADDS(X(k), X(k), pi))
        a(ADDS(Y(k), Y(k), pr, neg=True))
      else:                # new = o + i p
        a(ADDS(X(k), X(k), pi, neg=True))
        a(ADDS(Y(k), Y(k), pr))
    lane_pipeline(20, 4, comb)
    next_op()

  c0 = LN_COEF['car']
  a.label('L_bfl_a')                               # h: alpha = +1 on the 0-lane, -1 on the 1-lane
  if DT.wide:
    a(f'v_mov_b32 v{c0}, 0')
    a(f'v_mov_b32 v{c0 + 1}, 0x3ff00000')
    a(f'v_mov_b32 v{LN_TMP}, 0xbff00000')
    a(f'v_cndmask_b32 v{c0 + 1}, v{c0 + 1}, v{LN_TMP}, vcc')
  else:
    a(f'v_mov_b32 v{LN_TMP}, -1.0')
    a(f'v_cndmask_b32 v{c0}, 1.0, v{LN_TMP}, vcc')
  bf_lane('a')
  a.label('L_bfl_b')                               # yroot / yroot^+: beta = g[0] on the 0-lane, g[1] on the 1-lane
  a('s_load_dwordx4 s[52:55], s[36:37], 0x20')
  a('s_waitcnt lgkmcnt(0)')
  if DT.wide:
    for d in range(2):
      a(f'v_mov_b32 v{c0 + d}, s{52 + d}')
      a(f'v_mov_b32 v{LN_TMP}, s{54 + d}')
      a(f'v_cndmask_b32 v{c0 + d}, v{c0 + d}, v{LN_TMP}, vcc')
  else:
    a(f'v_cvt_f32_f64 v{c0}, s[52:53]')
    a(f'v_cvt_f32_f64 v{LN_TMP}, s[54:55]')
    a(f'v_cndmask_b32 v{c0}, v{c0}, v{LN_TMP}, vcc')
  bf_lane('b')
  a.label('L_bfl_v')
  bf_lane('v')
  a.label('L_bfl_w')
  bf_lane('w')


  # ---- OP_LSWAP: lane bit tb (4 or 5) <-> register bit r (header field cm_reg), in place --
  # v_permlane{16,32}_swap exchanges the odd rows / upper half of one register with the even
  # rows / lower half of another: applied to slots (k, k^1) it moves the pair a lane-bit gate
  # acts on into ONE lane (registers k and k^1), i.e. afterwards that index bit is register
  # bit r and the old register bit r is the lane bit.  The planner emits the gate as a
  # register op in between and swaps back (the op is an involution).  No LDS traffic:
  # ds_bpermute issues once per ~6 cycles per CU, these run at VALU rate.
  a.label('L_lswap')                            # (section marker)
  for r in range(rb):
    for name, ins in ((f'L_lswap16_r{r}', 'v_permlane16_swap_b32'), (f'L_lswap32_r{r}', 'v_permlane32_swap_b32')):
      a.label(name)
      for k in range(nr):
        if k & (1 << r):
          continue
        for d in range(2 * W()):
          a(f'{ins} v{T(k) + d}, v{T(k | (1 << r)) + d}')
      # the thread's own bit moves with the exchange: header n_groups = the lane bit's index position,
      # cm_thread = that bit | the register bit's position (which holds 0 in a thread index)
      a('s_lshl_b64 s[72:73], 1, s47')
      a(f'v_and_b32 v{V_A}, s72, %6')
      a(f'v_and_b32 v{V_B}, s73, %7')
      a(f'v_or_b32 v{V_A}, v{V_A}, v{V_B}')
      a(f'v_cmp_ne_u32 vcc, 0, v{V_A}')
      a(f'v_xor_b32 v{V_A}, s48, %6')
      a(f'v_xor_b32 v{V_B}, s49, %7')
      a(f'v_cndmask_b32 %6, %6, v{V_A}, vcc')
      a(f'v_cndmask_b32 %7, %7, v{V_B}, vcc')
      next_op()

  # ---- OP_WSWAP: wave bit tb <-> register bit r (header field cm_reg) --------------------
  # The 2^W waves of a workgroup hold the tiles of ONE super-tile: they differ in W chosen
  # index bits ("wave bits").  A dense gate on such a bit pairs amplitudes of two waves; the
  # exchange below transposes that bit with register bit r: the wave whose bit is 0 hands its
  # slots with bit r set to the partner wave and receives the partner's other slots into them (and vice
  # versa), through a 2^W x (half x 16 lines) LDS buffer, `half` slots per pass, two barriers
  # per pass.  Afterwards the gate is a register op; the planner swaps back before the store.
  # The wave's own index bits change with the layout: header field cm_thread holds
  # (1 << old wave-bit position) | (1 << position of register bit r); it is XORed into the
  # tile index and the thread index when this wave's bit is 1.
  a.label('L_wswap')
  half = min(8, nr // 2)
  slot_bytes = 64 * 4 * W() * 2 // 2            # bytes one slot of one wave takes (64 lanes x complex)
  slot_bytes = 64 * 2 * W() * 4
  region = half * slot_bytes
  a('s_lshr_b32 s74, %8, s45')
  a('s_and_b32 s74, s74, 1')                    # x = this wave's bit
  a('s_lshl_b32 s75, 1, s45')
  a('s_xor_b32 s75, %8, s75')                   # partner wave
  a(f's_mul_i32 s72, %8, {region}')
  a('s_add_u32 s72, s72, %9')
  a(f's_mul_i32 s73, s75, {region}')
  a('s_add_u32 s73, s73, %9')
  a(f'v_lshlrev_b32 v16, {2 + W()}, %5')        # lane * bytes per amplitude
  a('v_add_u32 v17, s72, v16')                  # where this wave writes
  a('v_add_u32 v18, s73, v16')                  # where the partner wrote
  wr = 'ds_write_b128' if DT.wide else 'ds_write_b64'
  rd = 'ds_read_b128' if DT.wide else 'ds_read_b64'
  for r in range(rb):
    a(f's_cmp_eq_u32 s46, {r}')
    a(f's_cbranch_scc1 {L(f"L_wswap_r{r}")}')
  next_op()
  for r in range(rb):
    a.label(f'L_wswap_r{r}')
    a('s_cmp_eq_u32 s74, 0')
    a(f's_cbranch_scc0 {L(f"L_wswap_even_r{r}")}')
    for name, parity in ((f'L_wswap_odd_r{r}', 1), (f'L_wswap_even_r{r}', 0)):
      a.label(name)
      slots = [k for k in range(nr) if ((k >> r) & 1) == parity]
      for p0 in range(0, len(slots), half):
        part = slots[p0:p0 + half]
        for j, k in enumerate(part):
          a(f'{wr} v17, v[{T(k)}:{T(k) + 2 * W() - 1}] offset:{j * slot_bytes}')
        a('s_waitcnt lgkmcnt(0)')
        a('s_barrier')
        for j, k in enumerate(part):
          a(f'{rd} v[{T(k)}:{T(k) + 2 * W() - 1}], v18 offset:{j * slot_bytes}')
        a('s_waitcnt lgkmcnt(0)')
        a('s_barrier')
      if parity == 1:
        next_op()                # bit 0: index bits unchanged (both positions hold 0)
      else:
        a('s_xor_b64 %3, %3, s[48:49]')
        a('v_xor_b32 %6, s48, %6')
        a('v_xor_b32 %7, s49, %7')
        next_op()

  # ---- butterfly on lane bit 0..3 with DPP partner fetch (OPF_LANE_DPP) -------------------
  # new.re = o.re + beta_re * q.re ; new.im = o.im + beta_im * q.im, q = partner value, or the
  # partner with re/im exchanged (flags bit 8; v / v^+).  beta = +-1 per lane: g[0..3] =
  # beta_re(0-lane), beta_re(1-lane), beta_im(0-lane), beta_im(1-lane).
  a.label('L_dpp')
  a('s_load_dwordx8 s[52:59], s[36:37], 0x20')     # beta_re, beta_im per lane half
  a('s_lshl_b32 s75, 1, s45')
  a(f'v_and_b32 v{LN_TMP}, s75, %5')
  a(f'v_cmp_ne_u32 vcc, 0, v{LN_TMP}')
  BRE, BIM, Q, TQ = 18, (22 if DT.wide else 19), 24, 28     # complex64: (beta_re, beta_im) adjacent for the packed fma
  a('s_waitcnt lgkmcnt(0)')
  if DT.wide:
    for v, (lo, hi) in ((BRE, (52, 54)), (BIM, (56, 58))):
      for d in range(2):
        a(f'v_mov_b32 v{v + d}, s{lo + d}')
        a(f'v_mov_b32 v{LN_TMP}, s{hi + d}')
        a(f'v_cndmask_b32 v{v + d}, v{v + d}, v{LN_TMP}, vcc')
  else:
    for v, (lo, hi) in ((BRE, (52, 54)), (BIM, (56, 58))):
      a(f'v_cvt_f32_f64 v{v}, s[{lo}:{lo + 1}]')
      a(f'v_cvt_f32_f64 v{LN_TMP}, s[{hi}:{hi + 1}]')
      a(f'v_cndmask_b32 v{v}, v{v}, v{LN_TMP}, vcc')
  bre, bim = V2(BRE), V2(BIM)
  a('s_bfe_u32 s75, s51, 0x10008')              # 1: partner's re/im exchanged
  a('s_lshl_b32 s75, s75, 2')
  a('s_add_u32 s75, s75, s45')                  # 4*exchanged + tb
  for code in range(8):
    a(f's_cmp_eq_u32 s75, {code}')
    a(f's_cbranch_scc1 {L(f"L_dpp{code}")}')
  next_op()
  DPP1 = {0: ['quad_perm:[1,0,3,2]'], 1: ['quad_perm:[2,3,0,1]'],
          2: ['row_half_mirror', 'quad_perm:[3,2,1,0]'], 3: ['row_ror:8']}
  for code in range(8):
    tbv, exch = code & 3, code >> 2
    a.label(f'L_dpp{code}')
    a('s_nop 1')
    steps = DPP1[tbv]
    for k in range(nr):
      nd = 2 * W()
      # source dwords: q.re from the partner's re (or im when exchanged), q.im likewise
      src = [T(k) + d for d in range(nd)]
      if exch:
        src = src[W():] + src[:W()]
      if len(steps) == 1:
        for d in range(nd):
          a(f'v_mov_b32_dpp v{Q + d}, v{src[d]} {steps[0]} row_mask:0xf bank_mask:0xf')
      else:
        for d in range(nd):
          a(f'v_mov_b32_dpp v{TQ + d}, v{src[d]} {steps[0]} row_mask:0xf bank_mask:0xf')
        if nd < 3:
          a('s_nop 1')
        for d in range(nd):
          a(f'v_mov_b32_dpp v{Q + d}, v{TQ + d} {steps[1]} row_mask:0xf bank_mask:0xf')
      if DT.wide:
        a(FMA() + f' {X(k)}, {bre}, {V2(Q)}, {X(k)}')
        a(FMA() + f' {Y(k)}, {bim}, {V2(Q + W())}, {Y(k)}')
      else:
        a(f'v_pk_fma_f32 v[{T(k)}:{T(k) + 1}], v[{BRE}:{BRE + 1}], v[{Q}:{Q + 1}], v[{T(k)}:{T(k) + 1}]')
    next_op()


  # ---- real 2x2 on lane bit 0..3, partner by DPP moves (OPF_LANE_DPP on a REAL lane op) ------
  # new = ca*own + cb*partner; the control predicate is folded into the coefficients
  # (ca = 1, cb = 0 where it fails), so EXEC stays full and the two-step DPP moves may pass
  # through lanes the gate does not act on.
  a.label('L_lrd')
  a('s_lshl_b32 s74, 1, s45')
  a(f'v_and_b32 v{LN_TMP}, s74, %5')
  a(f'v_cmp_ne_u32 vcc, 0, v{LN_TMP}')
  CA, CB, Q2, TQ2 = 18, 18 + 2 * W(), 24, 28
  if DT.wide:
    for v, (lo, hi) in ((CA, (52, 64)), (CB, (56, 60))):
      for d in range(2):
        a(f'v_mov_b32 v{v + d}, s{lo + d}')
        a(f'v_mov_b32 v{LN_TMP}, s{hi + d}')
        a(f'v_cndmask_b32 v{v + d}, v{v + d}, v{LN_TMP}, vcc')
    # predicate fails: ca = 1.0, cb = 0.0
    a(f'v_mov_b32 v{LN_TMP}, 0x3ff00000')
    a(f'v_cndmask_b32 v{CA}, 0, v{CA}, s[68:69]')
    a(f'v_cndmask_b32 v{CA + 1}, v{LN_TMP}, v{CA + 1}, s[68:69]')
    a(f'v_cndmask_b32 v{CB}, 0, v{CB}, s[68:69]')
    a(f'v_cndmask_b32 v{CB + 1}, 0, v{CB + 1}, s[68:69]')
  else:
    a(f'v_cvt_f32_f64 v{CA}, s[52:53]')
    a(f'v_cvt_f32_f64 v{LN_TMP}, s[64:65]')
    a(f'v_cndmask_b32 v{CA}, v{CA}, v{LN_TMP}, vcc')
    a(f'v_cvt_f32_f64 v{CB}, s[56:57]')
    a(f'v_cvt_f32_f64 v{LN_TMP}, s[60:61]')
    a(f'v_cndmask_b32 v{CB}, v{CB}, v{LN_TMP}, vcc')
    a(f'v_cndmask_b32 v{CA}, 1.0, v{CA}, s[68:69]')
    a(f'v_cndmask_b32 v{CB}, 0, v{CB}, s[68:69]')
  for tbv in range(4):
    a(f's_cmp_eq_u32 s45, {tbv}')
    a(f's_cbranch_scc1 {L(f"L_lrd{tbv}")}')
  next_op()
  ca2, cb2 = V2(CA), V2(CB)
  for tbv in range(4):
    a.label(f'L_lrd{tbv}')
    a('s_nop 1')
    steps = DPP1[tbv]
    for k in range(nr):
      nd = 2 * W()
      skip = f'L_lrd{tbv}_{k}'
      a(f's_andn2_b32 s74, s76, {k}')             # register-bit controls: skip the slots they exclude
      a(f's_and_b32 s75, s77, {k}')
      a('s_or_b32 s74, s74, s75')
      a('s_cmp_eq_u32 s74, 0')
      a(f's_cbranch_scc0 {L(skip)}')
      if len(steps) == 1:
        for d in range(nd):
          a(f'v_mov_b32_dpp v{Q2 + d}, v{T(k) + d} {steps[0]} row_mask:0xf bank_mask:0xf')
      else:
        for d in range(nd):
          a(f'v_mov_b32_dpp v{TQ2 + d}, v{T(k) + d} {steps[0]} row_mask:0xf bank_mask:0xf')
        if nd < 3:
          a('s_nop 1')
        for d in range(nd):
          a(f'v_mov_b32_dpp v{Q2 + d}, v{TQ2 + d} {steps[1]} row_mask:0xf bank_mask:0xf')
      a(MUL() + f' {X(k)}, {ca2}, {X(k)}')
      a(MUL() + f' {Y(k)}, {ca2}, {Y(k)}')
      a(FMA() + f' {X(k)}, {cb2}, {V2(Q2)}, {X(k)}')
      a(FMA() + f' {Y(k)}, {cb2}, {V2(Q2 + W())}, {Y(k)}')
      a.label(skip)
    next_op()

  # ---- dense 2x2 on a lane bit: partner via ds_bpermute -------------------------------
  a.label('L_lane')
  a('s_lshl_b32 s74, 1, s45')                     # m = 1 << tb
  a(f'v_xor_b32 v{LN_ADDR}, s74, %5')
  a(f'v_lshlrev_b32 v{LN_ADDR}, 2, v{LN_ADDR}')   # bpermute byte address of the partner lane
  a(f'v_and_b32 v{LN_TMP}, s74, %5')
  a(f'v_cmp_ne_u32 vcc, 0, v{LN_TMP}')            # this lane holds the "1" element of the pair
  a.label('L_lane_c1')
  # deferred factor c of the preceding DIAG op lives in v[18:21] = the coefficient
  # registers: move it to v[34:37] and fetch the partner lane's c into v[26:29] first
  a('s_bitcmp1_b32 s51, 1')
  a(f's_cbranch_scc0 {L("L_lane_c0")}')
  cc = 34 if DT.wide else 38                        # f64: v[34:35], v[36:37]; f32: v38, v39
  a(MOV() + f' {V2(cc)}, {V2(D_C[0])}')
  a(MOV() + f' {V2(cc + W())}, {V2(D_C[1])}')
  for d in range(2 * W()):
    a(f'ds_bpermute_b32 v{LN_BUF[0] + d}, v{LN_ADDR}, v{cc + d}')
  a.label('L_lane_c0')
  # new = ca*mine + cb*other ; ca = hi ? g3 : g0 ; cb = hi ? g2 : g1
  src = {'car': (52, 64), 'cai': (54, 66), 'cbr': (56, 60), 'cbi': (58, 62)}
  if DT.wide:
    for name, v in LN_COEF.items():
      lo, hi = src[name]
      for d in range(2):
        a(f'v_mov_b32 v{v + d}, s{lo + d}')
        a(f'v_mov_b32 v{LN_TMP}, s{hi + d}')
        a(f'v_cndmask_b32 v{v + d}, v{v + d}, v{LN_TMP}, vcc')
  else:
    # the partner's c (USE_C) may be arriving in v26,v27: convert the matrix into the
    # cmul temporaries' neighbourhood instead (v30..v37), then select per lane
    load_matrix_f32(30)
    a(f'v_cndmask_b32 v{LN_COEF["car"]}, v30, v36, vcc')   # hi ? g3r : g0r
    a(f'v_cndmask_b32 v{LN_COEF["cai"]}, v31, v37, vcc')   # hi ? g3i : g0i
    a(f'v_cndmask_b32 v{LN_COEF["cbr"]}, v32, v34, vcc')   # hi ? g2r : g1r
    a(f'v_cndmask_b32 v{LN_COEF["cbi"]}, v33, v35, vcc')   # hi ? g2i : g1i
  car, cai, cbr, cbi = (V2(LN_COEF[n]) for n in ('car', 'cai', 'cbr', 'cbi'))
  # USE_C (flags bit 1): the preceding DIAG op left its per-lane factor c un-applied;
  # H.diag(c) acts as  new = (ca c_mine) mine + (cb c_other) other  -- two complex
  # products per LANE instead of one per amplitude.
  a('s_bitcmp1_b32 s51, 1')
  a(f's_cbranch_scc0 {L("L_lane_nc")}')
  a('s_waitcnt lgkmcnt(0)')
  cmul_vv(a, car, cai, V2(cc), V2(cc + W()), V2(LN_BUF[1]))
  cmul_vv(a, cbr, cbi, V2(LN_BUF[0]), V2(LN_BUF[0] + W()), V2(LN_BUF[1]))
  a.label('L_lane_nc')

  def shuf(k, buf):
    for d in range(2 * W()):
      a(f'ds_bpermute_b32 v{buf + d}, v{LN_ADDR}, v{T(k) + d}')

  def combine(k, buf):
    orr, oi = V2(buf), V2(buf + W())
    u0, u1 = V2(LN_T[0]), V2(LN_T[1])
    a(MUL() + f' {u0}, {car}, {X(k)}')
    a(MUL() + f' {u1}, {car}, {Y(k)}')
    a(FMA() + f' {u0}, -{cai}, {Y(k)}, {u0}')
    a(FMA() + f' {u1}, {cai}, {X(k)}, {u1}')
    a(FMA() + f' {u0}, {cbr}, {orr}, {u0}')
    a(FMA() + f' {u1}, {cbr}, {oi}, {u1}')
    a(FMA() + f' {u0}, -{cbi}, {oi}, {u0}')
    a(FMA() + f' {u1}, {cbi}, {orr}, {u1}')
    a('s_and_saveexec_b64 s[70:71], s[68:69]')
    a(MOV() + f' {X(k)}, {u0}')
    a(MOV() + f' {Y(k)}, {u1}')
    a('s_mov_b64 exec, s[70:71]')

  a('s_cmp_eq_u32 s46, 0')
  a(f's_cbranch_scc0 {L("L_lane_ctl")}')
  # fast path (no register-bit controls): shuffles of slot k+1 in flight while slot k combines
  shuf(0, LN_BUF[0])
  for k in range(nr):
    if k + 1 < nr:
      shuf(k + 1, LN_BUF[(k + 1) & 1])
      a(f's_waitcnt lgkmcnt({2 * W()})')
    else:
      a('s_waitcnt lgkmcnt(0)')
    combine(k, LN_BUF[k & 1])
  next_op()
  a.label('L_lane_ctl')
  for k in range(nr):
    skip = f'L_l_{k}'
    a(f's_andn2_b32 s74, s76, {k}')
    a(f's_and_b32 s75, s77, {k}')
    a('s_or_b32 s74, s74, s75')
    a('s_cmp_eq_u32 s74, 0')
    a(f's_cbranch_scc0 {L(skip)}')
    shuf(k, LN_BUF[0])
    a('s_waitcnt lgkmcnt(0)')
    combine(k, LN_BUF[0])
    a.label(skip)
  next_op()

  # ---- diagonal op: groups of phase factors -------------------------------------------
  # SGPRs here: s[48:49] tables base, s[52:67] group header (lane_mask reg_mask
  # oterm_off n_oterms re(2) im(2) flags ltab_off ntab tab_shift tab_off[4]),
  # s[76:91] chunk-table entries / oterm scratch, s[92:93] group cursor, s96 counter,
  # s72 = reg_mask of the group being applied (its header registers are already
  # being refilled with the NEXT group's header during the apply phase).
  cr, ci = V2(D_C[0]), V2(D_C[1])
  ur, ui = V2(D_U[0]), V2(D_U[1])
  fr, fi = V2(D_F[0]), V2(D_F[1])
  dt = V2(D_TMP[0])

  def cmul_su(sre, sim):
    """u *= the double-precision factor held in two SGPR pairs."""
    if DT.wide:
      cmul_vv(a, ur, ui, sre, sim, dt)
    else:
      a(f'v_cvt_f32_f64 v38, {sre}')          # v34..v37 may be receiving a lane-table entry
      a(f'v_cvt_f32_f64 v39, {sim}')
      cmul_vv(a, ur, ui, 'v38', 'v39', dt)

  # Control flow (a taken branch costs a wave ~50 cycles, see the op dispatch above): the host marks a
  # group GENERAL (flags bit 2) when it has a lane table, chunk tables or outside terms; every other
  # group -- f = phi0 on the lanes that satisfy lane_mask -- runs straight through.  The apply code is
  # reached through the handler table (entries NHID.., number in flags bits 8..15: 0 = no register mask,
  # then the 1- and 2-bit masks, last = any other mask), and every apply handler ends with its own copy of
  # the group head: two jumps per group instead of eight.
  masks = [1 << b for b in range(rb)] + [(1 << b0) | (1 << b1) for b0 in range(rb) for b1 in range(b0 + 1, rb)]
  gtargets = ['L_gm0'] + [f'L_gm{m}' for m in masks] + ['L_gmx']
  nuniq = [0]

  def f_from_u():                              # f = lane_ok ? u : 1
    a(f'v_and_b32 v{V_A}, s52, %5')
    a(f'v_cmp_eq_u32 vcc, s52, v{V_A}')
    if DT.wide:
      a(f'v_mov_b32 v{V_B}, 0x3ff00000')
      a(f'v_cndmask_b32 v{D_F[0]}, 0, v{D_U[0]}, vcc')
      a(f'v_cndmask_b32 v{D_F[0] + 1}, v{V_B}, v{D_U[0] + 1}, vcc')
      a(f'v_cndmask_b32 v{D_F[1]}, 0, v{D_U[1]}, vcc')
      a(f'v_cndmask_b32 v{D_F[1] + 1}, 0, v{D_U[1] + 1}, vcc')
    else:
      a(f'v_cndmask_b32 v{D_F[0]}, 1.0, v{D_U[0]}, vcc')
      a(f'v_cndmask_b32 v{D_F[1]}, 0, v{D_U[1]}, vcc')

  def u_from_header():
    if DT.wide:
      a(f'v_mov_b32 v{D_U[0]}, s56')
      a(f'v_mov_b32 v{D_U[0] + 1}, s57')         # u = phi0 (wave-uniform value held in VGPRs)
      a(f'v_mov_b32 v{D_U[1]}, s58')
      a(f'v_mov_b32 v{D_U[1] + 1}, s59')
    else:
      a(f'v_cvt_f32_f64 v{D_U[0]}, s[56:57]')
      a(f'v_cvt_f32_f64 v{D_U[1]}, s[58:59]')

  def grp_dispatch():
    # the header is consumed: remember reg_mask, advance, prefetch the NEXT group's header into the same
    # SGPRs (one past the last group is readable memory: oterms / tables follow) and jump to the apply code
    a('s_mov_b32 s72, s53')
    a('s_mov_b32 s73, s67')                      # tab_off[3]: the bit factors of a DG_BITFAC group
    a('s_bfe_u32 s74, s60, 0x80008')
    a('s_add_u32 s92, s92, 64')
    a('s_addc_u32 s93, s93, 0')
    a('s_add_u32 s96, s96, 1')
    a('s_load_dwordx16 s[52:67], s[92:93], 0x0')
    a('s_lshl_b32 s74, s74, 2')
    a(f's_add_u32 s74, s74, {4 * NHID}')
    a('s_add_u32 s98, s24, s74')
    a('s_addc_u32 s99, s25, 0')
    a('s_setpc_b64 s[98:99]')

  def grp_body():
    a('s_waitcnt lgkmcnt(0)')
    a('s_bitcmp1_b32 s60, 2')                    # DG_GENERAL
    a(f's_cbranch_scc1 {L("L_grp_gen")}')
    u_from_header()
    f_from_u()
    grp_dispatch()

  def grp_tail():
    a('s_cmp_lt_u32 s96, s47')
    a(f's_cbranch_scc0 {L("L_diag_end")}')
    grp_body()

  a.label('L_diag')
  if DT.wide:
    a(f'v_mov_b32 v{D_C[0]}, 0')
    a(f'v_mov_b32 v{D_C[0] + 1}, 0x3ff00000')   # c = 1.0 + 0.0i
    a(f'v_mov_b32 v{D_C[1]}, 0')
    a(f'v_mov_b32 v{D_C[1] + 1}, 0')
  else:
    a(f'v_mov_b32 v{D_C[0]}, 1.0')
    a(f'v_mov_b32 v{D_C[1]}, 0')
  a('s_mov_b32 s75, 0')                        # c modified?
  a('s_cmp_eq_u32 s47, 0')
  a(f's_cbranch_scc1 {L("L_next")}')
  a('s_add_u32 s48, s38, s43')                 # tables base = groups base + rel
  a('s_addc_u32 s49, s39, 0')
  a('s_lshl_b32 s74, s50, 6')                  # group_off * sizeof(DGroup)=64
  a('s_add_u32 s92, s38, s74')
  a('s_addc_u32 s93, s39, 0')
  a('s_mov_b32 s96, 0')
  a('s_load_dwordx16 s[52:67], s[92:93], 0x0')
  grp_body()

  # general group: lane table, chunk tables, outside terms
  a.label('L_grp_gen')
  u_from_header()
  a('s_bitcmp1_b32 s60, 0')                    # LTAB: start the 1-KiB lane-table load early
  a(f's_cbranch_scc0 {L("L_g1")}')
  a('s_lshl_b32 s74, s61, 4')
  a(f'v_lshlrev_b32 v{V_A}, 4, %5')             # 16-byte table entries, indexed by the LANE id
  a('s_bitcmp1_b32 s60, 1')                    # DG_LTAB_LDS: the kernel copied the lane tables to LDS
  a(f's_cbranch_scc0 {L("L_g0g")}')
  a('s_add_u32 s74, s74, %[ltab]')
  a(f'v_add_u32 v{V_A}, s74, v{V_A}')
  a(f'ds_read_b128 v[{D_LTAB}:{D_LTAB + 3}], v{V_A}')
  a(f's_branch {L("L_g1")}')
  a.label('L_g0g')
  a('s_add_u32 s98, s48, s74')
  a('s_addc_u32 s99, s49, 0')
  a(f'global_load_dwordx4 v[{D_LTAB}:{D_LTAB + 3}], v{V_A}, s[98:99]')
  a.label('L_g1')
  a('s_cmp_eq_u32 s62, 0')
  a(f's_cbranch_scc1 {L("L_g2")}')
  for t in range(4):                           # issue all chunk-table lookups, then one wait
    if t:
      a(f's_cmp_le_u32 s62, {t}')
      a(f's_cbranch_scc1 {L("L_g1w")}')
    a(f's_bfe_u32 s74, s63, {(8 << 16) | (8 * t)}')
    a('s_lshr_b64 s[72:73], %3, s74')
    a('s_and_b32 s72, s72, 0xff')
    a(f's_add_u32 s72, s72, s{64 + t}')
    a('s_lshl_b32 s72, s72, 4')
    a(f's_load_dwordx4 s[{76 + 4 * t}:{79 + 4 * t}], s[48:49], s72')
  a.label('L_g1w')
  a('s_waitcnt lgkmcnt(0)')
  for t in range(4):
    if t:
      a(f's_cmp_le_u32 s62, {t}')
      a(f's_cbranch_scc1 {L("L_g2")}')
    cmul_su(f's[{76 + 4 * t}:{77 + 4 * t}]', f's[{78 + 4 * t}:{79 + 4 * t}]')
  a.label('L_g2')
  a('s_cmp_eq_u32 s55, 0')
  a(f's_cbranch_scc1 {L("L_grp_f")}')
  a('s_mul_i32 s74, s54, 24')                  # oterm_off * sizeof(OTerm)=24
  a('s_add_u32 s94, s40, s74')
  a('s_addc_u32 s95, s41, 0')
  a('s_mov_b32 s97, 0')
  a.label('L_ot')
  a('s_load_dwordx2 s[84:85], s[94:95], 0x0')
  a('s_load_dwordx4 s[88:91], s[94:95], 0x8')
  a('s_waitcnt lgkmcnt(0)')
  a('s_and_b64 s[72:73], %3, s[84:85]')
  a('s_cmp_eq_u64 s[72:73], s[84:85]')
  a(f's_cbranch_scc0 {L("L_ot_n")}')
  cmul_su('s[88:89]', 's[90:91]')
  a.label('L_ot_n')
  a('s_add_u32 s94, s94, 24')
  a('s_addc_u32 s95, s95, 0')
  a('s_add_u32 s97, s97, 1')
  a('s_cmp_lt_u32 s97, s55')
  a(f's_cbranch_scc1 {L("L_ot")}')
  a.label('L_grp_f')
  a('s_bitcmp1_b32 s60, 0')
  a(f's_cbranch_scc0 {L("L_g3")}')
  a('s_waitcnt vmcnt(0) lgkmcnt(0)')           # f = ltab[lane] * u
  if DT.wide:
    lt_r, lt_i = V2(D_LTAB), V2(D_LTAB + 2)
  else:
    a(f'v_cvt_f32_f64 v38, v[{D_LTAB}:{D_LTAB + 1}]')
    a(f'v_cvt_f32_f64 v39, v[{D_LTAB + 2}:{D_LTAB + 3}]')
    lt_r, lt_i = 'v38', 'v39'
  a(MUL() + f' {fr}, {lt_r}, {ur}')
  a(MUL() + f' {fi}, {lt_r}, {ui}')
  a(FMA() + f' {fr}, -{lt_i}, {ui}, {fr}')
  a(FMA() + f' {fi}, {lt_i}, {ur}, {fi}')
  grp_dispatch()
  a.label('L_g3')
  f_from_u()
  grp_dispatch()

  # apply handlers
  a.label('L_gm0')
  cmul_vv(a, cr, ci, fr, fi, dt)               # reg_mask == 0: c *= f
  a('s_mov_b32 s75, 1')
  grp_tail()
  for m in masks:                              # 1- and 2-bit register masks: straight-line code
    a.label(f'L_gm{m}')
    slots = [k for k in range(nr) if (k & m) == m]
    for i in range(0, len(slots), 4):
      cmul_slots(a, slots[i:i + 4], fr, fi)
    grp_tail()
  a.label('L_gmx')
  for k in range(nr):                          # any other register mask
    skip = f'L_g_{k}'
    a(f's_andn2_b32 s74, s72, {k}')
    a('s_cmp_eq_u32 s74, 0')
    a(f's_cbranch_scc0 {L(skip)}')
    cmul_slots(a, [k], fr, fi)
    a.label(skip)
  grp_tail()

  # DG_BITFAC (planner.h): slots with register bit j set take f x the factors w_k of their other set bits;
  # the subsets of those bits are walked as a tree, parent factor x w_k -> child factor (kept in VGPRs for
  # the inner nodes: three levels), so every slot costs two complex products at most.
  def bitfac(j):
    others = [b for b in range(rb) if b != j][:4]
    free = [b for b in range(rb) if b != j and b not in others]
    a('s_lshl_b32 s73, s73, 4')
    a('s_load_dwordx16 s[76:91], s[48:49], s73')
    if DT.wide:
      wv = [(f's[{76 + 4 * t}:{77 + 4 * t}]', f's[{78 + 4 * t}:{79 + 4 * t}]') for t in range(4)]
      levels = [(V2(22), V2(24)), (V2(30), V2(32)), (V2(34), V2(36))]
      temps = [V2(16), V2(38)]
    else:
      # complex64: factors as (re, im) register pairs, packed arithmetic.  v18 v19 = c, v26 v27 = f.
      wv = [('v20', 'v21'), ('v22', 'v23'), ('v24', 'v25'), ('v28', 'v29')]
      levels = [('v30', 'v31'), ('v32', 'v33'), ('v34', 'v35')]
      temps = ['v[16:17]', 'v[36:37]', 'v[38:39]']

    def expand(slot):
      out = [slot]
      for b in free:
        out += [x | (1 << b) for x in out]
      return out

    def cmul_by(slots, pr, pi):
      if not DT.wide:
        f = vpair(pr)
        for i in range(0, len(slots), len(temps)):
          part = slots[i:i + len(temps)]
          for t, k in zip(temps, part):
            a(f'v_pk_mul_f32 {t}, v[{T(k)}:{T(k) + 1}], {f} op_sel_hi:[1,0]')
          for t, k in zip(temps, part):
            a(f'v_pk_fma_f32 v[{T(k)}:{T(k) + 1}], v[{T(k)}:{T(k) + 1}], {f}, {t} op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]')
        return
      for i in range(0, len(slots), len(temps)):
        part = slots[i:i + len(temps)]
        for t, k in zip(temps, part):
          a(MUL() + f' {t}, {Y(k)}, {pi}')
        for t, k in zip(temps, part):
          a(MUL() + f' {Y(k)}, {Y(k)}, {pr}')
        for t, k in zip(temps, part):
          a(FMA() + f' {Y(k)}, {X(k)}, {pi}, {Y(k)}')
        for t, k in zip(temps, part):
          a(FMA() + f' {X(k)}, {X(k)}, {pr}, -{t}')

    waited = [False]

    def need_w():
      if not waited[0]:
        a('s_waitcnt lgkmcnt(0)')
        if not DT.wide:
          for t in range(4):
            a(f'v_cvt_f32_f64 {wv[t][0]}, s[{76 + 4 * t}:{77 + 4 * t}]')
            a(f'v_cvt_f32_f64 {wv[t][1]}, s[{78 + 4 * t}:{79 + 4 * t}]')
        waited[0] = True

    def visit(F, rem, slot, level):
      last = rem[-1] if rem else None
      mine = expand(slot)
      leaf = expand(slot | (1 << others[last])) if rem else []
      cmul_by(mine + leaf, F[0], F[1])
      if rem:
        need_w()
        cmul_by(leaf, wv[last][0], wv[last][1])
      for idx, t in enumerate(rem[:-1]):
        need_w()
        C = levels[level]
        if DT.wide:
          a(MUL() + f' {C[0]}, {F[0]}, {wv[t][0]}')
          a(MUL() + f' {C[1]}, {F[0]}, {wv[t][1]}')
          a(FMA() + f' {C[0]}, -{F[1]}, {wv[t][1]}, {C[0]}')
          a(FMA() + f' {C[1]}, {F[1]}, {wv[t][0]}, {C[1]}')
        else:
          pk_cmul(a, vpair(C[0]), vpair(F[0]), vpair(wv[t][0]), temps[0])
        visit(C, rem[idx + 1:], slot | (1 << others[t]), level + 1)

    visit((fr, fi), list(range(len(others))), 1 << j, 0)

  for j in range(rb):
    a.label(f'L_gbf{j}')
    bitfac(j)
    grp_tail()

  a.label('L_diag_end')
  a('s_cmp_eq_u32 s75, 0')
  a(f's_cbranch_scc1 {L("L_next")}')
  a('s_bitcmp1_b32 s51, 0')                    # DEFER_C: the next (lane) op folds c into its matrix
  a(f's_cbranch_scc1 {L("L_next")}')
  for k in range(0, nr, 4):
    cmul_slots(a, list(range(k, min(k + 4, nr))), cr, ci)
  next_op()

  # ---- store the tile -------------------------------------------------------------------
  a.label('L_done')
  tile_io(store=True)
  prof_rec()
  prof_rec(wait_stores=True, real_at=127)
  a('s_nop 0')

  # s_branch / s_cbranch reach +-32 Ki dwords (128 KiB).  The complex64 RB=6 island is ~180 KiB of code:
  # its op sections are laid out on BOTH sides of the dispatcher (entry jumps over the first half), so
  # that every section is within reach of L_op / L_next / L_done and of the sections it jumps into.
  if len(a.lines) > 20000:
    lab = lambda nm: a.lines.index(f'{nm}_%=:')
    i_op, i_sec, i_done = lab('L_op'), lab('L_reg0'), lab('L_done')
    tops = [lab(nm) for nm in ('L_real', 'L_rrc0', 'L_lane_real', 'L_bf', 'L_bfl_e', 'L_lswap', 'L_wswap', 'L_dpp', 'L_lrd',
                               'L_lane', 'L_diag')]
    mid = (i_sec + i_done) // 2
    cut = min((i for i in tops if i >= mid), default=tops[-1])
    ends = ('s_branch ', 's_setpc_b64 ')
    assert a.lines[cut - 1].startswith(ends), 'the section before the cut must not fall through'
    assert a.lines[i_sec - 1].startswith(ends) and i_op < i_sec < cut < i_done
    a.lines = ([f's_branch {L("L_entry")}'] + a.lines[i_sec:cut] + [f'{L("L_entry")}:'] + a.lines[:i_sec] +
               a.lines[cut:])
  clob = ([f'v{i}' for i in range(TEMP_LO, T0 + 2 * W() * nr)] + [f's{i}' for i in range(16, 28)] + [f's{i}' for i in range(36, 100)] +
          (['s28', 's29', 's30', 's31', 's101'] if prof else []) + ['vcc', 'scc', 'memory'])
  names = {'0': 'blo', '1': 'bhi', '2': 'prm', '3': 'tidx', '4': 'voff', '5': 'lane', '6': 'itlo', '7': 'ithi',
           '8': 'wave', '9': 'lds'}
  lines = [re.sub(r'%(\d)(?!\d)', lambda m: '%[' + names[m.group(1)] + ']', ln) for ln in a.lines]
  body = '\n'.join(f'    "{ln}\\n\\t"' for ln in lines)
  cl = ', '.join(f'"{c}"' for c in clob)
  return (f'// GENERATED by tools/gen_sweep_asm.py (RB={rb}, {"complex128" if DT.wide else "complex64"}) -- do not edit.\n'
          f'asm volatile(\n{body}\n'
          '    : [tidx] "+s"(tile_idx), [itlo] "+v"(it_lo), [ithi] "+v"(it_hi)\n'
          '    : [blo] "s"(base_lo), [bhi] "s"(base_hi), [prm] "s"(prm), [voff] "v"(voff), [lane] "v"(lane_u),\n'
          '      [wave] "s"(wave_s), [lds] "s"(lds_base), [ltab] "s"(lds_ltab),\n'
          '      [sblo] "s"(sbase_lo), [sbhi] "s"(sbase_hi), [svoff] "v"(svoff)' + (', [prow] "s"(prof_row)' if prof else '') + '\n'
          f'    : {cl});\n')


def main():
  out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'qcc_amd', 'csrc')
  if PROF:
    path = os.path.join(out, 'sweep_island_prof_rb5.inc')
    with open(path, 'w') as f:
      f.write(gen(5, True, prof=True))
    print('wrote', path)
    return 0
  for wide in (True, False):
    for rb in (2, 3, 4, 5) + (() if wide else (6,)):     # complex64: 64 amplitudes per lane fit the same 128 VGPRs
      path = os.path.join(out, f'sweep_island_rb{rb}.inc' if wide else f'sweep_island_f32_rb{rb}.inc')
      with open(path, 'w') as f:
        f.write(gen(rb, wide))
      print('wrote', path, sum(1 for _ in open(path)), 'lines')


if __name__ == '__main__':
  sys.exit(main())
