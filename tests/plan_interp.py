"""Executes a sweep plan (qh_plan_export) with NumPy: TEST INFRASTRUCTURE.

The planner (qcc_amd/csrc/planner.h) turns a queue of gates into sweeps of ops over a tile
geometry; the GPU kernel interprets those ops on registers.  This module interprets the SAME
ops on a whole state vector in physical index space, so that `-m "not gpu"` tests can check
the planner's semantics (commutation, lazy diagonals, tables, butterfly scalars, layout
exchanges, fixed bits, shard predicates) against the oracle without a GPU.  It never runs in
the product path and shares no code with the kernel."""
import ctypes

import numpy as np

from qcc_amd import native

OP_DENSE_REG, OP_DENSE_LANE, OP_DIAG, OP_LSWAP, OP_WSWAP = 0, 1, 2, 3, 4
OPF_REAL, OPF_BFLY, OPF_LANE_DPP, OPF_SWAP_RI = 4, 8, 128, 256
OPF_ROT_P, OPF_ROT_M = 2048, 4096      # register butterfly behind the phase g[0] * (1 +- i) on its target
DG_LTAB = 1
DG_BITFAC = 8

OP_DT = np.dtype([('kind', '<u4'), ('tb', '<u4'), ('cm_reg', '<u4'), ('n_groups', '<u4'), ('cm_thread', '<u8'),
                  ('group_off', '<u4'), ('flags', '<u4'), ('g', '<f8', (8,))])
GROUP_DT = np.dtype([('lane_mask', '<u4'), ('reg_mask', '<u4'), ('oterm_off', '<u4'), ('n_oterms', '<u4'),
                     ('re', '<f8'), ('im', '<f8'), ('flags', '<u4'), ('ltab_off', '<u4'), ('ntab', '<u4'),
                     ('tab_shift', '<u4'), ('tab_off', '<u4', (4,))])
OTERM_DT = np.dtype([('mask', '<u8'), ('re', '<f8'), ('im', '<f8')])
assert OP_DT.itemsize == 96 and GROUP_DT.itemsize == 64 and OTERM_DT.itemsize == 24

# unit-entry butterfly matrices (planner.h butterfly_variant), WITHOUT their scalar
_BFLY = [np.array(m, dtype=np.complex128) for m in
         ([[1, 1], [1, -1]], [[1, -1], [1, 1]], [[1, 1], [-1, 1]], [[1, -1j], [-1j, 1]], [[1, 1j], [1j, 1]])]


def export_plan(handle):
  lib = native.load()
  need = ctypes.c_uint64()
  native.check(lib.qh_plan_export(handle, None, 0, ctypes.byref(need)))
  buf = np.zeros(need.value // 8, dtype=np.uint64)
  native.check(lib.qh_plan_export(handle, buf.ctypes.data, need.value, None))
  assert buf[0] == 0x51485033
  nsweeps, noop = int(buf[1]), int(buf[2])
  raw = buf.view(np.uint8)
  final_pos = [int(x) for x in raw[24:88]]
  pos = 88
  sweeps = []
  for _ in range(nsweeps):
    hdr = raw[pos:pos + 224].view('<i8')
    pos += 224
    sp = {'rb': int(hdr[0]), 'regpos': [int(x) for x in hdr[1:7]], 'regpos_store': [int(x) for x in hdr[7:13]],
          'lanehi': [int(x) for x in hdr[13:16]], 'nwave': int(hdr[16]), 'wavepos': [int(x) for x in hdr[17:19]],
          'fixed_ones': int(hdr[19]) & (2 ** 64 - 1), 'ntiles': int(hdr[20]), 'n_ltab': int(hdr[25]),
          'lane_low': int(hdr[26]), 'relayout': int(hdr[27]), 'final_pos': final_pos}
    sp['dest_pos'] = [int(x) for x in raw[pos:pos + 64]]
    pos += 64
    st = raw[pos:pos + 160].view('<i8')
    pos += 160
    sp['seat'], sp['seat_store'], sp['seat_dest'] = ([int(x) for x in st[:6]], [int(x) for x in st[6:12]],
                                                      [int(x) for x in st[12:18]])
    sp['wavepos_store'] = [int(x) for x in st[18:20]]
    kd = raw[pos:pos + 200].view('<i8')
    pos += 200
    sp['reg_dest'], sp['wave_dest'] = [int(x) for x in kd[:6]], [int(x) for x in kd[6:8]]
    sp['unit_runs'] = [(int(kd[9 + i]) & (2 ** 64 - 1), int(kd[17 + i])) for i in range(int(kd[8]))]
    for name, dt, count in (('ops', OP_DT, int(hdr[21])), ('groups', GROUP_DT, int(hdr[22])),
                            ('oterms', OTERM_DT, int(hdr[23])), ('tables', np.dtype('<f8'), int(hdr[24]))):
      nbytes = count * dt.itemsize
      sp[name] = raw[pos:pos + nbytes].view(dt).copy()
      pos += (nbytes + 7) // 8 * 8
    sweeps.append(sp)
  assert pos == raw.size
  return sweeps, noop


def _bit(idx, b):
  return ((idx >> np.uint64(b)) & np.uint64(1)).astype(bool)


def permute_bits(psi, nloc, dest_pos):
  """out[j] = psi[i] where bit p of i becomes bit dest_pos[p] of j (a relayout store)."""
  idx = np.arange(1 << nloc, dtype=np.uint64)
  j = np.zeros_like(idx)
  for p in range(nloc):
    j |= ((idx >> np.uint64(p)) & np.uint64(1)) << np.uint64(dest_pos[p])
  out = np.empty_like(psi)
  out[j] = psi
  return out


def run_plan(psi, sweeps, nloc, shard=0):
  """Apply the exported sweeps to the LOCAL shard `psi` (2^nloc amplitudes, physical order) in place.
  Relayout sweeps move index bits: the result is returned in the ORIGINAL bit order (the inverse of
  the plan's final_pos is applied at the end), as a caller that converts through the bit map sees it."""
  n = 1 << nloc
  idx = np.arange(n, dtype=np.uint64)
  gidx = idx | (np.uint64(shard) << np.uint64(nloc))          # global index (shard bits on top)
  for sp in sweeps:
    rb = sp['rb']
    regpos = list(sp['regpos'][:rb])
    low = sp['lane_low']
    lanepos = list(sp['seat'])          # index bit on lane bit i: the line bits and lanehi, seated by cost (planner.h choose_seats)
    assert sorted(lanepos) == sorted(list(range(low)) + list(sp['lanehi'][:6 - low])), 'seats are not the lane-resident bits'
    wavepos = list(sp['wavepos'][:sp['nwave']])
    fixed = np.uint64(sp['fixed_ones'])
    in_sweep = (idx & fixed) == fixed
    tab = sp['tables'].view(np.complex128)

    def coords():
      lane = np.zeros(n, dtype=np.uint64)
      for k, b in enumerate(lanepos):
        lane |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(k)
      slot = np.zeros(n, dtype=np.uint64)
      m = 0
      for k, b in enumerate(regpos):
        slot |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(k)
        m |= 1 << b
      for b in lanepos:
        m |= 1 << b
      tile = gidx & ~np.uint64(m)                               # what the kernel calls the tile index
      return lane, slot, tile

    for op in sp['ops']:
      kind, tb, flags = int(op['kind']), int(op['tb']), int(op['flags'])
      if kind == OP_LSWAP:
        r = int(op['cm_reg'])
        # kernel bookkeeping operands: the lane bit's index position, and both positions as a mask
        assert int(op['n_groups']) == lanepos[tb] and int(op['cm_thread']) == (1 << lanepos[tb]) | (1 << regpos[r])
        lanepos[tb], regpos[r] = regpos[r], lanepos[tb]
        continue
      if kind == OP_WSWAP:
        r = int(op['cm_reg'])
        assert int(op['cm_thread']) == (1 << wavepos[tb]) | (1 << regpos[r])
        wavepos[tb], regpos[r] = regpos[r], wavepos[tb]
        continue
      lane, slot, tile = coords()
      if kind == OP_DIAG:
        factor = np.ones(n, dtype=np.complex128)
        for g in sp['groups'][int(op['group_off']): int(op['group_off']) + int(op['n_groups'])]:
          u = np.full(n, complex(g['re'], g['im']))
          for t in range(int(g['ntab'])):
            sh = (int(g['tab_shift']) >> (8 * t)) & 0xff
            u = u * tab[int(g['tab_off'][t]) + ((tile >> np.uint64(sh)) & np.uint64(0xff)).astype(np.int64)]
          for ot in sp['oterms'][int(g['oterm_off']): int(g['oterm_off']) + int(g['n_oterms'])]:
            m = np.uint64(ot['mask'])
            u = np.where((tile & m) == m, u * complex(ot['re'], ot['im']), u)
          if int(g['flags']) & DG_LTAB:
            f = tab[int(g['ltab_off']) + lane.astype(np.int64)] * u
          else:
            lm = np.uint64(g['lane_mask'])
            f = np.where((lane & lm) == lm, u, 1.0)
          rm = np.uint64(g['reg_mask'])
          if int(g['flags']) & DG_BITFAC:
            # one register bit j; four more factors, one per register bit of the first four others
            j = int(g['reg_mask']).bit_length() - 1
            assert int(g['reg_mask']) == 1 << j
            others = [b for b in range(rb) if b != j][:4]
            w = tab[int(g['tab_off'][3]): int(g['tab_off'][3]) + 4]
            for t, b in enumerate(others):
              f = np.where((slot >> np.uint64(b)) & np.uint64(1) == 1, f * w[t], f)
          factor = np.where((slot & rm) == rm, factor * f, factor)
        psi[in_sweep] = (psi * factor.astype(psi.dtype))[in_sweep]
        continue
      # dense ops
      tgt = lanepos[tb] if kind == OP_DENSE_LANE else regpos[tb]
      # register controls: bits 0..4 of cm_reg must be one, bits 8..12 zero; thread controls:
      # (index & cm_thread) == cm_thread & ~zero-controls (which travel in n_groups / group_off)
      pos_reg, neg_reg = np.uint64(int(op['cm_reg']) & 0x3f), np.uint64((int(op['cm_reg']) >> 8) & 0x3f)
      cmt = int(op['cm_thread'])
      want = cmt & ~(int(op['n_groups']) | (int(op['group_off']) << 32))
      ok = (in_sweep & ((slot & pos_reg) == pos_reg) & ((slot & neg_reg) == np.uint64(0)) &
            ((gidx & np.uint64(cmt)) == np.uint64(want)))
      g8 = op['g']
      if flags & OPF_BFLY:
        v = (flags >> 4) & 7
        if flags & OPF_LANE_DPP:
          # new.re = own.re + b_re*q.re ; new.im = own.im + b_im*q.im ; q = partner (re/im exchanged if SWAP_RI)
          hi = _bit(idx, tgt)
          partner = psi[idx ^ np.uint64(1 << tgt)]
          q = (partner.imag + 1j * partner.real) if flags & OPF_SWAP_RI else partner
          bre = np.where(hi, g8[1], g8[0])
          bim = np.where(hi, g8[3], g8[2])
          new = (psi.real + bre * q.real) + 1j * (psi.imag + bim * q.imag)
          psi[ok] = new.astype(psi.dtype)[ok]
          continue
        m = _BFLY[v]
        if flags & (OPF_ROT_P | OPF_ROT_M):
          assert kind != OP_DENSE_LANE and int(op['cm_reg']) == 0 and int(op['cm_thread']) == 0
          m = m @ np.diag([1.0, g8[0] * (1 + 1j if flags & OPF_ROT_P else 1 - 1j)])
      else:
        m = g8.view(np.complex128).reshape(2, 2)
      lo = ok & ~_bit(idx, tgt)
      i0 = idx[lo]
      i1 = i0 | np.uint64(1 << tgt)
      a, b = psi[i0].astype(np.complex128), psi[i1].astype(np.complex128)
      psi[i0] = (m[0, 0] * a + m[0, 1] * b).astype(psi.dtype)
      psi[i1] = (m[1, 0] * a + m[1, 1] * b).astype(psi.dtype)
    assert regpos == list(sp['regpos_store'][:rb]), 'store layout disagrees with the exported one'
    if sp['relayout']:
      # the tile goes to the second buffer contiguously: lane bits on 0..5, register bit k on 6+k,
      # wave bit j on 6+rb+j, everything else above in order -- whatever the bits are now
      assert fixed == 0
      assert lanepos == list(sp['seat_store']) and wavepos == list(sp['wavepos_store'][:sp['nwave']])
      assert sorted(sp['seat_dest']) == list(range(6)), 'the lanes of the tile are not stored on positions 0..5'
      for k, b in enumerate(lanepos):
        assert sp['dest_pos'][b] == sp['seat_dest'][k], 'store lane map disagrees with dest_pos'
      assert sorted(sp['dest_pos'][b] for b in regpos + wavepos) == list(range(6, 6 + len(regpos + wavepos))), \
          'the tile is not stored contiguously'
      assert sorted(sp['dest_pos'][:nloc]) == list(range(nloc)), 'dest_pos is not a permutation'
      # (the other index bits may go anywhere above the tile; the kernel moves them in at most 8 runs)
      # ... and what the kernel is handed says the same as dest_pos: slot / wave offsets and the runs of
      # unit-index bits, applied to every unit number
      assert [sp['dest_pos'][b] for b in regpos] == sp['reg_dest'][:rb]
      assert [sp['dest_pos'][b] for b in wavepos] == sp['wave_dest'][:len(wavepos)]
      tile_bits = set(lanepos + regpos + wavepos)
      outside = [b for b in range(nloc) if b not in tile_bits]
      tb = len(tile_bits)
      assert 1 <= len(sp['unit_runs']) <= 8
      units = np.arange(1 << len(outside), dtype=np.uint64)
      moved = np.zeros_like(units)
      for mask, shift in sp['unit_runs']:
        part = units & np.uint64(mask)
        moved |= (part << np.uint64(shift)) if shift >= 0 else (part >> np.uint64(-shift))
      want_u = np.zeros_like(units)
      for i, b in enumerate(outside):
        want_u |= ((units >> np.uint64(i)) & np.uint64(1)) << np.uint64(sp['dest_pos'][b] - tb)
      assert np.array_equal(moved, want_u), 'unit-index runs disagree with dest_pos'
      psi[:] = permute_bits(psi, nloc, sp['dest_pos'])
      continue
    # in place: the tile is stored with `regpos_store`; lane exchanges must have been undone
    assert lanepos == list(sp['seat']) == list(sp['seat_store']), 'lane layout not restored before the store'
    assert sorted(regpos + wavepos) == sorted(list(sp['regpos'][:rb]) + list(sp['wavepos'][:sp['nwave']]))
  if sweeps and any(sp['relayout'] for sp in sweeps):
    fp = sweeps[0]['final_pos']
    inv = [0] * nloc
    for p_ in range(nloc):
      inv[fp[p_]] = p_
    psi[:] = permute_bits(psi, nloc, inv)
  return psi
