"""Minimal circuit IR (API mirror of /root/reference/src/lib/ir.py)."""
import enum

from qcc_amd.lib import helper


class Op(enum.Enum):
    UNK = 0
    SINGLE = 1
    CTL = 2
    SECTION = 3
    END_SECTION = 4


class Node:
    """One recorded gate (or section marker)."""

    __slots__ = ('_opcode', '_name', '_idx0', '_idx1', '_gate', '_val')

    def __init__(self, opcode, name, idx0, idx1, gate, val):
        self._opcode, self._name = opcode, name
        self._idx0, self._idx1 = idx0, idx1
        self._gate, self._val = gate, val

    def __str__(self):
        if self.is_section():
            return f'|-- {self.name} ---'
        if self.is_single():
            text = f'{self.name}({self.idx0})'
        elif self.is_ctl():
            text = f'{self.name}({self.ctl}, {self.idx1})'
        else:
            text = ''
        if self._val:
            text += f'({helper.pi_fractions(self.val)})'
        return text

    def to_ctl(self, ctl):
        """Turn a single-qubit node into the same gate controlled by `ctl`."""
        self._opcode = Op.CTL
        self._idx1, self._idx0 = self._idx0, ctl
        self._name = 'c' + self._name

    def is_single(self):
        return self._opcode == Op.SINGLE

    def is_ctl(self):
        return self._opcode == Op.CTL

    def is_gate(self):
        return self._opcode in (Op.SINGLE, Op.CTL)

    def is_section(self):
        return self._opcode == Op.SECTION

    def is_end_section(self):
        return self._opcode == Op.END_SECTION

    opcode = property(lambda self: self._opcode)
    name = property(lambda self: self._name or '*unk*')
    desc = property(lambda self: self._name)
    val = property(lambda self: self._val)
    gate = property(lambda self: self._gate)

    @property
    def idx0(self):
        if not self.is_single():
            raise AssertionError('Invalid use of idx0(), must be single gate.')
        return self._idx0

    @property
    def ctl(self):
        if not self.is_ctl():
            raise AssertionError('Invalid use of ctl(), must be controlled gate.')
        return self._idx0

    @property
    def idx1(self):
        if not self.is_ctl():
            raise AssertionError('Invalid use of idx1(), must be controlled gate.')
        return self._idx1


class Ir:
    """Ordered gate list plus the register table."""

    def __init__(self):
        self.gates = []
        self.regs = []     # (global index, name, index within register)
        self.regset = []   # (name, size, Reg)
        self.nregs = 0
        self._ngates = 0

    def __str__(self):
        depth, out = 0, []
        for node in self.gates:
            if node.is_end_section():
                depth -= 1
                continue
            if node.is_section():
                depth += 1
            out.append('  ' * depth + str(node) + '\n')
        return ''.join(out)

    def reg(self, size, name, register):
        self.regset.append((name, size, register))
        self.regs.extend((self.nregs + i, name, i) for i in range(size))
        self.nregs += size

    def add_node(self, node):
        self.gates.append(node)
        self._ngates += 1

    def single(self, name, idx0, gate, val=None):
        self.add_node(Node(Op.SINGLE, name, idx0, None, gate, val))

    def controlled(self, name, idx0, idx1, gate, val=None):
        self.add_node(Node(Op.CTL, name, idx0, idx1, gate, val))

    def section(self, desc):
        self.gates.append(Node(Op.SECTION, desc, 0, 0, None, None))

    def end_section(self):
        self.gates.append(Node(Op.END_SECTION, 0, 0, 0, None, None))

    @property
    def ngates(self):
        return self._ngates
