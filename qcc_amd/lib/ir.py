"""Gate log of a circuit: what ``qc`` records when it is not (only) executing.

Interface-compatible with the reference's ``src/lib/ir.py`` (``Ir`` with
``single/controlled/section/end_section/reg/add_node`` and ``Node`` with the
``is_*`` predicates and ``idx0/ctl/idx1/val/gate/name/desc`` accessors), because
``qc.qc()``, ``qc.run()``, ``qc.inverse()`` and ``qc.control_by()`` are written
against it and user code inspects ``qc.ir.gates`` / ``qc.ir.ngates``.
"""
import enum

from qcc_amd.lib import helper


class Ir:
    """Append-only list of nodes plus a table of the registers that were declared."""

    def __init__(self):
        self.gates = []    # Node objects in program order (gates and section markers)
        self.regset = []   # one (name, size, Reg) per declared register
        self.regs = []     # one (global qubit index, register name, index in register) per qubit
        self.nregs = 0     # qubits declared so far
        self._count = 0    # gates only, markers excluded

    @property
    def ngates(self):
        return self._count

    # -- recording -------------------------------------------------------------------
    def add_node(self, node):
        self._count += 1
        self.gates.append(node)

    def single(self, name, idx0, gate, val=None):
        self.add_node(Node(Op.SINGLE, name, idx0, None, gate, val))

    def controlled(self, name, idx0, idx1, gate, val=None):
        self.add_node(Node(Op.CTL, name, idx0, idx1, gate, val))

    def section(self, desc):
        self.gates.append(Node(Op.SECTION, desc, 0, 0, None, None))

    def end_section(self):
        self.gates.append(Node(Op.END_SECTION, 0, 0, 0, None, None))

    def reg(self, size, name, register):
        first = self.nregs
        self.regset.append((name, size, register))
        for offset in range(size):
            self.regs.append((first + offset, name, offset))
        self.nregs = first + size

    # -- rendering --------------------------------------------------------------------
    def __str__(self):
        lines, indent = [], 0
        for node in self.gates:
            if node.is_end_section():
                indent -= 1
            else:
                if node.is_section():
                    indent += 1
                lines.append('  ' * indent + f'{node}\n')
        return ''.join(lines)


class Node:
    """A recorded single-qubit gate, controlled gate, or section marker."""

    def __init__(self, opcode, name, idx0, idx1, gate, val):
        self._what = opcode
        self._label = name
        self._first, self._second = idx0, idx1
        self._matrix, self._angle = gate, val

    # kind predicates
    def is_single(self):
        return self._what is Op.SINGLE

    def is_ctl(self):
        return self._what is Op.CTL

    def is_gate(self):
        return self.is_single() or self.is_ctl()

    def is_section(self):
        return self._what is Op.SECTION

    def is_end_section(self):
        return self._what is Op.END_SECTION

    def to_ctl(self, ctl):
        """Promote a single-qubit node to the same gate controlled by qubit `ctl`."""
        self._first, self._second = ctl, self._first
        self._label = 'c' + self._label
        self._what = Op.CTL

    # accessors (the index ones assert the node kind, like the reference)
    def _need(self, ok, what):
        if not ok:
            raise AssertionError(f'Invalid use of {what}')

    @property
    def opcode(self):
        return self._what

    @property
    def name(self):
        return self._label if self._label else '*unk*'

    @property
    def desc(self):
        return self._label

    @property
    def gate(self):
        return self._matrix

    @property
    def val(self):
        return self._angle

    @property
    def idx0(self):
        self._need(self.is_single(), 'idx0(), must be single gate.')
        return self._first

    @property
    def ctl(self):
        self._need(self.is_ctl(), 'ctl(), must be controlled gate.')
        return self._first

    @property
    def idx1(self):
        self._need(self.is_ctl(), 'idx1(), must be controlled gate.')
        return self._second

    def __str__(self):
        if self.is_section():
            return f'|-- {self.name} ---'
        if not self.is_gate():
            return ''
        where = f'{self._first}' if self.is_single() else f'{self._first}, {self._second}'
        angle = f'({helper.pi_fractions(self._angle)})' if self._angle else ''
        return f'{self.name}({where}){angle}'


class Op(enum.Enum):
    """Node kinds."""
    UNK = 0
    SINGLE = 1
    CTL = 2
    SECTION = 3
    END_SECTION = 4
