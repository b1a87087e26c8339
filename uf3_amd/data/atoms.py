"""
Minimal atomic-configuration container and extended-XYZ reader.

The featurizer, evaluator and tests only need the handful of ``ase.Atoms``
accessors listed in SURVEY section 8(b) (``get_positions``, ``get_atomic_numbers``,
``get_chemical_symbols``, ``get_cell``, ``get_pbc``/``pbc``, ``__len__``), so any
object providing them -- a real ``ase.Atoms`` or this class -- is accepted.
ASE is not a dependency (it is absent on the GPU box).
"""
import numpy as np

from uf3_amd.data.composition import chemical_symbols, symbols2numbers


class Atoms:
    def __init__(self, symbols=None, positions=None, numbers=None, cell=None, pbc=False,
                 calculator=None, info=None):
        if numbers is None:
            numbers = symbols2numbers(symbols) if symbols is not None else []
        self.numbers = np.array(numbers, dtype=np.int64).reshape(-1)
        n = len(self.numbers)
        self.positions = (np.zeros((n, 3)) if positions is None
                          else np.array(positions, dtype=np.float64).reshape(n, 3))
        self.set_cell(cell)
        self.set_pbc(pbc)
        self.calc = calculator
        self.info = dict(info or {})
        self.arrays = {}

    def set_cell(self, cell, scale_atoms=False):
        c = np.zeros((3, 3)) if cell is None else np.array(cell, dtype=np.float64)
        self.cell = np.diag(c) if c.shape == (3,) else c.reshape(3, 3)

    def set_pbc(self, pbc):
        p = np.zeros(3, dtype=bool)
        p[:] = False if pbc is None else pbc
        self._pbc = p

    pbc = property(lambda self: self._pbc, lambda self, v: self.set_pbc(v))

    def get_pbc(self):
        return self._pbc.copy()

    def get_cell(self):
        return self.cell.copy()

    def get_positions(self):
        return self.positions.copy()

    def set_positions(self, positions):
        self.positions = np.array(positions, dtype=np.float64).reshape(len(self), 3)

    def get_atomic_numbers(self):
        return self.numbers.copy()

    def get_chemical_symbols(self):
        return [chemical_symbols[z] for z in self.numbers]

    def get_volume(self):
        return abs(float(np.linalg.det(self.cell)))

    def __len__(self):
        return len(self.numbers)

    def copy(self):
        return Atoms(numbers=self.numbers.copy(), positions=self.positions.copy(),
                     cell=self.cell.copy(), pbc=self._pbc.copy(), info=self.info)

    def get_potential_energy(self, **kwargs):
        return self.calc.get_potential_energy(self)

    def get_forces(self):
        return self.calc.get_forces(self)


def read_extxyz(path):
    """Frames of an extended-XYZ file with Lattice / energy / forces (species pos forces ...)."""
    frames = []
    with open(path) as f:
        lines = f.read().splitlines()
    i = 0
    while i < len(lines) and lines[i].strip():
        n = int(lines[i])
        header = lines[i + 1]
        cell = np.array(header.split('Lattice="')[1].split('"')[0].split(), dtype=float).reshape(3, 3)
        info = {}
        if "energy=" in header:
            info["energy"] = float(header.split("energy=")[1].split()[0])
        rows = [ln.split() for ln in lines[i + 2:i + 2 + n]]
        atoms = Atoms([r[0] for r in rows], positions=[[float(x) for x in r[1:4]] for r in rows],
                      cell=cell, pbc=True, info=info)
        if len(rows[0]) >= 7:
            atoms.arrays["forces"] = np.array([[float(x) for x in r[4:7]] for r in rows])
        frames.append(atoms)
        i += 2 + n
    return frames
