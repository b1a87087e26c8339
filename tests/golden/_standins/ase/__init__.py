"""Bare-bones ASE stand-in (golden capture only): an Atoms container."""
import re
import sys
import types
import importlib.abc
import importlib.machinery
import numpy as np
from ase import symbols as _sym


def _parse_formula(f):
    out = []
    for el, n in re.findall(r"([A-Z][a-z]?)(\d*)", f):
        out.extend([el] * (int(n) if n else 1))
    return out


class Atoms:
    def __init__(self, symbols=None, positions=None, numbers=None, pbc=None, cell=None,
                 calculator=None, **kw):
        if numbers is not None:
            z = list(numbers)
        elif isinstance(symbols, str):
            z = _sym.symbols2numbers(_parse_formula(symbols))
        elif symbols is None:
            z = []
        else:
            z = [s if isinstance(s, (int, np.integer)) else _sym.atomic_numbers[s] for s in symbols]
        self.numbers = np.array(z, dtype=int)
        n = len(self.numbers)
        self.positions = (np.zeros((n, 3)) if positions is None
                          else np.array(positions, dtype=float).reshape(n, 3))
        self.set_cell(cell)
        self.set_pbc(pbc)
        self.calc = calculator
        self.info = {}
        self.arrays = {}

    def set_cell(self, cell, scale_atoms=False):
        if cell is None:
            c = np.zeros((3, 3))
        else:
            c = np.array(cell, dtype=float)
            if c.shape == (3,):
                c = np.diag(c)
        self.cell = c

    def set_pbc(self, pbc):
        if pbc is None:
            pbc = False
        p = np.zeros(3, dtype=bool)
        p[:] = pbc
        self._pbc = p

    @property
    def pbc(self):
        return self._pbc

    @pbc.setter
    def pbc(self, v):
        self.set_pbc(v)

    def get_pbc(self):
        return self._pbc.copy()

    def get_cell(self):
        return self.cell.copy()

    def get_positions(self):
        return self.positions.copy()

    def set_positions(self, p):
        self.positions = np.array(p, dtype=float).reshape(len(self), 3)

    def get_atomic_numbers(self):
        return self.numbers.copy()

    def get_chemical_symbols(self):
        return [_sym.chemical_symbols[z] for z in self.numbers]

    def get_volume(self):
        return abs(np.linalg.det(self.cell))

    def __len__(self):
        return len(self.numbers)

    def copy(self):
        a = Atoms(numbers=self.numbers.copy(), positions=self.positions.copy(),
                  pbc=self._pbc.copy(), cell=self.cell.copy())
        return a

    def __delitem__(self, idx):
        mask = np.ones(len(self), dtype=bool)
        mask[np.asarray(idx, dtype=int)] = False
        self.numbers = self.numbers[mask]
        self.positions = self.positions[mask]

    def translate(self, d):
        self.positions = self.positions + np.asarray(d)

    def set_calculator(self, calc):
        self.calc = calc

    def get_potential_energy(self, **kw):
        return self.calc.get_potential_energy(self)

    def get_forces(self):
        return self.calc.get_forces(self)


class _Lazy(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Any other ase.* submodule is imported by the reference but never used."""

    def find_spec(self, name, path, target=None):
        if name.startswith("ase.") and name not in ("ase.symbols", "ase.calculators",
                                                     "ase.calculators.calculator"):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []
        m.__getattr__ = lambda attr: type(attr, (), {})
        return m

    def exec_module(self, module):
        pass


sys.meta_path.append(_Lazy())
