"""Experiment: per-phase cycle counters of k_eval (library built with -DUF3_PHASE_TIMING at exp/libuf3hip_phase.so):
share of wave residency spent in the candidate walk + pair splines, the list build, the triplets, the write-back."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["UF3_LIB_PATH"] = os.path.join(ROOT, "exp", "libuf3hip_phase.so")
import numpy as np
from uf3_amd import _lib, synthetic
from uf3_amd.regression import least_squares as ls
from uf3_amd.forcefield import calculator
basis = synthetic.notebook_basis(['V', 'Mo', 'W'])
atoms = synthetic.lattice_frame("bcc", (25, 25, 40), 3.165, [23, 42, 74], 4000)
model = ls.WeightedLinearModel(basis)
coeff = np.random.default_rng(11).normal(0, 0.05, basis.n_feats); coeff[basis.col_idx] = 0.0
model.coefficients = coeff
calc = calculator.UFCalculator(model, md_skin=float(os.environ.get("SKIN", "0.5")))
for _ in range(3): calc.evaluate_frames([atoms])
ctx = _lib.get_context(None)
buf = (ctypes.c_ulonglong * 16)()
ctx.lib.uf3_debug_phase(buf)
calc.evaluate_frames([atoms])
ctx.lib.uf3_debug_phase(buf)
names = {9: "candidate walk (MD route: list filter)", 10: "pair splines of the last batch", 11: "3-body list (sort, store, LDS copy)",
         14: "per-bond leg tables, knot records to LDS (TAB)", 12: "triplets", 13: "forces on the list entries -> HBM / inbox", 15: "wave sums, stores"}
tot = sum(buf[i] for i in names)
for i, n in names.items():
    print(f"{n:44s} {buf[i]:>14d} {100.0 * buf[i] / tot:5.1f}%")
