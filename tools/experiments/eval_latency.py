"""Latency of one UFCalculator call (host entry: upload, neighbour stage, evaluation, download) on small frames."""
import json
import time

import numpy as np

from uf3_amd import synthetic
from uf3_amd.forcefield import calculator
from uf3_amd.regression import least_squares as ls


def main():
    out = {}
    basis = synthetic.notebook_basis(['W'])
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(1).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    calc = calculator.UFCalculator(model)
    for reps in ((4, 4, 4), (8, 8, 8), (16, 16, 16)):
        atoms = synthetic.lattice_frame("bcc", reps, 3.165, [74], seed=3)
        for _ in range(5):
            calc.evaluate_frames([atoms])
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            calc.evaluate_frames([atoms])
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            calc.evaluate_frames([atoms], virial=True)
        dtv = (time.perf_counter() - t0) / n
        out[f"{len(atoms)}_atoms"] = {"us_per_call": 1e6 * dt, "us_per_call_with_virial": 1e6 * dtv}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
