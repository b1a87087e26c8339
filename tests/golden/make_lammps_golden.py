#!/usr/bin/env python3
"""
Golden LAMMPS potential files (build container only, like make_golden.py): runs the reference's own writer
(lammps_plugin/scripts/generate_uf3_lammps_pots.py:57-165) on the model JSONs its tests hold and stores the text it
produces as tests/golden/lammps_<model>.uf3.  The DATE field of the header lines is the only non-deterministic part;
the test masks it.
"""
import importlib.util
import os
import sys
import tempfile
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "_standins"))
sys.path.insert(0, REF)
warnings.simplefilter("ignore")

from uf3.regression import least_squares  # noqa: E402

spec = importlib.util.spec_from_file_location("gen", os.path.join(REF, "lammps_plugin", "scripts", "generate_uf3_lammps_pots.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)

for name, spacing in [("model_unary", "nk"), ("model_binary", "nk"), ("model_2and3", "uk")]:
    model = least_squares.WeightedLinearModel.from_json(os.path.join(HERE, name + ".json"))
    cs = model.bspline_config.chemical_system
    with tempfile.TemporaryDirectory() as tmp:
        gen.write_uf3_lammps_pot_files(chemical_sys=cs, model=model, knots_spacing_type=spacing, pot_dir=tmp,
                                       uf3_lammps_pot_name="pot.uf3", author="golden", lammps_units="metal")
        text = open(os.path.join(tmp, "pot.uf3")).read()
    open(os.path.join(HERE, f"lammps_{name}.uf3"), "w").write(text)
    print(name, len(text.splitlines()), "lines")
