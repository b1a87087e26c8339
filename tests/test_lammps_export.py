"""LAMMPS potential export against files written by the reference's own writer (tests/golden/make_lammps_golden.py)."""
import os
import re

import pytest

from uf3_amd.forcefield import lammps_pot
from uf3_amd.regression import least_squares as ls
from _util import GOLDEN

DATE = re.compile(r"DATE: \d{4}-\d\d-\d\d \d\d:\d\d:\d\d")


@pytest.mark.parametrize("name,spacing", [("model_unary", "nk"), ("model_binary", "nk"), ("model_2and3", "uk")])
def test_potential_file_matches_reference_writer(name, spacing, tmp_path):
    model = ls.WeightedLinearModel.from_json(os.path.join(GOLDEN, name + ".json"))
    cs = model.bspline_config.chemical_system
    lammps_pot.write_uf3_lammps_pot_files(chemical_sys=cs, model=model, knots_spacing_type=spacing,
                                          pot_dir=str(tmp_path / "pots"), uf3_lammps_pot_name="pot.uf3",
                                          author="golden", lammps_units="metal")
    got = open(tmp_path / "pots" / "pot.uf3").read()
    want = open(os.path.join(GOLDEN, f"lammps_{name}.uf3")).read()
    assert DATE.sub("DATE: *", got) == DATE.sub("DATE: *", want)


def test_bad_spacing_type_and_cli(tmp_path, capsys):
    model = ls.WeightedLinearModel.from_json(os.path.join(GOLDEN, "model_binary.json"))
    with pytest.raises(ValueError):
        lammps_pot.format_uf3_lammps_pot(model.bspline_config.chemical_system, model, "xx")
    lammps_pot.main(["-a", "me", "-u", "metal", "-m", os.path.join(GOLDEN, "model_binary.json"), "-d", str(tmp_path)])
    out = capsys.readouterr().out
    name = "".join(model.bspline_config.chemical_system.element_list) + ".uf3"
    assert os.path.exists(tmp_path / name) and "pair_style\tuf3 2 2" in out
