"""
LAMMPS ``pair_style uf3`` potential files from a fitted model (host side only).

Same text format as the reference's exporter (``lammps_plugin/scripts/generate_uf3_lammps_pots.py:57-165``): one
block per pair and per trio, each with a header line, the interaction line (``2B``/``3B``, elements, the trim
dictionaries as Python reprs, ``uk``/``nk``), cut-offs and knot counts, the knot vectors (``%.17g``), the number of
coefficients and the coefficients -- for trios the decompressed L x M x N grid, one ``n`` row per line, written
with ``str(float)`` like the reference.  ``tests/test_lammps_export.py`` compares against files written by the
reference's own function (only the DATE field differs).
"""
import argparse
import os
from datetime import datetime

from uf3_amd.regression import least_squares


def _fmt(values):
    return " ".join('{:.17g}'.format(v) for v in values)


def format_uf3_lammps_pot(chemical_sys, model, knots_spacing_type="nk", author="", lammps_units="metal", date=None):
    """The potential file as one string."""
    if knots_spacing_type not in ("uk", "nk"):
        raise ValueError(f"Supplied knot spacing type {knots_spacing_type}\n"
                         "is not a valid choice. Only uk or nk are valid types")
    basis = model.bspline_config
    date = date or datetime.now().strftime("%Y-%m-%d %H:%M:%S")
    header = f"#UF3 POT UNITS: {lammps_units} DATE: {date} AUTHOR: {author} CITATION:\n"
    sizes, starts = basis.get_interaction_partitions()[:2]
    out = []
    for pair in chemical_sys.interactions_map[2]:
        knots = basis.knots_map[pair]
        block = header
        block += f"2B {pair[0]} {pair[1]} {basis.leading_trim} {basis.trailing_trim} {knots_spacing_type}\n"
        block += f"{basis.r_max_map[pair]} {len(knots)}\n"
        block += _fmt(knots) + "\n"
        block += f"{sizes[pair]}\n"
        block += _fmt(model.coefficients[starts[pair]:starts[pair] + sizes[pair]]) + "\n"
        block += "#\n"
        out.append(block)
    if 3 in basis.interactions_map:
        solutions = least_squares.arrange_coefficients(model.coefficients, basis)
        for trio in basis.interactions_map[3]:
            knots = basis.knots_map[trio]
            r_max = basis.r_max_map[trio]
            block = header
            block += f"3B {trio[0]} {trio[1]} {trio[2]} {basis.leading_trim} {basis.trailing_trim} {knots_spacing_type}\n"
            block += f"{r_max[2]} {r_max[1]} {r_max[0]} {len(knots[2])} {len(knots[1])} {len(knots[0])}\n"
            for leg in (2, 1, 0):
                block += _fmt(knots[leg]) + "\n"
            grid = basis.decompress_3B(solutions[trio], trio)
            block += f"{grid.shape[0]} {grid.shape[1]} {grid.shape[2]}\n"
            for i in range(grid.shape[0]):
                for j in range(grid.shape[1]):
                    block += ' '.join(map(str, grid[i, j])) + "\n"
            block += "#\n"
            out.append(block)
    return "".join(out)


def write_uf3_lammps_pot_files(chemical_sys, model, knots_spacing_type, pot_dir, uf3_lammps_pot_name, author,
                               lammps_units):
    """Writes ``pot_dir/uf3_lammps_pot_name`` (overwriting), creating the directory if needed."""
    if not os.path.exists(pot_dir):
        os.mkdir(pot_dir)
    text = format_uf3_lammps_pot(chemical_sys, model, knots_spacing_type, author, lammps_units)
    with open(os.path.join(pot_dir, uf3_lammps_pot_name), "w") as fp:
        fp.write(text)
    return text


def main(argv=None):
    parser = argparse.ArgumentParser(description="Generate UF3 LAMMPS potential file")
    parser.add_argument('-a', '--author', required=True, help="Author Name Seperated by '_'")
    parser.add_argument('-u', '--units', required=True, help="LAMMPS Units")
    parser.add_argument('-m', '--model', required=True, help="UF3 Model JSON file")
    parser.add_argument('-d', '--directory', default=".", help="Directory path (default: current directory)")
    parser.add_argument('-k', '--knots_spacing_type', default="nk",
                        help="Knot spacing type, uk (uniform spacing) or nk (non-uniform spacing) (default: nk)")
    args = parser.parse_args(argv)
    model = least_squares.WeightedLinearModel.from_json(args.model)
    chemical_sys = model.bspline_config.chemical_system
    name = "".join(chemical_sys.element_list) + ".uf3"
    write_uf3_lammps_pot_files(chemical_sys=chemical_sys, model=model, knots_spacing_type=args.knots_spacing_type,
                               pot_dir=args.directory, uf3_lammps_pot_name=name, author=args.author,
                               lammps_units=args.units)
    print("\n\n***Add the following line to the lammps input script***\n\n")
    print("pair_style\tuf3 %i %i" % (model.bspline_config.degree, len(chemical_sys.element_list)))
    print("pair_coeff\t* * " + args.directory + "/" + name + " " + " ".join(chemical_sys.element_list))


if __name__ == "__main__":
    main()
