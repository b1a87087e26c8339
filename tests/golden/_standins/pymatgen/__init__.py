"""Stand-in: the reference's LAMMPS export script imports pymatgen at module level but its writer never uses it."""
