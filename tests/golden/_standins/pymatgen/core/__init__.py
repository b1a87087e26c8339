class Element:        # imported, unused by write_uf3_lammps_pot_files
    pass
