all_changes = ['positions', 'numbers', 'cell', 'pbc', 'initial_charges', 'initial_magmoms']


class Calculator:
    implemented_properties = []

    def __init__(self, **kwargs):
        self.results = {}
        self.atoms = None

    def calculate(self, atoms=None, properties=None, system_changes=all_changes):
        if atoms is not None:
            self.atoms = atoms.copy()

    def get_property(self, name, atoms=None, allow_calculation=True):
        self.results = {}
        self.calculate(atoms, [name], all_changes)
        return self.results[name]

    def get_potential_energy(self, atoms=None, force_consistent=False):
        return self.get_property('energy', atoms)

    def get_forces(self, atoms=None):
        return self.get_property('forces', atoms)
