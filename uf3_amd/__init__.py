"""uf3_amd: MI355X-native hot path of UF3 (featurizer, normal equations, evaluator)."""
__version__ = "0.1.0"
