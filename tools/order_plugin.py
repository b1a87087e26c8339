"""pytest plugin for order-dependence hunts on the GPU box (contexts are shared per device and remember capacities / lists):
    PYTHONPATH=tools UF3_TEST_ORDER=reverse python -m pytest -p order_plugin tests -m gpu -q
    PYTHONPATH=tools UF3_TEST_ORDER=shuffle:7 ...      (seeded shuffle)"""
import os, random


def pytest_collection_modifyitems(session, config, items):
    how = os.environ.get("UF3_TEST_ORDER", "")
    if how == "reverse":
        items.reverse()
    elif how.startswith("shuffle"):
        seed = int(how.split(":")[1]) if ":" in how else 0
        random.Random(seed).shuffle(items)
