"""
Model / knot JSON files: tuple keys <-> "A-B-C" strings, arrays <-> lists.

Format-compatible with the reference's ``uf3/util/json_io.py`` (:11-83) so that
models written by either side load in the other.
"""
import json
import numpy as np


def encode_interaction_map(interaction_map):
    out = {}
    for key, value in interaction_map.items():
        if isinstance(value, list) and len(value) and isinstance(value[0], np.ndarray):
            value = [v.tolist() for v in value]
        if isinstance(value, np.ndarray):
            value = value.tolist()
        elif isinstance(value, dict):
            value = encode_interaction_map(value)
        elif isinstance(value, (np.floating, np.integer, np.bool_)):
            value = value.item()
        if isinstance(key, tuple):
            key = '-'.join(str(k) for k in key)
        out[key] = value
    return out


def decode_interaction_map(formatted_map):
    out = {}
    for key, value in formatted_map.items():
        if isinstance(value, list):
            if len(value) and isinstance(value[0], list):
                value = [np.array(row) for row in value]
            else:
                value = np.array(value)
        elif isinstance(value, dict):
            value = decode_interaction_map(value)
        if '-' in key:
            parts = key.split('-')
            try:
                parts = [int(p) for p in parts]
            except ValueError:
                pass
            key = tuple(parts)
        out[key] = value
    return out


class _RowEncoder(json.JSONEncoder):
    """Numeric vectors on one line, nested containers indented."""

    def iterencode(self, o, _one_shot=False):
        return iter([self._fmt(o, 0)])

    def _fmt(self, o, level):
        pad = " " * (self.indent or 0)
        if isinstance(o, (list, tuple)):
            if all(not isinstance(x, (list, tuple, dict)) for x in o):
                return "[" + ", ".join(json.dumps(x) for x in o) + "]"
            inner = [pad * (level + 1) + self._fmt(x, level + 1) for x in o]
            return "[\n" + ",\n".join(inner) + "\n" + pad * level + "]"
        if isinstance(o, dict):
            inner = [pad * (level + 1) + json.dumps(str(k)) + ": " + self._fmt(v, level + 1)
                     for k, v in o.items()]
            return "{\n" + ",\n".join(inner) + "\n" + pad * level + "}"
        return json.dumps(o)


def dump_interaction_map(interaction_map, indent=4, filename=None, write=False):
    text = json.dumps(encode_interaction_map(interaction_map), indent=indent, cls=_RowEncoder)
    if write:
        with open(filename, 'w') as f:
            f.write(text)
        return None
    return text


def load_interaction_map(filename):
    with open(filename, "r") as f:
        return decode_interaction_map(json.load(f))
