"""
ORACLE / TEST INFRASTRUCTURE ONLY. Tiny (de)serialiser for nested dict / list / ndarray / scalar records into a
pickle-free ``.npz``: arrays are stored under generated keys, the tree itself as one JSON string.
"""
import json
import numpy as np


def _enc(obj, store):
    if isinstance(obj, np.ndarray):
        key = "a%d" % len(store)
        store[key] = obj
        return {"@": key}
    if isinstance(obj, dict):
        return {"d": [[str(k), _enc(v, store)] for k, v in obj.items()]}
    if isinstance(obj, (list, tuple)):
        return {"l": [_enc(v, store) for v in obj]}
    if isinstance(obj, (np.bool_, bool)):
        return bool(obj)
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return {"f": float(obj).hex()}
    if isinstance(obj, float):
        return {"f": obj.hex()}
    if obj is None or isinstance(obj, (int, str)):
        return obj
    raise TypeError("cannot serialise %r" % type(obj))


def _dec(node, store):
    if isinstance(node, dict):
        if "@" in node:
            return store[node["@"]]
        if "d" in node:
            return {k: _dec(v, store) for k, v in node["d"]}
        if "l" in node:
            return [_dec(v, store) for v in node["l"]]
        if "f" in node:
            return float.fromhex(node["f"])
    return node


def save_records(path, records):
    store = {}
    tree = _enc(records, store)
    np.savez_compressed(path, __tree__=np.array(json.dumps(tree)), **store)


def load_records(path):
    with np.load(path, allow_pickle=False) as z:
        store = {k: z[k] for k in z.files if k != "__tree__"}
        tree = json.loads(str(z["__tree__"]))
    return _dec(tree, store)
