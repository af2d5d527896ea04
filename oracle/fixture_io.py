"""
ORACLE / TEST INFRASTRUCTURE ONLY. Tiny (de)serialiser for nested dict / list / ndarray / scalar records into a
pickle-free ``.npz``: arrays are stored under generated keys, the tree itself as one JSON string.
"""
import json
import numpy as np


def _enc(obj, store):
    if isinstance(obj, np.ndarray) and "__pool__" in store and obj.dtype == np.float64:
        # packed form: all float64 arrays of a file live in ONE pool array (thousands of tiny zip members are slow and big)
        pool = store["__pool__"]
        off = sum(a.size for a in pool)
        pool.append(np.ascontiguousarray(obj).reshape(-1))
        return {"p": [off, list(obj.shape)]}
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
        if "p" in node:
            off, shape = node["p"]
            n = int(np.prod(shape)) if len(shape) else 1
            return store["__pool__"][off:off + n].reshape(shape).copy()
        if "d" in node:
            return {k: _dec(v, store) for k, v in node["d"]}
        if "l" in node:
            return [_dec(v, store) for v in node["l"]]
        if "f" in node:
            return float.fromhex(node["f"])
    return node


def save_records(path, records, packed=False):
    store = {"__pool__": []} if packed else {}
    tree = _enc(records, store)
    if packed:
        pool = store.pop("__pool__")
        store["__pool__"] = np.concatenate(pool) if pool else np.zeros(0)
    np.savez_compressed(path, __tree__=np.array(json.dumps(tree, separators=(",", ":"))), **store)


def load_records(path):
    with np.load(path, allow_pickle=False) as z:
        store = {k: z[k] for k in z.files if k != "__tree__"}
        tree = json.loads(str(z["__tree__"]))
    return _dec(tree, store)
