"""Object ingestion: the oracle's restatement of check_inside_bounds /
process_object_list against golden vectors produced by the UNMODIFIED reference functions (oracle/gen_golden_objects.py).
No shimmed dependency is involved in this row, so its parity is pinned by the reference itself."""
import os
import numpy as np

from helpers import GOLDEN


def load_objects_golden():
    with np.load(os.path.join(GOLDEN, "objects_bounds.npz")) as z:
        d = {k: z[k] for k in z.files}
    return d


def check_backend_against_golden(backend):
    g = load_objects_golden()
    pts, flags = g["pts"], g["flags"].astype(bool)
    n = len(pts)
    out = backend.process_objects(pts[:, 0], pts[:, 1], np.zeros(n), np.zeros(n), np.full(n, 5.0))
    assert np.array_equal(out["on_track"].astype(bool), flags)                      # bit-exact verdicts
    assert 0.3 < flags.mean() < 0.8
    for k in range(int(g["n_lists"])):
        rec = g["l%d_in" % k]
        if rec.shape[0] == 0:
            continue
        o = backend.process_objects(rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4], dt=0.2)
        keep = o["on_track"].astype(bool)
        assert list(rec[keep, 5].astype(int)) == list(g["l%d_kept" % k])             # same objects survive, same order
        pred = np.column_stack((o["pred_x"][keep], o["pred_y"][keep]))
        assert np.allclose(pred, g["l%d_pred" % k], rtol=0.0, atol=1e-11)
        assert np.array_equal(o["radius"][keep], g["l%d_radius" % k])


def test_oracle_object_ingestion_matches_reference(oracle_backend):
    check_backend_against_golden(oracle_backend)
