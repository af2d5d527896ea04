"""
Host-side mirror of ``ObjectListInterface.process_object_list`` (graph_ltpl/data_objects/ObjectListInterface.py:75-153) --
SURVEY.md section 8f, rank 1: the step immediately in front of the path seam on every tick.

The per-object arithmetic -- on-track test ``check_inside_bounds`` (check_inside_bounds.py:7-59: closest centre-line point,
50-point interpolation between the bracketing bound points, squared-distance comparison), the 0.2 s constant-velocity
prediction (:117-127) and radius = length / 2 (:133) -- runs for ALL objects of the tick in one launch of
``ltpl_process_objects``; object-type dispatch, the 'prediction' pass-through, ``VehObject`` construction, time stamps and
log messages stay on the host with the reference's own classes.
"""
import numpy as np

PRED_DT = 0.2          # ObjectListInterface.py:121


def make_process_object_list(oli_module, session):
    """Returns a function with the signature of ``ObjectListInterface.process_object_list`` bound to ``session``."""
    P = "_ObjectListInterface__"

    def process_object_list(self, object_list: list) -> list:
        log = getattr(self, P + "log")
        if object_list is not None:
            setattr(self, P + "last_timestamp", oli_module.time.time())
            physical = []
            for object_el in object_list:
                if object_el['type'] in oli_module.KNOWN_OBJ_TYPES:
                    if object_el['type'] == "physical":
                        physical.append(object_el)
                else:
                    log.warning("Found non-supported object of type '%s' in object list!" % object_el['type'])
            new_vehicle_objects = []
            if physical:
                if session.current is None:
                    raise RuntimeError("object ingestion called before a GraphBase was bound to the backend")
                res = session.current[2].process_objects([o['X'] for o in physical], [o['Y'] for o in physical],
                                                         [o['theta'] for o in physical], [o['v'] for o in physical],
                                                         [o['length'] for o in physical], dt=PRED_DT)
                have_bounds = getattr(self, P + "bound1") is not None and getattr(self, P + "bound2") is not None
                for k, object_el in enumerate(physical):
                    if have_bounds and not res["on_track"][k]:
                        continue                                   # objects outside the track are ignored (:101-108)
                    if 'prediction' in object_el.keys():
                        pred = object_el['prediction']
                    else:
                        pred = np.array([[res["pred_x"][k], res["pred_y"][k]]])
                    new_vehicle_objects.append(oli_module.VehObject(id_in=object_el['id'],
                                                                    pos_in=[object_el['X'], object_el['Y']],
                                                                    psi_in=object_el['theta'],
                                                                    radius_in=float(res["radius"][k]),
                                                                    vel_in=object_el['v'],
                                                                    prediction_in=pred))
            setattr(self, P + "object_vehicles", new_vehicle_objects)
        else:
            last = getattr(self, P + "last_timestamp")
            if oli_module.time.time() - last > oli_module.TIME_WARNING:
                time_str = "so far" if last == 0.0 else "in the last %.2fs" % (oli_module.time.time() - last)
                log.warning("Did not receive an object list " + time_str + "! Check coms!")
        return getattr(self, P + "object_vehicles")

    return process_object_list
