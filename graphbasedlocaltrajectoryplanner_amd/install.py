"""
Drop-in installation of the MI355X backend behind the reference's own Python surface.

The reference has no plugin / FFI API; it resolves both hot-path seams through fully qualified module attributes at call
time:

  seam (1)  graph_ltpl.online_graph.src.main_online_path_gen.main_online_path_gen     (caller OTH.py:416-427)
  seam (2)  graph_ltpl.online_graph.src.VpForwardBackward.VpForwardBackward           (caller OTH.py:137-144)
  next row  graph_ltpl.data_objects.ObjectListInterface.ObjectListInterface.process_object_list  (caller Graph_LTPL.py:322)

``install()`` assigns this package's mirrors to those two attributes and wraps ``OnlineTrajectoryHandler.__init__`` so
that the lattice held by the reference's ``GraphBase`` is exported (``Lattice.from_graph_base``) and uploaded to HBM
(``ltpl_create``) while ``Graph_LTPL.graph_init()`` runs, not during the first tick. No file of the reference is
edited; ``Graph_LTPL.calc_paths()`` / ``calc_vel_profile()`` and the example scripts keep working unmodified.

There is no CPU fallback: the default backend factory is ``_capi.HipBackend`` and raises when the library or a HIP
device is missing. Tests in the build container (no GPU) pass an explicit ``backend_factory`` to exercise the HOST
logic of the mirrors against the real reference.
"""
import logging

from . import _capi
from .lattice import Lattice
from .path_gen import OnlinePathGenerator
from .object_ingest import make_process_object_list
from .vp_forward_backward import VpForwardBackward


class Session(object):
    """State shared by the two patched seams of one ``graph_ltpl`` module: one backend per GraphBase instance."""

    def __init__(self, backend_factory=None, device=-1, clock=None):
        self.backend_factory = backend_factory
        self.device = device
        self.clock = clock        # object with .time(); None = the time module (tests pass the reference's fake clock)
        self._by_gb = {}          # id(graph_base) -> (graph_base, Lattice, backend, OnlinePathGenerator)
        self.current = None       # most recently bound entry (what a VpForwardBackward constructed next will use)
        self.originals = {}

    def bind(self, graph_base):
        key = id(graph_base)
        ent = self._by_gb.get(key)
        if ent is None or ent[0] is not graph_base:
            lat = Lattice.from_graph_base(graph_base)
            if self.backend_factory is not None:
                backend = self.backend_factory(lat)
            else:
                backend = _capi.HipBackend(lat, device=self.device)
            ent = (graph_base, lat, backend, OnlinePathGenerator(lat, backend))
            self._by_gb[key] = ent
            logging.getLogger("local_trajectory_logger").info(
                "ltpl-hip: lattice uploaded (%d layers, %d nodes, %d edges, %d samples)"
                % (lat.num_layers, lat.num_nodes, lat.num_edges, lat.num_samples))
        self.current = ent
        return ent

    # seam (1): same signature as main_online_path_gen.py:11-21
    def main_online_path_gen(self, graph_base, start_node, obj_veh, obj_zone, action_sets=True, last_action_id=None,
                             max_solutions=1, const_path_seg=None, pos_est=None, last_solution_nodes=None,
                             w_last_edges=()):
        gen = self.bind(graph_base)[3]
        return gen(graph_base=graph_base, start_node=start_node, obj_veh=obj_veh, obj_zone=obj_zone,
                   action_sets=action_sets, last_action_id=last_action_id, max_solutions=max_solutions,
                   const_path_seg=const_path_seg, pos_est=pos_est, last_solution_nodes=last_solution_nodes,
                   w_last_edges=w_last_edges)


def install(graph_ltpl, backend_factory=None, device=-1, mode="seams", clock=None) -> Session:
    """
    Patch an already imported ``graph_ltpl`` package (the unmodified reference). Returns the Session; ``uninstall(session)``
    restores the reference's own implementations.

    mode="seams"    seam (1) ``main_online_path_gen`` and seam (2) ``VpForwardBackward`` are replaced; the reference's own
                    ``OnlineTrajectoryHandler`` keeps the iterative memory in Python
    mode="planner"  additionally the class ``OnlineTrajectoryHandler`` itself (constructed at Graph_LTPL.py:221-227) is replaced
                    by ``oth_adapter.PlannerOnlineTrajectoryHandler``: the state machine runs in C++ behind ``ltpl_planner_*``
                     -- a tick is a handful of C calls
    """
    if mode not in ("seams", "planner"):
        raise ValueError("install(): mode must be 'seams' or 'planner'")
    session = Session(backend_factory=backend_factory, device=device, clock=clock)
    mopg_mod = graph_ltpl.online_graph.src.main_online_path_gen
    vp_mod = graph_ltpl.online_graph.src.VpForwardBackward
    oth_cls = graph_ltpl.online_graph.src.OnlineTrajectoryHandler.OnlineTrajectoryHandler
    session.originals = {"path": (mopg_mod, "main_online_path_gen", mopg_mod.main_online_path_gen),
                         "vp": (vp_mod, "VpForwardBackward", vp_mod.VpForwardBackward),
                         "oth_init": (oth_cls, "__init__", oth_cls.__init__)}

    mopg_mod.main_online_path_gen = session.main_online_path_gen

    class BoundVpForwardBackward(VpForwardBackward):
        """seam (2) bound to the session's backend at construction time (OTH.py:137-144)."""

        def __init__(self, dyn_model_exp, drag_coeff, m_veh, len_veh, follow_control_type, follow_control_params,
                     glob_rl):
            if session.current is None:
                raise _capi.BackendError("VpForwardBackward constructed before any GraphBase was bound")
            VpForwardBackward.__init__(self, dyn_model_exp=dyn_model_exp, drag_coeff=drag_coeff, m_veh=m_veh,
                                       len_veh=len_veh, follow_control_type=follow_control_type,
                                       follow_control_params=follow_control_params, glob_rl=glob_rl,
                                       backend=session.current[2])

    vp_mod.VpForwardBackward = BoundVpForwardBackward

    orig_init = oth_cls.__init__

    def oth_init(self, graph_base, *args, **kwargs):
        session.bind(graph_base)                      # export + upload before the velocity planner is constructed
        return orig_init(self, graph_base, *args, **kwargs)

    oth_cls.__init__ = oth_init

    # next row in front of seam (1): object ingestion (ObjectListInterface.process_object_list)
    oli_mod = graph_ltpl.data_objects.ObjectListInterface
    oli_cls = oli_mod.ObjectListInterface
    session.originals["oli_process"] = (oli_cls, "process_object_list", oli_cls.process_object_list)
    oli_cls.process_object_list = make_process_object_list(oli_mod, session)

    if mode == "planner":
        from .oth_adapter import PlannerOnlineTrajectoryHandler
        oth_mod = graph_ltpl.online_graph.src.OnlineTrajectoryHandler

        class BoundPlannerOTH(PlannerOnlineTrajectoryHandler):
            pass
        BoundPlannerOTH.session = session
        session.originals["oth_cls"] = (oth_mod, "OnlineTrajectoryHandler", oth_mod.OnlineTrajectoryHandler)
        oth_mod.OnlineTrajectoryHandler = BoundPlannerOTH
    return session


def uninstall(session: Session) -> None:
    for owner, name, orig in session.originals.values():
        setattr(owner, name, orig)
    session.originals = {}
