"""
ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Deterministic re-statement of the reference's example driver loops (main_min_example.py:77-107,
main_std_example.py:98-135) on top of oracle/ref_env.py: fake clock advancing a fixed dt per tick, ideal-tracking ego
simulator ``vdc_dummy(iter_time=dt)`` (vdc_dummy.py:5-9), opponents from the reference's own ``ObjectlistDummy``
(objectlist_dummy.py:73-189). While the loop runs, recorders wrapped around the two drop-in seams capture every call's
inputs and outputs:

  seam (1)  graph_ltpl.online_graph.src.main_online_path_gen.main_online_path_gen   (caller OTH.py:416-427)
  seam (2)  graph_ltpl.online_graph.src.VpForwardBackward.VpForwardBackward methods (callers OTH.py:676-972)

The records are what oracle/gen_golden.py writes to tests/golden/*.npz.
"""

import copy
import os
import numpy as np

from . import ref_env

ZONE_EXAMPLE = {'sample_zone': [[64, 64, 64, 64, 64, 64, 64, 65, 65, 65, 65, 65, 65, 65, 66, 66, 66, 66, 66, 66, 66],
                                [0, 1, 2, 3, 4, 5, 6, 0, 1, 2, 3, 4, 5, 6, 0, 1, 2, 3, 4, 5, 6],
                                np.array([[-20.54, 227.56], [23.80, 186.64]]),
                                np.array([[-23.80, 224.06], [20.17, 183.60]])]}   # main_std_example.py:90-93


class SeamRecorder(object):
    """Wraps both seams of the reference in place and stores deep copies of every call."""

    def __init__(self, graph_ltpl, graph_base):
        self.gl = graph_ltpl
        self.gb = graph_base
        self.path_calls = []
        self.vel_calls = []
        self.zone_nodes = ([], [])
        self._orig = {}
        self._install()

    # ---- seam (1) -------------------------------------------------------------------------------------------------
    def _install(self):
        gl = self.gl
        mod = gl.online_graph.src.main_online_path_gen
        orig_path = mod.main_online_path_gen
        self._orig['path'] = (mod, 'main_online_path_gen', orig_path)

        # the zone node list reaches GraphBase only when it changes (gen_local_node_template.py:43,96)
        gb = self.gb
        orig_rm = gb.remove_nodes_filter
        self._orig['rm'] = (gb, 'remove_nodes_filter', orig_rm)

        def rm_wrap(layer_ids, node_ids, applied_filter="default", base=None):
            if applied_filter == "overtaking_zones":
                self.zone_nodes = ([int(x) for x in layer_ids], [int(x) for x in node_ids])
            return orig_rm(layer_ids=layer_ids, node_ids=node_ids, applied_filter=applied_filter, base=base)
        gb.remove_nodes_filter = rm_wrap

        def path_wrap(graph_base, start_node, obj_veh, obj_zone, action_sets=True, last_action_id=None,
                      max_solutions=1, const_path_seg=None, pos_est=None, last_solution_nodes=None, w_last_edges=()):
            rec = {
                'start_node': [int(start_node[0]), int(start_node[1])],
                'obj_pos': np.array([v.get_pos() for v in obj_veh], dtype=float).reshape(-1, 2),
                'obj_radius': np.array([v.get_radius() for v in obj_veh], dtype=float),
                'obj_vel': np.array([v.get_vel() for v in obj_veh], dtype=float),
                'obj_pred': [np.array(v.get_prediction(), dtype=float).reshape(-1, 2) for v in obj_veh],
                'action_sets': bool(action_sets),
                'last_action_id': last_action_id,
                'const_path_seg': None if const_path_seg is None else np.array(const_path_seg, dtype=float),
                'pos_est': None if pos_est is None else np.array(pos_est, dtype=float).reshape(-1),
                'last_solution_nodes': copy.deepcopy(last_solution_nodes),
                'w_last_edges': list(w_last_edges),
            }
            out = orig_path(graph_base=graph_base, start_node=start_node, obj_veh=obj_veh, obj_zone=obj_zone,
                            action_sets=action_sets, last_action_id=last_action_id, max_solutions=max_solutions,
                            const_path_seg=const_path_seg, pos_est=pos_est, last_solution_nodes=last_solution_nodes,
                            w_last_edges=w_last_edges)
            rec['zone_layers'], rec['zone_nodes'] = copy.deepcopy(self.zone_nodes)
            nodes, node_idx, coeff, path_param, red_len, closest = out
            rec['out'] = {
                'keys': list(nodes.keys()),
                'nodes': {k: [[int(a), int(b)] for a, b in nodes[k][0]] for k in nodes},
                'node_idx': {k: [int(i) for i in node_idx[k][0]] for k in nodes},
                'coeff': {k: np.array(coeff[k][0], dtype=float) for k in nodes},
                'path_param': {k: np.array(path_param[k][0], dtype=float) for k in nodes},
                'red_len': {k: bool(red_len[k][0]) for k in nodes},
                'closest_obj_index': None if closest is None else int(closest),
            }
            self.path_calls.append(rec)
            return out
        mod.main_online_path_gen = path_wrap

        # ---- seam (2) ---------------------------------------------------------------------------------------------
        vp_mod = gl.online_graph.src.VpForwardBackward
        cls = vp_mod.VpForwardBackward
        rec_list = self.vel_calls
        P = '_VpForwardBackward__'

        def state_of(obj):
            return {'vel_max': float(getattr(obj, P + 'vel_max')),
                    'gg_scale': float(getattr(obj, P + 'gg_scale')),
                    'old_gg_scale': float(getattr(obj, P + 'old_gg_scale')),
                    'ax_max_machines': np.array(getattr(obj, P + 'ax_max_machines'), dtype=float)}

        def wrap_method(name):
            orig = getattr(cls, name)
            self._orig['vp_' + name] = (cls, name, orig)

            def wrapped(obj, **kwargs):
                rec = {'method': name, 'state': state_of(obj),
                       'args': {k: (np.array(v, dtype=float) if isinstance(v, (np.ndarray, list, tuple))
                                    else float(v)) for k, v in kwargs.items()}}
                out = orig(obj, **kwargs)
                if isinstance(out, tuple):
                    rec['out'] = [np.array(o, dtype=float) if isinstance(o, (np.ndarray, list)) else o for o in out]
                else:
                    rec['out'] = np.array(out, dtype=float)
                rec_list.append(rec)
                return out
            setattr(cls, name, wrapped)

        for name in ('check_brake_prefix', 'calc_vel_profile', 'calc_vel_profile_follow', 'calc_vel_brake_em'):
            wrap_method(name)

    def uninstall(self):
        for owner, name, orig in self._orig.values():
            setattr(owner, name, orig)
        self._orig = {}


def make_planner(cache_dir, clock=None, track="monteblanco", online_overrides=None, offline_overrides=None):
    """Graph_LTPL instance (reference facade) initialised like the example scripts, visualisation / logging off.
    ``online_overrides``: {(section, key): value} written into a copy of the reference's online parameter file."""
    graph_ltpl, clock = ref_env.load_reference(clock)
    path_dict = ref_env.default_path_dict(cache_dir, track)
    if online_overrides:
        import configparser
        cp = configparser.ConfigParser()
        cp.optionxform = str
        cp.read(path_dict['ltpl_online_param_path'])
        for (sec, key), val in online_overrides.items():
            cp.set(sec, key, str(val))
        mod = os.path.join(cache_dir, "ltpl_config_online_mod.ini")
        with open(mod, "w") as fh:
            cp.write(fh)
        path_dict['ltpl_online_param_path'] = mod
    if offline_overrides:
        # a modified copy of the OFFLINE parameter file: its md5 keys the graph cache (main_offline_callback.py:57-68), so the
        # variant gets its own pickle next to the stock one
        import configparser
        cp = configparser.ConfigParser()
        cp.optionxform = str
        cp.read(path_dict['ltpl_offline_param_path'])
        tag = "_".join("%s%s" % (k, v) for (_, k), v in sorted(offline_overrides.items()))
        for (sec, key), val in offline_overrides.items():
            cp.set(sec, key, str(val))
        mod = os.path.join(cache_dir, "ltpl_config_offline_%s.ini" % tag)
        with open(mod, "w") as fh:
            cp.write(fh)
        path_dict['ltpl_offline_param_path'] = mod
        path_dict['graph_store_path'] = os.path.join(cache_dir, "stored_graph_%s_%s.pckl" % (track, tag))
    ltpl_obj = graph_ltpl.Graph_LTPL.Graph_LTPL(path_dict=path_dict, visual_mode=False, log_to_file=False)
    ltpl_obj.graph_init()
    graph_base = ltpl_obj._Graph_LTPL__graph_base
    return graph_ltpl, clock, ltpl_obj, graph_base, path_dict


def set_start(graph_ltpl, ltpl_obj, path_dict):
    refline = graph_ltpl.imp_global_traj.src.import_globtraj_csv.\
        import_globtraj_csv(import_path=path_dict['globtraj_input_path'])[0]
    pos_est = refline[0, :]
    heading_est = np.arctan2(np.diff(refline[0:2, 1]), np.diff(refline[0:2, 0])) - np.pi / 2
    ltpl_obj.set_startpos(pos_est=pos_est, heading_est=heading_est)
    return pos_est, 0.0


def opponents_c2(graph_ltpl, n_opp=8):
    """C2 opponent set: race-line followers s0_k = 250 + 280 k, vel_scale_k = 0.30 + 0.05 (k mod 4)."""
    Dummy = graph_ltpl.testing_tools.src.objectlist_dummy.ObjectlistDummy
    return [Dummy(dynamic=True, vel_scale=0.30 + 0.05 * (k % 4), s0=250.0 + 280.0 * k) for k in range(n_opp)]


def get_objects(dummies):
    obj_list = []
    for k, d in enumerate(dummies):
        for o in d.get_objectlist():
            o = dict(o)
            o['id'] = k + 1
            obj_list.append(o)
    return obj_list


def run_loop(graph_ltpl, clock, ltpl_obj, path_dict, n_ticks, dt=0.05, dummies=None, zones=None,
             action_pref=("right", "left", "straight", "follow"), on_tick=None, extra_objects=None, vel_kwargs=None):
    """The example drivers' online loop with a fixed time step. Returns per-tick exported trajectory sets.
    ``extra_objects(tick)`` may return further object-list dicts (objectlist_dummy.py:171-181 format) for that tick;
    ``vel_kwargs(tick)`` further keyword arguments of ``Graph_LTPL.calc_vel_profile`` (vel_max, gg_scale, ...)."""
    pos_est, vel_est = set_start(graph_ltpl, ltpl_obj, path_dict)
    traj_set = {'straight': None}
    exported = []
    for tick in range(n_ticks):
        clock.advance(dt)
        sel_action = None
        for sel_action in action_pref:
            if sel_action in traj_set.keys():
                break
        obj_list = get_objects(dummies) if dummies is not None else []
        if extra_objects is not None:
            obj_list = obj_list + list(extra_objects(tick))
        tick_paths = ltpl_obj.calc_paths(prev_action_id=sel_action, object_list=obj_list, blocked_zones=zones)
        if traj_set[sel_action] is not None:
            pos_est, vel_est = graph_ltpl.testing_tools.src.vdc_dummy.vdc_dummy(
                pos_est=pos_est,
                last_s_course=(traj_set[sel_action][0][:, 0]),
                last_path=(traj_set[sel_action][0][:, 1:3]),
                last_vel_course=(traj_set[sel_action][0][:, 5]),
                iter_time=dt)
        kw = {}
        if vel_kwargs is not None:
            # ``vel_kwargs(tick)`` or ``vel_kwargs(tick, path_dict)`` -- the latter sees the paths of this tick (Graph_LTPL.calc_paths's
            # first return value), which a location dependent ``local_gg`` dict has to match row for row (OTH.py:641-646)
            import inspect
            kw = vel_kwargs(tick, tick_paths) if len(inspect.signature(vel_kwargs).parameters) >= 2 else vel_kwargs(tick)
        traj_set, traj_id, _ = ltpl_obj.calc_vel_profile(pos_est=pos_est, vel_est=vel_est, **kw)
        exported.append({'sel_action': sel_action, 'pos_est': np.array(pos_est, dtype=float),
                         'vel_est': float(vel_est),
                         'traj': {k: np.array(v[0], dtype=float) for k, v in traj_set.items()},
                         'traj_id': dict(traj_id)})
        if on_tick is not None:
            on_tick(tick, exported[-1])
    return exported


class TickRecorder(object):
    """
    Records one dict per planning tick at the level of ``OnlineTrajectoryHandler`` (rows H1, H2, V0 of SURVEY.md §8a):
    inputs of ``update_objects`` / ``calc_paths`` (OTH.py:272-516), the outputs of ``get_ref_idx`` (OTH.py:518-601) and
    inputs / outputs of ``calc_vel_profile`` (OTH.py:603-1040), together with the fake-clock time the reference saw.
    Install AFTER a SeamRecorder (if both are used) so that the zone filter content is visible; uninstall restores the
    reference's methods.
    """
    P = '_OnlineTrajectoryHandler__'

    def __init__(self, graph_ltpl, clock, seam_recorder=None):
        self.gl, self.clock, self.seam = graph_ltpl, clock, seam_recorder
        self.ticks = []
        self.start = None
        self._cur = None
        self._orig = {}
        cls = graph_ltpl.online_graph.src.OnlineTrajectoryHandler.OnlineTrajectoryHandler
        P = self.P
        rec = self

        def wrap(name, fn):
            self._orig[name] = getattr(cls, name)
            setattr(cls, name, fn)

        o_update, o_paths, o_ref, o_vel = cls.update_objects, cls.calc_paths, cls.get_ref_idx, cls.calc_vel_profile

        def update_objects(oth, obj_veh, obj_zone):
            rec._cur = {'obj_pos': np.array([v.get_pos() for v in obj_veh], dtype=float).reshape(-1, 2),
                        'obj_radius': np.array([v.get_radius() for v in obj_veh], dtype=float),
                        'obj_vel': np.array([v.get_vel() for v in obj_veh], dtype=float),
                        'obj_pred': [np.array(v.get_prediction(), dtype=float).reshape(-1, 2) for v in obj_veh]}
            return o_update(oth, obj_veh=obj_veh, obj_zone=obj_zone)

        def calc_paths(oth, action_id_sel, idx_sel_traj):
            cur = rec._cur if rec._cur is not None else {}
            cur.update({'t': float(rec.clock.now), 'action_id_sel': action_id_sel, 'idx_sel_traj': int(idx_sel_traj)})
            out = o_paths(oth, action_id_sel=action_id_sel, idx_sel_traj=idx_sel_traj)
            path_param, start_node, nodes, const_seg = out
            if rec.seam is not None:
                cur['zone_layers'], cur['zone_nodes'] = copy.deepcopy(rec.seam.zone_nodes)
            node_idx = getattr(oth, P + 'last_action_set_node_idx')
            coeff = getattr(oth, P + 'last_action_set_coeff')
            red = getattr(oth, P + 'last_action_set_red_len')
            keys = list(path_param.keys())
            cur['paths'] = {
                'start_node': [int(start_node[0]), int(start_node[1])],
                'keys': keys,
                'nodes': {k: [[None if a is None else int(a), None if b is None else int(b)] for a, b in nodes[k][0]]
                          for k in keys},
                'node_idx': {k: [int(i) for i in node_idx[k][0]] for k in keys},
                'n_rows': {k: int(np.shape(path_param[k][0])[0]) for k in keys},
                'red_len': {k: bool(red[k][0]) for k in red.keys() if k in keys},
                'const_rows': -1 if const_seg is None else int(np.shape(const_seg)[0]),
                'closest_obj_index': getattr(oth, P + 'closest_obj_index'),
            }
            cur['_full_paths'] = {'path_param': {k: np.array(path_param[k][0], dtype=float) for k in keys},
                                  'coeff': {k: np.array(coeff[k][0], dtype=float) for k in keys}}
            rec._cur = cur
            return out

        def get_ref_idx(oth, action_id_sel, idx_sel_traj, pos_est):
            out = o_ref(oth, action_id_sel=action_id_sel, idx_sel_traj=idx_sel_traj, pos_est=pos_est)
            cut_index_pos, cut_layer, vel_plan, vel_course, acc_plan = out
            rec._cur['pos_est'] = np.array(pos_est, dtype=float).reshape(-1)
            rec._cur['ref_idx'] = {'cut_index_pos': int(cut_index_pos), 'cut_layer': int(cut_layer),
                                   'vel_plan': float(vel_plan), 'vel_course': np.array(vel_course, dtype=float),
                                   'acc_plan': float(acc_plan)}
            return out

        def calc_vel_profile(oth, **kw):
            cur = rec._cur
            cur['vel_args'] = {'vel_est': float(kw['vel_est']), 'vel_max': float(kw['vel_max']),
                               'gg_scale': float(kw['gg_scale']), 'safety_d': float(kw['safety_d']),
                               'ax_max_machines': np.array(kw['ax_max_machines'], dtype=float),
                               'local_gg': [float(kw['local_gg'][0]), float(kw['local_gg'][1])]
                               if not isinstance(kw['local_gg'], dict) else None,
                               # location dependent friction: the rows are a function of the path coordinates (friction_map), so the
                               # replay rebuilds them from ITS paths; kept here: the rows of the first key as a cross-check
                               'local_gg_first': None if not isinstance(kw['local_gg'], dict)
                               else np.array(list(kw['local_gg'].values())[0][0], dtype=float),
                               'incl_emerg_traj': bool(kw.get('incl_emerg_traj', False))}
            cur['backup_available'] = getattr(oth, P + 'backup_nodes') is not None
            out = o_vel(oth, **kw)
            bp, ids, stamp, _ = out
            keys = list(bp.keys())
            cur['vel'] = {'keys': keys, 'traj_id': {k: int(v) for k, v in ids.items()},
                          'digest': {k: [int(bp[k][0].shape[0]), float(bp[k][0][-1, 0]), float(bp[k][0][0, 5]),
                                         float(bp[k][0][-1, 5]), float(np.sum(bp[k][0][:, 5])),
                                         float(np.sum(bp[k][0][:, 6]))] for k in keys},
                          'keys_after': list(getattr(oth, P + 'last_action_set_nodes').keys())}
            cur['_full_traj'] = {k: np.array(bp[k][0], dtype=float) for k in keys}
            rec.ticks.append(cur)
            rec._cur = None
            return out

        o_init = cls.set_initial_pose

        def set_initial_pose(oth, start_pos, start_heading, start_vel=0.0, max_heading_offset=np.pi / 4):
            out = o_init(oth, start_pos=start_pos, start_heading=start_heading, start_vel=start_vel,
                         max_heading_offset=max_heading_offset)
            pp = getattr(oth, P + 'last_action_set_path_param')
            rec.start = {'pos': np.array(start_pos, dtype=float).reshape(-1), 'heading': float(np.squeeze(start_heading)),
                         'vel': float(start_vel), 'max_heading_offset': float(max_heading_offset),
                         'in_track': bool(out[0]), 'cor_heading': bool(out[1]),
                         'start_node': [int(v) for v in getattr(oth, P + 'start_node')],
                         'path_param': np.array(pp['straight'][0], dtype=float),
                         'coeff': np.array(getattr(oth, P + 'last_action_set_coeff')['straight'][0], dtype=float)}
            return out

        wrap('set_initial_pose', set_initial_pose)
        wrap('update_objects', update_objects)
        wrap('calc_paths', calc_paths)
        wrap('get_ref_idx', get_ref_idx)
        wrap('calc_vel_profile', calc_vel_profile)
        self._cls = cls

    def uninstall(self):
        for name, orig in self._orig.items():
            setattr(self._cls, name, orig)
        self._orig = {}

    def export(self, full_every=25, always_full=()):
        """Ticks as serialisable dicts; the full arrays (stitched paths, coefficients, trajectories) are kept on every
        ``full_every``-th tick, around every change of the offered action set and on the ticks in ``always_full``."""
        keep = set(range(0, len(self.ticks), full_every)) | set(always_full)
        prev = None
        for i, t in enumerate(self.ticks):
            sig = (tuple(t['paths']['keys']), tuple(k for k in t['vel']['keys'] if k != 'emergency'), t['action_id_sel'],
                   tuple(sorted(t['paths']['red_len'].items())))
            if sig != prev:
                keep.update((max(i - 1, 0), i))
            prev = sig
        out = []
        for i, t in enumerate(self.ticks):
            d = {k: v for k, v in t.items() if not k.startswith('_')}
            d['tick'] = i
            d['start'] = self.start if i == 0 else None
            d['full'] = None
            if i in keep:
                d['full'] = {'path_param': t['_full_paths']['path_param'], 'coeff': t['_full_paths']['coeff'],
                             'traj': t['_full_traj']}
            out.append(d)
        return out
