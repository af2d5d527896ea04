// planner_api.hpp -- bodies of the ltpl_planner_* entry points of include/ltpl_hip.h on top of planner_core.hpp. The extern "C"
// symbols themselves are defined by the library that binds a Compute backend: libltpl_hip.so (ltpl_hip.hip, HIP kernels).
#pragma once

#include "planner_core.hpp"

struct ltpl_planner { ltplp::Planner P; };

namespace ltplp {

inline int api_create(Compute* cmp, const HostLat& lat, const ltpl_planner_config* cfg, ltpl_planner** out, std::string* why)
{
    if (!cfg || !out || cfg->n_scen < 1) { *why = "planner: null argument or n_scen < 1"; delete cmp; return LTPL_ERR_INVALID_ARG; }
    if (cfg->n_w_last < 0 || cfg->n_w_last > LTPL_MAX_LAST_NODES - 1) { *why = "planner: n_w_last out of range"; delete cmp; return LTPL_ERR_INVALID_ARG; }
    ltpl_planner* p = new ltpl_planner();
    p->P.lat = lat; p->P.cmp = cmp;
    Config& c = p->P.cfg;
    c.n_scen = cfg->n_scen;
    if (cfg->n_w_last > 0) c.w_last.assign(cfg->w_last_edges, cfg->w_last_edges + cfg->n_w_last);
    c.v_max_offset = cfg->v_max_offset; c.delaycomp = cfg->delaycomp; c.calc_time_safety = cfg->calc_time_safety;
    c.calc_time_buffer_len = cfg->calc_time_buffer_len; c.filt_window_width = cfg->filt_window_width;
    c.dyn_model_exp = cfg->dyn_model_exp; c.drag_coeff = cfg->drag_coeff; c.m_veh = cfg->m_veh;
    c.follow_control_type = cfg->follow_control_type; c.c_p = cfg->c_p; c.k_p = cfg->k_p; c.k_d = cfg->k_d; c.tan_w = cfg->tan_w;
    p->P.sc.resize((size_t)cfg->n_scen);
    *out = p;
    return LTPL_OK;
}

inline int api_get_caps(const ltpl_planner* p, ltpl_planner_caps* caps)
{
    if (!p || !caps) return LTPL_ERR_INVALID_ARG;
    // a stitched path = rows of the previous path in front of the new start node + the new path
    caps->cap_rows = 2 * p->P.lat.max_path_pts + 64;
    caps->cap_nodes = 2 * p->P.lat.max_path_nodes + 8;
    return LTPL_OK;
}

inline int api_calc_paths(ltpl_planner* p, const ltpl_planner_paths_in* in)
{
    if (!p) return LTPL_ERR_INVALID_ARG;
    if (!in || !in->prev_action || !in->t_now || !in->veh_off || !in->pos_off || !in->zone_off)
        return p->P.fail(LTPL_ERR_INVALID_ARG, "planner: null input");
    return p->P.calc_paths(in->prev_action, in->t_now, in->veh_off, in->pos_off, in->veh_radius, in->veh_vel, in->pos_x, in->pos_y,
                           in->zone_off, in->zone_gid);
}

inline int api_calc_paths_begin(ltpl_planner* p, const ltpl_planner_paths_in* in)
{
    if (!p) return LTPL_ERR_INVALID_ARG;
    if (!in || !in->prev_action || !in->t_now || !in->veh_off || !in->pos_off) return p->P.fail(LTPL_ERR_INVALID_ARG, "planner: null input");
    return p->P.calc_paths_begin(in->prev_action, in->t_now, in->veh_off, in->pos_off, in->veh_radius, in->veh_vel, in->pos_x, in->pos_y);
}

inline int api_calc_paths_finish(ltpl_planner* p, const int32_t* zone_off, const int32_t* zone_gid)
{
    if (!p) return LTPL_ERR_INVALID_ARG;
    if (!zone_off) return p->P.fail(LTPL_ERR_INVALID_ARG, "planner: null input");
    return p->P.calc_paths_finish(zone_off, zone_gid);
}

inline int api_calc_vel_profile(ltpl_planner* p, const ltpl_planner_vel_in* in)
{
    if (!p) return LTPL_ERR_INVALID_ARG;
    if (!in || !in->pos_est_x || !in->pos_est_y || !in->vel_est || !in->vel_max || !in->gg_scale || !in->gg_ax || !in->gg_ay ||
        !in->safety_d || !in->ax_max_machines || in->n_ax_max_machines < 1)
        return p->P.fail(LTPL_ERR_INVALID_ARG, "planner: null input");
    if (in->n_ax_tables > 1 || in->ax_table_off || in->ax_table_idx)
        return p->P.fail(LTPL_ERR_UNSUPPORTED, "planner: machine tables per planner (ABI v6) are a feature of ltpl_fleet_*; ltpl_planner_* takes one table per call");
    const int n = (int)p->P.sc.size();
    std::vector<VelReq> req((size_t)n);
    for (int s = 0; s < n; ++s) {
        VelReq& r = req[(size_t)s];
        r.pos_x = in->pos_est_x[s]; r.pos_y = in->pos_est_y[s]; r.vel_est = in->vel_est[s]; r.vel_max = in->vel_max[s];
        r.gg_scale = in->gg_scale[s]; r.gg_ax = in->gg_ax[s]; r.gg_ay = in->gg_ay[s]; r.safety_d = in->safety_d[s];
        r.incl_emerg = in->incl_emerg_traj ? in->incl_emerg_traj[s] : 0;
        for (int k = 0; k < LTPL_PLANNER_MAX_KEYS; ++k) {
            r.gg_rows[k] = nullptr; r.gg_n[k] = 0;
            if (in->gg_row_off && in->gg_rows) {
                const int o0 = in->gg_row_off[s * LTPL_PLANNER_MAX_KEYS + k], o1 = in->gg_row_off[s * LTPL_PLANNER_MAX_KEYS + k + 1];
                if (o0 < 0 || o1 < o0) return p->P.fail(LTPL_ERR_INVALID_ARG, "planner: gg_row_off must be non-decreasing");
                r.gg_rows[k] = in->gg_rows + (size_t)o0 * 2; r.gg_n[k] = o1 - o0;
            }
        }
    }
    return p->P.calc_vel_profile(req.data(), in->ax_max_machines, in->n_ax_max_machines, nullptr);
}

// copy-out of one accessor array: nothing to do without a destination or for an empty source (whose data() may be null --
// memcpy must not be handed a null pointer, not even for zero bytes)
template <class T>
inline void copy_out(T* dst, const std::vector<T>& src, size_t n)
{
    if (dst && n > 0) std::memcpy(dst, src.data(), sizeof(T) * n);
}

inline int api_get_paths(const ltpl_planner* p, int scen, ltpl_planner_paths_view* v)
{
    if (!p || !v || scen < 0 || scen >= (int)p->P.sc.size()) return LTPL_ERR_INVALID_ARG;
    const Scn& S = p->P.sc[(size_t)scen];
    ltpl_planner_caps caps; api_get_caps(p, &caps);
    v->n_keys = 0;
    v->start_node[0] = S.has_start ? S.start_node[0] : -1; v->start_node[1] = S.has_start ? S.start_node[1] : -1;
    v->const_rows = S.const_rows; v->closest_obj_index = S.closest_obj_index;
    for (const Traj& T : S.last) {
        if (v->n_keys >= LTPL_PLANNER_MAX_KEYS) break;
        const int k = v->n_keys++;
        v->key_id[k] = T.id; v->n_rows[k] = T.rows(); v->n_nodes[k] = T.n_nodes(); v->red_len[k] = T.red_len ? 1 : 0;
        if (T.rows() > caps.cap_rows || T.n_nodes() > caps.cap_nodes) return LTPL_ERR_CAPACITY;
        copy_out(v->path_param[k], T.pp, T.pp.size());
        copy_out(v->coeff[k], T.coeff, std::min(T.coeff.size(), (size_t)caps.cap_nodes * 8));
        copy_out(v->nodes[k], T.nodes, T.nodes.size());
        copy_out(v->node_idx[k], T.node_idx, std::min(T.node_idx.size(), (size_t)caps.cap_nodes));
    }
    return LTPL_OK;
}

inline int api_get_trajectories(const ltpl_planner* p, int scen, ltpl_planner_traj_view* v)
{
    if (!p || !v || scen < 0 || scen >= (int)p->P.sc.size()) return LTPL_ERR_INVALID_ARG;
    const Scn& S = p->P.sc[(size_t)scen];
    ltpl_planner_caps caps; api_get_caps(p, &caps);
    v->n_keys = 0;
    v->cut_index_pos = S.cut_index_pos; v->cut_layer = S.cut_layer; v->vel_plan = S.vel_plan; v->acc_plan = S.acc_plan;
    v->n_vel_course = (int)S.vel_course.size();
    v->n_ids = 0;
    for (const auto& e : S.path_ids) if (v->n_ids < LTPL_PLANNER_MAX_KEYS) { v->id_key[v->n_ids] = e.first; v->id_val[v->n_ids] = e.second; ++v->n_ids; }
    copy_out(v->vel_course, S.vel_course, std::min(S.vel_course.size(), (size_t)caps.cap_rows));
    for (const BpTraj& B : S.last_bp) {
        if (v->n_keys >= LTPL_PLANNER_MAX_KEYS) break;
        const int k = v->n_keys++;
        v->key_id[k] = B.id; v->traj_id[k] = B.traj_id; v->n_rows[k] = B.rows();
        if (B.rows() > caps.cap_rows) return LTPL_ERR_CAPACITY;
        copy_out(v->traj[k], B.bp, B.bp.size());
    }
    return LTPL_OK;
}

}  // namespace ltplp
