// fleet_api.hpp -- host-side helpers shared by the two bindings of fleet_core.hpp: libltpl_hip.so (state in device memory, ltpl_fleet_*)
// and the CPU test harness oracle/fleet_host_shim.cpp (state in host memory). Start poses are computed by the host planner's
// set_start (planner_core.hpp, OTH.py:181-270) and converted into a planner block; the query views read a host image of one block.
#pragma once

#include <cstring>
#include <string>
#include <vector>

#include "fleet_core.hpp"
#include "planner_core.hpp"

namespace fleet {

inline FLat flat_of(const ltplp::HostLat& h)
{
    FLat f{};
    f.L = h.L; f.V = h.V; f.closed = h.closed ? 1 : 0;
    f.lat_offset = h.lat_offset; f.vel_decrease_lat = h.vel_decrease_lat; f.veh_width = h.veh_width; f.veh_length = h.veh_length;
    f.sampled_resolution = h.sampled_resolution;
    f.layer_off = h.layer_off.data(); f.rl_idx = h.rl_idx.data(); f.s_rl = h.s_rl.data(); f.vel_rl = h.vel_rl.data();
    f.node_x = h.node_x.data(); f.node_y = h.node_y.data(); f.race_x = h.race_x.data(); f.race_y = h.race_y.data();
    return f;
}

inline int check_config(const ltpl_planner_config* cfg, std::string* why)
{
    if (!cfg || cfg->n_scen < 1) { *why = "planner config: null argument or n_scen < 1"; return LTPL_ERR_INVALID_ARG; }
    if (cfg->n_w_last < 0 || cfg->n_w_last > LTPL_MAX_LAST_NODES - 1) { *why = "planner config: n_w_last out of range"; return LTPL_ERR_INVALID_ARG; }
    if (cfg->calc_time_buffer_len < 1 || cfg->calc_time_buffer_len > CALC_BUF) { *why = "planner config: calc_time_buffer_len must be 1 .. 16"; return LTPL_ERR_CAPACITY; }
    if (cfg->filt_window_width < 1 || cfg->filt_window_width % 2 != 1) { *why = "planner config: Window width of moving average filter must be odd! (tph.conv_filt)"; return LTPL_ERR_INVALID_ARG; }
    return LTPL_OK;
}

// scratch rows of a planner block hold two coordinate lists of a path's nodes (paths_pre) and the job tables address rows with ints
inline int check_dims(const Dims& D, std::string* why)
{
    if (2 * D.CN > D.SR || D.RV > D.R || D.R > D.SR) { *why = "planner: inconsistent block dimensions"; return LTPL_ERR_CAPACITY; }      // (holds by construction, make_dims)
    return LTPL_OK;
}

inline FCfg fcfg_of(const ltpl_planner_config* cfg)
{
    FCfg c{}; c.v_max_offset = cfg->v_max_offset; c.delaycomp = cfg->delaycomp; c.calc_time_safety = cfg->calc_time_safety;
    c.calc_time_buffer_len = cfg->calc_time_buffer_len; c.filt_window_width = cfg->filt_window_width;
    return c;
}

// OnlineTrajectoryHandler.set_initial_pose (planner_core.hpp) converted into a zeroed planner block image
// (`prev`: the planner's scalars before the call or nullptr -- set_initial_pose only re-initialises the iterative memory, OTH.py:161-179)
inline int start_block(const ltplp::HostLat& lat, const Dims& D, double x, double y, double heading, double vel, double mho,
                       int* in_track, int* cor_heading, unsigned char* image, const PlannerS* prev, std::string* why)
{
    ltplp::StartPose H;
    const int rc = ltplp::set_initial_pose(lat, x, y, heading, vel, mho, in_track, cor_heading, &H, why);
    if (rc) return rc;
    PlannerS keep{}; const bool carry = prev != nullptr;
    if (carry) keep = *prev;
    std::memset(image, 0, D.stride);
    Block B{image, D, nullptr};
    PlannerS& S = *B.S();
    if (carry) {
        S.traj_base_id = keep.traj_base_id; S.n_calc = keep.n_calc; std::memcpy(S.calc_buffer, keep.calc_buffer, sizeof(S.calc_buffer));
        S.has_old_gg = keep.has_old_gg; S.cut_index_pos = keep.cut_index_pos; S.cut_layer = keep.cut_layer; S.vel_plan = keep.vel_plan; S.acc_plan = keep.acc_plan;
    }
    S.v_start = H.v_start; S.has_start = H.has_start ? 1 : 0; S.start_node[0] = H.start_node[0]; S.start_node[1] = H.start_node[1];
    S.em_base_id = carry ? keep.em_base_id : LTPL_ACT_NONE; S.action_forced = H.action_forced; S.closest_obj_index = carry ? keep.closest_obj_index : -1;
    S.sel_action = S.raw_action = LTPL_ACT_NONE;
    S.const_rows = -1; S.old_gg_scale = carry ? keep.old_gg_scale : 1.0;
    S.has_last = H.has_last ? 1 : 0;
    if (H.has_last) {
        const ltplp::Traj& T = H.last;
        if (T.rows() > D.R || T.n_nodes() > D.CN) { *why = "fleet: start spline exceeds the row capacity"; return LTPL_ERR_CAPACITY; }
        TrajM& M = S.tm[0][0];
        M.id = T.id; M.red_len = T.red_len ? 1 : 0; M.rows = T.rows(); M.nc = (int)(T.coeff.size() / 8); M.nn = T.n_nodes(); M.ni = (int)T.node_idx.size();
        { const Rows r = B.pp(0, 0); for (int i = 0; i < T.rows(); ++i) for (int c = 0; c < 5; ++c) r.at(i, c) = T.pp[(size_t)i * 5 + c]; }
        std::memcpy(B.coeff(0, 0), T.coeff.data(), sizeof(double) * T.coeff.size());
        std::memcpy(B.nodes(0, 0), T.nodes.data(), sizeof(int) * T.nodes.size());
        std::memcpy(B.nidx(0, 0), T.node_idx.data(), sizeof(int) * T.node_idx.size());
        S.n_last = 1; S.last_slot[0] = 0;
    }
    return LTPL_OK;
}

inline void caps_of(const Dims& D, ltpl_planner_caps* c) { c->cap_rows = D.R; c->cap_nodes = D.CN; }

inline int paths_view(const Dims& D, const unsigned char* image, ltpl_planner_paths_view* v)
{
    Block B{const_cast<unsigned char*>(image), D, nullptr};
    const PlannerS& S = *B.S();
    v->n_keys = 0;
    v->start_node[0] = S.has_start ? S.start_node[0] : -1; v->start_node[1] = S.has_start ? S.start_node[1] : -1;
    v->const_rows = S.const_rows; v->closest_obj_index = S.closest_obj_index;
    for (int i = 0; i < S.n_last && v->n_keys < LTPL_PLANNER_MAX_KEYS; ++i) {
        const int sl = S.last_slot[i]; const TrajM& T = S.tm[S.cur_set][sl];
        const int k = v->n_keys++;
        v->key_id[k] = T.id; v->n_rows[k] = T.rows; v->n_nodes[k] = T.nn; v->red_len[k] = T.red_len;
        if (v->path_param[k]) { const Rows r = B.pp(S.cur_set, sl).from(T.r0); for (int i = 0; i < T.rows; ++i) for (int c = 0; c < 5; ++c) v->path_param[k][(size_t)i * 5 + c] = r.at(i, c); }   // (column-major in the state)
        if (v->coeff[k] && T.nc > 0) std::memcpy(v->coeff[k], B.coeff(S.cur_set, sl) + (size_t)T.c0 * 8, sizeof(double) * 8 * (size_t)T.nc);
        if (v->nodes[k] && T.nn > 0) std::memcpy(v->nodes[k], B.nodes(S.cur_set, sl) + (size_t)T.n0 * 2, sizeof(int) * 2 * (size_t)T.nn);
        if (v->node_idx[k] && T.ni > 0) std::memcpy(v->node_idx[k], B.nidx(S.cur_set, sl) + T.i0, sizeof(int) * (size_t)T.ni);
    }
    return LTPL_OK;
}

inline int traj_view(const Dims& D, const unsigned char* image, ltpl_planner_traj_view* v)
{
    Block B{const_cast<unsigned char*>(image), D, nullptr};
    const PlannerS& S = *B.S();
    v->n_keys = 0;
    v->cut_index_pos = S.cut_index_pos; v->cut_layer = S.cut_layer; v->vel_plan = S.vel_plan; v->acc_plan = S.acc_plan;
    v->n_vel_course = S.n_vel_course;
    v->n_ids = 0;
    for (int i = 0; i < S.n_ids && v->n_ids < LTPL_PLANNER_MAX_KEYS; ++i) { v->id_key[v->n_ids] = S.id_key[i]; v->id_val[v->n_ids] = S.id_val[i]; ++v->n_ids; }
    if (v->vel_course && S.n_vel_course > 0) std::memcpy(v->vel_course, B.velc(), sizeof(double) * (size_t)S.n_vel_course);
    for (int i = 0; i < S.n_bp && v->n_keys < LTPL_PLANNER_MAX_KEYS; ++i) {
        const int k = v->n_keys++;
        v->key_id[k] = S.bp_id[i]; v->traj_id[k] = S.bp_traj_id[i]; v->n_rows[k] = S.bp_rows[i];
        if (v->traj[k]) { const Rows r = B.bp(S.bp_slot[i]); for (int q = 0; q < S.bp_rows[i]; ++q) for (int c = 0; c < 7; ++c) v->traj[k][(size_t)q * 7 + c] = r.at(q, c); }
    }
    return LTPL_OK;
}

inline std::string err_text(int planner, int err)
{
    static const char* sites[] = {"", "neither 'straight' nor 'follow' in the last action set (KeyError, OTH.py:334)", "no start node (set_start first)",
        "stitched path exceeds the row capacity", "stitched path exceeds the node capacity", "cut_layer beyond the node list (IndexError, OTH.py:712)",
        "vel_plan > vel_max + 0.1 (brake prefix): the reference raises ValueError at OTH.py:919", "follow profile without points", "trajectory without nodes",
        "end node is None", "follow profile shorter than the path (OTH.py:830)", "velocity profile shorter than the path (OTH.py:919)",
        "fewer than 6 rows (IndexError, OTH.py:923)", "cut_layer beyond the backup plan", "backup plan shorter than the cut index",
        "backup brake profile length mismatch", "emergency profile without any trajectory (IndexError, OTH.py:1029)", "calc_time_buffer_len above 16",
        "local_gg in its dict form is not supported by the fleet (use the host planner)", "more velocity jobs than slots", "start layer without a planning range (end of an open track)",
        "velocity job longer than max_path_pts + 64 points",
        "local_gg rows of a path do not match its coordinates (OTH.py:641-646)",
        "emergency profile: the friction rows of the first path do not match the first trajectory (a backup plan): the reference raises "
        "RuntimeError 'Length of loc_gg and kappa must be equal!' (OTH.py:1031, calc_brake_emergency.py:31)"};
    const int site = (err >> 8) & 0xff;
    return "fleet: planner " + std::to_string(planner) + ": " + (site > 0 && site < (int)(sizeof(sites) / sizeof(sites[0])) ? sites[site] : "error");
}

}  // namespace fleet
