// planner_host.hpp -- ltpl_planner_* (include/ltpl_hip.h, ABI v3+): the reference's OnlineTrajectoryHandler
// (graph_ltpl/online_graph/src/OnlineTrajectoryHandler.py:24-1040) behind the C ABI for planners whose iterative memory lives on the HOST.
//
// ONE STATE MACHINE (round 4). The stages of a tick -- OTH.calc_paths in front of / behind seam (1), get_ref_idx, the four stages of
// calc_vel_profile around the launches of seam (2) -- are the functions of fleet_core.hpp, the source the fleet (ltpl_fleet_*) runs on
// the device with one wave64 per planner. This file instantiates them with the one-lane host policy (fleet::HostX) on planner blocks in
// host memory and drives the two seams through a `Compute` backend: the HIP kernels in libltpl_hip.so, the oracle's CPU arithmetic in
// the test harnesses (oracle/planner_host_shim.cpp, oracle/fleet_host_shim.cpp). Rounds 2 / 3 kept a second implementation of the same
// state machine for the host (planner_core.hpp, std::vector based) next to the fleet's and held the two together with differential tests;
// planner_core.hpp now only holds what is host-only by nature: the lattice tables (HostLat), set_initial_pose (the start spline,
// OTH.py:181-270) and the projections behind ltpl_const_segment_test / ltpl_raceline_s.
//
// Differences between the two front ends that remain, both in this file:
//   * errors: ltpl_planner_* validates a calc_vel_profile call for ALL planners before any planner's memory is cut (the reference's
//     ValueError / IndexError of one vehicle must not leave the others half-trimmed) and does not keep an error state; the fleet (and the
//     fleet harness, `sticky_errors`) reports per planner and keeps the failing planner's state until it gets a new start pose;
//   * the velocity jobs of a stage go to the backend in ONE call per distinct car (vel_max, machine table) of the call.
#pragma once

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "fleet_api.hpp"

namespace ltplp {

struct HostJobSet {
    int per = 0; std::vector<fleet::VelJob> jobs; std::vector<double> pool, out; std::vector<int> flags;
    void init(int N, int per_planner, int R)
    {
        per = per_planner; jobs.assign((size_t)N * per, fleet::VelJob{}); pool.assign((size_t)N * per * 4 * R, 0.0); out.assign((size_t)N * per * R, 0.0);
        flags.assign((size_t)N * per * 2, 0);
    }
    fleet::FJobs view() { return fleet::FJobs{jobs.data(), pool.data(), out.data(), flags.data(), per, nullptr, 0, -1}; }
};

struct HostPlanner {
    Compute* cmp = nullptr;                 // owned
    HostLat lat; fleet::FLat flat; fleet::FCfg cfg; fleet::Dims D; ltpl_planner_config pc; std::vector<double> w_last;
    std::vector<unsigned char> state, image, gg;         // gg: friction rows per planner (fleet::Dims::gg_stride), allocated by the first call that carries rows
    std::string err;
    bool sticky_errors = false;             // fleet semantics: a failing planner keeps its error until set_start (fleet harness)
    // inputs of the tick (the velocity stage reads the objects of calc_paths again)
    std::vector<int> prev_action, veh_off, pos_off; std::vector<double> t_now, radius, vel, px, py;
    // seam (1)
    std::vector<int> p_sl, p_sn, p_fl, p_la, p_cc, p_nl, p_ll, p_ln; std::vector<double> p_psi;
    std::vector<int> o_end, o_coi, o_con, o_na, o_id, o_valid, o_red, o_goal, o_nn, o_np, o_nt, o_nodes, o_nidx; std::vector<double> o_co, o_pp;
    HostJobSet JA, JB, JC;
    bool began = false;

    ~HostPlanner() { delete cmp; }
    fleet::FObj obj() { return fleet::FObj{prev_action.data(), t_now.data(), veh_off.data(), pos_off.data(), radius.data(), vel.data(), px.data(), py.data()}; }
    fleet::Block block(int p) const
    {
        return fleet::Block{const_cast<unsigned char*>(state.data()) + D.stride * (size_t)p, D,
                            gg.empty() ? nullptr : const_cast<unsigned char*>(gg.data()) + D.gg_stride * (size_t)p};
    }
    int fail(int code, const std::string& msg) { err = msg; return code; }

    // status of the call from the planners' error words; without sticky errors the words are cleared (the call is reported, not remembered)
    int first_error()
    {
        int rc = LTPL_OK;
        for (int p = 0; p < D.N; ++p) {
            fleet::PlannerS* S = block(p).S();
            if (S->err && rc == LTPL_OK) { err = fleet::err_text(p, S->err); rc = S->err & 0xff; }
            if (!sticky_errors) S->err = 0;
        }
        return rc;
    }

    int create(Compute* c, const HostLat& hl, const ltpl_planner_config* cf, std::string* why)
    {
        cmp = c;
        int rc = fleet::check_config(cf, why);
        if (rc) return rc;
        lat = hl; flat = fleet::flat_of(lat); cfg = fleet::fcfg_of(cf); pc = *cf; pc.w_last_edges = nullptr;
        if (cf->n_w_last > 0) w_last.assign(cf->w_last_edges, cf->w_last_edges + cf->n_w_last);
        D = fleet::make_dims(cf->n_scen, lat.max_path_nodes, lat.max_path_pts);
        if ((rc = fleet::check_dims(D, why))) return rc;
        state.assign(D.stride * (size_t)cf->n_scen, 0); image.resize(D.stride);
        for (int p = 0; p < cf->n_scen; ++p) {
            fleet::PlannerS* S = block(p).S();
            S->em_base_id = S->action_forced = S->sel_action = S->raw_action = LTPL_ACT_NONE; S->closest_obj_index = -1; S->const_rows = -1; S->old_gg_scale = 1.0;
        }
        JA.init(D.N, fleet::JOBS_A, D.R); JB.init(D.N, 1, D.R); JC.init(D.N, 1, D.R);
        return LTPL_OK;
    }

    // OnlineTrajectoryHandler.set_initial_pose (OTH.py:181-270)
    int set_start(int p, double x, double y, double heading, double v, double mho, int* in_track, int* cor_heading)
    {
        if (p < 0 || p >= D.N) return fail(LTPL_ERR_INVALID_ARG, "scenario index out of range");
        const fleet::PlannerS prev = *block(p).S();
        const int rc = fleet::start_block(lat, D, x, y, heading, v, mho, in_track, cor_heading, image.data(), &prev, &err);
        if (rc) return rc;
        std::memcpy(block(p).b, image.data(), D.stride);
        return LTPL_OK;
    }

    // OTH.update_objects + OTH.calc_paths in front of seam (1)
    int calc_paths_begin(const ltpl_planner_paths_in* in)
    {
        const int n = D.N;
        if (in->veh_off[0] != 0 || in->pos_off[0] != 0) return fail(LTPL_ERR_INVALID_ARG, "offset arrays must start at 0");
        const int nv = in->veh_off[n];
        if (nv < 0) return fail(LTPL_ERR_INVALID_ARG, "negative offsets");
        const int np_ = in->pos_off[nv];
        for (int v = 0; v < nv; ++v) if (in->pos_off[v + 1] - in->pos_off[v] < 1) return fail(LTPL_ERR_INVALID_ARG, "vehicle without position");
        if ((nv > 0 && !in->veh_radius) || (np_ > 0 && (!in->pos_x || !in->pos_y))) return fail(LTPL_ERR_INVALID_ARG, "planner: null vehicle / position arrays");
        prev_action.assign(in->prev_action, in->prev_action + n); t_now.assign(in->t_now, in->t_now + n);
        veh_off.assign(in->veh_off, in->veh_off + n + 1); pos_off.assign(in->pos_off, in->pos_off + nv + 1);
        radius.assign(in->veh_radius, in->veh_radius + nv); px.assign(in->pos_x, in->pos_x + np_); py.assign(in->pos_y, in->pos_y + np_);
        if (in->veh_vel) vel.assign(in->veh_vel, in->veh_vel + nv); else vel.assign((size_t)nv, 0.0);
        radius.push_back(0.0); vel.push_back(0.0); px.push_back(0.0); py.push_back(0.0);          // never empty
        p_sl.assign(n, 0); p_sn.assign(n, 0); p_fl.assign(n, 0); p_la.assign(n, LTPL_ACT_NONE); p_cc.assign(n, -1); p_psi.assign(n, 0.0); p_nl.assign(n, 0);
        p_ll.assign((size_t)n * LTPL_MAX_LAST_NODES, -1); p_ln.assign((size_t)n * LTPL_MAX_LAST_NODES, -1);
        fleet::FPathsIn pin{p_sl.data(), p_sn.data(), p_fl.data(), p_la.data(), p_cc.data(), p_psi.data(), p_nl.data(), p_ll.data(), p_ln.data()};
        LTPL_PROF(prof_pre, "planner.paths_pre");
        fleet::HostX x;
        for (int p = 0; p < n; ++p) {
            fleet::Block B = block(p);
            B.S()->ref_done = 0;                     // new paths: a reference index computed for the previous memory is stale
            if (!B.S()->err) fleet::paths_pre(x, flat, cfg, B, *B.S(), p, obj(), pin);
            // (a start layer without a planning range -- the end of an open track -- is reported by seam (1) itself)
            if (B.S()->err) { p_sl[p] = 0; p_sn[p] = lat.rl_idx[0]; p_fl[p] = LTPL_FLAG_ACTION_SETS; p_la[p] = LTPL_ACT_NONE; p_cc[p] = -1; p_nl[p] = 0; }   // a harmless scenario (its result is not looked at)
        }
        const int rc = first_error();
        began = rc == LTPL_OK || sticky_errors;
        return rc;
    }

    // seam (1) + OTH.calc_paths behind it
    int calc_paths_finish(const int32_t* zone_off, const int32_t* zone_gid)
    {
        if (!began) return fail(LTPL_ERR_INVALID_ARG, "planner: calc_paths_finish without calc_paths_begin");
        began = false;
        const int n = D.N, A = LTPL_MAX_ACTIONS, cn = D.cn, cp = D.cp;
        ltpl_paths_in in; std::memset(&in, 0, sizeof(in));
        static const double zero = 0.0;
        in.n_scen = n; in.n_w_last = (int)w_last.size(); in.w_last_edges = w_last.empty() ? &zero : w_last.data();
        in.start_layer = p_sl.data(); in.start_node = p_sn.data(); in.flags = p_fl.data(); in.last_action = p_la.data();
        in.const_closest = p_cc.data(); in.psi_s = p_psi.data();
        in.veh_off = veh_off.data(); in.pos_off = pos_off.data(); in.veh_radius = radius.data(); in.pos_x = px.data(); in.pos_y = py.data();
        in.zone_off = zone_off; in.zone_gid = zone_gid;
        in.n_last = p_nl.data(); in.last_layer = p_ll.data(); in.last_node = p_ln.data();
        o_end.resize(n); o_coi.resize(n); o_con.resize((size_t)n * 2); o_na.resize(n);
        for (auto* v : {&o_id, &o_valid, &o_red, &o_goal, &o_nn, &o_np, &o_nt}) v->resize((size_t)n * A);
        o_nodes.resize((size_t)n * A * cn); o_nidx.resize((size_t)n * A * cn); o_co.resize((size_t)n * A * cn * 8); o_pp.resize((size_t)n * A * cp * 5);
        ltpl_paths_out out; std::memset(&out, 0, sizeof(out));
        out.cap_nodes = cn; out.cap_pts = cp; out.end_layer = o_end.data(); out.closest_obj_index = o_coi.data(); out.closest_obj_node = o_con.data();
        out.n_actions = o_na.data(); out.action_id = o_id.data(); out.valid = o_valid.data(); out.reduced = o_red.data(); out.goal_layer = o_goal.data();
        out.n_nodes = o_nn.data(); out.n_pts = o_np.data(); out.n_ties = o_nt.data(); out.nodes = o_nodes.data(); out.node_idx = o_nidx.data();
        out.coeff = o_co.data(); out.path_param = o_pp.data();
        const int rc = cmp->plan_paths(&in, &out);
        if (rc) return fail(rc, std::string("planner: seam (1) failed: ") + cmp->last_error());
        fleet::FPathsOut po{o_coi.data(), o_na.data(), o_id.data(), o_valid.data(), o_red.data(), o_nn.data(), o_np.data(), o_nodes.data(),
                            o_nidx.data(), o_co.data(), o_pp.data()};
        LTPL_PROF(prof_post, "planner.paths_post");
        fleet::HostX x;
        for (int p = 0; p < n; ++p) { fleet::Block B = block(p); if (!B.S()->err) fleet::paths_post(x, flat, B, *B.S(), p, po); }
        return first_error();
    }

    int calc_paths(const ltpl_planner_paths_in* in)
    {
        const int n = D.N;
        if (!in->zone_off || in->zone_off[0] != 0) return fail(LTPL_ERR_INVALID_ARG, "planner: zone offsets missing");
        for (int s = 0; s < n; ++s) if (in->zone_off[s + 1] < in->zone_off[s]) return fail(LTPL_ERR_INVALID_ARG, "planner: zone offsets must not decrease");
        if (in->zone_off[n] > 0 && !in->zone_gid) return fail(LTPL_ERR_INVALID_ARG, "planner: zone node ids missing");
        const int rc = calc_paths_begin(in);
        if (rc && !sticky_errors) return rc;
        const int rc2 = calc_paths_finish(in->zone_off, in->zone_gid);
        return rc ? rc : rc2;
    }

    // OnlineTrajectoryHandler.get_ref_idx (OTH.py:518-601)
    int get_ref_idx(const double* ex, const double* ey)
    {
        fleet::HostX x;
        for (int p = 0; p < D.N; ++p) { fleet::Block B = block(p); if (!B.S()->err) { fleet::ref_idx(x, cfg, B, *B.S(), ex[p], ey[p]); B.S()->ref_done = 1; } }
        return first_error();
    }

    // the jobs of one stage through seam (2): ONE backend call per distinct car (vel_max, machine table) of the stage
    int run_jobs(const ltpl_vel_params& vp, HostJobSet& J)
    {
        std::vector<ltpl_vel_job> jobs; std::vector<ltpl_vel_result> res; std::vector<size_t> idx;
        for (size_t j = 0; j < J.jobs.size(); ++j) {
            const fleet::VelJob& v = J.jobs[j];
            if (v.n <= 0) continue;
            ltpl_vel_job jb; std::memset(&jb, 0, sizeof(jb));
            jb.mode = v.mode; jb.n = v.n; jb.n_el = v.n_el; jb.has_v_end = v.has_v_end; jb.v_start = v.v_start; jb.v_end = v.v_end; jb.v_ego = v.v_ego;
            jb.v_obj = v.v_obj; jb.safety_d = v.safety_d; jb.obj_dist = v.obj_dist; jb.obj_x = v.obj_x; jb.obj_y = v.obj_y;
            jb.kappa = J.pool.data() + v.off_kappa; jb.el_lengths = J.pool.data() + v.off_el; jb.loc_gg = J.pool.data() + v.off_gg;
            ltpl_vel_result r; r.vx = J.out.data() + v.off_out; r.too_close = 0; r.vel_bound = 1;
            jobs.push_back(jb); res.push_back(r); idx.push_back(j);
        }
        if (jobs.empty()) return LTPL_OK;
        std::vector<char> done(jobs.size(), 0);
        for (size_t i = 0; i < jobs.size(); ++i) {
            if (done[i]) continue;
            const fleet::VelJob& v = J.jobs[idx[i]];
            std::vector<ltpl_vel_job> gj; std::vector<ltpl_vel_result> gr; std::vector<size_t> gi;
            for (size_t k = i; k < jobs.size(); ++k) {
                const fleet::VelJob& w = J.jobs[idx[k]];
                if (!done[k] && w.v_max == v.v_max && w.axm_off == v.axm_off && w.n_axm == v.n_axm) { gj.push_back(jobs[k]); gr.push_back(res[k]); gi.push_back(k); done[k] = 1; }
            }
            ltpl_vel_params pj = vp;
            if (v.v_max > 0.0) pj.v_max = v.v_max;
            if (v.n_axm > 0) { pj.n_ax_max_machines = v.n_axm; pj.ax_max_machines = vp.ax_max_machines + 2 * (size_t)v.axm_off; }
            const int rc = cmp->vel_profile(&pj, (int)gj.size(), gj.data(), gr.data());
            if (rc) return fail(rc, std::string("planner: seam (2) failed: ") + cmp->last_error());
            for (size_t k = 0; k < gi.size(); ++k) res[gi[k]] = gr[k];
        }
        for (size_t i = 0; i < idx.size(); ++i) { J.flags[2 * idx[i]] = res[i].too_close; J.flags[2 * idx[i] + 1] = res[i].vel_bound; }
        return LTPL_OK;
    }

    // OnlineTrajectoryHandler.calc_vel_profile (OTH.py:603-1040)
    int calc_vel_profile(const ltpl_planner_vel_in* in)
    {
        const int n = D.N, MK = LTPL_PLANNER_MAX_KEYS;
        const int n_tab = in->n_ax_tables > 1 ? in->n_ax_tables : 0;
        if (n_tab) {
            if (!in->ax_table_off || !in->ax_table_idx) return fail(LTPL_ERR_INVALID_ARG, "planner: n_ax_tables > 1 without ax_table_off / ax_table_idx");
            if (in->ax_table_off[0] != 0 || in->ax_table_off[n_tab] != in->n_ax_max_machines) return fail(LTPL_ERR_INVALID_ARG, "planner: ax_table_off must run from 0 to n_ax_max_machines");
            for (int t = 0; t < n_tab; ++t) if (in->ax_table_off[t + 1] - in->ax_table_off[t] < 1) return fail(LTPL_ERR_INVALID_ARG, "planner: empty machine table");
            for (int s = 0; s < n; ++s) if (in->ax_table_idx[s] < 0 || in->ax_table_idx[s] >= n_tab) return fail(LTPL_ERR_INVALID_ARG, "planner: ax_table_idx out of range");
        }
        // capacity of the velocity stage's machine tables (64 rows per table) and the per-planner speed limits: checked HERE, in front of
        // the first stage -- the backend's own check would only fire after stage A has trimmed every planner's memory
        for (int t = 0; t < (n_tab ? n_tab : 1); ++t) {
            const int rows_t = n_tab ? in->ax_table_off[t + 1] - in->ax_table_off[t] : in->n_ax_max_machines;
            if (rows_t < 1 || rows_t > 64) return fail(LTPL_ERR_CAPACITY, "planner: a machine table holds 1 .. 64 rows");
        }
        for (int s = 0; s < n; ++s) if (!(in->vel_max[s] > 0.0)) return fail(LTPL_ERR_INVALID_ARG, "planner: vel_max must be positive");
        const bool rows = in->gg_row_off && in->gg_rows;
        if (rows && gg.empty()) gg.assign(D.gg_stride * (size_t)n, 0);
        if (rows) for (int i = 0; i < n * MK; ++i) if (in->gg_row_off[i] < 0 || in->gg_row_off[i + 1] < in->gg_row_off[i]) return fail(LTPL_ERR_INVALID_ARG, "planner: gg_row_off must be non-decreasing");
        ltpl_vel_params vp; std::memset(&vp, 0, sizeof(vp));
        vp.dyn_model_exp = pc.dyn_model_exp; vp.drag_coeff = pc.drag_coeff; vp.m_veh = pc.m_veh; vp.len_veh = lat.veh_length;
        vp.n_ax_max_machines = n_tab ? in->ax_table_off[1] - in->ax_table_off[0] : in->n_ax_max_machines; vp.ax_max_machines = in->ax_max_machines;
        vp.follow_control_type = pc.follow_control_type; vp.c_p = pc.c_p; vp.k_p = pc.k_p; vp.k_d = pc.k_d; vp.tan_w = pc.tan_w; vp.v_max = in->vel_max[0];
        fleet::FVelIn vin{in->pos_est_x, in->pos_est_y, in->vel_est, in->vel_max, in->gg_scale, in->gg_ax, in->gg_ay, in->safety_d, in->incl_emerg_traj,
                          n_tab ? in->ax_table_off : nullptr, n_tab ? in->ax_table_idx : nullptr, rows ? in->gg_row_off : nullptr, rows ? in->gg_rows : nullptr};
        fleet::HostX x; int rc;
        if (!sticky_errors) {
            // The failures that depend on the ARGUMENTS and the stored trajectories alone (cut layer beyond a path, friction rows of the
            // wrong length, a brake prefix, an empty follow window: E_CUT_LAYER, E_GG_ROWS, E_BRAKE_PREFIX, E_FOLLOW_EMPTY) are checked for
            // ALL planners before any planner's memory is touched: such an error of one planner of a batch (the reference's ValueError /
            // IndexError for that vehicle) must not leave the others half-trimmed. The reference index itself (OTH.get_ref_idx) cuts
            // nothing; a failing call forgets it again.
            // NOT covered: failures that only show while the stages run (job / row capacities E_CAP_JOBS, E_CAP_VEL, a velocity course
            // shorter than the cut E_VX_SHORT, E_ROW5, the emergency profile on friction rows E_EMERG_GG, a backend error inside run_jobs).
            // Those return after stage A has trimmed the planners' memories and advanced traj_base_id, and first_error() clears the
            // planner's error word: the failing planner is then in the state the REFERENCE object is in after the same exception (its
            // calc_vel_profile raises half way through as well, OTH.py:700-1040) -- callers re-initialise it with set_start, as the
            // reference's callers must. (The fleet entry points keep the error sticky per planner instead: ltpl_fleet_*.)
            auto fail_pre = [&](int p, int code, int site) {
                for (int q = 0; q < n; ++q) block(q).S()->ref_done = 0;
                return fail(code, fleet::err_text(p, code | (site << 8)));
            };
            for (int p = 0; p < n; ++p) {
                fleet::Block B = block(p); fleet::PlannerS& S = *B.S();
                if (!S.ref_done) { fleet::ref_idx(x, cfg, B, S, in->pos_est_x[p], in->pos_est_y[p]); S.ref_done = 1; }
                for (int k = 0; k < S.n_last; ++k) {
                    const fleet::TrajM& T = S.tm[S.cur_set][S.last_slot[k]];
                    const int c0 = S.cut_index_pos < 0 ? 0 : (S.cut_index_pos < T.rows ? S.cut_index_pos : T.rows), m = T.rows - c0;
                    if (S.cut_layer >= T.ni) return fail_pre(p, LTPL_ERR_INVALID_ARG, fleet::E_CUT_LAYER);
                    if (rows && k < MK) { const int g = in->gg_row_off[p * MK + k + 1] - in->gg_row_off[p * MK + k]; if (g > 0 && g != T.rows) return fail_pre(p, LTPL_ERR_INVALID_ARG, fleet::E_GG_ROWS); }
                    if (m > 0 && S.vel_plan > in->vel_max[p] + 0.1) return fail_pre(p, LTPL_ERR_UNSUPPORTED, fleet::E_BRAKE_PREFIX);
                    if (m > 0 && T.id == LTPL_ACT_FOLLOW && m - S.n_vel_course < 1) return fail_pre(p, LTPL_ERR_INVALID_ARG, fleet::E_FOLLOW_EMPTY);
                }
            }
        }
        fleet::FJobs ja = JA.view(), jb = JB.view(), jc = JC.view();
        { LTPL_PROF(prof_a, "planner.vel_stage_A"); for (int p = 0; p < n; ++p) { fleet::Block B = block(p); fleet::vel_a(x, flat, cfg, B, *B.S(), p, obj(), vin, ja); } }
        if ((rc = run_jobs(vp, JA))) return rc;
        { LTPL_PROF(prof_b, "planner.vel_stage_B"); for (int p = 0; p < n; ++p) { fleet::Block B = block(p); fleet::vel_b(x, cfg, B, *B.S(), p, ja, jb); } }
        if ((rc = run_jobs(vp, JB))) return rc;
        { LTPL_PROF(prof_c, "planner.vel_stage_C"); for (int p = 0; p < n; ++p) { fleet::Block B = block(p); fleet::vel_c(x, cfg, B, *B.S(), p, vin, jb, jc); } }
        ltpl_vel_params ve = vp; ve.dyn_model_exp = 1.0; ve.drag_coeff = 0.854; ve.m_veh = 1160.0;     // calc_brake_emergency.py:4-6,31-36
        if ((rc = run_jobs(ve, JC))) return rc;
        { LTPL_PROF(prof_d, "planner.vel_stage_D"); for (int p = 0; p < n; ++p) { fleet::Block B = block(p); fleet::vel_d(x, B, *B.S(), p, vin, jc); } }
        return first_error();
    }

    int get_paths(int p, ltpl_planner_paths_view* v) const
    {
        if (!v || p < 0 || p >= D.N) return LTPL_ERR_INVALID_ARG;
        return fleet::paths_view(D, block(p).b, v);
    }
    int get_trajectories(int p, ltpl_planner_traj_view* v) const
    {
        if (!v || p < 0 || p >= D.N) return LTPL_ERR_INVALID_ARG;
        return fleet::traj_view(D, block(p).b, v);
    }
};

}  // namespace ltplp

struct ltpl_planner { ltplp::HostPlanner P; };

namespace ltplp {

inline int api_create(Compute* cmp, const HostLat& lat, const ltpl_planner_config* cfg, ltpl_planner** out, std::string* why, bool sticky = false)
{
    if (!cfg || !out) { *why = "planner: null argument"; delete cmp; return LTPL_ERR_INVALID_ARG; }
    ltpl_planner* p = new ltpl_planner();
    p->P.sticky_errors = sticky;
    const int rc = p->P.create(cmp, lat, cfg, why);          // (takes ownership of cmp)
    if (rc) { delete p; return rc; }
    *out = p;
    return LTPL_OK;
}
inline int api_get_caps(const ltpl_planner* p, ltpl_planner_caps* caps) { if (!p || !caps) return LTPL_ERR_INVALID_ARG; fleet::caps_of(p->P.D, caps); return LTPL_OK; }
inline int api_calc_paths(ltpl_planner* p, const ltpl_planner_paths_in* in)
{
    if (!p) return LTPL_ERR_INVALID_ARG;
    if (!in || !in->prev_action || !in->t_now || !in->veh_off || !in->pos_off || !in->zone_off) return p->P.fail(LTPL_ERR_INVALID_ARG, "planner: null input");
    return p->P.calc_paths(in);
}
inline int api_calc_paths_begin(ltpl_planner* p, const ltpl_planner_paths_in* in)
{
    if (!p) return LTPL_ERR_INVALID_ARG;
    if (!in || !in->prev_action || !in->t_now || !in->veh_off || !in->pos_off) return p->P.fail(LTPL_ERR_INVALID_ARG, "planner: null input");
    return p->P.calc_paths_begin(in);
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
    return p->P.calc_vel_profile(in);
}
inline int api_get_paths(const ltpl_planner* p, int scen, ltpl_planner_paths_view* v) { return p ? p->P.get_paths(scen, v) : LTPL_ERR_INVALID_ARG; }
inline int api_get_trajectories(const ltpl_planner* p, int scen, ltpl_planner_traj_view* v) { return p ? p->P.get_trajectories(scen, v) : LTPL_ERR_INVALID_ARG; }

}  // namespace ltplp
