// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into or loaded by the product path.
//
// The fleet front end's semantics on the host: the planner state machine (graphbasedlocaltrajectoryplanner_amd/csrc/fleet_core.hpp, the
// source libltpl_hip.so runs with one wave per planner on DEVICE-resident state) through the product's host instantiation
// (csrc/planner_host.hpp) with the fleet's error behaviour -- every stage runs for every planner, a failing planner keeps its error word
// and its state until it gets a new start pose (`sticky_errors`) -- bound to the ORACLE's CPU arithmetic (oracle_compute.hpp). Purpose: the
// GPU-less build container replays the reference's tick recordings with the fleet's inputs (cars per planner, friction rows per planner)
// and error semantics (`-m "not gpu"`: "the host logic"). Exported names carry the prefix oracle_fleet_ so that they can never be mistaken
// for the product's ltpl_fleet_* symbols.
#include "oracle_compute.hpp"

namespace { std::string g_err; }

struct oracle_fleet { ltpl_planner* p = nullptr; };

extern "C" {
int oracle_fleet_create(const ltpl_lattice_desc* d, int max_path_nodes, int max_path_pts, const ltpl_planner_config* cfg, oracle_fleet** out)
{
    ltplp::HostLat lat;
    int rc = lat.init(d, max_path_nodes, max_path_pts, &g_err);
    if (rc) return rc;
    ltpl_planner* p = nullptr;
    if ((rc = ltplp::api_create(new OracleCompute(d), lat, cfg, &p, &g_err, /*sticky=*/true))) return rc;
    *out = new oracle_fleet(); (*out)->p = p;
    return LTPL_OK;
}
int oracle_fleet_destroy(oracle_fleet* f) { if (f) { delete f->p; delete f; } return LTPL_OK; }
int oracle_fleet_get_caps(const oracle_fleet* f, ltpl_planner_caps* c) { return f ? ltplp::api_get_caps(f->p, c) : LTPL_ERR_INVALID_ARG; }
const char* oracle_fleet_last_error(const oracle_fleet* f) { return f ? f->p->P.err.c_str() : g_err.c_str(); }
int oracle_fleet_set_start(oracle_fleet* f, int32_t p, double x, double y, double heading, double vel, double mho, int32_t* in_track, int32_t* cor_heading)
{
    return f ? f->p->P.set_start(p, x, y, heading, vel, mho, in_track, cor_heading) : LTPL_ERR_INVALID_ARG;
}
int oracle_fleet_calc_paths_begin(oracle_fleet* f, const ltpl_planner_paths_in* in) { return f ? ltplp::api_calc_paths_begin(f->p, in) : LTPL_ERR_INVALID_ARG; }
int oracle_fleet_calc_paths_finish(oracle_fleet* f, const int32_t* zo, const int32_t* zg) { return f ? ltplp::api_calc_paths_finish(f->p, zo, zg) : LTPL_ERR_INVALID_ARG; }
int oracle_fleet_calc_paths(oracle_fleet* f, const ltpl_planner_paths_in* in) { return f ? ltplp::api_calc_paths(f->p, in) : LTPL_ERR_INVALID_ARG; }
int oracle_fleet_get_ref_idx(oracle_fleet* f, const double* px, const double* py) { return (f && px && py) ? f->p->P.get_ref_idx(px, py) : LTPL_ERR_INVALID_ARG; }
int oracle_fleet_calc_vel_profile(oracle_fleet* f, const ltpl_planner_vel_in* in) { return f ? ltplp::api_calc_vel_profile(f->p, in) : LTPL_ERR_INVALID_ARG; }
int oracle_fleet_get_paths(const oracle_fleet* f, int32_t p, ltpl_planner_paths_view* v) { return f ? ltplp::api_get_paths(f->p, p, v) : LTPL_ERR_INVALID_ARG; }
int oracle_fleet_get_trajectories(const oracle_fleet* f, int32_t p, ltpl_planner_traj_view* v) { return f ? ltplp::api_get_trajectories(f->p, p, v) : LTPL_ERR_INVALID_ARG; }
}
