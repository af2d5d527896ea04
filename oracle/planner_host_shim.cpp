// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into or loaded by the product path.
//
// The product's host planner (graphbasedlocaltrajectoryplanner_amd/csrc/planner_host.hpp: the planner state machine of fleet_core.hpp --
// the C++ restatement of the reference's OnlineTrajectoryHandler -- on host memory) bound to the ORACLE's CPU arithmetic
// (oracle_compute.hpp) instead of the HIP kernels. Purpose: the GPU-less build container can drive the host logic through the
// recorded closed loops of tests/golden/*_ticks.npz (`-m "not gpu"` tests cover "the host logic"). On the GPU box the same
// recordings run through libltpl_hip.so (`-m gpu`). The exported names carry the prefix oracle_planner_ so that they can
// never be mistaken for the product's ltpl_planner_* symbols.
#include "oracle_compute.hpp"

namespace {
std::string g_err;
}

extern "C" {
int oracle_planner_create(const ltpl_lattice_desc* d, int max_path_nodes, int max_path_pts, const ltpl_planner_config* cfg,
                          ltpl_planner** out)
{
    ltplp::HostLat lat;
    int rc = lat.init(d, max_path_nodes, max_path_pts, &g_err);
    if (rc) return rc;
    return ltplp::api_create(new OracleCompute(d), lat, cfg, out, &g_err);
}
int oracle_const_segment_test(const ltpl_lattice_desc* d, const double* seg, int32_t n_rows, const double* pos_est, int32_t n_veh,
                              const double* veh_x, const double* veh_y, const double* veh_radius, int32_t* flags_out, int32_t* closest_out)
{
    ltplp::HostLat lat;
    if (lat.init(d, 0, 0, &g_err)) return LTPL_ERR_UNSUPPORTED;
    int in_const, besides, closest;
    ltplp::const_segment_test(lat, n_rows > 0 ? seg : nullptr, n_rows, pos_est, n_veh, veh_x, veh_y, veh_radius, &in_const, &besides, &closest);
    *flags_out = (in_const ? LTPL_FLAG_OBJ_IN_CONST : 0) | (besides ? LTPL_FLAG_OBJ_BESIDES : 0);
    *closest_out = closest;
    return LTPL_OK;
}
int oracle_raceline_s(const ltpl_lattice_desc* d, double x, double y, double* s_out)
{
    ltplp::HostLat lat;
    if (lat.init(d, 0, 0, &g_err)) return LTPL_ERR_UNSUPPORTED;
    *s_out = ltplp::raceline_s(lat, x, y);
    return LTPL_OK;
}
int oracle_planner_destroy(ltpl_planner* p) { delete p; return LTPL_OK; }
int oracle_planner_get_caps(const ltpl_planner* p, ltpl_planner_caps* c) { return ltplp::api_get_caps(p, c); }
const char* oracle_planner_last_error(const ltpl_planner* p) { return p ? p->P.err.c_str() : g_err.c_str(); }
int oracle_planner_set_start(ltpl_planner* p, int32_t scen, double x, double y, double heading, double vel, double mho,
                             int32_t* in_track, int32_t* cor_heading)
{
    return p ? p->P.set_start(scen, x, y, heading, vel, mho, in_track, cor_heading) : LTPL_ERR_INVALID_ARG;
}
int oracle_planner_calc_paths(ltpl_planner* p, const ltpl_planner_paths_in* in) { return ltplp::api_calc_paths(p, in); }
int oracle_planner_calc_paths_begin(ltpl_planner* p, const ltpl_planner_paths_in* in) { return ltplp::api_calc_paths_begin(p, in); }
int oracle_planner_calc_paths_finish(ltpl_planner* p, const int32_t* zo, const int32_t* zg) { return ltplp::api_calc_paths_finish(p, zo, zg); }
int oracle_planner_get_ref_idx(ltpl_planner* p, const double* px, const double* py) { return (p && px && py) ? p->P.get_ref_idx(px, py) : LTPL_ERR_INVALID_ARG; }
int oracle_planner_calc_vel_profile(ltpl_planner* p, const ltpl_planner_vel_in* in) { return ltplp::api_calc_vel_profile(p, in); }
int oracle_planner_get_paths(const ltpl_planner* p, int32_t scen, ltpl_planner_paths_view* v) { return ltplp::api_get_paths(p, scen, v); }
int oracle_planner_get_trajectories(const ltpl_planner* p, int32_t scen, ltpl_planner_traj_view* v) { return ltplp::api_get_trajectories(p, scen, v); }
}
