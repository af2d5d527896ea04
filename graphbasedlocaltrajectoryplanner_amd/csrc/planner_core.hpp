// planner_core.hpp -- what is HOST-ONLY by nature around the planner state machine (the state machine itself -- OTH.calc_paths, get_ref_idx,
// calc_vel_profile -- is fleet_core.hpp, instantiated for host memory by planner_host.hpp and for device memory by fleet_dev.hpp):
//
//   HostLat                host copy of the per-layer / per-node lattice tables
//   Compute                the two seams of include/ltpl_hip.h behind the host planner (HIP kernels in the product, the oracle in the harnesses)
//   set_initial_pose       OnlineTrajectoryHandler.set_initial_pose   graph_ltpl/online_graph/src/OnlineTrajectoryHandler.py:181-270
//                          (+ check_inside_bounds.py:7-59): the start spline from the pose into the lattice, once per run / restart
//   const_segment_test     main_online_path_gen.py:76-122 (behind ltpl_const_segment_test)
//   raceline_s             get_s_coord.py:8-121 on the race line (behind ltpl_raceline_s)
//
// Pure host C++ (no HIP types). O(n) projections of single points on polylines and one cubic: nothing here is on the path of a tick.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/ltpl_hip.h"
#include "host_prof.hpp"

namespace ltplp {

static const double kPi = 3.14159265358979323846;
static const double kInf = std::numeric_limits<double>::infinity();

// host copy of the per-layer / per-node lattice tables the state machine reads
struct HostLat {
    int L = 0, V = 0;
    bool closed = true;
    double lat_offset = 0, vel_decrease_lat = 0, veh_width = 0, veh_length = 0, sampled_resolution = 0;
    std::vector<int> layer_off, rl_idx;
    std::vector<double> s_rl, ref_x, ref_y, vel_rl, node_x, node_y, node_psi, race_x, race_y, nvx, nvy, w_right, w_left;
    int max_path_nodes = 0, max_path_pts = 0;

    int init(const ltpl_lattice_desc* d, int max_nodes, int max_pts, std::string* why)
    {
        L = d->num_layers; V = d->num_nodes; closed = d->closed != 0;
        lat_offset = d->lat_offset; vel_decrease_lat = d->vel_decrease_lat; veh_width = d->veh_width;
        veh_length = d->veh_length; sampled_resolution = d->sampled_resolution;
        max_path_nodes = max_nodes; max_path_pts = max_pts;
        layer_off.assign(d->layer_node_off, d->layer_node_off + L + 1);
        rl_idx.assign(d->raceline_index, d->raceline_index + L);
        s_rl.assign(d->s_raceline, d->s_raceline + L);
        ref_x.assign(d->refline_x, d->refline_x + L); ref_y.assign(d->refline_y, d->refline_y + L);
        vel_rl.assign(d->vel_raceline, d->vel_raceline + L);
        node_x.assign(d->node_x, d->node_x + V); node_y.assign(d->node_y, d->node_y + V);
        if (!d->raceline_x || !d->raceline_y || !d->node_psi) {
            *why = "planner: the lattice descriptor carries no raceline_x / raceline_y / node_psi columns";
            return LTPL_ERR_UNSUPPORTED;
        }
        race_x.assign(d->raceline_x, d->raceline_x + L); race_y.assign(d->raceline_y, d->raceline_y + L);
        node_psi.assign(d->node_psi, d->node_psi + V);
        if (d->normvec_x && d->normvec_y && d->width_right && d->width_left) {
            nvx.assign(d->normvec_x, d->normvec_x + L); nvy.assign(d->normvec_y, d->normvec_y + L);
            w_right.assign(d->width_right, d->width_right + L); w_left.assign(d->width_left, d->width_left + L);
        }
        return LTPL_OK;
    }
};

// the device arithmetic behind the state machine (seam (1) and seam (2) of include/ltpl_hip.h)
struct Compute {
    virtual int plan_paths(const ltpl_paths_in* in, ltpl_paths_out* out) = 0;
    virtual int vel_profile(const ltpl_vel_params* p, int n_jobs, const ltpl_vel_job* jobs, ltpl_vel_result* res) = 0;
    virtual const char* last_error() = 0;
    virtual ~Compute() {}
};

// ---------------------------------------------------------------------------------------------------------------------
// one point against one polyline (get_s_coord.py:8-121, closest_path_index.py:4-32)
// ---------------------------------------------------------------------------------------------------------------------
struct Poly {                       // strided view of a polyline's x / y columns
    const double* x; const double* y; int stride; int n;
    double px(int i) const { return x[(size_t)i * stride]; }
    double py(int i) const { return y[(size_t)i * stride]; }
};

inline int closest_index(const Poly& p, double qx, double qy)
{
    int best = 0; double bd = kInf;
    for (int i = 0; i < p.n; ++i) {
        const double dx = p.px(i) - qx, dy = p.py(i) - qy, d2 = dx * dx + dy * dy;
        if (d2 < bd) { bd = d2; best = i; }
    }
    return best;
}

inline double turn_angle(double ax, double ay, double bx, double by, double cx, double cy)
{
    double ang = std::atan2(cy - by, cx - bx) - std::atan2(ay - by, ax - bx);
    if (ang > kPi) ang -= 2 * kPi;
    else if (ang <= -kPi) ang += 2 * kPi;
    return ang;
}

struct Foot { double s; int i0, i1; };

// `s_at(i)` = the caller's s_array[i] (n_s entries). The reference prepends a zero when s_array[0] > 0.05
// (get_s_coord.py:67-68); a wrapped index -1 (closed lines) addresses the last entry of whichever array is in use.
template <class SFn>
inline Foot project_on_polyline(const Poly& p, double qx, double qy, bool closed, bool want_s, SFn s_at, int n_s)
{
    const int nb = closest_index(p, qx, qy);
    int i1, i2;
    if (closed) { i1 = nb - 1; i2 = nb + 1; if (i2 > p.n - 1) i2 = 0; }
    else { i1 = std::max(nb - 1, 0); i2 = std::min(nb + 1, p.n - 1); }
    const int i1p = i1 < 0 ? i1 + p.n : i1;                         // Python's negative index on the polyline
    const double a1 = std::fabs(turn_angle(p.px(nb), p.py(nb), qx, qy, p.px(i1p), p.py(i1p)));
    const double a2 = std::fabs(turn_angle(p.px(nb), p.py(nb), qx, qy, p.px(i2), p.py(i2)));
    Foot f; f.s = 0.0;
    if (want_s) {
        const bool first = a1 > a2;
        const int ia = first ? i1p : nb, ib = first ? nb : i2;
        const double ax = p.px(ia), ay = p.py(ia), bx = p.px(ib), by = p.py(ib);
        const double t = ((qx - ax) * (bx - ax) + (qy - ay) * (by - ay)) / ((bx - ax) * (bx - ax) + (by - ay) * (by - ay));
        const double fx = ax + t * (bx - ax), fy = ay + t * (by - ay);
        const double ds = std::sqrt((ax - fx) * (ax - fx) + (ay - fy) * (ay - fy));
        const bool shifted = n_s > 0 && s_at(0) > 0.05;
        const int n_ext = shifted ? n_s + 1 : n_s;
        int k = first ? i1 : nb;                                    // index into the (possibly extended) s array
        if (k < 0) k += n_ext;
        const double sv = shifted ? (k == 0 ? 0.0 : s_at(k - 1)) : s_at(k);
        f.s = sv + ds;
    }
    if (a1 >= a2) { f.i0 = i1; f.i1 = nb; } else { f.i0 = nb; f.i1 = i2; }
    return f;
}

// s coordinate of a position on the (closed) race line, one point per layer (get_s_coord with s_array = s_raceline)
inline double raceline_s(const HostLat& lat, double x, double y)
{
    Poly rl{lat.race_x.data(), lat.race_y.data(), 1, lat.L};
    return project_on_polyline(rl, x, y, true, true, [&](int i) { return lat.s_rl[(size_t)i]; }, lat.L).s;
}

// ---------------------------------------------------------------------------------------------------------------------
// constant-segment test in front of seam (1) (main_online_path_gen.py:76-122): is an object beside / on the part of the last
// path that stays constant? `seg` = rows [x, y, psi, kappa, el] of const_path_seg, `pos_est` = 2 doubles or null
// ---------------------------------------------------------------------------------------------------------------------
inline void const_segment_test(const HostLat& lat, const double* seg, int seg_rows, const double* pos_est, int n_veh,
                               const double* veh_x, const double* veh_y, const double* veh_radius, int* in_const, int* besides,
                               int* closest)
{
    *in_const = 0; *besides = 0; *closest = -1;
    if (!seg || seg_rows < 2) return;
    Poly rl{lat.race_x.data(), lat.race_y.data(), 1, lat.L};
    auto s_rl = [&](int i) { return lat.s_rl[(size_t)i]; };
    const double sx = pos_est ? pos_est[0] : seg[0], sy = pos_est ? pos_est[1] : seg[1];
    const double s_start = project_on_polyline(rl, sx, sy, true, true, s_rl, lat.L).s;
    const double s_end = project_on_polyline(rl, seg[(size_t)(seg_rows - 1) * 5], seg[(size_t)(seg_rows - 1) * 5 + 1], true, true, s_rl, lat.L).s;
    double smallest = kInf;
    for (int k = 0; k < n_veh; ++k) {
        const double s_obj = project_on_polyline(rl, veh_x[k], veh_y[k], true, true, s_rl, lat.L).s;
        if ((s_start <= s_obj && s_obj <= s_end) || (s_start > s_end && (s_obj > s_start || s_obj < s_end))) {
            *besides = 1;
            const double od = s_obj < s_start ? s_obj + lat.s_rl[(size_t)lat.L - 1] - s_start : s_obj - s_start;
            if (*closest < 0 || od < smallest) { *closest = k; smallest = od; }          // :96,113
            const double ref = std::pow(veh_radius[k] + lat.veh_width / 2, 2);
            for (int i = 0; i < seg_rows; ++i) {
                const double d2 = std::pow(seg[(size_t)i * 5] - veh_x[k], 2) + std::pow(seg[(size_t)i * 5 + 1] - veh_y[k], 2);
                if (d2 <= ref) { *in_const = 1; break; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the start pose
// ---------------------------------------------------------------------------------------------------------------------
#define LTPLP_NONE (-1)

struct Traj {                       // one entry of the reference's __last_action_set_* dicts (lists of length 1)
    int id = LTPL_ACT_NONE;
    std::vector<double> pp;         // rows [x, y, psi, kappa, el_length]
    std::vector<double> coeff;      // rows of 8
    std::vector<int> nodes;         // pairs [layer, node]; LTPLP_NONE = the reference's None
    std::vector<int> node_idx;
    bool red_len = false;
    int rows() const { return (int)(pp.size() / 5); }
    int n_nodes() const { return (int)(nodes.size() / 2); }
};

// -----------------------------------------------------------------------------------------------------------------
// check_inside_bounds.py:7-59 (start pose validation only)
// -----------------------------------------------------------------------------------------------------------------
inline bool inside_bounds(const HostLat& lat, double px, double py)
{
    const int L = lat.L;
    std::vector<double> b1x(L), b1y(L), b2x(L), b2y(L), cx(L), cy(L);
    for (int l = 0; l < L; ++l) {
        b1x[l] = lat.ref_x[l] + lat.nvx[l] * lat.w_right[l]; b1y[l] = lat.ref_y[l] + lat.nvy[l] * lat.w_right[l];
        b2x[l] = lat.ref_x[l] - lat.nvx[l] * lat.w_left[l];  b2y[l] = lat.ref_y[l] - lat.nvy[l] * lat.w_left[l];
        cx[l] = (b1x[l] + b2x[l]) / 2; cy[l] = (b1y[l] + b2y[l]) / 2;
    }
    Poly c{cx.data(), cy.data(), 1, L};
    const Foot f = project_on_polyline(c, px, py, true, false, [](int) { return 0.0; }, 0);
    const int a = f.i0 < 0 ? f.i0 + L : f.i0, b = f.i1;
    auto lin = [](double A, double B, int i) { return i == 49 ? B : (double)i * ((B - A) / 49.0) + A; };   // np.linspace(A, B, 50)
    int bi = 0; double bd = kInf;
    for (int i = 0; i < 50; ++i) {
        const double qx = lin(cx[a], cx[b], i) - px, qy = lin(cy[a], cy[b], i) - py, d2 = qx * qx + qy * qy;
        if (d2 < bd) { bd = d2; bi = i; }
    }
    const double l1x = lin(b1x[a], b1x[b], bi), l1y = lin(b1y[a], b1y[b], bi), l2x = lin(b2x[a], b2x[b], bi), l2y = lin(b2y[a], b2y[b], bi);
    const double d_track = (l1x - l2x) * (l1x - l2x) + (l1y - l2y) * (l1y - l2y);
    const double d1 = (l1x - px) * (l1x - px) + (l1y - py) * (l1y - py), d2 = (l2x - px) * (l2x - px) + (l2y - py) * (l2y - py);
    return !(d1 > d_track || d2 > d_track);
}

// what set_initial_pose leaves in the iterative memory (everything else is re-initialised, OTH.py:161-179)
struct StartPose {
    bool has_start = false; int start_node[2] = {0, 0};
    bool has_last = false; Traj last;       // the start spline as the only entry of the last action set ('straight')
    int action_forced = LTPL_ACT_NONE;
    double v_start = 0.0;
};

// -----------------------------------------------------------------------------------------------------------------
// OTH.set_initial_pose (OTH.py:181-270): start spline from the pose into the lattice
// -----------------------------------------------------------------------------------------------------------------
inline int set_initial_pose(const HostLat& lat, double x, double y, double heading, double vel, double max_heading_offset, int* in_track, int* cor_heading,
                            StartPose* out, std::string* why)
{
    StartPose& S = *out;
    S = StartPose();
    S.v_start = vel;
    *in_track = 1; *cor_heading = 1;
    if (lat.nvx.empty()) { *why = "planner: lattice without track bounds"; return LTPL_ERR_UNSUPPORTED; }
    if (!inside_bounds(lat, x, y)) { *in_track = 0; return LTPL_OK; }
    Poly nodes{lat.node_x.data(), lat.node_y.data(), 1, lat.V};
    const int gid = closest_index(nodes, x, y);                                    // GraphBase.get_closest_nodes, limit 1
    const int cl = (int)(std::upper_bound(lat.layer_off.begin(), lat.layer_off.end(), gid) - lat.layer_off.begin()) - 1;
    const int goal_layer = (cl + 2) % (lat.L - 1);                                 // OTH.py:226 (sic: L - 1)
    const int goal_node = lat.rl_idx[(size_t)goal_layer];
    S.start_node[0] = goal_layer; S.start_node[1] = goal_node; S.has_start = true;
    const int g = lat.layer_off[(size_t)goal_layer] + goal_node;
    const double ex = lat.node_x[(size_t)g], ey = lat.node_y[(size_t)g], epsi = lat.node_psi[(size_t)g];
    double hd = std::fabs(heading - epsi);
    if (hd > kPi) hd = std::fabs(2 * kPi - hd);
    if (hd > max_heading_offset) { *cor_heading = 0; return LTPL_OK; }
    // tph.calc_splines on two points = cubic Hermite segment with end slopes scaled by the chord length
    const double el = std::sqrt((ex - x) * (ex - x) + (ey - y) * (ey - y));
    double cxs[4], cys[4];
    {
        const double tx0 = std::cos(heading + kPi / 2) * el, ty0 = std::sin(heading + kPi / 2) * el;
        const double tx1 = std::cos(epsi + kPi / 2) * el, ty1 = std::sin(epsi + kPi / 2) * el;
        cxs[0] = x; cxs[1] = tx0; cxs[2] = 3.0 * (ex - x) - 2.0 * tx0 - tx1; cxs[3] = -2.0 * (ex - x) + tx0 + tx1;
        cys[0] = y; cys[1] = ty0; cys[2] = 3.0 * (ey - y) - 2.0 * ty0 - ty1; cys[3] = -2.0 * (ey - y) + ty0 + ty1;
    }
    // tph.calc_spline_lengths (15-point polyline), tph.interp_splines(stepsize_approx, incl_last_point=True)
    double len = 0.0;
    {
        double lx = cxs[0], ly = cys[0];
        for (int i = 1; i < 15; ++i) {
            const double t = i == 14 ? 1.0 : (double)i * (1.0 / 14.0);
            const double qx = cxs[0] + cxs[1] * t + cxs[2] * t * t + cxs[3] * t * t * t;
            const double qy = cys[0] + cys[1] * t + cys[2] * t * t + cys[3] * t * t * t;
            len += std::sqrt((qx - lx) * (qx - lx) + (qy - ly) * (qy - ly));
            lx = qx; ly = qy;
        }
    }
    const int n = (int)std::ceil(len / lat.sampled_resolution) + 1;
    Traj T; T.id = LTPL_ACT_STRAIGHT;
    T.pp.assign((size_t)n * 5, 0.0);
    const double step = n > 1 ? len / (double)(n - 1) : 0.0;
    for (int i = 0; i < n; ++i) {
        double t = (i == n - 1) ? 1.0 : ((double)i * step) / len;
        double px_ = cxs[0] + cxs[1] * t + cxs[2] * t * t + cxs[3] * t * t * t;
        double py_ = cys[0] + cys[1] * t + cys[2] * t * t + cys[3] * t * t * t;
        if (i == n - 1) { px_ = ((cxs[0] + cxs[1]) + cxs[2]) + cxs[3]; py_ = ((cys[0] + cys[1]) + cys[2]) + cys[3]; }
        const double xd = cxs[1] + 2 * cxs[2] * t + 3 * cxs[3] * t * t, yd = cys[1] + 2 * cys[2] * t + 3 * cys[3] * t * t;
        const double xdd = 2 * cxs[2] + 6 * cxs[3] * t, ydd = 2 * cys[2] + 6 * cys[3] * t;
        double psi = std::atan2(yd, xd) - kPi / 2;
        { const double sg = psi > 0 ? 1.0 : (psi < 0 ? -1.0 : 0.0); psi = sg * std::fmod(std::fabs(psi), 2 * kPi); if (psi >= kPi) psi -= 2 * kPi; else if (psi < -kPi) psi += 2 * kPi; }
        const double q = xd * xd + yd * yd;
        double* r = &T.pp[(size_t)i * 5];
        r[0] = px_; r[1] = py_; r[2] = psi; r[3] = (xd * ydd - yd * xdd) / std::pow(q, 1.5); r[4] = 0.0;
    }
    for (int i = 0; i + 1 < n; ++i) {
        const double dx = T.pp[(size_t)(i + 1) * 5] - T.pp[(size_t)i * 5], dy = T.pp[(size_t)(i + 1) * 5 + 1] - T.pp[(size_t)i * 5 + 1];
        T.pp[(size_t)i * 5 + 4] = std::sqrt(dx * dx + dy * dy);
    }
    T.coeff.assign(8, 0.0);
    for (int k = 0; k < 4; ++k) { T.coeff[(size_t)k] = cxs[k]; T.coeff[(size_t)4 + k] = cys[k]; }
    T.nodes = {LTPLP_NONE, LTPLP_NONE, goal_layer, goal_node};
    T.node_idx = {0, n - 1};
    S.last = T; S.has_last = true;
    S.action_forced = LTPL_ACT_STRAIGHT;
    return LTPL_OK;
}

}  // namespace ltplp
