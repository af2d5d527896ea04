// fleet_core.hpp -- the iterative memory of the reference's OnlineTrajectoryHandler for a FLEET of planners whose state lives in
// device memory (the caller of the hot path, "what comes next"). Same state machine as planner_core.hpp
// (which stays the product's host form for single planners and is pinned to the tick recordings), restated as wave-uniform SPMD code
// over fixed-capacity plain-data state so that ONE source compiles for
//   * the device: one wave64 per planner (ltpl_hip.hip, exec policy WaveX) -- the scalar control flow is executed redundantly by all
//     lanes on a private copy of the planner's scalars, the row loops (copies, projections, searches) are strided over the lanes;
//   * the host: exec policy HostX (one "lane"), used by the CPU test harness oracle/fleet_host_shim.cpp to replay the reference's
//     tick recordings through exactly this code without a GPU.
//
//   paths_pre   OnlineTrajectoryHandler.calc_paths, part in front of seam (1)   OTH.py:308-414 (+ main_online_path_gen.py:76-122)
//   paths_post  ... behind seam (1): stitch the new paths behind the constant part  OTH.py:429-513
//   ref_idx     OnlineTrajectoryHandler.get_ref_idx                              OTH.py:518-601
//   vel_a/b/c/d OnlineTrajectoryHandler.calc_vel_profile in four stages around the three launches of seam (2)   OTH.py:603-1040
//
// Restrictions (reported, not emulated): fixed capacities (rows / nodes per stitched path, planner_caps), errors are per-planner codes.
// Round 4: location dependent friction (local_gg as a dict of per-path rows, OTH.py:649-666) is part of the state machine -- the host
// planner (ltpl_planner_*) is the one-lane instantiation of THIS source (planner_host.hpp), not a second implementation.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>

#include "../../include/ltpl_hip.h"

#if defined(__HIPCC__)
#define FLT_FN __host__ __device__ inline
#else
#define FLT_FN inline
#endif

namespace fleet {

static constexpr double kPi = 3.14159265358979323846;
static constexpr int KEYS = LTPL_MAX_ACTIONS;       // primitives per tick
static constexpr int BPS = LTPL_PLANNER_MAX_KEYS;   // + 'emergency'
static constexpr int JOBS_A = 5;                    // stage-A jobs per planner: follow (2) + its reduced-length profile (1) + two more keys
static constexpr int CALC_BUF = 16;                 // cap of calc_time_buffer_len
#define FLT_NONE (-1)
FLT_FN double inf() { return HUGE_VAL; }

// error sites (S.err = LTPL_ERR_* | site << 8)
enum Site { E_BACKUP_KEY = 1, E_NO_START, E_CAP_ROWS, E_CAP_NODES, E_CUT_LAYER, E_BRAKE_PREFIX, E_FOLLOW_EMPTY, E_NO_NODES, E_END_NONE,
            E_FOLLOW_SHORT, E_VX_SHORT, E_ROW5, E_BACKUP_CUT, E_BACKUP_SHORT, E_BACKUP_LEN, E_EMERG_EMPTY, E_CALC_BUF, E_GG_DICT, E_CAP_JOBS, E_NO_RANGE, E_CAP_VEL,
            E_GG_ROWS, E_EMERG_GG };

// ---------------------------------------------------------------------------------------------------------------------
// plain-data state
// ---------------------------------------------------------------------------------------------------------------------
struct TrajM {                      // one entry of __last_action_set_* (windows into the slot's arrays: trimming moves the window)
    int id, red_len;
    int r0, rows;                   // path_param rows [x, y, psi, kappa, el]
    int c0, nc;                     // coeff rows of 8
    int n0, nn;                     // node pairs [layer, node]
    int i0, ni;                     // node indices
    double gax, gay;                // __last_action_set_path_gg: constant rows ...
    int has_gg, pad_;               // ... or (location dependent friction) one row per path coordinate in the slot's gg array, same window as the rows
};

struct Work {                       // per key of the current tick, stages A -> C
    int n, c0, vel_idx, pref_idx, v_idx, cut_index_layer;
    int job_follow, job_free, job_fb, job_backup;
    int generic, has_fb, too_close, vel_bound, keep, drop, empty;
    int gv_n, pad_;                 // rows of this key's friction rows from the cut on (0: constant friction), kept in Block::gv for stage C
    double vel_start;
};

struct PlannerS {
    int traj_base_id;
    int has_start, start_node[2];
    int has_last, n_last, last_slot[KEYS], cur_set;            // __last_action_set_*: slots of set `cur_set`, dict order
    int has_bp, n_bp, bp_slot[BPS], bp_id[BPS], bp_traj_id[BPS], bp_rows[BPS];      // __last_bp_action_set
    int has_stamp; double last_stamp;
    int last_cut_idx;
    int has_pos; double pos_est[2];
    int em_base_id;
    int has_backup, backup_set, backup_slot;                   // __backup_*: the previous tick's straight / follow entry (other set)
    int n_calc; double calc_buffer[CALC_BUF];
    double v_start;
    int action_forced, closest_obj_index;
    int has_old_gg; double old_gg_scale;
    int sel_action, raw_action, const_exists, const_rows, loc_path_start_idx, start_node_idx;
    int cut_index_pos, cut_layer; double vel_plan, acc_plan; int n_vel_course, ref_done;
    int n_ids, id_key[BPS], id_val[BPS];
    int err;
    int n_work;
    Work w[KEYS];
    TrajM tm[2][KEYS];
};

struct Dims {
    int N, R, CN, cn, cp;
    int RV;                         // points per velocity job (LDS-resident solver): cp + 64 <= R
    int SR;                         // doubles per scratch row: max(R, 2 CN) -- paths_pre keeps the node coordinates of a path (x | y) in one
    size_t stride;                  // bytes per planner block
    size_t o_traj, traj_bytes, o_bp, o_velc, o_sarr, o_vx, o_scr;
    // Friction rows (local_gg as a dict) live in their OWN array, allocated when the first call carries rows: [ax, ay] per path row for the
    // 2 x KEYS trajectory slots + KEYS windows "from the cut on" for stage C. Inside the planner block they made every fleet's state
    // 57 % larger (Monteblanco: 235 -> 370 KB per planner) and the fleet tick 5 % slower whether rows were ever passed or not (r04h).
    size_t gg_slot, gg_stride;
    FLT_FN size_t o_pp() const { return 0; }
    FLT_FN size_t o_coeff() const { return sizeof(double) * (size_t)R * 5; }
    FLT_FN size_t o_nodes() const { return o_coeff() + sizeof(double) * (size_t)CN * 8; }
    FLT_FN size_t o_nidx() const { return o_nodes() + sizeof(int) * (size_t)CN * 2; }
};

FLT_FN size_t align256(size_t x) { return (x + 255) / 256 * 256; }

inline Dims make_dims(int N, int max_path_nodes, int max_path_pts)
{
    Dims D;
    D.N = N; D.cn = max_path_nodes; D.cp = max_path_pts;
    D.R = 2 * max_path_pts + 64; D.CN = 2 * max_path_nodes + 8;            // = ltpl_planner_caps
    D.RV = max_path_pts + 64;
    D.SR = D.R > 2 * D.CN ? D.R : 2 * D.CN;
    D.traj_bytes = align256(D.o_nidx() + sizeof(int) * (size_t)D.CN);
    D.gg_slot = align256(sizeof(double) * (size_t)D.R * 2); D.gg_stride = D.gg_slot * 3 * KEYS;
    size_t o = align256(sizeof(PlannerS));
    D.o_traj = o; o += D.traj_bytes * 2 * KEYS;
    D.o_bp = o; o += align256(sizeof(double) * (size_t)D.R * 7) * BPS;
    D.o_velc = o; o += align256(sizeof(double) * (size_t)D.R);
    D.o_sarr = o; o += align256(sizeof(double) * (size_t)D.R) * KEYS;
    D.o_vx = o; o += align256(sizeof(double) * (size_t)D.R) * 2;
    D.o_scr = o; o += align256(sizeof(double) * (size_t)D.SR) * 2;
    D.stride = o;
    return D;
}

// Row tables of the state (path_param: 5 columns, trajectories: 7 columns) are stored COLUMN-major (column stride = R rows): the lanes of
// a wave walk rows, so every access is a contiguous run of one column. Row-major rows of 40 / 56 bytes made each load / store instruction
// touch 40-56 cache lines for 64 lanes (k_fleet_vel_b was bound by the L2 request rate of its trajectory stores). A window starting at row r
// is `from(r)`.
struct Rows {
    double* p; int ld;
    FLT_FN double& at(int i, int c) const { return p[(size_t)c * ld + i]; }
    FLT_FN Rows from(int r) const { return Rows{p + r, ld}; }
    FLT_FN const double* col(int c) const { return p + (size_t)c * ld; }
};

struct Block {                      // one planner's memory (`g`: its friction rows, null until a call carries rows)
    unsigned char* b; Dims D; unsigned char* g;
    FLT_FN PlannerS* S() const { return reinterpret_cast<PlannerS*>(b); }
    FLT_FN unsigned char* slot(int set, int k) const { return b + D.o_traj + D.traj_bytes * (size_t)(set * KEYS + k); }
    FLT_FN Rows pp(int set, int k) const { return Rows{reinterpret_cast<double*>(slot(set, k)), D.R}; }
    FLT_FN double* coeff(int set, int k) const { return reinterpret_cast<double*>(slot(set, k) + D.o_coeff()); }
    FLT_FN int* nodes(int set, int k) const { return reinterpret_cast<int*>(slot(set, k) + D.o_nodes()); }
    FLT_FN int* nidx(int set, int k) const { return reinterpret_cast<int*>(slot(set, k) + D.o_nidx()); }
    FLT_FN Rows gg(int set, int k) const { return Rows{reinterpret_cast<double*>(g + D.gg_slot * (size_t)(set * KEYS + k)), D.R}; }
    FLT_FN Rows gv(int k) const { return Rows{reinterpret_cast<double*>(g + D.gg_slot * (size_t)(2 * KEYS + k)), D.R}; }
    FLT_FN Rows bp(int k) const { return Rows{reinterpret_cast<double*>(b + D.o_bp + align256(sizeof(double) * (size_t)D.R * 7) * (size_t)k), D.R}; }
    FLT_FN double* velc() const { return reinterpret_cast<double*>(b + D.o_velc); }
    FLT_FN double* sarr(int k) const { return reinterpret_cast<double*>(b + D.o_sarr + align256(sizeof(double) * (size_t)D.R) * (size_t)k); }
    FLT_FN double* vx(int k) const { return reinterpret_cast<double*>(b + D.o_vx + align256(sizeof(double) * (size_t)D.R) * (size_t)k); }
    FLT_FN double* scr(int k) const { return reinterpret_cast<double*>(b + D.o_scr + align256(sizeof(double) * (size_t)D.SR) * (size_t)k); }
};

// lattice tables the state machine reads (device pointers on the device, host vectors in the harness)
struct FLat {
    int L, V, closed;
    double lat_offset, vel_decrease_lat, veh_width, veh_length, sampled_resolution;
    const int* layer_off; const int* rl_idx;
    const double* s_rl; const double* vel_rl; const double* node_x; const double* node_y; const double* race_x; const double* race_y;
};

struct FCfg {
    double v_max_offset, delaycomp, calc_time_safety; int calc_time_buffer_len, filt_window_width;
};

// per-tick inputs of calc_paths (ltpl_planner_paths_in, resident until the next calc_paths: the velocity stage reads the objects again)
struct FObj { const int* prev_action; const double* t_now; const int* veh_off; const int* pos_off; const double* radius; const double* vel;
              const double* px; const double* py; };
// seam (1): the arrays the path kernel reads / writes (ltpl_paths_in / ltpl_paths_out), one entry per planner
struct FPathsIn { int* start_layer; int* start_node; int* flags; int* last_action; int* const_closest; double* psi_s;
                  int* n_last; int* last_layer; int* last_node; };
struct FPathsOut { const int* closest_obj_index; const int* n_actions; const int* action_id; const int* valid; const int* reduced;
                   const int* n_nodes; const int* n_pts; const int* nodes; const int* node_idx; const double* coeff; const double* pp; };
// arguments of calc_vel_profile (ltpl_planner_vel_in)
struct FVelIn { const double* pos_x; const double* pos_y; const double* vel_est; const double* vel_max; const double* gg_scale;
                const double* gg_ax; const double* gg_ay; const double* safety_d; const int* incl_emerg;
                // ABI v6: machine tables per planner (null: one table for the call): first row of every table, table of every planner
                const int* ax_off; const int* ax_idx;
                // location dependent friction (local_gg as a dict, OTH.py:649-666; null: the constant tuple): rows gg_off[p * MK + k] ..
                // gg_off[p * MK + k + 1] of gg_rows ([ax, ay] per path coordinate) belong to planner p's k-th path key, MK = LTPL_PLANNER_MAX_KEYS
                const int* gg_off; const double* gg_rows; };

// seam (2): job table + pooled arrays (the layout k_vel_profile reads). Job slot j owns 4 R doubles of `pool` (kappa R | el R | gg 2 R)
// and R doubles of `out`; an unused slot has n = 0.
struct VelJob {
    int mode, n, n_el, has_v_end;
    int off_kappa, off_el, off_gg, off_out;
    double v_start, v_end, v_ego, v_obj, safety_d, obj_dist, obj_x, obj_y;
    // THE CAR of the job (ABI v6: a fleet of different cars): Graph_LTPL.calc_vel_profile's vel_max and ax_max_machines are arguments per
    // call = per vehicle (Graph_LTPL.py:344-351). v_max <= 0 / n_axm == 0: the parameter set of the launch.
    double v_max;
    int axm_off, n_axm;             // the job's machine table: rows [axm_off, axm_off + n_axm) of the call's stacked tables
    int gg_rows, lane_form;         // gg_rows 1: the friction limits differ from row to row (pool form, [ax, ay] per row); 0: gg[0], gg[1] for every row.
                                    // lane_form 1: the operands lie in the lane plane (FJobs::ke), not in the pool: the job belongs to a lane kernel
};
struct JobCar { double v_max; int axm_off, n_axm; };
// friction limits of a job's rows: constant (ax, ay) or rows of a [ax, ay] table (window `g`, first row `base`), times `scale`
struct GgSrc {
    Rows g; double ax, ay, scale;
    FLT_FN bool rows() const { return g.p != nullptr; }
    FLT_FN double at(int i, int c) const { return g.p ? g.at(i, c) * scale : (c ? ay : ax) * scale; }
};
FLT_FN GgSrc gg_const(double ax, double ay, double scale = 1.0) { return GgSrc{Rows{nullptr, 0}, ax, ay, scale}; }
FLT_FN GgSrc gg_table(const Rows& g, double scale = 1.0) { return GgSrc{g, 0.0, 0.0, scale}; }
FLT_FN JobCar car_of(const FVelIn& vin, int p)
{
    JobCar c{vin.vel_max[p], 0, 0};
    if (vin.ax_off && vin.ax_idx) { const int t = vin.ax_idx[p]; c.axm_off = vin.ax_off[t]; c.n_axm = vin.ax_off[t + 1] - vin.ax_off[t]; }
    return c;
}
// Lane plane (device only, stage-A table): the forward-backward jobs of slots >= 1 are solved one LANE per job (k_fleet_fb_lanes), so their
// operands go into a plane tiled by job and blocked by rows, (|kappa|, element length) as an fp64 pair (`ke`, layout = kep_base / kep_row of
// the batch velocity stage); plane index q = p (per_planner - 1) + slot - 1. The results come back job-major in `out` like every other
// job's. Null: operands through `pool` (host harness, follow and brake jobs).
#ifndef LTPL_VEL_F32_OPERANDS
typedef double ke_scalar_f;         // operand records of the lane kernels: fp64 pairs like ke_t in ltpl_hip.hip (the reference is fp64 throughout)
#else
typedef float ke_scalar_f;          // (the rounds-3..5 form, opt-in)
#endif
struct F2 { ke_scalar_f x, y; };
struct FJobs { VelJob* jobs; double* pool; const double* out; const int* flags; int per_planner; F2* ke; int ke_rows;
               int ke_follow; };    // >= 0: the follow jobs (slot 0) without friction rows go to the lane plane as well, job index ke_follow + p (round 5)
#ifndef LTPL_KE_RB
#define LTPL_KE_RB 8                // rows per block of the operand planes: the layout of kep_base / kep_row (paths_team.hpp)
#endif
FLT_FN unsigned kep_base_f(int job, int plane_rows) { return ((unsigned)(job >> 6) * ((unsigned)plane_rows / LTPL_KE_RB) * 64u + (unsigned)(job & 63)) * LTPL_KE_RB; }
FLT_FN unsigned kep_row_f(int r) { return ((unsigned)r / LTPL_KE_RB) * (64u * LTPL_KE_RB) + ((unsigned)r % LTPL_KE_RB); }
// result i of the job in `slot` of planner p
FLT_FN double job_out(const FJobs& J, const Dims& D, int p, int slot, int i) { return J.out[(size_t)(p * J.per_planner + slot) * D.R + i]; }

// ---------------------------------------------------------------------------------------------------------------------
// exec policy of the host build: one lane
// ---------------------------------------------------------------------------------------------------------------------
struct HostX {
    static constexpr int W = 1;
    int lane() const { return 0; }
    void sync() const {}
    void argmin(double&, int&) const {}
    bool any(bool b) const { return b; }
    template <class P> int find_first(int n, P pred) const { for (int i = 0; i < n; ++i) if (pred(i)) return i; return n; }
    // out[i] = term(0) + ... + term(i) in this (sequential) order
    template <class T> void scan_seq(int n, T term, double* out) const { double acc = 0.0; for (int i = 0; i < n; ++i) { acc += term(i); out[i] = acc; } }
};

// ---------------------------------------------------------------------------------------------------------------------
// one point against one polyline (get_s_coord.py:8-121, closest_path_index.py:4-32)
// ---------------------------------------------------------------------------------------------------------------------
struct Poly { const double* x; const double* y; int stride; int n;
              FLT_FN double px(int i) const { return x[(size_t)i * stride]; } FLT_FN double py(int i) const { return y[(size_t)i * stride]; } };

template <class X>
FLT_FN int closest_index(const X& x, const Poly& p, double qx, double qy)
{
    double bd = inf(); int best = 0x7fffffff;
    for (int i = x.lane(); i < p.n; i += X::W) {
        const double dx = p.px(i) - qx, dy = p.py(i) - qy, d2 = dx * dx + dy * dy;
        if (d2 < bd) { bd = d2; best = i; }
    }
    x.argmin(bd, best);
    return best == 0x7fffffff ? 0 : best;
}

FLT_FN double turn_angle(double ax, double ay, double bx, double by, double cx, double cy)
{
    double ang = atan2(cy - by, cx - bx) - atan2(ay - by, ax - bx);
    if (ang > kPi) ang -= 2 * kPi;
    else if (ang <= -kPi) ang += 2 * kPi;
    return ang;
}

struct Foot { double s; int i0, i1; };

// get_s_coord.py:34-47 picks the neighbour of the closest point by comparing |angle3pt(nb, pos, neighbour)| of the two neighbours. The wrapped
// difference of two atan2 values is the angle between u = nb - pos and v = neighbour - pos in [0, pi], and the cosine is monotone there:
// |ang1| > |ang2|  <=>  u.v1 / |v1| < u.v2 / |v2|  (|u| cancels). Two dot products and two square roots instead of four atan2 -- a projection
// is a dependent chain in these kernels and fp64 atan2 is ~1 us of it. Degenerate vectors (pos on a polyline point) take the atan2 form.
// Returns -1 / 0 / +1 for |ang1| < / == / > |ang2|. (Same rule as angle_order_dev of the velocity kernels.)
// (out of line: four inlined fp64 atan2 per projection set the register budget of the kernels that project, for a branch that is taken
// when a query sits exactly on a polyline point)
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline))
#endif
inline int angle_order_atan2(double nx, double ny, double px, double py, double x1, double y1, double x2, double y2)
{
    const double a1 = fabs(turn_angle(nx, ny, px, py, x1, y1)), a2 = fabs(turn_angle(nx, ny, px, py, x2, y2));
    return a1 > a2 ? 1 : (a1 < a2 ? -1 : 0);
}
FLT_FN int angle_order(double nx, double ny, double px, double py, double x1, double y1, double x2, double y2)
{
    const double ux = nx - px, uy = ny - py, v1x = x1 - px, v1y = y1 - py, v2x = x2 - px, v2y = y2 - py;
    const double n1 = v1x * v1x + v1y * v1y, n2 = v2x * v2x + v2y * v2y, nu = ux * ux + uy * uy;
    if (!(n1 > 0.0) || !(n2 > 0.0) || !(nu > 0.0)) return angle_order_atan2(nx, ny, px, py, x1, y1, x2, y2);
    const double c1 = (ux * v1x + uy * v1y) * sqrt(n2), c2 = (ux * v2x + uy * v2y) * sqrt(n1);
    return c1 < c2 ? 1 : (c1 > c2 ? -1 : 0);
}

// `s_arr[i * s_stride]` = the caller's s_array (n_s entries); see planner_core.hpp project_on_polyline for the index rules
template <class X>
FLT_FN Foot project_on_polyline(const X& x, const Poly& p, double qx, double qy, bool closed, bool want_s, const double* s_arr, int s_stride, int n_s)
{
    const int nb = closest_index(x, p, qx, qy);
    int i1, i2;
    if (closed) { i1 = nb - 1; i2 = nb + 1; if (i2 > p.n - 1) i2 = 0; }
    else { i1 = nb - 1 > 0 ? nb - 1 : 0; i2 = nb + 1 < p.n - 1 ? nb + 1 : p.n - 1; }
    const int i1p = i1 < 0 ? i1 + p.n : i1;
    const int ord = angle_order(p.px(nb), p.py(nb), qx, qy, p.px(i1p), p.py(i1p), p.px(i2), p.py(i2));      // |a1| vs |a2|
    Foot f; f.s = 0.0;
    if (want_s) {
        const bool first = ord > 0;
        const int ia = first ? i1p : nb, ib = first ? nb : i2;
        const double ax = p.px(ia), ay = p.py(ia), bx = p.px(ib), by = p.py(ib);
        const double t = ((qx - ax) * (bx - ax) + (qy - ay) * (by - ay)) / ((bx - ax) * (bx - ax) + (by - ay) * (by - ay));
        const double fx = ax + t * (bx - ax), fy = ay + t * (by - ay);
        const double ds = sqrt((ax - fx) * (ax - fx) + (ay - fy) * (ay - fy));
        const bool shifted = n_s > 0 && s_arr[0] > 0.05;
        const int n_ext = shifted ? n_s + 1 : n_s;
        int k = first ? i1 : nb;
        if (k < 0) k += n_ext;
        const double sv = shifted ? (k == 0 ? 0.0 : s_arr[(size_t)(k - 1) * s_stride]) : s_arr[(size_t)k * s_stride];
        f.s = sv + ds;
    }
    if (ord >= 0) { f.i0 = i1; f.i1 = nb; } else { f.i0 = nb; f.i1 = i2; }
    return f;
}

// constant-segment test in front of seam (1) (main_online_path_gen.py:76-122); seg = rows [x, y, psi, kappa, el]
template <class X>
FLT_FN void const_segment_test(const X& x, const FLat& lat, bool has_seg, const Rows& seg, int seg_rows, const double* pos_est, int n_veh,
                               const int* pos_off, const double* px, const double* py, const double* radius, int* in_const, int* besides, int* closest)
{
    *in_const = 0; *besides = 0; *closest = -1;
    if (!has_seg || seg_rows < 2) return;
    const Poly rl{lat.race_x, lat.race_y, 1, lat.L};
    const double sx = pos_est ? pos_est[0] : seg.at(0, 0), sy = pos_est ? pos_est[1] : seg.at(0, 1);
    const double s_start = project_on_polyline(x, rl, sx, sy, true, true, lat.s_rl, 1, lat.L).s;
    const double s_end = project_on_polyline(x, rl, seg.at(seg_rows - 1, 0), seg.at(seg_rows - 1, 1), true, true, lat.s_rl, 1, lat.L).s;
    double smallest = inf();
    for (int k = 0; k < n_veh; ++k) {
        const double vx = px[pos_off[k]], vy = py[pos_off[k]];
        const double s_obj = project_on_polyline(x, rl, vx, vy, true, true, lat.s_rl, 1, lat.L).s;
        if ((s_start <= s_obj && s_obj <= s_end) || (s_start > s_end && (s_obj > s_start || s_obj < s_end))) {
            *besides = 1;
            const double od = s_obj < s_start ? s_obj + lat.s_rl[lat.L - 1] - s_start : s_obj - s_start;
            if (*closest < 0 || od < smallest) { *closest = k; smallest = od; }
            const double rr = radius[k] + lat.veh_width / 2, ref = rr * rr;
            bool hit = false;
            for (int i = x.lane(); i < seg_rows; i += X::W) {
                const double dx = seg.at(i, 0) - vx, dy = seg.at(i, 1) - vy;
                if (dx * dx + dy * dy <= ref) hit = true;
            }
            if (x.any(hit)) *in_const = 1;
        }
    }
}

FLT_FN int find_last(const PlannerS& S, int id) { for (int i = 0; i < S.n_last; ++i) if (S.tm[S.cur_set][S.last_slot[i]].id == id) return S.last_slot[i]; return -1; }
FLT_FN int find_bp(const PlannerS& S, int id) { for (int i = 0; i < S.n_bp; ++i) if (S.bp_id[i] == id) return i; return -1; }
FLT_FN int fail(PlannerS& S, int code, int site) { if (!S.err) S.err = code | (site << 8); return code; }

// ---------------------------------------------------------------------------------------------------------------------
// OTH.update_objects + OTH.calc_paths in front of seam (1) (OTH.py:272-287, 308-414)
// ---------------------------------------------------------------------------------------------------------------------
template <class X>
FLT_FN void paths_pre(const X& x, const FLat& lat, const FCfg& cfg, const Block& B, PlannerS& S, int p, const FObj& ob, const FPathsIn& pin)
{
    const int prev_action = ob.prev_action[p];
    const double t_now = ob.t_now[p];
    S.closest_obj_index = -1;
    int sel = prev_action;
    S.raw_action = prev_action;
    if (sel == LTPL_ACT_EMERGENCY) sel = S.em_base_id;
    if (S.action_forced != LTPL_ACT_NONE) { sel = S.action_forced; S.action_forced = LTPL_ACT_NONE; }
    S.sel_action = sel;
    const int set = S.cur_set;
    const int lsel = S.has_last ? find_last(S, sel) : -1;
    const int bsel = S.has_bp ? find_bp(S, sel) : -1;
    S.const_exists = lsel >= 0;
    const bool planned_once = S.has_stamp != 0;
    const bool valid_last = planned_once && S.const_exists && bsel >= 0 && S.bp_rows[bsel] > 2;
    if (valid_last) {                                                          // backup plan (:329-344)
        int b = find_last(S, LTPL_ACT_FOLLOW);
        if (b < 0) b = find_last(S, LTPL_ACT_STRAIGHT);
        if (b < 0) { fail(S, LTPL_ERR_INVALID_ARG, E_BACKUP_KEY); return; }
        S.has_backup = 1; S.backup_set = set; S.backup_slot = b;
    } else S.has_backup = 0;

    int last_from = -1;                                                        // last solution = nodes of lsel from this pair on
    S.loc_path_start_idx = 0; S.start_node_idx = 0;
    if (planned_once && valid_last) {
        const TrajM& T = S.tm[set][lsel];
        const double calc_time = t_now - S.last_stamp;
        S.last_stamp = t_now;
        if (cfg.calc_time_buffer_len > CALC_BUF) { fail(S, LTPL_ERR_CAPACITY, E_CALC_BUF); return; }
        if (S.n_calc >= cfg.calc_time_buffer_len) { for (int i = 0; i + 1 < S.n_calc; ++i) S.calc_buffer[i] = S.calc_buffer[i + 1]; --S.n_calc; }
        S.calc_buffer[S.n_calc++] = calc_time;
        double sum = 0.0; for (int i = 0; i < S.n_calc; ++i) sum += S.calc_buffer[i];
        const double avg = sum / (double)S.n_calc;
        const double t_const = fmin(avg * cfg.calc_time_safety, 0.5);
        // index of the pose reached after t_const on the last trajectory (:370-378)
        const Rows bp = B.bp(S.bp_slot[bsel]); const int nb = S.bp_rows[bsel];
        double* cum = B.scr(0);
        x.scan_seq(nb - 2, [&](int i) { const double ds = bp.at(i + 2, 0) - bp.at(i + 1, 0), v = bp.at(i + 1, 5);
                                        return (v != 0.0) ? ds / v : inf(); }, cum);
        x.sync();
        const int ff = x.find_first(nb - 2, [&](int i) { return !(cum[i] <= t_const); });
        const int next_idx = (ff < nb - 2 ? ff : 0) + 1;
        // first node behind that pose (:381-393): project on the polyline of the node coordinates
        const Rows tp = B.pp(set, lsel).from(T.r0); const int* ni = B.nidx(set, lsel) + T.i0; const int* nd = B.nodes(set, lsel) + (size_t)T.n0 * 2;
        double* ncx = B.scr(1); double* ncy = ncx + T.ni;
        for (int i = x.lane(); i < T.ni; i += X::W) { const int r = ni[i]; ncx[i] = tp.at(r, 0); ncy[i] = tp.at(r, 1); }
        x.sync();
        const Poly np_{ncx, ncy, 1, T.ni};
        const Foot f = project_on_polyline(x, np_, bp.at(next_idx, 1), bp.at(next_idx, 2), false, false, nullptr, 0, 0);
        S.start_node_idx = f.i1;
        S.loc_path_start_idx = ni[S.start_node_idx];
        S.start_node[0] = nd[(size_t)S.start_node_idx * 2]; S.start_node[1] = nd[(size_t)S.start_node_idx * 2 + 1];
        S.has_start = 1;
        last_from = S.start_node_idx;
    } else {
        S.last_stamp = t_now; S.has_stamp = 1;
        if (S.const_exists && S.has_start) {
            const TrajM& T = S.tm[set][lsel];
            const int* nd = B.nodes(set, lsel) + (size_t)T.n0 * 2;
            int idx = -1;
            for (int i = 0; i < T.nn; ++i) if (nd[(size_t)i * 2] == S.start_node[0] && nd[(size_t)i * 2 + 1] == S.start_node[1]) { idx = i; break; }
            if (idx >= 0) {
                const int g = lat.layer_off[S.start_node[0]] + S.start_node[1];
                const Rows tp = B.pp(set, lsel).from(T.r0);
                const Poly pl{tp.col(0), tp.col(1), 1, T.rows};
                S.loc_path_start_idx = closest_index(x, pl, lat.node_x[g], lat.node_y[g]);
                S.start_node_idx = idx;
            }
        }
    }
    if (!S.has_start) { fail(S, LTPL_ERR_INVALID_ARG, E_NO_START); return; }
    // constant path segment (:412-414), test in front of seam (1), the packed inputs of seam (1) (:416-427)
    Rows seg{nullptr, 0}; bool has_seg = false; int seg_rows = 0;
    S.const_rows = -1;
    if (S.const_exists) { seg = B.pp(set, lsel).from(S.tm[set][lsel].r0); has_seg = true; seg_rows = S.loc_path_start_idx + 1; S.const_rows = seg_rows; }
    int in_const, besides, cc;
    const int v0 = ob.veh_off[p], nv = ob.veh_off[p + 1] - v0;
    const_segment_test(x, lat, has_seg, seg, seg_rows, S.has_pos ? S.pos_est : nullptr, nv, ob.pos_off + v0, ob.px, ob.py, ob.radius + v0, &in_const, &besides, &cc);
    int fl = LTPL_FLAG_ACTION_SETS;
    if (in_const) fl |= LTPL_FLAG_OBJ_IN_CONST;
    if (besides) fl |= LTPL_FLAG_OBJ_BESIDES;
    if (has_seg) fl |= LTPL_FLAG_HAS_PSI_S;
    int k = 0;
    if (x.lane() == 0) {
        pin.psi_s[p] = has_seg ? seg.at(seg_rows - 1, 2) : 0.0;
        pin.start_layer[p] = S.start_node[0]; pin.start_node[p] = S.start_node[1];
        pin.flags[p] = fl; pin.last_action[p] = sel; pin.const_closest[p] = cc;
        for (int i = 0; i < LTPL_MAX_LAST_NODES; ++i) { pin.last_layer[(size_t)p * LTPL_MAX_LAST_NODES + i] = -1; pin.last_node[(size_t)p * LTPL_MAX_LAST_NODES + i] = -1; }
    }
    if (last_from >= 0) {
        const TrajM& T = S.tm[set][lsel];
        const int* nd = B.nodes(set, lsel) + (size_t)T.n0 * 2;
        for (int i = last_from; i < T.nn && k < LTPL_MAX_LAST_NODES; ++i) {
            const int a = nd[(size_t)i * 2], b = nd[(size_t)i * 2 + 1];
            if (a == FLT_NONE || b == FLT_NONE) break;
            if (x.lane() == 0) { pin.last_layer[(size_t)p * LTPL_MAX_LAST_NODES + k] = a; pin.last_node[(size_t)p * LTPL_MAX_LAST_NODES + k] = b; }
            ++k;
        }
    }
    if (x.lane() == 0) pin.n_last[p] = k;
}

// ---------------------------------------------------------------------------------------------------------------------
// OTH.calc_paths behind seam (1): stitch the new paths behind the constant part (OTH.py:429-513)
// ---------------------------------------------------------------------------------------------------------------------
template <class X>
FLT_FN void paths_post(const X& x, const FLat& lat, const Block& B, PlannerS& S, int p, const FPathsOut& po)
{
    const Dims& D = B.D;
    const int A = LTPL_MAX_ACTIONS, cn = D.cn, cp = D.cp;
    const int set = S.cur_set, nset = 1 - set;
    const int lsel = S.const_exists ? find_last(S, S.sel_action) : -1;
    const int loc = S.loc_path_start_idx, sni = S.start_node_idx;
    TrajM old{}; Rows opp{nullptr, 0}; const double* oco = nullptr; const int* ond = nullptr; const int* oni = nullptr;
    if (lsel >= 0) {
        old = S.tm[set][lsel];
        opp = B.pp(set, lsel).from(old.r0); oco = B.coeff(set, lsel) + (size_t)old.c0 * 8;
        ond = B.nodes(set, lsel) + (size_t)old.n0 * 2; oni = B.nidx(set, lsel) + old.i0;
    }
    S.closest_obj_index = po.closest_obj_index[p];
    int nf = 0;
    for (int a = 0; a < po.n_actions[p]; ++a) {
        const size_t slot = (size_t)p * A + a;
        if (!po.valid[slot]) continue;
        TrajM& T = S.tm[nset][nf];
        T = TrajM{}; T.id = po.action_id[slot]; T.red_len = po.reduced[slot] != 0;
        const int nn = po.n_nodes[slot], npts = po.n_pts[slot];
        const int* nd = po.nodes + slot * cn; const int* ni = po.node_idx + slot * cn;
        const double* co = po.coeff + slot * cn * 8; const double* pp = po.pp + slot * cp * 5;
        const Rows tpp = B.pp(nset, nf); double* tco = B.coeff(nset, nf); int* tnd = B.nodes(nset, nf); int* tni = B.nidx(nset, nf);
        int rows, n_idx, n_nodes = 0, n_co = 0, pre = 0;
        if (lsel >= 0) {
            pre = loc > 0 ? loc : 0;
            rows = pre + npts;
            n_idx = sni < old.ni ? sni : old.ni;
            if (sni > 0) { n_nodes = sni; n_co = sni < old.nc ? sni : old.nc; }
        } else { rows = npts; n_idx = 0; }
        if (rows > D.R || (lsel >= 0 && pre > old.rows)) { fail(S, LTPL_ERR_CAPACITY, E_CAP_ROWS); return; }
        if (n_idx + nn > D.CN || n_nodes + nn > D.CN || n_co + nn > D.CN) { fail(S, LTPL_ERR_CAPACITY, E_CAP_NODES); return; }
        for (int c = 0; c < 5; ++c) for (int i = x.lane(); i < pre; i += X::W) tpp.at(i, c) = opp.at(i, c);
        for (int i = x.lane(); i < npts * 5; i += X::W) tpp.at(pre + i / 5, i % 5) = pp[i];          // (the path kernel's rows are row-major)
        for (int i = x.lane(); i < n_idx; i += X::W) tni[i] = oni[i];
        for (int i = x.lane(); i < nn; i += X::W) tni[n_idx + i] = ni[i] + (lsel >= 0 ? loc : 0);
        for (int i = x.lane(); i < n_nodes * 2; i += X::W) tnd[i] = ond[i];
        for (int i = x.lane(); i < nn; i += X::W) { tnd[(size_t)(n_nodes + i) * 2] = (S.start_node[0] + i) % lat.L; tnd[(size_t)(n_nodes + i) * 2 + 1] = nd[i]; }
        for (int i = x.lane(); i < n_co * 8; i += X::W) tco[i] = oco[i];
        for (int i = x.lane(); i < (nn - 1) * 8; i += X::W) tco[(size_t)n_co * 8 + i] = co[i];
        x.sync();
        if (lsel >= 0 && loc > 0 && old.rows == loc) {                                     // :449-454
            const int j = loc - 1;
            const double dx = tpp.at(j + 1, 0) - tpp.at(j, 0), dy = tpp.at(j + 1, 1) - tpp.at(j, 1);
            const double el = sqrt(dx * dx + dy * dy);
            x.sync();
            if (x.lane() == 0) tpp.at(j, 4) = el;
        }
        T.r0 = 0; T.rows = rows; T.c0 = 0; T.nc = n_co + (nn - 1 > 0 ? nn - 1 : 0); T.n0 = 0; T.nn = n_nodes + nn; T.i0 = 0; T.ni = n_idx + nn;
        ++nf;
    }
    if (nf == 0 && lsel >= 0 && S.const_rows > 2) {
        // blocked track: keep the constant segment including its end node (:474-506)
        const int loc1 = loc + 1, sni1 = sni + 1;
        TrajM T{}; T.id = S.sel_action; T.red_len = 1;
        T.rows = loc1 < old.rows ? loc1 : old.rows; T.ni = sni1 < old.ni ? sni1 : old.ni; T.nn = sni1 < old.nn ? sni1 : old.nn; T.nc = sni1 < old.nc ? sni1 : old.nc;
        const Rows tpp = B.pp(nset, 0); double* tco = B.coeff(nset, 0); int* tnd = B.nodes(nset, 0); int* tni = B.nidx(nset, 0);
        for (int c = 0; c < 5; ++c) for (int i = x.lane(); i < T.rows; i += X::W) tpp.at(i, c) = opp.at(i, c);
        for (int i = x.lane(); i < T.ni; i += X::W) tni[i] = oni[i];
        for (int i = x.lane(); i < T.nn * 2; i += X::W) tnd[i] = ond[i];
        for (int i = x.lane(); i < T.nc * 8; i += X::W) tco[i] = oco[i];
        S.tm[nset][0] = T;
        nf = 1;
    }
    x.sync();
    S.n_last = nf; for (int i = 0; i < nf; ++i) S.last_slot[i] = i;
    S.cur_set = nset; S.has_last = 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// OTH.get_ref_idx (OTH.py:518-601)
// ---------------------------------------------------------------------------------------------------------------------
template <class X>
FLT_FN void ref_idx(const X& x, const FCfg& cfg, const Block& B, PlannerS& S, double px, double py)
{
    S.pos_est[0] = px; S.pos_est[1] = py; S.has_pos = 1;
    const int b = S.has_bp ? find_bp(S, S.raw_action) : -1;
    const bool valid_last = b >= 0 && S.bp_rows[b] > 0;
    const bool valid_this = S.n_last > 0;
    int cut_index_layer = 0;
    S.n_vel_course = 0;
    if (valid_last) {
        const Rows bp = B.bp(S.bp_slot[b]); const int n = S.bp_rows[b];
        const Poly pl{bp.col(1), bp.col(2), 1, n};
        const Foot f = project_on_polyline(x, pl, px, py, false, false, nullptr, 0, 0);
        const int cut = f.i0;
        const int m = n - cut - 1;                                       // len(v_past) (:565-567)
        double* cum = B.scr(0);
        x.scan_seq(m, [&](int i) { const double ds = bp.at(cut + i + 1, 0) - bp.at(cut + i, 0), v = bp.at(cut + i, 5);
                                   return (v != 0.0) ? ds / v : inf(); }, cum);
        x.sync();
        const int ff = x.find_first(m, [&](int i) { return !(cum[i] <= cfg.delaycomp); });
        int vel_idx = (ff < m ? ff : 0) + 1;
        if (vel_idx > m - 1) vel_idx = m - 1;
        if (vel_idx < 0) vel_idx = 0;
        S.vel_plan = bp.at(cut + vel_idx, 5); S.acc_plan = bp.at(cut + vel_idx, 6);
        double* vc = B.velc();
        for (int i = x.lane(); i < vel_idx; i += X::W) vc[i] = bp.at(cut + i, 5);
        S.n_vel_course = vel_idx;
        S.cut_index_pos = S.last_cut_idx + cut;
        if (valid_this) {
            const TrajM& T = S.tm[S.cur_set][S.last_slot[0]];                // first key of the dict (:581)
            const int* ni = B.nidx(S.cur_set, S.last_slot[0]) + T.i0;
            int ff2 = 0;
            for (int i = 0; i < T.ni; ++i) if (!(ni[i] < S.cut_index_pos)) { ff2 = i; break; }
            S.cut_layer = ff2 - 2 > 0 ? ff2 - 2 : 0;
            cut_index_layer = T.ni == 0 ? 0 : ni[S.cut_layer];
        } else { S.cut_layer = 0; cut_index_layer = 0; }
        x.sync();
    } else {
        S.cut_index_pos = 0; S.cut_layer = 0; cut_index_layer = 0;
        S.vel_plan = S.v_start; S.acc_plan = 0.0;
    }
    S.last_cut_idx = S.cut_index_pos - cut_index_layer;
}

// tph.conv_filt(signal, filt_window, closed=False): centred moving average of odd width, the first / last half window keep their values
FLT_FN double conv_filt_at(const double* vx, int n, int width, int i)
{
    const int half = (width - 1) / 2;
    if (half < 1 || n < width || i < half || i >= n - half) return vx[i];
    double acc = 0.0;
    for (int k = i - half; k <= i + half; ++k) acc += vx[k] * (1.0 / (double)width);
    return acc;
}

// :925-941: vx filtered, ax from neighbours over np.diff(s), -5 at standstill
template <class X>
FLT_FN void finalize_bp(const X& x, const FCfg& cfg, const double* s_arr, const Rows& pv, const double* vx_raw, int n, const Rows& bp)
{
    for (int i = x.lane(); i < n; i += X::W) {
        const double v = conv_filt_at(vx_raw, n, cfg.filt_window_width, i);
        bp.at(i, 0) = s_arr[i]; bp.at(i, 1) = pv.at(i, 0); bp.at(i, 2) = pv.at(i, 1); bp.at(i, 3) = pv.at(i, 2); bp.at(i, 4) = pv.at(i, 3);
        bp.at(i, 5) = v;
        double ax = 0.0;
        if (i + 1 < n) {
            const double v1 = conv_filt_at(vx_raw, n, cfg.filt_window_width, i + 1);
            ax = (v1 * v1 - v * v) / (2 * (s_arr[i + 1] - s_arr[i]));
            if (fabs(v) <= 1e-8 && fabs(ax) <= 1e-8) ax = -5.0;
        }
        bp.at(i, 6) = ax;
    }
}

// job slot `slot` of planner p (slot 0 is reserved for the follow job of a tick: the device runs the follow jobs as their own launch)
template <class X>
FLT_FN int make_job(const X& x, const Dims& D, const FJobs& J, int p, int slot, int mode, const Rows& pv, const GgSrc& gsrc, int i0, int i1,
                    int n_el, double v_start, bool has_end, double v_end, const JobCar& car = JobCar{0.0, 0, 0})
{
    const int j = p * J.per_planner + slot;
    VelJob jb{};
    jb.v_max = car.v_max; jb.axm_off = car.axm_off; jb.n_axm = car.n_axm; jb.gg_rows = gsrc.rows() ? 1 : 0;
    jb.mode = mode; jb.n = i1 - i0; jb.n_el = n_el; jb.has_v_end = has_end ? 1 : 0; jb.v_start = v_start; jb.v_end = v_end;
    jb.off_kappa = j * 4 * D.R; jb.off_el = jb.off_kappa + D.R; jb.off_gg = jb.off_kappa + 2 * D.R; jb.off_out = j * D.R;
    double* kap = J.pool + jb.off_kappa; double* el = J.pool + jb.off_el; double* gg = J.pool + jb.off_gg;
    jb.lane_form = (J.ke && !gsrc.rows() && ((mode == LTPL_VEL_FB && slot >= 1) || (J.ke_follow >= 0 && mode == LTPL_VEL_FOLLOW_CONTROLLED && slot == 0))) ? 1 : 0;
    const bool lane_fb = J.ke && mode == LTPL_VEL_FB && slot >= 1 && !gsrc.rows();
    const bool lane_follow = J.ke && J.ke_follow >= 0 && mode == LTPL_VEL_FOLLOW_CONTROLLED && slot == 0 && !gsrc.rows();
    if (lane_fb || lane_follow) {
        F2* ke = J.ke + kep_base_f(lane_fb ? p * (J.per_planner - 1) + slot - 1 : J.ke_follow + p, J.ke_rows);
        for (int i = x.lane(); i < i1 - i0; i += X::W) {
            F2 r; r.x = (ke_scalar_f)fabs(pv.at(i0 + i, 3)); r.y = i < n_el ? (ke_scalar_f)pv.at(i0 + i, 4) : (ke_scalar_f)0;
            ke[kep_row_f(i)] = r;
        }
        if (x.lane() == 0) { gg[0] = gsrc.at(0, 0); gg[1] = gsrc.at(0, 1); }
    } else {
        for (int i = x.lane(); i < i1 - i0; i += X::W) { kap[i] = pv.at(i0 + i, 3); gg[(size_t)i * 2] = gsrc.at(i0 + i, 0); gg[(size_t)i * 2 + 1] = gsrc.at(i0 + i, 1); }
        for (int i = x.lane(); i < n_el; i += X::W) el[i] = pv.at(i0 + i, 4);
        if (n_el < 1 && x.lane() == 0) el[0] = 0.0;
    }
    if (x.lane() == 0) J.jobs[j] = jb;
    return slot;
}

// ---------------------------------------------------------------------------------------------------------------------
// OTH.calc_vel_profile, stage A: get_ref_idx, slicing (:700-731), job construction (:736-903)
// ---------------------------------------------------------------------------------------------------------------------
template <class X>
FLT_FN void vel_a(const X& x, const FLat& lat, const FCfg& cfg, const Block& B, PlannerS& S, int p, const FObj& ob, const FVelIn& vin, const FJobs& J)
{
    const Dims& D = B.D;
    for (int j = x.lane(); j < J.per_planner; j += X::W) J.jobs[p * J.per_planner + j].n = 0;
    x.sync();
    if (S.err) return;
    if (!S.ref_done) ref_idx(x, cfg, B, S, vin.pos_x[p], vin.pos_y[p]);
    S.ref_done = 0;
    const double vel_max = vin.vel_max[p], gg_scale = vin.gg_scale[p];
    S.traj_base_id += 10;
    if (!S.has_old_gg) { S.old_gg_scale = gg_scale; S.has_old_gg = 1; }
    S.n_bp = 0; S.has_bp = 1; S.n_ids = 0;
    const int vel_idx = S.n_vel_course;
    const int set = S.cur_set;
    int n_jobs = 1;                                        // next free slot (0: the follow job)
    S.n_work = S.n_last;
    // local gg of the stitched paths (:641-666): the caller's rows for a key ("each path coordinate must be represented by a row"), or the
    // constant tuple. Rows are kept next to the path rows (same window: they are trimmed with them) and, from the cut on, in Block::gv(k)
    // for stage C (the emergency profile uses the FIRST kept key's rows even when that key falls back to the backup plan). Its own loop
    // in front of the job construction: calls without rows (the common case) pay one uniform branch.
    if (vin.gg_off && vin.gg_rows && B.g) {
        const int MK = LTPL_PLANNER_MAX_KEYS;
        for (int k = 0; k < S.n_last; ++k) {
            const int sl = S.last_slot[k];
            TrajM& T = S.tm[set][sl];
            const int rows = T.rows;
            const int c0 = S.cut_index_pos < 0 ? 0 : (S.cut_index_pos < rows ? S.cut_index_pos : rows);
            int g_n = 0; const double* g_src = vin.gg_rows;
            if (k < MK) { const int o0 = vin.gg_off[p * MK + k]; g_n = vin.gg_off[p * MK + k + 1] - o0; g_src += (size_t)o0 * 2; }
            if (g_n > 0 && g_n != rows) { fail(S, LTPL_ERR_INVALID_ARG, E_GG_ROWS); return; }
            if (g_n > 0) {
                const Rows G = B.gg(set, sl).from(T.r0), gv = B.gv(k);
                for (int i = x.lane(); i < rows; i += X::W) {
                    const double ax = g_src[(size_t)i * 2], ay = g_src[(size_t)i * 2 + 1];
                    G.at(i, 0) = ax; G.at(i, 1) = ay;
                    if (i >= c0) { gv.at(i - c0, 0) = ax; gv.at(i - c0, 1) = ay; }
                }
            }
            T.has_gg = g_n > 0 ? 1 : 0;
        }
        x.sync();
    } else {
        for (int k = 0; k < S.n_last; ++k) S.tm[set][S.last_slot[k]].has_gg = 0;
    }
    for (int k = 0; k < S.n_last; ++k) {
        const int sl = S.last_slot[k];
        TrajM& T = S.tm[set][sl];
        Work& W = S.w[k];                                  // (in place: the scalars live in LDS on the device, a local copy costs ~20 registers)
        W = Work{}; W.vel_idx = vel_idx; W.job_follow = W.job_free = W.job_fb = W.job_backup = -1; W.vel_bound = 1;
        if (S.n_ids < BPS) { S.id_key[S.n_ids] = T.id; S.id_val[S.n_ids] = S.traj_base_id + (T.id >= 0 && T.id <= 3 ? T.id : 9); ++S.n_ids; }
        const int rows = T.rows;
        const int c0 = S.cut_index_pos < 0 ? 0 : (S.cut_index_pos < rows ? S.cut_index_pos : rows);
        const int m = rows - c0;
        if (S.cut_layer >= T.ni) { fail(S, LTPL_ERR_INVALID_ARG, E_CUT_LAYER); return; }
        const int cil = (B.nidx(set, sl) + T.i0)[S.cut_layer];
        W.cut_index_layer = cil;
        const Rows pv = B.pp(set, sl).from(T.r0 + c0);                           // action_set_path_param_vel: rows from cut_index_pos on
        W.c0 = T.r0 + c0;                                                        // (absolute row in the slot: the window below moves)
        W.gv_n = T.has_gg ? m : 0;
        {   // trim the memory for the next iteration, aligned with the nodes (:714-731): the windows move, node indices are re-based
            int* ni = B.nidx(set, sl) + T.i0 + S.cut_layer;
            const int cnt = T.ni - S.cut_layer;
            x.sync();
            for (int i = x.lane(); i < cnt; i += X::W) ni[i] -= cil;
            T.i0 += S.cut_layer; T.ni = cnt;
            const int c1 = cil < 0 ? 0 : (cil < rows ? cil : rows);
            T.r0 += c1; T.rows = rows - c1;
            T.gax = vin.gg_ax[p]; T.gay = vin.gg_ay[p];
            const int cc = S.cut_layer < T.nc ? S.cut_layer : T.nc;
            T.c0 += cc; T.nc -= cc;
            const int cnn = S.cut_layer < T.nn ? S.cut_layer : T.nn;
            T.n0 += cnn; T.nn -= cnn;
            x.sync();
        }
        W.n = m;
        if (m == 0) { W.empty = 1; continue; }
        double* s_arr = B.sarr(k);
        if (x.lane() == 0) s_arr[0] = 0.0;
        x.scan_seq(m - 1, [&](int i) { return pv.at(i, 4); }, s_arr + 1);                            // :743
        x.sync();
        if (S.vel_plan > vel_max + 0.1) { fail(S, LTPL_ERR_UNSUPPORTED, E_BRAKE_PREFIX); return; }       // (the reference raises, OTH.py:919)
        S.old_gg_scale = gg_scale;
        W.pref_idx = vel_idx; W.vel_start = S.vel_plan;
        const int pref = W.pref_idx;
        // friction limits of the rows of `pv`, times gg_scale (:944); built at the job (the closest-point searches of the follow branch in
        // between are the register peak of the stage: nothing of this is kept alive across them)
        auto gsrc_of = [&]() { return T.has_gg ? gg_table(B.gg(set, sl).from(W.c0), vin.gg_scale[p]) : gg_const(T.gax, T.gay, vin.gg_scale[p]); };
        if (n_jobs + (T.id == LTPL_ACT_FOLLOW ? 1 : 0) + ((T.id != LTPL_ACT_FOLLOW || T.red_len) ? 1 : 0) > J.per_planner) { fail(S, LTPL_ERR_CAPACITY, E_CAP_JOBS); return; }
        if (m > D.RV) { fail(S, LTPL_ERR_CAPACITY, E_CAP_VEL); return; }
        if (T.id == LTPL_ACT_FOLLOW) {                                                              // :763-830
            if (m - pref < 1) { fail(S, LTPL_ERR_INVALID_ARG, E_FOLLOW_EMPTY); return; }
            double obj_dist = 0.0, v_obj = 0.0, ox = vin.pos_x[p], oy = vin.pos_y[p];
            const int v0 = ob.veh_off[p], nv = ob.veh_off[p + 1] - v0;
            if (S.closest_obj_index >= 0 && S.closest_obj_index < nv) {
                const int vi = v0 + S.closest_obj_index;
                ox = ob.px[ob.pos_off[vi]]; oy = ob.py[ob.pos_off[vi]]; v_obj = ob.vel ? ob.vel[vi] : 0.0;
                double* cs = B.scr(0);                                                               // cumsum(path[:, 4]) (:777,782)
                x.scan_seq(m, [&](int i) { return pv.at(i, 4); }, cs);
                x.sync();
                const Poly pl{pv.col(0), pv.col(1), 1, m};
                const double s_obj = project_on_polyline(x, pl, ox, oy, false, true, cs, 1, m).s;
                const double s_sta = project_on_polyline(x, pl, S.pos_est[0], S.pos_est[1], false, true, cs, 1, m).s;
                obj_dist = s_obj - s_sta;
            }
            if (J.jobs[p * J.per_planner].n > 0) { fail(S, LTPL_ERR_CAPACITY, E_CAP_JOBS); return; }       // (two follow keys in one tick: not a thing)
            const int j = make_job(x, D, J, p, 0, LTPL_VEL_FOLLOW_CONTROLLED, pv, gsrc_of(), pref, m, m - pref, W.vel_start, false, 0.0, car_of(vin, p));
            x.sync();
            if (x.lane() == 0) {
                VelJob& jb = J.jobs[p * J.per_planner + j];
                jb.v_ego = vin.vel_est[p]; jb.v_obj = v_obj; jb.safety_d = vin.safety_d[p]; jb.obj_dist = obj_dist; jb.obj_x = ox; jb.obj_y = oy;
            }
            W.job_follow = j;
            W.job_free = make_job(x, D, J, p, n_jobs++, LTPL_VEL_FB, pv, gsrc_of(), pref, m, m - pref - 1, W.vel_start, false, 0.0, car_of(vin, p));
        }
        if (T.id != LTPL_ACT_FOLLOW || T.red_len) {                                                 // :834-903
            W.generic = 1;
            if (T.nn < 1) { fail(S, LTPL_ERR_INVALID_ARG, E_NO_NODES); return; }
            const int* nd = B.nodes(set, sl) + (size_t)T.n0 * 2;
            const int el_ = nd[(size_t)(T.nn - 1) * 2], en = nd[(size_t)(T.nn - 1) * 2 + 1];
            if (el_ < 0 || el_ >= lat.L) { fail(S, LTPL_ERR_INVALID_ARG, E_END_NONE); return; }
            int dn = en - lat.rl_idx[el_]; if (dn < 0) dn = -dn;
            const double raceline_offset = dn * lat.lat_offset;
            double v_end; int v_idx;
            if (T.red_len) {
                v_end = 0.0;
                // spl = sum of the first m - 1 element lengths = s_arr[m - 1]; first i with not (cumsum[i] < spl - 5), cumsum[i] = s_arr[i + 1]
                const double spl = s_arr[m - 1];
                const int ff = x.find_first(m - 1, [&](int i) { return !(s_arr[i + 1] < (spl - 5.0)); });
                v_idx = (ff < m - 1 ? ff : 0) + 1;
                if (v_idx == 1 && m > 1) v_idx = m;
            } else {
                v_end = lat.vel_rl[el_];
                v_end -= fmin(v_end * lat.vel_decrease_lat * raceline_offset, v_end);
                v_idx = m;
            }
            W.v_idx = v_idx;
            if (v_idx - pref > 1) { W.job_fb = make_job(x, D, J, p, n_jobs++, LTPL_VEL_FB, pv, gsrc_of(), pref, v_idx, v_idx - pref - 1, W.vel_start, true, v_end, car_of(vin, p)); W.has_fb = 1; }
        }
    }
    x.sync();
}

// ---------------------------------------------------------------------------------------------------------------------
// stage B: assemble trajectories (:824-941), decide keep / drop / backup (:943-1015); JB: one backup job slot per planner
// ---------------------------------------------------------------------------------------------------------------------
template <class X>
FLT_FN void vel_b(const X& x, const FCfg& cfg, const Block& B, PlannerS& S, int p, const FJobs& JA, const FJobs& JB)
{
    const Dims& D = B.D;
    if (x.lane() == 0) JB.jobs[p * JB.per_planner].n = 0;
    x.sync();
    if (S.err) return;
    const int set = S.cur_set;
    const double* vc = B.velc();
    int n_backup = 0;
    for (int k = 0; k < S.n_work; ++k) {
        Work& W = S.w[k];
        const int sl = S.last_slot[k];
        TrajM& T = S.tm[set][sl];
        const int m = W.n, vel_idx = W.vel_idx;
        W.vel_bound = 1;
        if (!W.empty) {
            const Rows pv = B.pp(set, sl).from(W.c0);
            double* vxf = B.vx(0); double* vxg = B.vx(1);
            const double* vx = nullptr;
            bool have_bp = false;
            if (W.job_follow >= 0) {
                const int jf = p * JA.per_planner + W.job_follow;
                W.too_close = JA.flags[2 * jf]; W.vel_bound = JA.flags[2 * jf + 1];
                const double* f = JA.out + (size_t)jf * D.R; const int nf = m - W.pref_idx;
                const bool has_u = W.job_free >= 0;                                                       // np.minimum(vx_profile, vx_compl) (:310)
                int len = vel_idx + nf;
                if (len > m) len = m;
                if (len != m) { fail(S, LTPL_ERR_INVALID_ARG, E_FOLLOW_SHORT); return; }
                for (int i = x.lane(); i < m; i += X::W) {
                    double v;
                    if (i < vel_idx) v = vc[i];
                    else { v = f[i - vel_idx]; if (has_u) { const double w = job_out(JA, D, p, W.job_free, i - vel_idx); v = v < w ? v : w; } }
                    vxf[i] = v;
                }
                have_bp = true;
            }
            if (W.generic) {
                const bool has_g = W.has_fb != 0;
                int ng = W.has_fb ? W.v_idx - W.pref_idx : 1;
                const int n_prof = ng;
                if (W.v_idx != m || W.v_idx <= 2) ng += (m - W.v_idx > 0 ? m - W.v_idx : 0);                          // :901-903
                const double g0 = has_g ? job_out(JA, D, p, W.job_fb, 0) : 0.0;
                W.vel_bound = fabs(g0 - S.vel_plan) < cfg.v_max_offset ? 1 : 0;                                      // :906-911
                int len = vel_idx + ng;
                if (len > m) len = m;
                if (len != m) { fail(S, LTPL_ERR_INVALID_ARG, E_VX_SHORT); return; }
                for (int i = x.lane(); i < m; i += X::W) {
                    double v;
                    if (i < vel_idx) v = vc[i];
                    else { const int q = i - vel_idx; v = (q < n_prof && has_g) ? job_out(JA, D, p, W.job_fb, q) : 0.0; }
                    vxg[i] = v;
                }
                x.sync();
                vx = vxg;
                if (have_bp) {
                    if (m < 6) { fail(S, LTPL_ERR_INVALID_ARG, E_ROW5); return; }                                    // (IndexError at OTH.py:923)
                    if (vxf[5] < vxg[5]) vx = vxf;
                }
            } else { x.sync(); vx = vxf; }
            finalize_bp(x, cfg, B.sarr(k), pv, vx, m, B.bp(k));
            x.sync();
        }
        const bool sf = T.id == LTPL_ACT_FOLLOW || T.id == LTPL_ACT_STRAIGHT;
        if (W.vel_bound || sf) {
            if (W.vel_bound || !S.has_backup) W.keep = 1;
            else {
                // recursive infeasibility: brake on the previous solution (:950-1006)
                const TrajM Bm = S.tm[S.backup_set][S.backup_slot];
                const int bs = S.backup_set, bk = S.backup_slot;
                const int cl = S.cut_layer, cil = W.cut_index_layer;
                if (cl > Bm.ni) { fail(S, LTPL_ERR_INVALID_ARG, E_BACKUP_CUT); return; }
                if (n_backup >= JB.per_planner) { fail(S, LTPL_ERR_CAPACITY, E_BACKUP_LEN); return; }
                const int br = Bm.rows;
                const int c1 = cil < 0 ? 0 : (cil < br ? cil : br);
                const int i0 = S.cut_index_pos + vel_idx;
                if (i0 >= br) { fail(S, LTPL_ERR_INVALID_ARG, E_BACKUP_SHORT); return; }
                const Rows bpp = B.pp(bs, bk).from(Bm.r0);
                if (br - i0 > D.RV) { fail(S, LTPL_ERR_CAPACITY, E_CAP_VEL); return; }
                // brake job on the backup rows [i0, br): no gg_scale (:229-255)
                const Rows bgg = B.gg(bs, bk).from(Bm.r0);
                W.job_backup = make_job(x, D, JB, p, n_backup++, LTPL_VEL_BRAKE, bpp, Bm.has_gg ? gg_table(bgg) : gg_const(Bm.gax, Bm.gay), i0, br, br - i0 - 1,
                                        S.vel_plan, false, 0.0);
                // the key's memory becomes the (trimmed) backup
                const Rows tpp = B.pp(set, sl); double* tco = B.coeff(set, sl); int* tnd = B.nodes(set, sl); int* tni = B.nidx(set, sl);
                const double* bco = B.coeff(bs, bk) + (size_t)Bm.c0 * 8; const int* bnd = B.nodes(bs, bk) + (size_t)Bm.n0 * 2; const int* bni = B.nidx(bs, bk) + Bm.i0;
                TrajM N = T;
                N.r0 = 0; N.rows = br - c1; N.i0 = 0; N.ni = Bm.ni - cl;
                const int cc = cl < Bm.nc ? cl : Bm.nc, cnn = cl < Bm.nn ? cl : Bm.nn;
                N.c0 = 0; N.nc = Bm.nc - cc; N.n0 = 0; N.nn = Bm.nn - cnn;
                N.gax = Bm.gax; N.gay = Bm.gay; N.has_gg = Bm.has_gg;
                x.sync();
                for (int c = 0; c < 5; ++c) for (int i = x.lane(); i < N.rows; i += X::W) tpp.at(i, c) = bpp.at(c1 + i, c);
                if (Bm.has_gg) { const Rows tgg = B.gg(set, sl); for (int c = 0; c < 2; ++c) for (int i = x.lane(); i < N.rows; i += X::W) tgg.at(i, c) = bgg.at(c1 + i, c); }
                for (int i = x.lane(); i < N.ni; i += X::W) tni[i] = bni[cl + i] - cil;
                for (int i = x.lane(); i < N.nc * 8; i += X::W) tco[i] = bco[(size_t)cc * 8 + i];
                for (int i = x.lane(); i < N.nn * 2; i += X::W) tnd[i] = bnd[(size_t)cnn * 2 + i];
                x.sync();
                T = N;
                W.keep = 1;
            }
        } else W.drop = 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// stage C: backup trajectories, commit, emergency job (:986-1034, calc_brake_emergency.py:9-45); JC: one job slot per planner
// ---------------------------------------------------------------------------------------------------------------------
template <class X>
FLT_FN void vel_c(const X& x, const FCfg& cfg, const Block& B, PlannerS& S, int p, const FVelIn& vin, const FJobs& JB, const FJobs& JC)
{
    const Dims& D = B.D;
    if (x.lane() == 0) JC.jobs[p * JC.per_planner].n = 0;
    x.sync();
    if (S.err) return;
    const double* vc = B.velc();
    int nk = 0; int kept[KEYS];
    for (int k = 0; k < S.n_work; ++k) {
        Work& W = S.w[k];
        const int sl = S.last_slot[k];
        const TrajM& T = S.tm[S.cur_set][sl];
        if (W.drop) continue;                                                                      // :1007-1025
        int rows = W.n;
        if (W.job_backup >= 0) {
            const TrajM& Bm = S.tm[S.backup_set][S.backup_slot];
            const Rows bpp = B.pp(S.backup_set, S.backup_slot).from(Bm.r0);
            const int c0 = S.cut_index_pos, br = Bm.rows, m = br - c0;
            const double* o = JB.out + (size_t)(p * JB.per_planner + W.job_backup) * D.R;
            const int no = br - (S.cut_index_pos + W.vel_idx);
            if (S.n_vel_course + no != m || m > D.R) { fail(S, LTPL_ERR_INVALID_ARG, E_BACKUP_LEN); return; }
            double* vraw = B.vx(0); double* s_arr = B.scr(0);
            for (int i = x.lane(); i < m; i += X::W) vraw[i] = i < S.n_vel_course ? vc[i] : o[i - S.n_vel_course];
            if (x.lane() == 0) s_arr[0] = 0.0;
            x.scan_seq(m - 1, [&](int i) { return bpp.at(c0 + i, 4); }, s_arr + 1);
            x.sync();
            const Rows bp = B.bp(k), q = bpp.from(c0);
            for (int i = x.lane(); i < m; i += X::W) {                                                 // :996-1004: ax over the element lengths themselves
                const double v = conv_filt_at(vraw, m, cfg.filt_window_width, i);
                bp.at(i, 0) = s_arr[i]; bp.at(i, 1) = q.at(i, 0); bp.at(i, 2) = q.at(i, 1); bp.at(i, 3) = q.at(i, 2); bp.at(i, 4) = q.at(i, 3); bp.at(i, 5) = v;
                double ax = 0.0;
                if (i + 1 < m) {
                    const double v1 = conv_filt_at(vraw, m, cfg.filt_window_width, i + 1);
                    ax = (v1 * v1 - v * v) / (2 * q.at(i, 4));
                    if (fabs(v) <= 1e-8 && fabs(ax) <= 1e-8) ax = -5.0;
                }
                bp.at(i, 6) = ax;
            }
            x.sync();
            rows = m;
        }
        const int q = S.n_bp++;
        S.bp_slot[q] = k; S.bp_id[q] = T.id; S.bp_rows[q] = rows;
        S.bp_traj_id[q] = S.traj_base_id + (T.id >= 0 && T.id <= 3 ? T.id : 9);                         // ACTION_ID_MAP (:14-17,696-697)
        kept[nk++] = sl;
    }
    S.n_last = nk; for (int i = 0; i < nk; ++i) S.last_slot[i] = kept[i];
    if (vin.incl_emerg && vin.incl_emerg[p]) {                                                        // :1028-1034
        if (S.n_bp == 0) { fail(S, LTPL_ERR_INVALID_ARG, E_EMERG_EMPTY); return; }
        S.em_base_id = S.bp_id[0];
        const Rows base = B.bp(S.bp_slot[0]); const int m = S.bp_rows[0];
        if (m > D.RV) { fail(S, LTPL_ERR_CAPACITY, E_CAP_VEL); return; }
        const int j = p * JC.per_planner;
        VelJob jb{};
        jb.mode = LTPL_VEL_BRAKE; jb.n = m; jb.n_el = m - 1; jb.v_start = m > 0 ? base.at(0, 5) : 0.0;
        jb.off_kappa = j * 4 * D.R; jb.off_el = jb.off_kappa + D.R; jb.off_gg = jb.off_kappa + 2 * D.R; jb.off_out = j * D.R;
        double* kap = JC.pool + jb.off_kappa; double* el = JC.pool + jb.off_el; double* gg = JC.pool + jb.off_gg;
        // local gg of the emergency profile: the first kept key's rows from the cut on (W.gv of stage A), the constant tuple behind them
        const double gax = vin.gg_ax[p], gay = vin.gg_ay[p];
        const int k0 = S.bp_slot[0]; const int gvn = S.w[k0].gv_n; const Rows gv = B.gv(k0);
        // With rows, the reference hands the rows of the first key's CURRENT path to the brake solver next to the kappa of the first
        // TRAJECTORY (OTH.py:1029-1036): when that trajectory is the backup plan (other length) tph raises "Length of loc_gg and kappa
        // must be equal!" (reproduced with the unmodified reference, oracle/gen_golden.py 'ggmapdrop' notes). Reported, not papered over.
        if (gvn > 0 && gvn != m) { fail(S, LTPL_ERR_INVALID_ARG, E_EMERG_GG); return; }
        jb.gg_rows = gvn > 0 ? 1 : 0;
        for (int i = x.lane(); i < m; i += X::W) { kap[i] = base.at(i, 4); gg[(size_t)i * 2] = gvn > 0 ? gv.at(i, 0) : gax; gg[(size_t)i * 2 + 1] = gvn > 0 ? gv.at(i, 1) : gay; }
        for (int i = x.lane(); i + 1 < m; i += X::W) el[i] = base.at(i + 1, 0) - base.at(i, 0);
        if (m < 2 && x.lane() == 0) el[0] = 0.0;
        if (x.lane() == 0) JC.jobs[j] = jb;
    }
    x.sync();
}

// stage D: the emergency trajectory from the brake profile (calc_brake_emergency.py:38-45)
template <class X>
FLT_FN void vel_d(const X& x, const Block& B, PlannerS& S, int p, const FVelIn& vin, const FJobs& JC)
{
    const Dims& D = B.D;
    if (S.err || !(vin.incl_emerg && vin.incl_emerg[p])) return;
    const Rows base = B.bp(S.bp_slot[0]), em = B.bp(BPS - 1); const int m = S.bp_rows[0];
    const double* v = JC.out + (size_t)(p * JC.per_planner) * D.R;
    for (int i = x.lane(); i < m; i += X::W) {
        for (int c = 0; c < 5; ++c) em.at(i, c) = base.at(i, c);
        em.at(i, 5) = v[i];
        em.at(i, 6) = (i + 1 < m) ? (v[i + 1] * v[i + 1] - v[i] * v[i]) / (2 * (base.at(i + 1, 0) - base.at(i, 0))) : 0.0;
    }
    const int q = S.n_bp++;
    S.bp_slot[q] = BPS - 1; S.bp_id[q] = LTPL_ACT_EMERGENCY; S.bp_rows[q] = m; S.bp_traj_id[q] = S.bp_traj_id[0];
    if (S.n_ids < BPS) { S.id_key[S.n_ids] = LTPL_ACT_EMERGENCY; S.id_val[S.n_ids] = S.bp_traj_id[0]; ++S.n_ids; }
    x.sync();
}

}  // namespace fleet
