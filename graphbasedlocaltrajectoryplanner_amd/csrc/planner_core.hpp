// planner_core.hpp -- the iterative memory of the reference's OnlineTrajectoryHandler as a C++ state machine, batched over
// independent scenarios ("planners"): SURVEY.md section 8a rows H1 (calc_paths), H2 (get_ref_idx), V0 (calc_vel_profile) and
// section 8f rank 2. One C call per tick stage instead of Python dict / list churn:
//
//   set_start          OnlineTrajectoryHandler.set_initial_pose   graph_ltpl/online_graph/src/OnlineTrajectoryHandler.py:181-270
//   calc_paths         OnlineTrajectoryHandler.calc_paths         OTH.py:289-516  (+ the constant-segment test of
//                      main_online_path_gen.py:76-122 in front of seam (1))
//   calc_vel_profile   OnlineTrajectoryHandler.get_ref_idx        OTH.py:518-601
//                      + OnlineTrajectoryHandler.calc_vel_profile OTH.py:603-1040  (+ calc_brake_emergency.py:9-45)
//
// Pure host C++ (no HIP types): every piece of arithmetic that the reference delegates to seam (1) / seam (2) goes through the
// `Compute` interface. libltpl_hip.so binds it to the HIP kernels (ltpl_hip.hip); the product has no CPU implementation of it.
// What runs here is the reference's control flow: index bookkeeping, slicing / stitching of trajectories, O(n) projections
// of single points on polylines (get_s_coord.py:8-121) and the final column assembly.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
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
// state
// ---------------------------------------------------------------------------------------------------------------------
#define LTPLP_NONE (-1)

struct Traj {                       // one entry of the reference's __last_action_set_* dicts (lists of length 1)
    int id = LTPL_ACT_NONE;
    std::vector<double> pp;         // rows [x, y, psi, kappa, el_length]
    std::vector<double> gg;         // rows [ax, ay]           (__last_action_set_path_gg)
    std::vector<double> coeff;      // rows of 8
    std::vector<int> nodes;         // pairs [layer, node]; LTPLP_NONE = the reference's None
    std::vector<int> node_idx;
    bool red_len = false;
    int rows() const { return (int)(pp.size() / 5); }
    int n_nodes() const { return (int)(nodes.size() / 2); }
};

struct BpTraj { int id = LTPL_ACT_NONE; std::vector<double> bp; int traj_id = 0; int rows() const { return (int)(bp.size() / 7); } };

struct ObjVeh { double x, y, radius, vel; std::vector<double> pos; };   // pos: pairs, own position first, then the prediction

struct Scn {
    int traj_base_id = 0;
    bool has_start = false; int start_node[2] = {0, 0};
    bool has_last = false;  std::vector<Traj> last;          // insertion order of the reference's dict keys
    bool has_bp = false;    std::vector<BpTraj> last_bp;
    bool has_stamp = false; double last_stamp = 0.0;
    int last_cut_idx = 0;
    bool has_pos = false; double pos_est[2] = {0, 0};
    int em_base_id = LTPL_ACT_NONE;
    bool has_backup = false; Traj backup;
    std::vector<double> calc_buffer;
    double v_start = 0.0;
    int action_forced = LTPL_ACT_NONE;
    int closest_obj_index = -1;
    std::vector<ObjVeh> veh;
    bool has_old_gg = false; double old_gg_scale = 1.0;      // VpForwardBackward.py:50-84
    // transient between the two halves of calc_paths
    int sel_action = LTPL_ACT_NONE, raw_action = LTPL_ACT_NONE;
    bool const_exists = false; int const_rows = -1;
    int loc_path_start_idx = 0, start_node_idx = 0;
    // outputs of the last get_ref_idx
    int cut_index_pos = 0, cut_layer = 0; double vel_plan = 0.0, acc_plan = 0.0; std::vector<double> vel_course;
    bool ref_done = false;                                  // get_ref_idx already ran for the current tick
    std::vector<std::pair<int, int>> path_ids;              // action_set_path_id of the last tick (key, id), dropped keys included

    Traj* find_last(int id) { for (auto& t : last) if (t.id == id) return &t; return nullptr; }
    BpTraj* find_bp(int id) { for (auto& t : last_bp) if (t.id == id) return &t; return nullptr; }
};

struct Config {
    int n_scen = 1;
    std::vector<double> w_last;
    double v_max_offset = 0.1, delaycomp = 0.1, calc_time_safety = 2.0;
    int calc_time_buffer_len = 5, filt_window_width = 1;
    double dyn_model_exp = 1.0, drag_coeff = 0.85, m_veh = 1000.0;
    int follow_control_type = 0; double c_p = 1.25, k_p = 0.2, k_d = 0.025, tan_w = 1.0;
};

// arguments of Graph_LTPL.calc_vel_profile (Graph_LTPL.py:344-408), per scenario
struct VelReq {
    double pos_x, pos_y, vel_est, vel_max, gg_scale, gg_ax, gg_ay, safety_d; int incl_emerg;
    // location dependent friction (local_gg as a dict, OTH.py:649-666): per path key (dict order) rows [ax, ay], or nullptr / 0 rows
    const double* gg_rows[LTPL_PLANNER_MAX_KEYS] = {}; int gg_n[LTPL_PLANNER_MAX_KEYS] = {};
};

// Persistent worker threads for the per-planner loops of a BATCH of planners (the state machines of different planners share nothing
// but the read-only lattice): run(f) calls f(t) for t = 0 .. size() - 1, f(0) on the calling thread, and returns when all are done. An
// exception thrown by any f is re-thrown on the calling thread after the others have finished (the ABI's try blocks sit there).
class WorkerPool {
public:
    explicit WorkerPool(int workers, int spin_us = 300) : spin_us_(spin_us) { for (int i = 0; i < workers; ++i) th.emplace_back([this, i] { loop(i + 1); }); }
    ~WorkerPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; gen.fetch_add(1, std::memory_order_release); }
        cv_go.notify_all();
        for (std::thread& t : th) t.join();
    }
    int size() const { return (int)th.size() + 1; }
    void run(const std::function<void(int)>& f)
    {
        { std::lock_guard<std::mutex> lk(mu); job = &f; pending.store((int)th.size(), std::memory_order_relaxed); eptr = nullptr; gen.fetch_add(1, std::memory_order_release); }
        if (sleepers.load(std::memory_order_acquire) > 0) cv_go.notify_all();
        std::exception_ptr mine;
        try { f(0); } catch (...) { mine = std::current_exception(); }
        // the regions are short: wait for the stragglers by polling first
        const auto t0 = std::chrono::steady_clock::now();
        while (pending.load(std::memory_order_acquire) != 0) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                std::unique_lock<std::mutex> lk(mu);
                cv_done.wait(lk, [this] { return pending.load(std::memory_order_acquire) == 0; });
                break;
            }
        }
        std::exception_ptr theirs;
        { std::lock_guard<std::mutex> lk(mu); job = nullptr; theirs = eptr; }
        if (mine) std::rethrow_exception(mine);
        if (theirs) std::rethrow_exception(theirs);
    }
private:
    void loop(int id)
    {
        long seen = 0;
        for (;;) {
            // A sleeping thread woken through the condition variable starts on the waker's core and only runs once the waker blocks
            // (measured: short regions then execute one after the other), so a worker polls the generation counter for spin_us_ after its
            // last job -- the regions of one tick follow each other within that window -- and only then goes to sleep.
            const auto t0 = std::chrono::steady_clock::now();
            bool got = false;
            while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us_)) {
                if (gen.load(std::memory_order_acquire) != seen) { got = true; break; }
            }
            const std::function<void(int)>* f;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (!got) {
                    sleepers.fetch_add(1, std::memory_order_release);
                    cv_go.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen; });
                    sleepers.fetch_sub(1, std::memory_order_release);
                }
                seen = gen.load(std::memory_order_acquire);
                if (stop) return;
                f = job;
            }
            std::exception_ptr e;
            try { (*f)(id); } catch (...) { e = std::current_exception(); }
            if (e) { std::lock_guard<std::mutex> lk(mu); if (!eptr) eptr = e; }
            if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(mu); cv_done.notify_one(); }
        }
    }
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const std::function<void(int)>* job = nullptr;
    std::exception_ptr eptr;
    std::atomic<long> gen{0};
    std::atomic<int> pending{0}, sleepers{0};
    int spin_us_;
    bool stop = false;
};

struct Planner {
    HostLat lat;
    Config cfg;
    Compute* cmp = nullptr;
    std::vector<Scn> sc;
    std::string err;

    // packed staging of seam (1)
    std::vector<int> p_start_layer, p_start_node, p_flags, p_last_action, p_const_closest, p_veh_off, p_pos_off, p_zone_off,
        p_zone_gid, p_n_last, p_last_layer, p_last_node;
    std::vector<double> p_psi_s, p_radius, p_px, p_py;
    std::vector<int> o_end_layer, o_coi, o_con, o_n_actions, o_action_id, o_valid, o_reduced, o_goal_layer, o_n_nodes, o_n_pts,
        o_n_ties, o_nodes, o_node_idx;
    std::vector<double> o_coeff, o_pp;

    // Batches (OPT-IN, LTPL_PLANNER_THREADS=<n>; default 1 = serial): the per-planner loops (objects + paths_pre, paths_post, stage A of the
    // velocity step: ~85 % of the host time of a tick, ~30 us per planner) run on n threads once a call carries at least kParMin
    // planners. The compute calls (kernel launches) always stay on the calling thread. Results are identical to the serial run (ranges
    // are contiguous and merged in order). EXPERIMENTAL: in the build container the regions only get faster while the workers are still
    // polling (LTPL_PLANNER_SPIN_US after their last job; 8 threads: 115 / 78 / 389 instead of 357 / 260 / 861 us) -- workers that went to
    // sleep take milliseconds to come back there and the tick gets SLOWER. Not measured on the GPU box yet.
    static constexpr int kParMin = 32, kParGrain = 8;
    int n_threads = default_threads();
    int spin_us = std::getenv("LTPL_PLANNER_SPIN_US") ? std::atoi(std::getenv("LTPL_PLANNER_SPIN_US")) : 300;
    std::unique_ptr<WorkerPool> pool;
    static int default_threads()
    {
        int t = 1;
        if (const char* e = std::getenv("LTPL_PLANNER_THREADS")) t = std::atoi(e);
        return std::min(std::max(t, 1), 64);
    }
    static std::string*& tl_err() { static thread_local std::string* p = nullptr; return p; }     // error sink of a worker's range

    ~Planner() { pool.reset(); delete cmp; }
    int fail(int code, const std::string& why) { (tl_err() ? *tl_err() : err) = why; return code; }
    // body(s0, s1, t) -> status for the planners [s0, s1) of part t; parts are contiguous and ordered, so "the first error" is the
    // error of the first failing part. Returns the number of parts through *parts (1 = ran serially on the calling thread).
    template <class F>
    int for_planner_ranges(int n, int* parts, F&& body)
    {
        const int T = (n >= kParMin && n_threads > 1) ? std::min(n_threads, n / kParGrain) : 1;
        if (parts) *parts = T;
        if (T <= 1) return body(0, n, 0);
        if (!pool || pool->size() != n_threads) pool.reset(new WorkerPool(n_threads - 1, spin_us));
        std::vector<int> rc((size_t)T, LTPL_OK);
        std::vector<std::string> errs((size_t)T);
        const int chunk = (n + T - 1) / T;
        pool->run([&](int t) {
            if (t >= T) return;
            const int s0 = t * chunk, s1 = std::min(n, s0 + chunk);
            if (s0 >= s1) return;
            struct Sink { std::string*& p; Sink(std::string*& q, std::string* to) : p(q) { p = to; } ~Sink() { p = nullptr; } } sink(tl_err(), &errs[(size_t)t]);
            rc[(size_t)t] = body(s0, s1, t);
        });
        for (int t = 0; t < T; ++t) if (rc[(size_t)t]) { err = errs[(size_t)t]; return rc[(size_t)t]; }
        return LTPL_OK;
    }
    int fail_cmp(int code) { err = cmp->last_error() ? cmp->last_error() : "compute backend failed"; return code; }

    // -----------------------------------------------------------------------------------------------------------------
    // check_inside_bounds.py:7-59 (start pose validation only)
    // -----------------------------------------------------------------------------------------------------------------
    bool inside_bounds(double px, double py) const
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

    static void reinit_memory(Scn& S)                   // OTH.py:161-179
    {
        S.has_start = false; S.has_last = false; S.last.clear(); S.has_bp = false; S.last_bp.clear();
        S.has_stamp = false; S.last_cut_idx = 0; S.has_pos = false;
        S.ref_done = false;                             // a reference index of the memory that is gone must not be reused
    }

    // -----------------------------------------------------------------------------------------------------------------
    // OTH.set_initial_pose (OTH.py:181-270): start spline from the pose into the lattice
    // -----------------------------------------------------------------------------------------------------------------
    int set_start(int s, double x, double y, double heading, double vel, double max_heading_offset, int* in_track, int* cor_heading)
    {
        if (s < 0 || s >= (int)sc.size()) return fail(LTPL_ERR_INVALID_ARG, "scenario index out of range");
        Scn& S = sc[(size_t)s];
        S.v_start = vel;
        *in_track = 1; *cor_heading = 1;
        reinit_memory(S);
        if (lat.nvx.empty()) return fail(LTPL_ERR_UNSUPPORTED, "planner: lattice without track bounds");
        if (!inside_bounds(x, y)) { *in_track = 0; return LTPL_OK; }
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
        S.last.clear(); S.last.push_back(T); S.has_last = true;
        S.action_forced = LTPL_ACT_STRAIGHT;
        return LTPL_OK;
    }

    void const_segment_test(const Scn& S, const double* seg, int seg_rows, int* in_const, int* besides, int* closest) const
    {
        std::vector<double> vx(S.veh.size()), vy(S.veh.size()), vr(S.veh.size());
        for (size_t k = 0; k < S.veh.size(); ++k) { vx[k] = S.veh[k].x; vy[k] = S.veh[k].y; vr[k] = S.veh[k].radius; }
        ltplp::const_segment_test(lat, seg, seg_rows, S.has_pos ? S.pos_est : nullptr, (int)S.veh.size(), vx.data(), vy.data(), vr.data(),
                                  in_const, besides, closest);
    }

    // -----------------------------------------------------------------------------------------------------------------
    // OTH.calc_paths, part in front of seam (1) (OTH.py:308-414)
    // -----------------------------------------------------------------------------------------------------------------
    int paths_pre(int s, int prev_action, double t_now)
    {
        Scn& S = sc[(size_t)s];
        int sel = prev_action;
        S.raw_action = prev_action;                    // Graph_LTPL hands the untranslated id to get_ref_idx (Graph_LTPL.py:384-387)
        if (sel == LTPL_ACT_EMERGENCY) sel = S.em_base_id;                               // :309-310
        if (S.action_forced != LTPL_ACT_NONE) { sel = S.action_forced; S.action_forced = LTPL_ACT_NONE; }
        S.sel_action = sel;
        Traj* lsel = S.has_last ? S.find_last(sel) : nullptr;
        BpTraj* bsel = S.has_bp ? S.find_bp(sel) : nullptr;
        S.const_exists = lsel != nullptr;
        const bool planned_once = S.has_stamp;
        const bool valid_last = planned_once && S.const_exists && bsel && bsel->rows() > 2;
        // backup plan = last straight / follow trajectory (:329-344)
        if (valid_last) {
            Traj* b = S.find_last(LTPL_ACT_FOLLOW);
            if (!b) b = S.find_last(LTPL_ACT_STRAIGHT);
            if (!b) return fail(LTPL_ERR_INVALID_ARG, "planner: neither 'straight' nor 'follow' in the last action set (the reference raises KeyError, OTH.py:334)");
            S.backup = *b; S.has_backup = true;
        } else S.has_backup = false;

        std::vector<int> last_sol;
        bool has_last_sol = false;
        S.loc_path_start_idx = 0; S.start_node_idx = 0;
        if (planned_once && valid_last) {
            const double calc_time = t_now - S.last_stamp;
            S.last_stamp = t_now;
            if ((int)S.calc_buffer.size() >= cfg.calc_time_buffer_len) S.calc_buffer.erase(S.calc_buffer.begin());
            S.calc_buffer.push_back(calc_time);
            double sum = 0.0; for (double c : S.calc_buffer) sum += c;
            const double avg = sum / (double)S.calc_buffer.size();
            const double t_const = std::min(avg * cfg.calc_time_safety, 0.5);
            // index of the pose reached after t_const on the last trajectory (:370-378)
            const double* bp = bsel->bp.data(); const int nb = bsel->rows();
            int first_false = 0; bool found = false; double cum = 0.0;
            for (int i = 0; i < nb - 2; ++i) {
                const double ds = bp[(size_t)(i + 2) * 7] - bp[(size_t)(i + 1) * 7], v = bp[(size_t)(i + 1) * 7 + 5];
                cum += (v != 0.0) ? ds / v : kInf;
                if (!(cum <= t_const)) { first_false = i; found = true; break; }
            }
            const int next_idx = (found ? first_false : 0) + 1;
            // first node behind that pose (:381-393)
            const int nn = (int)lsel->node_idx.size();
            std::vector<double> ncx((size_t)nn), ncy((size_t)nn);
            for (int i = 0; i < nn; ++i) { const int r = lsel->node_idx[(size_t)i]; ncx[(size_t)i] = lsel->pp[(size_t)r * 5]; ncy[(size_t)i] = lsel->pp[(size_t)r * 5 + 1]; }
            Poly np_{ncx.data(), ncy.data(), 1, nn};
            const Foot f = project_on_polyline(np_, bp[(size_t)next_idx * 7 + 1], bp[(size_t)next_idx * 7 + 2], false, false, [](int) { return 0.0; }, 0);
            S.start_node_idx = f.i1;
            S.loc_path_start_idx = lsel->node_idx[(size_t)S.start_node_idx];
            S.start_node[0] = lsel->nodes[(size_t)S.start_node_idx * 2]; S.start_node[1] = lsel->nodes[(size_t)S.start_node_idx * 2 + 1];
            S.has_start = true;
            last_sol.assign(lsel->nodes.begin() + (size_t)S.start_node_idx * 2, lsel->nodes.end());
            has_last_sol = true;
        } else {
            S.last_stamp = t_now; S.has_stamp = true;
            if (S.const_exists && S.has_start) {
                int idx = -1;
                for (int i = 0; i < lsel->n_nodes(); ++i)
                    if (lsel->nodes[(size_t)i * 2] == S.start_node[0] && lsel->nodes[(size_t)i * 2 + 1] == S.start_node[1]) { idx = i; break; }
                if (idx >= 0) {
                    const int g = lat.layer_off[(size_t)S.start_node[0]] + S.start_node[1];
                    Poly pl{lsel->pp.data(), lsel->pp.data() + 1, 5, lsel->rows()};
                    S.loc_path_start_idx = closest_index(pl, lat.node_x[(size_t)g], lat.node_y[(size_t)g]);
                    S.start_node_idx = idx;
                }
            }
        }
        if (!S.has_start) return fail(LTPL_ERR_INVALID_ARG, "planner: no start node (call ltpl_planner_set_start first)");
        // constant path segment (:412-414) and the packed seam-(1) call (:416-427)
        const double* seg = nullptr; int seg_rows = 0;
        S.const_rows = -1;
        if (S.const_exists) { seg = lsel->pp.data(); seg_rows = S.loc_path_start_idx + 1; S.const_rows = seg_rows; }
        int in_const, besides, cc;
        const_segment_test(S, seg, seg_rows, &in_const, &besides, &cc);
        int f = LTPL_FLAG_ACTION_SETS;
        if (in_const) f |= LTPL_FLAG_OBJ_IN_CONST;
        if (besides) f |= LTPL_FLAG_OBJ_BESIDES;
        if (seg) { f |= LTPL_FLAG_HAS_PSI_S; p_psi_s[(size_t)s] = seg[(size_t)(seg_rows - 1) * 5 + 2]; } else p_psi_s[(size_t)s] = 0.0;
        p_start_layer[(size_t)s] = S.start_node[0]; p_start_node[(size_t)s] = S.start_node[1];
        p_flags[(size_t)s] = f; p_last_action[(size_t)s] = sel; p_const_closest[(size_t)s] = cc;
        int k = 0;
        if (has_last_sol)
            for (size_t i = 0; i + 1 < last_sol.size() && k < LTPL_MAX_LAST_NODES; i += 2) {
                if (last_sol[i] == LTPLP_NONE || last_sol[i + 1] == LTPLP_NONE) break;
                p_last_layer[(size_t)s * LTPL_MAX_LAST_NODES + k] = last_sol[i];
                p_last_node[(size_t)s * LTPL_MAX_LAST_NODES + k] = last_sol[i + 1];
                ++k;
            }
        p_n_last[(size_t)s] = k;
        return LTPL_OK;
    }

    // -----------------------------------------------------------------------------------------------------------------
    // OTH.calc_paths behind seam (1): stitch the new paths behind the constant part (OTH.py:429-513)
    // -----------------------------------------------------------------------------------------------------------------
    int paths_post(int s)
    {
        Scn& S = sc[(size_t)s];
        const int A = LTPL_MAX_ACTIONS, cn = lat.max_path_nodes, cp = lat.max_path_pts;
        Traj* lsel = S.const_exists ? S.find_last(S.sel_action) : nullptr;
        Traj old;                                          // the dicts are replaced below; the old entry is still read (moved out: S.last is
        if (lsel) old = std::move(*lsel);                  // not looked at again before it is replaced at the end of this function)
        const int loc = S.loc_path_start_idx, sni = S.start_node_idx;
        std::vector<Traj> fresh;
        S.closest_obj_index = o_coi[(size_t)s];
        for (int a = 0; a < o_n_actions[(size_t)s]; ++a) {
            const size_t slot = (size_t)s * A + a;
            if (!o_valid[slot]) continue;
            Traj T; T.id = o_action_id[slot]; T.red_len = o_reduced[slot] != 0;
            const int nn = o_n_nodes[slot], npts = o_n_pts[slot];
            const int* nd = &o_nodes[slot * cn]; const int* ni = &o_node_idx[slot * cn];
            const double* co = &o_coeff[slot * cn * 8]; const double* pp = &o_pp[slot * cp * 5];
            if (lsel) {
                if (loc > 0) {
                    T.pp.assign(old.pp.begin(), old.pp.begin() + (size_t)loc * 5);
                    T.pp.insert(T.pp.end(), pp, pp + (size_t)npts * 5);
                    if (old.rows() == loc) {                                                 // :449-454
                        const int j = loc - 1;
                        const double dx = T.pp[(size_t)(j + 1) * 5] - T.pp[(size_t)j * 5], dy = T.pp[(size_t)(j + 1) * 5 + 1] - T.pp[(size_t)j * 5 + 1];
                        T.pp[(size_t)j * 5 + 4] = std::sqrt(dx * dx + dy * dy);
                    }
                } else T.pp.assign(pp, pp + (size_t)npts * 5);
                T.node_idx.assign(old.node_idx.begin(), old.node_idx.begin() + std::min<size_t>((size_t)sni, old.node_idx.size()));
                for (int i = 0; i < nn; ++i) T.node_idx.push_back(ni[i] + loc);
                if (sni > 0) {
                    T.nodes.assign(old.nodes.begin(), old.nodes.begin() + (size_t)sni * 2);
                    T.coeff.assign(old.coeff.begin(), old.coeff.begin() + std::min<size_t>((size_t)sni * 8, old.coeff.size()));
                }
            } else {
                T.pp.assign(pp, pp + (size_t)npts * 5);
                for (int i = 0; i < nn; ++i) T.node_idx.push_back(ni[i]);
            }
            for (int i = 0; i < nn; ++i) { T.nodes.push_back((S.start_node[0] + i) % lat.L); T.nodes.push_back(nd[i]); }
            T.coeff.insert(T.coeff.end(), co, co + (size_t)(nn - 1) * 8);
            fresh.push_back(std::move(T));
        }
        if (fresh.empty() && lsel && S.const_rows > 2) {
            // blocked track: keep the constant segment including its end node (:474-506)
            const int loc1 = loc + 1, sni1 = sni + 1;
            Traj T; T.id = S.sel_action; T.red_len = true;
            T.pp.assign(old.pp.begin(), old.pp.begin() + (size_t)std::min(loc1, old.rows()) * 5);
            T.node_idx.assign(old.node_idx.begin(), old.node_idx.begin() + std::min<size_t>((size_t)sni1, old.node_idx.size()));
            T.nodes.assign(old.nodes.begin(), old.nodes.begin() + std::min<size_t>((size_t)sni1 * 2, old.nodes.size()));
            T.coeff.assign(old.coeff.begin(), old.coeff.begin() + std::min<size_t>((size_t)sni1 * 8, old.coeff.size()));
            fresh.push_back(std::move(T));
        }
        S.last = std::move(fresh); S.has_last = true;
        return LTPL_OK;
    }

    // calc_paths in two halves so that a caller that owns the zone bookkeeping (gen_local_node_template.py:42-99 needs the start
    // node of THIS search) can step in between: begin = OTH.update_objects + OTH.py:308-414, finish = seam (1) + OTH.py:429-513
    std::vector<int> b_veh_off, b_pos_off; std::vector<double> b_radius, b_px, b_py;
    bool began = false;

    int calc_paths_begin(const int* prev_action, const double* t_now, const int* veh_off, const int* pos_off, const double* veh_radius,
                         const double* veh_vel, const double* pos_x, const double* pos_y)
    {
        const int n = (int)sc.size();
        p_start_layer.assign((size_t)n, 0); p_start_node.assign((size_t)n, 0); p_flags.assign((size_t)n, 0);
        p_last_action.assign((size_t)n, LTPL_ACT_NONE); p_const_closest.assign((size_t)n, -1); p_psi_s.assign((size_t)n, 0.0);
        p_n_last.assign((size_t)n, 0);
        p_last_layer.assign((size_t)n * LTPL_MAX_LAST_NODES, -1); p_last_node.assign((size_t)n * LTPL_MAX_LAST_NODES, -1);
        if (veh_off[0] != 0 || pos_off[0] != 0) return fail(LTPL_ERR_INVALID_ARG, "offset arrays must start at 0");
        const int nv = veh_off[n], np_ = pos_off[nv];
        b_veh_off.assign(veh_off, veh_off + n + 1); b_pos_off.assign(pos_off, pos_off + nv + 1);
        b_radius.assign(veh_radius, veh_radius + nv); b_px.assign(pos_x, pos_x + np_); b_py.assign(pos_y, pos_y + np_);
        b_radius.push_back(0.0); b_px.push_back(0.0); b_py.push_back(0.0);          // never empty
        // OTH.update_objects (OTH.py:272-287), then the part of OTH.calc_paths in front of seam (1); planner by planner, in parallel for batches
        LTPL_PROF(prof_pre, "planner.paths_pre");
        const int rc_all = for_planner_ranges(n, nullptr, [&](int s0, int s1, int) {
            for (int s = s0; s < s1; ++s) {
                Scn& S = sc[(size_t)s];
                S.veh.clear();
                for (int v = veh_off[s]; v < veh_off[s + 1]; ++v) {
                    ObjVeh o; o.radius = veh_radius[v]; o.vel = veh_vel ? veh_vel[v] : 0.0;
                    if (pos_off[v + 1] - pos_off[v] < 1) return fail(LTPL_ERR_INVALID_ARG, "vehicle without position");
                    o.x = pos_x[pos_off[v]]; o.y = pos_y[pos_off[v]];
                    for (int p = pos_off[v]; p < pos_off[v + 1]; ++p) { o.pos.push_back(pos_x[p]); o.pos.push_back(pos_y[p]); }
                    S.veh.push_back(std::move(o));
                }
                S.closest_obj_index = -1;
                S.ref_done = false;                 // new paths: a reference index computed for the previous memory is stale
                const int rc = paths_pre(s, prev_action[s], t_now[s]);
                if (rc) return rc;
            }
            return (int)LTPL_OK;
        });
        if (rc_all) return rc_all;
        began = true;
        return LTPL_OK;
    }

    int calc_paths_finish(const int* zone_off, const int* zone_gid)
    {
        if (!began) return fail(LTPL_ERR_INVALID_ARG, "planner: calc_paths_finish without calc_paths_begin");
        began = false;
        const int n = (int)sc.size(), A = LTPL_MAX_ACTIONS, cn = lat.max_path_nodes, cp = lat.max_path_pts;
        ltpl_paths_in in; std::memset(&in, 0, sizeof(in));
        in.n_scen = n; in.n_w_last = (int)cfg.w_last.size();
        static const double zero = 0.0;
        in.w_last_edges = cfg.w_last.empty() ? &zero : cfg.w_last.data();
        in.start_layer = p_start_layer.data(); in.start_node = p_start_node.data(); in.flags = p_flags.data();
        in.last_action = p_last_action.data(); in.const_closest = p_const_closest.data(); in.psi_s = p_psi_s.data();
        in.veh_off = b_veh_off.data(); in.pos_off = b_pos_off.data(); in.veh_radius = b_radius.data(); in.pos_x = b_px.data(); in.pos_y = b_py.data();
        in.zone_off = zone_off; in.zone_gid = zone_gid;
        in.n_last = p_n_last.data(); in.last_layer = p_last_layer.data(); in.last_node = p_last_node.data();
        o_end_layer.resize((size_t)n); o_coi.resize((size_t)n); o_con.resize((size_t)n * 2); o_n_actions.resize((size_t)n);
        for (auto* v : {&o_action_id, &o_valid, &o_reduced, &o_goal_layer, &o_n_nodes, &o_n_pts, &o_n_ties}) v->resize((size_t)n * A);
        o_nodes.resize((size_t)n * A * cn); o_node_idx.resize((size_t)n * A * cn);
        o_coeff.resize((size_t)n * A * cn * 8); o_pp.resize((size_t)n * A * cp * 5);
        ltpl_paths_out out; std::memset(&out, 0, sizeof(out));
        out.cap_nodes = cn; out.cap_pts = cp;
        out.end_layer = o_end_layer.data(); out.closest_obj_index = o_coi.data(); out.closest_obj_node = o_con.data();
        out.n_actions = o_n_actions.data(); out.action_id = o_action_id.data(); out.valid = o_valid.data();
        out.reduced = o_reduced.data(); out.goal_layer = o_goal_layer.data(); out.n_nodes = o_n_nodes.data();
        out.n_pts = o_n_pts.data(); out.n_ties = o_n_ties.data(); out.nodes = o_nodes.data(); out.node_idx = o_node_idx.data();
        out.coeff = o_coeff.data(); out.path_param = o_pp.data();
        int rc = cmp->plan_paths(&in, &out);
        if (rc) return fail_cmp(rc);
        LTPL_PROF(prof_post, "planner.paths_post");
        return for_planner_ranges(n, nullptr, [&](int s0, int s1, int) {
            for (int s = s0; s < s1; ++s) { const int r = paths_post(s); if (r) return r; }
            return (int)LTPL_OK;
        });
    }

    int calc_paths(const int* prev_action, const double* t_now, const int* veh_off, const int* pos_off, const double* veh_radius,
                   const double* veh_vel, const double* pos_x, const double* pos_y, const int* zone_off, const int* zone_gid)
    {
        int rc = calc_paths_begin(prev_action, t_now, veh_off, pos_off, veh_radius, veh_vel, pos_x, pos_y);
        if (rc) return rc;
        return calc_paths_finish(zone_off, zone_gid);
    }

    // -----------------------------------------------------------------------------------------------------------------
    // OTH.get_ref_idx (OTH.py:518-601)
    // -----------------------------------------------------------------------------------------------------------------
    void ref_idx(Scn& S, double px, double py)
    {
        S.pos_est[0] = px; S.pos_est[1] = py; S.has_pos = true;
        BpTraj* b = S.has_bp ? S.find_bp(S.raw_action) : nullptr;
        const bool valid_last = b && b->rows() > 0;
        const bool valid_this = !S.last.empty();
        int cut_index_layer = 0;
        S.vel_course.clear();
        if (valid_last) {
            const double* bp = b->bp.data(); const int n = b->rows();
            Poly pl{bp + 1, bp + 2, 7, n};
            const Foot f = project_on_polyline(pl, px, py, false, false, [](int) { return 0.0; }, 0);
            const int cut = f.i0;
            const int m = n - cut - 1;                                   // len(v_past) (:565-567)
            int first_false = 0; bool found = false; double cum = 0.0;
            for (int i = 0; i < m; ++i) {
                const double ds = bp[(size_t)(cut + i + 1) * 7] - bp[(size_t)(cut + i) * 7], v = bp[(size_t)(cut + i) * 7 + 5];
                cum += (v != 0.0) ? ds / v : kInf;
                if (!(cum <= cfg.delaycomp)) { first_false = i; found = true; break; }
            }
            int vel_idx = std::min((found ? first_false : 0) + 1, m - 1);
            if (vel_idx < 0) vel_idx = 0;
            S.vel_plan = bp[(size_t)(cut + vel_idx) * 7 + 5]; S.acc_plan = bp[(size_t)(cut + vel_idx) * 7 + 6];
            for (int i = cut; i < cut + vel_idx; ++i) S.vel_course.push_back(bp[(size_t)i * 7 + 5]);
            S.cut_index_pos = S.last_cut_idx + cut;
            if (valid_this) {
                const std::vector<int>& ni = S.last[0].node_idx;           // first key of the dict (:581)
                int ff = 0;
                for (size_t i = 0; i < ni.size(); ++i) if (!(ni[i] < S.cut_index_pos)) { ff = (int)i; break; }
                S.cut_layer = std::max(ff - 2, 0);
                cut_index_layer = ni.empty() ? 0 : ni[(size_t)S.cut_layer];
            } else { S.cut_layer = 0; cut_index_layer = 0; }
        } else {
            S.cut_index_pos = 0; S.cut_layer = 0; cut_index_layer = 0;
            S.vel_plan = S.v_start; S.acc_plan = 0.0;
        }
        S.last_cut_idx = S.cut_index_pos - cut_index_layer;
    }

    // get_ref_idx as its own call (Graph_LTPL.py:380-383 calls it right before calc_vel_profile); calc_vel_profile then reuses it
    int get_ref_idx(const double* px, const double* py)
    {
        for (size_t s = 0; s < sc.size(); ++s) { ref_idx(sc[s], px[s], py[s]); sc[s].ref_done = true; }
        return LTPL_OK;
    }

    // -----------------------------------------------------------------------------------------------------------------
    // OTH.calc_vel_profile (OTH.py:603-1040), batched: all seam-(2) jobs of a stage go out in one launch
    // -----------------------------------------------------------------------------------------------------------------
    struct Work {                       // per (scenario, action key) of the current tick
        int s; size_t key;              // index into S.last
        std::vector<double> pv, gv;     // action_set_path_param_vel / _gg (rows from cut_index_pos on)
        std::vector<double> s_arr;
        int cut_index_layer = 0;
        int n = 0, vel_idx = 0, pref_idx = 0, v_idx = 0;
        double vel_start = 0.0;
        int job_follow = -1, job_free = -1, job_fb = -1, job_backup = -1;
        bool generic = false, has_fb = false;
        int too_close = 0, vel_bound = 1;
        std::vector<double> bp;         // result rows of 7
        bool keep = false, drop = false, empty = false;
    };
    struct JobBuf { std::vector<double> kappa, el, gg, out; };

    int run_jobs(const ltpl_vel_params& vp, std::vector<ltpl_vel_job>& jobs, std::vector<JobBuf>& bufs, std::vector<ltpl_vel_result>& res)
    {
        if (jobs.empty()) return LTPL_OK;
        res.resize(jobs.size());
        for (size_t j = 0; j < jobs.size(); ++j) {
            jobs[j].kappa = bufs[j].kappa.data(); jobs[j].el_lengths = bufs[j].el.data(); jobs[j].loc_gg = bufs[j].gg.data();
            bufs[j].out.assign((size_t)jobs[j].n, 0.0);
            res[j].vx = bufs[j].out.data(); res[j].too_close = 0; res[j].vel_bound = 1;
        }
        const int rc = cmp->vel_profile(&vp, (int)jobs.size(), jobs.data(), res.data());
        return rc ? fail_cmp(rc) : LTPL_OK;
    }

    // tph.conv_filt(signal, filt_window, closed=False) (OTH.py:928-930, :988-990): centred moving average of odd width; the first and
    // the last half window keep their values (np.convolve(..., "same") only replaces [half, n - half))
    static void conv_filt_open(const double* vx, int n, int width, std::vector<double>* out)
    {
        out->assign(vx, vx + n);
        const int half = (width - 1) / 2;
        if (half < 1 || n < width) return;              // width 1: identity (the stock value, ltpl_config_online.ini:60)
        for (int i = half; i < n - half; ++i) {
            double acc = 0.0;
            for (int k = i - half; k <= i + half; ++k) acc += vx[k] * (1.0 / (double)width);
            (*out)[(size_t)i] = acc;
        }
    }

    void finalize_bp(const double* s_arr, const double* pv, const double* vx_in, int n, std::vector<double>* bp) const
    {
        // :925-941: vx filtered (conv_filt), ax from neighbours (tph.calc_ax_profile over np.diff(s)), -5 at standstill
        std::vector<double> vxf;
        conv_filt_open(vx_in, n, cfg.filt_window_width, &vxf);
        const double* vx = vxf.data();
        bp->assign((size_t)n * 7, 0.0);
        for (int i = 0; i < n; ++i) {
            double* r = &(*bp)[(size_t)i * 7];
            r[0] = s_arr[i]; r[1] = pv[(size_t)i * 5]; r[2] = pv[(size_t)i * 5 + 1]; r[3] = pv[(size_t)i * 5 + 2]; r[4] = pv[(size_t)i * 5 + 3];
            r[5] = vx[i];
            if (i + 1 < n) {
                double ax = (std::pow(vx[i + 1], 2) - std::pow(vx[i], 2)) / (2 * (s_arr[i + 1] - s_arr[i]));
                if (std::fabs(vx[i]) <= 1e-8 && std::fabs(ax) <= 1e-8) ax = -5.0;
                r[6] = ax;
            }
        }
    }

    int calc_vel_profile(const VelReq* req, const double* ax_max_machines, int n_axm, const double* /*t_now*/)
    {
        const int n = (int)sc.size();
        LTPL_PROF(prof_a, "planner.vel_stage_A");
        if (cfg.filt_window_width < 1 || cfg.filt_window_width % 2 != 1)
            return fail(LTPL_ERR_INVALID_ARG, "planner: Window width of moving average filter must be odd! (tph.conv_filt)");
        ltpl_vel_params vp; std::memset(&vp, 0, sizeof(vp));
        vp.dyn_model_exp = cfg.dyn_model_exp; vp.drag_coeff = cfg.drag_coeff; vp.m_veh = cfg.m_veh; vp.len_veh = lat.veh_length;
        vp.n_ax_max_machines = n_axm; vp.ax_max_machines = ax_max_machines; vp.follow_control_type = cfg.follow_control_type;
        vp.c_p = cfg.c_p; vp.k_p = cfg.k_p; vp.k_d = cfg.k_d; vp.tan_w = cfg.tan_w;
        // one parameter set per launch: vel_max is part of it, so scenarios must agree (they do for one planner; a batch
        // with differing vel_max is split by the caller)
        vp.v_max = req[0].vel_max;
        for (int s = 1; s < n; ++s) if (req[s].vel_max != req[0].vel_max) return fail(LTPL_ERR_UNSUPPORTED, "planner: vel_max must be the same for all scenarios of a call");

        // Everything that can make the call fail is checked for ALL planners before any planner's memory is touched: an error of one
        // planner of a batch (the reference's ValueError / IndexError for that vehicle) must not leave the others half-trimmed.
        // (A failing pre-check must leave NO trace: the reference index it computed belongs to this call's position estimate, so the
        //  flag that lets stage A skip ref_idx is cleared again for every planner before the error is returned -- otherwise the next
        //  calc_vel_profile without a get_ref_idx in front of it would cut with this call's stale indices.)
        auto fail_pre = [&](int code, const std::string& msg) { for (Scn& S : sc) S.ref_done = false; return fail(code, msg); };
        for (int s = 0; s < n; ++s) {
            Scn& S = sc[(size_t)s];
            if (!S.ref_done) { ref_idx(S, req[s].pos_x, req[s].pos_y); S.ref_done = true; }      // (OTH.get_ref_idx: no iterative memory is cut here)
            for (size_t k = 0; k < S.last.size(); ++k) {
                const Traj& T = S.last[k];
                const int rows = T.rows(), m = rows - std::min(std::max(S.cut_index_pos, 0), rows);
                if (S.cut_layer >= (int)T.node_idx.size())
                    return fail_pre(LTPL_ERR_INVALID_ARG, "planner " + std::to_string(s) + ": cut_layer beyond the node list (the reference raises IndexError, OTH.py:712)");
                if (k < (size_t)LTPL_PLANNER_MAX_KEYS && req[s].gg_rows[k] && req[s].gg_n[k] > 0 && req[s].gg_n[k] != rows)
                    return fail_pre(LTPL_ERR_INVALID_ARG, "planner " + std::to_string(s) + ": local_gg rows of a path do not match its coordinates (OTH.py:641-646)");
                if (m > 0 && S.vel_plan > req[s].vel_max + 0.1)
                    return fail_pre(LTPL_ERR_UNSUPPORTED, "planner " + std::to_string(s) + ": vel_plan > vel_max + 0.1 (brake prefix): the reference raises ValueError at OTH.py:919");
                if (m > 0 && T.id == LTPL_ACT_FOLLOW && m - (int)S.vel_course.size() < 1)
                    return fail_pre(LTPL_ERR_INVALID_ARG, "planner " + std::to_string(s) + ": follow profile without points");
            }
        }
        std::vector<Work> work;
        std::vector<ltpl_vel_job> jobs; std::vector<JobBuf> bufs; std::vector<ltpl_vel_result> res;
        work.reserve((size_t)n * 3); jobs.reserve((size_t)n * 4); bufs.reserve((size_t)n * 4);     // (<= 3 keys and <= 4 jobs per planner, typically)
        // ---- stage A: get_ref_idx, slicing (:700-731), job construction (:736-903) ---------------------------------------
        // (planners [s0, s1) into the given lists: a batch is cut into ranges that fill their own lists in parallel, merged in order below)
        auto stage_a = [&](int s0, int s1, std::vector<Work>& work, std::vector<ltpl_vel_job>& jobs, std::vector<JobBuf>& bufs) -> int {
        for (int s = s0; s < s1; ++s) {
            Scn& S = sc[(size_t)s];
            const VelReq& R = req[s];
            if (!S.ref_done) ref_idx(S, R.pos_x, R.pos_y);
            S.ref_done = false;
            S.traj_base_id += 10;
            // VpForwardBackward.update_dyn_parameters (:65-84)
            if (!S.has_old_gg) { S.old_gg_scale = R.gg_scale; S.has_old_gg = true; }
            S.last_bp.clear(); S.has_bp = true; S.path_ids.clear();
            const int vel_idx = (int)S.vel_course.size();
            for (size_t k = 0; k < S.last.size(); ++k) {
                Traj& T = S.last[k];
                Work W; W.s = s; W.key = k; W.vel_idx = vel_idx;
                S.path_ids.push_back({T.id, S.traj_base_id + (T.id >= 0 && T.id <= 3 ? T.id : 9)});
                const int rows = T.rows();
                // local gg rows of the stitched path: constant friction expanded to one row per path coordinate (:651-666), or the
                // caller's rows for this key (location dependent friction, :641-646: "each path coordinate must be represented by a row")
                std::vector<double> gg_full((size_t)rows * 2);
                if (k < (size_t)LTPL_PLANNER_MAX_KEYS && R.gg_rows[k] && R.gg_n[k] > 0) {
                    if (R.gg_n[k] != rows) return fail(LTPL_ERR_INVALID_ARG, "planner: local_gg rows of a path do not match its coordinates (OTH.py:641-646)");
                    std::memcpy(gg_full.data(), R.gg_rows[k], sizeof(double) * 2 * (size_t)rows);
                } else
                    for (int i = 0; i < rows; ++i) { gg_full[(size_t)i * 2] = R.gg_ax; gg_full[(size_t)i * 2 + 1] = R.gg_ay; }
                const int c0 = std::min(std::max(S.cut_index_pos, 0), rows);
                W.pv.assign(T.pp.begin() + (size_t)c0 * 5, T.pp.end());
                W.gv.assign(gg_full.begin() + (size_t)c0 * 2, gg_full.end());
                if (S.cut_layer >= (int)T.node_idx.size()) return fail(LTPL_ERR_INVALID_ARG, "planner: cut_layer beyond the node list (the reference raises IndexError, OTH.py:712)");
                const int cil = T.node_idx[(size_t)S.cut_layer];
                W.cut_index_layer = cil;
                {   // trim the memory for the next iteration, aligned with the nodes (:714-731)
                    std::vector<int> ni(T.node_idx.begin() + S.cut_layer, T.node_idx.end());
                    for (int& v : ni) v -= cil;
                    T.node_idx.swap(ni);
                    const int c1 = std::min(std::max(cil, 0), rows);
                    T.pp.erase(T.pp.begin(), T.pp.begin() + (size_t)c1 * 5);
                    T.gg.assign(gg_full.begin() + (size_t)c1 * 2, gg_full.end());
                    const size_t cc = std::min<size_t>((size_t)S.cut_layer * 8, T.coeff.size());
                    T.coeff.erase(T.coeff.begin(), T.coeff.begin() + cc);
                    const size_t cnn = std::min<size_t>((size_t)S.cut_layer * 2, T.nodes.size());
                    T.nodes.erase(T.nodes.begin(), T.nodes.begin() + cnn);
                }
                const int m = (int)(W.pv.size() / 5);
                W.n = m;
                if (m == 0) { W.empty = true; work.push_back(std::move(W)); continue; }
                W.s_arr.assign((size_t)m, 0.0);
                for (int i = 1; i < m; ++i) W.s_arr[(size_t)i] = W.s_arr[(size_t)i - 1] + W.pv[(size_t)(i - 1) * 5 + 4];   // :743
                // VpForwardBackward.check_brake_prefix (:86-139): with a non-empty prefix the reference cannot assemble the
                // trajectory (vx is shorter than s at OTH.py:919 / :830 -> ValueError), so the branch is reported, not emulated
                if (S.vel_plan > R.vel_max + 0.1) return fail(LTPL_ERR_UNSUPPORTED, "planner: vel_plan > vel_max + 0.1 (brake prefix): the reference raises ValueError at OTH.py:919");
                S.old_gg_scale = R.gg_scale;
                W.pref_idx = vel_idx; W.vel_start = S.vel_plan;
                const int pref = W.pref_idx;
                auto make_job = [&](int mode, int i0, int i1, int n_el, double v_start, bool has_end, double v_end) {
                    ltpl_vel_job jb; std::memset(&jb, 0, sizeof(jb));
                    JobBuf B;
                    jb.mode = mode; jb.n = i1 - i0; jb.n_el = n_el; jb.has_v_end = has_end ? 1 : 0;
                    jb.v_start = v_start; jb.v_end = v_end;
                    B.kappa.reserve((size_t)(i1 - i0)); B.gg.reserve((size_t)(i1 - i0) * 2); B.el.reserve((size_t)std::max(n_el, 1));
                    for (int i = i0; i < i1; ++i) {
                        B.kappa.push_back(W.pv[(size_t)i * 5 + 3]);
                        B.gg.push_back(W.gv[(size_t)i * 2] * R.gg_scale); B.gg.push_back(W.gv[(size_t)i * 2 + 1] * R.gg_scale);
                    }
                    for (int i = i0; i < i0 + n_el; ++i) B.el.push_back(W.pv[(size_t)i * 5 + 4]);
                    if (B.el.empty()) B.el.push_back(0.0);
                    jobs.push_back(jb); bufs.push_back(std::move(B));
                    return (int)jobs.size() - 1;
                };
                if (T.id == LTPL_ACT_FOLLOW) {                                             // :763-830
                    if (m - pref < 1) return fail(LTPL_ERR_INVALID_ARG, "planner: follow profile without points");
                    double obj_dist = 0.0, v_obj = 0.0, ox = R.pos_x, oy = R.pos_y;
                    if (S.closest_obj_index >= 0 && S.closest_obj_index < (int)S.veh.size()) {
                        const ObjVeh& o = S.veh[(size_t)S.closest_obj_index];
                        ox = o.x; oy = o.y; v_obj = o.vel;
                        // s_array = cumsum(path[:, 4]) (:777,782)
                        std::vector<double> cs((size_t)m); double acc = 0.0;
                        for (int i = 0; i < m; ++i) { acc += W.pv[(size_t)i * 5 + 4]; cs[(size_t)i] = acc; }
                        Poly pl{W.pv.data(), W.pv.data() + 1, 5, m};
                        auto s_at = [&](int i) { return cs[(size_t)i]; };
                        const double s_obj = project_on_polyline(pl, ox, oy, false, true, s_at, m).s;
                        const double s_sta = project_on_polyline(pl, S.pos_est[0], S.pos_est[1], false, true, s_at, m).s;
                        obj_dist = s_obj - s_sta;
                    }
                    // the two halves of the follow mode are independent (calc_vel_profile_follow.py:151-294 vs :297-307): two
                    // jobs = two waves in parallel on the device, intersected in stage B (:310)
                    const int j = make_job(LTPL_VEL_FOLLOW_CONTROLLED, pref, m, m - pref, W.vel_start, false, 0.0);
                    jobs[(size_t)j].v_ego = R.vel_est; jobs[(size_t)j].v_obj = v_obj; jobs[(size_t)j].safety_d = R.safety_d;
                    jobs[(size_t)j].obj_dist = obj_dist; jobs[(size_t)j].obj_x = ox; jobs[(size_t)j].obj_y = oy;
                    W.job_follow = j;
                    W.job_free = make_job(LTPL_VEL_FB, pref, m, m - pref - 1, W.vel_start, false, 0.0);
                }
                if (T.id != LTPL_ACT_FOLLOW || T.red_len) {                                // :834-903
                    W.generic = true;
                    const int nn = T.n_nodes();
                    if (nn < 1) return fail(LTPL_ERR_INVALID_ARG, "planner: trajectory without nodes");
                    const int el_ = T.nodes[(size_t)(nn - 1) * 2], en = T.nodes[(size_t)(nn - 1) * 2 + 1];
                    if (el_ < 0 || el_ >= lat.L) return fail(LTPL_ERR_INVALID_ARG, "planner: end node is None");
                    const double raceline_offset = std::abs(en - lat.rl_idx[(size_t)el_]) * lat.lat_offset;
                    double v_end; int v_idx;
                    if (T.red_len) {
                        v_end = 0.0;
                        double spl = 0.0; for (int i = 0; i < m - 1; ++i) spl += W.pv[(size_t)i * 5 + 4];
                        int ff = 0; double c = 0.0;
                        for (int i = 0; i < m - 1; ++i) { c += W.pv[(size_t)i * 5 + 4]; if (!(c < (spl - 5.0))) { ff = i; break; } }
                        v_idx = ff + 1;
                        if (v_idx == 1 && m > 1) v_idx = m;
                    } else {
                        v_end = lat.vel_rl[(size_t)el_];
                        v_end -= std::min(v_end * lat.vel_decrease_lat * raceline_offset, v_end);
                        v_idx = m;
                    }
                    W.v_idx = v_idx;
                    if (v_idx - pref > 1) { W.job_fb = make_job(LTPL_VEL_FB, pref, v_idx, v_idx - pref - 1, W.vel_start, true, v_end); W.has_fb = true; }
                }
                work.push_back(std::move(W));
            }
        }
        return LTPL_OK;
        };
        {
            struct Part { std::vector<Work> work; std::vector<ltpl_vel_job> jobs; std::vector<JobBuf> bufs; };
            std::vector<Part> part((size_t)std::max(n_threads, 1));
            int parts = 1;
            const int rc_a = for_planner_ranges(n, &parts, [&](int s0, int s1, int t) {
                return parts == 1 ? stage_a(s0, s1, work, jobs, bufs) : stage_a(s0, s1, part[(size_t)t].work, part[(size_t)t].jobs, part[(size_t)t].bufs);
            });
            if (rc_a) return rc_a;
            if (parts > 1)
                for (int t = 0; t < parts; ++t) {                 // ranges are ordered by planner: same order as the serial loop
                    Part& P = part[(size_t)t];
                    const int off = (int)jobs.size();
                    for (Work& W : P.work) {
                        if (W.job_follow >= 0) W.job_follow += off;
                        if (W.job_free >= 0) W.job_free += off;
                        if (W.job_fb >= 0) W.job_fb += off;
                        if (W.job_backup >= 0) W.job_backup += off;
                        work.push_back(std::move(W));
                    }
                    jobs.insert(jobs.end(), P.jobs.begin(), P.jobs.end());
                    for (JobBuf& B : P.bufs) bufs.push_back(std::move(B));
                }
        }
        prof_a.stop();
        int rc = run_jobs(vp, jobs, bufs, res);
        if (rc) return rc;
        LTPL_PROF(prof_b, "planner.vel_stage_B");

        // ---- stage B: assemble trajectories (:824-941), decide keep / drop / backup (:943-1015) ---------------------------
        std::vector<ltpl_vel_job> jobs2; std::vector<JobBuf> bufs2; std::vector<ltpl_vel_result> res2;
        for (Work& W : work) {
            Scn& S = sc[(size_t)W.s];
            Traj& T = S.last[W.key];
            const int m = W.n, vel_idx = W.vel_idx;
            std::vector<double> vx_follow, vx;
            bool have_bp = false;
            W.vel_bound = 1;
            if (!W.empty) {
                if (W.job_follow >= 0) {
                    const ltpl_vel_result& r = res[(size_t)W.job_follow];
                    W.too_close = r.too_close; W.vel_bound = r.vel_bound;
                    vx_follow = S.vel_course;
                    {
                        std::vector<double> f = bufs[(size_t)W.job_follow].out;
                        if (W.job_free >= 0) {
                            const std::vector<double>& u = bufs[(size_t)W.job_free].out;          // np.minimum(vx_profile, vx_compl) (:310)
                            for (size_t i = 0; i < f.size() && i < u.size(); ++i) f[i] = f[i] < u[i] ? f[i] : u[i];
                        }
                        vx_follow.insert(vx_follow.end(), f.begin(), f.end());
                    }
                    if ((int)vx_follow.size() > m) vx_follow.resize((size_t)m);
                    if ((int)vx_follow.size() != m) return fail(LTPL_ERR_INVALID_ARG, "planner: follow profile shorter than the path (the reference raises at OTH.py:830)");
                    have_bp = true;
                }
                if (W.generic) {
                    std::vector<double> g;
                    if (W.has_fb) g = bufs[(size_t)W.job_fb].out; else g.assign(1, 0.0);
                    if (W.v_idx != m || W.v_idx <= 2) g.insert(g.end(), (size_t)std::max(m - W.v_idx, 0), 0.0);      // :901-903
                    W.vel_bound = std::fabs(g[0] - S.vel_plan) < cfg.v_max_offset ? 1 : 0;                        // :906-911
                    vx = S.vel_course; vx.insert(vx.end(), g.begin(), g.end());
                    if ((int)vx.size() > m) vx.resize((size_t)m);
                    if ((int)vx.size() != m) return fail(LTPL_ERR_INVALID_ARG, "planner: velocity profile shorter than the path (the reference raises at OTH.py:919)");
                    if (have_bp) {
                        // :923 compares ROW 5 of both arrays; only column 5 can differ, so the whole vx column switches
                        if (m < 6) return fail(LTPL_ERR_INVALID_ARG, "planner: fewer than 6 rows (the reference raises IndexError at OTH.py:923)");
                        if (vx_follow[5] < vx[5]) vx = vx_follow;
                    }
                } else vx = vx_follow;
                finalize_bp(W.s_arr.data(), W.pv.data(), vx.data(), m, &W.bp);
            }
            const bool sf = T.id == LTPL_ACT_FOLLOW || T.id == LTPL_ACT_STRAIGHT;
            if (W.vel_bound || sf) {
                if (W.vel_bound || !S.has_backup) W.keep = true;
                else {
                    // recursive infeasibility: brake on the previous solution (:950-1006)
                    const Traj& B = S.backup;
                    const int cl = S.cut_layer, cil = W.cut_index_layer;
                    if (cl > (int)B.node_idx.size()) return fail(LTPL_ERR_INVALID_ARG, "planner: cut_layer beyond the backup plan");
                    T.node_idx.assign(B.node_idx.begin() + cl, B.node_idx.end());
                    for (int& v : T.node_idx) v -= cil;
                    const int br = B.rows();
                    const int c1 = std::min(std::max(cil, 0), br);
                    T.pp.assign(B.pp.begin() + (size_t)c1 * 5, B.pp.end());
                    T.gg.assign(B.gg.begin() + std::min<size_t>((size_t)c1 * 2, B.gg.size()), B.gg.end());
                    T.coeff.assign(B.coeff.begin() + std::min<size_t>((size_t)cl * 8, B.coeff.size()), B.coeff.end());
                    T.nodes.assign(B.nodes.begin() + std::min<size_t>((size_t)cl * 2, B.nodes.size()), B.nodes.end());
                    const int i0 = S.cut_index_pos + vel_idx;
                    if (i0 >= br) return fail(LTPL_ERR_INVALID_ARG, "planner: backup plan shorter than the cut index");
                    ltpl_vel_job jb; std::memset(&jb, 0, sizeof(jb)); JobBuf Bf;
                    jb.mode = LTPL_VEL_BRAKE; jb.n = br - i0; jb.n_el = jb.n - 1; jb.v_start = S.vel_plan;
                    for (int i = i0; i < br; ++i) {
                        Bf.kappa.push_back(B.pp[(size_t)i * 5 + 3]);
                        Bf.gg.push_back(B.gg[(size_t)i * 2]); Bf.gg.push_back(B.gg[(size_t)i * 2 + 1]);       // no gg_scale (:229-255)
                    }
                    for (int i = i0; i < br - 1; ++i) Bf.el.push_back(B.pp[(size_t)i * 5 + 4]);
                    if (Bf.el.empty()) Bf.el.push_back(0.0);
                    jobs2.push_back(jb); bufs2.push_back(std::move(Bf));
                    W.job_backup = (int)jobs2.size() - 1;
                    W.keep = true;
                }
            } else W.drop = true;
        }
        prof_b.stop();
        if ((rc = run_jobs(vp, jobs2, bufs2, res2))) return rc;
        LTPL_PROF(prof_c, "planner.vel_stage_C");

        // ---- stage C: backup trajectories, commit, emergency profile ---------------------------------------------------------
        std::vector<ltpl_vel_job> jobs3; std::vector<JobBuf> bufs3; std::vector<ltpl_vel_result> res3;
        std::vector<std::pair<int, int>> em_of;       // (scenario, job)
        std::vector<std::vector<double>> gv_first((size_t)n);       // action_set_path_param_gg of the first remaining key
        {
            size_t wi = 0;
            for (int s = 0; s < n; ++s) {
                Scn& S = sc[(size_t)s];
                std::vector<Traj> kept;
                const size_t nk = S.last.size();
                for (size_t k = 0; k < nk; ++k, ++wi) {
                    Work& W = work[wi];
                    Traj& T = S.last[k];
                    if (W.drop) continue;                                                   // :1007-1025
                    if (W.job_backup >= 0) {
                        const Traj& B = S.backup;
                        const int c0 = S.cut_index_pos, br = B.rows(), m = br - c0;
                        std::vector<double> vx_raw = S.vel_course, vx;
                        vx_raw.insert(vx_raw.end(), bufs2[(size_t)W.job_backup].out.begin(), bufs2[(size_t)W.job_backup].out.end());
                        if ((int)vx_raw.size() != m) return fail(LTPL_ERR_INVALID_ARG, "planner: backup brake profile length mismatch");
                        conv_filt_open(vx_raw.data(), m, cfg.filt_window_width, &vx);             // :986-990
                        std::vector<double> s_arr((size_t)m, 0.0);
                        for (int i = 1; i < m; ++i) s_arr[(size_t)i] = s_arr[(size_t)i - 1] + B.pp[(size_t)(c0 + i - 1) * 5 + 4];
                        // :996-1004: ax over the element lengths themselves, not over np.diff(s)
                        W.bp.assign((size_t)m * 7, 0.0);
                        for (int i = 0; i < m; ++i) {
                            double* r = &W.bp[(size_t)i * 7]; const double* p = &B.pp[(size_t)(c0 + i) * 5];
                            r[0] = s_arr[(size_t)i]; r[1] = p[0]; r[2] = p[1]; r[3] = p[2]; r[4] = p[3]; r[5] = vx[(size_t)i];
                            if (i + 1 < m) {
                                double ax = (std::pow(vx[(size_t)i + 1], 2) - std::pow(vx[(size_t)i], 2)) / (2 * p[4]);
                                if (std::fabs(vx[(size_t)i]) <= 1e-8 && std::fabs(ax) <= 1e-8) ax = -5.0;
                                r[6] = ax;
                            }
                        }
                    }
                    BpTraj bt; bt.id = T.id; bt.bp = std::move(W.bp);
                    bt.traj_id = S.traj_base_id + (T.id >= 0 && T.id <= 3 ? T.id : 9);        // ACTION_ID_MAP (:14-17,696-697)
                    if (S.last_bp.empty()) gv_first[(size_t)s] = W.gv;
                    S.last_bp.push_back(std::move(bt));
                    kept.push_back(std::move(T));
                }
                S.last = std::move(kept);
                if (req[s].incl_emerg) {                                                    // :1028-1034, calc_brake_emergency.py:9-45
                    if (S.last_bp.empty()) return fail(LTPL_ERR_INVALID_ARG, "planner: emergency profile without any trajectory (the reference raises IndexError, OTH.py:1029)");
                    S.em_base_id = S.last_bp[0].id;
                    const BpTraj& base = S.last_bp[0];
                    const int m = base.rows();
                    ltpl_vel_job jb; std::memset(&jb, 0, sizeof(jb)); JobBuf Bf;
                    jb.mode = LTPL_VEL_BRAKE; jb.n = m; jb.n_el = m - 1; jb.v_start = m > 0 ? base.bp[5] : 0.0;
                    const std::vector<double>& gv = gv_first[(size_t)s];
                    for (int i = 0; i < m; ++i) {
                        Bf.kappa.push_back(base.bp[(size_t)i * 7 + 4]);
                        Bf.gg.push_back((size_t)i * 2 + 1 < gv.size() ? gv[(size_t)i * 2] : req[s].gg_ax);
                        Bf.gg.push_back((size_t)i * 2 + 1 < gv.size() ? gv[(size_t)i * 2 + 1] : req[s].gg_ay);
                    }
                    for (int i = 0; i + 1 < m; ++i) Bf.el.push_back(base.bp[(size_t)(i + 1) * 7] - base.bp[(size_t)i * 7]);
                    if (Bf.el.empty()) Bf.el.push_back(0.0);
                    jobs3.push_back(jb); bufs3.push_back(std::move(Bf));
                    em_of.push_back({s, (int)jobs3.size() - 1});
                }
            }
        }
        if (!jobs3.empty()) {
            ltpl_vel_params ve = vp;                    // calc_brake_emergency.py:4-6,31-36: own vehicle constants, exponent 1
            ve.dyn_model_exp = 1.0; ve.drag_coeff = 0.854; ve.m_veh = 1160.0;
            if ((rc = run_jobs(ve, jobs3, bufs3, res3))) return rc;
            for (auto& e : em_of) {
                Scn& S = sc[(size_t)e.first];
                const BpTraj& base = S.last_bp[0];
                const int m = base.rows();
                const std::vector<double>& v = bufs3[(size_t)e.second].out;
                BpTraj em; em.id = LTPL_ACT_EMERGENCY; em.traj_id = base.traj_id;
                em.bp.assign((size_t)m * 7, 0.0);
                for (int i = 0; i < m; ++i) {
                    double* r = &em.bp[(size_t)i * 7];
                    for (int c = 0; c < 5; ++c) r[c] = base.bp[(size_t)i * 7 + c];
                    r[5] = v[(size_t)i];
                    if (i + 1 < m) r[6] = (std::pow(v[(size_t)i + 1], 2) - std::pow(v[(size_t)i], 2)) / (2 * (base.bp[(size_t)(i + 1) * 7] - base.bp[(size_t)i * 7]));
                }
                const int base_traj_id = base.traj_id;          // (`base` refers into last_bp: the push_back below may reallocate it)
                S.last_bp.push_back(std::move(em));
                S.path_ids.push_back({LTPL_ACT_EMERGENCY, base_traj_id});
            }
        }
        return LTPL_OK;
    }
};

}  // namespace ltplp
