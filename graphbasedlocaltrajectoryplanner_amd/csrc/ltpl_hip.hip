// ltpl_hip.hip -- MI355X (gfx950 / CDNA4) backend of the graph_ltpl online hot path. C ABI: include/ltpl_hip.h.
//
// One workgroup (256 threads = 4 wave64) plans one scenario (= one call of the reference's main_online_path_gen):
//   phase 1  obstacle -> closest reference-line layer (wave-level lexicographic min-reduction)        [M1]
//   phase 2  obstacle x edge-sample collision mask, lanes over the coalesced sample arrays of a layer
//            transition, result = bit per horizon edge in LDS                                          [M2]
//   phase 3  closest object / node, action-template choice                                            [M3, T1]
//   phase 4  four layered min-plus sweeps in parallel, one wave per filter (planning_range, default,
//            overtake_left, overtake_right); lane = destination node, private min over its in-edges
//            (CSC, no atomics), frontier distances double-buffered in LDS, parents in LDS              [F1, S1]
//   phase 5  horizon back-off / reduced-horizon logic on the per-layer goal table                      [S2]
//   phase 6  one wave per offered primitive: backtrack, gather, tridiagonal C2 spline solve,
//            re-sampling, heading / curvature                                                          [G1, P1-P3]
// The lattice (< 15 MB for Monteblanco) is uploaded once and is L2 / Infinity-Cache resident afterwards.
// Everything is IEEE fp64 and compiled with -ffp-contract=off so that masks, arg-mins and path costs are
// bit-identical to the NumPy arithmetic of the reference.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ltpl_hip.h"

#define WG_THREADS 256
#define NUM_WAVES 4
#define MAX_POS 192      // obstacle positions (own + predicted) per scenario
#define MAX_VEH 96       // vehicles per scenario
#define NFILT 4
#define F_PR 0
#define F_DEF 1
#define F_LEFT 2
#define F_RIGHT 3
#define D_PI 3.14159265358979323846

// ---------------------------------------------------------------------------------------------------------------------
// device-side views
// ---------------------------------------------------------------------------------------------------------------------
struct DevLat {
    int L, V, E, S, G;
    int mode;
    double min_plan_horizon, veh_width, sampled_resolution, lat_offset, vel_decrease_lat, veh_length;
    const int* layer_off; const int* rl_idx;
    const double* s_rl; const double* ref_x; const double* ref_y; const double* vel_rl;
    const double* node_x; const double* node_y; const double* vgoal;
    const int* in_ptr; const int* edge_src; const double* edge_cost; const double* edge_len; const int* samp_ptr;
    const double* sx; const double* sy; const double* spsi; const double* slen;
    const double* glob_rl;
};

struct DevPathsIn {
    int n_scen, n_w_last;
    const double* w_last;
    const int* start_layer; const int* start_node; const int* flags; const int* last_action; const int* const_closest;
    const double* psi_s;
    const int* veh_off; const int* pos_off; const double* veh_radius; const double* pos_x; const double* pos_y;
    const int* zone_off; const int* zone_gid;
    const int* n_last; const int* last_layer; const int* last_node;
};

struct DevPathsOut {
    int cap_nodes, cap_pts;
    int* end_layer; int* closest_obj_index; int* closest_obj_node; int* n_actions;
    int* action_id; int* valid; int* reduced; int* goal_layer; int* n_nodes; int* n_pts; int* n_ties;
    int* nodes; int* node_idx; double* coeff; double* path_param;
};

// dynamic-LDS plan (byte offsets), computed once per lattice on the host
struct LdsPlan {
    int kpad;          // padded max nodes per layer
    int hmax;          // max layers in a planning range (H + 1)
    int off_blocked;   // u32[(ehmax + 31) / 32]
    int off_zone;      // u32[(nhmax + 31) / 32]
    int off_dist;      // double[NFILT][2][kpad]
    int off_par;       // uchar2 [NFILT][hmax][kpad]  (.x = source node, .y = in-edge rank | tie bit 0x80)
    int off_best;      // int[NFILT][hmax]           goal node of layer j (-1 none), bit 30 = goal tie
    int off_path;      // per wave: path scratch (see PATH_* below)
    int path_stride;   // bytes per wave
    int words_blocked, words_zone;
    int total;
};

// ---------------------------------------------------------------------------------------------------------------------
// wave helpers (wave64)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_min3(double& k1, double& k2, int& idx)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        double o1 = __shfl_xor(k1, m), o2 = __shfl_xor(k2, m);
        int oi = __shfl_xor(idx, m);
        bool take = (o1 < k1) || (o1 == k1 && ((o2 < k2) || (o2 == k2 && oi < idx)));
        if (take) { k1 = o1; k2 = o2; idx = oi; }
    }
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int& total)
{
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
}

__device__ __forceinline__ double normalize_psi_dev(double psi)
{
    double sgn = (psi > 0.0) ? 1.0 : ((psi < 0.0) ? -1.0 : 0.0);
    double out = sgn * fmod(fabs(psi), 2.0 * D_PI);
    if (out >= D_PI) out -= 2.0 * D_PI;
    else if (out < -D_PI) out += 2.0 * D_PI;
    return out;
}

// per-scenario scalars shared by the workgroup
struct TickShared {
    int start_layer, start_node, flags, end_layer, H;
    int e_base, EH, n_base, NH;
    int n_pos, n_veh, pos0, veh0;
    int closest_idx, cl, cn, have_cn;
    int n_act, filt[LTPL_MAX_ACTIONS], name[LTPL_MAX_ACTIONS];
    int need[NFILT];
    int start_ok[NFILT];
    int slot_valid[LTPL_MAX_ACTIONS], slot_j[LTPL_MAX_ACTIONS];
    int n_last; int last_layer[LTPL_MAX_LAST_NODES]; int last_node[LTPL_MAX_LAST_NODES];
};

__device__ __forceinline__ bool node_removed(const TickShared& ts, const unsigned* zone_bits, const DevLat& lat,
                                             int f, int layer, int n, int gid)
{
    int nl = gid - ts.n_base; if (nl < 0) nl += lat.V;
    if (zone_bits[nl >> 5] & (1u << (nl & 31))) return true;
    if (f == F_LEFT && layer == ts.cl && n >= ts.cn) return true;     // main_online_path_gen.py:148-152
    if (f == F_RIGHT && layer == ts.cl && n < ts.cn) return true;     // main_online_path_gen.py:155-159
    return false;
}

// per-wave path scratch layout (doubles first for alignment)
#define PATH_NARR 9   // kx ky el mx my cpx cpy (7 double arrays of hmax) + pedge, pidx (2 int arrays of hmax + 1)

// ---------------------------------------------------------------------------------------------------------------------
// the path kernel (seam 1)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG_THREADS) void k_plan_paths(DevLat lat, DevPathsIn in, DevPathsOut out, LdsPlan lp)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ TickShared ts;
    __shared__ int sh_pos_layer[MAX_POS];

    const int s = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int L = lat.L;

    unsigned* blocked_bits = reinterpret_cast<unsigned*>(smem + lp.off_blocked);
    unsigned* zone_bits = reinterpret_cast<unsigned*>(smem + lp.off_zone);
    double* dist = reinterpret_cast<double*>(smem + lp.off_dist);
    uchar2* par = reinterpret_cast<uchar2*>(smem + lp.off_par);
    int* best = reinterpret_cast<int*>(smem + lp.off_best);

    // ---- phase 0: scenario scalars, planning range (gen_local_node_template.py:101-147) ----------------------------
    if (wave == 0) {
        const int sl = in.start_layer[s];
        int cnt = 0;
        if (lat.mode == 0) {
            double des = lat.s_rl[sl] + lat.min_plan_horizon;
            if (des > lat.s_rl[L - 1]) des -= lat.s_rl[L - 1];
            // bisect_left on a sorted array = number of entries < des
            for (int l0 = 0; l0 < L; l0 += 64) {
                int l = l0 + lane;
                bool lt = (l < L) && (lat.s_rl[l] < des);
                cnt += __popcll(__ballot(lt));
            }
        } else {
            cnt = (sl + (int)lat.min_plan_horizon) % L;
        }
        if (lane == 0) {
            ts.start_layer = sl; ts.start_node = in.start_node[s]; ts.flags = in.flags[s];
            int el = cnt >= L ? L - 1 : cnt;      // host rejects lattices where this could clamp
            ts.end_layer = el;
            int H = el - sl; if (H < 0) H = L - sl + el;
            ts.H = H;
            int first = sl + 1; if (first >= L) first -= L;
            ts.e_base = lat.in_ptr[lat.layer_off[first]];
            int e_end = lat.in_ptr[lat.layer_off[el + 1]];
            int EH = e_end - ts.e_base; if (EH < 0) EH += lat.E;
            ts.EH = EH;
            ts.n_base = lat.layer_off[sl];
            int NH = lat.layer_off[el + 1] - ts.n_base; if (NH <= 0) NH += lat.V;
            ts.NH = NH;
            ts.veh0 = in.veh_off[s]; ts.n_veh = in.veh_off[s + 1] - ts.veh0;
            ts.pos0 = in.pos_off[ts.veh0]; ts.n_pos = in.pos_off[ts.veh0 + ts.n_veh] - ts.pos0;
            int nl = in.n_last[s]; ts.n_last = nl;
            for (int i = 0; i < LTPL_MAX_LAST_NODES; ++i) {
                ts.last_layer[i] = in.last_layer[s * LTPL_MAX_LAST_NODES + i];
                ts.last_node[i] = in.last_node[s * LTPL_MAX_LAST_NODES + i];
            }
        }
    }
    for (int i = tid; i < lp.words_blocked; i += WG_THREADS) blocked_bits[i] = 0u;
    for (int i = tid; i < lp.words_zone; i += WG_THREADS) zone_bits[i] = 0u;
    __syncthreads();

    // zone-removed nodes of the "overtaking_zones" filter (gen_local_node_template.py:96; GraphBase.py:713-745)
    for (int i = in.zone_off[s] + tid; i < in.zone_off[s + 1]; i += WG_THREADS) {
        int nl = in.zone_gid[i] - ts.n_base; if (nl < 0) nl += lat.V;
        if (nl < ts.NH) atomicOr(&zone_bits[nl >> 5], 1u << (nl & 31));
    }

    // ---- phase 1: closest reference-line layer per obstacle position (get_intersec_edges.py:40-51) -----------------
    for (int p = wave; p < ts.n_pos; p += NUM_WAVES) {
        const double px = in.pos_x[ts.pos0 + p], py = in.pos_y[ts.pos0 + p];
        double bd = INFINITY, dummy = 0.0; int bl = 0x7fffffff;
        for (int l = lane; l < L; l += 64) {
            double dx = lat.ref_x[l] - px, dy = lat.ref_y[l] - py;
            double d2 = dx * dx + dy * dy;
            if (d2 < bd) { bd = d2; bl = l; }
        }
        wave_min3(bd, dummy, bl);
        if (lane == 0) {
            const int ol = bl, sl = ts.start_layer, el = ts.end_layer;
            bool gate = (sl - 1 <= ol && ol <= el + 1) || (sl > el && (sl - 1 <= ol || ol <= el + 1));
            sh_pos_layer[p] = gate ? ol : -1;
        }
    }
    __syncthreads();

    // ---- phase 2: obstacle x edge-sample mask (GraphBase.get_intersec_edges_in_range, GraphBase.py:567-646) --------
    // Work item = (position, window transition); each wave takes items round-robin, lanes stride the contiguous
    // sample arrays of that transition. Window = layers [ol-1, ol+1] with the reference's wrap quirks (:597-600):
    // the transition ol -> ol+1 is only part of it for ol <= L-2.
    for (int item = wave; item < 2 * ts.n_pos; item += NUM_WAVES) {
        const int p = item >> 1, second = item & 1;
        const int ol = sh_pos_layer[p];
        if (ol < 0) continue;
        if (second && ol > L - 2) continue;
        int b = ol + second; if (b >= L) b -= L;               // destination layer of the transition
        int jb = b - ts.start_layer; if (jb < 0) jb += L;
        if (jb < 1 || jb > ts.H) continue;                    // both end points must lie in the planning range
        // vehicle of this position -> radius
        int vlo = 0, vhi = ts.n_veh;                           // last vehicle with pos_off <= pos0 + p
        while (vhi - vlo > 1) { int mid = (vlo + vhi) >> 1; if (in.pos_off[ts.veh0 + mid] <= ts.pos0 + p) vlo = mid; else vhi = mid; }
        const double rr = in.veh_radius[ts.veh0 + vlo] + lat.veh_width / 2;
        double ref = rr * rr;
        ref += (lat.sampled_resolution * lat.sampled_resolution) / 4;
        const double px = in.pos_x[ts.pos0 + p], py = in.pos_y[ts.pos0 + p];
        const int e0 = lat.in_ptr[lat.layer_off[b]], e1 = lat.in_ptr[lat.layer_off[b + 1]];
        const int s0 = lat.samp_ptr[e0], s1 = lat.samp_ptr[e1];
        for (int k = s0 + lane; k < s1; k += 64) {
            double x = lat.sx[k] - px, y = lat.sy[k] - py;
            if (x * x + y * y <= ref) {
                int lo = e0, hi = e1;                          // edge owning sample k
                while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (lat.samp_ptr[mid] <= k) lo = mid; else hi = mid; }
                int el_ = lo - ts.e_base; if (el_ < 0) el_ += lat.E;
                atomicOr(&blocked_bits[el_ >> 5], 1u << (el_ & 31));
            }
        }
    }

    // ---- phase 3: closest object (gen_local_node_template.py:191-213) and action template (mopg.py:124-174) --------
    if (tid == 0) {
        int ci = -1, cd = -1, cl = -1;
        for (int k = 0; k < ts.n_veh; ++k) {
            int plast = in.pos_off[ts.veh0 + k + 1] - 1 - ts.pos0;
            int ol = (plast >= 0 && plast < ts.n_pos && in.pos_off[ts.veh0 + k + 1] > in.pos_off[ts.veh0 + k])
                         ? sh_pos_layer[plast] : -1;
            if (ol >= 0) {
                int ld = ol - ts.start_layer; if (ld < 0) ld = L - ts.start_layer + ol;
                if (ld <= ts.H && (cd < 0 || ld < cd)) { cd = ld; ci = k; cl = ol; }
            }
        }
        ts.closest_idx = ci; ts.cl = cl; ts.have_cn = cd >= 0; ts.cn = -1;
    }
    __syncthreads();
    if (wave == 0 && ts.have_cn) {
        const int p = in.pos_off[ts.veh0 + ts.closest_idx];
        const double px = in.pos_x[p], py = in.pos_y[p];
        const int v0 = lat.layer_off[ts.cl], K = lat.layer_off[ts.cl + 1] - v0;
        double bd = INFINITY, dummy = 0.0; int bn = 0x7fffffff;
        for (int n = lane; n < K; n += 64) {
            double dx = lat.node_x[v0 + n] - px, dy = lat.node_y[v0 + n] - py;
            double d2 = dx * dx + dy * dy;
            if (d2 < bd) { bd = d2; bn = n; }
        }
        wave_min3(bd, dummy, bn);
        if (lane == 0) ts.cn = bn;
    }
    __syncthreads();
    if (tid == 0) {
        const int flags = ts.flags;
        const bool action_sets = flags & LTPL_FLAG_ACTION_SETS, in_const = flags & LTPL_FLAG_OBJ_IN_CONST,
                   besides = flags & LTPL_FLAG_OBJ_BESIDES;
        int closest_idx = ts.closest_idx;
        if (in.const_closest[s] >= 0) closest_idx = in.const_closest[s];
        int n_act = 0;
        for (int f = 0; f < NFILT; ++f) ts.need[f] = 0;
        if (action_sets && (in_const || besides)) {
            ts.filt[n_act] = F_PR; ts.name[n_act++] = LTPL_ACT_FOLLOW;
            const int la = in.last_action[s];
            if (!in_const && (la == LTPL_ACT_LEFT || la == LTPL_ACT_RIGHT)) { ts.filt[n_act] = F_DEF; ts.name[n_act++] = la; }
            else if (!in_const) {
                ts.filt[n_act] = F_DEF; ts.name[n_act++] = LTPL_ACT_LEFT;
                ts.filt[n_act] = F_DEF; ts.name[n_act++] = LTPL_ACT_RIGHT;
            }
        } else if (action_sets && closest_idx >= 0 && ts.have_cn) {
            ts.filt[0] = F_PR; ts.name[0] = LTPL_ACT_FOLLOW;
            ts.filt[1] = F_LEFT; ts.name[1] = LTPL_ACT_LEFT;
            ts.filt[2] = F_RIGHT; ts.name[2] = LTPL_ACT_RIGHT;
            n_act = 3;
        } else {
            ts.filt[0] = F_DEF; ts.name[0] = LTPL_ACT_STRAIGHT; n_act = 1;
        }
        ts.n_act = n_act;
        for (int a = 0; a < n_act; ++a) ts.need[ts.filt[a]] = 1;
        out.end_layer[s] = ts.end_layer;
        out.closest_obj_index[s] = closest_idx;
        out.closest_obj_node[2 * s] = ts.have_cn ? ts.cl : -1;
        out.closest_obj_node[2 * s + 1] = ts.have_cn ? ts.cn : -1;
        out.n_actions[s] = n_act;
    }
    __syncthreads();

    // ---- phase 4: layered min-plus sweeps, wave = filter (GraphBase.search_graph_layer, GraphBase.py:854-894) ------
    // dist[v in layer j] = min over in-edges (u, v) of dist[u] + cost(u, v); strict '<' updates; among exact ties the
    // predecessor with the smaller dist[u], then the smaller node id wins (= order in which Dijkstra settles them).
    {
        const int f = wave;
        const bool active = ts.need[f] != 0;
        const int H = ts.H, kpad = lp.kpad;
        const int n_fac = min(ts.n_last - 1, in.n_w_last);
        double* d0 = dist + (size_t)(f * 2) * kpad;
        if (active) {
            const int sl = ts.start_layer, sn = ts.start_node;
            const int K0 = lat.layer_off[sl + 1] - lat.layer_off[sl];
            bool ok = sn >= 0 && sn < K0 && !node_removed(ts, zone_bits, lat, f, sl, sn, lat.layer_off[sl] + sn);
            for (int n = lane; n < kpad; n += 64) d0[n] = (ok && n == sn) ? 0.0 : INFINITY;
            if (lane == 0) { ts.start_ok[f] = ok; best[f * lp.hmax] = -1; }
        }
        __syncthreads();
        for (int j = 1; j <= H; ++j) {
            if (active) {
                int b = ts.start_layer + j; if (b >= L) b -= L;
                const int v0 = lat.layer_off[b], Kb = lat.layer_off[b + 1] - v0;
                const double* dprev = dist + (size_t)(f * 2 + ((j - 1) & 1)) * kpad;
                double* dcur = dist + (size_t)(f * 2 + (j & 1)) * kpad;
                uchar2* pj = par + ((size_t)f * lp.hmax + j) * kpad;
                // cost discount along the previous solution (gen_local_node_template.py:154-162): edge j-1 -> j
                int fac_src = -1, fac_dst = -1; double fac = 1.0;
                if (j - 1 < n_fac) {
                    int pl = ts.last_layer[j - 1], nl_ = ts.last_layer[j];
                    int pb = b - 1; if (pb < 0) pb += L;
                    if (pl == pb && nl_ == b) { fac_src = ts.last_node[j - 1]; fac_dst = ts.last_node[j]; fac = in.w_last[j - 1]; }
                }
                double g1 = INFINITY, g2 = INFINITY; int gn = 0x7fffffff;
                for (int n = lane; n < Kb; n += 64) {
                    const int v = v0 + n;
                    double bestc = INFINITY, bestdu = INFINITY; int bk = 0, bsrc = 0, tie = 0;
                    if (!node_removed(ts, zone_bits, lat, f, b, n, v)) {
                        const int e0 = lat.in_ptr[v], e1 = lat.in_ptr[v + 1];
                        for (int e = e0; e < e1; ++e) {
                            const int src = lat.edge_src[e];
                            const double du = dprev[src];
                            double c = lat.edge_cost[e];
                            if (f != F_PR) {
                                int el_ = e - ts.e_base; if (el_ < 0) el_ += lat.E;
                                if (blocked_bits[el_ >> 5] & (1u << (el_ & 31))) continue;
                            }
                            if (!(du < INFINITY)) continue;
                            if (src == fac_src && n == fac_dst) c *= fac;
                            const double cand = du + c;
                            if (cand < bestc) { bestc = cand; bestdu = du; bk = e - e0; bsrc = src; tie = 0; }
                            else if (cand == bestc) {
                                tie = 1;
                                if (du < bestdu) { bestdu = du; bk = e - e0; bsrc = src; }
                            }
                        }
                    }
                    dcur[n] = bestc;
                    pj[n] = make_uchar2((unsigned char)bsrc, (unsigned char)(bk | (tie ? 0x80 : 0)));
                    if (bestc < INFINITY) {
                        // virtual goal edge (GraphBase.py:188-194)
                        double tot = bestc + lat.vgoal[v];
                        if (tot < g1 || (tot == g1 && (bestc < g2 || (bestc == g2 && n < gn)))) { g1 = tot; g2 = bestc; gn = n; }
                    }
                }
                for (int n = Kb + lane; n < kpad; n += 64) dcur[n] = INFINITY;
                double m1 = g1, m2 = g2; int mn = gn;
                wave_min3(m1, m2, mn);
                int ntie = __popcll(__ballot(g1 == m1 && g1 < INFINITY));   // NOTE: per-lane best only (Kb <= 64 exact)
                if (lane == 0) best[f * lp.hmax + j] = (m1 < INFINITY) ? (mn | (ntie > 1 ? (1 << 30) : 0)) : -1;
            }
            __syncthreads();
        }
    }

    // ---- phase 5: search loop with horizon back-off (main_online_path_gen.py:187-248) -------------------------------
    if (tid == 0) {
        const bool in_const = ts.flags & LTPL_FLAG_OBJ_IN_CONST;
        int mod_j = ts.H;
        for (int a = 0; a < LTPL_MAX_ACTIONS; ++a) {
            const int slot = s * LTPL_MAX_ACTIONS + a;
            ts.slot_valid[a] = 0; ts.slot_j[a] = 0;
            if (a >= ts.n_act) {
                out.action_id[slot] = LTPL_ACT_NONE; out.valid[slot] = 0; out.reduced[slot] = 0; out.goal_layer[slot] = -1;
                out.n_nodes[slot] = 0; out.n_pts[slot] = 0; out.n_ties[slot] = 0;
                continue;
            }
            const int f = ts.filt[a]; int nm = ts.name[a];
            bool found = false;
            for (;;) {
                if (mod_j == 0) break;
                found = ts.start_ok[f] && best[f * lp.hmax + mod_j] >= 0;
                if (found || !(nm == LTPL_ACT_FOLLOW || nm == LTPL_ACT_STRAIGHT)) break;
                mod_j -= 1;
            }
            const bool reduced = mod_j != ts.H;
            int goal = ts.start_layer + mod_j; if (goal >= L) goal -= L;
            if (reduced) {
                const int cl = ts.cl, sl = ts.start_layer;
                bool in_mod = ts.have_cn && ((sl <= cl && cl <= goal) || (sl > goal && (cl >= sl || cl <= goal)));
                if (!in_const && ts.have_cn && !in_mod) {
                    if (nm == LTPL_ACT_FOLLOW || nm == LTPL_ACT_STRAIGHT) nm = LTPL_ACT_STRAIGHT;
                    else found = false;
                }
            }
            out.action_id[slot] = nm; out.reduced[slot] = reduced ? 1 : 0; out.goal_layer[slot] = goal;
            out.valid[slot] = found ? 1 : 0;
            if (!found) { out.n_nodes[slot] = 0; out.n_pts[slot] = 0; out.n_ties[slot] = 0; }
            ts.slot_valid[a] = found ? 1 : 0; ts.slot_j[a] = mod_j;
        }
    }
    __syncthreads();

    // ---- phase 6: wave a assembles primitive a (main_online_path_gen.py:250-328) ------------------------------------
    if (wave < LTPL_MAX_ACTIONS && wave < ts.n_act && ts.slot_valid[wave]) {
        const int a = wave, slot = s * LTPL_MAX_ACTIONS + a, f = ts.filt[a], J = ts.slot_j[a], N = J;   // N segments
        unsigned char* pw = smem + lp.off_path + (size_t)a * lp.path_stride;
        const int hm = lp.hmax;
        double* kx = reinterpret_cast<double*>(pw);
        double* ky = kx + hm; double* el = ky + hm; double* mx = el + hm; double* my = mx + hm;
        double* cpx = my + hm; double* cpy = cpx + hm;
        int* pedge = reinterpret_cast<int*>(cpy + hm); int* pidx = pedge + hm + 1;
        int* o_nodes = out.nodes + (size_t)slot * out.cap_nodes;
        int* o_idx = out.node_idx + (size_t)slot * out.cap_nodes;
        double* o_coeff = out.coeff + (size_t)slot * out.cap_nodes * 8;
        double* o_pp = out.path_param + (size_t)slot * out.cap_pts * 5;

        // backtrack along the LDS parent table (lane 0), count exact ties on the way
        if (lane == 0) {
            int bj = best[f * hm + J];
            int ties = (bj >> 30) & 1;
            int n = bj & 0xffff;
            for (int j = J; j >= 1; --j) {
                int b = ts.start_layer + j; if (b >= L) b -= L;
                o_nodes[j] = n;
                uchar2 pr = par[((size_t)f * hm + j) * lp.kpad + n];
                pedge[j - 1] = lat.in_ptr[lat.layer_off[b] + n] + (pr.y & 0x7f);
                ties += (pr.y >> 7) & 1;
                n = pr.x;
            }
            o_nodes[0] = n;
            out.n_nodes[slot] = J + 1;
            out.n_ties[slot] = ties;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        // gather: rows per edge, node row indices, knots, element lengths (:260-297)
        int run = 0;
        for (int i0 = 0; i0 < N; i0 += 64) {
            const int i = i0 + lane;
            int take = 0, e = 0, k0 = 0, k1 = 0;
            if (i < N) {
                e = pedge[i]; k0 = lat.samp_ptr[e]; k1 = lat.samp_ptr[e + 1];
                take = (i == N - 1) ? (k1 - k0) : (k1 - k0 - 1);
            }
            int tot; int off = wave_excl_scan(take, lane, tot);
            if (i < N) {
                pidx[i] = run + off;
                kx[i] = lat.sx[k0]; ky[i] = lat.sy[k0]; el[i] = lat.edge_len[e];
                if (i == N - 1) { kx[N] = lat.sx[k1 - 1]; ky[N] = lat.sy[k1 - 1]; pidx[N] = run + off + take - 1; }
            }
            run += tot;
        }
        const int n_pts = run;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int i = lane; i <= N; i += 64) o_idx[i] = pidx[i];
        if (lane == 0) out.n_pts[slot] = n_pts;

        // tph.calc_splines (main_online_path_gen.py:299-309) as the equivalent clamped C2 spline in the cumulated
        // el_lengths parameter: tridiagonal system in the knot slopes m_i, Thomas algorithm; lane 0 -> x, lane 1 -> y
        if (lane < 2) {
            const int e_first = pedge[0], e_last = pedge[N - 1];
            const double psi_s = (ts.flags & LTPL_FLAG_HAS_PSI_S) ? in.psi_s[s] : lat.spsi[lat.samp_ptr[e_first]];
            const double psi_e = lat.spsi[lat.samp_ptr[e_last + 1] - 1];
            const double* kk = lane == 0 ? kx : ky;
            double* m = lane == 0 ? mx : my;
            double* cp = lane == 0 ? cpx : cpy;
            const double m0 = lane == 0 ? cos(psi_s + D_PI / 2) : sin(psi_s + D_PI / 2);
            const double mN = lane == 0 ? cos(psi_e + D_PI / 2) : sin(psi_e + D_PI / 2);
            m[0] = m0; m[N] = mN;
            if (N >= 2) {
                double cprev = 0.0, dprev_ = 0.0;
                for (int i = 1; i <= N - 1; ++i) {
                    const double h0 = el[i - 1], h1 = el[i];
                    const double ai = 1.0 / h0, ci = 1.0 / h1, bi = 2.0 * (ai + ci);
                    double di = 3.0 * ((kk[i] - kk[i - 1]) / (h0 * h0) + (kk[i + 1] - kk[i]) / (h1 * h1));
                    if (i == 1) di -= ai * m0;
                    if (i == N - 1) di -= ci * mN;
                    const double cc = (i == N - 1) ? 0.0 : ci;
                    const double denom = (i == 1) ? bi : (bi - ai * cprev);
                    const double cpi = cc / denom;
                    const double dpi = (i == 1) ? di / denom : (di - ai * dprev_) / denom;
                    cp[i] = cpi; m[i] = dpi; cprev = cpi; dprev_ = dpi;
                }
                for (int i = N - 2; i >= 1; --i) m[i] = m[i] - cp[i] * m[i + 1];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        // coefficients per segment, t in [0, 1]: a0 = k_i, a1 = m_i h, a2 = 3 d - 2 T0 - T1, a3 = -2 d + T0 + T1
        for (int i = lane; i < N; i += 64) {
            const double h = el[i];
            {
                const double T0 = mx[i] * h, T1 = mx[i + 1] * h, dlt = kx[i + 1] - kx[i];
                o_coeff[i * 8 + 0] = kx[i]; o_coeff[i * 8 + 1] = T0;
                o_coeff[i * 8 + 2] = 3.0 * dlt - 2.0 * T0 - T1; o_coeff[i * 8 + 3] = -2.0 * dlt + T0 + T1;
            }
            {
                const double T0 = my[i] * h, T1 = my[i + 1] * h, dlt = ky[i + 1] - ky[i];
                o_coeff[i * 8 + 4] = ky[i]; o_coeff[i * 8 + 5] = T0;
                o_coeff[i * 8 + 6] = 3.0 * dlt - 2.0 * T0 - T1; o_coeff[i * 8 + 7] = -2.0 * dlt + T0 + T1;
            }
        }

        // tph.interp_splines(stepnum_fixed) + tph.calc_head_curv_an (:311-322); column 4 keeps the offline spacing
        for (int r = lane; r < n_pts; r += 64) {
            int lo = 0, hi = N;                                // segment i with pidx[i] <= r < pidx[i+1] (last: <=)
            while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (pidx[mid] <= r) lo = mid; else hi = mid; }
            const int i = lo, k = r - pidx[i];
            const int n_i = pidx[i + 1] - pidx[i] + 1;
            const double t = (k == n_i - 1) ? 1.0 : (double)k * (1.0 / (double)(n_i - 1));
            const double h = el[i];
            const double Tx0 = mx[i] * h, Tx1 = mx[i + 1] * h, dx_ = kx[i + 1] - kx[i];
            const double Ty0 = my[i] * h, Ty1 = my[i + 1] * h, dy_ = ky[i + 1] - ky[i];
            const double ax0 = kx[i], ax1 = Tx0, ax2 = 3.0 * dx_ - 2.0 * Tx0 - Tx1, ax3 = -2.0 * dx_ + Tx0 + Tx1;
            const double ay0 = ky[i], ay1 = Ty0, ay2 = 3.0 * dy_ - 2.0 * Ty0 - Ty1, ay3 = -2.0 * dy_ + Ty0 + Ty1;
            const double t2 = t * t, t3 = t2 * t;
            double x = ((ax0 + ax1 * t) + ax2 * t2) + ax3 * t3;
            double y = ((ay0 + ay1 * t) + ay2 * t2) + ay3 * t3;
            if (r == n_pts - 1) { x = ((ax0 + ax1) + ax2) + ax3; y = ((ay0 + ay1) + ay2) + ay3; }
            const double xd = ax1 + 2.0 * ax2 * t + 3.0 * ax3 * t2, yd = ay1 + 2.0 * ay2 * t + 3.0 * ay3 * t2;
            const double xdd = 2.0 * ax2 + 6.0 * ax3 * t, ydd = 2.0 * ay2 + 6.0 * ay3 * t;
            const double q = xd * xd + yd * yd;
            double* row = o_pp + (size_t)r * 5;
            row[0] = x; row[1] = y;
            row[2] = normalize_psi_dev(atan2(yd, xd) - D_PI / 2);
            row[3] = (xd * ydd - yd * xdd) / (q * sqrt(q));
            row[4] = lat.slen[lat.samp_ptr[pedge[i]] + k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static thread_local std::string g_create_error;

struct ltpl_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    DevLat lat{};
    LdsPlan lp{};
    ltpl_caps caps{};
    std::vector<void*> dev_allocs;
    // staging
    void* h_in = nullptr; size_t h_in_cap = 0;
    void* d_in = nullptr; size_t d_in_cap = 0;
    void* h_out = nullptr; size_t h_out_cap = 0;
    void* d_out = nullptr; size_t d_out_cap = 0;
};

#define HIP_TRY(h, call)                                                                                              \
    do {                                                                                                              \
        hipError_t e_ = (call);                                                                                       \
        if (e_ != hipSuccess) {                                                                                       \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                             \
            return LTPL_ERR_HIP;                                                                                      \
        }                                                                                                             \
    } while (0)

template <typename T>
static int upload(ltpl_handle* h, const T* src, size_t n, const T** dst)
{
    void* p = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    HIP_TRY(h, hipMalloc(&p, bytes));
    h->dev_allocs.push_back(p);
    if (n) HIP_TRY(h, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return LTPL_OK;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// planning-range statistics over all start layers (sizes the LDS plan and the output capacities)
static int horizon_stats(const ltpl_lattice_desc* d, int* hmax, int* ehmax, int* nhmax, int* ptsmax, int* kmax,
                         int* degmax, std::string* why)
{
    const int L = d->num_layers;
    std::vector<int> edges_into(L, 0), maxsamp_into(L, 0);
    *kmax = 0; *degmax = 0;
    for (int l = 0; l < L; ++l) {
        int K = d->layer_node_off[l + 1] - d->layer_node_off[l];
        if (K > *kmax) *kmax = K;
        for (int v = d->layer_node_off[l]; v < d->layer_node_off[l + 1]; ++v) {
            int deg = d->in_ptr[v + 1] - d->in_ptr[v];
            if (deg > *degmax) *degmax = deg;
            edges_into[l] += deg;
            for (int e = d->in_ptr[v]; e < d->in_ptr[v + 1]; ++e) {
                int ns = d->samp_ptr[e + 1] - d->samp_ptr[e];
                if (ns < 2) { *why = "edge with fewer than 2 samples"; return LTPL_ERR_INVALID_ARG; }
                if (ns > maxsamp_into[l]) maxsamp_into[l] = ns;
                int pl = (l + L - 1) % L;
                if (d->edge_src[e] < 0 || d->edge_src[e] >= d->layer_node_off[pl + 1] - d->layer_node_off[pl]) {
                    *why = "edge_src out of range"; return LTPL_ERR_INVALID_ARG;
                }
            }
        }
    }
    *hmax = *ehmax = *nhmax = *ptsmax = 0;
    for (int sl = 0; sl < L; ++sl) {
        int el;
        if (d->plan_horizon_mode == 0) {
            double des = d->s_raceline[sl] + d->min_plan_horizon;
            if (des > d->s_raceline[L - 1]) des -= d->s_raceline[L - 1];
            int lo = 0, hi = L;
            while (lo < hi) { int mid = (lo + hi) / 2; if (d->s_raceline[mid] < des) lo = mid + 1; else hi = mid; }
            el = lo;
        } else el = (sl + (int)d->min_plan_horizon) % L;
        if (el >= L) { *why = "planning horizon runs past the last layer (track shorter than the horizon?)"; return LTPL_ERR_UNSUPPORTED; }
        int H = el - sl; if (H < 0) H = L - sl + el;
        if (H <= 0 || H >= L - 1) { *why = "planning range covers the whole track; not supported"; return LTPL_ERR_UNSUPPORTED; }
        int eh = 0, nh = d->layer_node_off[sl + 1] - d->layer_node_off[sl], pts = 1;
        for (int j = 1; j <= H; ++j) {
            int b = (sl + j) % L;
            eh += edges_into[b]; nh += d->layer_node_off[b + 1] - d->layer_node_off[b];
            pts += maxsamp_into[b] - 1;
        }
        if (H + 1 > *hmax) *hmax = H + 1;
        if (eh > *ehmax) *ehmax = eh;
        if (nh > *nhmax) *nhmax = nh;
        if (pts > *ptsmax) *ptsmax = pts;
    }
    return LTPL_OK;
}

extern "C" int ltpl_version(void) { return LTPL_ABI_VERSION; }

extern "C" const char* ltpl_last_error(const ltpl_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int ltpl_destroy(ltpl_handle* h)
{
    if (!h) return LTPL_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    for (void* p : h->dev_allocs) (void)hipFree(p);
    if (h->d_in) (void)hipFree(h->d_in);
    if (h->d_out) (void)hipFree(h->d_out);
    if (h->h_in) (void)hipHostFree(h->h_in);
    if (h->h_out) (void)hipHostFree(h->h_out);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return LTPL_OK;
}

extern "C" int ltpl_create(const ltpl_lattice_desc* d, int device, ltpl_handle** out_handle)
{
    g_create_error.clear();
    if (!d || !out_handle) { g_create_error = "null argument"; return LTPL_ERR_INVALID_ARG; }
    if (!d->closed) { g_create_error = "only closed tracks are supported"; return LTPL_ERR_UNSUPPORTED; }
    if (d->num_layers < 4 || d->num_nodes < 1 || d->num_edges < 1) { g_create_error = "empty lattice"; return LTPL_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "no HIP device visible"; return LTPL_ERR_NO_DEVICE; }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev) { g_create_error = "device index out of range"; return LTPL_ERR_NO_DEVICE; }

    int hmax, ehmax, nhmax, ptsmax, kmax, degmax;
    int rc = horizon_stats(d, &hmax, &ehmax, &nhmax, &ptsmax, &kmax, &degmax, &g_create_error);
    if (rc) return rc;
    if (kmax > 255 || degmax > 127) { g_create_error = "more than 255 nodes per layer or 127 in-edges per node"; return LTPL_ERR_CAPACITY; }

    ltpl_handle* h = new ltpl_handle();
    h->device = device;
    auto fail = [&](int code) { g_create_error = h->err; ltpl_destroy(h); return code; };
    if (hipSetDevice(device) != hipSuccess) { h->err = "hipSetDevice failed"; return fail(LTPL_ERR_HIP); }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return fail(LTPL_ERR_HIP); }

    DevLat& L = h->lat;
    L.L = d->num_layers; L.V = d->num_nodes; L.E = d->num_edges; L.S = d->num_samples; L.G = d->num_glob_rl;
    L.mode = d->plan_horizon_mode; L.min_plan_horizon = d->min_plan_horizon; L.veh_width = d->veh_width;
    L.sampled_resolution = d->sampled_resolution; L.lat_offset = d->lat_offset;
    L.vel_decrease_lat = d->vel_decrease_lat; L.veh_length = d->veh_length;
#define UP(field, src, n) if ((rc = upload(h, src, (size_t)(n), &L.field)) != LTPL_OK) return fail(rc)
    UP(layer_off, d->layer_node_off, L.L + 1); UP(rl_idx, d->raceline_index, L.L);
    UP(s_rl, d->s_raceline, L.L); UP(ref_x, d->refline_x, L.L); UP(ref_y, d->refline_y, L.L);
    UP(vel_rl, d->vel_raceline, L.L);
    UP(node_x, d->node_x, L.V); UP(node_y, d->node_y, L.V); UP(vgoal, d->vgoal_cost, L.V);
    UP(in_ptr, d->in_ptr, L.V + 1); UP(edge_src, d->edge_src, L.E); UP(edge_cost, d->edge_cost, L.E);
    UP(edge_len, d->edge_len, L.E); UP(samp_ptr, d->samp_ptr, L.E + 1);
    UP(sx, d->samp_x, L.S); UP(sy, d->samp_y, L.S); UP(spsi, d->samp_psi, L.S); UP(slen, d->samp_len, L.S);
    UP(glob_rl, d->glob_rl, (size_t)L.G * 5);
#undef UP

    LdsPlan& lp = h->lp;
    lp.kpad = (int)align_up((size_t)kmax, 4);
    lp.hmax = hmax + 1;
    lp.words_blocked = (ehmax + 31) / 32 + 1;
    lp.words_zone = (nhmax + 31) / 32 + 1;
    size_t off = 0;
    lp.off_dist = (int)off; off += sizeof(double) * NFILT * 2 * lp.kpad;
    lp.path_stride = (int)align_up(sizeof(double) * 7 * lp.hmax + sizeof(int) * 2 * (lp.hmax + 1), 16);
    lp.off_path = (int)off; off += (size_t)lp.path_stride * LTPL_MAX_ACTIONS;
    lp.off_best = (int)off; off += sizeof(int) * NFILT * lp.hmax; off = align_up(off, 16);
    lp.off_blocked = (int)off; off += sizeof(unsigned) * lp.words_blocked; off = align_up(off, 16);
    lp.off_zone = (int)off; off += sizeof(unsigned) * lp.words_zone; off = align_up(off, 16);
    lp.off_par = (int)off; off += sizeof(uchar2) * NFILT * (size_t)lp.hmax * lp.kpad; off = align_up(off, 16);
    lp.total = (int)off;
    if (off > 150 * 1024) {
        h->err = "planning horizon too large for the LDS-resident sweep (" + std::to_string(off) + " B > 150 KiB)";
        return fail(LTPL_ERR_CAPACITY);
    }
    if (lp.total > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_plan_paths), hipFuncAttributeMaxDynamicSharedMemorySize,
                                lp.total) != hipSuccess) { h->err = "cannot raise dynamic LDS limit"; return fail(LTPL_ERR_HIP); }
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { h->err = "hipGetDeviceProperties failed"; return fail(LTPL_ERR_HIP); }
    h->caps.max_path_nodes = hmax; h->caps.max_path_pts = ptsmax; h->caps.max_horizon_edges = ehmax;
    h->caps.device = device; h->caps.num_cus = prop.multiProcessorCount; h->caps.lds_bytes_paths = lp.total;
    *out_handle = h;
    return LTPL_OK;
}

extern "C" int ltpl_get_caps(const ltpl_handle* h, ltpl_caps* caps)
{
    if (!h || !caps) return LTPL_ERR_INVALID_ARG;
    *caps = h->caps;
    return LTPL_OK;
}

// ---- staging arena ---------------------------------------------------------------------------------------------------
struct Arena {
    size_t size = 0;
    size_t add(size_t bytes) { size_t o = size; size = align_up(size + bytes, 16); return o; }
};

static int ensure(ltpl_handle* h, void** hp, size_t* hcap, void** dp, size_t* dcap, size_t need)
{
    if (need > *hcap) {
        if (*hp) (void)hipHostFree(*hp);
        *hp = nullptr; *hcap = 0;
        size_t cap = align_up(need + need / 2, 4096);
        HIP_TRY(h, hipHostMalloc(hp, cap, hipHostMallocDefault));
        *hcap = cap;
    }
    if (need > *dcap) {
        if (*dp) (void)hipFree(*dp);
        *dp = nullptr; *dcap = 0;
        size_t cap = align_up(need + need / 2, 4096);
        HIP_TRY(h, hipMalloc(dp, cap));
        *dcap = cap;
    }
    return LTPL_OK;
}

struct InLayout {
    size_t w_last, start_layer, start_node, flags, last_action, const_closest, psi_s, veh_off, pos_off, veh_radius,
        pos_x, pos_y, zone_off, zone_gid, n_last, last_layer, last_node, total;
    int n_veh, n_pos, n_zone;
};

static int validate_and_layout(ltpl_handle* h, const ltpl_paths_in* in, InLayout* lo)
{
    if (!in || in->n_scen < 1) { h->err = "n_scen < 1"; return LTPL_ERR_INVALID_ARG; }
    const int n = in->n_scen;
    if (in->n_w_last < 0 || in->n_w_last > LTPL_MAX_LAST_NODES - 1) { h->err = "n_w_last out of range"; return LTPL_ERR_INVALID_ARG; }
    if (in->veh_off[0] != 0 || in->zone_off[0] != 0) { h->err = "offset arrays must start at 0"; return LTPL_ERR_INVALID_ARG; }
    lo->n_veh = in->veh_off[n]; lo->n_zone = in->zone_off[n];
    if (lo->n_veh < 0 || lo->n_zone < 0) { h->err = "negative offsets"; return LTPL_ERR_INVALID_ARG; }
    if (in->pos_off[0] != 0) { h->err = "pos_off must start at 0"; return LTPL_ERR_INVALID_ARG; }
    lo->n_pos = in->pos_off[lo->n_veh];
    for (int s = 0; s < n; ++s) {
        const int sl = in->start_layer[s];
        if (sl < 0 || sl >= h->lat.L) { h->err = "start_layer out of range"; return LTPL_ERR_INVALID_ARG; }
        const int nv = in->veh_off[s + 1] - in->veh_off[s];
        if (nv < 0 || nv > MAX_VEH) { h->err = "more than 96 vehicles in one scenario"; return LTPL_ERR_CAPACITY; }
        const int np = in->pos_off[in->veh_off[s + 1]] - in->pos_off[in->veh_off[s]];
        if (np < 0 || np > MAX_POS) { h->err = "more than 192 obstacle positions in one scenario"; return LTPL_ERR_CAPACITY; }
        if (in->n_last[s] < 0 || in->n_last[s] > LTPL_MAX_LAST_NODES) { h->err = "n_last out of range"; return LTPL_ERR_INVALID_ARG; }
        for (int v = in->veh_off[s]; v < in->veh_off[s + 1]; ++v)
            if (in->pos_off[v + 1] - in->pos_off[v] < 1) { h->err = "vehicle without position"; return LTPL_ERR_INVALID_ARG; }
    }
    for (int i = 0; i < lo->n_zone; ++i)
        if (in->zone_gid[i] < 0 || in->zone_gid[i] >= h->lat.V) { h->err = "zone node id out of range"; return LTPL_ERR_INVALID_ARG; }
    Arena a;
    lo->w_last = a.add(sizeof(double) * (size_t)(in->n_w_last + 1));
    lo->start_layer = a.add(sizeof(int) * (size_t)n); lo->start_node = a.add(sizeof(int) * (size_t)n);
    lo->flags = a.add(sizeof(int) * (size_t)n); lo->last_action = a.add(sizeof(int) * (size_t)n);
    lo->const_closest = a.add(sizeof(int) * (size_t)n); lo->psi_s = a.add(sizeof(double) * (size_t)n);
    lo->veh_off = a.add(sizeof(int) * (size_t)(n + 1)); lo->pos_off = a.add(sizeof(int) * (size_t)(lo->n_veh + 1));
    lo->veh_radius = a.add(sizeof(double) * (size_t)(lo->n_veh + 1));
    lo->pos_x = a.add(sizeof(double) * (size_t)(lo->n_pos + 1)); lo->pos_y = a.add(sizeof(double) * (size_t)(lo->n_pos + 1));
    lo->zone_off = a.add(sizeof(int) * (size_t)(n + 1)); lo->zone_gid = a.add(sizeof(int) * (size_t)(lo->n_zone + 1));
    lo->n_last = a.add(sizeof(int) * (size_t)n);
    lo->last_layer = a.add(sizeof(int) * (size_t)n * LTPL_MAX_LAST_NODES);
    lo->last_node = a.add(sizeof(int) * (size_t)n * LTPL_MAX_LAST_NODES);
    lo->total = a.size;
    return LTPL_OK;
}

static void pack_in(const ltpl_paths_in* in, const InLayout& lo, unsigned char* hb, const unsigned char* db, DevPathsIn* di)
{
    const int n = in->n_scen;
#define CP(field, src, count, T) do { if ((count) > 0) memcpy(hb + lo.field, src, sizeof(T) * (size_t)(count)); } while (0)
    CP(w_last, in->w_last_edges, in->n_w_last, double);
    CP(start_layer, in->start_layer, n, int); CP(start_node, in->start_node, n, int); CP(flags, in->flags, n, int);
    CP(last_action, in->last_action, n, int); CP(const_closest, in->const_closest, n, int); CP(psi_s, in->psi_s, n, double);
    CP(veh_off, in->veh_off, n + 1, int); CP(pos_off, in->pos_off, lo.n_veh + 1, int);
    CP(veh_radius, in->veh_radius, lo.n_veh, double); CP(pos_x, in->pos_x, lo.n_pos, double); CP(pos_y, in->pos_y, lo.n_pos, double);
    CP(zone_off, in->zone_off, n + 1, int); CP(zone_gid, in->zone_gid, lo.n_zone, int);
    CP(n_last, in->n_last, n, int); CP(last_layer, in->last_layer, n * LTPL_MAX_LAST_NODES, int);
    CP(last_node, in->last_node, n * LTPL_MAX_LAST_NODES, int);
#undef CP
    di->n_scen = n; di->n_w_last = in->n_w_last;
    di->w_last = reinterpret_cast<const double*>(db + lo.w_last);
    di->start_layer = reinterpret_cast<const int*>(db + lo.start_layer);
    di->start_node = reinterpret_cast<const int*>(db + lo.start_node);
    di->flags = reinterpret_cast<const int*>(db + lo.flags);
    di->last_action = reinterpret_cast<const int*>(db + lo.last_action);
    di->const_closest = reinterpret_cast<const int*>(db + lo.const_closest);
    di->psi_s = reinterpret_cast<const double*>(db + lo.psi_s);
    di->veh_off = reinterpret_cast<const int*>(db + lo.veh_off);
    di->pos_off = reinterpret_cast<const int*>(db + lo.pos_off);
    di->veh_radius = reinterpret_cast<const double*>(db + lo.veh_radius);
    di->pos_x = reinterpret_cast<const double*>(db + lo.pos_x);
    di->pos_y = reinterpret_cast<const double*>(db + lo.pos_y);
    di->zone_off = reinterpret_cast<const int*>(db + lo.zone_off);
    di->zone_gid = reinterpret_cast<const int*>(db + lo.zone_gid);
    di->n_last = reinterpret_cast<const int*>(db + lo.n_last);
    di->last_layer = reinterpret_cast<const int*>(db + lo.last_layer);
    di->last_node = reinterpret_cast<const int*>(db + lo.last_node);
}

struct OutLayout {
    size_t end_layer, closest_obj_index, closest_obj_node, n_actions, action_id, valid, reduced, goal_layer, n_nodes,
        n_pts, n_ties, nodes, node_idx, coeff, path_param, total;
};

static void layout_out(int n, int cap_nodes, int cap_pts, OutLayout* lo)
{
    Arena a; const size_t A = LTPL_MAX_ACTIONS;
    lo->end_layer = a.add(sizeof(int) * (size_t)n); lo->closest_obj_index = a.add(sizeof(int) * (size_t)n);
    lo->closest_obj_node = a.add(sizeof(int) * (size_t)n * 2); lo->n_actions = a.add(sizeof(int) * (size_t)n);
    lo->action_id = a.add(sizeof(int) * n * A); lo->valid = a.add(sizeof(int) * n * A);
    lo->reduced = a.add(sizeof(int) * n * A); lo->goal_layer = a.add(sizeof(int) * n * A);
    lo->n_nodes = a.add(sizeof(int) * n * A); lo->n_pts = a.add(sizeof(int) * n * A); lo->n_ties = a.add(sizeof(int) * n * A);
    lo->nodes = a.add(sizeof(int) * n * A * (size_t)cap_nodes); lo->node_idx = a.add(sizeof(int) * n * A * (size_t)cap_nodes);
    lo->coeff = a.add(sizeof(double) * n * A * (size_t)cap_nodes * 8);
    lo->path_param = a.add(sizeof(double) * n * A * (size_t)cap_pts * 5);
    lo->total = a.size;
}

static void bind_out(unsigned char* db, const OutLayout& lo, int cap_nodes, int cap_pts, DevPathsOut* d)
{
    d->cap_nodes = cap_nodes; d->cap_pts = cap_pts;
    d->end_layer = reinterpret_cast<int*>(db + lo.end_layer);
    d->closest_obj_index = reinterpret_cast<int*>(db + lo.closest_obj_index);
    d->closest_obj_node = reinterpret_cast<int*>(db + lo.closest_obj_node);
    d->n_actions = reinterpret_cast<int*>(db + lo.n_actions);
    d->action_id = reinterpret_cast<int*>(db + lo.action_id); d->valid = reinterpret_cast<int*>(db + lo.valid);
    d->reduced = reinterpret_cast<int*>(db + lo.reduced); d->goal_layer = reinterpret_cast<int*>(db + lo.goal_layer);
    d->n_nodes = reinterpret_cast<int*>(db + lo.n_nodes); d->n_pts = reinterpret_cast<int*>(db + lo.n_pts);
    d->n_ties = reinterpret_cast<int*>(db + lo.n_ties); d->nodes = reinterpret_cast<int*>(db + lo.nodes);
    d->node_idx = reinterpret_cast<int*>(db + lo.node_idx); d->coeff = reinterpret_cast<double*>(db + lo.coeff);
    d->path_param = reinterpret_cast<double*>(db + lo.path_param);
}

static void scatter_out(const unsigned char* hb, const OutLayout& lo, int n, ltpl_paths_out* out)
{
    const size_t A = LTPL_MAX_ACTIONS, cn = (size_t)out->cap_nodes, cp = (size_t)out->cap_pts;
    memcpy(out->end_layer, hb + lo.end_layer, sizeof(int) * (size_t)n);
    memcpy(out->closest_obj_index, hb + lo.closest_obj_index, sizeof(int) * (size_t)n);
    memcpy(out->closest_obj_node, hb + lo.closest_obj_node, sizeof(int) * (size_t)n * 2);
    memcpy(out->n_actions, hb + lo.n_actions, sizeof(int) * (size_t)n);
    memcpy(out->action_id, hb + lo.action_id, sizeof(int) * n * A);
    memcpy(out->valid, hb + lo.valid, sizeof(int) * n * A);
    memcpy(out->reduced, hb + lo.reduced, sizeof(int) * n * A);
    memcpy(out->goal_layer, hb + lo.goal_layer, sizeof(int) * n * A);
    memcpy(out->n_nodes, hb + lo.n_nodes, sizeof(int) * n * A);
    memcpy(out->n_pts, hb + lo.n_pts, sizeof(int) * n * A);
    memcpy(out->n_ties, hb + lo.n_ties, sizeof(int) * n * A);
    memcpy(out->nodes, hb + lo.nodes, sizeof(int) * n * A * cn);
    memcpy(out->node_idx, hb + lo.node_idx, sizeof(int) * n * A * cn);
    memcpy(out->coeff, hb + lo.coeff, sizeof(double) * n * A * cn * 8);
    memcpy(out->path_param, hb + lo.path_param, sizeof(double) * n * A * cp * 5);
}

extern "C" int ltpl_plan_paths(ltpl_handle* h, const ltpl_paths_in* in, ltpl_paths_out* out)
{
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (!out) { h->err = "null output"; return LTPL_ERR_INVALID_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    InLayout li;
    int rc = validate_and_layout(h, in, &li);
    if (rc) return rc;
    if (out->cap_nodes < h->caps.max_path_nodes || out->cap_pts < h->caps.max_path_pts) {
        h->err = "output capacity below ltpl_caps.max_path_nodes / max_path_pts"; return LTPL_ERR_CAPACITY;
    }
    OutLayout lo;
    layout_out(in->n_scen, out->cap_nodes, out->cap_pts, &lo);
    if ((rc = ensure(h, &h->h_in, &h->h_in_cap, &h->d_in, &h->d_in_cap, li.total))) return rc;
    if ((rc = ensure(h, &h->h_out, &h->h_out_cap, &h->d_out, &h->d_out_cap, lo.total))) return rc;
    DevPathsIn di; DevPathsOut dout;
    pack_in(in, li, static_cast<unsigned char*>(h->h_in), static_cast<const unsigned char*>(h->d_in), &di);
    bind_out(static_cast<unsigned char*>(h->d_out), lo, out->cap_nodes, out->cap_pts, &dout);
    HIP_TRY(h, hipMemcpyAsync(h->d_in, h->h_in, li.total, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_plan_paths, dim3(in->n_scen), dim3(WG_THREADS), h->lp.total, h->stream, h->lat, di, dout, h->lp);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(h->h_out, h->d_out, lo.total, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    scatter_out(static_cast<const unsigned char*>(h->h_out), lo, in->n_scen, out);
    return LTPL_OK;
}

// --- velocity seam / fused tick: implemented in the next build step ---------------------------------------------------
extern "C" int ltpl_vel_profile(ltpl_handle* h, const ltpl_vel_params*, int, const ltpl_vel_job*, ltpl_vel_result*)
{
    if (!h) return LTPL_ERR_INVALID_ARG;
    h->err = "ltpl_vel_profile: not built yet"; return LTPL_ERR_UNSUPPORTED;
}
extern "C" int ltpl_tick_batch(ltpl_handle* h, const ltpl_paths_in*, const ltpl_tick_vel_in*, ltpl_paths_out*, ltpl_tick_vel_out*)
{
    if (!h) return LTPL_ERR_INVALID_ARG;
    h->err = "ltpl_tick_batch: not built yet"; return LTPL_ERR_UNSUPPORTED;
}
extern "C" int ltpl_batch_upload(ltpl_handle* h, const ltpl_paths_in*, const ltpl_tick_vel_in*, int32_t, int32_t)
{
    if (!h) return LTPL_ERR_INVALID_ARG;
    h->err = "ltpl_batch_upload: not built yet"; return LTPL_ERR_UNSUPPORTED;
}
extern "C" int ltpl_batch_run(ltpl_handle* h, int, float*)
{
    if (!h) return LTPL_ERR_INVALID_ARG;
    h->err = "ltpl_batch_run: not built yet"; return LTPL_ERR_UNSUPPORTED;
}
extern "C" int ltpl_batch_download(ltpl_handle* h, ltpl_paths_out*, ltpl_tick_vel_out*)
{
    if (!h) return LTPL_ERR_INVALID_ARG;
    h->err = "ltpl_batch_download: not built yet"; return LTPL_ERR_UNSUPPORTED;
}
