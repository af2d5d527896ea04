// fleet_dev.hpp -- included at the end of ltpl_hip.hip. The FLEET (include/ltpl_hip.h, ABI v5): N planners whose iterative memory
// (the state of OnlineTrajectoryHandler) lives in DEVICE memory and is advanced by kernels -- one wave64 per planner runs the
// wave-uniform state machine of fleet_core.hpp -- around the launches of the path kernel (seam 1) and the velocity kernel (seam 2):
//
//   calc_paths        H2D inputs | k_fleet_paths_pre | k_paths | k_fleet_paths_post
//   calc_vel_profile  H2D inputs | k_fleet_vel_a | k_vel_profile (<= 5 jobs per planner) | k_fleet_vel_b | k_vel_profile (backup brake jobs)
//                     | k_fleet_vel_c | k_vel_profile (emergency jobs) | k_fleet_vel_d
//
// No host work per planner and no host synchronisation inside a tick (the per-call entry points synchronise once at their end to
// report errors; the tape form replays pre-uploaded inputs for many ticks without any). The host planner (planner_core.hpp) stays
// the form for single vehicles; both are pinned to the same tick recordings of the reference (tests/test_gpu_fleet.py, and through the
// one-lane host build of the same source tests/test_fleet_host_logic.py).

// exec policy of the device build: one wave64, lane = threadIdx.x
struct WaveX {
    static constexpr int W = 64;
    int l;
    __device__ int lane() const { return l; }
    // hand-over of GLOBAL memory between the lanes of the wave (they share the CU's vector cache: workgroup scope is enough)
    __device__ void sync() const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // (distance, index) minimum over the wave. The distance alone first: 12 cross-lane moves instead of 18, and a unique minimum -- the normal
    // case of a closest-point search -- needs no index compare at all; ties take the two-key reduction (same result). The inlined two-key
    // reductions were 18 % of the static instructions of k_fleet_paths_pre (line-table listing, round 4).
    __device__ void argmin(double& d, int& i) const
    {
        const double m = wave_min_f64(d);
        const unsigned long long eq = __ballot(d == m);
        if (eq != 0ull && (eq & (eq - 1ull)) == 0ull) { i = __builtin_amdgcn_readlane(i, __ffsll((long long)eq) - 1); d = m; }
        else wave_min2(d, i);
    }
    __device__ bool any(bool b) const { return __ballot(b) != 0ull; }
    template <class P> __device__ int find_first(int n, P pred) const { return wave_find_first(n, l, pred); }
    // out[i] = term(0) + ... + term(i) in the sequential order of np.cumsum, systolic (see wave_cumsum_seq)
    template <class T> __device__ void scan_seq(int n, T term, double* out) const
    {
        double carry = 0.0;
        for (int base = 0; base < n; base += 64) {
            const int cnt = (n - base) < 64 ? (n - base) : 64;
            const double e = term((l < cnt) ? base + l : base + cnt - 1);
            double s_in = carry, s_out = 0.0;
            for (int it = 0; it < cnt; it += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { s_out = s_in + e; s_in = wave_shr1_f64(s_out, carry); }
            }
            if (l < cnt) out[base + l] = s_out;
            carry = readlane_f64(s_out, cnt - 1);
        }
    }
};

static_assert(sizeof(fleet::VelJob) == sizeof(DevVelJob) && offsetof(fleet::VelJob, obj_y) == offsetof(DevVelJob, obj_y), "job layouts must agree");

struct FleetArgs { fleet::Dims D; fleet::FLat lat; fleet::FCfg cfg; unsigned char* state; int* err_word; const int* rng_end;
                   unsigned char* gg; };      // gg: friction rows of the planners (null until a call carries rows)

// The planner's scalars (PlannerS, ~1 KB) live in LDS while a kernel works on them: every lane executes the scalar control flow on the SAME
// copy (uniform reads are LDS broadcasts; all lanes store the same value to the same address, in lockstep), instead of 64 private copies in
// scratch memory (first version: 2 x 64 KB of scratch traffic per planner and kernel, six kernels per tick -- the fleet was bound by it).
static_assert(sizeof(fleet::PlannerS) % 8 == 0, "PlannerS is copied in 8-byte words");
__device__ __forceinline__ void fleet_load(const WaveX& x, const fleet::Block& B, fleet::PlannerS* S)
{
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(B.S());
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(S);
    for (int i = x.lane(); i < (int)(sizeof(fleet::PlannerS) / 8); i += 64) dst[i] = src[i];
    x.sync();
}
#define FLEET_AXM_ROWS 512          // machine-table rows of one call, all tables together (the lane kernel keeps them in LDS)
#define FLEET_ERR_SHIFT 13
#define FLEET_ERR_MASK 0x1fff
static_assert((fleet::E_EMERG_GG << 8 | 0xff) <= FLEET_ERR_MASK, "error word: site bits");
__device__ __forceinline__ void fleet_store(const WaveX& x, const fleet::Block& B, const fleet::PlannerS* S, int p, int* err_word)
{
    x.sync();
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(S);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(B.S());
    for (int i = x.lane(); i < (int)(sizeof(fleet::PlannerS) / 8); i += 64) dst[i] = src[i];
    // error word: planner + 1 in bits 13.., S.err (code | site << 8; sites need FIVE bits -- E_CAP_VEL = 21) in the low 13 bits
    if (x.lane() == 0 && S->err) atomicCAS(err_word, 0, ((p + 1) << FLEET_ERR_SHIFT) | (S->err & FLEET_ERR_MASK));
}

// OTH.calc_paths in front of seam (1) for planner p (scalars in LDS)
__device__ __forceinline__ void fleet_pre_body(const WaveX& x, const FleetArgs& F, const fleet::Block& B, fleet::PlannerS& S, int p, const fleet::FObj& ob,
                                               const fleet::FPathsIn& pin)
{
    if (!S.err) fleet::paths_pre(x, F.lat, F.cfg, B, S, p, ob, pin);
    if (!S.err && F.rng_end[S.start_node[0]] < 0) fleet::fail(S, LTPL_ERR_INVALID_ARG, fleet::E_NO_RANGE);
    if (S.err && x.lane() == 0) {
        // the path kernel still runs for this planner: give it a harmless scenario (its result is not looked at)
        int l0 = 0; while (l0 < F.lat.L - 1 && F.rng_end[l0] < 0) ++l0;
        pin.start_layer[p] = l0; pin.start_node[p] = F.lat.rl_idx[l0]; pin.flags[p] = LTPL_FLAG_ACTION_SETS; pin.last_action[p] = LTPL_ACT_NONE;
        pin.const_closest[p] = -1; pin.psi_s[p] = 0.0; pin.n_last[p] = 0;
    }
}

__global__ __launch_bounds__(64) void k_fleet_paths_pre(FleetArgs F, fleet::FObj ob, fleet::FPathsIn pin)
{
    const int p = blockIdx.x; const WaveX x{(int)threadIdx.x};
    const fleet::Block B{F.state + F.D.stride * (size_t)p, F.D, F.gg ? F.gg + F.D.gg_stride * (size_t)p : nullptr};
    __shared__ fleet::PlannerS S;
    fleet_load(x, B, &S);
    fleet_pre_body(x, F, B, S, p, ob, pin);
    fleet_store(x, B, &S, p, F.err_word);
}

__global__ __launch_bounds__(64) void k_fleet_paths_post(FleetArgs F, fleet::FPathsOut po)
{
    const int p = blockIdx.x; const WaveX x{(int)threadIdx.x};
    const fleet::Block B{F.state + F.D.stride * (size_t)p, F.D, F.gg ? F.gg + F.D.gg_stride * (size_t)p : nullptr};
    __shared__ fleet::PlannerS S;
    fleet_load(x, B, &S);
    if (!S.err) fleet::paths_post(x, F.lat, B, S, p, po);
    fleet_store(x, B, &S, p, F.err_word);
}

__global__ __launch_bounds__(64) void k_fleet_ref_idx(FleetArgs F, const double* px, const double* py)
{
    const int p = blockIdx.x; const WaveX x{(int)threadIdx.x};
    const fleet::Block B{F.state + F.D.stride * (size_t)p, F.D, F.gg ? F.gg + F.D.gg_stride * (size_t)p : nullptr};
    __shared__ fleet::PlannerS S;
    fleet_load(x, B, &S);
    if (!S.err) { fleet::ref_idx(x, F.cfg, B, S, px[p], py[p]); S.ref_done = 1; }
    fleet_store(x, B, &S, p, F.err_word);
}

// DIGEST of every planner's result of the last tick (round 5: a check of the WHOLE fleet against a recording instead of copying sampled
// planner blocks to the host): per planner LTPL_FLEET_DIGEST doubles --
//   [0] error word, [1] cut_index_pos, [2] cut_layer, [3] keys of the trajectory set, [4] ids, [5] vel_plan, [6] n_vel_course, [7] acc_plan,
//   per key k (LTPL_PLANNER_MAX_KEYS): [8 + 7 k ..] key id, trajectory id, rows, s of the last row, vx of the first / last row, sum of vx
//   per id k: [8 + 7 K + 2 k ..] key id, id value
// -- the quantities the tick recordings hold for every tick (tick_replay.check_trajectories). One wave per planner, read only.
#define LTPL_FLEET_DIGEST (8 + 9 * LTPL_PLANNER_MAX_KEYS)
__global__ __launch_bounds__(64) void k_fleet_digest(FleetArgs F, double* out)
{
    const int p = blockIdx.x; const WaveX x{(int)threadIdx.x};
    const fleet::Block B{F.state + F.D.stride * (size_t)p, F.D, F.gg ? F.gg + F.D.gg_stride * (size_t)p : nullptr};
    __shared__ fleet::PlannerS S;
    fleet_load(x, B, &S);
    double* o = out + (size_t)p * LTPL_FLEET_DIGEST;
    constexpr int K = LTPL_PLANNER_MAX_KEYS;
    for (int i = x.lane(); i < LTPL_FLEET_DIGEST; i += 64) o[i] = 0.0;
    x.sync();
    if (x.lane() == 0) {
        o[0] = (double)S.err; o[1] = (double)S.cut_index_pos; o[2] = (double)S.cut_layer; o[3] = (double)S.n_bp; o[4] = (double)S.n_ids;
        o[5] = S.vel_plan; o[6] = (double)S.n_vel_course; o[7] = S.acc_plan;
        for (int i = 0; i < S.n_ids && i < K; ++i) { o[8 + 7 * K + 2 * i] = (double)S.id_key[i]; o[8 + 7 * K + 2 * i + 1] = (double)S.id_val[i]; }
    }
    for (int i = 0; i < S.n_bp && i < K; ++i) {
        const fleet::Rows r = B.bp(S.bp_slot[i]);
        const int n = S.bp_rows[i];
        double acc = 0.0;
        for (int q = x.lane(); q < n; q += 64) acc += r.at(q, 5);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
        if (x.lane() == 0) {
            double* k = o + 8 + 7 * i;
            k[0] = (double)S.bp_id[i]; k[1] = (double)S.bp_traj_id[i]; k[2] = (double)n;
            k[3] = n > 0 ? r.at(n - 1, 0) : 0.0; k[4] = n > 0 ? r.at(0, 5) : 0.0; k[5] = n > 0 ? r.at(n - 1, 5) : 0.0; k[6] = acc;
        }
    }
}

#ifndef LTPL_FLEET_VELA_WAVES
#define LTPL_FLEET_VELA_WAVES 4       // waves per SIMD the stage-A kernels are compiled for (their register budget): A/B in profiles/r05k_fleet_occ_ab.txt
#endif
__global__ __launch_bounds__(64, LTPL_FLEET_VELA_WAVES) void k_fleet_vel_a(FleetArgs F, fleet::FObj ob, fleet::FVelIn vin, fleet::FJobs JA)
{
    const int p = blockIdx.x; const WaveX x{(int)threadIdx.x};
    const fleet::Block B{F.state + F.D.stride * (size_t)p, F.D, F.gg ? F.gg + F.D.gg_stride * (size_t)p : nullptr};
    __shared__ fleet::PlannerS S;
    fleet_load(x, B, &S);
    fleet::vel_a(x, F.lat, F.cfg, B, S, p, ob, vin, JA);
    fleet_store(x, B, &S, p, F.err_word);
}

// TAPE mode (both calls of a tick and the next tick's inputs are known): stages that follow each other without a seam in between run as
// ONE kernel -- the scalars are loaded / stored once and a kernel boundary (drain + refill of 8 192 short waves) goes away:
//   paths_post + vel_a            (behind seam (1))
//   vel_c | vel_d + paths_pre     (behind the last velocity launch of tick t: the first kernel of tick t + 1)
__global__ __launch_bounds__(64, LTPL_FLEET_VELA_WAVES) void k_fleet_post_vel_a(FleetArgs F, fleet::FPathsOut po, fleet::FObj ob, fleet::FVelIn vin, fleet::FJobs JA)
{
    const int p = blockIdx.x; const WaveX x{(int)threadIdx.x};
    const fleet::Block B{F.state + F.D.stride * (size_t)p, F.D, F.gg ? F.gg + F.D.gg_stride * (size_t)p : nullptr};
    __shared__ fleet::PlannerS S;
    fleet_load(x, B, &S);
    if (!S.err) fleet::paths_post(x, F.lat, B, S, p, po);
    x.sync();
    fleet::vel_a(x, F.lat, F.cfg, B, S, p, ob, vin, JA);
    fleet_store(x, B, &S, p, F.err_word);
}
template <bool D>      // D: the tick had an emergency launch -> stage D in front of the next tick; else stage C
__global__ __launch_bounds__(64) void k_fleet_tail_pre(FleetArgs F, fleet::FVelIn vin, fleet::FJobs J0, fleet::FJobs JC, fleet::FObj ob_next, fleet::FPathsIn pin)
{
    const int p = blockIdx.x; const WaveX x{(int)threadIdx.x};
    const fleet::Block B{F.state + F.D.stride * (size_t)p, F.D, F.gg ? F.gg + F.D.gg_stride * (size_t)p : nullptr};
    __shared__ fleet::PlannerS S;
    fleet_load(x, B, &S);
    if constexpr (D) fleet::vel_d(x, B, S, p, vin, JC);
    else fleet::vel_c(x, F.cfg, B, S, p, vin, J0, JC);
    x.sync();
    fleet_pre_body(x, F, B, S, p, ob_next, pin);
    fleet_store(x, B, &S, p, F.err_word);
}

__global__ __launch_bounds__(64) void k_fleet_vel_b(FleetArgs F, fleet::FJobs JA, fleet::FJobs JB)
{
    const int p = blockIdx.x; const WaveX x{(int)threadIdx.x};
    const fleet::Block B{F.state + F.D.stride * (size_t)p, F.D, F.gg ? F.gg + F.D.gg_stride * (size_t)p : nullptr};
    __shared__ fleet::PlannerS S;
    fleet_load(x, B, &S);
    fleet::vel_b(x, F.cfg, B, S, p, JA, JB);
    fleet_store(x, B, &S, p, F.err_word);
}

__global__ __launch_bounds__(64) void k_fleet_vel_c(FleetArgs F, fleet::FVelIn vin, fleet::FJobs JB, fleet::FJobs JC)
{
    const int p = blockIdx.x; const WaveX x{(int)threadIdx.x};
    const fleet::Block B{F.state + F.D.stride * (size_t)p, F.D, F.gg ? F.gg + F.D.gg_stride * (size_t)p : nullptr};
    __shared__ fleet::PlannerS S;
    fleet_load(x, B, &S);
    fleet::vel_c(x, F.cfg, B, S, p, vin, JB, JC);
    fleet_store(x, B, &S, p, F.err_word);
}

__global__ __launch_bounds__(64) void k_fleet_vel_d(FleetArgs F, fleet::FVelIn vin, fleet::FJobs JC)
{
    const int p = blockIdx.x; const WaveX x{(int)threadIdx.x};
    const fleet::Block B{F.state + F.D.stride * (size_t)p, F.D, F.gg ? F.gg + F.D.gg_stride * (size_t)p : nullptr};
    __shared__ fleet::PlannerS S;
    fleet_load(x, B, &S);
    fleet::vel_d(x, B, S, p, vin, JC);
    fleet_store(x, B, &S, p, F.err_word);
}

// The forward-backward jobs of a tick (slots >= 1 of the stage-A table), one LANE per job: lane_fb_profile of the batch velocity stage on
// the fleet's lane planes. The wave-per-job form computes every step of the recurrence on 64 lanes for one useful result (220 us per tick
// of 8 192 planners); here a wave advances 64 profiles per step.
static_assert(sizeof(fleet::F2) == sizeof(ke_t), "lane plane records: the pair type of the batch velocity stage");
template <int EM, bool AXM1>
__global__ __launch_bounds__(64) void k_fleet_fb_lanes(DevVelParams p, const DevVelJob* jobs, const double* pool, const ke_t* ke, int ke_rows,
                                                       double* outp, int cap, int n_planners, int per, double* out)
{
    extern __shared__ double axm_s[];                         // ALL machine tables of the call, 16 B per row (every lane indexes its own job's table);
                                                              // sized by the launch: a fixed 8 KB took LDS from the follow jobs running next to this kernel (r04i)
    const int lane = threadIdx.x;
    for (int i = lane; i < 2 * p.n_axm; i += 64) axm_s[i] = p.axm[i];
    __syncthreads();
    const int q = blockIdx.x * 64 + lane;
    if (q >= n_planners * (per - 1)) return;
    const int pl = q / (per - 1), slot = q % (per - 1) + 1;
    const DevVelJob* jp = jobs + (size_t)pl * per + slot;
    const int n = jp->n;
    if (n <= 0 || jp->mode != LTPL_VEL_FB || jp->gg_rows) return;      // (jobs with friction ROWS: k_vel_profile<.., GG, SEL 3>, one wave per job)
    LaneProf L; L.KE = ke + kep_base(q, ke_rows);
    double* D = outp + tile_base(q, cap);
    const double cax = pool[jp->off_gg], cay = pool[jp->off_gg + 1];
    // the car of the job: its own vel_max / machine table (fleet::VelJob, ABI v6), else the launch's
    DevVelParams pj = p;
    const double* axm_j = axm_s;
#ifndef LTPL_LANES_UNIFORM_CAR
    if (jp->v_max > 0.0) pj.v_max = jp->v_max;
    if (jp->n_axm > 0) { pj.n_axm = jp->n_axm; axm_j = axm_s + 2 * jp->axm_off; }
#endif
    lane_fb_profile<EM, AXM1>(L, D, 0, n, cax, cay, pj, axm_j, pj.v_max, jp->v_start, jp->has_v_end != 0, jp->v_end);
    // results job-major like every other job's (a lane writes its own row: scattered 8-byte stores, but this launch leaves half of the
    // SIMDs idle and runs next to the follow jobs -- the readers in k_fleet_vel_b then find coalesced rows)
    double* o = out + jp->off_out;
    for (int i = 0; i < n; ++i) o[i] = sqrt(D[(size_t)i * 64]);
}
// The follow jobs of a tick (slot 0 of the stage-A table, LTPL_VEL_FOLLOW_CONTROLLED without friction rows), one LANE per job (round 5): the
// controlled part of calc_vel_profile_follow.py:78-294 exactly as the batch velocity stage runs it (lane_follow_controlled, k_vel_lanes) on
// the job's scalars. The wave-per-job form (k_vel_profile<.., 2>: every step of the recurrences on 64 lanes for one useful result, 14 jobs
// resident per CU) was the longest kernel of a fleet tick -- 180 us for 8 192 planners, more than the path search; this form needs
// n_planners / 64 waves. The unconstrained profile of a follow key is a forward-backward job of its own (k_fleet_fb_lanes) and the
// intersection is stage B's, as before.
#ifndef LTPL_FOLLOW_LANES_MIN_PLANNERS
#define LTPL_FOLLOW_LANES_MIN_PLANNERS 12288
#endif
template <int EM, bool AXM1>
__global__ __launch_bounds__(64) void k_fleet_follow_lanes(DevLat lat, DevVelParams p, const DevVelJob* jobs, const double* pool, const ke_t* ke, int ke_rows,
                                                           int ke_follow, double* P2, double* P3, int cap, int n_planners, int per, double* out, int* flags)
{
    extern __shared__ double axm_s[];                         // ALL machine tables of the call (every lane indexes its own job's table)
    const int lane = threadIdx.x;
    for (int i = lane; i < 2 * p.n_axm; i += 64) axm_s[i] = p.axm[i];
    __syncthreads();
    const int q0 = blockIdx.x * 64 + lane;
    const int q = q0 < n_planners ? q0 : n_planners - 1;      // (idle lanes repeat the last planner: the race line scan below runs on all 64 lanes)
    const DevVelJob* jp = jobs + (size_t)q * per;
    const int n = jp->n;
    const bool mine = q0 < n_planners && n > 0 && jp->mode == LTPL_VEL_FOLLOW_CONTROLLED && !jp->gg_rows;   // (jobs with friction ROWS: k_vel_profile<.., GG, SEL 2>)
    if (__ballot(mine) == 0ull) return;
    const int idx_s_opp = lane_globrl_index(lat, jp->obj_x, jp->obj_y);
    if (!mine) return;
    LaneProf L; L.KE = ke + kep_base(ke_follow + q, ke_rows);
    double* D2 = P2 + tile_base(q, cap); double* D3 = P3 + tile_base(q, cap);
    const double cax = pool[jp->off_gg], cay = pool[jp->off_gg + 1];
    DevVelParams pj = p;
    const double* axm_j = axm_s;
    if (jp->v_max > 0.0) pj.v_max = jp->v_max;
    if (jp->n_axm > 0) { pj.n_axm = jp->n_axm; axm_j = axm_s + 2 * jp->axm_off; }
    const LaneFollowOut fo = lane_follow_controlled<EM, AXM1>(lat, L, D2, D3, n, cax, cay, pj, axm_j, jp->v_start, jp->v_ego, jp->v_obj, jp->obj_dist,
                                                              jp->safety_d, idx_s_opp, nullptr, -1);
    // "vx_profile" (:289 / :294): brake profile in front of row n_decel - 1 (everywhere without a second segment), segment profile up to
    // the stop index, zeros behind; results job-major like every other job's
    double* o = out + jp->off_out;
    for (int i = 0; i < n; ++i) {
        const bool from_b = !fo.two_seg || i < fo.n_decel - 1;
        const double w = from_b ? D2[(size_t)i * 64] : (i > fo.stop_idx ? 0.0 : D3[(size_t)i * 64]);
        o[i] = sqrt(w);
    }
    flags[2 * ((size_t)q * per)] = fo.too_close; flags[2 * ((size_t)q * per) + 1] = fo.vel_bound;
}
typedef void (*fleet_follow_kernel_t)(DevLat, DevVelParams, const DevVelJob*, const double*, const ke_t*, int, int, double*, double*, int, int, int, double*, int*);
static fleet_follow_kernel_t fleet_follow_kernel_of(int v)
{
    switch (v) {
        case 0: return k_fleet_follow_lanes<0, false>; case 1: return k_fleet_follow_lanes<0, true>;
        case 2: return k_fleet_follow_lanes<1, false>; case 3: return k_fleet_follow_lanes<1, true>;
        case 4: return k_fleet_follow_lanes<2, false>; default: return k_fleet_follow_lanes<2, true>;
    }
}
typedef void (*fleet_lanes_kernel_t)(DevVelParams, const DevVelJob*, const double*, const ke_t*, int, double*, int, int, int, double*);
static fleet_lanes_kernel_t fleet_lanes_kernel_of(int v)
{
    switch (v) {
        case 0: return k_fleet_fb_lanes<0, false>; case 1: return k_fleet_fb_lanes<0, true>;
        case 2: return k_fleet_fb_lanes<1, false>; case 3: return k_fleet_fb_lanes<1, true>;
        case 4: return k_fleet_fb_lanes<2, false>; default: return k_fleet_fb_lanes<2, true>;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
struct FleetJobsDev { fleet::VelJob* jobs = nullptr; double* pool = nullptr; double* out = nullptr; int* flags = nullptr; int per = 0;
                      fleet::F2* ke = nullptr; int ke_rows = 0; double* outp = nullptr;          // lane planes (stage-A table only): operands, profile state
                      int ke_follow = -1; double* fp2 = nullptr; double* fp3 = nullptr;          // follow jobs in the lane plane: first job index; brake / segment profile planes
                      fleet::FJobs view() const { return fleet::FJobs{jobs, pool, out, flags, per, ke, ke_rows, ke_follow}; } };

// the inputs of one tick in device memory (one arena per tick of a tape, or the fleet's own for the per-call entry points)
struct FleetTickIn {
    void* d_buf = nullptr; size_t cap = 0;
    fleet::FObj ob{}; const int* zone_off = nullptr; const int* zone_gid = nullptr;
    fleet::FVelIn vin{}; const double* axm = nullptr; int n_axm = 0; double vel_max = 0.0; int any_emerg = 0;
    int has_gg = 0;                          // the call carries friction rows (local_gg as a dict) for some planner
    int n_axm_total = 0, multi_axm = 0;      // ABI v6: rows of all machine tables together; several tables (jobs carry their own): n_axm = rows of table 0
    bool has_paths = false, has_vel = false;
};

struct ltpl_fleet {
    ltpl_handle* h = nullptr;
    std::string err;
    fleet::Dims D{}; fleet::FCfg cfg{}; ltpl_planner_config pc{};
    FleetArgs args{};
    unsigned char* d_state = nullptr; unsigned char* d_gg = nullptr; int* d_err = nullptr; const double* d_w_last = nullptr; int n_w_last = 0;
    std::vector<void*> allocs;
    // seam (1) arrays
    fleet::FPathsIn pin{}; DevPathsOut dout{}; void* d_out = nullptr;
    FleetJobsDev JA, JB, JC;
    hipStream_t stream2 = nullptr; hipEvent_t ev_a = nullptr, ev_b = nullptr;      // the lane kernel of the forward-backward jobs runs next to the follow jobs
    FleetTickIn cur, curv;                            // inputs of the per-call entry points: calc_paths / calc_vel_profile
    void* h_stage = nullptr; size_t h_stage_cap = 0;  // page-locked staging of the inputs
    std::vector<FleetTickIn> tape;
    std::vector<unsigned char> image;                 // host image of one planner block (queries: the views of get_paths / get_trajectories point into it)
    std::vector<unsigned char> start_image;           // scratch image of set_start / set_start_range (a rejected pose leaves `image` alone)
    double* d_digest = nullptr;                       // [N][LTPL_FLEET_DIGEST] of ltpl_fleet_digest, allocated at the first call (in `allocs`)
    bool began = false;
    // some call of this fleet carried friction rows (local_gg as a dict): from then on a planner's MEMORY may hold a backup plan with rows of
    // its own (PlannerS / Block::gg), and vel_b builds the backup brake job from them whatever the current call carries (OTH.py:963-968) --
    // the brake / emergency job launches (JB, JC) keep the rows form of the kernel. (Every pool-form job holds one [ax, ay] row per point,
    // constants replicated, so the rows form serves jobs without rows as well.) Round 4 chose by the current call alone: a backup plan with
    // rows was then solved with its FIRST row's limits in a tick that passed none (advisor finding).
    bool seen_gg = false;
    size_t vel_lds = 0, vel_lds_lite = 0, vel_lds_gg = 0, vel_lds_lite_gg = 0;
    bool tape_fuse = !(getenv("LTPL_FLEET_NO_FUSE") && atoi(getenv("LTPL_FLEET_NO_FUSE")) != 0);     // tape runs with fused stage kernels (results identical)
    ~ltpl_fleet()
    {
        if (h) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); --h->n_planners; }
        if (stream2) { (void)hipStreamSynchronize(stream2); (void)hipStreamDestroy(stream2); }
        if (ev_a) (void)hipEventDestroy(ev_a);
        if (ev_b) (void)hipEventDestroy(ev_b);
        for (void* p : allocs) (void)hipFree(p);
        if (cur.d_buf) (void)hipFree(cur.d_buf);
        if (curv.d_buf) (void)hipFree(curv.d_buf);
        for (FleetTickIn& t : tape) if (t.d_buf) (void)hipFree(t.d_buf);
        if (h_stage) (void)hipHostFree(h_stage);
    }
};
static std::string* abi_err_of(const ltpl_fleet* f) { return f ? const_cast<std::string*>(&f->err) : nullptr; }

#define FLEET_TRY(f, call)                                                                                            \
    do {                                                                                                              \
        hipError_t e_ = (call);                                                                                       \
        if (e_ != hipSuccess) { (f)->err = std::string(#call) + ": " + hipGetErrorString(e_); return LTPL_ERR_HIP; }  \
    } while (0)

template <class T>
static int fleet_alloc(ltpl_fleet* f, size_t n, T** out, bool zero = true)
{
    void* p = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    FLEET_TRY(f, hipMalloc(&p, bytes));
    f->allocs.push_back(p);
    if (zero) FLEET_TRY(f, hipMemset(p, 0, bytes));
    *out = static_cast<T*>(p);
    return LTPL_OK;
}
template <class T>
static int fleet_upload(ltpl_fleet* f, const std::vector<T>& v, const T** out)
{
    T* p = nullptr;
    int rc = fleet_alloc(f, v.size(), &p, false);
    if (rc) return rc;
    if (!v.empty()) FLEET_TRY(f, hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    *out = p;
    return LTPL_OK;
}

static int fleet_jobs_alloc(ltpl_fleet* f, FleetJobsDev* J, int per)
{
    const size_t n = (size_t)f->D.N * (size_t)per;
    if (n * 4 * (size_t)f->D.R > 0x7fffffffull) { f->err = "fleet: job pool offsets exceed 31 bits (fewer planners per fleet)"; return LTPL_ERR_CAPACITY; }
    J->per = per;
    int rc;
    if ((rc = fleet_alloc(f, n, &J->jobs))) return rc;
    if ((rc = fleet_alloc(f, n * 4 * (size_t)f->D.R, &J->pool))) return rc;
    if ((rc = fleet_alloc(f, n * (size_t)f->D.R, &J->out))) return rc;
    return fleet_alloc(f, n * 2, &J->flags);
}

extern "C" int ltpl_fleet_create(ltpl_handle* h, const ltpl_planner_config* cfg, ltpl_fleet** out)
try {
    g_create_error.clear();
    if (!h || !out) { g_create_error = "fleet: null argument"; return LTPL_ERR_INVALID_ARG; }
    if (!h->has_hostlat) { g_create_error = "fleet: the lattice was created without raceline_x / raceline_y / node_psi"; return LTPL_ERR_UNSUPPORTED; }
    int rc = fleet::check_config(cfg, &g_create_error);
    if (rc) return rc;
    if (hipSetDevice(h->device) != hipSuccess) { g_create_error = "fleet: hipSetDevice failed"; return LTPL_ERR_HIP; }
    std::unique_ptr<ltpl_fleet> f(new ltpl_fleet());
    f->h = h; ++h->n_planners;
    f->cfg = fleet::fcfg_of(cfg); f->pc = *cfg; f->pc.w_last_edges = nullptr;
    f->D = fleet::make_dims(cfg->n_scen, h->caps.max_path_nodes, h->caps.max_path_pts);
    const int N = f->D.N;
    auto bail = [&](int code) { g_create_error = f->err; return code; };
    if ((rc = fleet::check_dims(f->D, &f->err))) return bail(rc);
    if (N >= (1 << (31 - FLEET_ERR_SHIFT)) - 1) { f->err = "fleet: more planners than the error word can name (fewer planners per fleet)"; return bail(LTPL_ERR_CAPACITY); }
    if ((rc = fleet_alloc(f.get(), f->D.stride * (size_t)N, &f->d_state))) return bail(rc);
    if ((rc = fleet_alloc(f.get(), 4, &f->d_err))) return bail(rc);
    {   // all planners start without memory (the fields whose "nothing" is not zero)
        std::vector<unsigned char> img(f->D.stride, 0);
        fleet::PlannerS* S = reinterpret_cast<fleet::PlannerS*>(img.data());
        S->em_base_id = S->action_forced = S->sel_action = S->raw_action = LTPL_ACT_NONE; S->closest_obj_index = -1; S->const_rows = -1; S->old_gg_scale = 1.0;
        for (int p = 0; p < N; ++p)
            if (hipMemcpy(f->d_state + f->D.stride * (size_t)p, img.data(), sizeof(fleet::PlannerS), hipMemcpyHostToDevice) != hipSuccess) { g_create_error = "fleet: state upload failed"; return LTPL_ERR_HIP; }
    }
    // lattice tables of the state machine
    const ltplp::HostLat& hl = h->hostlat;
    fleet::FLat fl = fleet::flat_of(hl);
    fl.layer_off = h->lat.layer_off; fl.rl_idx = h->lat.rl_idx; fl.s_rl = h->lat.s_rl; fl.vel_rl = h->lat.vel_rl; fl.node_x = h->lat.node_x; fl.node_y = h->lat.node_y;
    if ((rc = fleet_upload(f.get(), hl.race_x, &fl.race_x))) return bail(rc);
    if ((rc = fleet_upload(f.get(), hl.race_y, &fl.race_y))) return bail(rc);
    std::vector<double> wl(cfg->n_w_last > 0 ? cfg->w_last_edges : nullptr, cfg->n_w_last > 0 ? cfg->w_last_edges + cfg->n_w_last : nullptr);
    f->n_w_last = cfg->n_w_last; wl.push_back(0.0);
    if ((rc = fleet_upload(f.get(), wl, &f->d_w_last))) return bail(rc);
    f->args.D = f->D; f->args.lat = fl; f->args.cfg = f->cfg; f->args.state = f->d_state; f->args.gg = nullptr; f->args.err_word = f->d_err; f->args.rng_end = h->lat.rng_end;
    // seam (1)
    if ((rc = fleet_alloc(f.get(), (size_t)N, &f->pin.start_layer))) return bail(rc);
    if ((rc = fleet_alloc(f.get(), (size_t)N, &f->pin.start_node))) return bail(rc);
    if ((rc = fleet_alloc(f.get(), (size_t)N, &f->pin.flags))) return bail(rc);
    if ((rc = fleet_alloc(f.get(), (size_t)N, &f->pin.last_action))) return bail(rc);
    if ((rc = fleet_alloc(f.get(), (size_t)N, &f->pin.const_closest))) return bail(rc);
    if ((rc = fleet_alloc(f.get(), (size_t)N, &f->pin.psi_s))) return bail(rc);
    if ((rc = fleet_alloc(f.get(), (size_t)N, &f->pin.n_last))) return bail(rc);
    if ((rc = fleet_alloc(f.get(), (size_t)N * LTPL_MAX_LAST_NODES, &f->pin.last_layer))) return bail(rc);
    if ((rc = fleet_alloc(f.get(), (size_t)N * LTPL_MAX_LAST_NODES, &f->pin.last_node))) return bail(rc);
    {
        OutLayout lo; layout_out(N, f->D.cn, f->D.cp, &lo);
        unsigned char* d = nullptr;
        if ((rc = fleet_alloc(f.get(), lo.total, &d))) return bail(rc);
        bind_out(d, lo, f->D.cn, f->D.cp, &f->dout);
    }
    if ((rc = fleet_jobs_alloc(f.get(), &f->JA, fleet::JOBS_A))) return bail(rc);
    {   // lane planes of the forward-backward jobs (slots 1 .. JOBS_A - 1), tiles of 64 jobs; round 5: behind them the tiles of the follow
        // jobs (slot 0), which run one LANE per job as well (k_fleet_follow_lanes; LTPL_FLEET_FOLLOW_WAVES=1 keeps the wave-per-job form)
        const size_t tiles = ((size_t)N * (fleet::JOBS_A - 1) + 63) / 64, ftiles = ((size_t)N + 63) / 64;
        // same-box A/B on the mixed tape (profiles/r05d_fleet_follow_ab.txt): 32 768 planners 15.9 M planner-ticks/s (lanes) against 14.8 M (waves);
        // 8 192 planners 12.3 M against 12.5 M -- a small fleet's follow kernel hides behind the forward-backward lane kernel on the second
        // stream either way, a large fleet's wave-per-job kernel is nine rounds of 14 jobs per CU. The lane form is the default from
        // LTPL_FOLLOW_LANES_MIN_PLANNERS (12 288) planners on; LTPL_FLEET_FOLLOW_WAVES=1 / =0 forces one or the other.
        const char* fw = getenv("LTPL_FLEET_FOLLOW_WAVES");
        const bool follow_lanes = fw ? atoi(fw) == 0 : N >= LTPL_FOLLOW_LANES_MIN_PLANNERS;
        f->JA.ke_rows = (f->D.RV + 7) / 8 * 8;
        if ((rc = fleet_alloc(f.get(), (tiles + (follow_lanes ? ftiles : 0)) * 64 * (size_t)f->JA.ke_rows, &f->JA.ke))) return bail(rc);
        if ((rc = fleet_alloc(f.get(), tiles * 64 * (size_t)f->D.RV, &f->JA.outp))) return bail(rc);
        if (follow_lanes) {
            f->JA.ke_follow = (int)(tiles * 64);
            if ((rc = fleet_alloc(f.get(), ftiles * 64 * (size_t)f->D.RV, &f->JA.fp2))) return bail(rc);
            if ((rc = fleet_alloc(f.get(), ftiles * 64 * (size_t)f->D.RV, &f->JA.fp3))) return bail(rc);
        }
    }
    if ((rc = fleet_jobs_alloc(f.get(), &f->JB, 1))) return bail(rc);
    if ((rc = fleet_jobs_alloc(f.get(), &f->JC, 1))) return bail(rc);
    if (hipStreamCreateWithFlags(&f->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&f->ev_a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&f->ev_b, hipEventDisableTiming) != hipSuccess) return bail((f->err = "fleet: stream / event creation failed", LTPL_ERR_HIP));
    f->vel_lds = vel_scratch_bytes(f->D.RV, false, false); f->vel_lds_lite = vel_scratch_bytes(f->D.RV, false, false, true);
    f->vel_lds_gg = vel_scratch_bytes(f->D.RV, true, false); f->vel_lds_lite_gg = vel_scratch_bytes(f->D.RV, true, false, true);
    if (f->vel_lds > 150 * 1024) return bail((f->err = "fleet: velocity profile too long for the LDS-resident solver", LTPL_ERR_CAPACITY));
    f->image.resize(f->D.stride);
    *out = f.release();
    return LTPL_OK;
} LTPL_ABI_CATCH(nullptr)

extern "C" int ltpl_fleet_destroy(ltpl_fleet* f) { delete f; return LTPL_OK; }
extern "C" int ltpl_fleet_get_caps(const ltpl_fleet* f, ltpl_planner_caps* c) { if (!f || !c) return LTPL_ERR_INVALID_ARG; fleet::caps_of(f->D, c); return LTPL_OK; }
extern "C" const char* ltpl_fleet_last_error(const ltpl_fleet* f) { return f ? f->err.c_str() : g_create_error.c_str(); }

// error word of the kernels -> status + message; clears the word
static int fleet_check(ltpl_fleet* f)
{
    int w = 0;
    FLEET_TRY(f, hipMemcpyAsync(&w, f->d_err, sizeof(int), hipMemcpyDeviceToHost, f->h->stream));
    FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
    if (!w) return LTPL_OK;
    FLEET_TRY(f, hipMemsetAsync(f->d_err, 0, sizeof(int), f->h->stream));
    const int p = (w >> FLEET_ERR_SHIFT) - 1, e = w & FLEET_ERR_MASK;
    f->err = fleet::err_text(p, e) + " (the fleet keeps the planner's error state: set a new start pose to clear it)";
    return e & 0xff;
}

extern "C" int ltpl_fleet_set_start(ltpl_fleet* f, int32_t p, double x, double y, double heading, double vel, double mho, int32_t* in_track, int32_t* cor_heading)
try {
    if (!f || !in_track || !cor_heading) return LTPL_ERR_INVALID_ARG;
    if (p < 0 || p >= f->D.N) { f->err = "fleet: planner index out of range"; return LTPL_ERR_INVALID_ARG; }
    FLEET_TRY(f, hipSetDevice(f->h->device));
    FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
    fleet::PlannerS prev;
    unsigned char* blk = f->d_state + f->D.stride * (size_t)p;
    FLEET_TRY(f, hipMemcpy(&prev, blk, sizeof(prev), hipMemcpyDeviceToHost));
    if (f->start_image.size() != f->image.size()) f->start_image.assign(f->image.size(), 0);
    const int rc = fleet::start_block(f->h->hostlat, f->D, x, y, heading, vel, mho, in_track, cor_heading, f->start_image.data(), &prev, &f->err);
    if (rc) return rc;                                // (rejected pose: neither the planner's block nor the query image was touched)
    f->image = f->start_image;
    FLEET_TRY(f, hipMemcpy(blk, f->image.data(), f->D.stride, hipMemcpyHostToDevice));
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(f))

// the same start pose for the planners [p0, p1) in ONE call (a fleet that starts from a grid, a benchmark that starts thousands of
// planners): the start spline is computed once, its block image goes to the device once and is copied there planner by planner; what
// set_initial_pose carries over from a planner's previous life (trajectory id counter, calculation-time buffer, ...: start_block) is read
// back and patched per planner. (ltpl_fleet_set_start per planner: two synchronous copies of the whole block each -- 0.4 s for 8 192.)
extern "C" int ltpl_fleet_set_start_range(ltpl_fleet* f, int32_t p0, int32_t p1, double x, double y, double heading, double vel, double mho,
                                          int32_t* in_track, int32_t* cor_heading)
try {
    if (!f || !in_track || !cor_heading) return LTPL_ERR_INVALID_ARG;
    if (p0 < 0 || p1 > f->D.N || p0 >= p1) { f->err = "fleet: planner range out of bounds"; return LTPL_ERR_INVALID_ARG; }
    // the start pose first: it only needs the host's lattice tables, and a pose off the track (or a heading the track does not allow) must
    // cost neither a device synchronisation nor a read-back, and must leave the planners' memory untouched
    // (built in a scratch image: a rejected pose leaves the image the views of earlier get_paths / get_trajectories calls point into alone)
    if (f->start_image.size() != f->image.size()) f->start_image.assign(f->image.size(), 0);
    const int rc = fleet::start_block(f->h->hostlat, f->D, x, y, heading, vel, mho, in_track, cor_heading, f->start_image.data(), nullptr, &f->err);
    if (rc) return rc;
    f->image = f->start_image;
    FLEET_TRY(f, hipSetDevice(f->h->device));
    FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
    const size_t cnt = (size_t)(p1 - p0), hs = sizeof(fleet::PlannerS);
    std::vector<fleet::PlannerS> prev(cnt);
    unsigned char* blk0 = f->d_state + f->D.stride * (size_t)p0;
    FLEET_TRY(f, hipMemcpy2D(prev.data(), hs, blk0, f->D.stride, hs, cnt, hipMemcpyDeviceToHost));
    // per-planner scalars: the image's header with the fields start_block carries over from the planner's state before the call
    const fleet::PlannerS base = *reinterpret_cast<const fleet::PlannerS*>(f->image.data());
    std::vector<fleet::PlannerS> hdr(cnt, base);
    for (size_t q = 0; q < cnt; ++q) {
        fleet::PlannerS& S = hdr[q]; const fleet::PlannerS& keep = prev[q];
        S.traj_base_id = keep.traj_base_id; S.n_calc = keep.n_calc; std::memcpy(S.calc_buffer, keep.calc_buffer, sizeof(S.calc_buffer));
        S.has_old_gg = keep.has_old_gg; S.cut_index_pos = keep.cut_index_pos; S.cut_layer = keep.cut_layer; S.vel_plan = keep.vel_plan; S.acc_plan = keep.acc_plan;
        S.em_base_id = keep.em_base_id; S.closest_obj_index = keep.closest_obj_index; S.old_gg_scale = keep.old_gg_scale;
    }
    // From here on the range is being rewritten: a HIP failure below (the device is gone -- every later call fails as well) leaves the blocks
    // of the range UNDEFINED; the error is returned and the fleet is to be destroyed. The image goes to the device once, with the first
    // planner's own header already in place, then device-to-device into every other block of the range, then the other planners' headers.
    std::memcpy(f->image.data(), &hdr[0], hs);
    FLEET_TRY(f, hipMemcpy(blk0, f->image.data(), f->D.stride, hipMemcpyHostToDevice));
    for (size_t q = 1; q < cnt; ++q)
        FLEET_TRY(f, hipMemcpyAsync(blk0 + f->D.stride * q, blk0, f->D.stride, hipMemcpyDeviceToDevice, f->h->stream));
    FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
    FLEET_TRY(f, hipMemcpy2D(blk0, f->D.stride, hdr.data(), hs, hs, cnt, hipMemcpyHostToDevice));
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(f))

static int fleet_stage(ltpl_fleet* f, size_t bytes)
{
    if (bytes <= f->h_stage_cap) return LTPL_OK;
    FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
    if (f->h_stage) (void)hipHostFree(f->h_stage);
    f->h_stage = nullptr; f->h_stage_cap = 0;
    const size_t cap = align_up(bytes + bytes / 2, 4096);
    FLEET_TRY(f, hipHostMalloc(&f->h_stage, cap, hipHostMallocDefault));
    f->h_stage_cap = cap;
    return LTPL_OK;
}

// packs both halves of a tick's inputs into ONE device arena (t->d_buf) and binds the views; `pin` / `vin` may be null (that half keeps its
// previous binding only if the arena is not reallocated, so the per-call entry points always pass what they need)
static int fleet_pack_inputs(ltpl_fleet* f, FleetTickIn* t, const ltpl_planner_paths_in* pin, const ltpl_planner_vel_in* vin, bool use_zones)
{
    const int N = f->D.N;
    Arena a;
    size_t o_pa = 0, o_tn = 0, o_vo = 0, o_po = 0, o_ra = 0, o_ve = 0, o_px = 0, o_py = 0, o_zo = 0, o_zg = 0;
    int nv = 0, np_ = 0, nz = 0;
    if (pin) {
        if (!pin->prev_action || !pin->t_now || !pin->veh_off || !pin->pos_off) { f->err = "fleet: null input"; return LTPL_ERR_INVALID_ARG; }
        if (pin->veh_off[0] != 0 || pin->pos_off[0] != 0) { f->err = "offset arrays must start at 0"; return LTPL_ERR_INVALID_ARG; }
        nv = pin->veh_off[N];
        if (nv < 0) { f->err = "negative offsets"; return LTPL_ERR_INVALID_ARG; }
        np_ = pin->pos_off[nv];
        if ((nv > 0 && !pin->veh_radius) || (np_ > 0 && (!pin->pos_x || !pin->pos_y))) { f->err = "fleet: null vehicle / position arrays"; return LTPL_ERR_INVALID_ARG; }
        for (int s = 0; s < N; ++s) {
            const int c = pin->veh_off[s + 1] - pin->veh_off[s];
            if (c < 0 || c > MAX_VEH) { f->err = "more than 96 vehicles for one planner"; return LTPL_ERR_CAPACITY; }
            const int q = pin->pos_off[pin->veh_off[s + 1]] - pin->pos_off[pin->veh_off[s]];
            if (q < 0 || q > MAX_POS) { f->err = "more than 192 obstacle positions for one planner"; return LTPL_ERR_CAPACITY; }
            for (int v = pin->veh_off[s]; v < pin->veh_off[s + 1]; ++v)
                if (pin->pos_off[v + 1] - pin->pos_off[v] < 1) { f->err = "vehicle without position"; return LTPL_ERR_INVALID_ARG; }
        }
        if (use_zones) {
            if (!pin->zone_off || pin->zone_off[0] != 0) { f->err = "fleet: zone offsets missing"; return LTPL_ERR_INVALID_ARG; }
            for (int s = 0; s < N; ++s) if (pin->zone_off[s + 1] < pin->zone_off[s]) { f->err = "fleet: zone offsets must not decrease"; return LTPL_ERR_INVALID_ARG; }
            nz = pin->zone_off[N];
            if (nz > 0 && !pin->zone_gid) { f->err = "fleet: zone node ids missing"; return LTPL_ERR_INVALID_ARG; }
            for (int i = 0; i < nz; ++i) if (pin->zone_gid[i] < 0 || pin->zone_gid[i] >= f->h->lat.V) { f->err = "zone node id out of range"; return LTPL_ERR_INVALID_ARG; }
        }
        o_pa = a.add(4 * (size_t)N); o_tn = a.add(8 * (size_t)N); o_vo = a.add(4 * (size_t)(N + 1)); o_po = a.add(4 * (size_t)(nv + 1));
        o_ra = a.add(8 * (size_t)(nv + 1)); o_ve = a.add(8 * (size_t)(nv + 1)); o_px = a.add(8 * (size_t)(np_ + 1)); o_py = a.add(8 * (size_t)(np_ + 1));
        o_zo = a.add(4 * (size_t)(N + 1)); o_zg = a.add(4 * (size_t)(nz + 1));
    }
    size_t o_v[8] = {0}, o_em = 0, o_axm = 0, o_axo = 0, o_axi = 0, o_ggo = 0, o_ggr = 0;
    int n_tab = 0, n_gg = 0;
    if (vin) {
        if (!vin->pos_est_x || !vin->pos_est_y || !vin->vel_est || !vin->vel_max || !vin->gg_scale || !vin->gg_ax || !vin->gg_ay || !vin->safety_d ||
            !vin->ax_max_machines || vin->n_ax_max_machines < 1) { f->err = "fleet: null input"; return LTPL_ERR_INVALID_ARG; }
        // location dependent friction (local_gg as a dict, OTH.py:649-666): rows per planner and path key
        if ((vin->gg_row_off != nullptr) != (vin->gg_rows != nullptr)) { f->err = "fleet: gg_row_off and gg_rows come together"; return LTPL_ERR_INVALID_ARG; }
        if (vin->gg_row_off) {
            const int MK = LTPL_PLANNER_MAX_KEYS;
            if (vin->gg_row_off[0] != 0) { f->err = "fleet: gg_row_off must start at 0"; return LTPL_ERR_INVALID_ARG; }
            for (int i = 0; i < N * MK; ++i) if (vin->gg_row_off[i + 1] < vin->gg_row_off[i]) { f->err = "fleet: gg_row_off must be non-decreasing"; return LTPL_ERR_INVALID_ARG; }
            n_gg = vin->gg_row_off[N * MK];
            if (n_gg > 0 && f->vel_lds_gg > 150 * 1024) { f->err = "fleet: velocity profile with friction rows too long for the LDS-resident solver"; return LTPL_ERR_CAPACITY; }
            if (n_gg > 0 && !f->d_gg) {        // the planners' friction rows: allocated by the first call that carries rows (no kernel of this fleet is in flight that could read the pointer)
                FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
                int rc_ = fleet_alloc(f, f->D.gg_stride * (size_t)N, &f->d_gg, false);
                if (rc_) return rc_;
                f->args.gg = f->d_gg;
            }
        }
        // a fleet of different cars (ABI v6): vel_max per planner, machine tables per planner (ax_table_off / ax_table_idx)
        n_tab = vin->n_ax_tables > 1 ? vin->n_ax_tables : 0;
        if (n_tab) {
            if (!vin->ax_table_off || !vin->ax_table_idx) { f->err = "fleet: n_ax_tables > 1 without ax_table_off / ax_table_idx"; return LTPL_ERR_INVALID_ARG; }
            if (vin->ax_table_off[0] != 0 || vin->ax_table_off[n_tab] != vin->n_ax_max_machines) { f->err = "fleet: ax_table_off must run from 0 to n_ax_max_machines"; return LTPL_ERR_INVALID_ARG; }
            for (int t_ = 0; t_ < n_tab; ++t_) {
                const int rows = vin->ax_table_off[t_ + 1] - vin->ax_table_off[t_];
                if (rows < 1 || rows > 64) { f->err = "fleet: a machine table needs 1 .. 64 rows"; return rows < 1 ? LTPL_ERR_INVALID_ARG : LTPL_ERR_CAPACITY; }
            }
            if (vin->n_ax_max_machines > FLEET_AXM_ROWS) { f->err = "fleet: more than 512 machine-table rows in one call"; return LTPL_ERR_CAPACITY; }
            for (int s = 0; s < N; ++s) if (vin->ax_table_idx[s] < 0 || vin->ax_table_idx[s] >= n_tab) { f->err = "fleet: ax_table_idx out of range"; return LTPL_ERR_INVALID_ARG; }
        } else if (vin->n_ax_max_machines > 64) { f->err = "ax_max_machines with more than 64 rows"; return LTPL_ERR_CAPACITY; }
        for (int s = 0; s < N; ++s) if (!(vin->vel_max[s] > 0.0)) { f->err = "fleet: vel_max must be positive"; return LTPL_ERR_INVALID_ARG; }
        for (int k = 0; k < 8; ++k) o_v[k] = a.add(8 * (size_t)N);
        o_em = a.add(4 * (size_t)N); o_axm = a.add(16 * (size_t)vin->n_ax_max_machines);
        if (n_tab) { o_axo = a.add(4 * (size_t)(n_tab + 1)); o_axi = a.add(4 * (size_t)N); }
        if (n_gg > 0) { o_ggo = a.add(4 * ((size_t)N * LTPL_PLANNER_MAX_KEYS + 1)); o_ggr = a.add(16 * (size_t)n_gg); }
    }
    int rc = fleet_stage(f, a.size);
    if (rc) return rc;
    if (a.size > t->cap) {
        FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
        if (t->d_buf) (void)hipFree(t->d_buf);
        t->d_buf = nullptr; t->cap = 0; t->has_paths = t->has_vel = false;
        const size_t cap = align_up(a.size + a.size / 2, 4096);
        FLEET_TRY(f, hipMalloc(&t->d_buf, cap));
        t->cap = cap;
    }
    // the staging buffer is reused by the next call: wait for the previous copy out of it
    FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
    unsigned char* hb = static_cast<unsigned char*>(f->h_stage); unsigned char* db = static_cast<unsigned char*>(t->d_buf);
    if (pin) {
        memcpy(hb + o_pa, pin->prev_action, 4 * (size_t)N); memcpy(hb + o_tn, pin->t_now, 8 * (size_t)N);
        memcpy(hb + o_vo, pin->veh_off, 4 * (size_t)(N + 1)); memcpy(hb + o_po, pin->pos_off, 4 * (size_t)(nv + 1));
        if (nv) { memcpy(hb + o_ra, pin->veh_radius, 8 * (size_t)nv); if (pin->veh_vel) memcpy(hb + o_ve, pin->veh_vel, 8 * (size_t)nv); else memset(hb + o_ve, 0, 8 * (size_t)nv); }
        if (np_) { memcpy(hb + o_px, pin->pos_x, 8 * (size_t)np_); memcpy(hb + o_py, pin->pos_y, 8 * (size_t)np_); }
        if (use_zones) { memcpy(hb + o_zo, pin->zone_off, 4 * (size_t)(N + 1)); if (nz) memcpy(hb + o_zg, pin->zone_gid, 4 * (size_t)nz); }
        else memset(hb + o_zo, 0, 4 * (size_t)(N + 1));
        t->ob = fleet::FObj{reinterpret_cast<const int*>(db + o_pa), reinterpret_cast<const double*>(db + o_tn), reinterpret_cast<const int*>(db + o_vo),
                            reinterpret_cast<const int*>(db + o_po), reinterpret_cast<const double*>(db + o_ra), reinterpret_cast<const double*>(db + o_ve),
                            reinterpret_cast<const double*>(db + o_px), reinterpret_cast<const double*>(db + o_py)};
        t->zone_off = reinterpret_cast<const int*>(db + o_zo); t->zone_gid = reinterpret_cast<const int*>(db + o_zg);
        t->has_paths = true;
    }
    if (vin) {
        const double* src[8] = {vin->pos_est_x, vin->pos_est_y, vin->vel_est, vin->vel_max, vin->gg_scale, vin->gg_ax, vin->gg_ay, vin->safety_d};
        for (int k = 0; k < 8; ++k) memcpy(hb + o_v[k], src[k], 8 * (size_t)N);
        int any = 0;
        if (vin->incl_emerg_traj) { memcpy(hb + o_em, vin->incl_emerg_traj, 4 * (size_t)N); for (int s = 0; s < N; ++s) any |= vin->incl_emerg_traj[s]; }
        else memset(hb + o_em, 0, 4 * (size_t)N);
        memcpy(hb + o_axm, vin->ax_max_machines, 16 * (size_t)vin->n_ax_max_machines);
        auto dp = [&](int k) { return reinterpret_cast<const double*>(db + o_v[k]); };
        if (n_tab) { memcpy(hb + o_axo, vin->ax_table_off, 4 * (size_t)(n_tab + 1)); memcpy(hb + o_axi, vin->ax_table_idx, 4 * (size_t)N); }
        t->vin = fleet::FVelIn{dp(0), dp(1), dp(2), dp(3), dp(4), dp(5), dp(6), dp(7), reinterpret_cast<const int*>(db + o_em),
                               n_tab ? reinterpret_cast<const int*>(db + o_axo) : nullptr, n_tab ? reinterpret_cast<const int*>(db + o_axi) : nullptr,
                               n_gg > 0 ? reinterpret_cast<const int*>(db + o_ggo) : nullptr, n_gg > 0 ? reinterpret_cast<const double*>(db + o_ggr) : nullptr};
        if (n_gg > 0) { memcpy(hb + o_ggo, vin->gg_row_off, 4 * ((size_t)N * LTPL_PLANNER_MAX_KEYS + 1)); memcpy(hb + o_ggr, vin->gg_rows, 16 * (size_t)n_gg); }
        t->has_gg = n_gg > 0 ? 1 : 0;
        t->axm = reinterpret_cast<const double*>(db + o_axm); t->vel_max = vin->vel_max[0]; t->any_emerg = any;
        t->n_axm_total = vin->n_ax_max_machines; t->multi_axm = n_tab ? 1 : 0;
        t->n_axm = n_tab ? vin->ax_table_off[1] - vin->ax_table_off[0] : vin->n_ax_max_machines;
        t->has_vel = true;
    }
    if (a.size) FLEET_TRY(f, hipMemcpyAsync(t->d_buf, f->h_stage, a.size, hipMemcpyHostToDevice, f->h->stream));
    return LTPL_OK;
}

static int fleet_launch_paths(ltpl_fleet* f, const FleetTickIn& t, bool pre, bool rest, bool post = true, fleet::FPathsOut* po_out = nullptr)
{
    ltpl_handle* h = f->h; const int N = f->D.N; hipStream_t st = h->stream;
    if (pre) {
        hipLaunchKernelGGL(k_fleet_paths_pre, dim3(N), dim3(64), 0, st, f->args, t.ob, f->pin);
        FLEET_TRY(f, hipGetLastError());
    }
    if (!rest) return LTPL_OK;
    DevPathsIn di;
    di.n_scen = N; di.n_w_last = f->n_w_last; di.w_last = f->d_w_last;
    di.start_layer = f->pin.start_layer; di.start_node = f->pin.start_node; di.flags = f->pin.flags; di.last_action = f->pin.last_action;
    di.const_closest = f->pin.const_closest; di.psi_s = f->pin.psi_s;
    di.veh_off = t.ob.veh_off; di.pos_off = t.ob.pos_off; di.veh_radius = t.ob.radius; di.pos_x = t.ob.px; di.pos_y = t.ob.py;
    di.zone_off = t.zone_off; di.zone_gid = t.zone_gid;
    di.n_last = f->pin.n_last; di.last_layer = f->pin.last_layer; di.last_node = f->pin.last_node;
    di.order = nullptr;                 // (planner p is planned by block p: the start layers live in device memory, the fleet does not reorder)
    const int nw = (N >= h->nw1_min_scen && h->batch_nw == 1) ? 1 : NUM_WAVES;
    const int rc = launch_paths(h, nw, N, st, di, f->dout);
    if (rc) { f->err = h->err; return rc; }
    const fleet::FPathsOut po{f->dout.closest_obj_index, f->dout.n_actions, f->dout.action_id, f->dout.valid, f->dout.reduced, f->dout.n_nodes, f->dout.n_pts,
                              f->dout.nodes, f->dout.node_idx, f->dout.coeff, f->dout.path_param};
    if (po_out) *po_out = po;
    if (!post) return LTPL_OK;          // (tape mode: paths_post runs inside the first kernel of the velocity step)
    hipLaunchKernelGGL(k_fleet_paths_post, dim3(N), dim3(64), 0, st, f->args, po);
    FLEET_TRY(f, hipGetLastError());
    return LTPL_OK;
}

// one launch of the velocity kernel over a job table. sel 1: forward-backward / brake jobs of all slots ("lite" LDS scratch); sel 2: the
// follow jobs (slot 0 of every planner)
// kernel variant of a launch: with several machine tables (a fleet of different cars) the interpolating form, whatever table 0 looks like
static int fleet_variant(const ltpl_vel_params& vp, bool multi) { const int v = vel_variant(&vp); return multi ? (v & ~1) : v; }

static int fleet_launch_vel_jobs(ltpl_fleet* f, const ltpl_vel_params& vp, const double* d_axm, const FleetJobsDev& J, int sel, bool multi = false, bool rows = false)
{
    ltpl_handle* h = f->h;
    DevVelParams p;
    int rc = make_vel_params(h, &vp, d_axm, &p);
    if (rc) { f->err = h->err; return rc; }
    // `rows`: the call carries friction rows -- every pool-form job holds one [ax, ay] row per point (constants replicated), so the GG
    // form of the kernel serves all of them; sel 3 then picks the forward-backward jobs the lane kernel left out
    const int v = fleet_variant(vp, multi);
    vel_kernel_t kern = rows ? (sel == 1 ? vel_kernel_rows_of<1>(v) : sel == 2 ? vel_kernel_rows_of<2>(v) : vel_kernel_rows_of<3>(v))
                             : (sel == 1 ? vel_kernel_const_of<1>(v) : vel_kernel_const_of<2>(v));
    const size_t lds = rows ? (sel == 2 ? f->vel_lds_gg : f->vel_lds_lite_gg) : (sel == 1 ? f->vel_lds_lite : f->vel_lds);
    if (lds > 48 * 1024) FLEET_TRY(f, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DoneSignal done; done.host_flag = nullptr; done.dev_count = nullptr; done.seq = 0u;
    const unsigned blocks = (unsigned)(sel != 2 ? f->D.N * J.per : f->D.N);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, h->stream, h->lat, p, reinterpret_cast<const DevVelJob*>(J.jobs),
                       J.pool, J.out, J.flags, f->D.RV, h->lp4.dbg, done, sel != 2 ? 1 : J.per);
    FLEET_TRY(f, hipGetLastError());
#ifdef LTPL_EXPERIMENT
    if (h->d_dbg && J.per > 1) {        // LTPL_DEBUG_TIMING=1 (experiment build): cycle stamps of the first 256 blocks of the launch, per slot index
        std::vector<long long> v((size_t)256 * DBG_SLOTS);
        if (hipMemcpy(v.data(), h->d_dbg, v.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
            const int groups = sel == 1 ? J.per : 1;
            for (int r = 0; r < groups; ++r) {
                fprintf(stderr, "[ltpl dbg] fleet k_vel_profile sel %d slot %d:", sel, r);
                for (int k = 0; k + 1 < 16; ++k) {
                    double acc = 0; int cnt = 0;
                    for (int b = r; b < 256; b += groups) { const long long a = v[(size_t)b * DBG_SLOTS + k], c = v[(size_t)b * DBG_SLOTS + k + 1]; if (a > 0 && c > a) { acc += (double)(c - a); ++cnt; } }
                    fprintf(stderr, " %8.0f", cnt ? acc / cnt : 0.0);
                }
                fprintf(stderr, "\n");
            }
            (void)hipMemset(h->d_dbg, 0, v.size() * sizeof(long long));
        }
    }
#endif
    return LTPL_OK;
}

// `fused_post`: paths_post of this tick has not run yet (tape mode) -> first kernel = paths_post + vel_a; `next`: the tick that follows on the
// tape -> its paths_pre runs inside the last state-machine kernel of this tick
static int fleet_launch_vel(ltpl_fleet* f, const FleetTickIn& t, const fleet::FPathsOut* fused_post = nullptr, const FleetTickIn* next = nullptr)
{
    ltpl_handle* h = f->h; const int N = f->D.N; hipStream_t st = h->stream;
    static const double one_row[2] = {0.0, 0.0};
    ltpl_vel_params vp; memset(&vp, 0, sizeof(vp));
    vp.dyn_model_exp = f->pc.dyn_model_exp; vp.drag_coeff = f->pc.drag_coeff; vp.m_veh = f->pc.m_veh; vp.len_veh = h->hostlat.veh_length;
    vp.n_ax_max_machines = t.n_axm; vp.ax_max_machines = one_row;      // (the table itself is read on the device: t.axm)
    vp.follow_control_type = f->pc.follow_control_type; vp.c_p = f->pc.c_p; vp.k_p = f->pc.k_p; vp.k_d = f->pc.k_d; vp.tan_w = f->pc.tan_w; vp.v_max = t.vel_max;
    int rc;
    if (fused_post) hipLaunchKernelGGL(k_fleet_post_vel_a, dim3(N), dim3(64), 0, st, f->args, *fused_post, t.ob, t.vin, f->JA.view());
    else hipLaunchKernelGGL(k_fleet_vel_a, dim3(N), dim3(64), 0, st, f->args, t.ob, t.vin, f->JA.view());
    FLEET_TRY(f, hipGetLastError());
    {   // forward-backward jobs (slots >= 1), one lane per job, on the second stream: 512 long waves for 8 192 planners -- next to the follow jobs
        DevVelParams p;
        if ((rc = make_vel_params(h, &vp, t.axm, &p))) { f->err = h->err; return rc; }
        p.n_axm = t.n_axm_total;           // (the lane kernel stages ALL tables of the call; a job without its own table reads the first p.n_axm... rows of table 0 only when there is one table)
        const unsigned waves = (unsigned)(((size_t)N * (fleet::JOBS_A - 1) + 63) / 64);
        FLEET_TRY(f, hipEventRecord(f->ev_a, st));
        FLEET_TRY(f, hipStreamWaitEvent(f->stream2, f->ev_a, 0));
        hipLaunchKernelGGL(fleet_lanes_kernel_of(fleet_variant(vp, t.multi_axm != 0)), dim3(waves), dim3(64), 16 * (size_t)(t.n_axm_total > 0 ? t.n_axm_total : 1), f->stream2, p, reinterpret_cast<const DevVelJob*>(f->JA.jobs),
                           (const double*)f->JA.pool, reinterpret_cast<const ke_t*>(f->JA.ke), f->JA.ke_rows, f->JA.outp, f->D.RV, N, (int)fleet::JOBS_A,
                           f->JA.out);
        FLEET_TRY(f, hipGetLastError());
        FLEET_TRY(f, hipEventRecord(f->ev_b, f->stream2));
    }
    const bool rows = t.has_gg != 0, multi = t.multi_axm != 0;
    if (rows) f->seen_gg = true;
    const bool rows_mem = rows || f->seen_gg;          // jobs built from the planners' memory (backup plans) may carry rows of an earlier tick
    if (f->JA.ke_follow >= 0) {                                                                  // follow jobs (slot 0) without friction rows: one LANE per job (round 5)
        DevVelParams p;
        if ((rc = make_vel_params(h, &vp, t.axm, &p))) { f->err = h->err; return rc; }
        p.n_axm = t.n_axm_total;
        hipLaunchKernelGGL(fleet_follow_kernel_of(fleet_variant(vp, multi)), dim3((unsigned)((N + 63) / 64)), dim3(64), 16 * (size_t)(t.n_axm_total > 0 ? t.n_axm_total : 1), st,
                           h->lat, p, reinterpret_cast<const DevVelJob*>(f->JA.jobs), (const double*)f->JA.pool, reinterpret_cast<const ke_t*>(f->JA.ke), f->JA.ke_rows,
                           f->JA.ke_follow, f->JA.fp2, f->JA.fp3, f->D.RV, N, (int)fleet::JOBS_A, f->JA.out, f->JA.flags);
        FLEET_TRY(f, hipGetLastError());
    }
    if ((rows || f->JA.ke_follow < 0) && (rc = fleet_launch_vel_jobs(f, vp, t.axm, f->JA, 2, multi, rows))) return rc;   // ... with friction rows (or LTPL_FLEET_FOLLOW_WAVES=1): one wave per job
    if (rows && (rc = fleet_launch_vel_jobs(f, vp, t.axm, f->JA, 3, multi, true))) return rc;    // forward-backward jobs with friction rows
    FLEET_TRY(f, hipStreamWaitEvent(st, f->ev_b, 0));
    hipLaunchKernelGGL(k_fleet_vel_b, dim3(N), dim3(64), 0, st, f->args, f->JA.view(), f->JB.view());
    FLEET_TRY(f, hipGetLastError());
    if ((rc = fleet_launch_vel_jobs(f, vp, t.axm, f->JB, 1, multi, rows_mem))) return rc;
    if (next && !t.any_emerg) hipLaunchKernelGGL(k_fleet_tail_pre<false>, dim3(N), dim3(64), 0, st, f->args, t.vin, f->JB.view(), f->JC.view(), next->ob, f->pin);
    else hipLaunchKernelGGL(k_fleet_vel_c, dim3(N), dim3(64), 0, st, f->args, t.vin, f->JB.view(), f->JC.view());
    FLEET_TRY(f, hipGetLastError());
    if (t.any_emerg) {
        ltpl_vel_params ve = vp; ve.dyn_model_exp = 1.0; ve.drag_coeff = 0.854; ve.m_veh = 1160.0;       // calc_brake_emergency.py:4-6,31-36
        if ((rc = fleet_launch_vel_jobs(f, ve, t.axm, f->JC, 1, false, rows_mem))) return rc;
        if (next) hipLaunchKernelGGL(k_fleet_tail_pre<true>, dim3(N), dim3(64), 0, st, f->args, t.vin, f->JB.view(), f->JC.view(), next->ob, f->pin);
        else hipLaunchKernelGGL(k_fleet_vel_d, dim3(N), dim3(64), 0, st, f->args, t.vin, f->JC.view());
        FLEET_TRY(f, hipGetLastError());
    }
    return LTPL_OK;
}

static int fleet_enter(ltpl_fleet* f)
{
    FLEET_TRY(f, hipSetDevice(f->h->device));
    drop_resident(f->h);            // (the path kernel's LDS plan / parent slabs are shared with the handle's other entry points)
    return LTPL_OK;
}

extern "C" int ltpl_fleet_calc_paths_begin(ltpl_fleet* f, const ltpl_planner_paths_in* in)
try {
    if (!f || !in) return LTPL_ERR_INVALID_ARG;
    int rc = fleet_enter(f);
    if (rc) return rc;
    if ((rc = fleet_pack_inputs(f, &f->cur, in, nullptr, false))) return rc;
    if ((rc = fleet_launch_paths(f, f->cur, true, false))) return rc;
    f->began = true;
    return fleet_check(f);
} LTPL_ABI_CATCH(abi_err_of(f))

extern "C" int ltpl_fleet_calc_paths_finish(ltpl_fleet* f, const int32_t* zone_off, const int32_t* zone_gid)
try {
    if (!f) return LTPL_ERR_INVALID_ARG;
    if (!zone_off || !f->began) { f->err = "fleet: calc_paths_finish without calc_paths_begin"; return LTPL_ERR_INVALID_ARG; }
    f->began = false;
    int rc = fleet_enter(f);
    if (rc) return rc;
    // the zone lists go into their own small arena behind the tick's inputs
    const int N = f->D.N, nz = zone_off[N];
    if (zone_off[0] != 0 || nz < 0) { f->err = "offset arrays must start at 0"; return LTPL_ERR_INVALID_ARG; }
    for (int s = 0; s < N; ++s) if (zone_off[s + 1] < zone_off[s]) { f->err = "fleet: zone offsets must not decrease"; return LTPL_ERR_INVALID_ARG; }
    if (nz > 0 && !zone_gid) { f->err = "fleet: zone node ids missing"; return LTPL_ERR_INVALID_ARG; }
    for (int i = 0; i < nz; ++i) if (zone_gid[i] < 0 || zone_gid[i] >= f->h->lat.V) { f->err = "zone node id out of range"; return LTPL_ERR_INVALID_ARG; }
    int* d_zo = nullptr; int* d_zg = nullptr;
    struct Guard { void* a = nullptr; void* b = nullptr; ~Guard() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); } } g;
    FLEET_TRY(f, hipMalloc(reinterpret_cast<void**>(&d_zo), 4 * (size_t)(N + 1))); g.a = d_zo;
    FLEET_TRY(f, hipMalloc(reinterpret_cast<void**>(&d_zg), 4 * (size_t)(nz + 1))); g.b = d_zg;
    FLEET_TRY(f, hipMemcpy(d_zo, zone_off, 4 * (size_t)(N + 1), hipMemcpyHostToDevice));
    if (nz) FLEET_TRY(f, hipMemcpy(d_zg, zone_gid, 4 * (size_t)nz, hipMemcpyHostToDevice));
    FleetTickIn t = f->cur; t.zone_off = d_zo; t.zone_gid = d_zg;
    if ((rc = fleet_launch_paths(f, t, false, true))) return rc;
    return fleet_check(f);
} LTPL_ABI_CATCH(abi_err_of(f))

extern "C" int ltpl_fleet_calc_paths(ltpl_fleet* f, const ltpl_planner_paths_in* in)
try {
    if (!f || !in) return LTPL_ERR_INVALID_ARG;
    int rc = fleet_enter(f);
    if (rc) return rc;
    if ((rc = fleet_pack_inputs(f, &f->cur, in, nullptr, true))) return rc;
    if ((rc = fleet_launch_paths(f, f->cur, true, true))) return rc;
    return fleet_check(f);
} LTPL_ABI_CATCH(abi_err_of(f))

extern "C" int ltpl_fleet_get_ref_idx(ltpl_fleet* f, const double* px, const double* py)
try {
    if (!f || !px || !py) return LTPL_ERR_INVALID_ARG;
    int rc = fleet_enter(f);
    if (rc) return rc;
    const size_t n = (size_t)f->D.N;
    double* d = nullptr;
    struct Guard { void* a = nullptr; ~Guard() { if (a) (void)hipFree(a); } } g;
    FLEET_TRY(f, hipMalloc(reinterpret_cast<void**>(&d), 16 * n)); g.a = d;
    FLEET_TRY(f, hipMemcpy(d, px, 8 * n, hipMemcpyHostToDevice));
    FLEET_TRY(f, hipMemcpy(d + n, py, 8 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fleet_ref_idx, dim3(f->D.N), dim3(64), 0, f->h->stream, f->args, (const double*)d, (const double*)(d + n));
    FLEET_TRY(f, hipGetLastError());
    return fleet_check(f);
} LTPL_ABI_CATCH(abi_err_of(f))

extern "C" int ltpl_fleet_calc_vel_profile(ltpl_fleet* f, const ltpl_planner_vel_in* in)
try {
    if (!f || !in) return LTPL_ERR_INVALID_ARG;
    int rc = fleet_enter(f);
    if (rc) return rc;
    if (!f->cur.has_paths) { f->err = "fleet: calc_vel_profile before calc_paths"; return LTPL_ERR_INVALID_ARG; }
    // the objects of the tick stay where calc_paths put them: the velocity inputs go into a second arena
    FleetTickIn& v = f->curv;
    if ((rc = fleet_pack_inputs(f, &v, nullptr, in, false))) return rc;
    FleetTickIn t = f->cur; t.vin = v.vin; t.axm = v.axm; t.n_axm = v.n_axm; t.vel_max = v.vel_max; t.any_emerg = v.any_emerg;
    t.n_axm_total = v.n_axm_total; t.multi_axm = v.multi_axm; t.has_gg = v.has_gg;
    if ((rc = fleet_launch_vel(f, t))) return rc;
    return fleet_check(f);
} LTPL_ABI_CATCH(abi_err_of(f))

extern "C" int ltpl_fleet_digest(ltpl_fleet* f, double* out, int32_t cap_doubles_per_planner)
try {
    if (!f || !out) return LTPL_ERR_INVALID_ARG;
    if (cap_doubles_per_planner != LTPL_FLEET_DIGEST) { f->err = "fleet: digest record size mismatch"; return LTPL_ERR_INVALID_ARG; }
    FLEET_TRY(f, hipSetDevice(f->h->device));
    const size_t n = (size_t)f->D.N * LTPL_FLEET_DIGEST;
    if (!f->d_digest) {                               // once per fleet (freed with the fleet's other allocations): no hipFree -- an implicit device synchronisation -- per call
        const int rc = fleet_alloc(f, n, &f->d_digest, false);
        if (rc) return rc;
    }
    double* d = f->d_digest;
    hipLaunchKernelGGL(k_fleet_digest, dim3(f->D.N), dim3(64), 0, f->h->stream, f->args, d);
    FLEET_TRY(f, hipGetLastError());
    FLEET_TRY(f, hipMemcpyAsync(out, d, sizeof(double) * n, hipMemcpyDeviceToHost, f->h->stream));
    FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(f))

static int fleet_fetch(ltpl_fleet* f, int p)
{
    if (p < 0 || p >= f->D.N) { f->err = "fleet: planner index out of range"; return LTPL_ERR_INVALID_ARG; }
    FLEET_TRY(f, hipSetDevice(f->h->device));
    FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
    FLEET_TRY(f, hipMemcpy(f->image.data(), f->d_state + f->D.stride * (size_t)p, f->D.stride, hipMemcpyDeviceToHost));
    return LTPL_OK;
}
extern "C" int ltpl_fleet_get_paths(ltpl_fleet* f, int32_t p, ltpl_planner_paths_view* v)
try {
    if (!f || !v) return LTPL_ERR_INVALID_ARG;
    const int rc = fleet_fetch(f, p);
    return rc ? rc : fleet::paths_view(f->D, f->image.data(), v);
} LTPL_ABI_CATCH(abi_err_of(f))
extern "C" int ltpl_fleet_get_trajectories(ltpl_fleet* f, int32_t p, ltpl_planner_traj_view* v)
try {
    if (!f || !v) return LTPL_ERR_INVALID_ARG;
    const int rc = fleet_fetch(f, p);
    return rc ? rc : fleet::traj_view(f->D, f->image.data(), v);
} LTPL_ABI_CATCH(abi_err_of(f))

// ---- tape: pre-uploaded inputs of many ticks, replayed without host synchronisation ------------------------------------------------
extern "C" int ltpl_fleet_tape_clear(ltpl_fleet* f)
try {
    if (!f) return LTPL_ERR_INVALID_ARG;
    FLEET_TRY(f, hipSetDevice(f->h->device));
    FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
    for (FleetTickIn& t : f->tape) if (t.d_buf) (void)hipFree(t.d_buf);
    f->tape.clear();
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(f))

extern "C" int ltpl_fleet_tape_append(ltpl_fleet* f, const ltpl_planner_paths_in* pin, const ltpl_planner_vel_in* vin)
try {
    if (!f || !pin || !vin) return LTPL_ERR_INVALID_ARG;
    FLEET_TRY(f, hipSetDevice(f->h->device));
    FleetTickIn t;
    const int rc = fleet_pack_inputs(f, &t, pin, vin, true);
    if (rc) { if (t.d_buf) (void)hipFree(t.d_buf); return rc; }
    f->tape.push_back(t);
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(f))

extern "C" int ltpl_fleet_tape_run(ltpl_fleet* f, int32_t first, int32_t count, float* ms_total)
try {
    if (!f) return LTPL_ERR_INVALID_ARG;
    if (first < 0 || count < 1 || (size_t)first + (size_t)count > f->tape.size()) { f->err = "fleet: tape range out of bounds"; return LTPL_ERR_INVALID_ARG; }
    int rc = fleet_enter(f);
    if (rc) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Guard { hipEvent_t* a; hipEvent_t* b; ~Guard() { if (*a) (void)hipEventDestroy(*a); if (*b) (void)hipEventDestroy(*b); } } g{&e0, &e1};
    FLEET_TRY(f, hipEventCreate(&e0)); FLEET_TRY(f, hipEventCreate(&e1));
    FLEET_TRY(f, hipStreamSynchronize(f->h->stream));
    FLEET_TRY(f, hipEventRecord(e0, f->h->stream));
    const bool fuse = f->tape_fuse;
    for (int i = first; i < first + count; ++i) {
        const FleetTickIn& t = f->tape[(size_t)i];
        if (!fuse) {
            if ((rc = fleet_launch_paths(f, t, true, true))) return rc;
            if ((rc = fleet_launch_vel(f, t))) return rc;
            continue;
        }
        // fused stages: paths_pre of tick i ran inside the last kernel of tick i - 1 (except for the first tick of the run)
        fleet::FPathsOut po{};
        if ((rc = fleet_launch_paths(f, t, i == first, true, false, &po))) return rc;
        if ((rc = fleet_launch_vel(f, t, &po, i + 1 < first + count ? &f->tape[(size_t)i + 1] : nullptr))) return rc;
    }
    FLEET_TRY(f, hipEventRecord(e1, f->h->stream));
    FLEET_TRY(f, hipEventSynchronize(e1));
    if (ms_total) FLEET_TRY(f, hipEventElapsedTime(ms_total, e0, e1));
    f->cur.has_paths = false;           // (the objects of the last tick live in the tape: a per-call calc_vel_profile needs its own calc_paths first)
    return fleet_check(f);
} LTPL_ABI_CATCH(abi_err_of(f))
