/*
 * ltpl_hip.h -- C ABI of libltpl_hip.so, the MI355X (gfx950) backend for the online hot path of graph_ltpl.
 *
 * The reference (TUMFTM/GraphBasedLocalTrajectoryPlanner) is pure Python and has no FFI; the natural seams are two
 * Python call sites that it resolves by fully qualified attribute lookup on every tick:
 *
 *   seam (1)  graph_ltpl/online_graph/src/main_online_path_gen.py:11-21,333-334   main_online_path_gen(...)
 *             called from graph_ltpl/online_graph/src/OnlineTrajectoryHandler.py:416-427
 *   seam (2)  graph_ltpl/online_graph/src/VpForwardBackward.py:11-255             class VpForwardBackward
 *             constructed at OnlineTrajectoryHandler.py:137-144, called at :676, :747-752, :789-798, :872-878, :967-972
 *
 * Every entry point below names the reference interface it replaces. All functions are extern "C", take plain
 * pointers / sizes (C-contiguous float64 / int32 host buffers owned by the caller) and return an int status
 * (LTPL_OK == 0). "No path found" is NOT an error (valid == 0). No C++ exception crosses the ABI. A handle owns its
 * device memory, pinned staging and one HIP stream; calls on one handle are synchronous on return and must not be
 * issued concurrently; different handles may be used from different threads / processes.
 *
 * Environment switches (all read ONCE, at ltpl_create / ltpl_planner_create). Every switch of the RELEASE library selects among
 * code paths that produce identical results (the GPU test-suite runs them); none skips work:
 *   LTPL_NO_FIXED_PLAN=1        batch kernel with the runtime LDS plan instead of a compile-time plan class (any lattice)
 *   LTPL_FORCE_LONG_HORIZON=1   parent tables in global memory (the mode lattices with very long planning ranges get automatically)
 *   LTPL_BATCH_NW=4             four-wave teams also for batches; LTPL_NW1_MIN_SCEN=<n>: smallest batch that uses one-wave teams (64)
 *   LTPL_FORCE_FUSED=1          fused k_tick also for batches; LTPL_NO_OVERLAP=1: resident batches on one stream (no pipelining)
 *   LTPL_NO_SCEN_ORDER=1        batches of >= 2048 scenarios planned in the caller's order instead of sorted by start layer (identical results)
 *   LTPL_FOLLOW_EMIT_MIN_SCEN=<n>  smallest pipeline batch whose follow jobs are finished by the lane kernel instead of k_vel_final (8192)
 *   LTPL_PIPELINE_MIN_SCEN=<n>  smallest tick batch that runs the one-wave pipeline instead of the fused tick kernel (default: more than two
 *                               fused workgroups per compute unit -- 513 on the MI355X; rounds 1-5: 64)
 *   LTPL_FINAL_Y=<n>            row-chunk blocks per tile of the final velocity kernel (8)
 *   LTPL_ZC_OUT=0 / LTPL_ZC_IN=1   zero-copy outputs (default on) / inputs (default off) of calls with <= 8 scenarios
 *   LTPL_POLL=1 (+ LTPL_POLL_SYNC_EVERY, LTPL_POLL_QUERY)   completion of small calls through a polled word instead of a stream sync
 *   LTPL_NO_SELFTEST=1          skip the create-time self-test (one-wave vs four-wave kernel on probe scenarios)
 *   LTPL_HOST_PROF=1            host-side timing table of the entry points on stderr at exit
 *   LTPL_FLEET_NO_FUSE=1        fleet tape runs with one kernel per stage instead of the fused stage kernels (read by ltpl_fleet_create)
 *   LTPL_FLEET_FOLLOW_WAVES=1/0 fleet follow jobs one WAVE per job / one LANE per job (default: lanes from 12 288 planners on; ltpl_fleet_create)
 *   LTPL_NO_LAYER_GRID=1        closest reference-line layer of an obstacle position by the scan over all layers, no create-time grid
 *   LTPL_PERSISTENT_TICK=1      ltpl_create behaves like ltpl_create_ex(LTPL_CREATE_PERSISTENT_TICK); LTPL_PERSIST_IDLE_MS=<ms>: idle limit (250)
 *   LTPL_TICK_GRAPH=1           the single fused tick (copy in -> kernel [-> copy out]) as ONE hipGraph launch (measured: slower; off)
 *   LTPL_VEL_CUS=<n>            velocity streams of the resident-batch pipeline confined to n compute units by a CU mask (measured: the
 *                               velocity chain becomes the bottleneck below ~128 CUs, no gain above; off)
 * Timing / fault-injection switches (LTPL_ABLATE, LTPL_EXP_SKIP, LTPL_LDS_POISON, LTPL_SCRATCH_POISON, LTPL_DEBUG_TIMING,
 * LTPL_DEBUG_OCC) skip work, overwrite memory or instrument kernels; they are compiled into the EXPERIMENT build only
 * (-DLTPL_EXPERIMENT -> libltpl_hip_exp.so, used by tools/ and one fault-injection test) and do not exist in libltpl_hip.so.
 */
#ifndef LTPL_HIP_H
#define LTPL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LTPL_ABI_VERSION 9        /* v7 (round 5, additive): ltpl_paths_kernel_symbol, ltpl_layer_grid, ltpl_fleet_digest; v8 (additive): ltpl_assembly_records;
                                     v9 (round 6, additive): ltpl_create_ex, ltpl_tick_persistent_stop / _stats */

/* status codes */
#define LTPL_OK               0
#define LTPL_ERR_INVALID_ARG  1
#define LTPL_ERR_NO_DEVICE    2
#define LTPL_ERR_HIP          3
#define LTPL_ERR_CAPACITY     4
#define LTPL_ERR_UNSUPPORTED  5
#define LTPL_ERR_EXCEPTION    6    /* a C++ exception (out of host memory, ...) was caught at the ABI; message in ltpl_last_error */

/* action primitives: ACTION_ID_MAP, OnlineTrajectoryHandler.py:14-17 */
#define LTPL_ACT_STRAIGHT 0
#define LTPL_ACT_FOLLOW   1
#define LTPL_ACT_LEFT     2
#define LTPL_ACT_RIGHT    3
#define LTPL_ACT_NONE    (-1)
#define LTPL_ACT_EMERGENCY 4   /* key 'emergency' of the exported trajectory set (OnlineTrajectoryHandler.py:1028-1034) */

#define LTPL_MAX_ACTIONS     3   /* at most 3 primitives are offered per tick (main_online_path_gen.py:128-174) */
#define LTPL_MAX_LAST_NODES  8   /* nodes of the previous solution used for the cost discount (w_last_edges)      */

/* per-scenario flag bits of ltpl_paths_in.flags */
#define LTPL_FLAG_ACTION_SETS     1   /* action_sets=True                    main_online_path_gen.py:15          */
#define LTPL_FLAG_OBJ_IN_CONST    2   /* obj_in_const_path                   main_online_path_gen.py:77,118-122  */
#define LTPL_FLAG_OBJ_BESIDES     4   /* object_besides_const_path           main_online_path_gen.py:78,104-105  */
#define LTPL_FLAG_HAS_PSI_S       8   /* const_path_seg is not None -> psi_s main_online_path_gen.py:300-303     */

typedef struct ltpl_handle ltpl_handle;

/* ------------------------------------------------------------------------------------------------------------------
 * Offline lattice, struct-of-arrays (what GraphBase holds in igraph attributes: GraphBase.py:93-119,163-194,409-439).
 * Node global id = layer_node_off[layer] + node. Edges are stored CSC by destination node, in-edges sorted by source
 * node id; an edge always connects layer l to layer (l+1) mod num_layers (gen_edges.py:52-61).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t num_layers;
    int32_t num_nodes;
    int32_t num_edges;
    int32_t num_samples;
    int32_t num_glob_rl;            /* rows of glob_rl                                                            */
    int32_t closed;                 /* GraphBase.closed; 0 = open track (planning range clamped to the last layer)   */
    int32_t plan_horizon_mode;      /* 0 = 'distance', 1 = 'layers'        gen_local_node_template.py:104-133     */
    int32_t reserved0;
    double  min_plan_horizon;
    double  lat_resolution;
    double  lat_offset;
    double  veh_width;
    double  veh_length;
    double  sampled_resolution;
    double  vel_decrease_lat;
    /* per layer [num_layers] */
    const int32_t* layer_node_off;  /* [num_layers + 1]                                                           */
    const int32_t* raceline_index;
    const double*  s_raceline;
    const double*  refline_x;
    const double*  refline_y;
    const double*  vel_raceline;
    /* per node [num_nodes] */
    const double*  node_x;
    const double*  node_y;
    const double*  vgoal_cost;      /* cost of the virtual goal edge of that node              GraphBase.py:188   */
    /* per edge [num_edges] */
    const int32_t* in_ptr;          /* [num_nodes + 1]                                                            */
    const int32_t* edge_src;        /* source node index inside the previous layer                                */
    const double*  edge_cost;       /* offline_cost                                                               */
    const double*  edge_len;        /* spline_length                                                              */
    const int32_t* samp_ptr;        /* [num_edges + 1]                                                            */
    /* per sample [num_samples]: columns 0,1,2,4 of spline_param (GraphBase.py:425-436); kappa (col 3) is never read
     * online because main_online_path_gen.py:318-322 overwrites it                                               */
    const double*  samp_x;
    const double*  samp_y;
    const double*  samp_psi;
    const double*  samp_len;
    /* fine global race line, row-major [num_glob_rl][5] = s, x, y, kappa, vel    (GraphBase.glob_rl)             */
    const double*  glob_rl;
    /* track bounds per layer (ObjectListInterface.set_track_data, ObjectListInterface.py:49-73; fed from GraphBase at
     * Graph_LTPL.py:232-235): bound1 = refline + normvec * width_right, bound2 = refline - normvec * width_left       */
    const double*  normvec_x;       /* [num_layers]                                                               */
    const double*  normvec_y;
    const double*  width_right;
    const double*  width_left;
    /* ABI v3, only read by the planner entry points (ltpl_planner_*); may be NULL otherwise:
     * race line point per layer (GraphBase.raceline, main_online_path_gen.py:86-101) and heading per node
     * (vertex attribute psi, GraphBase.py:163-168; OnlineTrajectoryHandler.py:232-246)                              */
    const double*  raceline_x;      /* [num_layers]                                                               */
    const double*  raceline_y;
    const double*  node_psi;        /* [num_nodes]                                                                */
} ltpl_lattice_desc;

typedef struct {
    int32_t max_path_nodes;         /* capacity needed per path: nodes                                            */
    int32_t max_path_pts;           /* capacity needed per path: samples                                          */
    int32_t max_horizon_edges;
    int32_t device;
    int32_t num_cus;
    int32_t lds_bytes_paths;        /* dynamic LDS of the path kernel                                             */
} ltpl_caps;

/* ------------------------------------------------------------------------------------------------------------------
 * seam (1): batched main_online_path_gen. One "scenario" = one call of the reference function.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_scen;
    int32_t n_w_last;               /* len(w_last_edges)                                                          */
    const double*  w_last_edges;    /* [n_w_last]  online cost factors     gen_local_node_template.py:155-162     */
    const int32_t* start_layer;     /* [n_scen]                                                                   */
    const int32_t* start_node;      /* [n_scen]                                                                   */
    const int32_t* flags;           /* [n_scen]  LTPL_FLAG_*                                                      */
    const int32_t* last_action;     /* [n_scen]  LTPL_ACT_* of last_action_id or LTPL_ACT_NONE                    */
    const int32_t* const_closest;   /* [n_scen]  object index chosen by the constant-segment test
                                                 (main_online_path_gen.py:97-115) or -1                           */
    const double*  psi_s;           /* [n_scen]  const_path_seg[-1, 2] when LTPL_FLAG_HAS_PSI_S                   */
    /* vehicles: vehicle k of scenario s is v = veh_off[s] + k; its positions are pos_off[v] .. pos_off[v+1]-1,
     * the first one being VehObject.get_pos(), the others VehObject.get_prediction() rows
     * (gen_local_node_template.py:169-189)                                                                       */
    const int32_t* veh_off;         /* [n_scen + 1]                                                               */
    const int32_t* pos_off;         /* [n_veh_total + 1]                                                          */
    const double*  veh_radius;      /* [n_veh_total]                                                              */
    const double*  pos_x;           /* [n_pos_total]                                                              */
    const double*  pos_y;           /* [n_pos_total]                                                              */
    /* zone-blocked nodes ("overtaking_zones" filter, gen_local_node_template.py:96): global node ids             */
    const int32_t* zone_off;        /* [n_scen + 1]                                                               */
    const int32_t* zone_gid;        /* [n_zone_total]                                                             */
    /* previous solution (last_solution_nodes): n_last[s] <= LTPL_MAX_LAST_NODES leading nodes                    */
    const int32_t* n_last;          /* [n_scen]                                                                   */
    const int32_t* last_layer;      /* [n_scen * LTPL_MAX_LAST_NODES]                                             */
    const int32_t* last_node;       /* [n_scen * LTPL_MAX_LAST_NODES]                                             */
} ltpl_paths_in;

typedef struct {
    int32_t cap_nodes;              /* caller-chosen capacities, >= ltpl_caps.max_path_nodes / max_path_pts       */
    int32_t cap_pts;
    /* per scenario */
    int32_t* end_layer;             /* [n_scen]                                                                   */
    int32_t* closest_obj_index;     /* [n_scen]     -1 = None                                                     */
    int32_t* closest_obj_node;      /* [n_scen * 2] layer, node; -1 = None                                        */
    int32_t* n_actions;             /* [n_scen]     number of action slots used (template length)                 */
    /* per scenario and action slot a < LTPL_MAX_ACTIONS, index s * LTPL_MAX_ACTIONS + a; slots are in the order the
     * reference inserts its dict keys (follow / straight first)                                                  */
    int32_t* action_id;             /* final name (after the follow -> straight rename, :232-237) or NONE         */
    int32_t* valid;                 /* 1 = key present in the reference's dicts                                   */
    int32_t* reduced;               /* action_set_red_len                                                         */
    int32_t* goal_layer;            /* layer the path ends in                                                     */
    int32_t* n_nodes;               /* entries of nodes / node_idx / coeff behind n_nodes and rows of path_param (and of vx / ax of the
                                     * tick outputs) behind n_pts are UNSPECIFIED padding, as are slots a >= n_actions and invalid slots */
    int32_t* n_pts;
    int32_t* n_ties;                /* exact cost ties met while picking predecessors along this sweep            */
    int32_t* nodes;                 /* [.. * cap_nodes]      node index per layer, layer i = (start + i) mod L    */
    int32_t* node_idx;              /* [.. * cap_nodes]      row of every node in path_param                      */
    double*  coeff;                 /* [.. * cap_nodes * 8]  (n_nodes-1) rows [x a0..a3, y a0..a3]                */
    double*  path_param;            /* [.. * cap_pts * 5]    n_pts rows [x, y, psi, kappa, el_length]             */
} ltpl_paths_out;

/* ------------------------------------------------------------------------------------------------------------------
 * seam (2): the arithmetic behind class VpForwardBackward.
 * ------------------------------------------------------------------------------------------------------------------ */
#define LTPL_VEL_FB      0   /* VpForwardBackward.calc_vel_profile  :194-227 -> tph.calc_vel_profile(closed=False) */
#define LTPL_VEL_BRAKE   1   /* tph.calc_vel_profile_brake behind check_brake_prefix :115-122, calc_vel_brake_em  */
#define LTPL_VEL_FOLLOW  2   /* VpForwardBackward.calc_vel_profile_follow :141-192 -> calc_vel_profile_follow.py  */
#define LTPL_VEL_FOLLOW_CONTROLLED 3   /* the same without the final intersection with the unconstrained profile
                                          (calc_vel_profile_follow.py:78-294 only): a caller that wants the two independent halves of
                                          the follow mode in parallel submits this job + an LTPL_VEL_FB job without v_end and takes
                                          the element-wise minimum (:297-310) itself                                          */

typedef struct {
    double  dyn_model_exp;          /* VpForwardBackward.__init__ :22-29                                          */
    double  drag_coeff;
    double  m_veh;
    double  len_veh;
    double  v_max;                  /* update_dyn_parameters :65-84                                               */
    int32_t n_ax_max_machines;
    int32_t follow_control_type;    /* 0 = 'PD', 1 = 'PDtan'          calc_vel_profile_follow.py:65-75            */
    const double* ax_max_machines;  /* [n_ax_max_machines * 2] rows [v, ax]                                       */
    double  c_p, k_p, k_d, tan_w;   /* follow controller parameters   params/ltpl_config_online.ini:41-50         */
} ltpl_vel_params;

typedef struct {
    int32_t mode;                   /* LTPL_VEL_*                                                                 */
    int32_t n;                      /* kappa.size                                                                 */
    int32_t n_el;                   /* el_lengths.size: n-1 (FB, BRAKE) or n (FOLLOW, OTH.py:791 hands over n)    */
    int32_t has_v_end;
    const double* kappa;            /* [n]                                                                        */
    const double* el_lengths;       /* [n_el]                                                                     */
    const double* loc_gg;           /* [n * 2] rows [ax_max, ay_max], gg_scale already applied                    */
    double  v_start;
    double  v_end;
    /* FOLLOW only (calc_vel_profile_follow.py:78-95) */
    double  v_ego, v_obj, safety_d, obj_dist, obj_x, obj_y;
} ltpl_vel_job;

typedef struct {
    double*  vx;                    /* [n] output profile                                                         */
    int32_t  too_close;             /* FOLLOW: calc_vel_profile_follow.py:146-149                                 */
    int32_t  vel_bound;             /* FOLLOW: vel_bound_fulfilled                                                */
} ltpl_vel_result;

/* ------------------------------------------------------------------------------------------------------------------
 * fused tick: seam (1) followed by the per-primitive velocity stage of OnlineTrajectoryHandler.calc_vel_profile
 * (OTH.py:688-941) on the freshly planned paths (no constant prefix: cut_index_pos = 0, vel_course empty).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    const ltpl_vel_params* params;
    double  gg_ax, gg_ay;           /* constant local gg (ax, ay) times gg_scale   OTH.py:651-666                 */
    double  gg_brake_scale;         /* old_gg_scale / gg_scale used by the brake prefix  VpForwardBackward.py:114 */
    double  safety_d;
    double  v_max_offset;           /* ACTIONSET.v_max_offset                      OTH.py:102,907                 */
    const double* vel_plan;         /* [n_scen]                                                                   */
    const double* vel_est;          /* [n_scen]                                                                   */
    const double* pos_est_x;        /* [n_scen]                                                                   */
    const double* pos_est_y;        /* [n_scen]                                                                   */
    const double* veh_vel;          /* [n_veh_total] VehObject.get_vel()                                          */
} ltpl_tick_vel_in;

typedef struct {
    double*  vx;                    /* [n_scen * LTPL_MAX_ACTIONS * cap_pts]                                      */
    double*  ax;                    /* [n_scen * LTPL_MAX_ACTIONS * cap_pts]                                      */
    int32_t* vel_bound;             /* [n_scen * LTPL_MAX_ACTIONS]                                                */
    int32_t* too_close;             /* [n_scen * LTPL_MAX_ACTIONS]                                                */
} ltpl_tick_vel_out;

/* ------------------------------------------------------------------------------------------------------------------
 * object ingestion: the arithmetic of ObjectListInterface.process_object_list
 * (ObjectListInterface.py:75-153) for objects of type "physical": on-track test check_inside_bounds
 * (check_inside_bounds.py:7-59), constant-velocity prediction over dt (:117-127, dt = 0.2 s) and radius = length / 2 (:133).
 * One flat list of objects (the caller concatenates the objects of all scenarios).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_obj;
    int32_t reserved0;
    double  dt;                     /* prediction horizon, 0.2 in the reference                                  */
    const double* x;                /* [n_obj] object_el['X']                                                     */
    const double* y;                /* [n_obj] object_el['Y']                                                     */
    const double* theta;            /* [n_obj] heading, 0 = north                                                 */
    const double* v;                /* [n_obj]                                                                    */
    const double* length;           /* [n_obj]                                                                    */
} ltpl_objects_in;

typedef struct {
    int32_t* on_track;              /* [n_obj] 1 = inside the track bounds                                        */
    double*  pred_x;                /* [n_obj] X - sin(theta) v dt                                                */
    double*  pred_y;                /* [n_obj] Y + cos(theta) v dt                                                */
    double*  radius;                /* [n_obj] length / 2                                                         */
} ltpl_objects_out;

/* --- lifecycle ---------------------------------------------------------------------------------------------------- */
/* Uploads the lattice to HBM once (replaces the pickled GraphBase handed to OnlineTrajectoryHandler,
 * Graph_LTPL.py:202-229). device < 0 selects the current device. */
int ltpl_create(const ltpl_lattice_desc* lattice, int device, ltpl_handle** out_handle);
/* v9: the same with flags. LTPL_CREATE_PERSISTENT_TICK: single-scenario ltpl_tick_batch calls are served by a RESIDENT kernel (one
 * workgroup, started at the first such call) that receives every tick through a mailbox in page-locked memory instead of being launched
 * per tick -- the latency form for the reference's own use, one car and one tick at a time inside its 0.1 s budget
 * (OnlineTrajectoryHandler.py:353-366). Results are bit-identical to the launched form (same device code). Engaged for lattices whose
 * single-tick kernel has a compile-time LDS plan (ltpl_persistent_stats.enabled tells); otherwise the call behaves like ltpl_create.
 * A resident kernel never completes, so device-wide synchronisations of the process (hipDeviceSynchronize, hipFree) wait for it: every
 * other entry point of the handle stops it first, ltpl_tick_persistent_stop does so on request, and it leaves by itself after
 * LTPL_PERSIST_IDLE_MS (250) ms without a tick -- the next tick starts it again. Environment: LTPL_PERSISTENT_TICK=1 sets the flag for
 * ltpl_create. */
#define LTPL_CREATE_PERSISTENT_TICK 1u
int ltpl_create_ex(const ltpl_lattice_desc* lattice, int device, uint32_t flags, ltpl_handle** out_handle);
typedef struct {
    int32_t enabled;                /* 1: single ticks of this handle go through the resident kernel                         */
    int32_t resident;               /* 1: the kernel is resident right now                                                    */
    int64_t ticks;                  /* ticks served through the mailbox                                                       */
    int64_t launches;               /* kernel starts (first tick, after idling out, after another entry point stopped it)     */
    double  device_us_mean;         /* device-side time per tick: "sequence number seen" .. "outputs fenced out", mean / last */
    double  device_us_last;
    double  idle_ms;                /* idle limit in force                                                                    */
} ltpl_persistent_stats;
int ltpl_tick_persistent_stop(ltpl_handle* handle);       /* the resident kernel leaves now (no-op when none is resident)           */
int ltpl_tick_persistent_stats(ltpl_handle* handle, ltpl_persistent_stats* stats);
int ltpl_destroy(ltpl_handle* handle);
int ltpl_get_caps(const ltpl_handle* handle, ltpl_caps* caps);
const char* ltpl_last_error(const ltpl_handle* handle);   /* NULL handle -> last create() error */
int ltpl_version(void);
/* diagnostics: symbol-name prefix (Itanium mangling) of the path kernel this handle launches for batches (team_waves = 1) or single
 * ticks (team_waves = 4) -- the LDS plan class is chosen per lattice at ltpl_create. Profiles under profiles/ carry a digest of that
 * kernel's instruction stream; bench.py uses this to check that a committed counter pass describes the library it runs. */
const char* ltpl_paths_kernel_symbol(const ltpl_handle* handle, int32_t team_waves);

/* --- seam (1): main_online_path_gen.py:11 ------------------------------------------------------------------------- */
int ltpl_plan_paths(ltpl_handle* handle, const ltpl_paths_in* in, ltpl_paths_out* out);

/* --- diagnostics of seam (1): the same call with the obstacle x edge mask exported (GraphBase.get_intersec_edges_in_range,
 *     GraphBase.py:567-646; the set of edges gen_local_node_template.py:164-203 deletes from the "default" filter) exactly as the
 *     path kernel computed it in its LDS bitmap: blocked[s * num_edges + e] = 1 if edge e (global id) is blocked in scenario s.
 *     The kernel tests every edge of an obstacle's 3-layer window that lies in the planning range, whether or not its end nodes
 *     are removed by the zone filter (the reference only tests edges of the active "planning_range" filter; edges at removed
 *     nodes cannot be used either way). team_waves: 0 = the form ltpl_plan_paths would choose for n_scen, 1 = one-wave batch
 *     kernel, 4 = four-wave latency kernel. tests/test_edge_mask.py compares the bitmap with the oracle's bit for bit. ------- */
int ltpl_plan_paths_mask(ltpl_handle* handle, const ltpl_paths_in* in, ltpl_paths_out* out, int32_t team_waves, uint8_t* blocked);

/* --- constant-segment test in front of seam (1): main_online_path_gen.py:76-122 (host side, O(#objects) projections on the
 *     race line). seg = rows [x, y, psi, kappa, el] of const_path_seg (n_rows may be 0), pos_est = 2 doubles or NULL;
 *     flags_out receives LTPL_FLAG_OBJ_IN_CONST | LTPL_FLAG_OBJ_BESIDES bits, closest_out the object index or -1 ---------- */
int ltpl_const_segment_test(const ltpl_handle* handle, const double* seg, int32_t n_rows, const double* pos_est, int32_t n_veh,
                            const double* veh_x, const double* veh_y, const double* veh_radius, int32_t* flags_out,
                            int32_t* closest_out);

/* --- global s coordinate of a position on the race line: get_s_coord(ref_line=raceline, s_array=s_raceline, closed=True)
 *     (get_s_coord.py:8-99; call sites Graph_LTPL.py:436-440 for the log row, main_online_path_gen.py:86-101) ------------- */
int ltpl_raceline_s(const ltpl_handle* handle, double x, double y, double* s_out);

/* --- diagnostics (host only, no device needed): the per-edge capsule table ltpl_create derives for the obstacle mask
 *     (GraphBase.get_intersec_edges_in_range, GraphBase.py:567-646). The path kernel decides "no sample of the edge is within
 *     the obstacle's threshold" / "some sample is" from it without reading the samples whenever either is certain, and runs the
 *     reference's exact sample test otherwise. capsules_out: 8 floats per edge (Ax, Ay, ABx, ABy, 1 / |AB|^2, dev, (gap / 2)^2,
 *     packed sample range); slack_out: the fp32 rounding bound added to both decisions. tests/test_capsule_cull.py checks the
 *     conservativeness of the decisions against the exact test on the real lattices. ------------------------------------- */
int ltpl_edge_capsules(int32_t n_edges, const int32_t* samp_ptr, const double* samp_x, const double* samp_y, int32_t n_samples,
                       float* capsules_out, float* slack_out);

/* --- diagnostics (host only, no device needed): the closest-layer grid ltpl_create derives for the path kernel's first phase (closest
 *     reference-line layer of an obstacle position = np.argmin over all layers, get_intersec_edges.py:40-51). Per grid cell the at most two
 *     intervals of layers that can be closest to any point of the cell; the kernel evaluates the reference's exact distances on those and on
 *     all layers for positions outside the grid / in cells marked "full scan". origin_cell: x0, y0, 1 / cell size; dims: nx, ny; cells (may
 *     be NULL to query the dimensions first): 4 ints per cell, row-major in y: first layer and length of interval 1 and 2, length -1 = full
 *     scan. tests/test_layer_grid.py checks against the brute-force argmin that the true answer is always among the candidates. --------- */
int ltpl_layer_grid(int32_t n_layers, const double* ref_x, const double* ref_y, double* origin_cell, int32_t* dims, int32_t* cells,
                    int32_t cap_cells);

/* --- diagnostics (host only, no device needed): the per-node / per-edge records ltpl_create derives for the path assembly
 *     (main_online_path_gen.py:260-328: the nodes of a path -> its edges -> the spline samples' coordinates and end headings). node_rec_out:
 *     4 ints per node = first in-edge (CSC id) + the source nodes of its first 12 in-edges as bytes (0xff = none); edge_rec_out: 10 doubles
 *     per edge = first sample | #samples << 32 (bit pattern), edge length, x, y of the first and of the last sample, sin, cos of the first
 *     and of the last sample's heading. tests/test_assembly_records.py checks them against the lattice arrays they replace. ----------- */
int ltpl_assembly_records(int32_t n_nodes, int32_t n_edges, const int32_t* in_ptr, const int32_t* edge_src, const double* edge_len,
                          const int32_t* samp_ptr, const double* samp_x, const double* samp_y, const double* samp_psi,
                          int32_t* node_rec_out, double* edge_rec_out);

/* --- object ingestion: ObjectListInterface.py:75-153, check_inside_bounds.py:7-59 --------------------------------- */
int ltpl_process_objects(ltpl_handle* handle, const ltpl_objects_in* in, ltpl_objects_out* out);

/* --- seam (2): VpForwardBackward.py:86,141,194,229 ---------------------------------------------------------------- */
int ltpl_vel_profile(ltpl_handle* handle, const ltpl_vel_params* params, int n_jobs, const ltpl_vel_job* jobs,
                     ltpl_vel_result* results);

/* --- fused tick (throughput path): Graph_LTPL.calc_paths + calc_vel_profile arithmetic, Graph_LTPL.py:300,344 ------ */
int ltpl_tick_batch(ltpl_handle* handle, const ltpl_paths_in* in, const ltpl_tick_vel_in* vin,
                    ltpl_paths_out* out, ltpl_tick_vel_out* vout);

/* Compact result form of ltpl_tick_batch for callers that only need what Graph_LTPL.calc_vel_profile hands out (the
 * trajectory set, Graph_LTPL.py:396-408): rows [s, x, y, psi, kappa, vx, ax] of every valid action slot, trimmed to max_rows
 * (EXPORT.nmbr_export_points) and packed back to back on the DEVICE, so that only the bytes in use cross PCIe (the capacity
 * slabs of ltpl_paths_out / ltpl_tick_vel_out are mostly padding). `rows` may be any host memory; memory from ltpl_host_alloc
 * (page-locked) is written by DMA directly, other memory goes through the handle's staging buffer. */
typedef struct {
    int32_t  max_rows;              /* rows kept per trajectory, 0 = all                                          */
    int32_t  reserved0;
    int64_t  capacity_rows;         /* capacity of `rows` in rows of 7 doubles                                    */
    /* per scenario and action slot, index s * LTPL_MAX_ACTIONS + a */
    int32_t* action_id;             /* LTPL_ACT_* or LTPL_ACT_NONE                                                */
    int32_t* n_rows;                /* 0 = no trajectory in this slot                                             */
    int32_t* vel_bound;             /* OTH.py:906-911                                                             */
    int32_t* reduced;               /* action_set_red_len                                                         */
    int64_t* row_off;               /* first row of the slot in `rows`                                            */
    double*  rows;                  /* [total_rows * 7]                                                           */
    int64_t  total_rows;            /* out                                                                        */
} ltpl_traj_out;
int ltpl_tick_batch_compact(ltpl_handle* handle, const ltpl_paths_in* in, const ltpl_tick_vel_in* vin, ltpl_traj_out* out);
void* ltpl_host_alloc(size_t bytes);      /* page-locked host memory for ltpl_traj_out.rows; NULL on failure      */
void  ltpl_host_free(void* p);

/* Device-resident variant for benchmarks: upload the batch once, replay the fused kernel, download on demand.
 * The resident batch lives in the handle's staging buffers: ANY other entry point on the same handle (ltpl_plan_paths,
 * ltpl_tick_batch, ltpl_vel_profile, ltpl_process_objects, the planner calls) drops it, after which ltpl_batch_run /
 * ltpl_batch_download return LTPL_ERR_INVALID_ARG ("no resident batch") until the next ltpl_batch_upload.
 * ltpl_batch_run enqueues `reps` launches on the handle's stream and, when ms_total != NULL, brackets them with HIP
 * events on that stream and waits. */
int ltpl_batch_upload(ltpl_handle* handle, const ltpl_paths_in* in, const ltpl_tick_vel_in* vin,
                      int32_t cap_nodes, int32_t cap_pts);
int ltpl_batch_run(ltpl_handle* handle, int reps, float* ms_total);
int ltpl_batch_download(ltpl_handle* handle, ltpl_paths_out* out, ltpl_tick_vel_out* vout);
/* Profiling variant of ltpl_batch_run: HIP events between the kernels of the pipeline on the handle's stream.
 * ms_kernels[3] = summed durations over `reps` of {path kernel, follow preparation, velocity lane kernel}. */
int ltpl_batch_run_profile(ltpl_handle* handle, int reps, float* ms_kernels);
/* Average duration (ms) of the path kernel inside the last timed ltpl_batch_run, measured with HIP events recorded around
 * every launch of it on the handle's stream, i.e. under the same overlap with the velocity kernels as the timed region. */
int ltpl_batch_last_paths_ms(ltpl_handle* handle, float* ms_avg);

/* ------------------------------------------------------------------------------------------------------------------
 * Offline lattice build: the per-edge arithmetic of gen_edges.py:11-164 (two-point cubic from node
 * pose to node pose, arc-length sampling tph.interp_splines(stepsize_approx), heading / curvature tph.calc_head_curv_an,
 * turn-radius / velocity filter :127-140), of GraphBase.update_edge (element lengths, spline length, GraphBase.py:421-436) and
 * the curvature terms of gen_offline_cost.py:57-62 for ALL candidate edges of a track in one launch (lane = edge). Needs no
 * lattice handle. The host side (node skeleton, pruning, assembly) is graphbasedlocaltrajectoryplanner_amd/offline_build.py.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_edges;
    int32_t cap_samples;            /* capacity per edge of ltpl_offline_edges_out.samples                        */
    double  stepsize_approx;        /* SAMPLING.stepsize_approx                                                   */
    double  kappa_max_turn;         /* 1 / VEHICLE.veh_turn                                                       */
    const double*  start_x;         /* [n_edges] pose of the start node                                           */
    const double*  start_y;
    const double*  start_psi;
    const double*  end_x;           /* [n_edges] pose of the end node                                             */
    const double*  end_y;
    const double*  end_psi;
    const double*  kappa_max_vel;   /* [n_edges] 10 / (vel_raceline[start layer] * min_vel_race)^2   gen_edges.py:131-132 */
    const int32_t* raceline_edge;   /* [n_edges] 1: race line node -> race line node: coefficients are GIVEN (closed spline
                                                 through the race line, gen_edges.py:40-42,75-78) and the filter does not apply */
    const double*  given_coeff;     /* [n_edges * 8] x a0..a3, y a0..a3 (only read where raceline_edge != 0)      */
} ltpl_offline_edges_in;

typedef struct {
    int32_t* n_samples;             /* [n_edges]  (> cap_samples: the edge did not fit, samples not written)      */
    int32_t* valid;                 /* [n_edges]  1 = kept by the curvature filter                                */
    double*  coeff;                 /* [n_edges * 8]                                                              */
    double*  length;                /* [n_edges]  spline_length = sum of the element lengths                      */
    double*  kappa_avg;             /* [n_edges]  sum |kappa| / n_samples                  gen_offline_cost.py:57 */
    double*  kappa_range;           /* [n_edges]  |max kappa - min kappa|                  gen_offline_cost.py:61 */
    double*  samples;               /* [n_edges * cap_samples * 5] rows x, y, psi, kappa, el_length               */
} ltpl_offline_edges_out;

int ltpl_offline_edges(int device, const ltpl_offline_edges_in* in, ltpl_offline_edges_out* out);

/* ------------------------------------------------------------------------------------------------------------------
 * ABI v3 -- the planner: the iterative memory of class OnlineTrajectoryHandler
 * (graph_ltpl/online_graph/src/OnlineTrajectoryHandler.py:24-1040) behind the C ABI, batched over n_scen independent
 * planners that share one lattice handle (section 8f rank 2). One tick =
 *   ltpl_planner_calc_paths        Graph_LTPL.calc_paths        (Graph_LTPL.py:300-340) = OTH.update_objects + OTH.calc_paths
 *   ltpl_planner_calc_vel_profile  Graph_LTPL.calc_vel_profile  (Graph_LTPL.py:344-408) = OTH.get_ref_idx + OTH.calc_vel_profile
 * Object ingestion in front of it is ltpl_process_objects; zone bookkeeping (ObjectListInterface.update_zone) stays with
 * the caller, who passes the node ids the "overtaking_zones" filter currently removes (as for ltpl_plan_paths).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ltpl_planner ltpl_planner;

#define LTPL_PLANNER_MAX_KEYS 4     /* <= 3 primitives per tick + 'emergency' */

typedef struct {
    int32_t n_scen;
    int32_t n_w_last;               /* COST.w_last_edges                 OTH.py:109                               */
    const double* w_last_edges;
    double  v_max_offset;           /* ACTIONSET.v_max_offset            OTH.py:102                               */
    double  delaycomp;              /* DELAY.delaycomp                   OTH.py:117                               */
    double  calc_time_safety;       /* CALC_TIME.calc_time_safety        OTH.py:121                               */
    int32_t calc_time_buffer_len;   /* CALC_TIME.calc_time_buffer_len    OTH.py:122   (1 .. 16; reference default 5)  */
    int32_t filt_window_width;      /* SMOOTHING.filt_window_width       OTH.py:107 (only 1 is supported)         */
    double  dyn_model_exp, drag_coeff, m_veh;          /* OTH.__init__ arguments -> VpForwardBackward (OTH.py:137-144) */
    int32_t follow_control_type;    /* 0 = 'PD', 1 = 'PDtan'             OTH.py:112-114                           */
    int32_t reserved0;
    double  c_p, k_p, k_d, tan_w;
} ltpl_planner_config;

typedef struct {
    const int32_t* prev_action;     /* [n_scen] LTPL_ACT_* incl. LTPL_ACT_EMERGENCY: prev_action_id (Graph_LTPL.py:300)   */
    const double*  t_now;           /* [n_scen] the caller's time.time() (OTH.py:353-354,395)                     */
    /* objects of the tick (output of ObjectListInterface.process_object_list): same layout as ltpl_paths_in        */
    const int32_t* veh_off;         /* [n_scen + 1]                                                               */
    const int32_t* pos_off;         /* [n_veh_total + 1]                                                          */
    const double*  veh_radius;      /* [n_veh_total]                                                              */
    const double*  veh_vel;         /* [n_veh_total] VehObject.get_vel()                                          */
    const double*  pos_x;           /* [n_pos_total] own position first, then the prediction                      */
    const double*  pos_y;
    const int32_t* zone_off;        /* [n_scen + 1]                                                               */
    const int32_t* zone_gid;        /* [n_zone_total] global node ids removed by the "overtaking_zones" filter    */
} ltpl_planner_paths_in;

typedef struct {                    /* arguments of Graph_LTPL.calc_vel_profile (Graph_LTPL.py:344-351)           */
    const double*  pos_est_x;       /* [n_scen]                                                                   */
    const double*  pos_est_y;
    const double*  vel_est;
    const double*  vel_max;         /* [n_scen] one value per planner (ABI v6; ltpl_planner_* groups its velocity jobs per car), > 0                 */
    const double*  gg_scale;
    const double*  gg_ax;           /* [n_scen] constant local_gg tuple (ax, ay)                                  */
    const double*  gg_ay;
    const double*  safety_d;
    const int32_t* incl_emerg_traj; /* [n_scen]                                                                   */
    int32_t n_ax_max_machines;      /* rows of ax_max_machines (all tables together)                              */
    int32_t n_ax_tables;            /* ABI v6: 0 or 1 = ONE machine table for every planner of the call (was reserved0)            */
    const double*  ax_max_machines; /* [n_ax_max_machines * 2] rows [v, ax]                                       */
    /* ABI v4 -- LOCATION DEPENDENT FRICTION, local_gg as a dict {action id: [rows x (ax, ay)]} (OnlineTrajectoryHandler.py:649-666;
     * Graph_LTPL.py:360-365). Both NULL: constant friction (gg_ax / gg_ay). Otherwise rows gg_row_off[s * LTPL_PLANNER_MAX_KEYS + k]
     * .. gg_row_off[s * LTPL_PLANNER_MAX_KEYS + k + 1] of gg_rows belong to planner s and its k-th path key (key order of
     * ltpl_planner_paths_view); a key's row count must equal n_rows of that path (every path coordinate is represented by a row),
     * a key with zero rows falls back to (gg_ax, gg_ay)                                                            */
    const int32_t* gg_row_off;      /* [n_scen * LTPL_PLANNER_MAX_KEYS + 1]                                       */
    const double*  gg_rows;         /* [total rows * 2] rows [ax, ay]                                             */
    /* ABI v6 -- A FLEET OF DIFFERENT CARS (ltpl_fleet_*, and since the round-4 merge of the two state machines ltpl_planner_* as well:
     * the host planner groups its jobs per car; every table 1 .. 64 rows, every vel_max > 0, checked before any memory is cut):
     * Graph_LTPL.calc_vel_profile
     * takes vel_max and ax_max_machines per call, i.e. per vehicle (Graph_LTPL.py:344-351). n_ax_tables > 1: ax_max_machines holds that
     * many tables back to back, table t = rows ax_table_off[t] .. ax_table_off[t + 1] (ax_table_off[n_ax_tables] = n_ax_max_machines),
     * planner s uses table ax_table_idx[s]. Both NULL with n_ax_tables <= 1: one table for all.                                    */
    const int32_t* ax_table_off;    /* [n_ax_tables + 1] first row of every table                                  */
    const int32_t* ax_table_idx;    /* [n_scen] table of every planner                                              */
} ltpl_planner_vel_in;

/* Sizes a caller needs for the query buffers below: rows per (stitched) path / trajectory, nodes per path. */
typedef struct { int32_t cap_rows; int32_t cap_nodes; } ltpl_planner_caps;

typedef struct {
    /* state after calc_paths: path_dict of Graph_LTPL.calc_paths, keys in the reference's dict order             */
    int32_t  n_keys;
    int32_t  key_id[LTPL_PLANNER_MAX_KEYS];
    int32_t  n_rows[LTPL_PLANNER_MAX_KEYS];
    int32_t  n_nodes[LTPL_PLANNER_MAX_KEYS];
    int32_t  red_len[LTPL_PLANNER_MAX_KEYS];
    int32_t  start_node[2];
    int32_t  const_rows;            /* rows of const_path_seg or -1 (None)                                        */
    int32_t  closest_obj_index;     /* -1 = None                                                                  */
    double*  path_param[LTPL_PLANNER_MAX_KEYS];   /* caller buffers [cap_rows * 5] or NULL                        */
    double*  coeff[LTPL_PLANNER_MAX_KEYS];        /* caller buffers [cap_nodes * 8] or NULL                       */
    int32_t* nodes[LTPL_PLANNER_MAX_KEYS];        /* caller buffers [cap_nodes * 2] pairs (layer, node), -1 = None */
    int32_t* node_idx[LTPL_PLANNER_MAX_KEYS];     /* caller buffers [cap_nodes]                                   */
} ltpl_planner_paths_view;

typedef struct {
    /* result of calc_vel_profile: action_set / action_set_id of Graph_LTPL.calc_vel_profile (untrimmed)          */
    int32_t  n_keys;
    int32_t  key_id[LTPL_PLANNER_MAX_KEYS];
    int32_t  traj_id[LTPL_PLANNER_MAX_KEYS];
    int32_t  n_rows[LTPL_PLANNER_MAX_KEYS];
    int32_t  cut_index_pos, cut_layer;            /* outputs of get_ref_idx (OTH.py:601)                          */
    double   vel_plan, acc_plan;
    int32_t  n_vel_course;
    /* action_set_path_id of OTH.py:696-697,1034: one id per key of the tick INCLUDING keys dropped for a broken velocity
     * bound (the reference never removes them from this dict)                                                    */
    int32_t  n_ids;
    int32_t  id_key[LTPL_PLANNER_MAX_KEYS];
    int32_t  id_val[LTPL_PLANNER_MAX_KEYS];
    double*  traj[LTPL_PLANNER_MAX_KEYS];         /* caller buffers [cap_rows * 7] rows [s, x, y, psi, kappa, vx, ax] or NULL */
    double*  vel_course;                          /* caller buffer [cap_rows] or NULL                             */
} ltpl_planner_traj_view;

int ltpl_planner_create(ltpl_handle* handle, const ltpl_planner_config* cfg, ltpl_planner** out_planner);
int ltpl_planner_destroy(ltpl_planner* planner);
int ltpl_planner_get_caps(const ltpl_planner* planner, ltpl_planner_caps* caps);
const char* ltpl_planner_last_error(const ltpl_planner* planner);
/* OnlineTrajectoryHandler.set_initial_pose (OTH.py:181-270) for planner `scen` */
int ltpl_planner_set_start(ltpl_planner* planner, int32_t scen, double x, double y, double heading, double vel,
                           double max_heading_offset, int32_t* in_track, int32_t* cor_heading);
int ltpl_planner_calc_paths(ltpl_planner* planner, const ltpl_planner_paths_in* in);
/* The same call in two halves for callers that own the zone bookkeeping (gen_local_node_template.py:42-99 decides on the
 * START NODE OF THIS SEARCH): _begin = OTH.update_objects + OTH.py:308-414 (the zone members of `in` are ignored; the start
 * nodes are then visible through ltpl_planner_get_paths), _finish = seam (1) + OTH.py:429-513 with the zone node ids. */
int ltpl_planner_calc_paths_begin(ltpl_planner* planner, const ltpl_planner_paths_in* in);
int ltpl_planner_calc_paths_finish(ltpl_planner* planner, const int32_t* zone_off, const int32_t* zone_gid);
/* OnlineTrajectoryHandler.get_ref_idx (OTH.py:518-601) on its own; optional -- ltpl_planner_calc_vel_profile runs it when it was
 * not called for the tick. Results: cut_index_pos .. vel_course of ltpl_planner_traj_view. */
int ltpl_planner_get_ref_idx(ltpl_planner* planner, const double* pos_est_x, const double* pos_est_y);
int ltpl_planner_calc_vel_profile(ltpl_planner* planner, const ltpl_planner_vel_in* in);
/* copy-out of planner `scen`'s state (fills the counts, copies the arrays whose pointers are non-NULL) */
int ltpl_planner_get_paths(const ltpl_planner* planner, int32_t scen, ltpl_planner_paths_view* view);
int ltpl_planner_get_trajectories(const ltpl_planner* planner, int32_t scen, ltpl_planner_traj_view* view);

/* ------------------------------------------------------------------------------------------------------------------
 * ABI v5 -- the FLEET: the same planner (the iterative memory of OnlineTrajectoryHandler, OTH.py:24-1040) for MANY vehicles on
 * one lattice with the state in DEVICE memory. Every stage of a tick that ltpl_planner_* runs on the host per planner
 * (OTH.py:308-414 in front of seam (1), :429-513 behind it, get_ref_idx :518-601, the slicing / job construction / trajectory
 * assembly / backup and emergency branches of calc_vel_profile :603-1040) runs as a kernel with one wave64 per planner
 * (csrc/fleet_core.hpp) between the launches of the path kernel and the velocity kernel: no host work per planner, no host
 * synchronisation inside a tick. Entry points and structs are those of ltpl_planner_* (same argument meaning, same views), so a
 * caller switches by the prefix. Differences, all reported and none silent:
 *   - local_gg in both forms since ABI v6 (gg_row_off / gg_rows: friction rows per planner and path key; rows that do not match
 *     the coordinates of the path are an error of that planner, OTH.py:641-646);
 *   - the conditions on which the reference raises (OTH.py:334, :712, :830, :919, :923, :1029, ...) are detected per planner on the
 *     device: the call returns the status of the FIRST failing planner ("fleet: planner N: ..."), that planner keeps its error
 *     state (its later ticks are skipped) until ltpl_fleet_set_start gives it a new pose; the other planners are not affected;
 *   - calc_time_buffer_len <= 16; capacities as ltpl_planner_caps.
 * The TAPE form replays pre-uploaded inputs: ltpl_fleet_tape_append packs the inputs of one tick (both calls' arguments) into
 * device memory, ltpl_fleet_tape_run advances all planners through ticks [first, first + count) back to back on the handle's
 * stream and reports the device time between the first and the last launch -- the closed-loop throughput of the hot path with
 * state carried from tick to tick (bench.py extra.closed_loop_device).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct ltpl_fleet ltpl_fleet;

int ltpl_fleet_create(ltpl_handle* handle, const ltpl_planner_config* cfg /* n_scen = planners */, ltpl_fleet** out_fleet);
int ltpl_fleet_destroy(ltpl_fleet* fleet);
int ltpl_fleet_get_caps(const ltpl_fleet* fleet, ltpl_planner_caps* caps);
const char* ltpl_fleet_last_error(const ltpl_fleet* fleet);
/* OnlineTrajectoryHandler.set_initial_pose for one planner (computed on the host, uploaded as that planner's state) */
int ltpl_fleet_set_start(ltpl_fleet* fleet, int32_t planner, double x, double y, double heading, double vel,
                         double max_heading_offset, int32_t* in_track, int32_t* cor_heading);
/* the same start pose for the planners [first, past_last) in one call (start spline computed once, block image copied on the device) */
int ltpl_fleet_set_start_range(ltpl_fleet* fleet, int32_t first, int32_t past_last, double x, double y, double heading, double vel,
                               double max_heading_offset, int32_t* in_track, int32_t* cor_heading);
int ltpl_fleet_calc_paths(ltpl_fleet* fleet, const ltpl_planner_paths_in* in);
int ltpl_fleet_calc_paths_begin(ltpl_fleet* fleet, const ltpl_planner_paths_in* in);
int ltpl_fleet_calc_paths_finish(ltpl_fleet* fleet, const int32_t* zone_off, const int32_t* zone_gid);
int ltpl_fleet_get_ref_idx(ltpl_fleet* fleet, const double* pos_est_x, const double* pos_est_y);
int ltpl_fleet_calc_vel_profile(ltpl_fleet* fleet, const ltpl_planner_vel_in* in);
/* copy-out of one planner's state (a device-to-host copy of its block, then as ltpl_planner_get_*) */
int ltpl_fleet_get_paths(ltpl_fleet* fleet, int32_t planner, ltpl_planner_paths_view* view);
int ltpl_fleet_get_trajectories(ltpl_fleet* fleet, int32_t planner, ltpl_planner_traj_view* view);
/* digest of EVERY planner's result of the last tick, computed on the device (one wave per planner, read only): per planner
 * LTPL_FLEET_DIGEST_DOUBLES doubles -- [0] error word, [1] cut_index_pos, [2] cut_layer, [3] n_keys, [4] n_ids, [5] vel_plan,
 * [6] n_vel_course, [7] acc_plan; per key k: [8 + 7 k ..] key id, trajectory id, rows, s of the last row, vx of the first and the last row,
 * sum of vx; per id k: [8 + 7 K + 2 k ..] key id, id value (K = LTPL_PLANNER_MAX_KEYS). What the reference's tick recordings hold for every
 * tick: a whole fleet is checked against a recording without copying planner blocks (bench.py extra.closed_loop_device_mixed). */
#define LTPL_FLEET_DIGEST_DOUBLES (8 + 9 * LTPL_PLANNER_MAX_KEYS)
int ltpl_fleet_digest(ltpl_fleet* fleet, double* out /* [n_planners * LTPL_FLEET_DIGEST_DOUBLES] */, int32_t doubles_per_planner);
int ltpl_fleet_tape_clear(ltpl_fleet* fleet);
int ltpl_fleet_tape_append(ltpl_fleet* fleet, const ltpl_planner_paths_in* paths_in, const ltpl_planner_vel_in* vel_in);
int ltpl_fleet_tape_run(ltpl_fleet* fleet, int32_t first, int32_t count, float* ms_total /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* LTPL_HIP_H */
