/*
 * meshenv.h -- C-ABI of libmeshenv_hip.so: the MI355X-native, vectorised replacement for the
 * reference's BoudaryEnv step()/reset() hot path.
 *
 * The reference (ZhuoQiuMcgill/ReinforcementLearning4MeshGeneration) is pure Python and has no
 * FFI layer; the boundary it exposes for this path is the Gym class surface
 *   rl/boundary_env.py:18     class BoudaryEnv(MeshGeneration, gym.Env)
 *   rl/boundary_env.py:67-84  reset()            -> meshenv_reset
 *   rl/boundary_env.py:113-263 step(action)      -> meshenv_step
 *   rl/baselines/dummy_vec_env.py:41-51 step_wait (the vectorised caller)  -> one meshenv_step per call
 * so every entry point below cites the Python method it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MESHENV_E_* code on failure;
 *     meshenv_last_error() returns a human-readable message for the last failure on that handle
 *     (or for a failed meshenv_create when passed NULL);
 *   - pointers named *_dev are DEVICE pointers (e.g. torch.Tensor.data_ptr() of a tensor on the
 *     handle's GPU), owned by the caller, and must stay valid until the stream work that uses
 *     them has completed; pointers named *_host are host pointers;
 *   - a handle is bound to one GPU and one HIP stream; all work is stream-ordered on that stream;
 *     handles are not thread-safe;
 *   - there is no CPU fallback: every call fails with MESHENV_E_HIP if the GPU is unavailable.
 */
#ifndef MESHENV_H
#define MESHENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MESHENV_ABI_VERSION 1

#define MESHENV_OBS_DIM 18 /* observation_space shape, rl/boundary_env.py:38-39 */
#define MESHENV_ACT_DIM 3  /* action_space shape, rl/boundary_env.py:27 */

enum {
    MESHENV_OK = 0,
    MESHENV_E_ARG = -1,    /* bad argument (null pointer, size out of range, unsupported parameter) */
    MESHENV_E_HIP = -2,    /* a HIP runtime call failed / no usable GPU */
    MESHENV_E_RANGE = -3,  /* env / domain index out of range */
    MESHENV_E_STATE = -4   /* call not valid in the handle's current state */
};

/* per-env status bits returned by meshenv_get_status() */
enum {
    MESHENV_ST_NO_REFERENCE = 1, /* candidate list empty: the reference's find_next_state() returned None
                                    (rl/boundary_env.py:569-571); obs is all zeros */
    MESHENV_ST_LOG_OVERFLOW = 2  /* more elements / new vertices than log_capacity; counts stay exact,
                                    the log holds the first log_capacity entries */
};

/* Constants of rl/boundary_env.py:26-56, general/mesh.py:26-28 and general/components.py:721-722.
 * meshenv_default_params() fills the reference's values.  neighbor_num and radius_num fix the
 * observation layout and must stay 6 and 3. */
typedef struct MeshEnvParams {
    int32_t struct_size;        /* sizeof(MeshEnvParams), for ABI checking */
    int32_t neighbor_num;       /* 6    rl/boundary_env.py:32 */
    int32_t radius_num;         /* 3    rl/boundary_env.py:33 */
    int32_t fail_limit;         /* 100  rl/boundary_env.py:260 */
    int32_t log_capacity;       /* elements (and new vertices) logged per env per episode; 0 = no log */
    int32_t reserved;
    double radius;              /* 4      rl/boundary_env.py:35 */
    double max_ref_angle;       /* 0.972*pi  general/mesh.py:26 */
    double key_lambda;          /* 0.618  general/mesh.py:213 */
    double min_degree;          /* 0.01*pi general/components.py:722 */
    double max_degree;          /* 0.99*pi general/components.py:721 */
    double same_point_eps;      /* 0.001  rl/boundary_env.py:601 */
    double ray_length;          /* 10000  general/mesh.py:540 */
} MeshEnvParams;

typedef struct MeshEnv MeshEnv; /* opaque */

void meshenv_default_params(MeshEnvParams *p);

int meshenv_abi_version(void);

/* Number of HIP devices visible (0 if none / runtime unusable). */
int meshenv_device_count(void);

/*
 * Construct n_envs environments over n_domains polygons.
 * Replaces BoudaryEnv.__init__ (rl/boundary_env.py:21-65) + MeshGeneration.__init__
 * (general/mesh.py:17-30) for a whole batch.
 *
 *   dom_offsets_host[n_domains+1]  vertex offsets of each domain ring in dom_xy_host
 *   dom_xy_host[2*total]           clockwise rings, (x, y) interleaved, float64
 *   dom_consts_host[3*n_domains]   per domain: original_area, est_min_l, est_crit_l
 *                                  (Boundary2D.poly_area, general/components.py:477-479;
 *                                   MeshGeneration.estimate_area_range, general/mesh.py:679-692)
 *   env_domain_host[n_envs]        domain index of every environment
 *   stream                         hipStream_t to bind (NULL = the device's default stream)
 *
 * The environments come up already reset (their first observation is available through
 * meshenv_reset with a NULL mask).
 */
int meshenv_create(int device, int n_domains, const int32_t *dom_offsets_host, const double *dom_xy_host,
                   const double *dom_consts_host, int n_envs, const int32_t *env_domain_host,
                   const MeshEnvParams *params, void *stream, MeshEnv **out);

/*
 * The same with one GENERATED domain per environment (BASELINE.json configs[4]: ui/GenerateRandomPolygon.py envs with
 * variable vertex count): env k runs on the polygon of Python's random.Random(seed0 + k) --
 *   generatePolygon(250, 250, 100, 0.55, 0.7, numVerts)   ui/GenerateRandomPolygon.py:5-49, 63 (numVerts = num_verts, or
 *                                                         randint(8, 64) from the same stream when num_verts == 0)
 *   consecutive duplicate pixels dropped, coordinates / 100 (general/polygon.py:110-117), made clockwise by the
 *   drawing UI's rule (ui/tk-ui.py:84-101, 169-176), every edge split into ceil(length / edge) equal pieces with an even
 *   vertex total, coordinates rounded to 4 places --
 * generated, and its constants (poly_area, estimate_area_range) computed, on the device (csrc/meshenv_domgen.h); the host
 * side is only a prefix sum over the ring lengths.  reinforcementlearning4meshgeneration_amd.domains.random_domain(seed)
 * is the host restatement of the same ring (bit-identical; tests/test_gpu_domgen.py).  Fails with MESHENV_E_STATE if a
 * polygon keeps fewer than 5 distinct pixels (the host function would draw another one from the stream).
 */
int meshenv_create_random(int device, int n_envs, uint64_t seed0, int num_verts, double edge, const MeshEnvParams *params,
                          void *stream, MeshEnv **out);

/*
 * The same generator followed by the REFERENCE's densification instead of the uniform split: Density.calculate_density
 * (ui/tk-ui.py:252-276; arithmetic-progression spacing from density(prev) * base_length to density(k) * base_length on
 * every edge, Python round half-to-even, one middle point dropped on the last edge for an even total), the UI's
 * clockwise rule on the result (:84-101, 169-176) and read_polygon's / 100 (general/polygon.py:110-117) -- the route
 * reinforcementlearning4meshgeneration_amd.domains.density_domain(random_polygon_px(seed), base_length) restates on the
 * host, bit-identical (tests/test_gpu_domgen.py).  base_length is in PIXELS (45 = the 0.45 spacing of the uniform mode),
 * density the value every generated vertex carries.  Where the reference raises ZeroDivisionError -- an edge of 0.5 - 1.5
 * spacings gets round(...) == 0 interior points and (B - A) / 0 -- no environment can be built:
 *   seeds_host   [n_envs] nullable: the seed of every ring (NULL: seed0 + k), so that a caller can leave those seeds out
 *   raises_host  [n_envs] nullable: 1 where calculate_density raises for that ring; 2 where the polygon itself is not
 *                generated on the device (fewer than 5 distinct pixels: the host function would draw another one)
 *   out          nullable: NULL = probe only (fills raises_host, builds nothing)
 * Returns MESHENV_E_STATE, *out = NULL and raises_host filled if any ring raises (MeshVecEnv.from_random_density probes a
 * seed range first and then passes the seeds that are defined).  cos / sin of the edge directions come from the host
 * libm (integer pixel offsets: a finite table), so the rings are the reference's bit for bit.
 */
int meshenv_create_random_density(int device, int n_envs, uint64_t seed0, const uint64_t *seeds_host, int num_verts,
                                  double base_length, double density, const MeshEnvParams *params, void *stream, MeshEnv **out,
                                  uint8_t *raises_host);

/*
 * Density.calculate_density + clockwise rule + / 100 for EXPLICIT pixel polygons with per-vertex densities (handle-free,
 * synchronous; the drawn-domain route of ui/tk-ui.py):
 *   poly_offsets_host [n_polys+1], pixels_host [2*total] float64 (x, y), integer_pixels != 0: the coordinates are Python
 *   ints (the UI's event coordinates; then every value must be integral) -- the reference's arithmetic differs between
 *   int and float operands (exact int squares and an int 0 in atan2 against pow(x, 2.0) and -0.0),
 *   densities_host [total] nullable (1.0),
 *   count_host [n_polys] ring lengths, status_host [n_polys]: 0 ok, 1 the reference raises ZeroDivisionError, 2 more than
 *   2048 ring points, 3 fewer than 3 distinct pixels; xy_host [2*cap_points] nullable: the rings of the polygons with
 *   status 0, back to back in polygon order (MESHENV_E_RANGE if they do not fit).  3..256 vertices per polygon.
 */
int meshenv_density_rings(int device, int n_polys, const int32_t *poly_offsets_host, const double *pixels_host, int integer_pixels,
                          const double *densities_host, double base_length, int32_t *count_host, uint8_t *status_host,
                          double *xy_host, int64_t cap_points);

/* Host-side readout of one domain of the handle's table (synchronises the stream): xy_host[2*cap_points] receives the
 * ring, *n_out its length, consts_host[3] (nullable) original_area, estimated_area_range[0]**2, [1]**2. */
int meshenv_get_domain(MeshEnv *h, int domain, double *xy_host, int cap_points, int32_t *n_out, double *consts_host);

void meshenv_destroy(MeshEnv *h);

const char *meshenv_last_error(const MeshEnv *h);

/* Rebind the handle to another HIP stream (e.g. torch.cuda.current_stream().cuda_stream). */
int meshenv_set_stream(MeshEnv *h, void *stream);

int meshenv_num_envs(const MeshEnv *h);
int meshenv_max_ring(const MeshEnv *h);
/* Environments per workgroup of the single-step kernel: 1 = k_step<false> (one wave per workgroup), 8 / 16 =
 * k_step_group<G> (chosen at creation from n_envs, see meshenv_create in csrc/meshenv_hip.hip). */
int meshenv_group_size(const MeshEnv *h);
/* Which single-step kernel meshenv_step launches: 0 = k_step<false> (one wave per workgroup), 1 = k_step_group<G>
 * (checks, workgroup barrier, updates dealt over the SIMDs), 2 = k_step_spec<G> (no barrier: an action that survives the
 * cheap exact tests is extracted speculatively by an idle wavefront while its checks finish).
 * 3 = k_step<false, ., true>: after a front smoothing (meshenv_smooth with interior = 0, or a meshenv_move that went through
 * smooth_pave) and until the next reset of ALL envs rings may hold vertices off the 1e-4 lattice, whose clockwise angles
 * can sit exactly on a rounding boundary; steps then run the one-wave-per-env kernel in the instantiation that decides
 * those like the reference's libm (meshenv_atan2_exact).
 * 4 = k_step_group<16, ., true>: the CU-group kernel with its LDS packed by each ring's own length (batches of mixed
 * domains whose sixteen longest rings would not fit one CU).  5 = k_step_group<G, ., false, true>: the CU-group kernel of
 * batches whose ring stride is at most 64 slots (every ring pass is one 64-lane pass, no chunk loops).
 * 6 = k_step<false, true, false, true>: the one-wave-per-env kernel of such batches (default geometry constants).
 * 7 / 8 = k_step<false, true, false, false | true, true>: the throughput-regime forms of 0 / 6 (batches staged record-first,
 * from 8192 envs): a rule-0 quad with a corner that cannot be valid is rejected before the point-in-polygon pass.
 * (2 exists in -DMESHENV_DEV builds of the library only.) */
int meshenv_step_kernel(const MeshEnv *h);
/* The smoothing kernels evaluate `x ** 2` like the reference's libm (CPython's float ** 2 is pow(x, 2.0), which glibc does
 * not round correctly: it differs from x * x in 0.085 % of the arguments) through a restatement of glibc's pow
 * (csrc/meshenv_libm.h) that the library validates against the libm of the running process at the first smoothing call.
 * Returns 1: validated, the smoothers square as the reference does; 0: this libm computes pow differently (or
 * MESHENV_LIBM_EXACT=0 is set), the smoothers use x * x and agree with the reference to one ulp per squaring instead of
 * bit for bit; -1: not checked yet (no smoothing call so far).  The reference has no counterpart: it IS that libm. */
int meshenv_libm_exact(const MeshEnv *h);
/* Every clockwise angle (Vertex.to_find_clockwise_angle, general/components.py:99-108) is -atan2(cross, dot) rounded to
 * 1e-4 rad.  The kernels take ocml's atan2, which is within an ulp of the libm the reference calls, and re-evaluate the
 * few angles that sit on a rounding boundary (k + 0.5) e-4 -- the front smoother constructs such angles -- with a
 * restatement of glibc's atan2 (csrc/meshenv_libm.h) whose table is read from the libm image of the running process and
 * which is validated against that libm's atan2 when the first handle is created.  Returns 1: validated, the quantised
 * angles are the reference's in every case; 0: table not found / another algorithm (the boundary cases then take a
 * correctly rounded atan2, which agrees with glibc's in 99.93 % of them), or MESHENV_LIBM_EXACT=0 (ocml only).
 * Host-only: needs no GPU. */
int meshenv_atan2_exact(void);

/*
 * reset(): rl/boundary_env.py:67-84 for every env whose mask byte is non-zero (all envs when
 * mask_dev is NULL).  obs_dev[n_envs*18] receives the first observation of the reset envs and the
 * current observation of the others.
 */
int meshenv_reset(MeshEnv *h, const uint8_t *mask_dev, float *obs_dev);

/*
 * reset(static=True): rl/boundary_env.py:67 -> find_next_state(static=static) -> PointEnvironment(static=True)
 * (general/components.py:1213-1218): observation entry 1 (row 0, second value) carries 0 instead of the area ratio.
 * is_static = 0 is meshenv_reset.  Also empties the move() API's not_valid_points of the reset envs
 * (rl/boundary_env.py:73).
 */
int meshenv_reset_static(MeshEnv *h, const uint8_t *mask_dev, float *obs_dev, int is_static);

/*
 * step(): rl/boundary_env.py:113-263 for all envs, one kernel launch.
 *   actions_dev[n_envs*3] float32   (rule type, x, y) as SB3 hands them over
 *   obs_dev[n_envs*18]    float32   next observation (after auto-reset: the first observation of
 *                                   the new episode, like SB3's DummyVecEnv)
 *   reward_dev[n_envs]    float64   the reference returns np.float64 (rl/boundary_env.py:263)
 *   done_dev[n_envs]      uint8
 *   complete_dev[n_envs]  uint8     info['is_complete']
 *   terminal_obs_dev      nullable; for envs with done != 0 receives the last observation of the
 *                         finished episode (SB3's infos[k]["terminal_observation"])
 *   auto_reset            non-zero: envs that finish are reset inside the same launch
 */
int meshenv_step(MeshEnv *h, const float *actions_dev, float *obs_dev, double *reward_dev, uint8_t *done_dev,
                 uint8_t *complete_dev, float *terminal_obs_dev, int auto_reset);

/*
 * n_steps consecutive step() calls in ONE launch (environments are independent, so a wave can
 * advance its env without any inter-wave synchronisation).  actions_dev is [n_steps][n_envs][3];
 * reward_dev/done_dev/complete_dev are [n_steps][n_envs]; obs_dev[n_envs*18] receives the
 * observation after the last step; auto_reset as in meshenv_step.  Used for open-loop (scripted /
 * random-policy) rollouts; a policy in the loop calls meshenv_step.
 */
int meshenv_rollout(MeshEnv *h, int n_steps, const float *actions_dev, float *obs_dev, double *reward_dev,
                    uint8_t *done_dev, uint8_t *complete_dev, int auto_reset);

/*
 * move(new_point, type): rl/boundary_env.py:265-432, the deterministic extraction API the reference's ANN / testbed
 * scripts drive, for all envs in one launch and for Python-float arguments:
 *   points_dev[n_envs*2]  float64  (radius fraction, angle): the candidate point is
 *                                  base_length * radius * r * (cos a, sin a) in the reference vertex's frame, rounded
 *                                  to 6 places, scale 1 (rl/boundary_env.py:266-269, 105)
 *   type_dev[n_envs]      float64  rule selector: <= 0.3 the rule -1 quad, >= 0.7 the rule +1 quad, else the new point
 *                                  (TYPE_THRESHOLD, rl/boundary_env.py:19, 286-310; no find_same_point here)
 *   obs_dev[n_envs*18]    float32  next observation, static form (see meshenv_reset_static)
 *   done_dev / complete_dev        done: ring <= 5 after a valid move; complete: ring <= 4
 *   code_dev[n_envs]      uint8    MESHENV_MOVE_*
 * The reward is always 0 in the reference and is not returned.  A rejected move appends the reference vertex to the
 * env's not_valid_points, which the next selection skips (general/mesh.py:284-288); a valid move empties the list.
 * current_area and failed_num are not touched (the reference's move() does not).  No auto-reset: reset the envs with
 * done != 0 or code >= MESHENV_MOVE_RAISES through meshenv_reset_static with a mask.
 */
enum {
    MESHENV_MOVE_OK = 0,
    MESHENV_MOVE_NONE = 1,            /* the observation is None (zeros): no reference vertex on a ring of <= 4 */
    MESHENV_MOVE_RAISES = 2,          /* ring <= 5 (or no reference vertex) on entry: the reference raises
                                         UnboundLocalError (`is_complete` unbound, rl/boundary_env.py:283-285, 432);
                                         nothing is changed, the cached observation is returned */
    MESHENV_MOVE_NEEDS_SMOOTHING = 3, /* only on handles without a usable element log (log_capacity = 0, overflow, or
                                         too large for the smoother's LDS): the smooth_pave below cannot be run and the
                                         episode ends instead, done = 1, complete = 0 */
    MESHENV_MOVE_SMOOTH_RAISES = 4    /* the reference raises out of this move(): inside that smooth_pave
                                         (MESHENV_SMOOTH_RAISES below) or, after a smooth_pave that accepted a NaN vertex
                                         (MESHENV_SMOOTH_NONFINITE), in the find_next_state that follows it -- in the
                                         second case not_valid_points has already been emptied, :416-422; done = 1, the
                                         caller resets */
};
/* No selectable reference vertex on a front of more than 4 vertices (every candidate is listed in not_valid_points):
 * the reference runs smooth_pave (front + interior smoothing, candidate rebuild; rl/boundary_env.py:405-412), ends the
 * episode if not_valid_points repeats the list of the previous such smoothing (first entry, last entry and length;
 * last_not_valid_points survives reset(), :416-420), empties the list and selects again (:422-426).  meshenv_move does
 * the same in the same call (the kernels of meshenv_smooth under a mask of those envs); such a move returns
 * MESHENV_MOVE_OK / MESHENV_MOVE_NONE like any other. */
int meshenv_move(MeshEnv *h, const double *points_dev, const double *type_dev, float *obs_dev, uint8_t *done_dev,
                 uint8_t *complete_dev, uint8_t *code_dev);

/* MeshGeneration.smooth_pave(boundary.vertices, updated_boundary.vertices, iteration=..., interior=...),
 * general/mesh.py:790-795, on the RUNNING episode (which = 0) of every env with mask_dev[e] != 0 (mask_dev NULL: all).  The
 * Vertex.segments graph the reference walks is rebuilt from the element log, so the handle needs log_capacity > 0.
 *
 * interior != 0 (the post-processing call of general/EBRD.py:393): smooth_fixed_vertices (general/mesh.py:1258-1288) --
 *   Gauss-Seidel relaxation of the generated vertices that are off the front, in boundary.vertices order, until the moved
 *   vertices' coordinate sum changes by <= 0.001 or `iteration` sweeps -- then find_reference_candidates(0)
 *   (general/mesh.py:233-261) on the front.  The front, the reference vertex and the observation are unchanged (as in the
 *   reference), the candidate list is the rebuilt one; the first step after the call must be a meshenv_step or meshenv_step_actor, not a meshenv_rollout (it commits
 *   the re-selection the rebuild parked: the reference re-selects at the end of every step, accepted or not).
 *   is_static and obs_dev are ignored.
 * interior == 0 (what move() runs when no reference vertex is selectable, rl/boundary_env.py:405-420):
 *   smooth_current_boundary_3 (general/mesh.py:939-1028) moves the generated vertices ON the front first, then the two
 *   steps above; because the front moved, the point environment is recomputed at once -- find_next_state(static =
 *   is_static != 0), the call move() makes right after smooth_pave -- and obs_dev [n_envs][18] (nullable) receives its
 *   observation (zeros where the reference returns None: status bit MESHENV_ST_NO_REFERENCE).
 *
 * which = 1 (interior != 0 only): the same relaxation on the ARCHIVED episode of every selected env (the mesh
 *   meshenv_get_last_episode reads, e.g. an episode that auto-reset ended by truncation); its front is recovered from the
 *   logs (vertices on edges used an odd number of times by domain ring + elements); no candidate list is involved;
 *   MESHENV_SMOOTH_NOT_FINISHED where nothing is archived yet.
 *
 * The vertex log (meshenv_get_elements, meshenv_element_quality) holds the moved coordinates afterwards.
 *   sweeps_dev [n_envs] int32, nullable: sweeps of the interior relaxation; MESHENV_SMOOTH_SKIPPED for masked-out envs;
 *              MESHENV_SMOOTH_LOG_OVERFLOW (status bit MESHENV_ST_LOG_OVERFLOW: graph incomplete) and
 *              MESHENV_SMOOTH_DEGREE (a vertex with more than 16 neighbours) leave the env untouched;
 *              MESHENV_SMOOTH_RAISES: the reference raises inside a vertex construction of the front smoother
 *              (math.sqrt of a negative number; a zero divisor between Python operands, i.e. two coincident DOMAIN
 *              vertices) -- the vertices moved before that point stay moved, nothing else ran.  A zero divisor with a
 *              NumPy-scalar operand (the coordinates of generated vertices are np.float64) only warns in the reference
 *              and yields nan: reproduced -- the nan position fails is_inside_boundary and the vertex stays where it
 *              is -- except where the reference ACCEPTS the nan position (every test against the surrounding polygon
 *              reads False for the original position too): MESHENV_SMOOTH_NONFINITE, the smoother stops at that vertex;
 *              the reference goes on with a NaN vertex and raises in its next find_next_state (int(nan),
 *              general/components.py:1249)
 *   diff_dev   [n_envs] float64, nullable: the last |sum - previous sum| (what the reference prints) */
enum {
    MESHENV_SMOOTH_SKIPPED = -1, MESHENV_SMOOTH_LOG_OVERFLOW = -2, MESHENV_SMOOTH_DEGREE = -3,
    MESHENV_SMOOTH_NOT_FINISHED = -4, /* meshenv_smooth_final on a front of more than 5 vertices */
    MESHENV_SMOOTH_INDEX_ERROR = -5,  /* the reference raises IndexError here (empty common-neighbour list) */
    MESHENV_SMOOTH_RAISES = -6,
    MESHENV_SMOOTH_NONFINITE = -7
};
int meshenv_smooth(MeshEnv *h, int which, const uint8_t *mask_dev, int iteration, int interior, int is_static,
                   int32_t *sweeps_dev, double *diff_dev, float *obs_dev);

/* MeshGeneration.smooth(boundary.vertices, lr_1, lr_2, iteration), general/mesh.py:1290-1392 -- the post-processing of a
 * FINISHED mesh (general/EBRD.py:391: front of <= 5 vertices) -- on every env with mask_dev[e] != 0:
 *   which = 0: the running episode, ended complete and not reset yet (step with auto_reset = 0, smooth, read, reset);
 *   which = 1: the archived episode (what meshenv_get_last_episode reads: the finished mesh auto-reset left behind); its
 *              front, of which smooth() only needs the membership, is recovered from the logs (the vertices on edges used
 *              an odd number of times by domain ring + elements; a front of 4 closed by the last element = that element).
 * Every generated vertex, front vertices included, by the number of elements around it -- 4th-vertex estimates for 1 and 2
 * (general/mesh.py:1305-1361), the Laplacian step otherwise -- until the coordinate sum of the whole vertex list changes by
 * <= 0.001 or `iteration` sweeps.  The reference's defaults are lr_1 = lr_2 = 0.999, iteration = 400.  The vertex log (and,
 * for which = 0, the front's coordinates) hold the result; the candidate list of the finished episode is not rebuilt -- the
 * next call on such an env is a reset.  sweeps_dev / diff_dev as meshenv_smooth, plus MESHENV_SMOOTH_NOT_FINISHED (front
 * > 5, nothing archived, archived episode truncated) and MESHENV_SMOOTH_INDEX_ERROR (env untouched). */
int meshenv_smooth_final(MeshEnv *h, int which, const uint8_t *mask_dev, int iteration, double lr_1, double lr_2,
                         int32_t *sweeps_dev, double *diff_dev);

/* Host-side readout of one env's not_valid_points (synchronises the stream): xy_host[2*cap_points], *count = length. */
int meshenv_get_not_valid(MeshEnv *h, int env, double *xy_host, int cap_points, int32_t *count);

/* The same list as vertex ids (the reference's list holds Vertex objects: identity), and the summary of
 * last_not_valid_points (rl/boundary_env.py:48,416-422: set where move() smooths, compared by first entry, last entry and
 * length, never cleared by reset()): last_host[4] = first id, last id, length, 1 if it was recorded in the running
 * episode (reset() deep-copies the domain, rl/boundary_env.py:69: vertices of an earlier episode are different objects).  ids_host / last_host nullable. */
int meshenv_get_not_valid_ids(MeshEnv *h, int env, int32_t *ids_host, int cap_ids, int32_t *count, int32_t *last_host);

/*
 * Multi-GPU exchange message.  With msg_dev != NULL every following meshenv_step / meshenv_rollout also writes
 * msg_dev[n_envs*21] float32 = (obs[18] | reward | done | complete) per env -- the buffer a rank hands to
 * all_gather -- straight from the step kernel (no separate pack kernels).  NULL switches it off.  The pointer
 * may be changed between steps (double buffering against an in-flight collective).
 */
#define MESHENV_MSG_DIM 21
int meshenv_set_packed_output(MeshEnv *h, float *msg_dev);

/* Per-env status bits (MESHENV_ST_*) after the last step/reset; status_dev[n_envs]. */
int meshenv_get_status(MeshEnv *h, uint8_t *status_dev);

/*
 * Host-side readout of one environment (synchronises the stream).  Any pointer may be NULL.
 *   ring_ids_host[max_ring]      global vertex ids of updated_boundary.vertices, in ring order
 *   ring_xy_host[2*max_ring]
 *   cand_key_host[max_ring]      cached candidate key per ring slot (NaN where not a candidate)
 *   cand_stamp_host[max_ring]    insertion stamp per ring slot (larger = nearer the list head
 *                                among equal keys; INT32_MIN where not a candidate)
 *   scalars_host[8]: ring length, reference ring index, n_elements (len(generated_meshes)),
 *                    failed_num, n_vertices (len(boundary.vertices)), status bits, domain, n0
 *   fscalars_host[2]: current_area, base_length
 */
int meshenv_get_state(MeshEnv *h, int env, int32_t *ring_ids_host, double *ring_xy_host, double *cand_key_host,
                      int32_t *cand_stamp_host, int32_t *scalars_host, double *fscalars_host);

/*
 * generated_meshes / boundary.vertices of one env (rl/boundary_env.py:192, general/mesh.py:589-590):
 *   quads_host[4*cap]      global vertex ids per element, in creation order
 *   vertex_xy_host[2*cap]  coordinates of all vertices ever (initial ring first, then new ones)
 * cap_elems / cap_verts are the capacities of the caller's buffers; *n_elem / *n_vert receive the
 * number of entries written.
 */
int meshenv_get_elements(MeshEnv *h, int env, int32_t *quads_host, int cap_elems, double *vertex_xy_host,
                         int cap_verts, int32_t *n_elem, int32_t *n_vert);

/*
 * The same for the last FINISHED episode of the env.  The reference's evaluation callback reads
 * env.generated_meshes after `done` and before it resets (rl/baselines/CustomizeCallback.py:131-133); under
 * auto-reset the kernel has already started the next episode by then, so every reset that ends an episode with at
 * least one element archives that episode's log (double-buffered in HBM, no copy).  *flags: bit 0 = is_complete,
 * bit 1 = the log overflowed log_capacity; *episodes: number of episodes archived so far (0 = nothing yet, n_elem 0).
 */
int meshenv_get_last_episode(MeshEnv *h, int env, int32_t *quads_host, int cap_elems, double *vertex_xy_host,
                             int cap_verts, int32_t *n_elem, int32_t *n_vert, int32_t *flags, int32_t *episodes);

/*
 * Per-element quality report, computed on the device for every env at once (one launch).
 *   which      0 = the running episodes, 1 = the archived (last finished) episodes
 *   elem_dev   [n_envs][log_capacity][MESHENV_QUALITY_DIM] float64, nullable: per element
 *                min corner angle (deg), max corner angle (deg), scaled Jacobian, stretch, taper, 'robust',
 *                area, 'default' -- Mesh.get_quality(type) / compute_area() of general/components.py:863-950, the
 *                in-repo forms of the five Verdict measures Measurement/quality_verdict.py:133-148 requests;
 *                rows >= count are left untouched
 *   stats_dev  [n_envs][MESHENV_QUALITY_DIM][4] float64, nullable: minimum, average, maximum, variance per
 *                measure over the env's elements (what DumpQualityStats prints, quality_verdict.py:77-90)
 *   count_dev  [n_envs] int32, nullable: number of elements reported per env
 * Stream-ordered on the handle's stream.
 */
#define MESHENV_QUALITY_DIM 8
int meshenv_element_quality(MeshEnv *h, int which, double *elem_dev, double *stats_dev, int32_t *count_dev);

/*
 * MeshGeneration.get_quality(element, index) (general/mesh.py:1728-1747; called as env.get_quality(env.generated_meshes[i], 4)
 * by rl/baselines/testbed.py:194-197, and by generate_meshes_canvas, general/mesh.py:1770-1778, for the labels of
 * save_meshes) for n arbitrary quads, on the device:
 *   quad_xy_dev [n][4][2] float64, vertices in Mesh.vertices order; out_dev [n] float64
 *   index 0 = Mesh.get_quality() 'default', 1 = compute_element_quality (general/mesh.py:1714-1726), 3 = 'stretch',
 *         4 = 'robust', 5 = 'strong' (general/components.py:863-930).
 * Indices 2 and 6 add compute_ele_boundary_quality of the ring right after the extraction; that value exists only
 * inside the step that extracted the element (it is the step's reward term) -> MESHENV_E_ARG.
 * Stream-ordered on the handle's stream.
 */
int meshenv_quad_quality(MeshEnv *h, int n, const double *quad_xy_dev, int index, double *out_dev);

/*
 * Work counters since creation (roofline accounting), summed over envs:
 *   out_host[0] env steps executed, [1] valid extractions, [2] sum of ring lengths over all steps,
 *   [3] sum of ring lengths over valid steps.
 */
int meshenv_counters(MeshEnv *h, uint64_t *out_host);

/*
 * Kernel timing with HIP events on the handle's stream (bench.py's roofline figure).  meshenv_set_timing(h, k),
 * k > 0: every other group of k consecutive meshenv_step / meshenv_rollout launches is bracketed by one event pair
 * and reported as (elapsed / k), the average launch duration inside the group including the launch-to-launch gap
 * (an event pair costs several microseconds of stream time on this stack, so bracketing single ~20 us launches would
 * both slow the loop and inflate the figure).  k = 0 switches timing off.  Arming creates a pool of
 * MESHENV_TIMING_POOL event pairs and clears the record; meshenv_kernel_times() synchronises and copies the
 * per-launch averages (milliseconds, in launch order) of the groups recorded since then -- at most the newest
 * MESHENV_TIMING_POOL -- into ms_host[cap] and clears the record.  Call it before changing k.
 */
#define MESHENV_TIMING_POOL 4096
int meshenv_set_timing(MeshEnv *h, int enable);
int meshenv_kernel_times(MeshEnv *h, float *ms_host, int cap, int32_t *n_out);

/*
 * Fused SAC actor forward -- the caller of the hot path, not part of the environment.
 * Architecture fixed to the reference's policy (rl/baselines/RL_Mesh.py:183-196: MlpPolicy, ReLU,
 * net_arch [128, 128, 128]; SB3's squashed-Gaussian actor): latent = MLP(obs[18]); mean / log_std = Linear(latent);
 * action = low + 0.5 * (tanh(mean + exp(clamp(log_std, -20, 2)) * noise) + 1) * (high - low); noise_dev = NULL gives
 * the deterministic action.  One launch instead of ~25, so obs -> action -> step stays on the GPU.
 * Weights: host pointers in torch.nn.Linear layout ([out][in], row-major), float32.
 */
typedef struct MeshActor MeshActor;
int meshenv_actor_create(int device, void *stream, MeshActor **out);
void meshenv_actor_destroy(MeshActor *a);
int meshenv_actor_set_stream(MeshActor *a, void *stream);
int meshenv_actor_load(MeshActor *a, const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                       const float *b3, const float *w_mu, const float *b_mu, const float *w_log_std,
                       const float *b_log_std, const float *low, const float *high);
int meshenv_actor_forward(MeshActor *a, int n, const float *obs_dev, const float *noise_dev, float *actions_dev);
/*
 * The stochastic action with the noise drawn inside the kernel: eps[env][k] is a standard normal from Philox4x32-10
 * keyed by `seed` with counter (env, `counter`) and Box-Muller -- the caller passes a fresh `counter` per rollout step
 * (same seed + counter -> same actions).  eps_out_dev [n,3] (nullable) receives the noise that was used, so that
 * meshenv_actor_forward(obs, eps_out) reproduces the actions bit for bit.  Saves the separate random-number launch.
 */
int meshenv_actor_sample(MeshActor *a, int n, const float *obs_dev, uint64_t seed, uint64_t counter, float *actions_dev,
                         float *eps_out_dev);

/* One launch per vector step of the closed RL loop: meshenv_step(actions_dev, ...) followed by meshenv_actor_sample /
 * meshenv_actor_forward on the observations it produced -- i.e. obs_dev, reward_dev, done_dev, complete_dev,
 * terminal_obs_dev exactly as meshenv_step writes them, and actions_next_dev [n_envs][3] = the policy's actions for the
 * NEXT step (sample != 0: SAC's stochastic actor with in-kernel Philox noise keyed by (seed, counter), eps_out_dev nullable;
 * sample == 0: the mean action).  actions_next_dev must not alias actions_dev (ping-pong two buffers).  When the batch runs
 * on the CU-group kernel (256 * 16 envs, default parameters) both halves run
 * in ONE kernel (csrc/meshenv_fused.h: the same 16 envs per workgroup, the actor's forward after the step's barrier);
 * otherwise two launches are issued.  Results are identical either way.  Env and actor must share device and stream
 * (MESHENV_E_STATE otherwise: the stream is what orders the two halves). */
int meshenv_step_actor(MeshEnv *h, MeshActor *a, const float *actions_dev, float *obs_dev, double *reward_dev, uint8_t *done_dev,
                       uint8_t *complete_dev, float *terminal_obs_dev, int auto_reset, int sample, uint64_t seed, uint64_t counter,
                       float *actions_next_dev, float *eps_out_dev);

/*
 * MeshGeneration.extract_samples_2(meshes, n_neighbor, n_radius, radius, index, quality_threshold), general/mesh.py:1438-1489
 * (the data-preparation step of the ANN scripts: general/EBRD.py:414, 579 with (2, 3, radius 4, index 1) and
 * general/post_processing.py:532 with (3, 3, radius 6, index 5)) for the generated mesh of EVERY env in one launch
 * (csrc/meshenv_samples.h): which = 0 the running episode's elements, 1 the archived (finished) episode.  Needs
 * log_capacity > 0.  Two calls:
 *   1. offsets_dev = NULL: count_dev [n_envs] int64 = samples of every env, status_dev [n_envs] uint8 (0 ok, 1 masked out,
 *      2 log overflow, 3 a vertex with more than 16 neighbours, 4 more than 32 close vertices in a sector: count 0);
 *   2. offsets_dev [n_envs + 1] int64 = where env k's samples start / end (the exclusive prefix sum of the counts; count_dev
 *      and status_dev as the first call left them), samples_dev
 *      [total][2 * (2 n_neighbor + n_radius)], outputs_dev [total][2], types_dev [total] float64: the three lists the
 *      reference returns, env after env, in the reference's order (element, corner, right path, sector tuple, left path).
 * Values: bit-identical to the reference's except the entries of the synthetic sector points, whose coordinates pass
 * through cos / sin of an unquantised angle (<= 2 ulp); reinforcementlearning4meshgeneration_amd.samples is the host
 * restatement (tests/test_gpu_samples.py, tests/test_samples_cpu.py).
 */
int meshenv_extract_samples(MeshEnv *h, int which, const uint8_t *mask_dev, int n_neighbor, int n_radius, double radius, int index,
                            double quality_threshold, int64_t *count_dev, uint8_t *status_dev, const int64_t *offsets_dev,
                            double *samples_dev, double *outputs_dev, double *types_dev);

/* T vector steps of the closed loop in ONE launch (csrc/meshenv_fused.h, k_step_group_actor_T): the workgroups of the
 * fused kernel never talk to each other, so each loops [step its 16 envs -> actor forward] T times on its own -- one launch
 * ramp per T steps, and the workgroups drift apart instead of waiting, every step, for the CU with the most extractions.
 *   actions_dev   [T+1][n][3]  slice 0 = the actions of the first step (input); slice t + 1 = the policy's actions on
 *                              the observations of step t (output: slice T feeds the next call)
 *   obs_dev [T][n][18], reward_dev [T][n], done_dev [T][n], complete_dev [T][n], terminal_obs_dev [T][n][18] (nullable),
 *   eps_out_dev [T][n][3] (nullable): slice t = what meshenv_step_actor would have written at step t; the noise counter of
 *   step t is counter + t.
 * Bit-identical to T calls of meshenv_step_actor(counter + t) on the slices (that is also what runs when the batch is not
 * on the fused kernel: T x the one- or two-launch path).  The reference has no counterpart: its loop is SB3's
 * collect_rollouts calling policy and env.step() alternately (rl/baselines/RL_Mesh.py:186-228). */
int meshenv_step_actor_multi(MeshEnv *h, MeshActor *a, int T, float *actions_dev, float *obs_dev, double *reward_dev, uint8_t *done_dev,
                         uint8_t *complete_dev, float *terminal_obs_dev, int auto_reset, int sample, uint64_t seed, uint64_t counter,
                         float *eps_out_dev);

/*
 * Test hook: evaluate one device geometry primitive on n items (in_per_item doubles each) and copy the results
 * back, so the parity tests can compare the device primitives with the oracle's one by one.
 *   what 0 round(python float, 4)   1 round(np.float64, 4)   2 Vertex.to_find_clockwise_angle (6 doubles: s, p1, p2)
 *        3 Segment.is_cross (8 doubles: a1, a2, b1, b2)  4 collinearity class, fast + 2*exact (2 doubles: c, d)
 *        5 round(np.float32, 4)      6 Point2D.distance_to (4 doubles)   7 sqrt_pos == sqrt (1 double)
 *        8 quantised clockwise angle, fast form against exact (2 doubles: c, d): 0 equal, 1 guard band, 2 mismatch
 *        9 / 10 x / y of a front-smoother vertex construction (9 doubles: which = 0 middle_vertex, 1 side_vertex,
 *               2 indention_vertex (general/mesh.py:805-909), 3 Mesh.estimate_4th_vertex (factor, suggest_dist or < 0);
 *               vertex, p1, p2; angle; dist), NaN where it is undefined
 */
int meshenv_selftest(int device, int what, int n, int in_per_item, const double *in_host, double *out_host);

#ifdef __cplusplus
}
#endif
#endif /* MESHENV_H */
