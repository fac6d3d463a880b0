/*
 * mrca_env.h -- C ABI of the MI355X-native multi-robot collision-avoidance environment.
 *
 * This is the drop-in boundary for ONE path of Acmece/rl-collision-avoidance: the per-robot
 * ROS/Stage simulation behind the `StageWorld` class.  The reference has no C plugin API for
 * this path; what a caller binds to is (a) the StageWorld method set and (b) the stageros topic
 * contract underneath it.  Every entry point below names the reference interface it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream);
 *   - every pointer suffixed _dev is DEVICE memory on the env's GPU, everything else is host;
 *   - all calls are asynchronous on `stream`, there is no hidden synchronisation;
 *   - return value: 0 (MRCA_OK) or a negative mrca_status; mrca_last_error() gives the text
 *     for the calling thread;
 *   - one env per GPU, not thread-safe per handle;
 *   - the env owns one contiguous device arena (either caller-provided or hipMalloc'ed) laid out
 *     structure-of-arrays, every field 256-byte aligned; mrca_get_field() exposes the fields
 *     zero-copy.
 *
 * Robot numbering: robot n lives in world n / robots_per_world with local index
 * n % robots_per_world (the reference's `index` = MPI rank, ppo_stage1.py:168).  Robots only
 * interact (collide, see each other's bodies with the lidar) inside their world.
 */
#ifndef MRCA_ENV_H
#define MRCA_ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 6): MRCA_F_HIT_BITS -- what a beam hit is a bit plane of its own, MRCA_F_SCAN_RING holds plain ranges (5: the sign
 * bit of a ring entry); mrca_step_many's run-ahead schedule (chains < 0: the chained one).
 * 5 (round 5): mrca_policy_tail takes fc1_b_dev (may be NULL) after h1_dev; added since 4, all additive: mrca_step_worlds,
 * mrca_move_worlds, mrca_observe_worlds, mrca_step_many, mrca_adam_step, mrca_policy_heads(_backward), mrca_relu_cat(_backward), mrca_rollout_rows +
 * mrca_rollout_store_state / _outcome, status bits for mrca_check.  4: the frame history became a ring of raw scans (MRCA_F_SCAN_RING, MRCA_F_RING_HEAD). */
#define MRCA_ABI_VERSION 6

typedef struct mrca_env mrca_env; /* opaque */

enum mrca_status {
    MRCA_OK = 0,
    MRCA_ERR_INVALID = -1,     /* bad argument / config */
    MRCA_ERR_HIP = -2,         /* a HIP runtime call failed */
    MRCA_ERR_NOMEM = -3,       /* arena too small / allocation failed */
    MRCA_ERR_UNSUPPORTED = -4  /* e.g. robots_per_world > 64 together with group-synchronous episodes */
};

/* episode structure */
enum mrca_auto_reset {
    MRCA_AUTO_NONE = 0,  /* circle_test.py:36-83 : nothing resets, first terminal event latched  */
    MRCA_AUTO_ROBOT = 1, /* ppo_stage1.py:51-58  : each robot starts a new episode when done     */
    MRCA_AUTO_GROUP = 2  /* ppo_stage2.py:72-107 : done robots idle until their whole group done */
};

/* per-local-index rule for reset_pose / generate_goal_point */
enum mrca_reset_mode {
    MRCA_RESET_TABLE = 0, /* model/utils.py:6-63 tables (stage_world2.py:213-214, circle_world.py:205) */
    MRCA_RESET_DISC = 1,  /* stage_world1.py:251-274                                                  */
    MRCA_RESET_REGION = 2 /* stage_world2.py:250-287 (robots 34..43)                                   */
};

/* get_reward_and_terminate's third return value (stage_world1.py:194,201,208) */
enum mrca_result { MRCA_RESULT_NONE = 0, MRCA_RESULT_REACH = 1, MRCA_RESULT_CRASH = 2, MRCA_RESULT_TIMEOUT = 3 };

/* device-resident fields; shapes in [] with N = num_worlds*robots_per_world, B = beams, F = frames */
enum mrca_field {
    MRCA_F_POSE = 0,      /* f32 [N,3]   get_self_stateGT   stage_world1.py:116 ; base_pose_ground_truth stageros.cpp:575 */
    MRCA_F_SPEED,         /* f32 [N,2]   get_self_speed     stage_world1.py:143 ; odom twist stageros.cpp:543-558        */
    MRCA_F_SPEED_GT,      /* f32 [N,2]   get_self_speedGT   stage_world1.py:119 ; stageros.cpp:585-590                   */
    MRCA_F_GOAL,          /* f32 [N,2]   env.goal_point     stage_world1.py:173                                          */
    MRCA_F_INIT_POSE,     /* f32 [N,3]   env.init_pose      stage_world1.py:263                                          */
    MRCA_F_SCAN,          /* f32 [N,B]   base_scan ranges   stageros.cpp:479-516 : the newest scan, a COPY out of
                           *             MRCA_F_SCAN_RING -- current after every call with lazy_obs = 0, otherwise after
                           *             mrca_materialize(MRCA_VIEW_SCAN)                                                    */
    MRCA_F_OBS,           /* f32 [N,F,B] get_laser_observation x frame deque  stage_world1.py:122-140, ppo_stage1.py:59-60,87-89
                           *             x / 6 - 0.5 of the last F scans in deque order (oldest frame first), made from
                           *             MRCA_F_SCAN_RING: current after every call with lazy_obs = 0, otherwise after
                           *             mrca_materialize(MRCA_VIEW_OBS) */
    MRCA_F_LOCAL_GOAL,    /* f32 [N,2]   get_local_goal     stage_world1.py:155-160                                      */
    MRCA_F_REWARD,        /* f32 [N]     get_reward_and_terminate[0]  stage_world1.py:180-211                            */
    MRCA_F_DONE,          /* u8  [N]     get_reward_and_terminate[1]                                                     */
    MRCA_F_RESULT,        /* u8  [N]     get_reward_and_terminate[2] as enum mrca_result                                 */
    MRCA_F_FIRST_RESULT,  /* u8  [N]     first terminal event since reset (success-rate metric)               */
    MRCA_F_CRASHED,       /* u8  [N]     get_crash_state    stage_world1.py:149 ; is_crashed stageros.cpp:562-564        */
    MRCA_F_LIVE,          /* u8  [N]     `liveflag`         ppo_stage2.py:52,85-86                                       */
    MRCA_F_FRESH,         /* u8  [N]     1 where the last call started a new episode for the robot                      */
    MRCA_F_T,             /* i32 [N]     the `step` argument of get_reward_and_terminate (ppo_stage1.py:57,118)          */
    MRCA_F_EPISODE,       /* i32 [N]     episode counter (RNG stream position)                                          */
    MRCA_F_PREV_DIST,     /* f32 [N]     self.distance      stage_world1.py:176-177,185-186                              */
    MRCA_F_SCAN_RING,     /* f32 [N,F,B] the last F scans of every robot (RAW ranges, 0..6 m, never negative) as a ring (ABI 4): a
                           *             tick writes ONE row per robot -- the tick's only per-beam store -- logical frame f
                           *             (0 = oldest) of robot n is slot (head[n] + 1 + f) mod F.  (ABI 3 kept a ring of
                           *             NORMALISED frames next to MRCA_F_SCAN: every beam was stored twice.  ABI 4-5 kept
                           *             what a beam hit in the SIGN BIT of its entry; since ABI 6 that is MRCA_F_HIT_BITS
                           *             and an entry is the plain range.)                                               */
    MRCA_F_RING_HEAD,     /* u8  [N]     slot of robot n's newest scan in MRCA_F_SCAN_RING                                */
    MRCA_F_HIT_BITS,      /* u64 [N,F,B/64] (ABI 6) what every beam of the ring hit, one bit per beam, slot by slot like the
                           *             ring: bit (b & 63) of word b >> 6 set = beam b returned from ANOTHER ROBOT
                           *             (ranger_return 0.5, stage1.world:95 -- what stageros turns into LaserScan intensity
                           *             0, stageros.cpp:506), clear = the floorplan or nothing (range 6.0)              */
    MRCA_F_COUNT
};

typedef struct mrca_config {
    int32_t abi_version;       /* MRCA_ABI_VERSION */
    int32_t device;            /* HIP device ordinal */
    int32_t num_worlds;
    int32_t robots_per_world;  /* 1..64: one wavefront per world; > 64: per-robot threads + per-tick broad phase */
    int32_t beams;             /* 512 (stage1.world:14); 64..1024, multiple of 64 */
    int32_t frames;            /* 3 (LASER_HIST, ppo_stage1.py:24); 1..8 */
    /* occupancy grid shared by all worlds; cells outside it are free */
    int32_t map_width, map_height, map_words_per_row;   /* cells; at most 16384 per side */
    float map_cell, map_x0, map_y0; /* cell edge [m]; world coords of the lower-left corner */
    const uint32_t* map_bits;       /* host, [map_height][map_words_per_row], bit b of word w = column 32w+b */
    /* reward / episode rules (stage_world1.py:180-211 and the stage2 / circle variants) */
    int32_t timeout;        /* 150 / 200 / 10000 */
    float w_thresh;         /* 1.05 / 1.05 / 0.7 */
    int32_t pre_dist_zero;  /* stage_world2.py:170-171, circle_world.py:166-167 quirk */
    int32_t auto_reset;     /* enum mrca_auto_reset */
    uint64_t seed;          /* Philox key */
    /* per local index (robots_per_world entries each); NULL = all MRCA_RESET_DISC / zeros */
    const int32_t* reset_mode;
    const int32_t* goal_mode;
    const float* init_table;  /* [R,3] */
    const float* goal_table;  /* [R,2] */
    const int32_t* group_id;  /* [R] in 0..15, model/utils.py:83 */
    /* Fidelity mode (ABI 2).  0 = robots collide when their 0.44 x 0.38 rectangles overlap (exact, resolution-free).
     * res > 0 [m] = Stage's rule on a raster of `res` metres (worlds/stage1.world:3: 0.2): robots collide when their
     * OUTLINES SHARE A RASTER CELL, i.e. up to one cell apart.  res >= 0.1; not with robots_per_world > 64. */
    float collision_raster;
    /* ABI 3 / 4.  0: MRCA_F_SCAN and MRCA_F_OBS are brought up to date by every mrca_reset / mrca_step* -- what a caller
     * written against ABI 2 expects; since round 6 the stepping calls' ray cast writes the two views of its robots itself
     * (8 kB per robot beside its 2 kB ring row: 180 instead of 247 M agent-steps/s at 4096 robots; as a pass of its own behind
     * every ray cast it was 131).  1: only mrca_materialize() does that; callers that read MRCA_F_SCAN_RING +
     * MRCA_F_RING_HEAD (mrca_lidar_features does) never pay for it. */
    int32_t lazy_obs;
    /* ABI 3, fidelity.  0: a robot that is not acting any more (MRCA_F_LIVE = 0: finished, waiting for its group,
     * ppo_stage2.py:72-107) is commanded (0, 0), and MRCA_F_SPEED restarts at 0 with every episode.  1: what stageros
     * does -- SetSpeed persists (stageros.cpp:272-280; the watchdog of :466-471 is global and the other robots keep it
     * fed): such a robot keeps driving at the last command it was given, still collides and is still seen, and the odom
     * twist get_self_speed reads (stage_world1.py:106-108,146-147) survives reset_pose's teleport. */
    int32_t hold_velocity;
} mrca_config;

/* Bytes of device arena an env with this config needs (256-byte granules). */
int mrca_arena_bytes(const mrca_config* cfg, size_t* bytes_out);

/* Replaces: StageWorld.__init__ (stage_world1.py:17-84) + the stageros process it talks to
 * (stageros.cpp:311-437).  arena_dev may be NULL (the library allocates) or caller-owned device
 * memory of at least mrca_arena_bytes() (e.g. a torch tensor's data_ptr). */
/* (Beside the arena an env with robots_per_world <= 64 allocates the run-ahead ring of mrca_step_many -- 53 B per robot and
 * slot, at most 255 slots / 256 MB; 55 MB at 4096 robots -- one stream + events for the move launches and one per world range in
 * use.  If that allocation fails the env works without it: mrca_step_many then runs the chained schedule.  Every env also keeps a
 * 560-byte copy of its internal view in device memory: the two kernels of the tick read it there instead of taking it as a
 * kernel argument -- a launch with <= 104 bytes of arguments costs the host 2.6 us, one with more 3.2 - 3.5.) */
int mrca_create(const mrca_config* cfg, void* arena_dev, size_t arena_bytes, mrca_env** env_out);
int mrca_destroy(mrca_env* env);

/* Replaces: reset_world / reset_pose / control_pose / generate_goal_point
 * (stage_world1.py:162-177,213-223,237-249; stageros.cpp:260-296).
 *   mask_dev  u8[N] or NULL (= all robots): which robots begin a new episode
 *   poses_dev f32[N,3] or NULL: teleport targets (control_pose); NULL = sample per reset_mode
 *   goals_dev f32[N,2] or NULL: goal points; NULL = sample per goal_mode
 * Re-casts the lidar for the masked robots and fills all F frames with the fresh scan. */
int mrca_reset(mrca_env* env, const uint8_t* mask_dev, const float* poses_dev, const float* goals_dev, void* stream);

/* Replaces one trip round the loop body of ppo_stage1.py:75-91:
 *   control_vel (cmd_vel -> SetSpeed, stageros.cpp:272-280) for every robot,
 *   one Stage tick (UpdateWorld, stageros.cpp:445-449),
 *   get_reward_and_terminate(step), get_laser_observation, get_local_goal, get_self_speed.
 * actions_dev f32[N,2] = (v, omega) already clipped by the caller (ppo_stage1.py:170, model/ppo.py:75). */
int mrca_step(mrca_env* env, const float* actions_dev, void* stream);

/* The same tick for ONE world sharded over several GPUs (a single circle of 50 000 robots): every
 * rank holds the whole world, feeds the commands of ALL robots (one all-gather of act[N,2] per tick -- the exchange
 * step) and advances all of them -- the ordered collision pass needs every provisional pose, and identical arithmetic
 * keeps the replicas bit-identical -- but casts the lidar only for its own robots
 * [first_robot, first_robot + num_robots): scan / obs / local_goal of the other robots are left untouched.  The
 * replicated move phase bounds the speed-up.  Measured (round 6, one rank's share timed on one GPU,
 * profiles/r06_b_bigworld_shards8.jsonl): the move phase is ~20 % of a tick at every size -- 43.7 of 143 us at
 * 50 000 robots, 98 of 474 at 200 000, 244 of 1190 at 500 000 -- so 8 GPUs project to 2.2x / 3.1x / 3.1x (a jam: 2.5 / 3.8 /
 * 4.3x) and the ceiling, ~3.3x, does not rise with the world: there is no size from which the replicated design alone reaches
 * 6x.  Passing it takes a sharded move phase (spatial slabs + the transitive closure of each slab's lower-indexed neighbours
 * and a second exchange per tick); one GPU does 400 M agent-steps/s at 500 000 robots as it is. */
int mrca_step_slice(mrca_env* env, const float* actions_dev, int32_t first_robot, int32_t num_robots, void* stream);

/* The same tick for the worlds [first_world, first_world + num_worlds) ONLY: their robots advance and are observed, every
 * other world is left exactly as it is.  Worlds never interact (one `stageros` process per world in the reference:
 * stage_world1.py:17-84), so a caller may step disjoint world ranges on DIFFERENT streams -- the latency-bound move launch
 * of one range then runs next to the ray cast of another (bench.py --schedule chained) -- or give the policy of
 * one range the time the simulator spends on the other.  actions_dev is still f32[N,2] indexed by robot; only the rows of the
 * range are read.  Calls on overlapping ranges must be ordered by the caller (same stream, or events).
 * robots_per_world > 64: only the full range (MRCA_ERR_UNSUPPORTED otherwise: one world's move phase is one launch chain). */
int mrca_step_worlds(mrca_env* env, const float* actions_dev, int32_t first_world, int32_t num_worlds, void* stream);
/* ... and its two launches one by one: mrca_move_worlds = control_vel + the Stage tick + get_reward_and_terminate + episode
 * bookkeeping of the range (stage_world1.py:225-234,180-211; stageros.cpp:445-449), mrca_observe_worlds = the ray cast at the
 * poses it left + get_local_goal (stageros.cpp:479-516, stage_world1.py:126-160) (+ the two views with lazy_obs = 0).
 * mrca_step_worlds(r) == mrca_move_worlds(r) then mrca_observe_worlds(r) on one stream.  What the split is for: a caller with
 * two streams and an event can hold two world ranges half a tick apart -- range B's move launch goes out when range A's has
 * finished, i.e. next to A's ray cast, tick after tick (bench.py --chains 2: the schedule, DESIGN.md 5.9: what it buys). */
int mrca_move_worlds(mrca_env* env, const float* actions_dev, int32_t first_world, int32_t num_worlds, void* stream);
int mrca_observe_worlds(mrca_env* env, int32_t first_world, int32_t num_worlds, void* stream);

/* num_ticks ticks of every world from ONE call, with commands that are already on the device (a scripted scenario, a replayed
 * log, the benchmark's action pool): tick k (k = 0 .. num_ticks - 1) takes actions_dev[(first_tick + k) % num_actions], each
 * f32[N,2] as for mrca_step.  Equal, field for field, to num_ticks calls of mrca_step.
 * Because every command is known up front, tick k + 1's move launch does not depend on tick k's ray cast -- only on what the
 * ray cast READS of a move launch: pose, head record, goal, fresh flag, outline.  The call therefore lets the move launches
 * RUN AHEAD (since round 6): tick 0's goes out on `stream`, the others on a stream the env owns, back to back, each writing
 * those five fields into a slot of its own (a ring of <= 255 slots beside the arena, 53 B per robot and slot, at most 256 MB; the call's last
 * tick writes the env's own fields, so MRCA_F_POSE etc. are current when the call's work is done and never in between); the
 * ray casts of chains = P contiguous world ranges run on P streams (range 0 on `stream`, the others on streams the env created
 * in mrca_create for P <= 2, at first use beyond), tick after tick, each behind the event "tick k's move launch is through".  `stream` waits for
 * all of them before the call's work counts as done, and they for everything queued on `stream` before the call.  What a tick
 * then costs is its ray casts alone.  The host only enqueues (no synchronisation, nothing spins on the
 * device); the call is capturable into a hipGraph like any other (chains > 2: call it once outside the capture first).
 * Streams: the HIP runtime maps a process's streams onto a few hardware queues, and two streams that share one run their
 * kernels one after the other (tools/queue_alias_probe.hip): the schedule stays correct and silently loses its overlap.  At the
 * FIRST call on a given `stream` (outside a capture) the env therefore warms its streams and checks, with two 40 us probe
 * kernels per pair, that its move stream and its ranges' streams really run next to `stream` and to each other, replacing the
 * ones that do not: ~1 ms, once, and the one place where a call of this library synchronises (`stream` and the env's own
 * streams).  Make that first call before a timed or latency-critical region.
 * chains = -P: round 5's schedule instead -- P chains `move, ray cast, move, ray cast ...` of one world range each, half a tick
 * apart (kept for A/B runs: tools/region_sweep.py --schedule chained).  robots_per_world > 64: one chain, in order. */
int mrca_step_many(mrca_env* env, const float* const* actions_dev, int32_t num_actions, int32_t first_tick, int32_t num_ticks,
                   int32_t chains, void* stream);

/* The reference-shaped views of the ring, for all robots (asynchronous on `stream`; needed only with lazy_obs = 1):
 * what & MRCA_VIEW_SCAN: MRCA_F_SCAN := every robot's newest scan; what & MRCA_VIEW_OBS: MRCA_F_OBS := x / 6 - 0.5
 * (stage_world1.py:140) of the ring in deque order. */
enum mrca_view { MRCA_VIEW_SCAN = 1, MRCA_VIEW_OBS = 2 };
int mrca_materialize(mrca_env* env, int32_t what, void* stream);
/* out_dev f32[N,B] := x / 6 - 0.5 of every robot's newest scan -- the ONE observation row per tick a rollout buffer that
 * stores single frames keeps (ppo_stage1.py:87-89 appends exactly this row to the deque). */
int mrca_newest_obs(mrca_env* env, float* out_dev, void* stream);

/* out_dev[i] := in_dev[i] / 6 - 0.5 (stage_world1.py:140) for `count` floats, rounded exactly as MRCA_F_OBS is -- for a caller
 * that holds rows of MRCA_F_SCAN_RING (several envs concatenated, a slice of a sharded world) and wants observations.
 * count % 4 == 0, both buffers 16-byte aligned; in place allowed. */
int mrca_normalize_scans(const float* in_dev, float* out_dev, size_t count, void* stream);

/* get_laser_observation for a StageWorld constructed with beam_num != the lidar's sample count (stage_world1.py:126-139:
 * the sparse scan is a left half picked ascending and a right half picked descending from the raw one):
 * out_dev f32[N,F,beam_num] := x / 6 - 0.5 of beams index_dev[0..beam_num) of every frame of the stack, deque order.
 * index_dev i32[beam_num], each in [0, beams): the caller's table (the reference advances a float64 index by
 * raw / beam_num per pick and truncates; mrca/vec_env.py:sparse_beam_index restates that loop). */
int mrca_sparse_obs(mrca_env* env, const int32_t* index_dev, int32_t beam_num, float* out_dev, void* stream);

/* Synchronises `stream` and reports (then clears) the env's sticky device-side status word: MRCA_OK, or MRCA_ERR_HIP with
 * mrca_last_error() saying what went wrong on the device since the last check.  Today one condition: the ordered
 * collision pass of a world with more than 64 robots ran out of its (very long) bounded wait and left a robot
 * undecided.  mrca_step itself never synchronises; call this wherever a host round trip is acceptable (end of an
 * evaluation, once per PPO update). */
int mrca_check(mrca_env* env, void* stream);

/* Zero-copy access to a field: device pointer, byte offset inside the arena and byte size. */
int mrca_get_field(mrca_env* env, int field, void** ptr_dev_out, size_t* offset_out, size_t* bytes_out);

/* Replaces: generate_train_data (model/ppo.py:122-139) -- reverse GAE scan, one thread per robot.
 *   rewards_dev f32[T,N], values_dev f32[T,N], last_value_dev f32[N], dones_dev u8[T,N]
 *   targets_dev f32[T,N], advs_dev f32[T,N] */
int mrca_gae(const float* rewards_dev, const float* values_dev, const float* last_value_dev,
             const uint8_t* dones_dev, float gamma, float lam, int32_t T, int32_t N,
             float* targets_dev, float* advs_dev, void* stream);

/* Diagnostics */
int mrca_abi_version(void);
const char* mrca_last_error(void);
/* Per-kernel timing: the move launch and the ray-cast launch of a timed step are stamped with their own BEGIN and END
 * (hipExtLaunchKernel's start / stop events: the dispatch's timestamps, what rocprofv3 reports -- event records AROUND a
 * launch read ~2.5 us longer per kernel), up to 1024 steps between reads.
 * mrca_read_timing synchronises on the last recorded event, returns the summed durations of the
 * recorded launches in milliseconds and clears the ring. */
int mrca_enable_timing(mrca_env* env, int32_t on); /* on = n > 0: time every n-th step; 0: off */
int mrca_read_timing(mrca_env* env, float* move_ms_total, float* ray_ms_total, int32_t* launches);
/* Mean microseconds an event pair on `stream` reads with nothing between its two records (`samples` pairs, each
 * synchronised): the marker time every (event, kernel, event) figure of mrca_read_timing contains once per kernel. */
int mrca_event_pair_overhead(void* stream, int32_t samples, float* us_out);
/* Rollout-path front end of the lidar actor-critic (model/net.py:19-25,37-49,57-69: Conv1d(3,32,k5,s2,p1) -> ReLU ->
 * Conv1d(32,32,k3,s2,p1) -> ReLU for the actor and the critic tower), fused into one kernel: fp32 in, fp32 MFMA
 * accumulate, the 32 x 255 intermediate never leaves the CU.
 *   obs_dev  f32[N,3,512]   the observation stacks: MRCA_F_OBS (deque order) with obs_head_dev = NULL, or a ring with
 *                           obs_head_dev = its head slots (u8[N]): the kernel then reads frame f of robot n from slot
 *                           (head[n] + 1 + f) mod 3 while staging
 *   raw_scans               0: obs_dev holds normalised observations; 1: it holds RAW ranges (MRCA_F_SCAN_RING with
 *                           MRCA_F_RING_HEAD) and the kernel applies x / 6 - 0.5 (stage_world1.py:140) while staging --
 *                           bit-identical to reading the materialised MRCA_F_OBS
 *   w1_dev   f32[2,32,3,5]  b1_dev f32[2,32]    act_fea_cv1 / crt_fea_cv1 weight and bias, tower-major
 *   w2_dev   f32[2,32,32,3] b2_dev f32[2,32]    act_fea_cv2 / crt_fea_cv2
 *   feat_dev f32[2,N,4096]  out: tower-major, each row in the flatten order of [32,128] (what act_fc1 / crt_fc1 eat)
 * frames must be 3 and beams 512 (MRCA_ERR_UNSUPPORTED otherwise). */
int mrca_lidar_features(const float* obs_dev, const uint8_t* obs_head_dev, int32_t raw_scans, int32_t n_robots,
                        int32_t frames, int32_t beams, const float* w1_dev, const float* b1_dev, const float* w2_dev, const float* b2_dev,
                        float* feat_dev, void* stream);
/* The same with the stacks addressed THROUGH A ROW TABLE instead of gathered: frames_dev is a matrix of normalised frames
 * f32[*,512] and rows_dev i32[n_samples,3] the row of each sample's three frames, oldest first.  What the PPO update reads of a
 * rollout buffer that stores ONE frame per tick (model/ppo.py:143-194 indexes obs_batch[sampler]; here the stack of (tick t,
 * robot i) is three rows of the frame store, mrca/ppo.py FrameRows): the minibatch's [n,3,512] copy -- 100 MB written and read
 * per 16 384 samples -- is never made.  Same arithmetic, same results bit for bit. */
int mrca_lidar_features_rows(const float* frames_dev, const int32_t* rows_dev, int32_t n_samples, int32_t frames, int32_t beams,
                             const float* w1_dev, const float* b1_dev, const float* w2_dev, const float* b2_dev, float* feat_dev,
                             void* stream);

/* The rest of the rollout inference behind fc1 in one kernel (model/net.py:41-55,61-70: ReLU, cat with goal and speed,
 * fc2 + ReLU of both towers, actor1 / actor2 / critic heads with sigmoid / tanh; model/ppo.py:57-82 generate_action:
 * a = mean + exp(logstd) * noise, log-density, clip to the action bounds).  fp32, exact-fp32 MFMA for fc2.
 *   h1_dev    f32[2,N,256]  act_fc1 / crt_fc1 outputs before the ReLU (tower-major)
 *   fc1_b_dev f32[2,256]    act_fc1.bias, crt_fc1.bias: added while h1 is staged -- or NULL when h1 already includes them
 *   goal_dev  f32[N,2]  speed_dev f32[N,2]          MRCA_F_LOCAL_GOAL, MRCA_F_SPEED
 *   fc2_w_dev f32[2,260,128] (input-major: act_fc2.weight^T, crt_fc2.weight^T)   fc2_b_dev f32[2,128]
 *   head_w_dev f32[128,2] (columns actor1.weight, actor2.weight)  head_b_dev f32[2]  critic_w_dev f32[128]  critic_b_dev f32[1]
 *   logstd_dev f32[2]   noise_dev f32[N,2] standard normal draws, or NULL: the deterministic mean action
 *                       (generate_action_no_sampling, model/ppo.py:84-107)
 *   lo_dev, hi_dev f32[2]   the action bounds (ppo_stage1.py:170)
 *   out: value_dev f32[N], action_dev f32[N,2] (UNclipped sample: what the rollout buffer stores), logprob_dev f32[N],
 *        scaled_dev f32[N,2] (clipped: what drives the robot), mean_dev f32[N,2] */
int mrca_policy_tail(const float* h1_dev, const float* fc1_b_dev, const float* goal_dev, const float* speed_dev, int32_t n_robots,
                     const float* fc2_w_dev, const float* fc2_b_dev, const float* head_w_dev, const float* head_b_dev,
                     const float* critic_w_dev, const float* critic_b_dev, const float* logstd_dev, const float* noise_dev,
                     const float* lo_dev, const float* hi_dev, float* value_dev, float* action_dev, float* logprob_dev,
                     float* scaled_dev, float* mean_dev, void* stream);

/* Backward pass of the same front end for the PPO update (model/ppo.py:158-192 back-propagates the loss through
 * act_fea_cv1/2 and crt_fea_cv1/2 of model/net.py:19-25 for every minibatch): given the gradient with respect to the
 * forward kernel's output it returns the gradients of both towers' convolution weights and biases; h1 is recomputed,
 * g1 never leaves the registers, per-wave partial sums are added in a fixed order (deterministic).
 *   obs_dev   f32[N,3,512]   the minibatch's observation stacks (no gradient: data)
 *   w1_dev f32[2,32,3,5]  b1_dev f32[2,32]  w2_dev f32[2,32,32,3]      the weights the forward ran with
 *   feat_dev  f32[2,N,4096]  the forward's output (its sign pattern is the second ReLU's mask)
 *   gfeat_act_dev, gfeat_crt_dev  f32[N,4096] each: dLoss / dfeat of the actor and of the critic tower (two buffers:
 *                            they are the outputs of two independent fc1 backward GEMMs)
 *   dw1_dev f32[2,32,3,5]  db1_dev f32[2,32]  dw2_dev f32[2,32,32,3]  db2_dev f32[2,32]   out (overwritten)
 *   scratch_dev              caller-owned device scratch of at least mrca_lidar_features_backward_scratch() bytes
 *                            on the CURRENT device */
int mrca_lidar_features_backward_scratch(size_t* bytes_out);
int mrca_lidar_features_backward(const float* obs_dev, int32_t n_robots, int32_t frames, int32_t beams,
                                 const float* w1_dev, const float* b1_dev, const float* w2_dev, const float* feat_dev,
                                 const float* gfeat_act_dev, const float* gfeat_crt_dev, float* dw1_dev, float* db1_dev,
                                 float* dw2_dev, float* db2_dev, void* scratch_dev, size_t scratch_bytes, void* stream);
/* ... and its row-table form (see mrca_lidar_features_rows): frames_dev f32[*,512], rows_dev i32[n_samples,3]. */
int mrca_lidar_features_backward_rows(const float* frames_dev, const int32_t* rows_dev, int32_t n_samples, int32_t frames,
                                      int32_t beams, const float* w1_dev, const float* b1_dev, const float* w2_dev,
                                      const float* feat_dev, const float* gfeat_act_dev, const float* gfeat_crt_dev, float* dw1_dev,
                                      float* db1_dev, float* dw2_dev, float* db2_dev, void* scratch_dev, size_t scratch_bytes,
                                      void* stream);

/* The loss tail of the PPO update, values AND gradients, in one launch (model/ppo.py:172-185 / :238-251: importance ratio,
 * clipped surrogate, value loss x value_coef, entropy bonus; log-density of model/utils.py:90-97) -- what PyTorch runs as
 * ~30 element-wise launches forward and as many backward per minibatch.
 *   mean_dev f32[n,2]  value_dev f32[n]  logstd_dev f32[2]      the network's outputs for the minibatch
 *   action_dev f32[n,2]  old_logprob_dev f32[n]  adv_dev f32[n]  target_dev f32[n]      the minibatch's stored rows
 *   out_dev f32[8]:  loss, policy loss, value loss, entropy, k3 estimate of KL(old || new), dloss/dlogstd[0], [1], 0
 *   gmean_dev f32[n,2] = dloss/dmean,  gvalue_dev f32[n] = dloss/dvalue      (autograd's tie rules for min / clamp)
 *   scratch_dev: caller-owned, at least mrca_ppo_loss_scratch() bytes, ZEROED ONCE before its first use (a launch leaves
 *   it ready for the next); sums are combined in a fixed order: bit-identical from run to run. */
int mrca_ppo_loss_scratch(size_t* bytes_out);
int mrca_ppo_loss(const float* mean_dev, const float* value_dev, const float* logstd_dev, const float* action_dev,
                  const float* old_logprob_dev, const float* adv_dev, const float* target_dev, int32_t n, float clip_value,
                  float value_coef, float coeff_entropy, float* out_dev, float* gmean_dev, float* gvalue_dev,
                  void* scratch_dev, size_t scratch_bytes, void* stream);

/* One optimiser step of the PPO update on flat buffers, one launch: torch.optim.Adam's rule (ppo_stage1.py:176 Adam(lr);
 * one step per minibatch, model/ppo.py:187-189) with no weight decay and no amsgrad --
 *     m <- m + (g - m)(1 - beta1),  v <- v beta2 + (1 - beta2) g g,
 *     p <- p - lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * in fp32, the bias corrections formed in double.
 *   param_dev, exp_avg_dev, exp_avg_sq_dev  f32[n]  updated in place;  grad_dev f32[n];  all four 16-byte aligned
 *   step  the number of THIS step, starting at 1 */
int mrca_adam_step(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n,
                   double lr, double beta1, double beta2, double eps, int32_t step, void* stream);

/* The three output heads of the actor-critic in the PPO update (model/net.py:47-55,61-63: mean = [sigmoid(actor1(a)),
 * tanh(actor2(a))], value = critic(c); three Linear(128, 1)), forward and backward, as row operations instead of ~25 library
 * launches per minibatch.
 *   a_dev, c_dev  f32[n,128]  the actor / critic tower's features (act_fc2 / crt_fc2 outputs after their ReLU), 16-byte aligned
 *   w_*_dev f32[128], b_*_dev f32[1]   actor1 / actor2 / critic weight rows and biases (any 4-byte alignment)
 *   relu_inputs   1: a_dev / c_dev are fc2's outputs BEFORE their ReLU -- it is applied as they are loaded, and its mask to
 *                 da_dev / dc_dev on the way back (which are then the gradients of those pre-activations); 0: as given
 *   out: mean_dev f32[n,2], value_dev f32[n] */
int mrca_policy_heads(const float* a_dev, const float* c_dev, int32_t n, const float* w_actor1_dev, const float* b_actor1_dev,
                      const float* w_actor2_dev, const float* b_actor2_dev, const float* w_critic_dev, const float* b_critic_dev,
                      int32_t relu_inputs, float* mean_dev, float* value_dev, void* stream);
/* ... and its backward pass: given dLoss/dmean (gmean_dev f32[n,2] or NULL = zero) and dLoss/dvalue (gvalue_dev f32[n] or
 * NULL) and the forward's mean_dev, the feature gradients da_dev, dc_dev f32[n,128] and dw_dev f32[387] = dW actor1[128],
 * dW actor2[128], dW critic[128], db actor1, db actor2, db critic (per-wave partial sums added in a fixed order in float64:
 * deterministic).  scratch_dev: caller-owned, at least mrca_policy_heads_backward_scratch() bytes, 16-byte aligned. */
int mrca_policy_heads_backward_scratch(size_t* bytes_out);
int mrca_policy_heads_backward(const float* a_dev, const float* c_dev, const float* mean_dev, const float* gmean_dev,
                               const float* gvalue_dev, int32_t n, const float* w_actor1_dev, const float* w_actor2_dev,
                               const float* w_critic_dev, int32_t relu_inputs, float* da_dev, float* dc_dev, float* dw_dev,
                               void* scratch_dev, size_t scratch_bytes, void* stream);
/* The same, and dz_bias_dev f32[256] = the column sums of da_dev [0,128) and dc_dev [128,256): with relu_inputs = 1 the gradients
 * of act_fc2.bias / crt_fc2.bias, the layers whose outputs a_dev / c_dev are (model/net.py:45,59) -- summed where da / dc are
 * formed, in the same fixed order, instead of by a reduction over the two matrices just written. */
int mrca_policy_heads_backward_bias(const float* a_dev, const float* c_dev, const float* mean_dev, const float* gmean_dev,
                                    const float* gvalue_dev, int32_t n, const float* w_actor1_dev, const float* w_actor2_dev,
                                    const float* w_critic_dev, int32_t relu_inputs, float* da_dev, float* dc_dev, float* dw_dev,
                                    float* dz_bias_dev, void* scratch_dev, size_t scratch_bytes, void* stream);

/* out[n,260] = [relu(h1[n,256]), goal[n,2], speed[n,2]]: F.relu(act_fc1(a)) and torch.cat((a, goal, speed), dim=-1) of
 * model/net.py:43-45 (the critic tower alike, :57-59) in one launch, and its backward dh1[n,256] = gout[:, :256] where h1 > 0
 * (goal and speed are data).  h1 / out / gout / dh1 16-byte aligned, goal / speed 8-byte. */
int mrca_relu_cat(const float* h1_dev, const float* goal_dev, const float* speed_dev, int32_t n, float* out_dev, void* stream);
int mrca_relu_cat_backward(const float* h1_dev, const float* gout_dev, int32_t n, float* dh1_dev, void* stream);
/* ... with db_dev f32[256] = the column sums of dh1: the gradient of the bias of the fc1 layer that produced h1 (model/net.py:41,57),
 * per-workgroup sums added in a fixed order in float64 (deterministic).  scratch_dev: caller-owned, at least
 * mrca_relu_cat_backward_bias_scratch() bytes, 16-byte aligned. */
int mrca_relu_cat_backward_bias_scratch(size_t* bytes_out);
int mrca_relu_cat_backward_bias(const float* h1_dev, const float* gout_dev, int32_t n, float* dh1_dev, float* db_dev,
                                void* scratch_dev, size_t scratch_bytes, void* stream);

/* The learner's rollout buffer, written by the library: what the reference appends to `buff` every step and turns into arrays
 * before the update (ppo_stage1.py:102-103; model/ppo.py:22-54 transform_buffer), kept on the device with ONE lidar frame per
 * tick (the stack at tick t shares F - 1 frames with the stack at t - 1) plus, per tick and robot, the rows of `frames` that
 * make up its stack (a robot that restarted: F times its fresh scan, ppo_stage1.py:59-60).  All pointers are device memory of
 * the caller (mrca/ppo.py RolloutBuffer); T = horizon, N / F / B = the env's robots / frames / beams. */
typedef struct mrca_rollout_rows {
    float* frames;     /* f32[T + F - 1][N][B]   rows 0 .. F-2: the older frames of the first tick's stacks (the caller's) */
    int64_t* fidx;     /* i64[T][N][F]           rows of `frames` that are the stack of (tick, robot), oldest first */
    int64_t* cur;      /* i64[N][F]              the same for the tick being stored (carried from tick to tick) */
    float* goal;       /* f32[T][N][2]  MRCA_F_LOCAL_GOAL */
    float* speed;      /* f32[T][N][2]  MRCA_F_SPEED */
    float* action;     /* f32[T][N][2]  the UNclipped sample (model/ppo.py:75) */
    float* logprob;    /* f32[T][N] */
    float* value;      /* f32[T][N] */
    float* reward;     /* f32[T][N] */
    uint8_t* done;     /* u8[T][N] */
    int32_t horizon;   /* T */
} mrca_rollout_rows;

/* Row *tick_dev of the buffer BEFORE the env steps, one launch: the newest observation frame x / 6 - 0.5 from the env's scan
 * ring into frames[t + F - 1], the stack's row indices (tick 0: rows 0 .. F-1), the env's local goal and speed, and the
 * policy's action_dev f32[N,2] / logprob_dev f32[N] / value_dev f32[N].  The row comes from DEVICE memory (int64[1]): no host
 * value enters the launch, a captured tick replays for every row of the horizon.  A counter outside [0, T) stores nothing and
 * sets a sticky status bit (mrca_check). */
int mrca_rollout_store_state(mrca_env* env, const mrca_rollout_rows* rows, const int64_t* tick_dev, const float* action_dev,
                             const float* logprob_dev, const float* value_dev, void* stream);
/* ... and AFTER the env stepped, one launch: MRCA_F_REWARD / MRCA_F_DONE into row *tick_dev, then *tick_dev += 1 (by the last
 * workgroup to finish).  ticket_dev: uint32[1] of the caller, zero before the first call (every call leaves it at zero). */
int mrca_rollout_store_outcome(mrca_env* env, const mrca_rollout_rows* rows, int64_t* tick_dev, uint32_t* ticket_dev,
                               void* stream);

#ifdef MRCA_PROFILING
/* PROFILING BUILD ONLY (csrc/build.sh --profiling -> libmrca_env_prof.so, used by tools/ablate.py); the product
 * library neither exports this symbol nor contains the switches.  Results are WRONG while any of bits 0-5 is
 * set: 1 = skip robot-robot lidar tests, 2 = skip the grid march, 8 / 16 / 32 = move kernel without its outline
 * test / collision loop / resets.  Launch-shape knobs (results unchanged):
 * bits 8-10 = k > 0: 1 << (k-1) beams per marching thread; bit 11: a dedicated preparation wave; bit 12: the beams of a
 * thread marched in lock step.  64 (results unchanged): the ray cast's phase stamps drain the memory queues first.
 * 0 restores the product path. */
int mrca_set_debug_flags(mrca_env* env, int32_t flags);
/* s_memtime ticks between the move kernel's phase stamps of the last launch, averaged over worlds: [0..7] = state loaded
 * and integrated | clearance + broad phase | patches in LDS | outline walks | ordered collision pass | reward / ballots
 * | restarts | stores drained; [8] = entry to end.  Every stamp drains the memory queues first. */
int mrca_debug_move_stamps(mrca_env* env, double* avg_ticks_out);
/* s_memtime stamps of the last ray-cast launch, lane 0 of wave 0 (builds the neighbour list, then marches) and of wave 1
 * (marches only): out[w * 7 + k] = mean over workgroups of stamp k - the workgroup's entry, k = entry | loads requested |
 * neighbour list built | beams marched | through the barrier | slab tests done | stores issued; out[14..16] = 0 (reserved:
 * s_memtime counters have unrelated origins across the chip, stamps of different workgroups cannot be compared). */
int mrca_debug_ray_stamps(mrca_env* env, double* out /* [17] */);
/* s_memtime ticks per robot a wave of the last mrca_lidar_features launch spent in conv1's tile pairs 0..3 (out[0..3]) and
 * conv2's tile pairs 0 / 1 (out[4], out[5]); out[6] = robots per wave; out[7] = shader clock during the loop [GHz].  See
 * tools/fwd_phases.py. */
int mrca_debug_fwd_stamps(double* out /* [8] */);
/* the same for the last mrca_lidar_features_backward launch: out[0..6] = ticks per item in: scan staged | gradient rows staged |
 * conv1 recompute | conv2 wgrad | conv2 dgrad | ReLU mask | conv1 wgrad; out[8] = items per wave; out[9] = shader clock [GHz].
 * See tools/bwd_phases.py. */
int mrca_debug_bwd_stamps(double* out /* [10] */);
#endif

#ifdef __cplusplus
}
#endif
#endif /* MRCA_ENV_H */
