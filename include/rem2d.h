/*
 * rem2d.h -- C ABI of the MI355X-native batched 2D rigid-body stepper (librem2d.so).
 *
 * Drop-in boundary for ONE path of FrankVeenstra/gym_rem2D: the world.Step() hot loop inside
 * Modular2D.step()/reset().  In the reference that path crosses an FFI into pybox2d
 * (Box2D==2.3.10, SWIG); each entry point below names the pybox2d calls / reference lines it
 * replaces, batched over N independent worlds (one creature each).
 *
 * Conventions: plain C, no torch types.  Every function returns 0 on success or a negative
 * REM2D_E_* code; rem2d_last_error() gives the text (thread-local).  Nothing throws across
 * the boundary.  `state` is caller-owned device memory (e.g. one torch.uint8 tensor); the
 * handle owns only its scratch.  One handle = one device = one stream at a time; distinct
 * handles are independent.  No function synchronises the device except the *_read helpers.
 *
 * Data layout in HBM: structure of arrays, one 4- or 8-byte element per (creature, lane),
 * lane fastest.  One lane = one rigid body (+ the revolute joint to its parent + the
 * controller of the node that created it); `lanes` (2..64, power of two) consecutive lanes
 * of one 64-wide wavefront form one creature, so a wave steps 64/lanes creatures in lockstep.
 */
#ifndef REM2D_H
#define REM2D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REM2D_ABI_VERSION 11 /* 2: + rem2d_worlds_step(_ex), rem2d_tree_diversity, rem2d_compile_lsystem; 3: + rem2d_world_set_tiles;
                               4: + rem2d_world_set_tile_shape, rem2d_plan_tiles_shape; 5: + rem2d_world_adopt; 6: + rem2d_groups_step(_ex), rem2d_capacity;
                               7: + rem2d_worlds_launch_info; 8: + rem2d_world_set_option / get_option (the library reads no environment variable), rem2d_world_set_order, rem2d_selftest_scalar;
                               9: + rem2d_build_id, rem2d_mutate_trees, count-only compilers (out == NULL); worlds of tile shapes 0 and 2 are refused in one launch;
                               10: REM2D_OPT_FUSE_VELPOST = 2 (the step train, the new default), REM2D_ERR_HANDOVER, rem2d_worlds_launch_info reports 2 for it;
                               11: + rem2d_world_handover_failures, REM2D_OPT_TRAIN_FAULT; rem2d_world_step_time_ms counts env-steps, not brackets */

enum {
    REM2D_OK = 0,
    REM2D_E_INVALID = -1, /* bad argument */
    REM2D_E_HIP = -2,     /* HIP runtime error (text in rem2d_last_error) */
    REM2D_E_STATE = -3,   /* call order (e.g. step before set_terrain / reset) */
    REM2D_E_NOMEM = -4
};

/* flags */
#define REM2D_FLAG_CONTINUOUS 1u         /* b2World continuousPhysics (SolveTOI) */
#define REM2D_FLAG_SLEEP_RESET_ALWAYS 2u /* b2Body::SetAwake(true) always zeroes sleepTime */
#define REM2D_FLAG_NO_SLEEP 4u           /* b2World(doSleep=False) */
/* evaluate() leaves its loop once an individual's fitness is final (REM2D_main.py:366-372: reward < -10 or
 * > ENV_LENGTH).  With this flag a wavefront whose creatures are ALL in that state (REM2D_F_FROZEN) is not
 * stepped any more; fitness is unaffected, the bodies of such creatures simply stop where they were. */
#define REM2D_FLAG_SKIP_FROZEN 8u
/* Launch shape, no effect on results: the position-iteration kernel deals the creatures to the wavefronts anew in every
 * step -- those that used all 60 position iterations in the last step (the same ~12 % from step to step) share wavefronts
 * instead of keeping 70 % of them busy for 60 iterations; since ABI v8 the velocity tiles go through the same creature order
 * (tile slot e = creature order[e]), so the one-launch form works with it.  The dealing is by atomics, i.e. in arrival order:
 * it destroys the static schedule order inside the tiles, which is why round 4's policy is the STABLE re-ordering of
 * REM2D_OPT_REBALANCE / rem2d_world_set_order instead (gym_rem2d_amd.env.BatchedModular2D); the flag stays as an option. */
#define REM2D_FLAG_RETILE 16u

#define REM2D_MAX_LANES 64
/* Capacity of one body's contact bookkeeping.  Box2D has no such cap (Modular2DEnv.py:634 solves whatever touches); the
 * library is built twice from the same source: librem2d.so with the slots below and librem2d_wide.so (-DREM2D_WIDE) with
 * 32 pair slots / 12 solver slots, into which the host re-runs the (rare) creatures that set an overflow bit -- see
 * gym_rem2d_amd.evaluate.run_episode.  rem2d_capacity() reports what a loaded library was built with; the state arena of
 * a wide world is laid out for its own slot count. */
#ifdef REM2D_WIDE
#define REM2D_CONTACT_SLOTS 32
#define REM2D_SOLVER_SLOTS 12
#else
#define REM2D_CONTACT_SLOTS 24 /* broadphase pair slots per body */
#define REM2D_SOLVER_SLOTS 6  /* touching contacts per body that enter the solver */
#endif

typedef struct rem2d_world rem2d_world;

typedef struct {
    int32_t n_envs; /* creatures (independent b2Worlds) */
    int32_t lanes;  /* lanes per creature: 2, 4, 8, 16, 32 or 64 */
    uint32_t flags; /* REM2D_FLAG_* */
    int32_t device; /* HIP device ordinal */
} rem2d_world_cfg;

/* Morphology upload: DEVICE pointers to [n_envs*lanes] arrays (lane fastest).  Lane 0 of a
 * creature is the root (no joint).  What each field replaces in the reference:
 *   shape,hx,hy,x,y,angle : world.CreateDynamicBody(position, angle, fixtures=fixtureDef(
 *                           polygonShape(box=(hx,hy)) | b2CircleShape(radius=hx), density=1,
 *                           friction=0.1)) -- simple_module.py:286-298, circular_module.py:191-202
 *   parent,ax..by,torque,lower,upper : world.CreateJoint(revoluteJointDef(bodyA, bodyB,
 *                           localAnchorA, localAnchorB, enableMotor, enableLimit, maxMotorTorque,
 *                           lowerAngle, upperAngle)) -- module_utility.py:19-32
 *   jround                : packed solver schedule derived from b2World::Solve's island joint order
 *                           (gym_rem2d_amd/compiler.py joint_rounds / pipeline_schedule):
 *                           bits 0-7 joint round, 8-15 contact slot, 16-23 pipeline period
 *   amp,phase,freq,offset,istate : node.controller -- Controller/m_controller.py:5-21 */
typedef struct {
    const int32_t *shape; /* 0 none, 1 box, 2 circle */
    const float *hx, *hy, *x, *y, *angle;
    const int32_t *parent; /* lane of the parent body inside the creature, -1 = none */
    const int32_t *jround;
    const float *ax, *ay, *bx, *by, *torque, *lower, *upper;
    const double *amp, *phase, *freq, *offset, *istate;
} rem2d_morph;

/* State fields that can be viewed in place (rem2d_world_field).  Per-lane fields have
 * n_envs_padded*lanes elements, per-creature fields n_envs_padded, contact fields
 * REM2D_CONTACT_SLOTS * n_envs_padded*lanes (slot-major). */
enum {
    /* per lane, float */
    REM2D_F_PX = 0, REM2D_F_PY, REM2D_F_ANG, REM2D_F_VX, REM2D_F_VY, REM2D_F_W, REM2D_F_SLEEPT,
    REM2D_F_HX, REM2D_F_HY, REM2D_F_INVM, REM2D_F_INVI,
    REM2D_F_FATLX, REM2D_F_FATLY, REM2D_F_FATUX, REM2D_F_FATUY,
    REM2D_F_JAX, REM2D_F_JAY, REM2D_F_JBX, REM2D_F_JBY, REM2D_F_JTORQUE, REM2D_F_JLOWER, REM2D_F_JUPPER,
    REM2D_F_JIMPX, REM2D_F_JIMPY, REM2D_F_JIMPZ, REM2D_F_JMOTORIMP, REM2D_F_JMOTORSPEED,
    /* per lane, int32 */
    REM2D_F_SHAPE, REM2D_F_PARENT, REM2D_F_JROUND, REM2D_F_AWAKE, REM2D_F_JLIMIT, REM2D_F_CCOUNT,
    /* per lane, double */
    REM2D_F_CAMP, REM2D_F_CPHASE, REM2D_F_CFREQ, REM2D_F_COFFSET, REM2D_F_CISTATE,
    /* per contact slot x lane */
    REM2D_F_CEDGE, REM2D_F_CINFO, REM2D_F_CKEY0, REM2D_F_CKEY1, REM2D_F_CN0, REM2D_F_CN1, REM2D_F_CT0,
    REM2D_F_CT1,
    /* per creature */
    REM2D_F_WOD /* f64 */, REM2D_F_FITNESS /* f64 */, REM2D_F_REWARD /* f32 */, REM2D_F_DONE /* i32 */,
    REM2D_F_EVERDONE /* i32 */, REM2D_F_FROZEN /* i32 */, REM2D_F_STEPS /* i32 */, REM2D_F_INVDT0 /* f32 */,
    REM2D_F_NEWFIX /* i32 */, REM2D_F_ERR /* i32 */, REM2D_F_POSITERS /* i32 */, REM2D_F_TOIEVENTS /* i32 */,
    REM2D_F_COUNT
};
enum { REM2D_DT_F32 = 0, REM2D_DT_I32 = 1, REM2D_DT_F64 = 2 };

/* error bits in REM2D_F_ERR */
#define REM2D_ERR_PAIR_OVERFLOW 1   /* more than REM2D_CONTACT_SLOTS fat-AABB pairs on a body */
#define REM2D_ERR_SOLVER_OVERFLOW 2 /* more than REM2D_SOLVER_SLOTS touching contacts on a body */
#define REM2D_ERR_HANDOVER 4        /* step train (REM2D_OPT_FUSE_VELPOST = 2): a block's previous step was published from another XCD, or a
                                      wait for it ran into its 2 s limit -- the creature's state is not to be trusted.  NOT a capacity
                                      problem: the remedy is the same creature on per-step launches (REM2D_OPT_FUSE_VELPOST = 1) in the
                                      same build, not the wide build; rem2d_world_handover_failures counts them per world */

int rem2d_abi_version(void);
/* Identity of this build: the hash of the library's sources (gym_rem2d_amd/csrc/, this header) and compile flags that the
 * builder passed in (-DREM2D_BUILD_ID), "unidentified" for a hand-made build.  The Python loader recomputes it from the sources
 * beside the library and refuses a mismatch; bench.py prints it.  (b2World has no counterpart: pybox2d's wheel is versioned by
 * pip, /root/reference/requirements.txt:1.) */
const char *rem2d_build_id(void);
const char *rem2d_last_error(void);
/* REM2D_CONTACT_SLOTS / REM2D_SOLVER_SLOTS of this build (either pointer may be NULL) */
int rem2d_capacity(int32_t *contact_slots, int32_t *solver_slots);
/* How a step of these worlds (one step group: what rem2d_worlds_step would take) is launched -- for tools that name kernels
 * (bench.py, profiles); results never depend on it.  tile_shape: 3 / 1 / 0 = 64 / 128 / 256 bodies per tile of the velocity
 * kernel; fused_velpost: 2 = the step train (rem2d_step_train_kernel: all steps of a call and all phases of a step in one launch),
 * 1 = the velocity tiles and the position iterations of a 64-lane block share one launch per step (rem2d_velpost_kernel: pre ->
 * velpost -> toi_heavy), 0 = rem2d_vel4_kernel and rem2d_post_multi_kernel.  Either pointer may be NULL. */
int rem2d_worlds_launch_info(rem2d_world *const *worlds, int32_t n_worlds, int32_t *tile_shape, int32_t *fused_velpost);

/* Bytes of device memory the caller must provide for a world of this shape. */
size_t rem2d_state_bytes(const rem2d_world_cfg *cfg);
int32_t rem2d_padded_envs(const rem2d_world_cfg *cfg);

/* Box2D.b2World() for n_envs worlds (Modular2DEnv.py:144,572): gravity (0,-10), sleeping,
 * warm starting and continuous physics as pybox2d defaults (flags select variants). */
int rem2d_world_create(const rem2d_world_cfg *cfg, void *state_dev, size_t state_bytes, rem2d_world **out);
/* world teardown (Modular2DEnv.py:175-186 _destroy + garbage-collected b2World) */
int rem2d_world_destroy(rem2d_world *w);

/* _generate_terrain's world.CreateStaticBody(fixtures=fd_edge / fd_polygon) calls
 * (Modular2DEnv.py:226-306): npts polyline points -> npts-1 edge bodies; polys [npolys][4][2]
 * hardcore boxes in creation order.  HOST pointers (tiny, uploaded once, shared by all worlds). */
int rem2d_world_set_terrain(rem2d_world *w, const float *xs, const float *ys, int32_t npts, const float *polys,
                            int32_t npolys, float friction);

/* Modular2D.reset (Modular2DEnv.py:565-598): destroy + re-create every world and build the
 * robots (create_robot :517-563) from the uploaded layout; wall of death back to 0. */
int rem2d_world_reset(rem2d_world *w, const rem2d_morph *morph_dev, void *stream);

/* Instead of rem2d_world_reset: the caller has filled EVERY field of the state arena itself, creature by creature, from the
 * arena of another world of the same lane count, terrain and flags between two steps (the fields of rem2d_world_field:
 * all of a creature's lanes, pair slots and per-creature words; padding creatures zero) -- e.g. the survivors of an
 * evaluate() episode moved into a smaller world, so that the wavefronts of the dead stop costing anything
 * (gym_rem2d_amd.env.BatchedModular2D.compact).  No counterpart in the reference (a b2World cannot be copied); the
 * creatures continue bit-identically because nothing outside the arena survives a step.  Follow with
 * rem2d_world_set_tiles for the new creature order. */
int rem2d_world_adopt(rem2d_world *w);

/* A population that lives in several worlds (one per lane count) reads reward / done in POPULATION order: with these
 * caller-owned device arrays set, every env-step also writes creature e's reward (float) and done (0 / 1 byte, the
 * storage of a torch.bool tensor) to reward_dev[index_dev[e]] / done_dev[index_dev[e]] -- what `step()` returns
 * (Modular2DEnv.py:642-653) without a gather per world on the host side.  index_dev: [n_envs] int32.  Three NULLs
 * switch it off.  The arrays must outlive the world or be reset before they are freed. */
int rem2d_world_set_outputs(rem2d_world *w, float *reward_dev, uint8_t *done_dev, const int32_t *index_dev);

/* Work partition of the velocity kernel (no counterpart in the reference: Box2D walks one island at a time,
 * b2World::Solve).  The 180 velocity iterations of world.Step (Modular2DEnv.py:634) run one TILE per wavefront: tile t
 * = creatures [tile_start[t], tile_start[t+1]).  Rules, checked here or flagged as REM2D_ERR_SOLVER_OVERFLOW by the
 * kernel: tile_start[0] = 0, strictly increasing, tile_start[n_tiles] in [n_envs, padded envs]; at most 256 lanes per
 * tile; at most 64 joints per schedule phase (joint round mod the tile's largest pipeline period) per tile.  Optional:
 * rem2d_world_create installs a valid default (128 / lanes creatures per tile); a host that knows the morphologies
 * packs tighter (gym_rem2d_amd.compiler.Morphology.tiles).  HOST pointer, copied. */
int rem2d_world_set_tiles(rem2d_world *w, const int32_t *tile_start, int32_t n_tiles);
/* Greedy tile plan from a morphology batch (HOST arrays in the layout of rem2d_morph: parent[n_envs*lanes],
 * jround[n_envs*lanes] = round | contact slot << 8 | period << 16).  Consecutive creatures are packed into a tile while
 * it stays within 256 lanes, 64 joints per schedule phase and max_creatures creatures (0: the default, 32 -- keeps the
 * tile's touching manifolds within the 128 that rem2d_vel4_kernel holds in registers for typical morphologies).
 * n_padded >= n_envs: creatures [n_envs, n_padded) are empty padding (rem2d_padded_envs).  tile_start_out needs room
 * for n_padded + 1 entries; *n_tiles_out receives the tile count. */
int rem2d_plan_tiles(const int32_t *parent, const int32_t *jround, int32_t n_envs, int32_t lanes, int32_t n_padded,
                     int32_t max_creatures, int32_t *tile_start_out, int32_t *n_tiles_out);
/* Tile shape of the velocity kernel for this world (again a launch shape without a counterpart in the reference):
 * 3 = 64 lanes per tile, one joint per kernel lane (4 wavefronts per SIMD; the default, fastest up to ~130 000 creatures
 * per GPU), 1 = 128 lanes with two joint register sets (4 wavefronts per SIMD, a third fewer wave-instructions: faster once
 * instruction issue alone limits the step, from ~130 000 creatures per GPU on), 2 = 192 lanes with three sets (3 wavefronts
 * per SIMD), 0 = 256 lanes with one register set per schedule phase (2 wavefronts per SIMD), 4 = 128 lanes with the static
 * phase -> register-set map (3 wavefronts per SIMD: what fixed-morphology populations want).  Replaces the tile table by the default plan of the new shape: call rem2d_world_set_tiles
 * afterwards with a plan made for the same shape (rem2d_plan_tiles_shape).  Results do not depend on it. */
int rem2d_world_set_tile_shape(rem2d_world *w, int32_t tile_shape);
/* rem2d_plan_tiles for a given tile shape (0 .. 4; -1 = the default, 3) */
int rem2d_plan_tiles_shape(const int32_t *parent, const int32_t *jround, int32_t n_envs, int32_t lanes, int32_t n_padded,
                           int32_t max_creatures, int32_t tile_shape, int32_t *tile_start_out, int32_t *n_tiles_out);

/* Creature order of a world (a launch shape again, no counterpart in the reference, no effect on results: creatures are
 * independent).  With an order installed, slot e of the velocity kernel's tile table and of the position kernel's 64-lane
 * blocks is creature order_dev[e] instead of creature e: the host may keep creatures of similar CURRENT cost together -- a tile
 * costs what its most expensive creature costs -- e.g. those that used every position iteration in the last steps
 * (REM2D_F_POSITERS), which stay the same for many steps (gym_rem2d_amd.env.BatchedModular2D.rebalance does that every few
 * dozen steps with a stable sort, so that the static schedule order survives within each class; REM2D_FLAG_RETILE is the
 * same indirection re-dealt by the kernel itself in every step, in arrival order).  order_dev: DEVICE array of
 * rem2d_padded_envs() int32, a permutation of 0 .. padded - 1 (the caller's responsibility), copied asynchronously on
 * `stream`; NULL: back to the identity.  Refused for a world created with REM2D_FLAG_RETILE.  Any order is valid with any
 * tile plan: the flexible tile shapes (1, 2, 3) take whatever creatures a tile's slots name, the static ones (0, 4), whose
 * plan by rem2d_plan_tiles depends on the morphologies a tile holds, keep the arena order in their velocity tiles and follow
 * the order in their position blocks only. */
int rem2d_world_set_order(rem2d_world *w, const int32_t *order_dev, void *stream);

/* Launch options of a world.  Like the tile shape they are launch shapes / scheduling hints without a counterpart in the
 * reference (b2World::Step has no such knobs) and NO result depends on them: every combination reproduces the same bits
 * (tests/test_parity_gpu.py::test_other_formulations_match_committed_digests runs them in one process).  The library reads
 * no environment variable; a host that wants experiment overrides sets them here (gym_rem2d_amd._lib maps REM2D_* variables
 * onto these calls for bench.py and tools/).  The options of the FIRST world of a step group steer that group's launches.
 *   REM2D_OPT_PIPELINE        3 (default): pre -> velocity tiles -> position kernel; 0: the fused body-per-lane step kernel
 *   REM2D_OPT_FUSE_VELPOST    2 (default): the step train -- all steps of a call in ONE launch, a workgroup per (step, 64-lane block)
 *                             runs pre, the velocity tile, the position block and the TOI solve of its own bodies and hands the
 *                             block to its next step through a flag (rem2d_step_train_kernel; 64-lane tiles, not REM2D_FLAG_RETILE);
 *                             1: velocity tiles + position iterations of a 64-lane block in one launch per step; 0: two
 *   REM2D_OPT_PRIO            s_setprio mask, default 5: bit 1 slow velocity tiles, bit 4 the TOI solve's wavefronts; 0 off
 *   REM2D_OPT_PRIO_T1 / _T2   slot-cost thresholds (7 per tick + 10 per contact sub-slot) for priority 1 / 3, default 60 / 75
 *   REM2D_OPT_HEAVY_PER_WAVE  bodies of the TOI work list per wavefront, 1..64, default 1
 *   REM2D_OPT_DEBUG           diagnostic builds only (-DREM2D_V4_PROBES), default 0
 *   REM2D_OPT_REBALANCE       N > 0: every N env-steps the library re-makes the world's creature order on the device (what
 *                             rem2d_world_set_order installs from the host: the creatures that used every position iteration in
 *                             the last step first, a stable partition); 0 (default) off.  Refused with REM2D_FLAG_RETILE.  The step
 *                             train re-orders in front of a launch once N or more steps have run since the last time (a call is
 *                             cut only where it is itself longer than N steps)
 *   REM2D_OPT_TRAIN_FAULT     test hook of the step train's hand-over check, default 0 = off.  s | m << 16: the workgroups of step s
 *                             (1-based inside a launch) of the blocks with block % m == 0 (m <= 1: all) are TOLD that their hand-over
 *                             failed; with bit 30 also set their flag is never published instead, so that they wait into the 2 s
 *                             limit and every later wait of the launch ends at once.  The arithmetic is untouched; the creatures of
 *                             those blocks carry REM2D_ERR_HANDOVER and rem2d_world_handover_failures moves
 *                             (tests/test_handover_gpu.py: the host's re-evaluation on per-step launches gives the oracle's fitness)
 * Returns REM2D_E_INVALID for an unknown key or a value outside the option's range. */
enum {
    REM2D_OPT_PIPELINE = 0, REM2D_OPT_FUSE_VELPOST, REM2D_OPT_PRIO, REM2D_OPT_PRIO_T1, REM2D_OPT_PRIO_T2,
    REM2D_OPT_HEAVY_PER_WAVE, REM2D_OPT_DEBUG, REM2D_OPT_REBALANCE, REM2D_OPT_TRAIN_FAULT, REM2D_OPT_COUNT
};
int rem2d_world_set_option(rem2d_world *w, int32_t key, int32_t value);
int rem2d_world_get_option(const rem2d_world *w, int32_t key, int32_t *value);

/* n_steps x Modular2D.step (Modular2DEnv.py:607-653): wod.update, controller sweep
 * (m_controller.py:17-21), PID -> joint.motorSpeed (:600-605,:631-632),
 * world.Step(1/50, 180, 60) (:634), reward / done (:642-649) and evaluate()'s fitness rule
 * (REM2D_main.py:362-377).  Asynchronous on `stream` (hipStream_t). */
int rem2d_world_step(rem2d_world *w, int32_t n_steps, void *stream);
/* same with explicit b2World::Step arguments (dt, velocityIterations, positionIterations); iteration counts outside
 * 0..8192 are refused with REM2D_E_INVALID (the solver loops count ticks = iterations x schedule period in 16 bits) */
int rem2d_world_step_ex(rem2d_world *w, int32_t n_steps, float dt, int32_t vel_iters, int32_t pos_iters,
                        void *stream);

/* n_steps x Modular2D.step for SEVERAL worlds of one device in one launch per kernel: the lane buckets
 * (2/4/8/../64 lanes per creature) of one population, which the reference steps as independent envs
 * (REM2D_main.py:256-267 pool.map over individuals).  Same result as calling rem2d_world_step on each
 * world; the merged grid lets the GPU pack the small buckets next to the big one instead of running
 * them one after the other.  All worlds must share the device and the REM2D_FLAG_CONTINUOUS setting;
 * at most REM2D_MAX_WORLDS_PER_STEP worlds.  Kernel timing is booked on worlds[0]. */
#define REM2D_MAX_WORLDS_PER_STEP 8
int rem2d_worlds_step(rem2d_world *const *worlds, int32_t n_worlds, int32_t n_steps, void *stream);
int rem2d_worlds_step_ex(rem2d_world *const *worlds, int32_t n_worlds, int32_t n_steps, float dt, int32_t vel_iters,
                         int32_t pos_iters, void *stream);

/* n_steps x Modular2D.step for a population that has been cut into STEP GROUPS (independent parts -- the reference steps
 * every individual as an independent env, REM2D_main.py:256-267 pool.map, :362-368 the per-individual step loop -- each
 * a set of lane-bucket worlds as in rem2d_worlds_step), every group on a stream of its own so that one group's chain of
 * kernels runs under the others'.  One call queues the whole job: the group streams first wait for everything queued on
 * `stream` so far (fork), then step l of EVERY group is queued before step l + 1 of any (round-robin: no group's stream
 * runs dry while the host queues another group's train), and finally `stream` waits for every group (join).  A group
 * whose `stream` is NULL runs on `stream` itself.  Same result as rem2d_worlds_step on every group.
 * flags: REM2D_STEP_GRAPH -- capture the call's launches and fork / join edges into a hipGraph the first time and replay it
 * with one hipGraphLaunch afterwards (re-captured when a world's tiles / outputs / terrain change; ignored while kernel
 * timing is on).  With REM2D_OPT_REBALANCE the position of the call in the N-step cadence is part of the replay's identity, so
 * the re-ordering launches fall on the same env-steps as without the flag (a call length that does not divide N needs
 * N / gcd(N, n_steps) captures). */
#define REM2D_MAX_STEP_GROUPS 16
#define REM2D_STEP_GRAPH 1u
typedef struct {
    rem2d_world *const *worlds; /* the group's lane-bucket worlds, <= REM2D_MAX_WORLDS_PER_STEP */
    int32_t n_worlds;
    void *stream;               /* hipStream_t of this group; NULL: the call's `stream` */
} rem2d_step_group;
int rem2d_groups_step(const rem2d_step_group *groups, int32_t n_groups, int32_t n_steps, void *stream, uint32_t flags);
int rem2d_groups_step_ex(const rem2d_step_group *groups, int32_t n_groups, int32_t n_steps, float dt, int32_t vel_iters,
                         int32_t pos_iters, void *stream, uint32_t flags);

/* Host-side genotype -> phenotype for L-system genomes (Encodings/LSystem.py:144-199 create, then
 * Modular2DEnv.py:517-563 create_robot with simple_module.py:147-199,231-313, circular_module.py:138-221 and
 * module_utility.py:7-33), native and multi-threaded: n genomes in, the [creature][lane] arrays of
 * rem2d_world_reset out.  HOST pointers throughout.  SoA over genomes; a genome has n_types module
 * prototypes (the reference: 4 boxes + 4 circles, REM2D_main.py:69-77) and one rewriting rule per type. */
typedef struct rem2d_lsystem_genomes {
    int32_t n, n_types;
    const int32_t *mod_shape;                 /* [n][n_types] 1 box (Standard2D), 2 circle (Circular2D) */
    const double *mod_width, *mod_height;     /* [n][n_types] box */
    const double *mod_radius;                 /* [n][n_types] circle */
    const double *mod_angle, *mod_torque;     /* [n][n_types] */
    const double *ctl_amp, *ctl_phase, *ctl_freq, *ctl_offset; /* [n][n_types] prototype controller */
    const int32_t *rule_n;                    /* [n][n_types] products of the rule for that type (0..3) */
    const int32_t *rule_site;                 /* [n][n_types][3] 0 left, 1 right, 2 top */
    const int32_t *rule_ref;                  /* [n][n_types][3] module type of the product */
} rem2d_lsystem_genomes;
/* out: host arrays of n*lanes elements each (same fields as rem2d_morph, written, not read);
 * n_bodies[n]: bodies of every creature.  Fails (and marks the creature with n_bodies = -1) if a creature
 * needs more than `lanes` lanes.  max_modules <= 63.  n_threads <= 0: all hardware threads.
 * out == NULL (all three compilers): expression and create_robot's rules only -- n_bodies is filled, nothing else is written
 * (the static cost key by which a sharded job deals its individuals to the ranks, gym_rem2d_amd.evaluate.shard_balanced). */
int rem2d_compile_lsystem(const rem2d_lsystem_genomes *genomes, int32_t tree_depth, int32_t max_modules,
                          double terrain_height, int32_t lanes, const rem2d_morph *out, int32_t *n_bodies,
                          int32_t n_threads);

/* The same for phenotype trees of any encoding, node by node: Encodings/Direct_Encoding.py:18-27 (the genome IS the
 * tree, one module object per node), Network_Encoding.py:97-129,171-206 (one module + controller per network query),
 * i.e. whatever genome.create() returned, flattened in Tree.getNodes() order.  HOST pointers, [n][max_nodes]. */
typedef struct {
    int32_t n, max_nodes;                     /* creatures, nodes per creature (<= 64) */
    const int32_t *node_count;                /* [n] nodes of each tree */
    const int32_t *index, *parent;            /* node.index, node.parent (-1: root) */
    const int32_t *site;                      /* parent_connection_coordinates: -1 none, 0 left, 1 right, 2 top */
    const int32_t *shape;                     /* 1 box (Standard2D), 2 circle (Circular2D) */
    const double *width, *height, *radius;    /* module_.width / height / radius */
    const double *angle, *torque;             /* module_.angle / torque */
    const double *ctl_amp, *ctl_phase, *ctl_freq, *ctl_offset; /* node.controller */
} rem2d_tree_batch;
int rem2d_compile_trees(const rem2d_tree_batch *trees, double terrain_height, int32_t lanes, const rem2d_morph *out,
                        int32_t *n_bodies, int32_t n_threads);

/* DirectEncoding.mutate (Encodings/Direct_Encoding.py:82-139: removal of subtrees, growth on free connection sites, then
 * module.mutate / controller.mutate of every visited node -- simple_module.py:70-85, circular_module.py:67-80,
 * m_controller.py:50-58) for a whole population of direct-encoding genomes held as node tables, IN PLACE: what
 * REM2D_main.py:280-298 does to every offspring of every generation, one python object at a time.  HOST pointers, [n][max_nodes]
 * in Tree.getNodes() order (a node's subtree is the run of columns behind it; parent = the parent's column, -1 for the root);
 * the same tables rem2d_compile_trees reads (its `index` is the column).  The reference's algorithm with its list-iterator
 * semantics and recounts; its own random stream: one generator per individual seeded from (seed, individual), so the result
 * does not depend on n_threads. */
typedef struct {
    int32_t n, max_nodes;                      /* individuals, columns of the tables (<= 64) */
    int32_t *node_count;                       /* [n] */
    int32_t *parent, *site, *shape;            /* [n][max_nodes]: parent column, 0 left / 1 right / 2 top (-1 root), 1 box / 2 circle / 0 padding */
    double *width, *height, *radius, *angle, *torque;
    double *ctl_amp, *ctl_phase, *ctl_freq, *ctl_offset;
    int32_t max_modules, max_depth;            /* DirectEncoding.maxModules / maxDepth (20 / 8) */
    int32_t n_box, n_circle;                   /* the module list's prototypes a new node copies (REM2D_main.py:69-77: 4 + 4) */
    double proto_box_width, proto_box_height, proto_circle_radius, proto_angle, proto_torque; /* their (never mutated) defaults */
    /* class constants, handed over like rem2d_network_genomes' */
    double box_min_width, box_max_width, box_min_height, box_max_height, box_min_angle, box_max_angle;
    double circle_min_radius, circle_max_radius, circle_min_angle, circle_max_angle;
    double ctl_max_amp, ctl_max_phase, ctl_max_offset, ctl_max_freq;
} rem2d_tree_population;
int rem2d_mutate_trees(const rem2d_tree_population *pop, double morph_rate, double rate, double sigma, uint64_t seed,
                       int32_t n_threads);

/* The network encoding end to end (Encodings/Network_Encoding.py:42-222: update / iterate / create / recursiveNodeGen --
 * one network query per free connection site decides whether a child exists, its module type, shape (setMorph,
 * simple_module.py:86-93, circular_module.py) and controller (setControl, m_controller.py)), then create_robot as above.
 * The network is this build's synthetic feed-forward CPPN (neat-python is absent): 3 inputs, n_hidden nodes with one of
 * four activations (0 tanh, 1 sin, 2 gauss-like exp(-x^2)*2-1, 3 clamp), 10 tanh outputs.  HOST pointers. */
typedef struct {
    int32_t n, n_types, n_hidden, max_modules; /* genomes, module prototypes per genome, hidden nodes, NNEncoding.maxModules */
    const double *w1;                          /* [n][n_hidden][4]  (3 inputs + bias) */
    const int32_t *a1;                         /* [n][n_hidden]     activation id */
    const double *w2;                          /* [n][10][n_hidden + 1] */
    const int32_t *mod_shape;                  /* [n][n_types] prototypes (genome.moduleList), as in rem2d_lsystem_genomes */
    const double *mod_width, *mod_height, *mod_radius, *mod_angle, *mod_torque;
    const double *ctl_amp, *ctl_phase, *ctl_freq, *ctl_offset;
    /* class constants of the module / controller classes (simple_module.py:33-39, circular_module.py, m_controller.py:9-12) */
    double box_min_width, box_max_width, box_min_height, box_max_height, box_min_angle, box_max_angle;
    double circle_min_radius, circle_max_radius, circle_min_angle, circle_max_angle;
    double ctl_max_amp, ctl_max_phase, ctl_max_offset, ctl_max_freq;
} rem2d_network_genomes;
int rem2d_compile_network(const rem2d_network_genomes *genomes, int32_t tree_depth, double terrain_height, int32_t lanes,
                          const rem2d_morph *out, int32_t *n_bodies, int32_t n_threads);

/* Population diversity (DataAnalysis/AdvancedDataAnalysis.py:291-313 compare_distance, :367-381
 * tree_edit_distance): out[c] = sum over t != c of the number of nodes of c whose (x, y) position does not
 * occur in t plus the number of nodes of t that no node of c sits on; positions are binary64 and compared
 * with ==.  pos_dev [n_trees][max_nodes][2] f64 (get_tree_pos layout, computed by the caller),
 * count_dev [n_trees] i32 (nodes per tree, <= max_nodes <= 64), out_dev [n_trees] i64.  Device pointers;
 * asynchronous on `stream`.  Independent of any rem2d_world. */
int rem2d_tree_diversity(const double *pos_dev, const int32_t *count_dev, int32_t n_trees, int32_t max_nodes,
                         int64_t *out_dev, int32_t device, void *stream);

/* Self-test of the scalar helpers every solver is built from, element by element on the device: out_dev [5][n] =
 * b2Min(a, b), b2Max(a, b), b2Clamp(a, lo = b, hi = c) (b2Math.h; one v_med3_f32 each here), and b2Rot(a).s / .c (b2Rot::Set;
 * "rem2d trig").  So that a test can feed the special values no trajectory reaches -- NaN, infinities,
 * denormals, zeros of either sign -- through the device code and through the oracle's form
 * (tests/test_parity_gpu.py::test_scalar_helpers_special_values lists where the two may differ: the sign of a zero
 * result and NaN operands, nothing else).  Device pointers; asynchronous on `stream`.  Independent of any rem2d_world. */
int rem2d_selftest_scalar(const float *a_dev, const float *b_dev, const float *c_dev, int32_t n, float *out_dev,
                          int32_t device, void *stream);

/* In-place view of a state field: byte offset into `state`, element count, REM2D_DT_*.
 * Replaces the per-object reads body.position / body.angle / joint.angle and the per-step
 * return values reward / done. */
int rem2d_world_field(const rem2d_world *w, int32_t field, size_t *offset_bytes, size_t *count, int32_t *dtype);

/* Average device time of the step kernel over the launches since the last call, measured
 * with HIP events on the launch stream (bench.py's roofline leg).  Synchronises. */
int rem2d_world_kernel_time_ms(rem2d_world *w, double *total_ms, int64_t *launches);
/* The same for the whole kernel sequence of one env-step (pre, velocity, post, TOI scan, TOI solve) of the tile pipeline:
 * device time between the first kernel's start and the last kernel's end, summed over the steps since the last call.
 * `steps` counts ENV-STEPS: a step train's bracket spans a whole launch and adds the steps of that launch. */
int rem2d_world_step_time_ms(rem2d_world *w, double *total_ms, int64_t *steps);
/* How many hand-overs of this world's step trains have failed so far (REM2D_ERR_HANDOVER: the count of (step, block) workgroups,
 * booked on the FIRST world of a launch / step group): read from a word in pinned host memory that the kernels add to, so the call
 * neither synchronises nor touches the device; it reports what the launches that have finished so far saw.  clear != 0 resets it.
 * (b2World::Step cannot fail this way: Modular2DEnv.py:634 is one synchronous call.  A host checks this after a step call and
 * before trusting reward / done -- gym_rem2d_amd.env.BatchedModular2D.step raises HandoverError.) */
int rem2d_world_handover_failures(const rem2d_world *w, int64_t *count, int32_t clear);
/* on = 0: off; on = 1: on, room for 4096 timed launches between two read-backs; on > 1: room for `on` launches.  The
 * event pairs are created here, not inside the step calls. */
int rem2d_world_enable_timing(rem2d_world *w, int32_t on);

#ifdef __cplusplus
}
#endif
#endif
