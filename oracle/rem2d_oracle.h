/*
 * rem2d_oracle.h -- CPU restatement of the gym_rem2D world.Step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the third-party
 * dependency Box2D==2.3.10 (pybox2d, /root/reference/requirements.txt:1),
 * which is absent from /root/reference and from this image, and the reference
 * ships no tests or golden vectors for it.  This file restates the published
 * Box2D 2.3.x algorithm for exactly the feature subset
 * that Modular2DEnv.py / simple_module.py / circular_module.py /
 * module_utility.py drive; it is checked against analytic known-answer tests
 * (tests/test_oracle_kat.py) and, for the Python side of the path, against
 * fixtures captured from the importable reference (tests/golden/).
 */
#ifndef REM2D_ORACLE_H
#define REM2D_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flags for rem2d_oracle_world_create */
#define O_FLAG_CONTINUOUS 1u         /* b2World continuousPhysics (SolveTOI), SURVEY A.8 */
#define O_FLAG_SLEEP_RESET_ALWAYS 2u /* b2Body::SetAwake(true) always zeroes sleepTime (pre-2.3.1 variant) */
#define O_FLAG_NO_SLEEP 4u           /* b2World(doSleep=False) */
#define O_FLAG_TOI_TRANSPARENT_STATICS 8u /* SolveTOI without the static bodies' sweep.alpha0 / island-flag bookkeeping
                                            * (round-1 form; bit-identical results, tests/test_oracle_kat.py) */

#define O_MAX_BODIES 64
#define O_MAX_BODY_CONTACTS 32 /* >= the wide HIP build's pair slots (include/rem2d.h REM2D_WIDE) */

typedef struct o_terrain o_terrain;
typedef struct o_world o_world;

/* Morphology + controller batch in [env][lane] layout (lane fastest).  Slot 0
 * of every env is the root; slot s>=1 carries its body and the joint that ties
 * it to `parent[s]` (joint index s-1 == creation order, Modular2DEnv.py:517-563). */
typedef struct {
    int32_t n_envs, lanes;
    const int32_t *shape;  /* 0 none, 1 box (simple_module.py:286), 2 circle (circular_module.py:191) */
    const float *hx, *hy;  /* box half extents; circle: hx = radius */
    const float *x, *y, *angle;
    const int32_t *parent; /* -1: none */
    const float *ax, *ay, *bx, *by; /* localAnchorA/B (module_utility.py:10-17) */
    const float *torque, *lower, *upper;
    const double *amp, *phase, *freq, *offset, *istate; /* Controller/m_controller.py:5-21 */
} o_morph;

/* Terrain: npts polyline heights -> npts-1 edge bodies (Modular2DEnv.py:294-306);
 * polys: [npoly][4][2] static boxes of the hardcore track (Modular2DEnv.py:217-275),
 * listed in creation order (they precede the edges). */
o_terrain *rem2d_oracle_terrain_create(const float *xs, const float *ys, int npts,
                                       const float *polys, int npoly, float friction);
void rem2d_oracle_terrain_destroy(o_terrain *);

o_world *rem2d_oracle_world_create(const o_terrain *, unsigned flags);
void rem2d_oracle_world_destroy(o_world *);
int rem2d_oracle_add_box(o_world *, float hx, float hy, float x, float y, float angle);
int rem2d_oracle_add_circle(o_world *, float r, float x, float y, float angle);
int rem2d_oracle_add_joint(o_world *, int bodyA, int bodyB, float ax, float ay, float bx, float by,
                           float maxTorque, float lower, float upper);
void rem2d_oracle_set_controller(o_world *, int joint, double amp, double phase, double freq,
                                 double offset, double istate);
void rem2d_oracle_set_motor_speed(o_world *, int joint, float speed);
void rem2d_oracle_set_velocity(o_world *, int body, float vx, float vy, float w);
void rem2d_oracle_set_gravity(o_world *, float gx, float gy);
/* state re-synchronisation for tests/test_box2d_pin.py (what pybox2d exposes of a b2World, see rem2d_oracle.c) */
void rem2d_oracle_set_body_state(o_world *, int body, float x, float y, float angle, float vx, float vy, float w, int awake);
void rem2d_oracle_set_joint_impulses(o_world *, int joint, float ix, float iy, float iz, float motorImpulse);
void rem2d_oracle_set_contact_impulses(o_world *, int body, int k, float n0, float n1, float t0, float t1);
double rem2d_oracle_get_wod(const o_world *);
void rem2d_oracle_get_controller_state(const o_world *, double *out);
/* b2World::Step */
void rem2d_oracle_world_step(o_world *, float dt, int velIters, int posIters);
/* Modular2D.step (Modular2DEnv.py:607-653): wod, controllers, PID, Step(1/50,180,60), reward/done */
void rem2d_oracle_env_step(o_world *, double *reward, int *done);
void rem2d_oracle_env_step_ex(o_world *, float dt, int velIters, int posIters, double *reward, int *done);

int rem2d_oracle_num_bodies(const o_world *);
int rem2d_oracle_num_joints(const o_world *);
/* out[n][8] = x y angle vx vy w sleepTime awake */
void rem2d_oracle_get_bodies(const o_world *, float *out);
/* out[n][4] = invMass invI mass I */
void rem2d_oracle_get_mass(const o_world *, float *out);
/* out[n][6] = impulse.x impulse.y impulse.z motorImpulse motorSpeed limitState */
void rem2d_oracle_get_joints(const o_world *, float *out);
/* island joint order of the last Solve (b2World::Solve DFS), returns count */
int rem2d_oracle_get_island_joint_order(const o_world *, int32_t *out);
/* per-body contact list in list order (head first).  out[k][8] =
 *   static index, pointCount, manifold type, touching, key0, key1, (bits of) nImp0.. -> see below
 * fout[k][4] = normalImpulse0, normalImpulse1, tangentImpulse0, tangentImpulse1.  Returns count. */
int rem2d_oracle_get_contacts(const o_world *, int body, int32_t *out, float *fout);
/* manifold geometry of contact k of body: out[8] = localNormal.xy localPoint.xy p0.xy p1.xy */
void rem2d_oracle_get_manifold(const o_world *, int body, int k, float *out);
/* body fat AABB: out[4] */
void rem2d_oracle_get_fat_aabb(const o_world *, int body, float *out);
int rem2d_oracle_position_iterations(const o_world *); /* of last Solve */
int rem2d_oracle_toi_events(const o_world *);          /* cumulative TOI sub-steps */
/* times SolveTOI had to advance a dynamic body's sweep because its static partner's alpha0 was ahead */
int rem2d_oracle_toi_dynamic_advances(const o_world *);

/* standalone pieces for known-answer tests */
void rem2d_oracle_sincosf(float a, float *s, float *c);
int rem2d_oracle_kat_scalar(const float *a, const float *b, const float *c, int32_t n, float *out);
double rem2d_oracle_sin(double x);
void rem2d_oracle_box_mass(float hx, float hy, float *mass, float *I);
void rem2d_oracle_circle_mass(float r, float *mass, float *I);

/* Batch driver used by parity tests and bench.py's cpu_baseline leg: builds one
 * world per env from the morphology, runs n_steps of Modular2D.step on n_threads
 * host threads and writes final state.
 *   bodies_out [n_envs][lanes][8]   (as rem2d_oracle_get_bodies)
 *   reward_out [n_envs] (double), done_out [n_envs] (int32; sticky "was ever done"),
 *   fitness_out [n_envs] (double; evaluate() rule REM2D_main.py:350-378, may be NULL)
 *   trace_out  [n_steps][n_envs][lanes][3] x,y,angle per step (may be NULL)
 * Returns 0 on success. */
int rem2d_oracle_batch_run(const o_terrain *, const o_morph *, int n_steps, int n_threads,
                           unsigned flags, float *bodies_out, double *reward_out,
                           int32_t *done_out, double *fitness_out, float *trace_out);
int rem2d_oracle_batch_run_caps(const o_terrain *, const o_morph *, int n_steps, int n_threads, unsigned flags,
                                float *bodies_out, double *reward_out, int32_t *done_out, double *fitness_out,
                                int32_t *caps_out /* [N][3]: max pairs / max touching manifolds on one body, refused pairs */);

/* bench.py's cpu_baseline leg: `settle` untimed steps of every creature (worlds kept), then ONE continuous wall-clock window
 * around `window` further steps of all of them (OpenMP over creatures).  *seconds_out = the window's wall time. */
int rem2d_oracle_batch_window(const o_terrain *, const o_morph *, int settle, int window, int n_threads, unsigned flags,
                              double *seconds_out, double *reward_out);

/* TOI sub-steps and forced dynamic-sweep advances (see rem2d_oracle_toi_dynamic_advances) summed over all worlds
 * that rem2d_oracle_batch_run has stepped since the last reset. */
void rem2d_oracle_batch_toi_stats(long long *events, long long *dynamic_advances, int reset);

/* Build a single world from env `e` of a morphology batch. */
o_world *rem2d_oracle_world_from_morph(const o_terrain *, const o_morph *, int e, unsigned flags);

#ifdef __cplusplus
}
#endif
#endif
