/*
 * rem2d_cpu.c -- the C ABI of include/rem2d.h once more, on HOST pointers, backed by the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY (SURVEY.md 8b proposed a `rem2d_cpu_*` twin of the boundary; like everything under
 * oracle/ it may be loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
 * product package).  Every entry point has the signature of its `rem2d_*` counterpart with the prefix `rem2d_cpu_`;
 * `state` is caller-owned HOST memory with the SAME field layout as the device arena (rem2d_cpu_world_field returns
 * the same offsets as rem2d_world_field), so a test can drive both libraries with one call sequence and compare the
 * two arenas field by field (tests/test_cpu_twin.py).  One creature = one oracle world (rem2d_oracle.c, included
 * below), stepped by rem2d_oracle_env_step_ex; the arena is refreshed at the end of every step call.
 *
 * Fields the twin maintains: every per-lane pose / velocity / sleep / joint-impulse / limit-state / controller field,
 * the static per-lane fields written by reset, the pair lists (edge, point count | type << 8, feature keys, warm-start
 * impulses, in list order) and every per-creature field except REM2D_F_NEWFIX.  Scheduling entry points of the GPU
 * library (set_tiles, timing) are accepted and ignored; the host-only entry points of librem2d.so (rem2d_plan_tiles,
 * rem2d_compile_*) need no twin.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "../include/rem2d.h"
#include "rem2d_oracle.c"

/* ---- arena layout: restated from gym_rem2d_amd/csrc/rem2d_state.h (make_layout / field_place) ---- */
enum { CG_LANE4 = 0, CG_LANE8 = 1, CG_SLOT4 = 2, CG_ENV4 = 3, CG_ENV8 = 4 };
#define C_WAVE 64
#define C_L4_COUNT (REM2D_F_CCOUNT + 1)
#define C_L8_COUNT (REM2D_F_CISTATE - REM2D_F_CAMP + 1)
#define C_S4_COUNT (REM2D_F_CT1 - REM2D_F_CEDGE + 1)
#define C_E8_COUNT 2
#define C_E4_COUNT (REM2D_F_TOIEVENTS - REM2D_F_REWARD + 1)

typedef struct {
    int Np, Lp, K;
    size_t groupOff[5];
    size_t total;
} c_layout;

static int cfg_ok(const rem2d_world_cfg *cfg) {
    if (!cfg || cfg->n_envs <= 0) return 0;
    int k = cfg->lanes;
    return k >= 2 && k <= REM2D_MAX_LANES && (k & (k - 1)) == 0;
}
static c_layout c_make_layout(const rem2d_world_cfg *cfg) {
    c_layout L;
    int perWave = C_WAVE / cfg->lanes;
    L.K = cfg->lanes;
    L.Np = (cfg->n_envs + perWave - 1) / perWave * perWave;
    L.Lp = L.Np * cfg->lanes;
    size_t sizes[5];
    sizes[CG_LANE4] = (size_t)C_L4_COUNT * L.Lp * 4;
    sizes[CG_LANE8] = (size_t)C_L8_COUNT * L.Lp * 8;
    sizes[CG_SLOT4] = (size_t)C_S4_COUNT * REM2D_CONTACT_SLOTS * L.Lp * 4;
    sizes[CG_ENV4] = (size_t)C_E4_COUNT * L.Np * 4;
    sizes[CG_ENV8] = (size_t)C_E8_COUNT * L.Np * 8;
    const int order[5] = {CG_LANE8, CG_ENV8, CG_LANE4, CG_SLOT4, CG_ENV4}; /* 8-byte groups first */
    size_t o = 0;
    for (int k = 0; k < 5; ++k) {
        L.groupOff[order[k]] = o;
        o += sizes[order[k]];
        o = (o + 255) & ~(size_t)255;
    }
    L.total = o;
    return L;
}
static void c_field_place(const c_layout *L, int f, size_t *off, size_t *count, int *dtype) {
    int group, index, dt;
    if (f <= REM2D_F_CCOUNT) { group = CG_LANE4; index = f; dt = f >= REM2D_F_SHAPE ? REM2D_DT_I32 : REM2D_DT_F32; }
    else if (f <= REM2D_F_CISTATE) { group = CG_LANE8; index = f - REM2D_F_CAMP; dt = REM2D_DT_F64; }
    else if (f <= REM2D_F_CT1) { group = CG_SLOT4; index = f - REM2D_F_CEDGE; dt = f <= REM2D_F_CKEY1 ? REM2D_DT_I32 : REM2D_DT_F32; }
    else if (f == REM2D_F_WOD || f == REM2D_F_FITNESS) { group = CG_ENV8; index = f - REM2D_F_WOD; dt = REM2D_DT_F64; }
    else { group = CG_ENV4; index = f - REM2D_F_REWARD; dt = (f == REM2D_F_REWARD || f == REM2D_F_INVDT0) ? REM2D_DT_F32 : REM2D_DT_I32; }
    size_t n = group == CG_LANE4 || group == CG_LANE8 ? (size_t)L->Lp : group == CG_SLOT4 ? (size_t)L->Lp * REM2D_CONTACT_SLOTS : (size_t)L->Np;
    size_t esz = dt == REM2D_DT_F64 ? 8 : 4;
    if (off) *off = L->groupOff[group] + (size_t)index * n * esz;
    if (count) *count = n;
    if (dtype) *dtype = dt;
}

/* ---- the handle ---- */
typedef struct rem2d_cpu_world {
    rem2d_world_cfg cfg;
    c_layout L;
    char *arena;
    o_terrain *terrain;
    o_world **worlds; /* [n_envs] */
    int8_t *slotBody; /* [n_envs * lanes]: oracle body index of the lane, -1 = empty */
    int haveReset;
    float *outReward;
    uint8_t *outDone;
    const int32_t *outIndex;
    int32_t opt[REM2D_OPT_COUNT]; /* launch options: stored, never read (no launches on the host) */
} rem2d_cpu_world;

static _Thread_local char c_err[256];
static int c_fail(int code, const char *msg) {
    snprintf(c_err, sizeof c_err, "%s", msg);
    return code;
}
const char *rem2d_cpu_last_error(void) { return c_err; }
int rem2d_cpu_abi_version(void) { return REM2D_ABI_VERSION; }
const char *rem2d_cpu_build_id(void) { return "cpu-twin"; } /* (rem2d_build_id: the checker has no build to tell apart) */

size_t rem2d_cpu_state_bytes(const rem2d_world_cfg *cfg) { return cfg_ok(cfg) ? c_make_layout(cfg).total : 0; }
int32_t rem2d_cpu_padded_envs(const rem2d_world_cfg *cfg) { return cfg_ok(cfg) ? c_make_layout(cfg).Np : 0; }

#define FPTR(T, f) ((T *)(w->arena + foff[f]))

int rem2d_cpu_world_create(const rem2d_world_cfg *cfg, void *state_host, size_t state_bytes, rem2d_cpu_world **out) {
    if (!out) return c_fail(REM2D_E_INVALID, "out is NULL");
    *out = NULL;
    if (!cfg_ok(cfg)) return c_fail(REM2D_E_INVALID, "n_envs must be > 0 and lanes a power of two in 2..64");
    if (cfg->lanes > O_MAX_BODIES) return c_fail(REM2D_E_INVALID, "lanes exceed the oracle's body limit");
    c_layout L = c_make_layout(cfg);
    if (!state_host || state_bytes < L.total) return c_fail(REM2D_E_INVALID, "state buffer missing or smaller than rem2d_cpu_state_bytes");
    rem2d_cpu_world *w = (rem2d_cpu_world *)calloc(1, sizeof *w);
    if (!w) return c_fail(REM2D_E_NOMEM, "out of memory");
    w->cfg = *cfg;
    w->L = L;
    w->arena = (char *)state_host;
    w->worlds = (o_world **)calloc((size_t)cfg->n_envs, sizeof(o_world *));
    w->slotBody = (int8_t *)malloc((size_t)cfg->n_envs * cfg->lanes);
    if (!w->worlds || !w->slotBody) { free(w->worlds); free(w->slotBody); free(w); return c_fail(REM2D_E_NOMEM, "out of memory"); }
    memset(w->arena, 0, L.total);
    { static const int32_t def[REM2D_OPT_COUNT] = {3, 1, 5, 60, 75, 1, 0, 0, 0}; memcpy(w->opt, def, sizeof def); }
    *out = w;
    return REM2D_OK;
}
static void c_free_worlds(rem2d_cpu_world *w) {
    for (int e = 0; e < w->cfg.n_envs; ++e)
        if (w->worlds[e]) { rem2d_oracle_world_destroy(w->worlds[e]); w->worlds[e] = NULL; }
}
int rem2d_cpu_world_destroy(rem2d_cpu_world *w) {
    if (!w) return REM2D_OK;
    c_free_worlds(w);
    if (w->terrain) rem2d_oracle_terrain_destroy(w->terrain);
    free(w->worlds);
    free(w->slotBody);
    free(w);
    return REM2D_OK;
}
int rem2d_cpu_world_set_terrain(rem2d_cpu_world *w, const float *xs, const float *ys, int32_t npts, const float *polys,
                                int32_t npolys, float friction) {
    if (!w) return c_fail(REM2D_E_INVALID, "world is NULL");
    if (npts < 2 || !xs || !ys || npolys < 0 || (npolys > 0 && !polys)) return c_fail(REM2D_E_INVALID, "terrain needs >= 2 points");
    c_free_worlds(w); /* worlds hold a pointer to their terrain */
    w->haveReset = 0;
    if (w->terrain) rem2d_oracle_terrain_destroy(w->terrain);
    w->terrain = rem2d_oracle_terrain_create(xs, ys, npts, polys, npolys, friction);
    return w->terrain ? REM2D_OK : c_fail(REM2D_E_NOMEM, "terrain allocation failed");
}
int rem2d_cpu_world_set_outputs(rem2d_cpu_world *w, float *reward, uint8_t *done, const int32_t *index) {
    if (!w) return c_fail(REM2D_E_INVALID, "world is NULL");
    if (index && (!reward || !done)) return c_fail(REM2D_E_INVALID, "reward and done are required with an index");
    w->outReward = index ? reward : NULL;
    w->outDone = index ? done : NULL;
    w->outIndex = index;
    return REM2D_OK;
}
int rem2d_cpu_world_set_tiles(rem2d_cpu_world *w, const int32_t *tile_start, int32_t n_tiles) {
    (void)tile_start; (void)n_tiles;
    return w ? REM2D_OK : c_fail(REM2D_E_INVALID, "world is NULL"); /* a launch shape: nothing to do on the CPU */
}
int rem2d_cpu_world_adopt(rem2d_cpu_world *w) {
    (void)w; /* the twin's state lives in its oracle worlds, not in the arena: there is nothing it could adopt */
    return c_fail(REM2D_E_INVALID, "rem2d_cpu_world_adopt: not supported by the host-pointer twin");
}
int rem2d_cpu_world_set_tile_shape(rem2d_cpu_world *w, int32_t tile_shape) {
    if (!w) return c_fail(REM2D_E_INVALID, "world is NULL");
    if (tile_shape < 0 || tile_shape > 4) return c_fail(REM2D_E_INVALID, "tile shape must be 0 .. 4");
    return REM2D_OK; /* a launch shape: nothing to do on the CPU */
}
/* a creature order is a launch shape: nothing to do on the CPU (the twin steps creature by creature) */
int rem2d_cpu_world_set_order(rem2d_cpu_world *w, const int32_t *order, void *stream) {
    (void)order; (void)stream;
    if (!w) return c_fail(REM2D_E_INVALID, "world is NULL");
    if (w->cfg.flags & REM2D_FLAG_RETILE) return c_fail(REM2D_E_INVALID, "set_order: the world deals its creatures itself (REM2D_FLAG_RETILE)");
    return REM2D_OK;
}
/* launch options (include/rem2d.h REM2D_OPT_*): kept and handed back, nothing to steer on the CPU; same range checks */
int rem2d_cpu_world_set_option(rem2d_cpu_world *w, int32_t key, int32_t value) {
    static const int32_t lo[REM2D_OPT_COUNT] = {0, 0, 0, 0, 0, 1, 0, 0, 0}, hi[REM2D_OPT_COUNT] = {3, 1, 7, 1 << 20, 1 << 20, 64, 1 << 30, 1 << 20, 0x7fffffff};
    if (!w) return c_fail(REM2D_E_INVALID, "world is NULL");
    if (key < 0 || key >= REM2D_OPT_COUNT) return c_fail(REM2D_E_INVALID, "set_option: unknown option");
    if (value < lo[key] || value > hi[key] || (key == REM2D_OPT_PIPELINE && value != 0 && value != 3))
        return c_fail(REM2D_E_INVALID, "set_option: value out of range for this option");
    w->opt[key] = value;
    return REM2D_OK;
}
int rem2d_cpu_world_get_option(const rem2d_cpu_world *w, int32_t key, int32_t *value) {
    if (!w || !value) return c_fail(REM2D_E_INVALID, "get_option: NULL argument");
    if (key < 0 || key >= REM2D_OPT_COUNT) return c_fail(REM2D_E_INVALID, "get_option: unknown option");
    *value = w->opt[key];
    return REM2D_OK;
}
int rem2d_cpu_world_field(const rem2d_cpu_world *w, int32_t field, size_t *offset_bytes, size_t *count, int32_t *dtype) {
    if (!w || field < 0 || field >= REM2D_F_COUNT) return c_fail(REM2D_E_INVALID, "bad field id");
    int dt = 0;
    c_field_place(&w->L, field, offset_bytes, count, &dt);
    if (dtype) *dtype = dt;
    return REM2D_OK;
}

/* oracle worlds -> arena (dynamic fields) */
static void c_sync(rem2d_cpu_world *w) {
    size_t foff[REM2D_F_COUNT];
    for (int f = 0; f < REM2D_F_COUNT; ++f) c_field_place(&w->L, f, &foff[f], NULL, NULL);
    const int K = w->cfg.lanes, Lp = w->L.Lp;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int e = 0; e < w->cfg.n_envs; ++e) {
        const o_world *ow = w->worlds[e];
        if (!ow) continue;
        float st[O_MAX_BODIES * 8], js[O_MAX_BODIES * 6], fat[4];
        rem2d_oracle_get_bodies(ow, st);
        rem2d_oracle_get_joints(ow, js);
        for (int s = 0; s < K; ++s) {
            const int b = w->slotBody[e * K + s];
            const size_t i = (size_t)e * K + s;
            if (b < 0) continue;
            FPTR(float, REM2D_F_PX)[i] = st[b * 8 + 0]; FPTR(float, REM2D_F_PY)[i] = st[b * 8 + 1];
            FPTR(float, REM2D_F_ANG)[i] = st[b * 8 + 2]; FPTR(float, REM2D_F_VX)[i] = st[b * 8 + 3];
            FPTR(float, REM2D_F_VY)[i] = st[b * 8 + 4]; FPTR(float, REM2D_F_W)[i] = st[b * 8 + 5];
            FPTR(float, REM2D_F_SLEEPT)[i] = st[b * 8 + 6];
            FPTR(int32_t, REM2D_F_AWAKE)[i] = (int32_t)st[b * 8 + 7];
            rem2d_oracle_get_fat_aabb(ow, b, fat);
            FPTR(float, REM2D_F_FATLX)[i] = fat[0]; FPTR(float, REM2D_F_FATLY)[i] = fat[1];
            FPTR(float, REM2D_F_FATUX)[i] = fat[2]; FPTR(float, REM2D_F_FATUY)[i] = fat[3];
            if (b > 0) { /* joint b-1 ties body b to its parent (creation order) */
                const float *j = js + (b - 1) * 6;
                FPTR(float, REM2D_F_JIMPX)[i] = j[0]; FPTR(float, REM2D_F_JIMPY)[i] = j[1]; FPTR(float, REM2D_F_JIMPZ)[i] = j[2];
                FPTR(float, REM2D_F_JMOTORIMP)[i] = j[3]; FPTR(float, REM2D_F_JMOTORSPEED)[i] = j[4];
                FPTR(int32_t, REM2D_F_JLIMIT)[i] = (int32_t)j[5];
                FPTR(double, REM2D_F_CISTATE)[i] = ow->joints[b - 1].istate;
            }
            int32_t ci[O_MAX_BODY_CONTACTS * 8];
            float cf[O_MAX_BODY_CONTACTS * 4];
            const int n = rem2d_oracle_get_contacts(ow, b, ci, cf);
            FPTR(int32_t, REM2D_F_CCOUNT)[i] = n;
            for (int k = 0; k < n && k < REM2D_CONTACT_SLOTS; ++k) {
                const size_t q = (size_t)k * Lp + i;
                FPTR(int32_t, REM2D_F_CEDGE)[q] = ci[k * 8 + 0];
                FPTR(int32_t, REM2D_F_CINFO)[q] = ci[k * 8 + 1] | (ci[k * 8 + 2] << 8);
                FPTR(int32_t, REM2D_F_CKEY0)[q] = ci[k * 8 + 4]; FPTR(int32_t, REM2D_F_CKEY1)[q] = ci[k * 8 + 5];
                FPTR(float, REM2D_F_CN0)[q] = cf[k * 4 + 0]; FPTR(float, REM2D_F_CN1)[q] = cf[k * 4 + 1];
                FPTR(float, REM2D_F_CT0)[q] = cf[k * 4 + 2]; FPTR(float, REM2D_F_CT1)[q] = cf[k * 4 + 3];
            }
        }
        FPTR(double, REM2D_F_WOD)[e] = ow->wod;
        FPTR(float, REM2D_F_INVDT0)[e] = (float)ow->inv_dt0;
        FPTR(int32_t, REM2D_F_POSITERS)[e] = ow->lastPositionIterations;
        FPTR(int32_t, REM2D_F_TOIEVENTS)[e] = ow->toiEvents;
    }
}

int rem2d_cpu_world_reset(rem2d_cpu_world *w, const rem2d_morph *m, void *stream) {
    (void)stream;
    if (!w || !m) return c_fail(REM2D_E_INVALID, "world or morphology is NULL");
    if (!w->terrain) return c_fail(REM2D_E_STATE, "rem2d_cpu_world_set_terrain must precede reset");
    size_t foff[REM2D_F_COUNT];
    for (int f = 0; f < REM2D_F_COUNT; ++f) c_field_place(&w->L, f, &foff[f], NULL, NULL);
    const int K = w->cfg.lanes, N = w->cfg.n_envs;
    c_free_worlds(w);
    memset(w->arena, 0, w->L.total);
    o_morph om;
    om.n_envs = N; om.lanes = K;
    om.shape = m->shape; om.hx = m->hx; om.hy = m->hy; om.x = m->x; om.y = m->y; om.angle = m->angle;
    om.parent = m->parent; om.ax = m->ax; om.ay = m->ay; om.bx = m->bx; om.by = m->by;
    om.torque = m->torque; om.lower = m->lower; om.upper = m->upper;
    om.amp = m->amp; om.phase = m->phase; om.freq = m->freq; om.offset = m->offset; om.istate = m->istate;
    unsigned flags = 0;
    if (w->cfg.flags & REM2D_FLAG_CONTINUOUS) flags |= O_FLAG_CONTINUOUS;
    if (w->cfg.flags & REM2D_FLAG_SLEEP_RESET_ALWAYS) flags |= O_FLAG_SLEEP_RESET_ALWAYS;
    if (w->cfg.flags & REM2D_FLAG_NO_SLEEP) flags |= O_FLAG_NO_SLEEP;
    for (int e = 0; e < N; ++e) {
        w->worlds[e] = rem2d_oracle_world_from_morph(w->terrain, &om, e, flags);
        if (!w->worlds[e]) return c_fail(REM2D_E_NOMEM, "world allocation failed");
        float mass[O_MAX_BODIES * 4];
        rem2d_oracle_get_mass(w->worlds[e], mass);
        int nb = 0;
        for (int s = 0; s < K; ++s) {
            const size_t i = (size_t)e * K + s;
            const int b = m->shape[i] != 0 ? nb++ : -1;
            w->slotBody[i] = (int8_t)b;
            FPTR(int32_t, REM2D_F_SHAPE)[i] = m->shape[i];
            FPTR(int32_t, REM2D_F_PARENT)[i] = m->shape[i] != 0 ? m->parent[i] : -1;
            if (b < 0) continue;
            FPTR(float, REM2D_F_HX)[i] = m->hx[i]; FPTR(float, REM2D_F_HY)[i] = m->hy[i];
            FPTR(float, REM2D_F_INVM)[i] = mass[b * 4 + 0]; FPTR(float, REM2D_F_INVI)[i] = mass[b * 4 + 1];
            if (m->jround) FPTR(int32_t, REM2D_F_JROUND)[i] = m->jround[i];
            if (m->parent[i] >= 0) {
                FPTR(float, REM2D_F_JAX)[i] = m->ax[i]; FPTR(float, REM2D_F_JAY)[i] = m->ay[i];
                FPTR(float, REM2D_F_JBX)[i] = m->bx[i]; FPTR(float, REM2D_F_JBY)[i] = m->by[i];
                FPTR(float, REM2D_F_JTORQUE)[i] = m->torque[i];
                FPTR(float, REM2D_F_JLOWER)[i] = m->lower[i]; FPTR(float, REM2D_F_JUPPER)[i] = m->upper[i];
                FPTR(double, REM2D_F_CAMP)[i] = m->amp[i]; FPTR(double, REM2D_F_CPHASE)[i] = m->phase[i];
                FPTR(double, REM2D_F_CFREQ)[i] = m->freq[i]; FPTR(double, REM2D_F_COFFSET)[i] = m->offset[i];
            }
        }
    }
    w->haveReset = 1;
    c_sync(w);
    return REM2D_OK;
}

/* env_bookkeeping of the kernels (gym_rem2d_amd/csrc/rem2d_kernels.h): reward / done, evaluate()'s fitness rule */
static void c_bookkeeping(rem2d_cpu_world *w, const size_t *foff, int e, double reward, int done) {
    FPTR(float, REM2D_F_REWARD)[e] = (float)reward;
    FPTR(int32_t, REM2D_F_DONE)[e] = done;
    if (w->outIndex) {
        const int g = w->outIndex[e];
        w->outReward[g] = (float)reward;
        w->outDone[g] = (uint8_t)done;
    }
    if (done) FPTR(int32_t, REM2D_F_EVERDONE)[e] = 1;
    const int stepIdx = FPTR(int32_t, REM2D_F_STEPS)[e];
    if (!FPTR(int32_t, REM2D_F_FROZEN)[e]) {
        if (reward < -10.0) FPTR(int32_t, REM2D_F_FROZEN)[e] = 1;
        else if (reward > 100.0) {
            FPTR(double, REM2D_F_FITNESS)[e] = reward + (double)(10000 - stepIdx) / 10000.0;
            FPTR(int32_t, REM2D_F_FROZEN)[e] = 1;
        } else if (reward > 0.0) FPTR(double, REM2D_F_FITNESS)[e] = reward;
    }
    FPTR(int32_t, REM2D_F_STEPS)[e] = stepIdx + 1;
}

int rem2d_cpu_world_step_ex(rem2d_cpu_world *w, int32_t n_steps, float dt, int32_t vel_iters, int32_t pos_iters, void *stream) {
    (void)stream;
    if (!w) return c_fail(REM2D_E_INVALID, "world is NULL");
    if (!w->terrain || !w->haveReset) return c_fail(REM2D_E_STATE, "set_terrain and reset must precede step");
    if (n_steps < 0 || vel_iters < 0 || pos_iters < 0) return c_fail(REM2D_E_INVALID, "negative step or iteration count");
    size_t foff[REM2D_F_COUNT];
    for (int f = 0; f < REM2D_F_COUNT; ++f) c_field_place(&w->L, f, &foff[f], NULL, NULL);
    const int K = w->cfg.lanes, N = w->cfg.n_envs, perWave = C_WAVE / K;
    const int skipFrozen = (w->cfg.flags & REM2D_FLAG_SKIP_FROZEN) != 0;
    const int nWaves = (N + perWave - 1) / perWave;
    /* a wavefront (64 / lanes consecutive creatures) is the unit REM2D_FLAG_SKIP_FROZEN works on */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int wv = 0; wv < nWaves; ++wv) {
        const int e0 = wv * perWave, e1 = e0 + perWave < N ? e0 + perWave : N;
        for (int step = 0; step < n_steps; ++step) {
            if (skipFrozen) {
                int all = 1;
                for (int e = e0; e < e1; ++e) all = all && FPTR(int32_t, REM2D_F_FROZEN)[e] != 0;
                if (all) break;
            }
            for (int e = e0; e < e1; ++e) {
                double reward = 0.0;
                int done = 0;
                rem2d_oracle_env_step_ex(w->worlds[e], dt, vel_iters, pos_iters, &reward, &done);
                c_bookkeeping(w, foff, e, reward, done);
            }
        }
    }
    c_sync(w);
    return REM2D_OK;
}
int rem2d_cpu_world_step(rem2d_cpu_world *w, int32_t n_steps, void *stream) {
    return rem2d_cpu_world_step_ex(w, n_steps, (float)(1.0 / FPS), 6 * 30, 2 * 30, stream); /* Modular2DEnv.py:634 */
}
int rem2d_cpu_worlds_step_ex(rem2d_cpu_world *const *worlds, int32_t n_worlds, int32_t n_steps, float dt, int32_t vel_iters,
                             int32_t pos_iters, void *stream) {
    if (!worlds || n_worlds <= 0) return c_fail(REM2D_E_INVALID, "no worlds");
    if (vel_iters < 0 || pos_iters < 0 || vel_iters > 8192 || pos_iters > 8192) return c_fail(REM2D_E_INVALID, "velocity / position iterations must be in 0..8192");
    for (int i = 0; i < n_worlds; ++i) {
        const int rc = rem2d_cpu_world_step_ex(worlds[i], n_steps, dt, vel_iters, pos_iters, stream);
        if (rc != REM2D_OK) return rc;
    }
    return REM2D_OK;
}
int rem2d_cpu_worlds_step(rem2d_cpu_world *const *worlds, int32_t n_worlds, int32_t n_steps, void *stream) {
    return rem2d_cpu_worlds_step_ex(worlds, n_worlds, n_steps, (float)(1.0 / FPS), 6 * 30, 2 * 30, stream);
}
/* timing of GPU launches: nothing to report */
/* rem2d_groups_step: the step groups of a population, one after the other (streams and graphs mean nothing on the host) */
int rem2d_cpu_groups_step_ex(const rem2d_step_group *groups, int32_t n_groups, int32_t n_steps, float dt, int32_t vel_iters,
                             int32_t pos_iters, void *stream, uint32_t flags) {
    (void)flags;
    if (!groups || n_groups <= 0) return c_fail(REM2D_E_INVALID, "no step groups");
    if (n_groups > REM2D_MAX_STEP_GROUPS) return c_fail(REM2D_E_INVALID, "too many step groups");
    for (int g = 0; g < n_groups; ++g) {
        int rc = rem2d_cpu_worlds_step_ex((rem2d_cpu_world *const *)groups[g].worlds, groups[g].n_worlds, n_steps, dt, vel_iters,
                                          pos_iters, stream);
        if (rc != REM2D_OK) return rc;
    }
    return REM2D_OK;
}
int rem2d_cpu_groups_step(const rem2d_step_group *groups, int32_t n_groups, int32_t n_steps, void *stream, uint32_t flags) {
    return rem2d_cpu_groups_step_ex(groups, n_groups, n_steps, (float)(1.0 / FPS), 6 * 30, 2 * 30, stream, flags);
}
int rem2d_cpu_capacity(int32_t *contact_slots, int32_t *solver_slots) {
    if (contact_slots) *contact_slots = REM2D_CONTACT_SLOTS;
    if (solver_slots) *solver_slots = REM2D_SOLVER_SLOTS;
    return REM2D_OK;
}
/* (no tiles and no launches on the host: the shape of a GPU launch has no counterpart here) */
int rem2d_cpu_worlds_launch_info(rem2d_cpu_world *const *worlds, int32_t n_worlds, int32_t *tile_shape, int32_t *fused_velpost) {
    if (!worlds || n_worlds <= 0) return c_fail(REM2D_E_INVALID, "launch_info: bad world list");
    if (tile_shape) *tile_shape = -1;
    if (fused_velpost) *fused_velpost = 0;
    return REM2D_OK;
}
int rem2d_cpu_world_enable_timing(rem2d_cpu_world *w, int32_t on) { (void)on; return w ? REM2D_OK : c_fail(REM2D_E_INVALID, "world is NULL"); }
int rem2d_cpu_world_kernel_time_ms(rem2d_cpu_world *w, double *total_ms, int64_t *launches) {
    if (total_ms) *total_ms = 0.0;
    if (launches) *launches = 0;
    return w ? REM2D_OK : c_fail(REM2D_E_INVALID, "world is NULL");
}
/* (no launches on the host, so no hand-over between them: always 0) */
int rem2d_cpu_world_handover_failures(const rem2d_cpu_world *w, int64_t *count, int32_t clear) {
    (void)clear;
    if (!w || !count) return c_fail(REM2D_E_INVALID, "handover_failures: NULL argument");
    *count = 0;
    return REM2D_OK;
}
int rem2d_cpu_world_step_time_ms(rem2d_cpu_world *w, double *total_ms, int64_t *steps) {
    if (total_ms) *total_ms = 0.0;
    if (steps) *steps = 0;
    return w ? REM2D_OK : c_fail(REM2D_E_INVALID, "world is NULL");
}
