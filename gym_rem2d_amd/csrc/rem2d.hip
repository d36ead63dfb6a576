// rem2d.hip -- MI355X (gfx950) batched 2D rigid-body stepper + its C ABI (include/rem2d.h).
//
// Replaces the world.Step() hot loop of gym_rem2D's Modular2D.step()/reset()
// (gym_rem2D/envs/Modular2DEnv.py:565-653; engine semantics: Box2D 2.3.x, SURVEY.md
// Appendix A) for N independent creatures at once.
//
// Execution model (wave64, no MFMA -- this is VALU/latency bound, not a contraction):
//   * one lane = one rigid body, its revolute joint to the parent body and its contact
//     list; K = 2..32 consecutive lanes = one creature = one Box2D island; a 64-lane
//     wavefront (= one workgroup) steps 64/K creatures in lockstep;
//   * a body's pose/velocity, its joint's effective-mass terms and accumulated impulses and
//     up to REM2D_SOLVER_SLOTS contact constraints live in VGPRs for a whole launch
//     (n_steps steps), so HBM is touched once per launch for them;
//   * contacts only couple a body to static terrain, so all bodies solve their contacts in
//     parallel; joints couple two lanes and are solved in precomputed rounds that keep
//     Box2D's island order among joints sharing a body, exchanging body velocities /
//     positions through a 1.5 KB LDS mailbox -- the result is bit-identical to the sequential
//     Gauss-Seidel sweep of the CPU restatement;
//   * the per-body broadphase pair list (edge index, feature keys, warm-start impulses) is
//     SoA in HBM ([slot][lane], coalesced) and walked with rolled loops.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize (every binary32 operation rounded
// separately, like an x86-64 Box2D build; this is what makes bit-exact parity possible).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rem2d.h"

#define KC REM2D_CONTACT_SLOTS
#define KT REM2D_SOLVER_SLOTS   // touching contacts per body that can enter the solver
#define KR 3                    // ... of which this many are register resident in the velocity loop
#define WAVE 64
#define SCR_WORDS 9 // manifold scratch words per solver slot
// pair-slot map of a body's touching contacts: 5 bits per solver slot (REM2D_CONTACT_SLOTS <= 32), one or two words
static_assert(KC <= 32, "pair-slot indices are packed in 5 bits");
#if REM2D_SOLVER_SLOTS * 5 <= 32
typedef unsigned slotpack_t;
#define SP_WORDS 1
#else
typedef unsigned long long slotpack_t;
#define SP_WORDS 2
static_assert(KT * 5 <= 64, "pair-slot map: at most 12 solver slots");
#endif
#define SP_PUT(s, t) ((slotpack_t)(unsigned)(s) << (5 * (t)))
#define SP_GET(p, t) ((unsigned)((p) >> (5 * (t))) & 0x1fu)

// ---- b2Settings.h constants (same expressions as Box2D so that they fold identically) ----
#define B2_PI 3.14159265359f
#define B2_EPSILON FLT_EPSILON
#define B2_LINEAR_SLOP 0.005f
#define B2_ANGULAR_SLOP (2.0f / 180.0f * B2_PI)
#define B2_POLYGON_RADIUS (2.0f * B2_LINEAR_SLOP)
#define B2_AABB_EXTENSION 0.1f
#define B2_AABB_MULTIPLIER 2.0f
#define B2_MAX_LINEAR_CORRECTION 0.2f
#define B2_MAX_ANGULAR_CORRECTION (8.0f / 180.0f * B2_PI)
#define B2_MAX_TRANSLATION 2.0f
#define B2_MAX_TRANSLATION_SQ (B2_MAX_TRANSLATION * B2_MAX_TRANSLATION)
#define B2_MAX_ROTATION (0.5f * B2_PI)
#define B2_MAX_ROTATION_SQ (B2_MAX_ROTATION * B2_MAX_ROTATION)
#define B2_BAUMGARTE 0.2f
#define B2_TIME_TO_SLEEP 0.5f
#define B2_LINEAR_SLEEP_TOL 0.01f
#define B2_ANGULAR_SLEEP_TOL (2.0f / 180.0f * B2_PI)

enum { SHAPE_NONE = 0, SHAPE_BOX = 1, SHAPE_CIRCLE = 2 };
enum { MF_CIRCLES = 0, MF_FACE_A = 1, MF_FACE_B = 2 };
enum { LIM_INACTIVE = 0, LIM_AT_LOWER = 1, LIM_AT_UPPER = 2, LIM_EQUAL = 3 };
enum { CF_VERTEX = 0, CF_FACE = 1 };

#include "rem2d_state.h"
#include "rem2d_math.h"
#include "rem2d_narrowphase.h"
#include "rem2d_solver.h"
#include "rem2d_toi.h"
#include "rem2d_position.h"
#include "rem2d_kernels.h"
#include "rem2d_pipeline.h"
#include "rem2d_vel4.h"
#include "rem2d_diversity.h"

// =====================================================================================
// host side: handle + C ABI
// =====================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(REM2D_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct rem2d_world {
    rem2d_world_cfg cfg;
    Layout L;
    char *state;
    State S;
    Terrain T;
    float *terrainBuf;
    int *tilesDev; // [nTiles + 1] creature index where each tile of rem2d_vel4_kernel starts
    int nTiles;
    bool haveTerrain, haveReset;
    bool timing;
    double accumMs;
    int64_t launches;
    // event pairs around the dominant kernel of each launch; created by rem2d_world_enable_timing (outside any timed
    // region), recorded by the step calls, read back by rem2d_world_kernel_time_ms
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evPool;
    int evUsed;
    // second bracket: all kernels of one env-step (pre .. toi_heavy) of the tile pipeline
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evPoolStep;
    int tileShape; // REM2D_TILE_SHAPE id of rem2d_vel4_kernel for this world's tiles
    int evUsedStep;
    double accumMsStep;
    int64_t launchesStep;
    hipEvent_t evFork, evJoin; // fork / join edges of rem2d_groups_step (created on first use)
    uint64_t epoch;            // bumped whenever something the kernel arguments embed changes (graph replay key)
    int32_t opt[REM2D_OPT_COUNT]; // launch options (rem2d_world_set_option); results never depend on them
    int64_t stepsQueued;          // env-steps queued so far (the cadence of REM2D_OPT_REBALANCE)
    int64_t stepsAtOrder;         // ... at the step train's last re-ordering launch
    unsigned *trainFlags; size_t trainCap; // rem2d_step_train_kernel's hand-over flags (device; grown on demand, as the first world of a launch)
    unsigned *trainFailures;      // pinned host word: hand-overs of this world's trains that failed (rem2d_world_handover_failures)
    std::vector<int> evStepCount; // env-steps inside each bracket of evPoolStep (a step train's bracket holds a whole launch)
    bool hostOrder;               // rem2d_world_set_order installed an order (REM2D_STATE_ORDERED = hostOrder || REBALANCE > 0)
};
// the kernels go through State::order while the host has installed an order OR the library re-makes one every N steps
static void order_flag_update(rem2d_world *w) {
    if (w->hostOrder || w->opt[REM2D_OPT_REBALANCE] > 0) w->S.flags |= REM2D_STATE_ORDERED;
    else w->S.flags &= ~REM2D_STATE_ORDERED;
}

// Tile shape of rem2d_vel4_kernel (rem2d_world_set_tile_shape; rem2d_vel4.h explains the trade-off):
//   0: 256 bodies, 4 joint sets, 2 contact sets, 2 waves/SIMD   1: 128 bodies, 2 + 1 sets, 4 waves/SIMD (round 4)
//   2: 192 bodies, 3 + 1 sets, 3 waves/SIMD (round 4)
//   3: 64 bodies, 1 + 1 sets, 4 waves/SIMD (default: measured fastest on config 3; the same at 5 waves/SIMD spills: 24.7 M)
//   4: 128 bodies, 2 + 1 sets, 3 waves/SIMD with the static phase -> set map of rounds 2-3: what fixed-morphology populations
//      want (every creature the same schedule: nothing to rotate, and the 8-module chains are 5 % faster with it, 186 vs 176 M)
// Shapes 1-3 place joints flexibly (rem2d_vel4.h FLEX): any tile within 64 joints per set IN ALL fits.
struct TileShape { int sets, passes, csets, flex; };
#define REM2D_TILE_SHAPES 5
#define DEFAULT_TILE_SHAPE 3
static TileShape tile_shape(int id) {
    static const TileShape shapes[REM2D_TILE_SHAPES] = {{4, 4, 2, 0}, {2, 2, 1, 1}, {3, 3, 1, 1}, {1, 1, 1, 1}, {2, 2, 1, 0}};
    return shapes[(id >= 0 && id < REM2D_TILE_SHAPES) ? id : 3];
}
static bool tile_shape_ok(int id) { return id >= 0 && id < REM2D_TILE_SHAPES; }
// which kernel runs a merged launch of worlds planned for different shapes: the one that takes every world's tiles (a tile
// planned under the static phase -> set map fits a flexible kernel of the same size, not the other way round)
static int tile_shape_rank(int id) { return id == 3 ? 0 : (id == 4 ? 1 : (id == 1 ? 2 : (id == 2 ? 3 : 4))); }
// creatures per tile of the default plan (valid for every morphology of `lanes` lanes per creature)
static int default_tile_creatures(const TileShape &shp, int lanes) {
    // flexible shapes: a creature has fewer joints than lanes, so passes * 64 lanes never exceed sets * 64 joints (passes <= sets);
    // four sets = one phase per set: 128 / lanes creatures never have more than 64 joints in a phase
    int per = (shp.flex ? shp.passes * WAVE : (shp.sets >= 4 ? 128 : 64)) / lanes;
    if (per * lanes > shp.passes * WAVE) per = shp.passes * WAVE / lanes;
    return per < 1 ? 1 : per;
}
// Launch options of a world (include/rem2d.h REM2D_OPT_*): defaults and valid ranges.  None of them changes a result; the
// library reads no environment variable -- hosts that want overrides for experiments pass them here (gym_rem2d_amd._lib
// maps REM2D_* variables onto these calls for bench.py and the tools).
//   PIPELINE       3 = tile pipeline pre -> velocity tiles -> post (default), 0 = the fused body-per-lane step kernel of round 1,
//                  kept as an independently written second formulation that the parity suite runs against the same oracle
//   FUSE_VELPOST   velocity tiles and position iterations of a 64-lane block in one launch (rem2d_velpost_kernel) where the
//                  tile tables allow it.  The slowest velocity tile is usually the slowest position block as well, so
//                  max(v + p) is only a little less than max v + max p: +1.0 % on config 3 in 100-step blocks, +1.7 % on the
//                  driver's command (one launch per step and group less to wait for at the join of every call)
//   PRIO           issue priority (s_setprio) for the wavefronts expected to be the long ones of their launch -- bit 1: the
//                  velocity tiles with the most slots per iteration (cost 7 ticks + 10 sub-slots >= PRIO_T1: priority 1,
//                  >= PRIO_T2: priority 3), bit 4: the wavefronts of the TOI solve.  profiles/archive/r03_prio.txt: config 3 +4.6 %
//   HEAVY_PER_WAVE bodies of the TOI work list per wavefront of rem2d_toi_heavy_multi_kernel, 1..64 (one: a wavefront that
//                  holds two runs the union of their code paths)
//   DEBUG          diagnostic builds (-DREM2D_V4_PROBES) only: Vel4Args::dbg
//   REBALANCE      N > 0: every N env-steps the world's creature order is re-made on the device (rem2d_rebalance_kernel: the
//                  creatures that used every position iteration first, a stable partition), 0 = off
//   TRAIN_FAULT    test hook of the step train's hand-over check: s | m << 16 [| 1 << 30] -- the items (step s of a launch, s >= 1,
//                  blocks with block % m == 0; m <= 1: every block) are told that their hand-over failed, or (bit 30) never get
//                  their flag and wait into the limit.  What the kernels COMPUTE does not change; the creatures carry
//                  REM2D_ERR_HANDOVER and the world's failure counter moves (tests/test_handover_gpu.py)
static const int32_t kOptDefault[REM2D_OPT_COUNT] = {3, 2, 5, 60, 75, 1, 0, 0, 0};
static const int32_t kOptMin[REM2D_OPT_COUNT] = {0, 0, 0, 0, 0, 1, 0, 0, 0};
static const int32_t kOptMax[REM2D_OPT_COUNT] = {3, 2, 7, 1 << 20, 1 << 20, WAVE, 1 << 30, 1 << 20, 0x7fffffff};

static uint32_t __float_as_uint_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

extern "C" int rem2d_abi_version(void) { return REM2D_ABI_VERSION; }
// Build identity: the builder (gym_rem2d_amd/_lib.py build(), tools/build_variant.sh) hashes every file under csrc/, include/rem2d.h
// and the compile flags into -DREM2D_BUILD_ID; the loader recomputes the hash from the sources beside it and refuses a library
// built from anything else (the .so files are git-ignored yet travel to the GPU box: a stale one must not be benched silently).
// The marker in front lets a tool read the id from the file without dlopen.
#ifndef REM2D_BUILD_ID
#define REM2D_BUILD_ID "unidentified"
#endif
static const char kBuildId[] = "REM2D_BUILD_ID=" REM2D_BUILD_ID;
extern "C" const char *rem2d_build_id(void) { return kBuildId + 15; }
extern "C" const char *rem2d_last_error(void) { return g_err.c_str(); }
extern "C" int rem2d_capacity(int32_t *contact_slots, int32_t *solver_slots) {
    if (contact_slots) *contact_slots = REM2D_CONTACT_SLOTS;
    if (solver_slots) *solver_slots = REM2D_SOLVER_SLOTS;
    return REM2D_OK;
}

static bool cfg_ok(const rem2d_world_cfg *cfg) {
    if (!cfg || cfg->n_envs <= 0) return false;
    int k = cfg->lanes;
    return k == 2 || k == 4 || k == 8 || k == 16 || k == 32 || k == 64;
}
extern "C" size_t rem2d_state_bytes(const rem2d_world_cfg *cfg) {
    if (!cfg_ok(cfg)) return 0;
    return make_layout(cfg).total;
}
extern "C" int32_t rem2d_padded_envs(const rem2d_world_cfg *cfg) {
    if (!cfg_ok(cfg)) return 0;
    return make_layout(cfg).Np;
}

static uint64_t next_epoch() { // (worlds of different host threads draw from it)
    static std::atomic<uint64_t> e{0};
    return ++e;
}
static void bind_state(rem2d_world *w) {
    State &S = w->S;
    const Layout &L = w->L;
    S.lane4 = w->state + L.groupOff[G_LANE4];
    S.lane8 = w->state + L.groupOff[G_LANE8];
    S.slot4 = w->state + L.groupOff[G_SLOT4];
    S.env4 = w->state + L.groupOff[G_ENV4];
    S.env8 = w->state + L.groupOff[G_ENV8];
    S.Lp = (unsigned)L.Lp;
    S.Np = (unsigned)L.Np;
    S.nEnvs = (unsigned)w->cfg.n_envs;
    S.flags = w->cfg.flags;
    S.outReward = nullptr;
    S.outDone = nullptr;
    S.outIndex = nullptr;
}

extern "C" int rem2d_world_create(const rem2d_world_cfg *cfg, void *state_dev, size_t state_bytes, rem2d_world **out) {
    if (!out) return fail(REM2D_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!cfg_ok(cfg)) return fail(REM2D_E_INVALID, "cfg: n_envs must be > 0 and lanes one of 2,4,8,16,32,64");
    Layout L = make_layout(cfg);
    if (!state_dev || state_bytes < L.total) return fail(REM2D_E_INVALID, "state buffer missing or too small");
    // 32-bit per-lane byte offsets: (slots * Lp + lane) * 4 and (scratch words * Lp + lane) * 4 must fit
    if ((size_t)L.Lp * (SCR_TOTAL_WORDS + 1) * 4 >= ((size_t)1 << 32))
        return fail(REM2D_E_INVALID, "too many lanes for one world (n_envs * lanes must stay below ~5 million)");
    if (((uintptr_t)state_dev & 255) != 0) return fail(REM2D_E_INVALID, "state buffer must be 256-byte aligned");
    HIP_TRY(hipSetDevice(cfg->device));
    rem2d_world *w = new (std::nothrow) rem2d_world();
    if (!w) return fail(REM2D_E_NOMEM, "host allocation failed");
    w->cfg = *cfg;
    w->L = L;
    w->state = (char *)state_dev;
    w->terrainBuf = nullptr;
    w->haveTerrain = w->haveReset = false;
    w->timing = false;
    w->evUsed = 0;
    w->evUsedStep = 0;
    w->accumMsStep = 0.0;
    w->launchesStep = 0;
    w->accumMs = 0.0;
    w->launches = 0;
    w->evFork = w->evJoin = nullptr;
    w->epoch = next_epoch();
    for (int k = 0; k < REM2D_OPT_COUNT; ++k) w->opt[k] = kOptDefault[k];
    w->stepsQueued = 0;
    w->stepsAtOrder = 0;
    w->hostOrder = false;
    w->trainFlags = nullptr;
    w->trainCap = 0;
    w->trainFailures = nullptr;
    bind_state(w);
    w->S.scr = nullptr;
    hipError_t e = hipMalloc((void **)&w->S.scr, ((size_t)SCR_TOTAL_WORDS * L.Lp + L.Lp + 64) * sizeof(float));
    if (e != hipSuccess) {
        delete w;
        return fail(REM2D_E_HIP, std::string("hipMalloc(scratch): ") + hipGetErrorString(e));
    }
    w->S.toiWork = (int *)(w->S.scr + (size_t)SCR_TOTAL_WORDS * L.Lp);
    e = hipMemset(w->S.toiWork, 0, 64 * sizeof(int));
    if (e != hipSuccess) {
        (void)hipFree(w->S.scr);
        delete w;
        return fail(REM2D_E_HIP, std::string("hipMemset(scratch): ") + hipGetErrorString(e));
    }
    // creature order of the post kernel (rem2d_pipeline.h, "Dynamic re-tiling"): both halves start as the identity
    {
        std::vector<int> ord((size_t)2 * L.Np + 4, 0);
        for (int i = 0; i < L.Np; ++i) ord[i] = ord[(size_t)L.Np + i] = i;
        w->S.order = nullptr;
        e = hipMalloc((void **)&w->S.order, ord.size() * sizeof(int));
        if (e == hipSuccess) e = hipMemcpy(w->S.order, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (w->S.order) (void)hipFree(w->S.order);
            (void)hipFree(w->S.scr);
            delete w;
            return fail(REM2D_E_HIP, std::string("hipMalloc(order): ") + hipGetErrorString(e));
        }
    }
    // default tiles of the velocity kernel.  The joints of one schedule phase of a creature are a matching of its tree
    // (<= lanes / 2 of them), so 128 / lanes creatures never have more than 64 joints in a phase;
    // rem2d_world_set_tiles lets the host pack tiles tighter from the actual morphologies.
    {
        w->tileShape = DEFAULT_TILE_SHAPE;
        const int per = default_tile_creatures(tile_shape(w->tileShape), cfg->lanes);
        std::vector<int32_t> ts;
        for (int c = 0; c < L.Np; c += per) ts.push_back(c);
        ts.push_back(L.Np);
        w->tilesDev = nullptr;
        w->nTiles = 0;
        int rc = rem2d_world_set_tiles(w, ts.data(), (int32_t)ts.size() - 1);
        if (rc != REM2D_OK) {
            (void)hipFree(w->S.scr);
            (void)hipFree(w->S.order);
            delete w;
            return rc;
        }
    }
    *out = w;
    return REM2D_OK;
}

extern "C" int rem2d_world_adopt(rem2d_world *w) {
    if (!w) return fail(REM2D_E_INVALID, "world is NULL");
    if (!w->haveTerrain) return fail(REM2D_E_STATE, "rem2d_world_set_terrain must be called before adopt");
    w->haveReset = true; // the arena is the whole state between two steps: the scratch is rewritten by every step
    return REM2D_OK;
}

extern "C" int rem2d_world_set_outputs(rem2d_world *w, float *reward_dev, uint8_t *done_dev, const int32_t *index_dev) {
    if (!w) return fail(REM2D_E_INVALID, "world is NULL");
    if ((reward_dev == nullptr) != (done_dev == nullptr) || (reward_dev == nullptr) != (index_dev == nullptr))
        return fail(REM2D_E_INVALID, "set_outputs: pass all three pointers, or three NULLs to switch it off");
    w->S.outReward = reward_dev;
    w->S.outDone = done_dev;
    w->S.outIndex = index_dev;
    w->epoch = next_epoch();
    return REM2D_OK;
}

extern "C" int rem2d_world_set_tile_shape(rem2d_world *w, int32_t tile_shape_sel) {
    if (!w) return fail(REM2D_E_INVALID, "world is NULL");
    if (!tile_shape_ok(tile_shape_sel)) return fail(REM2D_E_INVALID, "tile shape must be 0 .. 4");
    if (w->tileShape == tile_shape_sel) return REM2D_OK;
    w->tileShape = tile_shape_sel;
    // the tile table in place may not fit the new shape: back to the default plan, valid for every morphology
    const int per = default_tile_creatures(tile_shape(w->tileShape), w->cfg.lanes);
    std::vector<int32_t> ts;
    for (int c = 0; c < w->L.Np; c += per) ts.push_back(c);
    ts.push_back(w->L.Np);
    return rem2d_world_set_tiles(w, ts.data(), (int32_t)ts.size() - 1);
}

extern "C" int rem2d_world_set_order(rem2d_world *w, const int32_t *order_dev, void *stream) {
    if (!w) return fail(REM2D_E_INVALID, "world is NULL");
    if (w->cfg.flags & REM2D_FLAG_RETILE) return fail(REM2D_E_INVALID, "set_order: the world deals its creatures itself (REM2D_FLAG_RETILE)");
    HIP_TRY(hipSetDevice(w->cfg.device));
    if (order_dev) {
        // both halves (the REM2D_FLAG_RETILE machinery reads the first, swaps in the second): what the kernels read stays put
        HIP_TRY(hipMemcpyAsync(w->S.order, order_dev, (size_t)w->L.Np * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIP_TRY(hipMemcpyAsync(w->S.order + w->L.Np, order_dev, (size_t)w->L.Np * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        w->hostOrder = true;
    } else {
        // back to the arena order.  With REM2D_OPT_REBALANCE on the kernels keep going through State::order (the library
        // re-makes it every N steps), so the identity is written there instead of dropping the indirection
        if (w->hostOrder && w->opt[REM2D_OPT_REBALANCE] > 0) {
            std::vector<int> ident((size_t)w->L.Np);
            for (int i = 0; i < w->L.Np; ++i) ident[(size_t)i] = i;
            HIP_TRY(hipDeviceSynchronize()); // (rare call: no kernel of any step group may still read the old order)
            HIP_TRY(hipMemcpy(w->S.order, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(w->S.order + w->L.Np, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        w->hostOrder = false;
    }
    order_flag_update(w);
    w->epoch = next_epoch();
    return REM2D_OK;
}

extern "C" int rem2d_world_set_option(rem2d_world *w, int32_t key, int32_t value) {
    if (!w) return fail(REM2D_E_INVALID, "world is NULL");
    if (key < 0 || key >= REM2D_OPT_COUNT) return fail(REM2D_E_INVALID, "set_option: unknown option");
    if (value < kOptMin[key] || value > kOptMax[key] || (key == REM2D_OPT_PIPELINE && value != 0 && value != 3))
        return fail(REM2D_E_INVALID, "set_option: value out of range for this option");
    if (key == REM2D_OPT_REBALANCE) {
        if (value > 0 && (w->cfg.flags & REM2D_FLAG_RETILE))
            return fail(REM2D_E_INVALID, "set_option: the world deals its creatures itself (REM2D_FLAG_RETILE)");
    }
    if (w->opt[key] != value) {
        w->opt[key] = value;
        w->epoch = next_epoch(); // (a captured replay of the old launch sequence is stale)
    }
    // REBALANCE: the kernels go through State::order from now on (the identity, or the host's order, until the first
    // rebalance) / directly again unless the host has an order installed (rem2d_world_set_order)
    if (key == REM2D_OPT_REBALANCE) order_flag_update(w);
    return REM2D_OK;
}
extern "C" int rem2d_world_get_option(const rem2d_world *w, int32_t key, int32_t *value) {
    if (!w || !value) return fail(REM2D_E_INVALID, "get_option: NULL argument");
    if (key < 0 || key >= REM2D_OPT_COUNT) return fail(REM2D_E_INVALID, "get_option: unknown option");
    *value = w->opt[key];
    return REM2D_OK;
}

extern "C" int rem2d_world_set_tiles(rem2d_world *w, const int32_t *tile_start, int32_t n_tiles) {
    if (!w || !tile_start || n_tiles <= 0) return fail(REM2D_E_INVALID, "tiles: NULL table or no tiles");
    if (tile_start[0] != 0 || tile_start[n_tiles] < w->cfg.n_envs || tile_start[n_tiles] > w->L.Np)
        return fail(REM2D_E_INVALID, "tiles must cover creatures [0, n_envs) (at most the padded count)");
    for (int t = 0; t < n_tiles; ++t) {
        const long long c = (long long)tile_start[t + 1] - tile_start[t];
        if (c <= 0) return fail(REM2D_E_INVALID, "tile starts must be strictly increasing");
        if (c * w->cfg.lanes > tile_shape(w->tileShape).passes * WAVE) return fail(REM2D_E_INVALID, "a tile holds too many lanes for the tile shape in use");
    }
    HIP_TRY(hipSetDevice(w->cfg.device));
    int *dev = nullptr;
    HIP_TRY(hipMalloc((void **)&dev, ((size_t)n_tiles + 1) * sizeof(int)));
    hipError_t e = hipMemcpy(dev, tile_start, ((size_t)n_tiles + 1) * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(dev);
        return fail(REM2D_E_HIP, std::string("hipMemcpy(tiles): ") + hipGetErrorString(e));
    }
    if (w->tilesDev) (void)hipFree(w->tilesDev); // hipFree waits for kernels that may still read the old table
    w->tilesDev = dev;
    w->nTiles = n_tiles;
    w->S.tiles = dev;
    // a regular table (tile t = creatures [t cap, (t + 1) cap), cap dividing the creatures of a 64-lane block -- or a multiple of
    // them: a tile of whole blocks): the velocity + position launch and the step trains can find the tiles of a block
    {
        const int cap = tile_start[1] - tile_start[0], cpb = WAVE / w->cfg.lanes;
        bool regular = w->cfg.lanes <= WAVE && cap > 0 && (cpb % cap == 0 || cap % cpb == 0) && tile_start[n_tiles] <= n_tiles * cap;
        for (int t = 0; t < n_tiles && regular; ++t) regular = tile_start[t] == t * cap;
        w->S.nTiles = n_tiles;
        w->S.tileCap = regular ? cap : 0;
    }
    w->epoch = next_epoch();
    return REM2D_OK;
}

extern "C" int rem2d_plan_tiles(const int32_t *parent, const int32_t *jround, int32_t n_envs, int32_t lanes, int32_t n_padded,
                                int32_t max_creatures, int32_t *tile_start_out, int32_t *n_tiles_out) {
    return rem2d_plan_tiles_shape(parent, jround, n_envs, lanes, n_padded, max_creatures, -1, tile_start_out, n_tiles_out);
}
extern "C" int rem2d_plan_tiles_shape(const int32_t *parent, const int32_t *jround, int32_t n_envs, int32_t lanes,
                                      int32_t n_padded, int32_t max_creatures, int32_t tile_shape_sel, int32_t *tile_start_out,
                                      int32_t *n_tiles_out) {
    if (!parent || !jround || !tile_start_out || !n_tiles_out) return fail(REM2D_E_INVALID, "plan_tiles: NULL argument");
    if (tile_shape_sel != -1 && !tile_shape_ok(tile_shape_sel))
        return fail(REM2D_E_INVALID, "plan_tiles: tile shape must be 0 .. 4 or -1 (the default, 3)");
    const TileShape sh = tile_shape(tile_shape_sel < 0 ? DEFAULT_TILE_SHAPE : tile_shape_sel);
    const int maxLanes = sh.passes * WAVE;
    if (n_envs <= 0 || n_padded < n_envs || lanes <= 0 || lanes > maxLanes) return fail(REM2D_E_INVALID, "plan_tiles: bad shape");
    if (max_creatures <= 0) {
        // (fewer, fuller tiles are fewer wave-instructions: with the step train, which issues at 0.9 of the chip's peak, two-lane
        // creatures 32 to a 64-lane tile are +1.0 % on config 3 over 16 to a tile, 8 to a tile -28 %; under the per-step launches
        // of rounds 2-4, bound by their chain, 16 was the better cap: 46.8 vs 46.4 M then)
        max_creatures = 32;
    }
    const int capBodies = maxLanes / lanes; // creatures per tile by lanes
    const int cap = max_creatures < capBodies ? max_creatures : capBodies;
    int nt = 0;
    tile_start_out[0] = 0;
    int first = 0;          // first creature of the open tile
    int P = 1;              // its period
    int cnt[V4_PHASES];     // its joints per register set (phases ph = s mod sets share set s)
    for (int s = 0; s < V4_PHASES; ++s) cnt[s] = 0;
    auto creature_period = [&](int e) {
        int p = 1;
        if (e < n_envs)
            for (int k = 0; k < lanes; ++k) {
                const int q = (jround[(size_t)e * lanes + k] >> 16) & 0xff;
                p = q > p ? q : p;
            }
        return p;
    };
    auto add_counts = [&](int e, int period, int *c) { // joints of creature e per register set under `period`
        if (e >= n_envs) return;
        for (int k = 0; k < lanes; ++k) {
            const size_t i = (size_t)e * lanes + k;
            if (parent[i] < 0) continue;
            const int ph = (jround[i] & 0xff) % period;
            if (ph < V4_PHASES) c[ph % sh.sets] += 1; // periods beyond V4_PHASES are refused by the kernel (REM2D_ERR_SOLVER_OVERFLOW)
        }
    };
    for (int e = 0; e < n_padded; ++e) {
        const int pe = creature_period(e);
        bool fits = (e - first) < cap;
        int c2[V4_PHASES], P2 = P;
        for (int s = 0; s < V4_PHASES; ++s) c2[s] = 0;
        if (fits) {
            P2 = pe > P ? pe : P;
            if (P2 != P) {
                for (int x = first; x < e; ++x) add_counts(x, P2, c2);
            } else {
                for (int s = 0; s < V4_PHASES; ++s) c2[s] = cnt[s];
            }
            add_counts(e, P2, c2);
            if (sh.flex) { // flexible placement (rem2d_vel4.h FLEX): 64 joints per register set in all
                int all = 0;
                for (int s = 0; s < V4_PHASES; ++s) all += c2[s];
                fits = fits && all <= sh.sets * WAVE;
            } else {
                for (int s = 0; s < V4_PHASES; ++s) fits = fits && c2[s] <= WAVE;
            }
        }
        if (!fits && e > first) { // close the tile before e, start a new one with e
            tile_start_out[++nt] = e;
            first = e;
            P2 = pe;
            for (int s = 0; s < V4_PHASES; ++s) c2[s] = 0;
            add_counts(e, P2, c2);
        }
        P = P2;
        for (int s = 0; s < V4_PHASES; ++s) cnt[s] = c2[s];
    }
    tile_start_out[++nt] = n_padded;
    *n_tiles_out = nt;
    return REM2D_OK;
}

static void drain_timing(rem2d_world *w) {
    // the device is drained once; after that every recorded pair is complete.  (No hipEventSynchronize per event: the
    // start / stop events of hipExtLaunchKernelGGL are not stream records, and waiting on one can block for ever.)
    if (w->evUsed > 0 || w->evUsedStep > 0) (void)hipDeviceSynchronize();
    for (int i = 0; i < w->evUsed; ++i) {
        auto &p = w->evPool[i];
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            w->accumMs += ms;
            w->launches += 1;
        }
    }
    w->evUsed = 0;
    for (int i = 0; i < w->evUsedStep; ++i) {
        auto &p = w->evPoolStep[i];
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            w->accumMsStep += ms;
            w->launchesStep += (size_t)i < w->evStepCount.size() ? w->evStepCount[(size_t)i] : 1;
        }
    }
    w->evUsedStep = 0;
    (void)hipGetLastError(); // an event pair that was never reached leaves hipErrorInvalidHandle / NotReady behind
}
static void free_timing(rem2d_world *w) {
    for (auto *pool : {&w->evPool, &w->evPoolStep}) {
        for (auto &p : *pool) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        pool->clear();
    }
    w->evUsed = w->evUsedStep = 0;
}
// first event of the next free pair, or nullptr when timing is off / the pool is used up (those launches go untimed)
static bool timing_begin(rem2d_world *w, hipStream_t st) {
    if (!w->timing || w->evUsed >= (int)w->evPool.size()) return false;
    return hipEventRecord(w->evPool[w->evUsed].first, st) == hipSuccess;
}
static void timing_end(rem2d_world *w, hipStream_t st) {
    (void)hipEventRecord(w->evPool[w->evUsed].second, st);
    w->evUsed += 1;
}

static void graphs_forget(const rem2d_world *w); // (the hipGraph replays of rem2d_groups_step, below)
extern "C" int rem2d_world_destroy(rem2d_world *w) {
    if (!w) return REM2D_OK;
    (void)hipSetDevice(w->cfg.device);
    graphs_forget(w);
    free_timing(w);
    if (w->evFork) (void)hipEventDestroy(w->evFork);
    if (w->evJoin) (void)hipEventDestroy(w->evJoin);
    if (w->S.scr) (void)hipFree(w->S.scr);
    if (w->S.order) (void)hipFree(w->S.order);
    if (w->tilesDev) (void)hipFree(w->tilesDev);
    if (w->terrainBuf) (void)hipFree(w->terrainBuf);
    if (w->trainFlags) (void)hipFree(w->trainFlags);
    if (w->trainFailures) (void)hipHostFree(w->trainFailures);
    delete w;
    return REM2D_OK;
}

// b2PolygonShape::Set for one hardcore box (2.3.1): weld, gift-wrap hull from the right-most (lowest)
// point, counter-clockwise, edge normals.  Host arithmetic in binary32, no FMA (-ffp-contract=off).
static bool host_poly_set(const float *xy /*[4][2]*/, float vx[4], float vy[4], float nx[4], float ny[4]) {
    float px[4], py[4];
    int n = 0;
    for (int i = 0; i < 4; ++i) {
        float x = xy[2 * i], y = xy[2 * i + 1];
        bool unique = true;
        for (int j = 0; j < n; ++j) {
            float dx = x - px[j], dy = y - py[j];
            if (dx * dx + dy * dy < 0.5f * B2_LINEAR_SLOP) { unique = false; break; }
        }
        if (unique) { px[n] = x; py[n] = y; ++n; }
    }
    if (n != 4) return false;
    int i0 = 0;
    float x0 = px[0];
    for (int i = 1; i < n; ++i) {
        float x = px[i];
        if (x > x0 || (x == x0 && py[i] < py[i0])) { i0 = i; x0 = x; }
    }
    int hull[4], m = 0, ih = i0;
    for (;;) {
        if (m >= 4) return false;
        hull[m] = ih;
        int ie = 0;
        for (int j = 1; j < n; ++j) {
            if (ie == ih) { ie = j; continue; }
            float rx = px[ie] - px[hull[m]], ry = py[ie] - py[hull[m]];
            float wx = px[j] - px[hull[m]], wy = py[j] - py[hull[m]];
            float c = rx * wy - ry * wx;
            if (c < 0.0f) ie = j;
            if (c == 0.0f && wx * wx + wy * wy > rx * rx + ry * ry) ie = j;
        }
        ++m;
        ih = ie;
        if (ie == i0) break;
    }
    if (m != 4) return false;
    for (int i = 0; i < 4; ++i) { vx[i] = px[hull[i]]; vy[i] = py[hull[i]]; }
    for (int i = 0; i < 4; ++i) {
        int i2 = i + 1 < 4 ? i + 1 : 0;
        float ex = vx[i2] - vx[i], ey = vy[i2] - vy[i];
        float tx = 1.0f * ey, ty = -1.0f * ex; // b2Cross(edge, 1.0f)
        float len = sqrtf(tx * tx + ty * ty);
        if (!(len < B2_EPSILON)) {
            float inv = 1.0f / len;
            tx *= inv;
            ty *= inv;
        }
        nx[i] = tx;
        ny[i] = ty;
    }
    return true;
}

extern "C" int rem2d_world_set_terrain(rem2d_world *w, const float *xs, const float *ys, int32_t npts, const float *polys,
                                       int32_t npolys, float friction) {
    if (!w || !xs || !ys || npts < 2) return fail(REM2D_E_INVALID, "terrain needs at least two polyline points");
    if (npolys < 0 || (npolys > 0 && !polys)) return fail(REM2D_E_INVALID, "bad hardcore polygon list");
    HIP_TRY(hipSetDevice(w->cfg.device));
    const int nEdge = npts - 1, nPoly = npolys, nS = nEdge + nPoly;
    std::vector<float> h((size_t)20 * nS, 0.0f);
    float *flx = h.data(), *fly = flx + nS, *fux = fly + nS, *fuy = fux + nS;
    float *vx = fuy + nS, *vy = vx + 4 * nS, *nx = vy + 4 * nS, *ny = nx + 4 * nS;
    const float r = B2_POLYGON_RADIUS;
    for (int i = 0; i < nPoly; ++i) {
        float pvx[4], pvy[4], pnx[4], pny[4];
        if (!host_poly_set(polys + (size_t)i * 8, pvx, pvy, pnx, pny))
            return fail(REM2D_E_INVALID, "hardcore polygons must be convex quads");
        float lx = 0, ly = 0, ux = 0, uy = 0;
        for (int k = 0; k < 4; ++k) {
            vx[k * nS + i] = pvx[k]; vy[k * nS + i] = pvy[k]; nx[k * nS + i] = pnx[k]; ny[k * nS + i] = pny[k];
            // b2PolygonShape::ComputeAABB with the identity transform
            float tx = (1.0f * pvx[k] - 0.0f * pvy[k]) + 0.0f, ty = (0.0f * pvx[k] + 1.0f * pvy[k]) + 0.0f;
            if (k == 0) { lx = ux = tx; ly = uy = ty; }
            else { lx = lx < tx ? lx : tx; ly = ly < ty ? ly : ty; ux = ux > tx ? ux : tx; uy = uy > ty ? uy : ty; }
        }
        flx[i] = (lx - r) - B2_AABB_EXTENSION; fly[i] = (ly - r) - B2_AABB_EXTENSION;
        fux[i] = (ux + r) + B2_AABB_EXTENSION; fuy[i] = (uy + r) + B2_AABB_EXTENSION;
    }
    for (int i = 0; i < nEdge; ++i) {
        const int s = nPoly + i;
        // b2EdgeShape::ComputeAABB with the identity transform, then the broadphase fattening
        float ax = (1.0f * xs[i] - 0.0f * ys[i]) + 0.0f, ay = (0.0f * xs[i] + 1.0f * ys[i]) + 0.0f;
        float bx = (1.0f * xs[i + 1] - 0.0f * ys[i + 1]) + 0.0f, by = (0.0f * xs[i + 1] + 1.0f * ys[i + 1]) + 0.0f;
        vx[0 * nS + s] = xs[i]; vy[0 * nS + s] = ys[i]; vx[1 * nS + s] = xs[i + 1]; vy[1 * nS + s] = ys[i + 1];
        float lx = ax < bx ? ax : bx, ly = ay < by ? ay : by;
        float ux = ax > bx ? ax : bx, uy = ay > by ? ay : by;
        lx = lx - r; ly = ly - r; ux = ux + r; uy = uy + r;
        flx[s] = lx - B2_AABB_EXTENSION; fly[s] = ly - B2_AABB_EXTENSION;
        fux[s] = ux + B2_AABB_EXTENSION; fuy[s] = uy + B2_AABB_EXTENSION;
    }
    float pitch = (xs[npts - 1] - xs[0]) / (float)nEdge;
    if (!(pitch > 0.0f)) return fail(REM2D_E_INVALID, "terrain xs must be increasing");
    // the edge scan assumes a (nearly) uniform pitch; verify so that the candidate range is conservative
    for (int i = 0; i < npts; ++i) {
        float expect = xs[0] + pitch * (float)i;
        if (fabsf(xs[i] - expect) > 0.1f * pitch) return fail(REM2D_E_INVALID, "terrain xs must be uniformly spaced");
    }
    if (w->terrainBuf) { (void)hipFree(w->terrainBuf); w->terrainBuf = nullptr; }
    HIP_TRY(hipMalloc((void **)&w->terrainBuf, h.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(w->terrainBuf, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    Terrain &T = w->T;
    T.nEdge = nEdge;
    T.nPoly = nPoly;
    T.nStatic = nS;
    T.flx = w->terrainBuf; T.fly = T.flx + nS; T.fux = T.fly + nS; T.fuy = T.fux + nS;
    T.vx = T.fuy + nS; T.vy = T.vx + 4 * nS; T.nx = T.vy + 4 * nS; T.ny = T.nx + 4 * nS;
    T.x0 = xs[0];
    T.invPitch = 1.0f / pitch;
    T.friction = sqrtf(friction * 0.1f); // b2MixFriction(terrain fixture, module fixture friction 0.1)
    w->haveTerrain = true;
    w->epoch = next_epoch();
    return REM2D_OK;
}

extern "C" int rem2d_world_reset(rem2d_world *w, const rem2d_morph *m, void *stream) {
    if (!w || !m) return fail(REM2D_E_INVALID, "world or morphology is NULL");
    if (!m->shape || !m->hx || !m->hy || !m->x || !m->y || !m->angle || !m->parent || !m->jround || !m->ax || !m->ay ||
        !m->bx || !m->by || !m->torque || !m->lower || !m->upper || !m->amp || !m->phase || !m->freq || !m->offset ||
        !m->istate)
        return fail(REM2D_E_INVALID, "morphology has NULL arrays");
    HIP_TRY(hipSetDevice(w->cfg.device));
    int threads = 256, blocks = (w->L.Lp + threads - 1) / threads;
    hipLaunchKernelGGL(rem2d_reset_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, w->S, *m, w->cfg.lanes);
    HIP_TRY(hipGetLastError());
    w->haveReset = true;
    return REM2D_OK;
}

static int pipeline_mode(const rem2d_world *w) { return w->opt[REM2D_OPT_PIPELINE] == 0 ? 0 : 3; }
// Worlds of one merged launch run in ONE kernel shape (tiles_plan: the highest tile_shape_rank).  Every mix is sound but one: a
// tile planned for shape 2 (192 bodies, <= 192 joints IN ALL, flexibly placed) may hold more than 64 joints in one schedule
// phase, which the static four-set kernel of shape 0 (one phase per register set) cannot take -- that launch would drop joints.
static bool shapes_mix_ok(rem2d_world *const *ws, int n_worlds) {
    bool has0 = false, has2 = false;
    for (int i = 0; i < n_worlds; ++i) {
        has0 = has0 || ws[i]->tileShape == 0;
        has2 = has2 || ws[i]->tileShape == 2;
    }
    return !(has0 && has2);
}
#define SHAPES_TRY(ws, n) \
    do { if (!shapes_mix_ok((ws), (n))) return fail(REM2D_E_INVALID, "worlds of tile shapes 0 and 2 cannot share a launch (step them in separate calls / groups)"); } while (0)

// The tile pipeline for one or several worlds (lane buckets) in one grid per kernel: pre and post run one body per
// lane; the velocity iterations run one tile per wavefront (rem2d_vel4.h).  TilePlan = the launch arguments of one env-step of
// one step group (they do not change from step to step); tiles_launch_step enqueues that step's kernels.
struct TilePlan {
    Batch B;
    Vel4Batch VB;
    StepArgs A;
    Vel4Args V;
    unsigned blocks, tiles;
    int launchShape;
    bool continuous;
    bool velpost; // one launch for the velocity iterations and post (rem2d_velpost_kernel)
    bool train;   // ... and ONE launch for all steps of a call, pre and the TOI solve included (rem2d_step_train_kernel; REM2D_OPT_FUSE_VELPOST = 2)
    bool train128; // ... the same for the 128-lane tile shapes 1 / 4 (rem2d_step_train128_kernel: an item = a tile's one or two blocks)
    unsigned items; // items per step of that launch: blocks (64-lane train) / tile-sized groups of blocks summed over the worlds (128-lane)
    rem2d_world *w0;
    rem2d_world *ws[REM2D_MAX_BATCH]; // the group's worlds (REM2D_OPT_REBALANCE runs per world)
    int nw;
};
static void tiles_plan(TilePlan &P, rem2d_world *const *ws, int n_worlds, float dt, int vel_iters, int pos_iters) {
    Batch &B = P.B;
    Vel4Batch &VB = P.VB;
    memset(&B, 0, sizeof(B));
    memset(&VB, 0, sizeof(VB));
    unsigned blocks = 0, tiles = 0;
    // tiles of the widest creatures first: their chains are the longest (period 3-4, more joint rounds), so they must
    // not start last
    int order[REM2D_MAX_BATCH];
    for (int i = 0; i < n_worlds; ++i) order[i] = i;
    for (int i = 1; i < n_worlds; ++i)
        for (int j = i; j > 0 && ws[order[j]]->cfg.lanes > ws[order[j - 1]]->cfg.lanes; --j) {
            const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
        }
    for (int i = 0; i < n_worlds; ++i) {
        rem2d_world *w = ws[order[i]];
        B.S[i] = w->S;
        B.T[i] = w->T;
        B.lanes[i] = w->cfg.lanes;
        blocks += (unsigned)w->L.Lp / WAVE;
        B.blockEnd[i] = blocks;
        VB.S[i] = w->S;
        VB.friction[i] = w->T.friction;
        VB.lanes[i] = w->cfg.lanes;
        tiles += (unsigned)w->nTiles;
        VB.tileEnd[i] = tiles;
    }
    B.n = VB.n = n_worlds;
    P.blocks = blocks;
    P.tiles = tiles;
    P.w0 = ws[0];
    P.nw = n_worlds;
    for (int i = 0; i < n_worlds; ++i) P.ws[i] = ws[i];
    // worlds planned for different tile shapes in one grid: the largest shape runs the smaller ones' tiles as well
    // (a tile within 64 joints fits any shape; one within 64 joints per phase pair fits the four-set shape)
    P.launchShape = 3;
    for (int i = 0; i < n_worlds; ++i) {
        const int id = ws[i]->tileShape;
        if (tile_shape_rank(id) > tile_shape_rank(P.launchShape)) P.launchShape = id;
    }
    P.velpost = ws[0]->opt[REM2D_OPT_FUSE_VELPOST] != 0 && P.launchShape == 3;
    for (int i = 0; i < n_worlds; ++i) // (a tile within a block: cap <= the creatures of a block)
        P.velpost = P.velpost && ws[i]->S.tileCap > 0 && ws[i]->S.tileCap * ws[i]->cfg.lanes <= WAVE && ws[i]->tileShape == 3; // (REM2D_FLAG_RETILE: tile slots and block slots go through the same creature order)
    // the step train: where the one-launch form is possible, for worlds whose creature order does not change inside a launch
    // (REM2D_FLAG_RETILE re-deals it in every step, grid-wide) and outside the fused kernel's diagnostics (REM2D_OPT_DEBUG)
    const bool wantTrain = ws[0]->opt[REM2D_OPT_FUSE_VELPOST] == 2 && ws[0]->opt[REM2D_OPT_DEBUG] == 0;
    P.train = P.velpost && wantTrain;
    for (int i = 0; i < n_worlds; ++i) P.train = P.train && !(ws[i]->cfg.flags & REM2D_FLAG_RETILE);
    P.items = blocks;
    // ... and its form for the 128-lane tile shapes (1 flexible, 4 static): regular tables of tiles of at most two blocks
    P.train128 = !P.train && wantTrain && (P.launchShape == 1 || P.launchShape == 4);
    if (P.train128) {
        unsigned items = 0;
        for (int i = 0; i < n_worlds; ++i) {
            const rem2d_world *w = ws[i];
            P.train128 = P.train128 && !(w->cfg.flags & REM2D_FLAG_RETILE) && w->S.tileCap > 0 && w->S.tileCap * w->cfg.lanes <= 2 * WAVE &&
                         (w->tileShape == 3 || w->tileShape == 1 || w->tileShape == 4);
            // the static kernel keeps the ARENA order in its velocity tiles (the host planned them for the creatures they hold) while
            // pre and post follow the creature order: under an order a tile's creatures are not the ones this workgroup's pre / post
            // handle -- legal between launches, not inside one workgroup.  Such worlds keep their per-step launches.
            if (P.launchShape == 4 && (w->S.flags & REM2D_STATE_ORDERED)) P.train128 = false;
            const unsigned lanesPerTile = (unsigned)w->S.tileCap * (unsigned)w->cfg.lanes, bpt = lanesPerTile > WAVE ? lanesPerTile / WAVE : 1u;
            items += ((unsigned)w->L.Lp / WAVE + bpt - 1) / bpt; // (the kernel's item -> world walk does the same sum)
        }
        if (P.train128) P.items = items;
    }
    P.train = P.train || P.train128;
    P.continuous = (ws[0]->cfg.flags & REM2D_FLAG_CONTINUOUS) != 0;
    P.A.nSteps = 1;
    P.A.dt = dt;
    P.A.velIters = vel_iters;
    P.A.posIters = pos_iters;
    // (the options of the group's first world steer the launch: a group is one launch sequence)
    P.A.heavyPerWave = ws[0]->opt[REM2D_OPT_HEAVY_PER_WAVE];
    P.A.prio = P.V.prio = ws[0]->opt[REM2D_OPT_PRIO];
    P.V.prioT1 = ws[0]->opt[REM2D_OPT_PRIO_T1];
    P.V.prioT2 = ws[0]->opt[REM2D_OPT_PRIO_T2];
    P.A.defer = P.continuous ? 2 : 0; // 2: post runs the TOI scan itself (the fused kernel's path keeps 1 = separate scan kernel)
    P.V.velIters = vel_iters;
    P.V.dt = dt;
    P.V.dbg = ws[0]->opt[REM2D_OPT_DEBUG];
}
// one env-step of one step group.  With timing on (rem2d_world_enable_timing; never inside a region whose wall time is
// being measured -- bench.py times kernels in a pass of its own) the dominant kernel gets its own begin / end timestamps
// (hipExtLaunchKernelGGL's start / stop events bracket exactly the kernel, which is what rocprofv3 --kernel-trace
// reports; events recorded on the stream around the launch also count the dispatch gaps) and the whole sequence a pair
// of stream events.
#ifndef REM2D_SHAPE1_WPS
#define REM2D_SHAPE1_WPS 4 // wavefronts per SIMD the 128-body tile shape is compiled for
#endif
#ifndef REM2D_SHAPE4_WPS
#define REM2D_SHAPE4_WPS 3 // the same for the static 128-body shape of the fixed-morphology populations (4: 128 VGPRs, 3 spilled; 65 536 8-module chains 164 instead of 185 M)
#endif
#ifndef REM2D_SHAPE1_PAIR
#define REM2D_SHAPE1_PAIR false // (lane-pair contact solves: a 128-body tile's manifolds rarely fit 32 lanes; without them 3 VGPR spills instead of 7, +2.7 % at 393 216 creatures)
#endif
static void tiles_launch_step(const TilePlan &P, hipStream_t st) {
    rem2d_world *w0 = P.w0;
    const dim3 grid(P.blocks), block(WAVE);
    // (the step's time bracket opens in front of the re-ordering launches: they belong to the step that needs them)
    const bool timedStep = w0->timing && w0->evUsedStep < (int)w0->evPoolStep.size() &&
                           hipEventRecord(w0->evPoolStep[w0->evUsedStep].first, st) == hipSuccess;
    for (int i = 0; i < P.nw; ++i) { // REM2D_OPT_REBALANCE: a new creature order every N env-steps, made from the last step's state
        rem2d_world *w = P.ws[i];
        const int every = w->opt[REM2D_OPT_REBALANCE];
        if (every > 0 && w->stepsQueued > 0 && w->stepsQueued % every == 0 && (w->S.flags & REM2D_STATE_ORDERED))
        {
            int threads = WAVE;
            while (threads < REBALANCE_MAX_THREADS && threads * 256 < w->cfg.n_envs) threads *= 2;
            hipLaunchKernelGGL(rem2d_rebalance_kernel, dim3(1), dim3(threads), 0, st, w->S, P.A.posIters);
        }
        w->stepsQueued += 1;
    }
    // (a merged launch of shapes 1 and 4: rank picks 1, whose flexible kernel takes the statically planned tiles too)
    if (P.launchShape == 3 || (P.launchShape == 1 && REM2D_SHAPE1_WPS >= 4)) hipLaunchKernelGGL(rem2d_pre_multi_kernel<4>, grid, block, 0, st, P.B, P.A);
    else hipLaunchKernelGGL(rem2d_pre_multi_kernel<3>, grid, block, 0, st, P.B, P.A);
    const bool timed = w0->timing && w0->evUsed < (int)w0->evPool.size();
    if (P.velpost) {
        if (timed) {
            hipExtLaunchKernelGGL(rem2d_velpost_kernel, grid, dim3(WAVE * REM2D_VELPOST_WAVES), 0, st, w0->evPool[w0->evUsed].first,
                                  w0->evPool[w0->evUsed].second, 0, P.B, P.A, P.V);
            w0->evUsed += 1;
        } else {
            hipLaunchKernelGGL(rem2d_velpost_kernel, grid, dim3(WAVE * REM2D_VELPOST_WAVES), 0, st, P.B, P.A, P.V);
        }
    } else {
        if (timed) {
            hipEvent_t e0 = w0->evPool[w0->evUsed].first, e1 = w0->evPool[w0->evUsed].second;
            switch (P.launchShape) {
            case 0: hipExtLaunchKernelGGL((rem2d_vel4_kernel<4, 4, 2, 2, false>), dim3(P.tiles), block, 0, st, e0, e1, 0, P.VB, P.V); break;
            case 1: hipExtLaunchKernelGGL((rem2d_vel4_kernel<2, 2, 1, REM2D_SHAPE1_WPS, REM2D_SHAPE1_PAIR>), dim3(P.tiles), block, 0, st, e0, e1, 0, P.VB, P.V); break;
            case 2: hipExtLaunchKernelGGL((rem2d_vel4_kernel<3, 3, 1, 3, true>), dim3(P.tiles), block, 0, st, e0, e1, 0, P.VB, P.V); break;
            case 4: hipExtLaunchKernelGGL((rem2d_vel4_kernel<2, 2, 1, REM2D_SHAPE4_WPS, false, false>), dim3(P.tiles), block, 0, st, e0, e1, 0, P.VB, P.V); break;
            default: hipExtLaunchKernelGGL((rem2d_vel4_kernel<1, 1, 1, 4, true>), dim3(P.tiles), block, 0, st, e0, e1, 0, P.VB, P.V); break;
            }
            w0->evUsed += 1;
        } else {
            switch (P.launchShape) {
            case 0: hipLaunchKernelGGL((rem2d_vel4_kernel<4, 4, 2, 2, false>), dim3(P.tiles), block, 0, st, P.VB, P.V); break;
            case 1: hipLaunchKernelGGL((rem2d_vel4_kernel<2, 2, 1, REM2D_SHAPE1_WPS, REM2D_SHAPE1_PAIR>), dim3(P.tiles), block, 0, st, P.VB, P.V); break;
            case 2: hipLaunchKernelGGL((rem2d_vel4_kernel<3, 3, 1, 3, true>), dim3(P.tiles), block, 0, st, P.VB, P.V); break;
            case 4: hipLaunchKernelGGL((rem2d_vel4_kernel<2, 2, 1, REM2D_SHAPE4_WPS, false, false>), dim3(P.tiles), block, 0, st, P.VB, P.V); break;
            default: hipLaunchKernelGGL((rem2d_vel4_kernel<1, 1, 1, 4, true>), dim3(P.tiles), block, 0, st, P.VB, P.V); break;
            }
        }
        hipLaunchKernelGGL(rem2d_post_multi_kernel, grid, block, 0, st, P.B, P.A);
    }
    if (P.continuous) hipLaunchKernelGGL(rem2d_toi_heavy_multi_kernel, grid, block, 0, st, P.B, P.A);
    if (timedStep) {
        (void)hipEventRecord(w0->evPoolStep[w0->evUsedStep].second, st);
        if (w0->evStepCount.size() < w0->evPoolStep.size()) w0->evStepCount.resize(w0->evPoolStep.size(), 1);
        w0->evStepCount[(size_t)w0->evUsedStep] = 1;
        w0->evUsedStep += 1;
    }
}
// The step train (rem2d_step_train_kernel): the steps of a call in one launch -- or in one launch per stretch between two
// re-orderings of the creature order (REM2D_OPT_REBALANCE), which run in front of the stretch they are due for.
// (the flag buffer and the failure counter of a train's first world: sized before anything of the launch is queued -- growing frees
// the old buffer, which waits for the launches that use it)
static int train_reserve(TilePlan &P) {
    rem2d_world *w0 = P.w0;
    const size_t need = TRAIN_FLAG_WORDS + (size_t)P.items;
    if (w0->trainCap < need) {
        if (w0->trainFlags) HIP_TRY(hipFree(w0->trainFlags));
        w0->trainFlags = nullptr; w0->trainCap = 0;
        HIP_TRY(hipMalloc(&w0->trainFlags, need * sizeof(unsigned)));
        w0->trainCap = need;
    }
    if (!w0->trainFailures) {
        HIP_TRY(hipHostMalloc((void **)&w0->trainFailures, sizeof(unsigned), hipHostMallocDefault));
        *w0->trainFailures = 0u;
    }
    return REM2D_OK;
}
static int tiles_launch_train(TilePlan &P, hipStream_t st, int n_steps) {
    rem2d_world *w0 = P.w0;
    const size_t need = TRAIN_FLAG_WORDS + (size_t)P.items;
    { int rc = train_reserve(P); if (rc != REM2D_OK) return rc; }
    const unsigned nPad = (P.items + 7u) & ~7u;
    const int fault = w0->opt[REM2D_OPT_TRAIN_FAULT];
    // The creature order is re-made in front of a launch once N (REM2D_OPT_REBALANCE) or more steps have run since the last time --
    // a cadence in launches, not in steps: a call is cut only where it is itself longer than N steps, never because a multiple
    // of N falls inside it (two short trains drain twice).  Any cadence gives the same bits.
    int l = 0;
    while (l < n_steps) {
        int seg = n_steps - l;
        if ((unsigned long long)nPad * (unsigned)seg > 0x7fffffffull) seg = (int)(0x7fffffffull / nPad);
        if (seg > (int)TRAIN_STEP_MASK) seg = (int)TRAIN_STEP_MASK; // (the flag's step field; nPad >= 8 keeps seg below it anyway)
        for (int i = 0; i < P.nw; ++i) {
            const int every = P.ws[i]->opt[REM2D_OPT_REBALANCE];
            if (every > 0 && seg > every) seg = every;
        }
        const bool timedStep = w0->timing && w0->evUsedStep < (int)w0->evPoolStep.size() &&
                               hipEventRecord(w0->evPoolStep[w0->evUsedStep].first, st) == hipSuccess;
        for (int i = 0; i < P.nw; ++i) {
            rem2d_world *w = P.ws[i];
            const int every = w->opt[REM2D_OPT_REBALANCE];
            if (every > 0 && w->stepsQueued > 0 && w->stepsQueued - w->stepsAtOrder >= every && (w->S.flags & REM2D_STATE_ORDERED)) {
                int threads = WAVE;
                while (threads < REBALANCE_MAX_THREADS && threads * 256 < w->cfg.n_envs) threads *= 2;
                hipLaunchKernelGGL(rem2d_rebalance_kernel, dim3(1), dim3(threads), 0, st, w->S, P.A.posIters);
                w->stepsAtOrder = w->stepsQueued;
            }
            w->stepsQueued += seg;
        }
        HIP_TRY(hipMemsetAsync(w0->trainFlags, 0, need * sizeof(unsigned), st));
        P.A.nSteps = seg;
        const dim3 grid(nPad * (unsigned)seg), block(WAVE);
        const bool timed = w0->timing && w0->evUsed < (int)w0->evPool.size();
        hipEvent_t e0 = timed ? w0->evPool[w0->evUsed].first : nullptr, e1 = timed ? w0->evPool[w0->evUsed].second : nullptr;
#define TRAIN_LAUNCH(KERNEL)                                                                                                              \
        do {                                                                                                                              \
            if (timed) hipExtLaunchKernelGGL(KERNEL, grid, block, 0, st, e0, e1, 0, P.B, P.A, P.V, w0->trainFlags, P.items, fault, w0->trainFailures); \
            else hipLaunchKernelGGL(KERNEL, grid, block, 0, st, P.B, P.A, P.V, w0->trainFlags, P.items, fault, w0->trainFailures);          \
        } while (0)
        if (!P.train128) TRAIN_LAUNCH(rem2d_step_train_kernel);
        else if (P.launchShape == 4) TRAIN_LAUNCH((rem2d_step_train128_kernel<false, REM2D_SHAPE4_WPS>));
        else TRAIN_LAUNCH((rem2d_step_train128_kernel<true, REM2D_SHAPE1_WPS>));
#undef TRAIN_LAUNCH
        if (timed) w0->evUsed += 1;
        if (timedStep) {
            (void)hipEventRecord(w0->evPoolStep[w0->evUsedStep].second, st);
            if (w0->evStepCount.size() < w0->evPoolStep.size()) w0->evStepCount.resize(w0->evPoolStep.size(), 1);
            w0->evStepCount[(size_t)w0->evUsedStep] = seg; // (this bracket holds `seg` env-steps)
            w0->evUsedStep += 1;
        }
        l += seg;
    }
    P.A.nSteps = 1;
    HIP_TRY(hipGetLastError());
    return REM2D_OK;
}
static int step_tiles(rem2d_world *const *ws, int n_worlds, int n_steps, float dt, int vel_iters, int pos_iters, hipStream_t st) {
    TilePlan P;
    tiles_plan(P, ws, n_worlds, dt, vel_iters, pos_iters);
    if (P.train) return tiles_launch_train(P, st, n_steps);
    for (int l = 0; l < n_steps; ++l) tiles_launch_step(P, st);
    HIP_TRY(hipGetLastError());
    return REM2D_OK;
}

// The fused formulation (REM2D_PIPELINE=0; round 1's body-per-lane step kernel + the TOI kernels) for one or several
// worlds in one grid per kernel: n_steps env-steps per launch in discrete mode, one per launch when the TOI kernels follow.
static int step_fused(rem2d_world *const *ws, int n_worlds, int n_steps, float dt, int vel_iters, int pos_iters, hipStream_t st) {
    Batch B;
    memset(&B, 0, sizeof(B));
    unsigned blocks = 0;
    for (int i = 0; i < n_worlds; ++i) {
        rem2d_world *w = ws[i];
        B.S[i] = w->S;
        B.T[i] = w->T;
        B.lanes[i] = w->cfg.lanes;
        blocks += (unsigned)w->L.Lp / WAVE;
        B.blockEnd[i] = blocks;
    }
    B.n = n_worlds;
    rem2d_world *w0 = ws[0];
    const bool continuous = (w0->cfg.flags & REM2D_FLAG_CONTINUOUS) != 0;
    StepArgs A;
    A.nSteps = continuous ? 1 : n_steps;
    A.dt = dt;
    A.velIters = vel_iters;
    A.posIters = pos_iters;
    A.heavyPerWave = w0->opt[REM2D_OPT_HEAVY_PER_WAVE];
    A.prio = w0->opt[REM2D_OPT_PRIO];
    A.defer = continuous ? 1 : 0;
    dim3 grid(blocks), block(WAVE);
    const int launches = continuous ? n_steps : 1;
    for (int l = 0; l < launches; ++l) {
        const bool timed = timing_begin(w0, st);
        hipLaunchKernelGGL(rem2d_step_multi_kernel, grid, block, 0, st, B, A);
        if (timed) timing_end(w0, st);
        if (continuous) {
            hipLaunchKernelGGL(rem2d_toi_scan_multi_kernel, grid, block, 0, st, B, A);
            hipLaunchKernelGGL(rem2d_toi_heavy_multi_kernel, grid, block, 0, st, B, A);
        }
    }
    HIP_TRY(hipGetLastError());
    return REM2D_OK;
}

extern "C" int rem2d_worlds_launch_info(rem2d_world *const *ws, int32_t n_worlds, int32_t *tile_shape_out, int32_t *fused_velpost) {
    if (!ws || n_worlds <= 0 || n_worlds > REM2D_MAX_BATCH) return fail(REM2D_E_INVALID, "launch_info: bad world list");
    for (int i = 0; i < n_worlds; ++i)
        if (!ws[i]) return fail(REM2D_E_INVALID, "world is NULL");
    if (pipeline_mode(ws[0]) != 3) { // the fused step kernel: no tiles
        if (tile_shape_out) *tile_shape_out = -1;
        if (fused_velpost) *fused_velpost = 0;
        return REM2D_OK;
    }
    static TilePlan P; // (large: kernel arguments of a whole group)
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    SHAPES_TRY(ws, n_worlds);
    tiles_plan(P, ws, n_worlds, 1.0f / 50.0f, 180, 60);
    if (tile_shape_out) *tile_shape_out = P.launchShape;
    if (fused_velpost) *fused_velpost = P.train ? 2 : (P.velpost ? 1 : 0); // (2 with tile shape 1 / 4: rem2d_step_train128_kernel)
    return REM2D_OK;
}

// The solver loops count ticks (iteration x schedule period + joint round) in 16 bits (wave_max_active): refuse iteration
// counts that would not fit instead of silently running fewer sweeps.  (Box2D's own arguments are 8 / 3; the reference
// passes 180 / 60, Modular2DEnv.py:634.)
#define REM2D_MAX_ITERS 8192
static bool iters_ok(int vel_iters, int pos_iters) {
    return vel_iters >= 0 && pos_iters >= 0 && vel_iters <= REM2D_MAX_ITERS && pos_iters <= REM2D_MAX_ITERS;
}
#define ITERS_TRY(v, p) \
    do { if (!iters_ok((v), (p))) return fail(REM2D_E_INVALID, "velocity / position iterations must be in 0..8192"); } while (0)

extern "C" int rem2d_world_step_ex(rem2d_world *w, int32_t n_steps, float dt, int32_t vel_iters, int32_t pos_iters,
                                   void *stream) {
    if (!w) return fail(REM2D_E_INVALID, "world is NULL");
    ITERS_TRY(vel_iters, pos_iters);
    if (!w->haveTerrain) return fail(REM2D_E_STATE, "rem2d_world_set_terrain must be called before step");
    if (!w->haveReset) return fail(REM2D_E_STATE, "rem2d_world_reset must be called before step");
    if (n_steps <= 0) return REM2D_OK;
    HIP_TRY(hipSetDevice(w->cfg.device));
    if (pipeline_mode(w) == 3) return step_tiles(&w, 1, n_steps, dt, vel_iters, pos_iters, (hipStream_t)stream);
    return step_fused(&w, 1, n_steps, dt, vel_iters, pos_iters, (hipStream_t)stream);
}
extern "C" int rem2d_worlds_step_ex(rem2d_world *const *ws, int32_t n_worlds, int32_t n_steps, float dt, int32_t vel_iters,
                                    int32_t pos_iters, void *stream) {
    if (!ws || n_worlds <= 0) return fail(REM2D_E_INVALID, "no worlds");
    ITERS_TRY(vel_iters, pos_iters);
    if (n_worlds == 1) return rem2d_world_step_ex(ws[0], n_steps, dt, vel_iters, pos_iters, stream);
    if (n_worlds > REM2D_MAX_WORLDS_PER_STEP) return fail(REM2D_E_INVALID, "too many worlds for one launch");
    static_assert(REM2D_MAX_WORLDS_PER_STEP == REM2D_MAX_BATCH, "batch size");
    for (int i = 0; i < n_worlds; ++i) {
        rem2d_world *w = ws[i];
        if (!w) return fail(REM2D_E_INVALID, "world is NULL");
        if (!w->haveTerrain) return fail(REM2D_E_STATE, "rem2d_world_set_terrain must be called before step");
        if (!w->haveReset) return fail(REM2D_E_STATE, "rem2d_world_reset must be called before step");
        if (w->cfg.device != ws[0]->cfg.device) return fail(REM2D_E_INVALID, "worlds of one launch must share the device");
        if ((w->cfg.flags & REM2D_FLAG_CONTINUOUS) != (ws[0]->cfg.flags & REM2D_FLAG_CONTINUOUS))
            return fail(REM2D_E_INVALID, "worlds of one launch must agree on REM2D_FLAG_CONTINUOUS");
    }
    if (n_steps <= 0) return REM2D_OK;
    rem2d_world *w0 = ws[0];
    HIP_TRY(hipSetDevice(w0->cfg.device));
    if (pipeline_mode(w0) == 3) {
        SHAPES_TRY(ws, n_worlds);
        return step_tiles(ws, n_worlds, n_steps, dt, vel_iters, pos_iters, (hipStream_t)stream);
    }
    return step_fused(ws, n_worlds, n_steps, dt, vel_iters, pos_iters, (hipStream_t)stream);
}
extern "C" int rem2d_worlds_step(rem2d_world *const *ws, int32_t n_worlds, int32_t n_steps, void *stream) {
    return rem2d_worlds_step_ex(ws, n_worlds, n_steps, (float)(1.0 / 50), 6 * 30, 2 * 30, stream);
}
extern "C" int rem2d_world_step(rem2d_world *w, int32_t n_steps, void *stream) {
    // Modular2DEnv.py:634  self.world.Step(1.0/FPS, 6*30, 2*30)
    return rem2d_world_step_ex(w, n_steps, (float)(1.0 / 50), 6 * 30, 2 * 30, stream);
}

// ---- all step groups of a population in one call (include/rem2d.h) ----
// Graph replay: the launches of one call (n_steps x groups x 3-4 kernels, fork / join edges between the streams) are
// captured once and replayed with one hipGraphLaunch per call; the kernel arguments are by-value structs that only
// change when a world's tile table / outputs / terrain change (epoch).
struct GraphEntry {
    uint64_t key;
    hipGraphExec_t exec;
    hipGraph_t graph;
    std::vector<const rem2d_world *> worlds; // whose kernel arguments and fork / join events the replay holds
};
static std::vector<GraphEntry> g_graphs;
// one lock for the replay cache and the capture stream: ctypes releases the GIL around the ABI calls, so two host threads
// may step / destroy different envs at once.  Held across capture, instantiate and launch (a capture in Relaxed mode on
// the one shared capture stream must not interleave with another) and while a destroy drops the replays of its world.
static std::mutex g_graphMu;
static void graph_entry_free(GraphEntry &e) {
    (void)hipDeviceSynchronize(); // (a replay may still be in flight)
    (void)hipGraphExecDestroy(e.exec);
    (void)hipGraphDestroy(e.graph);
}
// a world is going away: so must every replay that was captured with it (its events and scratch pointers)
static void graphs_forget(const rem2d_world *w) {
    std::lock_guard<std::mutex> lk(g_graphMu);
    for (size_t i = 0; i < g_graphs.size();) {
        bool uses = false;
        for (const rem2d_world *x : g_graphs[i].worlds) uses = uses || x == w;
        if (uses) {
            graph_entry_free(g_graphs[i]);
            g_graphs.erase(g_graphs.begin() + (long)i);
        } else {
            ++i;
        }
    }
}
static hipStream_t g_captureStream = nullptr;
static uint64_t mix64(uint64_t h, uint64_t v) {
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    return h;
}

static int groups_enqueue(const rem2d_step_group *groups, int n_groups, int n_steps, float dt, int vel_iters, int pos_iters,
                          hipStream_t origin, bool tiles) {
    // (a step train's flag buffer is sized BEFORE the fork: growing it frees the old one, a device-wide wait that does not belong
    // between a fork and its join, and an error here must not leave forked streams behind)
    std::vector<TilePlan> plans(tiles ? (size_t)n_groups : (size_t)0);
    if (tiles)
        for (int g = 0; g < n_groups; ++g) {
            tiles_plan(plans[g], groups[g].worlds, groups[g].n_worlds, dt, vel_iters, pos_iters);
            if (plans[g].train) { int rc = train_reserve(plans[g]); if (rc != REM2D_OK) return rc; }
        }
    // fork: every group stream waits for what the origin stream has queued so far
    rem2d_world *w00 = groups[0].worlds[0];
    bool forked = false;
    int rcAll = REM2D_OK; // (an error after the fork still runs the join: no group stream is left un-joined behind an error return)
    for (int g = 0; g < n_groups; ++g) {
        hipStream_t sg = groups[g].stream ? (hipStream_t)groups[g].stream : origin;
        if (sg == origin) continue;
        if (!forked) {
            HIP_TRY(hipEventRecord(w00->evFork, origin));
            forked = true;
        }
        HIP_TRY(hipStreamWaitEvent(sg, w00->evFork, 0));
    }
    if (tiles) {
        // round-robin: step l of every group is queued before step l + 1 of any, so that no group's stream runs dry while
        // the host is still busy queueing another group's whole train (and the groups start together)
        for (int g = 0; g < n_groups && rcAll == REM2D_OK; ++g) // (a group whose steps go in one launch: queued whole)
            if (plans[g].train) rcAll = tiles_launch_train(plans[g], groups[g].stream ? (hipStream_t)groups[g].stream : origin, n_steps);
        for (int l = 0; l < n_steps && rcAll == REM2D_OK; ++l)
            for (int g = 0; g < n_groups; ++g)
                if (!plans[g].train) tiles_launch_step(plans[g], groups[g].stream ? (hipStream_t)groups[g].stream : origin);
        if (rcAll == REM2D_OK) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) rcAll = fail(REM2D_E_HIP, std::string("groups_enqueue: ") + hipGetErrorString(e));
        }
    } else {
        for (int g = 0; g < n_groups && rcAll == REM2D_OK; ++g)
            rcAll = step_fused(groups[g].worlds, groups[g].n_worlds, n_steps, dt, vel_iters, pos_iters,
                               groups[g].stream ? (hipStream_t)groups[g].stream : origin);
    }
    // join: the origin stream waits for every group
    for (int g = 0; g < n_groups; ++g) {
        hipStream_t sg = groups[g].stream ? (hipStream_t)groups[g].stream : origin;
        if (sg == origin) continue;
        rem2d_world *wg = groups[g].worlds[0];
        hipError_t e = hipEventRecord(wg->evJoin, sg);
        if (e == hipSuccess) e = hipStreamWaitEvent(origin, wg->evJoin, 0);
        if (e != hipSuccess && rcAll == REM2D_OK) rcAll = fail(REM2D_E_HIP, std::string("groups_enqueue (join): ") + hipGetErrorString(e));
    }
    return rcAll;
}

extern "C" int rem2d_groups_step_ex(const rem2d_step_group *groups, int32_t n_groups, int32_t n_steps, float dt,
                                    int32_t vel_iters, int32_t pos_iters, void *stream, uint32_t flags) {
    if (!groups || n_groups <= 0) return fail(REM2D_E_INVALID, "no step groups");
    if (n_groups > REM2D_MAX_STEP_GROUPS) return fail(REM2D_E_INVALID, "too many step groups");
    ITERS_TRY(vel_iters, pos_iters);
    rem2d_world *w00 = nullptr;
    bool timing = false;
    for (int g = 0; g < n_groups; ++g) {
        if (!groups[g].worlds || groups[g].n_worlds <= 0) return fail(REM2D_E_INVALID, "a step group has no worlds");
        if (groups[g].n_worlds > REM2D_MAX_WORLDS_PER_STEP) return fail(REM2D_E_INVALID, "too many worlds in one step group");
        for (int i = 0; i < groups[g].n_worlds; ++i) {
            rem2d_world *w = groups[g].worlds[i];
            if (!w) return fail(REM2D_E_INVALID, "world is NULL");
            if (!w00) w00 = w;
            if (!w->haveTerrain) return fail(REM2D_E_STATE, "rem2d_world_set_terrain must be called before step");
            if (!w->haveReset) return fail(REM2D_E_STATE, "rem2d_world_reset must be called before step");
            if (w->cfg.device != w00->cfg.device) return fail(REM2D_E_INVALID, "step groups of one call must share the device");
            if ((w->cfg.flags & REM2D_FLAG_CONTINUOUS) != (groups[g].worlds[0]->cfg.flags & REM2D_FLAG_CONTINUOUS))
                return fail(REM2D_E_INVALID, "worlds of one launch must agree on REM2D_FLAG_CONTINUOUS");
            timing = timing || w->timing;
        }
    }
    if (n_steps <= 0) return REM2D_OK;
    HIP_TRY(hipSetDevice(w00->cfg.device));
    for (int g = 0; g < n_groups; ++g) {
        rem2d_world *wg = groups[g].worlds[0];
        if (!wg->evFork) HIP_TRY(hipEventCreateWithFlags(&wg->evFork, hipEventDisableTiming));
        if (!wg->evJoin) HIP_TRY(hipEventCreateWithFlags(&wg->evJoin, hipEventDisableTiming));
    }
    const bool tiles = pipeline_mode(w00) == 3;
    if (tiles)
        for (int g = 0; g < n_groups; ++g) SHAPES_TRY(groups[g].worlds, groups[g].n_worlds);
    hipStream_t origin = (hipStream_t)stream;
    bool train = false; // (a step train is one launch per call already, and sizes its flag buffer while it is queued: never captured)
    if (tiles)
        for (int g = 0; g < n_groups; ++g) {
            TilePlan P;
            tiles_plan(P, groups[g].worlds, groups[g].n_worlds, dt, vel_iters, pos_iters);
            train = train || P.train;
        }
    if (!(flags & REM2D_STEP_GRAPH) || timing || !tiles || train)
        return groups_enqueue(groups, n_groups, n_steps, dt, vel_iters, pos_iters, origin, tiles);

    // ---- graph replay ----
    std::lock_guard<std::mutex> graphLock(g_graphMu);
    uint64_t key = mix64(0x5bd1e995u, (uint64_t)n_steps);
    key = mix64(key, (uint64_t)__float_as_uint_host(dt));
    key = mix64(key, ((uint64_t)(uint32_t)vel_iters << 32) | (uint32_t)pos_iters);
    for (int g = 0; g < n_groups; ++g) {
        key = mix64(key, (uint64_t)(uintptr_t)groups[g].stream);
        for (int i = 0; i < groups[g].n_worlds; ++i) {
            const rem2d_world *w = groups[g].worlds[i];
            key = mix64(key, (uint64_t)(uintptr_t)w);
            key = mix64(key, w->epoch);
            // REM2D_OPT_REBALANCE: which steps of the call carry a re-ordering launch depends on where the call starts in the
            // cadence -- part of the key, so that a replay re-orders every N env-steps exactly like the plain enqueue does
            const int every = w->opt[REM2D_OPT_REBALANCE];
            if (every > 0) key = mix64(key, 0xabcd0000ull + (uint64_t)(w->stepsQueued % every));
        }
        key = mix64(key, 0xfeedull + (uint64_t)groups[g].n_worlds);
    }
    GraphEntry *hit = nullptr;
    for (auto &e : g_graphs)
        if (e.key == key) hit = &e;
    if (!hit) {
        if (!g_captureStream) HIP_TRY(hipStreamCreateWithFlags(&g_captureStream, hipStreamNonBlocking));
        // the capture starts on a stream of the library's own (the caller's may be the NULL stream, which cannot capture);
        // group streams are pulled into the capture by the fork events
        std::vector<rem2d_step_group> cg(groups, groups + n_groups);
        HIP_TRY(hipStreamBeginCapture(g_captureStream, hipStreamCaptureModeRelaxed));
        int rc = groups_enqueue(cg.data(), n_groups, n_steps, dt, vel_iters, pos_iters, g_captureStream, true);
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(g_captureStream, &graph);
        if (rc != REM2D_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (e != hipSuccess) return fail(REM2D_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            (void)hipGraphDestroy(graph);
            return fail(REM2D_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        }
        if (g_graphs.size() >= 16) { // call lengths come and go: drop the oldest replay
            graph_entry_free(g_graphs.front());
            g_graphs.erase(g_graphs.begin());
        }
        GraphEntry ge;
        ge.key = key; ge.exec = exec; ge.graph = graph;
        for (int g = 0; g < n_groups; ++g)
            for (int i = 0; i < groups[g].n_worlds; ++i) ge.worlds.push_back(groups[g].worlds[i]);
        g_graphs.push_back(ge);
        hit = &g_graphs.back();
    } else { // (a capture advances the worlds' step counters itself, tiles_launch_step; a replay does it here)
        for (int g = 0; g < n_groups; ++g)
            for (int i = 0; i < groups[g].n_worlds; ++i) groups[g].worlds[i]->stepsQueued += n_steps;
    }
    HIP_TRY(hipGraphLaunch(hit->exec, origin));
    return REM2D_OK;
}
extern "C" int rem2d_groups_step(const rem2d_step_group *groups, int32_t n_groups, int32_t n_steps, void *stream, uint32_t flags) {
    return rem2d_groups_step_ex(groups, n_groups, n_steps, (float)(1.0 / 50), 6 * 30, 2 * 30, stream, flags);
}

extern "C" int rem2d_tree_diversity(const double *pos_dev, const int32_t *count_dev, int32_t n_trees, int32_t max_nodes,
                                    int64_t *out_dev, int32_t device, void *stream) {
    if (!pos_dev || !count_dev || !out_dev) return fail(REM2D_E_INVALID, "NULL device pointer");
    if (n_trees < 0 || max_nodes <= 0 || max_nodes > DIV_MAX_NODES)
        return fail(REM2D_E_INVALID, "max_nodes must be 1..64");
    if (n_trees == 0) return REM2D_OK;
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(rem2d_tree_diversity_kernel, dim3((unsigned)n_trees), dim3(DIV_THREADS), 0, (hipStream_t)stream, pos_dev,
                       count_dev, n_trees, max_nodes, (long long *)out_dev);
    HIP_TRY(hipGetLastError());
    return REM2D_OK;
}

#include "rem2d_compile.h"

// ---- self-test of the scalar helpers the solvers are built from (include/rem2d.h rem2d_selftest_scalar) ----
__global__ void rem2d_selftest_scalar_kernel(const float *a, const float *b, const float *c, int n, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = fmin32(a[i], b[i]);
    out[(size_t)n + i] = fmax32(a[i], b[i]);
    out[2 * (size_t)n + i] = fclamp(a[i], b[i], c[i]);
    const Rot q = rot_set(a[i]);
    out[3 * (size_t)n + i] = q.s;
    out[4 * (size_t)n + i] = q.c;
}
extern "C" int rem2d_selftest_scalar(const float *a_dev, const float *b_dev, const float *c_dev, int32_t n, float *out_dev,
                                     int32_t device, void *stream) {
    if (!a_dev || !b_dev || !c_dev || !out_dev) return fail(REM2D_E_INVALID, "NULL device pointer");
    if (n <= 0) return REM2D_OK;
    HIP_TRY(hipSetDevice(device));
    hipLaunchKernelGGL(rem2d_selftest_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a_dev,
                       b_dev, c_dev, n, out_dev);
    HIP_TRY(hipGetLastError());
    return REM2D_OK;
}

extern "C" int rem2d_world_field(const rem2d_world *w, int32_t field, size_t *offset_bytes, size_t *count, int32_t *dtype) {
    if (!w || field < 0 || field >= REM2D_F_COUNT) return fail(REM2D_E_INVALID, "bad field id");
    int dt = 0;
    field_place(w->L, field, offset_bytes, count, &dt);
    if (dtype) *dtype = dt;
    return REM2D_OK;
}

extern "C" int rem2d_world_enable_timing(rem2d_world *w, int32_t on) {
    if (!w) return fail(REM2D_E_INVALID, "world is NULL");
    HIP_TRY(hipSetDevice(w->cfg.device));
    if (on) {
        drain_timing(w);
        const size_t want = on > 1 ? (size_t)on : 4096; // launches that can be timed before the next read-back
        while (w->evPool.size() < want) {
            hipEvent_t a = nullptr, b = nullptr;
            HIP_TRY(hipEventCreate(&a));
            hipError_t e = hipEventCreate(&b);
            if (e != hipSuccess) {
                (void)hipEventDestroy(a);
                return fail(REM2D_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
            }
            w->evPool.emplace_back(a, b);
        }
        while (w->evPoolStep.size() < want) {
            hipEvent_t a = nullptr, b = nullptr;
            HIP_TRY(hipEventCreate(&a));
            hipError_t e = hipEventCreate(&b);
            if (e != hipSuccess) {
                (void)hipEventDestroy(a);
                return fail(REM2D_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
            }
            w->evPoolStep.emplace_back(a, b);
        }
    }
    w->timing = on != 0;
    return REM2D_OK;
}
extern "C" int rem2d_world_kernel_time_ms(rem2d_world *w, double *total_ms, int64_t *launches) {
    if (!w) return fail(REM2D_E_INVALID, "world is NULL");
    HIP_TRY(hipSetDevice(w->cfg.device));
    drain_timing(w);
    if (total_ms) *total_ms = w->accumMs;
    if (launches) *launches = w->launches;
    w->accumMs = 0.0;
    w->launches = 0;
    return REM2D_OK;
}
#if defined(REM2D_TOI_STAMPS) || defined(REM2D_POS_STAMPS)
// diagnostic builds only: the TOI kernel's / the position solver's cycle counters (toiWork[0..16)), then zeroed
extern "C" int rem2d_world_debug_words(rem2d_world *w, int32_t *out16) {
    HIP_TRY(hipSetDevice(w->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out16, w->S.toiWork, 16 * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(w->S.toiWork + 1, 0, 15 * sizeof(int)));
    return REM2D_OK;
}
#endif
extern "C" int rem2d_world_handover_failures(const rem2d_world *w, int64_t *count, int32_t clear) {
    if (!w || !count) return fail(REM2D_E_INVALID, "handover_failures: NULL argument");
    // (pinned host memory the kernels add to at system scope: readable without a device call; what it holds is what the launches
    // that have FINISHED so far reported)
    *count = w->trainFailures ? (int64_t)__atomic_load_n(w->trainFailures, __ATOMIC_RELAXED) : 0;
    if (clear && w->trainFailures) __atomic_store_n(w->trainFailures, 0u, __ATOMIC_RELAXED);
    return REM2D_OK;
}
extern "C" int rem2d_world_step_time_ms(rem2d_world *w, double *total_ms, int64_t *steps) {
    if (!w) return fail(REM2D_E_INVALID, "world is NULL");
    HIP_TRY(hipSetDevice(w->cfg.device));
    drain_timing(w);
    if (total_ms) *total_ms = w->accumMsStep;
    if (steps) *steps = w->launchesStep;
    w->accumMsStep = 0.0;
    w->launchesStep = 0;
    return REM2D_OK;
}
