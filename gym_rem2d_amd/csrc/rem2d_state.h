// rem2d_state.h -- state arena layout, Terrain / State structs and field accessors.
// Part of the single translation unit rem2d.hip (see its header comment); not a stand-alone header.
#ifndef REM2D_STATE_H
#define REM2D_STATE_H

// =====================================================================================
// state arena: field-major with ONE stride per group, so that a kernel address is
//   (scalar group base + field * stride)  +  (32-bit per-lane byte offset shared by all fields)
// i.e. global_load/store with an SGPR base and one VGPR offset -- no per-array address VGPRs.
// =====================================================================================
struct FieldDesc { int group; int index; int dtype; }; // group: 0 lane4, 1 lane8, 2 slot4, 3 env4, 4 env8
enum { G_LANE4 = 0, G_LANE8 = 1, G_SLOT4 = 2, G_ENV4 = 3, G_ENV8 = 4 };
// per-lane 4-byte fields, same order as REM2D_F_PX .. REM2D_F_CCOUNT
enum {
    L_PX = 0, L_PY, L_ANG, L_VX, L_VY, L_W, L_SLEEPT, L_HX, L_HY, L_INVM, L_INVI, L_FATLX, L_FATLY, L_FATUX, L_FATUY,
    L_JAX, L_JAY, L_JBX, L_JBY, L_JTORQUE, L_JLOWER, L_JUPPER, L_JIMPX, L_JIMPY, L_JIMPZ, L_JMOTORIMP, L_JMOTORSPEED,
    L_SHAPE, L_PARENT, L_JROUND, L_AWAKE, L_JLIMIT, L_CCOUNT, L4_COUNT
};
enum { D_CAMP = 0, D_CPHASE, D_CFREQ, D_COFFSET, D_CISTATE, L8_COUNT };
enum { C_EDGE = 0, C_INFO, C_KEY0, C_KEY1, C_N0, C_N1, C_T0, C_T1, S4_COUNT };
enum { E_REWARD = 0, E_DONE, E_EVERDONE, E_FROZEN, E_STEPS, E_INVDT0, E_NEWFIX, E_ERR, E_POSITERS, E_TOIEVENTS, E4_COUNT };
enum { E_WOD = 0, E_FITNESS, E8_COUNT };

static FieldDesc field_desc(int f) {
    FieldDesc d;
    if (f <= REM2D_F_CCOUNT) { d.group = G_LANE4; d.index = f; d.dtype = f >= REM2D_F_SHAPE ? 1 : 0; }
    else if (f <= REM2D_F_CISTATE) { d.group = G_LANE8; d.index = f - REM2D_F_CAMP; d.dtype = 2; }
    else if (f <= REM2D_F_CT1) { d.group = G_SLOT4; d.index = f - REM2D_F_CEDGE; d.dtype = f <= REM2D_F_CKEY1 ? 1 : 0; }
    else if (f == REM2D_F_WOD || f == REM2D_F_FITNESS) { d.group = G_ENV8; d.index = f - REM2D_F_WOD; d.dtype = 2; }
    else { d.group = G_ENV4; d.index = f - REM2D_F_REWARD; d.dtype = (f == REM2D_F_REWARD || f == REM2D_F_INVDT0) ? 0 : 1; }
    return d;
}

struct Layout {
    int Np, Lp, K;
    size_t groupOff[5];
    size_t total;
};
static Layout make_layout(const rem2d_world_cfg *cfg) {
    Layout L;
    int perWave = WAVE / cfg->lanes;
    L.K = cfg->lanes;
    L.Np = (cfg->n_envs + perWave - 1) / perWave * perWave;
    L.Lp = L.Np * cfg->lanes;
    size_t o = 0;
    const size_t sizes[5] = {(size_t)L4_COUNT * L.Lp * 4, (size_t)L8_COUNT * L.Lp * 8, (size_t)S4_COUNT * KC * L.Lp * 4,
                             (size_t)E4_COUNT * L.Np * 4, (size_t)E8_COUNT * L.Np * 8};
    const int order[5] = {G_LANE8, G_ENV8, G_LANE4, G_SLOT4, G_ENV4}; // 8-byte groups first
    for (int k = 0; k < 5; ++k) {
        L.groupOff[order[k]] = o;
        o += sizes[order[k]];
        o = (o + 255) & ~(size_t)255;
    }
    L.total = o;
    return L;
}
static void field_place(const Layout &L, int f, size_t *off, size_t *count, int *dtype) {
    FieldDesc d = field_desc(f);
    size_t n = 0, esz = d.dtype == 2 ? 8 : 4;
    switch (d.group) {
    case G_LANE4: case G_LANE8: n = (size_t)L.Lp; break;
    case G_SLOT4: n = (size_t)L.Lp * KC; break;
    default: n = (size_t)L.Np; break;
    }
    if (off) *off = L.groupOff[d.group] + (size_t)d.index * n * esz;
    if (count) *count = n;
    if (dtype) *dtype = d.dtype;
}

struct Terrain { // static bodies at the origin, in creation (= broadphase proxy) order: hardcore boxes, then edges
    int nEdge, nPoly;
    const float *flx, *fly, *fux, *fuy; // fat AABB of every static proxy            [nPoly + nEdge]
    const float *vx, *vy;               // vertices [4][nPoly + nEdge] (edges use 0 and 1)
    const float *nx, *ny;               // polygon normals [4][nPoly + nEdge]
    int nStatic;
    float x0, invPitch;
    float friction; // b2MixFriction(terrain, module)
};
struct State {
    char *lane4, *lane8, *slot4, *env4, *env8; // group bases inside the caller's arena
    float *scr;                                // handle-owned: manifolds [KT][SCR_WORDS][Lp] + overflow constraints
    int *toiWork;                              // handle-owned: [0] count, [16..16+Lp) bodies that need the full TOI solve
    const int *tiles;                          // handle-owned: tile t of the velocity kernel = creatures [tiles[t], tiles[t+1])
    // handle-owned: creature order of the post kernel (dynamic re-tiling, rem2d_pipeline.h): [0, Np) the order post reads
    // in this step (slot -> creature), [Np, 2 Np) the order post writes for the next step, [2 Np] / [2 Np + 1] how many
    // creatures have been placed from the front / from the back of the latter
    int *order;
    // optional, caller-owned (rem2d_world_set_outputs): reward / done of creature e also go to outReward[outIndex[e]] /
    // outDone[outIndex[e]] -- the population-order arrays of a population that lives in several worlds
    float *outReward;
    unsigned char *outDone;
    const int *outIndex;
    unsigned Lp, Np, nEnvs, flags;
    // the tile table's shape when it is regular (every tile tileCap creatures, tileCap dividing a 64-lane block): the
    // velocity + position launch (rem2d_velpost_kernel) finds the tiles of a block with it; 0 = irregular table
    int nTiles, tileCap;
};
// State::flags = rem2d_world_cfg::flags | internal bits:
// the host installed a creature order (rem2d_world_set_order): slot e of the tile table / of a 64-lane block is creature
// order[e], fixed until the host changes it (REM2D_FLAG_RETILE: the same indirection, re-dealt by the position kernel itself)
#define REM2D_STATE_ORDERED 0x10000u
// accessors (S, gl and env must be in scope where they are used)
#define LF(f) (*(float *)(S.lane4 + (size_t)(f) * ((size_t)S.Lp * 4) + (gl) * 4u))
#define LI(f) (*(int *)(S.lane4 + (size_t)(f) * ((size_t)S.Lp * 4) + (gl) * 4u))
#define LD(f) (*(double *)(S.lane8 + (size_t)(f) * ((size_t)S.Lp * 8) + (gl) * 8u))
#define CF(f, o32) (*(float *)(S.slot4 + (size_t)(f) * ((size_t)S.Lp * (4 * KC)) + (o32) * 4u))
#define CI(f, o32) (*(int *)(S.slot4 + (size_t)(f) * ((size_t)S.Lp * (4 * KC)) + (o32) * 4u))
#define CU(f, o32) (*(unsigned *)(S.slot4 + (size_t)(f) * ((size_t)S.Lp * (4 * KC)) + (o32) * 4u))
#define EF(f) (*(float *)(S.env4 + (size_t)(f) * ((size_t)S.Np * 4) + (env) * 4u))
#define EI(f) (*(int *)(S.env4 + (size_t)(f) * ((size_t)S.Np * 4) + (env) * 4u))
#define ED(f) (*(double *)(S.env8 + (size_t)(f) * ((size_t)S.Np * 8) + (env) * 8u))
// scratch word k of the record that starts at 32-bit word offset base32 (= word0 * Lp + gl)
#define SW(base32, k) (*(float *)((char *)S.scr + ((base32) + (unsigned)(k) * S.Lp) * 4u))

#endif
