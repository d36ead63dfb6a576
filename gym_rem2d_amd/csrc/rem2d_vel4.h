// rem2d_vel4.h -- velocity kernel of the split pipeline, tile form (the default since round 2).
// Part of the single translation unit rem2d.hip; not a stand-alone header.
//
// b2Island::Solve's 180 velocity iterations are one long dependent chain per creature; what a wavefront can
// change is how many creatures ride on one chain.  rem2d_step_multi_kernel keeps one body per lane: a joint slot runs
// with 1/period of its lanes and a contact slot with ~5 % of them, so the population needs ~4 rounds of
// wavefronts, each paying the whole chain.  Here a wavefront owns a TILE: a run of consecutive creatures with
// at most 256 bodies whose joints number at most 64 in every phase of the modulo schedule (the host cuts the
// tiles, rem2d_plan_tiles / rem2d_world_set_tiles).  Lanes are constraints:
//   * joints: lane r of register set s holds the r-th joint of phase s (joint round mod period) -- in the tick
//     of phase s all of them fire together, so a joint slot runs with up to 64 active lanes;
//   * contacts: every touching manifold of the tile gets a lane of its own (CSETS register sets of 64).  A
//     body's manifolds must run in list order between its last joint of one iteration and its first joint of the
//     next (its WINDOW of ticks); manifolds of different bodies commute.  The kernel picks, per tile, the phase
//     that lies in the window of most touching bodies and puts all their manifolds there (sub-slot t = t-th
//     manifold), so that one iteration runs about max-manifolds-per-body contact sub-slots instead of one
//     batch per phase;
//   * body velocities (and 1/I) live in an LDS mailbox, one 16-byte record per body; hand-offs are wave-local
//     (lds_sync = s_waitcnt, no s_barrier).
// The whole 65 536-creature population is ~3000 tiles.  Arithmetic and ordering are those of the other forms
// (compiler.pipeline_schedule proves that any two operations sharing a body keep Box2D's sequential order for
// any period >= the creature's own), so the result is bit-identical to rem2d_step_multi_kernel and to the oracle.
#ifndef REM2D_VEL4_H
#define REM2D_VEL4_H

// Tile shapes (template parameters of the kernel; the host picks one per launch, rem2d.hip):
//   SETS    joint register sets per lane.  A joint of schedule phase ph lives in set ph mod SETS; the joints of the
//           phases that share a set take different lanes, so a tile holds at most 64 joints per set.
//   PASSES  64-body passes of the body role: a tile holds at most 64 * PASSES bodies.
//   CSETS   contact register sets: 64 * CSETS manifolds per tile in registers, the rest through scratch.
//   WPS     waves per SIMD the register allocation targets.
// More bodies per tile = fewer wavefronts and fuller joint slots; fewer registers = more resident wavefronts to hide
// the dependent-instruction latency (~10 cycles per instruction for a lone wavefront, measured with s_memtime).
#ifndef V4_PLUS1_GAIN
#define V4_PLUS1_GAIN 2    // contact sub-slots per iteration a tile must save to take the period P + 1 (one more joint slot per iteration)
#endif
// Largest schedule period the tick loop is unrolled for.  A module of the reference has three connection sites and the joint to its
// parent (simple_module.py:21-25, circular_module.py:23-27): at most four joints on a body, period <= 4 -- and a tile of period <= 3
// may still take one more (V4_PLUS1_GAIN).  Round 6: unrolled for 4 instead of 5 the loop is a fifth shorter and needs fewer scalar
// registers: +1.8 % on config 3, +3.1 % on config 4 (profiles/r06_experiments.txt 11; a period-4 tile loses its option on period 5,
// which measured as nothing).  The wide build -- the fallback for bodies outside the reference's domain -- keeps 5; a creature
// beyond a build's V4_PHASES is flagged REM2D_ERR_SOLVER_OVERFLOW like one beyond its contact slots.
#ifndef V4_PHASES
#ifdef REM2D_WIDE
#define V4_PHASES 5
#else
#define V4_PHASES 4
#endif
#endif
#define V4_MAX_BODIES 256  // bodies per tile in the widest shape (LDS mailbox size)
#define V4_MAX_PASSES (V4_MAX_BODIES / WAVE)

struct Vel4Args { int velIters; float dt; int prio, prioT1, prioT2; /* s_setprio for the tiles expected to be slow (rem2d.hip: prio_mode) */ int dbg; /* diagnostic builds (-DREM2D_V4_PROBES) only: 1 skip contact sub-slots, 2 skip joint slots, 8 s_memtime split, 16 start / end time of every wavefront */ };
#ifdef REM2D_V4_PROBES
#define V4_DBG(A) ((A).dbg)
#else
#define V4_DBG(A) 0

#endif

struct __attribute__((aligned(16))) V4Vel { float x, y, w, invI; };
template <int SETS, int PASSES> struct Vel4Shared {
    V4Vel vel[PASSES * WAVE];
    unsigned short jmap[SETS][WAVE];        // (set, lane) -> tile-local body id of the joint's child, 0xffff = none
    unsigned int cmap[PASSES * WAVE * KT];  // contact rank -> V4_C* key (body, sub-slot, manifold index, tick offset)
    int firstR[PASSES * WAVE], lastR[PASSES * WAVE]; // first / last joint round of every body (own joint and children's)
    // tile-local body id -> lane index in the state arena.  Consecutive creatures unless the world deals its creatures anew
    // every step (REM2D_FLAG_RETILE): then slot e of the tile table is creature order[e], for this kernel as for post
    unsigned glmap[PASSES * WAVE];
};

// one joint: what the 180 iterations read (constants) and write (accumulated impulses).  The inverse inertias come
// with the velocity records.  The limit-only terms of the 3x3 mass matrix (third column ezx, ezy; cross(ey, ez)) stay in
// registers too: some joint of a tile is at its limit in four slots out of five, so the wave recomputed them (19 of the
// ~110 instructions of a joint slot) in nearly every slot.
struct JointT {
    V2 rA, rB;
    float mA, mB;
    float exx, eyx, eyy, motorMass, det33, det22;
    float ezx, ezy, cyzx, cyzy, cyzz;
    float impX, impY, impZ, motorImp, motorSpeed, maxMotorImpulse;
    int key; // jA | jB << 8 | jround << 16 | limitState << 24 | phase << 26 | valid << 31
};
#define V4_JA(k) ((k) & 0xff)
#define V4_JB(k) (((k) >> 8) & 0xff)
#define V4_JROUND(k) (((k) >> 16) & 0xff)
#define V4_LIMIT(k) (((k) >> 24) & 0x3)
#define V4_JPHASE(k) (((k) >> 26) & 0x7)
#define V4_VALID(k) ((k) < 0)

DEV void v4_joint_load(const State &S, const unsigned *glmap, int K, int child, float h, int P, int rotation, JointT &J) {
    const unsigned Lp = S.Lp;
    const unsigned gl = glmap[child];
    const int jA = (child & ~(K - 1)) + LI(L_PARENT);
    const int jround = (LI(L_JROUND) & 0xff) + rotation; // (its creature's schedule runs `rotation` ticks late)
    const unsigned jb = (unsigned)SCR_JREC_BASE * Lp + gl;
    J.rA = mk(SW(jb, 0), SW(jb, 1));
    J.rB = mk(SW(jb, 2), SW(jb, 3));
    J.mB = LF(L_INVM);
    const float iB = LF(L_INVI);
    J.impX = LF(L_JIMPX); J.impY = LF(L_JIMPY); J.impZ = LF(L_JIMPZ); J.motorImp = LF(L_JMOTORIMP);
    J.motorSpeed = LF(L_JMOTORSPEED);
    const int limitState = LI(L_JLIMIT);
    J.maxMotorImpulse = h * LF(L_JTORQUE);
    float iA;
    {
        const unsigned gl = glmap[jA];
        J.mA = LF(L_INVM); iA = LF(L_INVI);
    }
    J.key = jA | (child << 8) | (jround << 16) | (limitState << 24) | ((jround % P) << 26) | (int)0x80000000;
    const float mA = J.mA, mB = J.mB;
    const V2 rA = J.rA, rB = J.rB;
    // effective-mass terms of b2RevoluteJoint::InitVelocityConstraints (same expressions as rem2d_step_multi_kernel)
    J.exx = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
    J.eyx = -rA.y * rA.x * iA - rB.y * rB.x * iB;
    const float ezx = -rA.y * iA - rB.y * iB;
    J.eyy = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
    const float ezy = rA.x * iA + rB.x * iB;
    const float ezz = iA + iB;
    J.motorMass = iA + iB;
    if (J.motorMass > 0.0f) J.motorMass = 1.0f / J.motorMass;
    const float cyzx = J.eyy * ezz - ezy * ezy;
    const float cyzy = ezy * ezx - J.eyx * ezz;
    const float cyzz = J.eyx * ezy - J.eyy * ezx;
    J.ezx = ezx; J.ezy = ezy; J.cyzx = cyzx; J.cyzy = cyzy; J.cyzz = cyzz;
    J.det33 = J.exx * cyzx + J.eyx * cyzy + ezx * cyzz;
    if (J.det33 != 0.0f) J.det33 = 1.0f / J.det33;
    J.det22 = J.exx * J.eyy - J.eyx * J.eyx;
    if (J.det22 != 0.0f) J.det22 = 1.0f / J.det22;
}

// b2RevoluteJoint::SolveVelocityConstraints (motor, then limit 3x3 / point 2x2) on the mailbox records of the
// joint's two bodies; same expression sequence as the joint slot of rem2d_step_multi_kernel
template <typename SH> DEV void v4_joint_slot(JointT &j, SH &sh) {
    const int a = V4_JA(j.key), b = V4_JB(j.key), limitState = V4_LIMIT(j.key);
    V4Vel ra = sh.vel[a], rb = sh.vel[b];
    V2 vA = mk(ra.x, ra.y), vB = mk(rb.x, rb.y);
    float wA = ra.w, wB = rb.w;
    const float iA = ra.invI, iB = rb.invI;
    if (limitState != LIM_EQUAL) {
        float Cdot = wB - wA - j.motorSpeed;
        float impulse = -j.motorMass * Cdot;
        float oldImpulse = j.motorImp;
        j.motorImp = fclamp(oldImpulse + impulse, -j.maxMotorImpulse, j.maxMotorImpulse);
        impulse = j.motorImp - oldImpulse;
        wA -= iA * impulse;
        wB += iB * impulse;
    }
    // (the relative velocity of the anchors and the application of the impulse are common to the limit (3 x 3) and the point
    // (2 x 2) solve: a wavefront usually holds joints of both kinds and runs both branches one after the other, so what
    // they share is computed once.  For a joint without an active limit the angular impulse is a literal 0 added to the
    // cross products -- x + 0 is x, bar the sign of a zero.)
    const V2 Cdot1 = vsub(vsub(vadd(vB, vcross_sv(wB, j.rB)), vA), vcross_sv(wA, j.rA));
    float ix, iy, iz = 0.0f;
    if (limitState != LIM_INACTIVE) {
        // ex = (exx, eyx, ezx), ey = (eyx, eyy, ezy), ez = (ezx, ezy, ezz) of m_mass; cyz = cross(ey, ez)
        const float ezx = j.ezx, ezy = j.ezy, ezz = iA + iB;
        const float cyzx = j.cyzx, cyzy = j.cyzy, cyzz = j.cyzz;
        float Cdot2 = wB - wA;
        float bx = Cdot1.x, by = Cdot1.y, bz = Cdot2;
        float sx = j.det33 * (bx * cyzx + by * cyzy + bz * cyzz);
        float cbx = by * ezz - bz * ezy, cby = bz * ezx - bx * ezz, cbz = bx * ezy - by * ezx;
        float sy = j.det33 * (j.exx * cbx + j.eyx * cby + ezx * cbz);
        float ebx = j.eyy * bz - ezy * by, eby = ezy * bx - j.eyx * bz, ebz = j.eyx * by - j.eyy * bx;
        float sz = j.det33 * (j.exx * ebx + j.eyx * eby + ezx * ebz);
        ix = -sx; iy = -sy; iz = -sz;
        if (limitState == LIM_EQUAL) {
            j.impX += ix; j.impY += iy; j.impZ += iz;
        } else {
            float newImpulse = j.impZ + iz;
            // (at the lower limit the impulse may not go negative, at the upper not positive: one comparison of the
            // impulse or its negation -- x > 0 <=> -x < 0, also for the zeros -- instead of two compares and a select)
            bool reduce = (limitState == LIM_AT_LOWER ? newImpulse : -newImpulse) < 0.0f;
            if (reduce) {
                V2 rhs = vadd(vneg(Cdot1), vscale(j.impZ, mk(ezx, ezy)));
                float rx = j.det22 * (j.eyy * rhs.x - j.eyx * rhs.y);
                float ry = j.det22 * (j.exx * rhs.y - j.eyx * rhs.x);
                ix = rx; iy = ry; iz = -j.impZ;
                j.impX += rx; j.impY += ry; j.impZ = 0.0f;
            } else {
                j.impX += ix; j.impY += iy; j.impZ += iz;
            }
        }
    } else {
        V2 bb = vneg(Cdot1);
        ix = j.det22 * (j.eyy * bb.x - j.eyx * bb.y);
        iy = j.det22 * (j.exx * bb.y - j.eyx * bb.x);
        j.impX += ix; j.impY += iy;
    }
    {
        V2 P = mk(ix, iy);
        vA = vsub(vA, vscale(j.mA, P));
        wA -= iA * (vcross(j.rA, P) + iz);
        vB = vadd(vB, vscale(j.mB, P));
        wB += iB * (vcross(j.rB, P) + iz);
    }
    ra.x = vA.x; ra.y = vA.y; ra.w = wA;
    rb.x = vB.x; rb.y = vB.y; rb.w = wB;
    sh.vel[a] = ra;
    sh.vel[b] = rb;
}

// one contact lane: the manifold's constraint, its body's inverse mass and its place in the schedule
struct ContactT {
    ContactC c;
    QuadRole r; // CPAIR: this lane's component of the pair solve (rem2d_solver.h)
    float mB;
    int key; // body | sub-slot << 8 | manifold index << 12 | first tick << 16 | (first tick mod P) << 24 | valid << 31
};
#define V4_CBODY(k) ((k) & 0xff)
#define V4_CSUB(k) (((k) >> 8) & 0xf)
#define V4_CT(k) (((k) >> 12) & 0xf)
static_assert(KT <= 16, "manifold index: 4 bits");
#define V4_COFF(k) (((k) >> 16) & 0xff)
#define V4_CPHASE(k) (((k) >> 24) & 0x7)

// the contact sub-slots of the tick `tick` (phase ph): every manifold scheduled here, in the order of its body's list
template <int CSETS, bool CPAIR, typename SH>
DEV void v4_contact_subslots(const State &S, ContactT (&C)[CSETS], SH &sh, int lane, int NC, bool spill,
                             bool pair, int nsub, int ph, int tick, int span, float mu) {
    const unsigned Lp = S.Lp;
    for (int t = 0; t < nsub; ++t) {
#pragma unroll
        for (int cs = 0; cs < CSETS; ++cs) {
            // off = ph (mod P) and off <= tick < off + iters P: the manifold runs in this tick, in sub-slot `sub`
            if (V4_VALID(C[cs].key) && V4_CSUB(C[cs].key) == t && V4_CPHASE(C[cs].key) == ph &&
                (unsigned)(tick - V4_COFF(C[cs].key)) < (unsigned)span) {
                const int b = V4_CBODY(C[cs].key);
                V4Vel v = sh.vel[b];
                if (!CPAIR || !pair) {
                    contact_solve(C[cs].c, C[cs].mB, v.invI, mu, v.x, v.y, v.w);
                    sh.vel[b] = v;
                } else { // two lanes per manifold: this one carries component (lane & 1) of the body's velocity
                    float vq = C[cs].r.isY ? v.y : v.x;
                    contact_solve_pair(C[cs].c, C[cs].r, C[cs].mB, v.invI, mu, vq, v.w);
                    if (!C[cs].r.isY) sh.vel[b].x = vq;
                    else { sh.vel[b].y = vq; sh.vel[b].w = v.w; }
                }
            }
        }
        if (spill) {
            for (int ci = CSETS * WAVE + lane; ci < NC; ci += WAVE) { // (a tile that spills is not in pair mode)
                const int e = (int)sh.cmap[ci];
                const int b = V4_CBODY(e);
                if (V4_CSUB(e) != t || V4_CPHASE(e) != ph || !((unsigned)(tick - V4_COFF(e)) < (unsigned)span)) continue;
                const unsigned gl = sh.glmap[b];
                const unsigned cb = (unsigned)(SCR_CC_BASE + V4_CT(e) * CC_WORDS) * Lp + gl;
                ContactC c;
                cc_load(S, cb, c);
                V4Vel v = sh.vel[b];
                contact_solve(c, LF(L_INVM), v.invI, mu, v.x, v.y, v.w);
                sh.vel[b] = v;
                SW(cb, 10) = c.n0; SW(cb, 11) = c.n1; SW(cb, 12) = c.t0; SW(cb, 13) = c.t1;
            }
        }
        lds_sync();
    }
}

template <int SETS, int PASSES, int CSETS, bool CPAIR, bool FLEXP = true>
DEV void vel4_body(const State &S, const float friction, const Vel4Args &A, unsigned tile, int K, Vel4Shared<SETS, PASSES> &sh) {
    const int lane = threadIdx.x;
    const unsigned Lp = S.Lp;
    // FLEX (up to three joint register sets): a joint whose phase's own set is full takes a lane of the other set, and the tick
    // loop looks at every set in every tick -- so a tile may rotate its creatures' schedules and take the period P + 1 (both
    // move joints between phases, hence between sets) without ever breaking the 64-joints-per-set rule the host planned for.
    // The four-set shape keeps the host's static phase -> set map (one inlined joint slot per phase instead of four).
    constexpr bool FLEX = FLEXP && SETS <= 3;
    const int c0 = S.tiles[tile], c1 = S.tiles[tile + 1];
    const int NB = (c1 - c0) * K;               // bodies of this tile (<= V4_MAX_BODIES, checked by the host)
    const unsigned long long rEntry = (V4_DBG(A) & (16 | 64)) ? __builtin_amdgcn_s_memrealtime() : 0; // 100 MHz, chip-wide
    // A creature order (REM2D_FLAG_RETILE / rem2d_world_set_order / the `rebalance` option) moves creatures between tiles.  The
    // flexible shapes take any composition (a creature has fewer joints than lanes); a tile of a static shape was planned by the
    // host for the creatures it holds in ARENA order (<= 64 joints per phase class), so those shapes keep the arena order here
    // and only the position blocks follow the creature order (the two kernels meet in the arena, per creature).
    const bool retile = FLEX && (S.flags & (REM2D_FLAG_RETILE | REM2D_STATE_ORDERED)) != 0;
    const int iters = A.velIters;
    const float h = A.dt, mu = friction;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (WAVE - lane));
    if (S.flags & REM2D_FLAG_SKIP_FROZEN) { // creatures pre left alone (it marks them 0x200): nothing to solve, nothing handed over
        // (pre's own criterion -- its 64-lane wavefront in ARENA order held finished creatures only -- read back per creature:
        // under a creature order a tile may hold finished creatures of wavefronts that pre did step)
        bool allSkipped = true;
        for (int e = c0 + lane; e < c1; e += WAVE) {
            const unsigned env = retile ? (unsigned)S.order[e] : (unsigned)e;
            const int m0 = __float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + env * (unsigned)K, 0));
            allSkipped = allSkipped && (m0 & 0x200) != 0;
        }
        if (__all(allSkipped ? 1 : 0)) return;
    }

#pragma unroll
    for (int s = 0; s < SETS; ++s) sh.jmap[s][lane] = 0xffff;
    // ---------------- body role (<= 4 passes of 64 bodies): publish velocities, deal joints and manifolds to lanes ----------------
    int sched[PASSES], misc[PASSES], parent[PASSES];
    int P = 1;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int bl = p * WAVE + lane;
        // (a lane beyond the tile's bodies maps to its own slot of the next creatures: never dereferenced, bl < NB guards)
        const unsigned ce = (unsigned)c0 + (unsigned)(bl / K);
        const unsigned gl = ((retile && bl < NB) ? (unsigned)S.order[ce] : ce) * (unsigned)K + (unsigned)(bl & (K - 1));
        sh.glmap[bl] = gl;
        sched[p] = 0; misc[p] = 0; parent[p] = -1;
        sh.firstR[bl] = 0x7fffffff;
        sh.lastR[bl] = -1;
        if (bl < NB) {
            misc[p] = __float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + gl, 0));
            sched[p] = LI(L_JROUND);
            parent[p] = LI(L_PARENT);
            V4Vel v;
            v.x = LF(L_VX); v.y = LF(L_VY); v.w = LF(L_W); v.invI = LF(L_INVI);
            sh.vel[bl] = v;
        }
        P = max(P, (sched[p] >> 16) & 0xff); // every creature of the tile, awake or not: the host cut the tile with this period
    }
    P = wave_max(P);
    lds_sync();
    // ---- every body's contact window.  Body b's manifolds of iteration i may run anywhere after its last joint of
    // that iteration (round last[b]) and before its first joint of the next one (round first[b] + P): ticks
    // last[b] .. first[b] + P - 1 (+ i P).  The host's contact slot offC[b] is one tick of that window.
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int bl = p * WAVE + lane;
        if ((misc[p] & 0x100) != 0 && parent[p] >= 0) {
            const int jr = sched[p] & 0xff, pl = (bl & ~(K - 1)) + parent[p];
            atomicMin(&sh.firstR[bl], jr); atomicMax(&sh.lastR[bl], jr);
            atomicMin(&sh.firstR[pl], jr); atomicMax(&sh.lastR[pl], jr);
        }
    }
    lds_sync();
    // ---- contact phases.  A contact sub-slot costs the wavefront the same whether one manifold or 64 run in it, and a tick
    // costs as many sub-slots as the most manifolds any body runs in it.  Two steps keep that sum small:
    // (1) per creature, its bodies are gathered in as few phases as possible: greedily, the phase inside the window of the
    //     most still unplaced touching bodies of the creature (ties: the later phase, the host's own preference) takes
    //     all of those, until every body is placed (<= P rounds).  A body runs all its manifolds in its tick, one per
    //     sub-slot.
    // (2) creatures are independent, so each one may run its whole schedule -- joint rounds and contact ticks alike --
    //     any number of ticks late: every creature is rotated so that its heaviest contact phase (the most manifolds on
    //     one body) falls on phase 0 of the tile.  Only with one joint register set: the host packed the others by phase.
    // (3) the period itself is a choice: the schedule is valid for any period >= the creatures' own (compiler.pipeline_schedule),
    //     and one more tick per iteration widens every body's window by one, so that bodies whose windows did not meet can
    //     share a contact tick.  A tile whose contact sub-slots outweigh a joint slot takes the longer period: the slowest
    //     tiles of a launch are such tiles (period 3-4 with 5-6 sub-slots per iteration, profiles/archive/r03_slow_tiles.txt).
    //     Again only with one joint register set.
    const unsigned long long groupLanes = (K >= WAVE ? ~0ull : ((1ull << K) - 1ull)) << (lane & ~(K - 1) & (WAVE - 1));
    int offB[PASSES], delta[PASSES];
    // the plan for period Pc: every body's contact tick offB (iteration 0) and its creature's rotation delta; returns the
    // contact sub-slots per iteration of the whole tile under that plan
    auto plan = [&](const int Pc) -> int {
        int wlo[PASSES], wlen[PASSES];
        bool placed[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int bl = p * WAVE + lane;
            const bool solve = (misc[p] & 0x100) != 0;
            const int nT = solve ? (misc[p] & 0xff) : 0;
            const int lastB = sh.lastR[bl];
            wlo[p] = lastB >= 0 ? lastB : 0;                          // first tick of the window (iteration 0)
            wlen[p] = lastB >= 0 ? sh.firstR[bl] + Pc - lastB : Pc;   // its length in ticks, 1 .. Pc
            offB[p] = wlo[p];
            placed[p] = nT == 0;
            delta[p] = 0;
        }
        int subs[V4_PHASES];
#pragma unroll
        for (int s = 0; s < V4_PHASES; ++s) subs[s] = 0;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int wl = wlo[p] % Pc;
            for (int round = 0; round < Pc && round < V4_PHASES; ++round) { // per creature (the lanes of groupLanes)
                int cover[V4_PHASES];
#pragma unroll
                for (int s = 0; s < V4_PHASES; ++s) {
                    int d = s - wl;
                    d = d < 0 ? d + Pc : d;
                    cover[s] = __popcll(__ballot(!placed[p] && s < Pc && d < wlen[p]) & groupLanes);
                }
                int cstar = 0;
#pragma unroll
                for (int s = 1; s < V4_PHASES; ++s) cstar = (s < Pc && cover[s] >= cover[cstar]) ? s : cstar;
                int dstar = cstar - wl;
                dstar = dstar < 0 ? dstar + Pc : dstar;
                if (!placed[p] && cover[cstar] > 0 && dstar < wlen[p]) {
                    offB[p] = wlo[p] + dstar;
                    placed[p] = true;
                }
            }
            const bool solve = (misc[p] & 0x100) != 0;
            const int nT = solve ? (misc[p] & 0xff) : 0;
            if (FLEX) {
                const int cph = offB[p] % Pc;
                int heavyPhase = 0, heavy = 0; // the creature's phase with the most manifolds on one body (ties: the first)
#pragma unroll
                for (int s = 0; s < V4_PHASES; ++s) {
                    int most = 0;
#pragma unroll
                    for (int n = 1; n <= KT; ++n) most = (__ballot(nT >= n && cph == s) & groupLanes) ? n : most;
                    if (s < Pc && most > heavy) { heavy = most; heavyPhase = s; }
                }
                delta[p] = heavy > 0 ? (Pc - heavyPhase) % Pc : 0;
            }
            const int cphr = (offB[p] + delta[p]) % Pc;
#pragma unroll
            for (int s = 0; s < V4_PHASES; ++s) {
#pragma unroll
                for (int n = 1; n <= KT; ++n)
                    if (__ballot(nT >= n && cphr == s)) subs[s] = max(subs[s], n);
            }
        }
        int total = 0;
#pragma unroll
        for (int s = 0; s < V4_PHASES; ++s) total += subs[s];
        return total;
    };
    {
        const int subsP = plan(P);
        // a joint slot costs ~0.7 of a HEAVY contact sub-slot (profiles/archive/r03_slow_tiles.txt: 850-1000 vs 1150-1400 cycles) but
        // about as much as a light one (800): only a gain of two sub-slots is worth a tick.  (Longer periods still -- up to
        // the creatures' round count, where every window contains one common tick -- need the tick loop rolled instead of
        // unrolled per phase; measured: the rolled loop costs 4 % per tick, more than the few tiles it helps gain.)
        if (FLEX && P < V4_PHASES && subsP >= 3) {
            int offP[PASSES], deltaP[PASSES]; // (the plan for P, kept: trying P + 1 overwrites it, and evaluating a plan is ~1 000 instructions)
#pragma unroll
            for (int p = 0; p < PASSES; ++p) { offP[p] = offB[p]; deltaP[p] = delta[p]; }
            const int subsQ = plan(P + 1);
            if (subsQ + V4_PLUS1_GAIN <= subsP) P = P + 1;
            else {
#pragma unroll
                for (int p = 0; p < PASSES; ++p) { offB[p] = offP[p]; delta[p] = deltaP[p]; } // (back to the plan for P)
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) sh.firstR[p * WAVE + lane] = delta[p]; // (the windows are done with: the joint role reads its creature's rotation here)
    lds_sync();

    int NC = 0, lastTick = -1, maxRound = -1, err = 0, maxT = 0;
    int jcount[SETS]; // lanes handed out in every register set (the phases ph = s mod SETS share set s)
#pragma unroll
    for (int s = 0; s < SETS; ++s) jcount[s] = 0;
    int subMax[V4_PHASES]; // contact sub-slots per phase = most manifolds any body runs in a tick of that phase
#pragma unroll
    for (int s = 0; s < V4_PHASES; ++s) subMax[s] = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int bl = p * WAVE + lane;
        const bool solve = (misc[p] & 0x100) != 0;
        const bool hasJ = solve && parent[p] >= 0;
        const int nT = solve ? (misc[p] & 0xff) : 0;
        const int jr = (sched[p] & 0xff) + delta[p];
        const int phase = jr % P;
        if (!FLEX) {
#pragma unroll
            for (int s = 0; s < SETS; ++s) {
                const unsigned long long m = __ballot(hasJ && phase % SETS == s);
                if (hasJ && phase % SETS == s) {
                    const int rank = jcount[s] + __popcll(m & below);
                    if (rank < WAVE) sh.jmap[s][rank] = (unsigned short)bl;
                    else err = REM2D_ERR_SOLVER_OVERFLOW; // the tile breaks the host's rule (<= 64 joints per register set)
                }
                jcount[s] += __popcll(m);
            }
        } else {
            bool placedJ = !hasJ;
#pragma unroll
            for (int k = 0; k < SETS; ++k) {       // k-th choice of set: the phase's own first, then the next one with room
#pragma unroll
                for (int s = 0; s < SETS; ++s) {
                    const bool want = !placedJ && (phase + k) % SETS == s;
                    const unsigned long long m = __ballot(want);
                    if (want) {
                        const int rank = jcount[s] + __popcll(m & below);
                        if (rank < WAVE) { sh.jmap[s][rank] = (unsigned short)bl; placedJ = true; }
                    }
                    jcount[s] = min(WAVE, jcount[s] + __popcll(m));
                }
            }
            if (!placedJ) err = REM2D_ERR_SOLVER_OVERFLOW; // more joints than the tile's register sets hold in all (host rule broken)
        }
        if (hasJ && phase >= V4_PHASES) err = REM2D_ERR_SOLVER_OVERFLOW;
        const int off = offB[p] + delta[p];
        const int cph = off % P;
        if (nT > 0) {
#pragma unroll
            for (int s = 0; s < V4_PHASES; ++s)
                if (cph == s) subMax[s] = max(subMax[s], nT);
            if (cph >= V4_PHASES) err = REM2D_ERR_SOLVER_OVERFLOW;
            if (iters > 0) lastTick = max(lastTick, off + (iters - 1) * P);
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) { // manifold t of every body: ranks in (pass, t, lane) order, sub-slot t
            const unsigned long long cm = __ballot(nT > t);
            if (nT > t) sh.cmap[NC + __popcll(cm & below)] = (unsigned)(bl | (t << 8) | (t << 12) | (off << 16) | (cph << 24));
            NC += __popcll(cm);
        }
        maxT = max(maxT, nT);
        if (iters > 0 && hasJ) lastTick = max(lastTick, jr + (iters - 1) * P);
        if (hasJ) maxRound = max(maxRound, jr);
    }
    const int nTicks = wave_max(lastTick) + 1;
    const int nRounds = wave_max(maxRound) + 1;
    maxT = wave_max(maxT);
#pragma unroll
    for (int s = 0; s < V4_PHASES; ++s) subMax[s] = (V4_DBG(A) & 1) ? 0 : wave_max(subMax[s]);
    lds_sync();
    // A launch lasts as long as its slowest tile, and a tile that shares its SIMD with three other wavefronts runs ~1.5x
    // slower than alone: the tiles with the most slots per iteration (7 : 10 = cost of a joint slot to a contact sub-slot)
    // get issue priority on their SIMD.  A scheduling hint only.
    if (A.prio & 1) {
        int cost = 7 * P;
#pragma unroll
        for (int s = 0; s < V4_PHASES; ++s) cost += 10 * subMax[s];
        if (cost >= A.prioT2) __builtin_amdgcn_s_setprio(3);
        else if (cost >= A.prioT1) __builtin_amdgcn_s_setprio(1);
    }

    // ---------------- joint role: one joint per phase and lane ----------------
    JointT J[SETS];
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
        J[s].key = 0;
        const int child = sh.jmap[s][lane];
        if (child != 0xffff) v4_joint_load(S, sh.glmap, K, child, h, P, sh.firstR[child], J[s]);
    }
    // ---------------- contact role: manifold `lane + 64 cs` of the tile; beyond CSETS * 64 through scratch ----------------
    ContactT C[CSETS];
    const bool pair = CPAIR && NC <= CSETS * (WAVE / 2); // wave-uniform
#pragma unroll
    for (int cs = 0; cs < CSETS; ++cs) {
        C[cs].key = 0;
        C[cs].c.count = 0;
        C[cs].mB = 0.0f;
        C[cs].r.tq = C[cs].r.nq = C[cs].r.r0c = C[cs].r.r1c = 0.0f;
        C[cs].r.isY = (lane & 1) != 0;
        C[cs].r.pt1 = false;
        // pair mode (CPAIR, and the tile's manifolds fit half the lanes -- a 64-body tile of 8- or 16-lane creatures has ~15):
        // manifold ci on the lane pair 2 ci, 2 ci + 1 of its register set, both lanes with the whole constraint and one
        // component each of the arithmetic (contact_solve_pair)
        const int ci = pair ? cs * (WAVE / 2) + (lane >> 1) : cs * WAVE + lane;
        if (ci < NC) {
            const int e = (int)sh.cmap[ci];
            const int b = V4_CBODY(e), t = V4_CT(e);
            const unsigned gl = sh.glmap[b];
            C[cs].key = e | (int)0x80000000;
            C[cs].mB = LF(L_INVM);
            cc_load(S, (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl, C[cs].c);
            C[cs].r = quad_role(C[cs].c, lane);
        }
    }
    const int CCAP = CSETS * WAVE; // manifolds in registers (classic mode; a tile in pair mode has at most half as many)
    const bool spill = NC > CCAP; // wave-uniform
    // ---------------- warm start: contacts (per body in list order), then joints in island rounds ----------------
    for (int t = 0; t < maxT; ++t) {
#pragma unroll
        for (int cs = 0; cs < CSETS; ++cs) {
            if (V4_VALID(C[cs].key) && V4_CT(C[cs].key) == t) {
                const int b = V4_CBODY(C[cs].key);
                V4Vel v = sh.vel[b];
                contact_warm_start(C[cs].c, C[cs].mB, v.invI, v.x, v.y, v.w);
                sh.vel[b] = v;
            }
        }
        if (spill) {
            for (int ci = CCAP + lane; ci < NC; ci += WAVE) {
                const int e = (int)sh.cmap[ci];
                const int b = V4_CBODY(e);
                if (V4_CT(e) != t) continue;
                const unsigned gl = sh.glmap[b];
                ContactC c;
                cc_load(S, (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl, c);
                V4Vel v = sh.vel[b];
                contact_warm_start(c, LF(L_INVM), v.invI, v.x, v.y, v.w);
                sh.vel[b] = v;
            }
        }
        lds_sync();
    }
    for (int r = 0; r < nRounds; ++r) {
#pragma unroll
        for (int s = 0; s < SETS; ++s) {
            if (V4_VALID(J[s].key) && V4_JROUND(J[s].key) == r) {
                const int a = V4_JA(J[s].key), b = V4_JB(J[s].key);
                V4Vel ra = sh.vel[a], rb = sh.vel[b];
                V2 vA = mk(ra.x, ra.y), vB = mk(rb.x, rb.y);
                float wA = ra.w, wB = rb.w;
                V2 Pw = mk(J[s].impX, J[s].impY);
                vA = vsub(vA, vscale(J[s].mA, Pw));
                wA -= ra.invI * (vcross(J[s].rA, Pw) + J[s].motorImp + J[s].impZ);
                vB = vadd(vB, vscale(J[s].mB, Pw));
                wB += rb.invI * (vcross(J[s].rB, Pw) + J[s].motorImp + J[s].impZ);
                ra.x = vA.x; ra.y = vA.y; ra.w = wA;
                rb.x = vB.x; rb.y = vB.y; rb.w = wB;
                sh.vel[a] = ra;
                sh.vel[b] = rb;
            }
        }
        lds_sync();
    }
    // ---------------- velocity iterations: ticks in groups of one period, phase = position in the group ----------------
    {
        const int span = iters * P; // joint k fires at ticks jround + i P, i < iters, i.e. while tick - jround < span
        const bool joints = !(V4_DBG(A) & 2);
        const bool stamp = (V4_DBG(A) & (8 | 32)) != 0; // diagnostic: s_memtime split of the loop (tools/vel4_probe.py); 32: four packed words
        unsigned long long tJ = 0, tC = 0, t0 = 0, t1 = 0, nSub = 0;
        const unsigned long long tStart = stamp ? __builtin_amdgcn_s_memtime() : 0;
        const unsigned long long rStart = stamp ? __builtin_amdgcn_s_memrealtime() : 0; // constant 100 MHz
        for (int base = 0; base < nTicks; base += P) {
#pragma unroll
            for (int s = 0; s < V4_PHASES; ++s) {
                if (s < P) {
                    const int tick = base + s;
                    if (stamp) t0 = __builtin_amdgcn_s_memtime();
                    // the joints of phase s: register set s mod SETS, the lanes whose joint has that phase (FLEX: then the other
                    // set's lanes of that phase -- joints that did not fit their own set; two joints of one tick never share a body)
                    {
                        JointT &Js = J[s % SETS];
                        if (joints && V4_VALID(Js.key) && (SETS == V4_PHASES || V4_JPHASE(Js.key) == s) &&
                            (unsigned)(tick - V4_JROUND(Js.key)) < (unsigned)span) v4_joint_slot(Js, sh);
                    }
                    if (FLEX && SETS > 1) {
#pragma unroll
                        for (int q = 1; q < SETS; ++q) {
                            JointT &Jo = J[(s + q) % SETS];
                            if (joints && V4_VALID(Jo.key) && V4_JPHASE(Jo.key) == s &&
                                (unsigned)(tick - V4_JROUND(Jo.key)) < (unsigned)span) v4_joint_slot(Jo, sh);
                        }
                    }
                    lds_sync();
                    if (stamp) { t1 = __builtin_amdgcn_s_memtime(); tJ += t1 - t0; }
                    v4_contact_subslots<CSETS, CPAIR>(S, C, sh, lane, NC, spill, pair, subMax[s], s, tick, span, mu);
                    if (stamp) { tC += __builtin_amdgcn_s_memtime() - t1; nSub += subMax[s]; }
                }
            }
        }
        if ((V4_DBG(A) & 32) && lane == 0 && c1 - c0 >= 4) { // the same for tiles of four creatures (tools/slow_tiles_probe.py)
            const unsigned long long tAll = __builtin_amdgcn_s_memtime() - tStart;
            unsigned env = (unsigned)c0;
            EI(E_TOIEVENTS) = (int)(tJ >> 6);
            env = (unsigned)c0 + 1; EI(E_TOIEVENTS) = (int)(tC >> 6);
            env = (unsigned)c0 + 2; EI(E_TOIEVENTS) = (int)(tAll >> 6);
            env = (unsigned)c0 + 3; EI(E_TOIEVENTS) = (int)((nSub << 16) | (unsigned long long)(nTicks & 0xffff));
        }
        if ((V4_DBG(A) & 8) && lane == 0 && c1 - c0 >= 5) { // debug words in the (otherwise unused while stepping) TOI event counters
            const unsigned long long tAll = __builtin_amdgcn_s_memtime() - tStart;
            unsigned env = (unsigned)c0;
            EI(E_TOIEVENTS) = (int)(tJ >> 4);
            env = (unsigned)c0 + 1; EI(E_TOIEVENTS) = (int)(tC >> 4);
            env = (unsigned)c0 + 2; EI(E_TOIEVENTS) = (int)(tAll >> 4);
            env = (unsigned)c0 + 3; EI(E_TOIEVENTS) = (int)nSub;
            env = (unsigned)c0 + 4; EI(E_TOIEVENTS) = nTicks;
            if (c1 - c0 >= 6) { env = (unsigned)c0 + 5; EI(E_TOIEVENTS) = (int)(__builtin_amdgcn_s_memrealtime() - rStart); }
        }
    }
    // ---------------- StoreImpulses, joint impulses, body velocities ----------------
#pragma unroll
    for (int cs = 0; cs < CSETS; ++cs) {
        if (V4_VALID(C[cs].key) && !(pair && (lane & 1))) { // (one lane of a pair stores)
            const int b = V4_CBODY(C[cs].key), t = V4_CT(C[cs].key);
            const unsigned gl = sh.glmap[b];
            const slotpack_t sp = sp_load(S, (unsigned)SCR_MISC_BASE * Lp + gl);
            const unsigned o = SP_GET(sp, t) * Lp + gl;
            CF(C_N0, o) = C[cs].c.n0;
            CF(C_T0, o) = C[cs].c.t0;
            if (C[cs].c.count > 1) {
                CF(C_N1, o) = C[cs].c.n1;
                CF(C_T1, o) = C[cs].c.t1;
            }
        }
    }
    if (spill) {
        for (int ci = CCAP + lane; ci < NC; ci += WAVE) {
            const int e = (int)sh.cmap[ci];
            const int b = V4_CBODY(e), t = V4_CT(e);
            const unsigned gl = sh.glmap[b];
            const unsigned cb = (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl;
            const slotpack_t sp = sp_load(S, (unsigned)SCR_MISC_BASE * Lp + gl);
            const unsigned o = SP_GET(sp, t) * Lp + gl;
            CF(C_N0, o) = SW(cb, 10);
            CF(C_T0, o) = SW(cb, 12);
            if (__float_as_int(SW(cb, 20)) > 1) {
                CF(C_N1, o) = SW(cb, 11);
                CF(C_T1, o) = SW(cb, 13);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
        if (V4_VALID(J[s].key)) {
            const unsigned gl = sh.glmap[V4_JB(J[s].key)];
            LF(L_JIMPX) = J[s].impX; LF(L_JIMPY) = J[s].impY; LF(L_JIMPZ) = J[s].impZ; LF(L_JMOTORIMP) = J[s].motorImp;
        }
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int bl = p * WAVE + lane;
        if (misc[p] & 0x100) {
            const unsigned gl = sh.glmap[bl];
            const V4Vel v = sh.vel[bl];
            LF(L_VX) = v.x; LF(L_VY) = v.y; LF(L_W) = v.w;
        }
    }
    err = wave_or(err);
    if (err) { // a tile that breaks the host's rule leaves joints of ANY of its creatures unsolved: all of them carry the flag
        for (int e = c0 + lane; e < c1; e += WAVE) {
            const unsigned env = retile ? (unsigned)S.order[e] : (unsigned)e;
            atomicOr(&EI(E_ERR), err);
        }
    }
    if ((V4_DBG(A) & 64) && lane == 0) { // diagnostic (tools/chain_probe.py): this wavefront's time in the kernel, 100 MHz ticks
        const unsigned env = (unsigned)c0;
        EI(E_TOIEVENTS) = (int)(__builtin_amdgcn_s_memrealtime() - rEntry);
    }
    if ((V4_DBG(A) & 16) && lane == 0 && c1 - c0 >= 2) { // diagnostic: when did this wavefront start and end (tools/dispatch_probe.py)
        unsigned env = (unsigned)c0;
        EI(E_TOIEVENTS) = (int)(rEntry & 0x7fffffff);
        env = (unsigned)c0 + 1; EI(E_TOIEVENTS) = (int)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
    }
}

// one launch over the tiles of up to REM2D_MAX_BATCH worlds (lane buckets of one population)
struct Vel4Batch {
    State S[REM2D_MAX_BATCH];
    float friction[REM2D_MAX_BATCH];
    unsigned tileEnd[REM2D_MAX_BATCH]; // exclusive prefix sums of the worlds' tile counts
    int lanes[REM2D_MAX_BATCH];
    int n;
};
template <int SETS, int PASSES, int CSETS, int WPS, bool CPAIR, bool FLEXP = true>
__global__ __launch_bounds__(WAVE, WPS) void rem2d_vel4_kernel(Vel4Batch B, Vel4Args A) {
    __shared__ Vel4Shared<SETS, PASSES> sh;
    unsigned tile = blockIdx.x;
    int b = 0;
    while (b + 1 < B.n && tile >= B.tileEnd[b]) ++b;
    if (b > 0) tile -= B.tileEnd[b - 1];
    vel4_body<SETS, PASSES, CSETS, CPAIR, FLEXP>(B.S[b], B.friction[b], A, tile, B.lanes[b], sh);
}

// ---------------------------------------------------------------------------------------------------
// velocity iterations + post in ONE launch (64-lane tiles whose table is regular: State::tileCap).  The launch of a phase
// lasts as long as its slowest wavefront, and the slowest tile of the velocity kernel is usually not the slowest block
// of the position kernel: one wavefront doing both for its 64 lanes costs the launch max(v + p) instead of max v + max p
// (tools/chain_probe.py: 5 % less on config 3's widest lane bucket alone in a launch; 2 % in the mix of a step group), and
// one launch gap less.  Same device functions, same order per creature: same bits.  128 VGPRs, no spills, 4 waves per
// SIMD -- with the velocity half inlined ONCE, in front of the lane-count dispatch of the position half (inside every case
// of it: 66 spilled VGPRs).  Optional (REM2D_FUSE_VELPOST=1, rem2d.hip): +1.0 % on config 3.
// ---------------------------------------------------------------------------------------------------
struct VelPostShared {
    union {
        Vel4Shared<1, 1> v;
        PosShared p;
    };
};
#ifndef REM2D_VELPOST_WAVES
#define REM2D_VELPOST_WAVES 1
#endif
__global__ __launch_bounds__(WAVE * REM2D_VELPOST_WAVES, 4) void rem2d_velpost_kernel(Batch B, StepArgs A, Vel4Args V) {
    __shared__ VelPostShared sh;
    if (REM2D_VELPOST_WAVES > 1 && threadIdx.x >= WAVE) return; // (experiment: what do idle helper wavefronts per block cost?)
    unsigned block = blockIdx.x;
    const int b = batch_find(B, block);
    {
        const int K = B.lanes[b];
        const unsigned cpb = (unsigned)(WAVE / K), cap = (unsigned)B.S[b].tileCap; // creatures per block / per tile; cap divides cpb (host)
        const unsigned t1 = (block + 1) * cpb / cap;
        for (unsigned t = block * cpb / cap; t < t1 && t < (unsigned)B.S[b].nTiles; ++t) {
            vel4_body<1, 1, 1, true>(B.S[b], B.T[b].friction, V, t, K, sh.v);
            lds_sync(); // (the next tile / the position solver reuse the mailbox)
        }
    }
    // the velocities, written per tile lane, are read per block lane below (the same lane unless a block holds several tiles)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    BATCH_DISPATCH(post_only_body, sh.p)
}

// ---------------------------------------------------------------------------------------------------
// The step TRAIN: all steps of a call in ONE launch, without a launch boundary between two steps or two phases of a step.
// One workgroup (a wavefront) per item (step, block), in launch order: item = step * nPad + slot.  It runs `pre` for the creatures
// of the block's slots (in the creature order), their velocity tile(s), their position block and -- continuous physics -- the TOI
// solve of the block's own bodies (the whole wavefront per body, one after the other: what rem2d_toi_heavy_multi_kernel does for
// its work list), then publishes "step s of block b done"; the item of the block's next step waits for that flag first.
//   * No deadlock: the workgroup of a block's previous step has a lower index, so it was dispatched earlier and runs or is done
//     (the forward-progress assumption of every decoupled look-back scan); every wait is bounded by the clock anyway.
//   * Hand-over inside an XCD: nPad = the blocks rounded up to a multiple of 8, workgroups go round-robin over the 8 XCDs, so all
//     steps of a block run on ONE XCD and its state never has to leave that XCD's L2.  Publisher: an EXPLICIT `s_waitcnt vmcnt(0)`
//     (the item's stores are acknowledged by the L2), then the flag.  Waiter: the poll, then an EXPLICIT `buffer_inv sc1` +
//     `s_dcache_inv` (its CU's vector L1 and scalar cache are dropped).  Both are inline assembly, not fences: a workgroup-scope
//     fence compiles to neither on gfx950, and tools/check_handover_asm.py (run by build() and the CPU tests) reads the two
//     sequences back from the code object.  An agent-scope release (an L2 write-back per item) costs 3 %
//     (profiles/r05_step_train.txt).  The flag carries the publisher's XCC_ID and the waiter compares: a hand-over from another
//     XCD or a wait that ran into its limit sets REM2D_ERR_HANDOVER on the block's creatures and counts into the world's
//     failure counter in host memory (rem2d_world_handover_failures) -- never a silent wrong result.  The host re-runs such
//     creatures on per-step launches (evaluate.run_episode).
//   * Same device functions, same order per creature as the separate launches: same bits (the parity suite runs this launch).
// A wavefront slot is refilled one block-step at a time and no launch waits for its slowest tile: config 3 1.30 -> 1.02 ms per
// env-step.  128 VGPRs; the TOI solve (226 VGPRs in its own kernel) spills here (656 spilled VGPRs, all in the post / TOI part: 1 %
// of the bodies go through it), 1 KB of LDS per wavefront (its island's manifolds in one column).
// ---------------------------------------------------------------------------------------------------
template <int K> DEV void pre_intile_body(const State &S, const Terrain &T, const StepArgs &A, unsigned block) { pre_body<K, true>(S, T, A, block); }
// post + (continuous physics) the TOI solve of this block's own bodies, one after the other, by the whole wavefront (G = 64 lanes
// per body, one LDS column); a failed hand-over (`bad`) is flagged on every creature of the block
template <int K>
DEV void post_toi_body(const State &S, const Terrain &T, const StepArgs &A, unsigned block, PosShared &psh, ToiSharedT<1> &ts1, int bad) {
    const bool heavy = post_body<K, true>(S, T, A, block, psh);
    const int lane = threadIdx.x;
    const unsigned slot = (block * WAVE + lane) / K;
    const unsigned env = (S.flags & REM2D_STATE_ORDERED) ? (unsigned)S.order[slot] : slot;
    if (bad && (lane & (K - 1)) == 0) EI(E_ERR) = EI(E_ERR) | REM2D_ERR_HANDOVER;
    if (A.defer != 2) return;
    const unsigned mygl = env * K + (unsigned)(lane & (K - 1));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    unsigned long long hm = __ballot(heavy ? 1 : 0);
#ifdef REM2D_PROBE_SKIP_TOI // (probe build only: what the TOI solves cost the train -- WRONG physics, bodies keep their discrete poses)
    hm = 0ull;
#endif
    while (hm) {
        const int h = __ffsll((long long)hm) - 1;
        hm &= hm - 1;
        const unsigned glh = (unsigned)__shfl((int)mygl, h);
        toi_heavy_one<K, 1>(S, T, A, glh, ts1, lane, WAVE);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}
#define TRAIN_FLAG_WORDS 16 // flags[0 .. 16) spare (flags[1]: "a wait ran into its limit"), flags[16 + block] = steps done | publisher's XCC_ID << 28
#define TRAIN_STEP_BITS 28
#define TRAIN_STEP_MASK ((1u << TRAIN_STEP_BITS) - 1u) // (the host cuts a launch at TRAIN_STEP_MASK steps: rem2d.hip tiles_launch_train)
#define TRAIN_WAIT_TICKS 200000000ull // 2 s of the 100 MHz clock: a hand-over that never comes must not hang the GPU
// REM2D_OPT_TRAIN_FAULT (a test hook, include/rem2d.h): value = s | m << 16 [| 1 << 30]; the items (step s of a launch, block % m == 0):
// plain -- are told that their hand-over failed; with bit 30 -- their predecessor (step s - 1) does not publish, so they wait
// into the limit and every later wait of the launch ends at once
#define TRAIN_FAULT_DROP (1 << 30)
DEV bool train_fault_hits(int fault, int step, unsigned blk) {
    const int s = fault & 0xffff;
    const unsigned m = (unsigned)((fault >> 16) & 0x3fff);
    return fault != 0 && step == s && (m <= 1u || blk % m == 0u);
}
// The waiter's half of a hand-over: lane 0 polls the item's flag until step `step` of the item is published (or the clock / another
// wait's verdict ends the wait), compares the publisher's XCD, books a failure; then the EXPLICIT acquire for the whole wavefront.
// Returns `bad` (uniform).  `fl`: index of the item's flag word; `blk`: what REM2D_OPT_TRAIN_FAULT's modulus counts.
DEV int train_acquire(unsigned *flags, unsigned fl, int step, unsigned xcd, int fault, unsigned blk, unsigned *failures) {
    int bad = 0;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned seen;
        while (((seen = __hip_atomic_load(&flags[fl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & TRAIN_STEP_MASK) < (unsigned)step) {
            __builtin_amdgcn_s_sleep(32);
            // one wait that ran into the limit ends every later wait of the launch at once (flags[1]): the launch drains in
            // milliseconds with REM2D_ERR_HANDOVER on what it touched instead of stalling 2 s per item
            if (__hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { bad = 1; break; }
            if (__builtin_amdgcn_s_memrealtime() - t0 > TRAIN_WAIT_TICKS) {
                __hip_atomic_store(&flags[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad = 1;
                break;
            }
        }
        if ((seen >> TRAIN_STEP_BITS) != xcd) bad = 1; // the item's previous step ran on another XCD: its stores may still sit in that L2
        if (!(fault & TRAIN_FAULT_DROP) && train_fault_hits(fault, step, blk)) bad = 1;
        // the host learns of it without reading the arena: a counter in pinned host memory (rem2d_world_handover_failures)
        if (bad) __hip_atomic_fetch_add(failures, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    bad = __builtin_amdgcn_readfirstlane(bad);
    // ACQUIRE, explicit: this CU's vector L1 and the scalar data cache (a 64-lane creature's per-creature words are uniform
    // loads) forget what they hold from an earlier step of the item; the XCD's L2 is current.  tools/check_handover_asm.py
    // holds the compiled kernels to this sequence (the poll loop, then buffer_inv sc1 + s_dcache_inv, before any other load).
    asm volatile("buffer_inv sc1\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    return bad;
}
// The publisher's half.  RELEASE, explicit: a workgroup-scope fence emits NO wait for outstanding stores on gfx950 (the CU's own L1
// is coherent for its own wavefronts, so LLVM needs none) -- but the reader of this flag runs on ANOTHER CU of the XCD.  vmcnt
// counts stores as well as loads here (no vscnt before gfx10) and a store leaves the count when the XCD's L2 has acknowledged it:
// after this wait every store of the item -- of every lane: one wavefront, one counter -- is in the L2 the next step reads from.
// The flag store must follow with no store of the item's state in between; tools/check_handover_asm.py checks exactly that in the
// code objects.
DEV void train_release(unsigned *flags, unsigned fl, int step, unsigned xcd, int fault, unsigned blk) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && !((fault & TRAIN_FAULT_DROP) && train_fault_hits(fault, step + 1, blk)))
        __hip_atomic_store(&flags[fl], (unsigned)(step + 1) | (xcd << TRAIN_STEP_BITS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define TRAIN_XCC_ID() ((unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u) // HW_REG_XCC_ID[3:0]

__global__ __launch_bounds__(WAVE, 4) void rem2d_step_train_kernel(Batch B, StepArgs A, Vel4Args V, unsigned *flags, unsigned nBlocks, int fault,
                                                                    unsigned *failures) {
    __shared__ VelPostShared sh;
    __shared__ ToiSharedT<1> ts1;
    const unsigned nPad = (nBlocks + 7u) & ~7u;
    const unsigned item = blockIdx.x;
    const int step = (int)(item / nPad);
    const unsigned slot = item % nPad;
    if (slot >= nBlocks) return;
    const unsigned blk = nBlocks - 1 - slot; // (the widest lane bucket, last in the batch, holds the long tiles: first)
    const unsigned xcd = TRAIN_XCC_ID();
    int bad = 0;
    if (step > 0) bad = train_acquire(flags, TRAIN_FLAG_WORDS + blk, step, xcd, fault, blk, failures);
    unsigned block = blk;
    const int b = batch_find(B, block);
    BATCH_DISPATCH(pre_intile_body)
    // what pre wrote per block lane is read per tile lane (constraints) below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    {
        const int K = B.lanes[b];
        const unsigned cpb = (unsigned)(WAVE / K), cap = (unsigned)B.S[b].tileCap; // creatures per block / per tile; cap divides cpb (host)
        const unsigned t1 = (block + 1) * cpb / cap;
        for (unsigned t = block * cpb / cap; t < t1 && t < (unsigned)B.S[b].nTiles; ++t) {
            vel4_body<1, 1, 1, true>(B.S[b], B.T[b].friction, V, t, K, sh.v);
            lds_sync(); // (the next tile / the position solver reuse the mailbox)
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    BATCH_DISPATCH(post_toi_body, sh.p, ts1, bad)
    train_release(flags, TRAIN_FLAG_WORDS + blk, step, xcd, fault, blk);
}

// ---------------------------------------------------------------------------------------------------
// The step train for the 128-lane tile shapes (1: flexible placement, populations from ~130 000 creatures on; 4: the static phase ->
// set map of uniform populations): an item = one velocity TILE = the tile's one or two 64-lane blocks.  A kernel of its own, not a
// parameter of the one above: shared in one body (round 5) the item -> world walk and the loops over a tile's blocks cost the 64-lane
// train 5 %.  Per item: find the world and the tile; wait for the tile's previous step; `pre` block by block; the velocity tile (two
// joint register sets, 128 bodies in the mailbox); `post` + the TOI solve block by block; publish.  Tables must be regular
// (State::tileCap creatures per tile, a whole number of blocks or a divisor of one -- rem2d_world_set_tiles); nItems = the tiles of
// one step summed over the worlds (rem2d.hip train_items does the same sum).  Same device functions, same order per creature: same
// bits.  The hand-over is the one above (train_acquire / train_release), the flag is per item.
// ---------------------------------------------------------------------------------------------------
struct Train128Shared {
    union {
        Vel4Shared<2, 2> v;
        PosShared p;
    };
};
template <bool FLEXP, int WPS>
__global__ __launch_bounds__(WAVE, WPS) void rem2d_step_train128_kernel(Batch B, StepArgs A, Vel4Args V, unsigned *flags, unsigned nItems, int fault,
                                                                         unsigned *failures) {
    __shared__ Train128Shared sh;
    __shared__ ToiSharedT<1> ts1;
    const unsigned nPad = (nItems + 7u) & ~7u;
    const unsigned item = blockIdx.x;
    const int step = (int)(item / nPad);
    const unsigned slot = item % nPad;
    if (slot >= nItems) return;
    const unsigned idx = nItems - 1 - slot; // (the widest lane bucket, last in the batch, holds the long tiles: first)
    // the item's world, its tile `it` there and the tile's blocks [blk0, blk0 + nb)
    unsigned it = idx;
    int b = 0;
    unsigned bpt = 1, blocksW = 0;
    for (;; ++b) {
        blocksW = B.blockEnd[b] - (b ? B.blockEnd[b - 1] : 0u);
        const unsigned lanesPerTile = (unsigned)B.S[b].tileCap * (unsigned)B.lanes[b];
        bpt = lanesPerTile > WAVE ? lanesPerTile / WAVE : 1u;
        const unsigned itemsW = (blocksW + bpt - 1) / bpt;
        if (it < itemsW || b + 1 >= B.n) break;
        it -= itemsW;
    }
    const unsigned blk0 = it * bpt, nb = blk0 + bpt <= blocksW ? bpt : blocksW - blk0;
    const unsigned xcd = TRAIN_XCC_ID();
    int bad = 0;
    if (step > 0) bad = train_acquire(flags, TRAIN_FLAG_WORDS + idx, step, xcd, fault, idx, failures);
    for (unsigned p = 0; p < nb; ++p) {
        unsigned block = blk0 + p;
        BATCH_DISPATCH(pre_intile_body)
    }
    // what pre wrote per block lane is read per tile lane (constraints) below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    {
        const int K = B.lanes[b];
        const unsigned cpb = (unsigned)(WAVE / K), cap = (unsigned)B.S[b].tileCap;
        // a tile of whole blocks: tile `it`; tiles inside one block (a small world): the block's tiles
        const bool whole = cap * (unsigned)K > WAVE;
        const unsigned ta = whole ? it : blk0 * cpb / cap, tb = whole ? it + 1 : (blk0 + 1) * cpb / cap;
        for (unsigned t = ta; t < tb && t < (unsigned)B.S[b].nTiles; ++t) {
            vel4_body<2, 2, 1, false, FLEXP>(B.S[b], B.T[b].friction, V, t, K, sh.v);
            lds_sync(); // (the next tile / the position solver reuse the mailbox)
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (unsigned p = 0; p < nb; ++p) {
        unsigned block = blk0 + p;
        BATCH_DISPATCH(post_toi_body, sh.p, ts1, bad)
        lds_sync(); // (the next block's position solver reuses the mailbox)
    }
    train_release(flags, TRAIN_FLAG_WORDS + idx, step, xcd, fault, idx);
}

#endif
