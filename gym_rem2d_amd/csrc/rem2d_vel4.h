// rem2d_vel4.h -- velocity kernel of the split pipeline, tile form (the default since round 2).
// Part of the single translation unit rem2d.hip; not a stand-alone header.
//
// b2Island::Solve's 180 velocity iterations are one long dependent chain per creature; what a wavefront can
// change is how many creatures ride on one chain.  rem2d_step_kernel keeps one body per lane: a joint slot runs
// with 1/period of its lanes and a contact slot with ~5 % of them, so the population needs ~4 rounds of
// wavefronts, each paying the whole chain.  Here a wavefront owns a TILE: a run of consecutive creatures with
// at most 256 bodies whose joints number at most 64 in every phase of the modulo schedule (the host cuts the
// tiles, compiler.Morphology.tiles / rem2d_world_set_tiles).  Lanes are constraints:
//   * joints: lane r of register set s holds the r-th joint of phase s (joint round mod period) -- in the tick
//     of phase s all of them fire together, so a joint slot runs with up to 64 active lanes;
//   * contacts: every touching manifold of the tile gets a lane of its own (V4_CSETS register sets of 64);
//     a body's manifolds are solved in list order in consecutive sub-slots (sub-slot t = the t-th manifold of
//     every body whose contact slot is this tick);
//   * body velocities live in an LDS mailbox ({vx, vy, w} as one 16-byte record per body); hand-offs are
//     wave-local (lds_sync = s_waitcnt, no s_barrier).
// The whole 65 536-creature population is ~2000 tiles, i.e. two resident wavefronts per SIMD: one round.
// Arithmetic and ordering are those of the other forms (compiler.pipeline_schedule proves that any two
// operations sharing a body keep Box2D's sequential order for any period >= the creature's own), so the result
// is bit-identical to rem2d_step_kernel and to the oracle.
#ifndef REM2D_VEL4_H
#define REM2D_VEL4_H

#define V4_SETS 4          // joint register sets = largest schedule period supported (reference modules: <= 4)
#define V4_CSETS 2         // contact register sets: 128 manifolds per tile in registers, the rest through scratch
#define V4_MAX_BODIES 256  // bodies per tile
#define V4_PASSES (V4_MAX_BODIES / WAVE)
#define V4_MAX_CONTACTS (V4_MAX_BODIES * KT)

struct Vel4Args { int velIters; float dt; };

struct __attribute__((aligned(16))) V4Vel { float x, y, w, pad; };
struct Vel4Shared {
    V4Vel vel[V4_MAX_BODIES];
    unsigned short jmap[V4_SETS][WAVE];     // (set, lane) -> tile-local body id of the joint's child, 0xffff = none
    unsigned short cmap[V4_MAX_CONTACTS];   // contact rank -> tile-local body id | manifold index << 8
};

// one joint: what the 180 iterations read (constants) and write (accumulated impulses)
struct JointT {
    V2 rA, rB;
    float mA, iA, mB, iB;
    float exx, eyx, ezx, eyy, ezy, ezz, motorMass, det33, det22, cyzx, cyzy, cyzz;
    float impX, impY, impZ, motorImp, motorSpeed, maxMotorImpulse;
    int key; // jA | jB << 8 | jround << 16 | limitState << 24 | valid << 31
};
#define V4_JA(k) ((k) & 0xff)
#define V4_JB(k) (((k) >> 8) & 0xff)
#define V4_JROUND(k) (((k) >> 16) & 0xff)
#define V4_LIMIT(k) (((k) >> 24) & 0x3)
#define V4_VALID(k) ((k) < 0)

DEV void v4_joint_load(const State &S, unsigned tb0, int K, int child, float h, JointT &J) {
    const unsigned Lp = S.Lp;
    const unsigned gl = tb0 + (unsigned)child;
    const int jA = (child & ~(K - 1)) + LI(L_PARENT);
    const int jround = LI(L_JROUND) & 0xff;
    const unsigned jb = (unsigned)SCR_JREC_BASE * Lp + gl;
    J.rA = mk(SW(jb, 0), SW(jb, 1));
    J.rB = mk(SW(jb, 2), SW(jb, 3));
    J.mB = LF(L_INVM); J.iB = LF(L_INVI);
    J.impX = LF(L_JIMPX); J.impY = LF(L_JIMPY); J.impZ = LF(L_JIMPZ); J.motorImp = LF(L_JMOTORIMP);
    J.motorSpeed = LF(L_JMOTORSPEED);
    const int limitState = LI(L_JLIMIT);
    J.maxMotorImpulse = h * LF(L_JTORQUE);
    {
        const unsigned gl = tb0 + (unsigned)jA;
        J.mA = LF(L_INVM); J.iA = LF(L_INVI);
    }
    J.key = jA | (child << 8) | (jround << 16) | (limitState << 24) | (int)0x80000000;
    const float mA = J.mA, iA = J.iA, mB = J.mB, iB = J.iB;
    const V2 rA = J.rA, rB = J.rB;
    // effective-mass terms of b2RevoluteJoint::InitVelocityConstraints (same expressions as rem2d_step_kernel)
    J.exx = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
    J.eyx = -rA.y * rA.x * iA - rB.y * rB.x * iB;
    J.ezx = -rA.y * iA - rB.y * iB;
    J.eyy = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
    J.ezy = rA.x * iA + rB.x * iB;
    J.ezz = iA + iB;
    J.motorMass = iA + iB;
    if (J.motorMass > 0.0f) J.motorMass = 1.0f / J.motorMass;
    J.cyzx = J.eyy * J.ezz - J.ezy * J.ezy;
    J.cyzy = J.ezy * J.ezx - J.eyx * J.ezz;
    J.cyzz = J.eyx * J.ezy - J.eyy * J.ezx;
    J.det33 = J.exx * J.cyzx + J.eyx * J.cyzy + J.ezx * J.cyzz;
    if (J.det33 != 0.0f) J.det33 = 1.0f / J.det33;
    J.det22 = J.exx * J.eyy - J.eyx * J.eyx;
    if (J.det22 != 0.0f) J.det22 = 1.0f / J.det22;
}

// b2RevoluteJoint::SolveVelocityConstraints (motor, then limit 3x3 / point 2x2) on the mailbox records of the
// joint's two bodies; same expression sequence as the joint slot of rem2d_step_kernel
DEV void v4_joint_slot(JointT &j, Vel4Shared &sh) {
    const int a = V4_JA(j.key), b = V4_JB(j.key), limitState = V4_LIMIT(j.key);
    const V4Vel ra = sh.vel[a], rb = sh.vel[b];
    V2 vA = mk(ra.x, ra.y), vB = mk(rb.x, rb.y);
    float wA = ra.w, wB = rb.w;
    if (limitState != LIM_EQUAL) {
        float Cdot = wB - wA - j.motorSpeed;
        float impulse = -j.motorMass * Cdot;
        float oldImpulse = j.motorImp;
        j.motorImp = fclamp(oldImpulse + impulse, -j.maxMotorImpulse, j.maxMotorImpulse);
        impulse = j.motorImp - oldImpulse;
        wA -= j.iA * impulse;
        wB += j.iB * impulse;
    }
    if (limitState != LIM_INACTIVE) {
        V2 Cdot1 = vsub(vsub(vadd(vB, vcross_sv(wB, j.rB)), vA), vcross_sv(wA, j.rA));
        float Cdot2 = wB - wA;
        float bx = Cdot1.x, by = Cdot1.y, bz = Cdot2;
        float sx = j.det33 * (bx * j.cyzx + by * j.cyzy + bz * j.cyzz);
        float cbx = by * j.ezz - bz * j.ezy, cby = bz * j.ezx - bx * j.ezz, cbz = bx * j.ezy - by * j.ezx;
        float sy = j.det33 * (j.exx * cbx + j.eyx * cby + j.ezx * cbz);
        float ebx = j.eyy * bz - j.ezy * by, eby = j.ezy * bx - j.eyx * bz, ebz = j.eyx * by - j.eyy * bx;
        float sz = j.det33 * (j.exx * ebx + j.eyx * eby + j.ezx * ebz);
        float ix = -sx, iy = -sy, iz = -sz;
        if (limitState == LIM_EQUAL) {
            j.impX += ix; j.impY += iy; j.impZ += iz;
        } else {
            float newImpulse = j.impZ + iz;
            bool reduce = limitState == LIM_AT_LOWER ? newImpulse < 0.0f : newImpulse > 0.0f;
            if (reduce) {
                V2 rhs = vadd(vneg(Cdot1), vscale(j.impZ, mk(j.ezx, j.ezy)));
                float rx = j.det22 * (j.eyy * rhs.x - j.eyx * rhs.y);
                float ry = j.det22 * (j.exx * rhs.y - j.eyx * rhs.x);
                ix = rx; iy = ry; iz = -j.impZ;
                j.impX += rx; j.impY += ry; j.impZ = 0.0f;
            } else {
                j.impX += ix; j.impY += iy; j.impZ += iz;
            }
        }
        V2 P = mk(ix, iy);
        vA = vsub(vA, vscale(j.mA, P));
        wA -= j.iA * (vcross(j.rA, P) + iz);
        vB = vadd(vB, vscale(j.mB, P));
        wB += j.iB * (vcross(j.rB, P) + iz);
    } else {
        V2 Cdot = vsub(vsub(vadd(vB, vcross_sv(wB, j.rB)), vA), vcross_sv(wA, j.rA));
        V2 bb = vneg(Cdot);
        V2 impulse = mk(j.det22 * (j.eyy * bb.x - j.eyx * bb.y), j.det22 * (j.exx * bb.y - j.eyx * bb.x));
        j.impX += impulse.x; j.impY += impulse.y;
        vA = vsub(vA, vscale(j.mA, impulse));
        wA -= j.iA * vcross(j.rA, impulse);
        vB = vadd(vB, vscale(j.mB, impulse));
        wB += j.iB * vcross(j.rB, impulse);
    }
    V4Vel oa, ob;
    oa.x = vA.x; oa.y = vA.y; oa.w = wA; oa.pad = 0.0f;
    ob.x = vB.x; ob.y = vB.y; ob.w = wB; ob.pad = 0.0f;
    sh.vel[a] = oa;
    sh.vel[b] = ob;
}

// one contact lane: the manifold's constraint, its body's inverse mass / inertia and its place in the schedule
struct ContactT {
    ContactC c;
    float mB, iB;
    int key; // body | manifold index << 8 | offC << 16 | (offC mod P) << 24 | valid << 31
};
#define V4_CBODY(k) ((k) & 0xff)
#define V4_CT(k) (((k) >> 8) & 0xff)
#define V4_COFF(k) (((k) >> 16) & 0xff)
#define V4_CPHASE(k) (((k) >> 24) & 0x7)

DEV void vel4_body(const State &S, const float friction, const Vel4Args &A, unsigned tile, int K, Vel4Shared &sh) {
    const int lane = threadIdx.x;
    const unsigned Lp = S.Lp;
    const int c0 = S.tiles[tile], c1 = S.tiles[tile + 1];
    const int NB = (c1 - c0) * K;               // bodies of this tile (<= V4_MAX_BODIES, checked by the host)
    const unsigned tb0 = (unsigned)c0 * (unsigned)K;
    const int iters = A.velIters;
    const float h = A.dt, mu = friction;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (WAVE - lane));
    if (S.flags & REM2D_FLAG_SKIP_FROZEN) { // pre skipped these creatures: nothing to solve, nothing handed over
        bool allFrozen = true;
        for (int e = c0 + lane; e < c1; e += WAVE) {
            const unsigned env = (unsigned)e;
            allFrozen = allFrozen && EI(E_FROZEN) != 0;
        }
        if (__all(allFrozen ? 1 : 0)) return;
    }

#pragma unroll
    for (int s = 0; s < V4_SETS; ++s) sh.jmap[s][lane] = 0xffff;
    // ---------------- body role (<= 4 passes of 64 bodies): publish velocities, deal joints and manifolds to lanes ----------------
    int sched[V4_PASSES], misc[V4_PASSES], parent[V4_PASSES];
    int P = 1;
#pragma unroll
    for (int p = 0; p < V4_PASSES; ++p) {
        const int bl = p * WAVE + lane;
        const unsigned gl = tb0 + (unsigned)bl;
        sched[p] = 0; misc[p] = 0; parent[p] = -1;
        if (bl < NB) {
            misc[p] = __float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + gl, 0));
            sched[p] = LI(L_JROUND);
            parent[p] = LI(L_PARENT);
            V4Vel v;
            v.x = LF(L_VX); v.y = LF(L_VY); v.w = LF(L_W); v.pad = 0.0f;
            sh.vel[bl] = v;
        }
        P = max(P, (sched[p] >> 16) & 0xff); // every creature of the tile, awake or not: the host cut the tile with this period
    }
    P = wave_max(P);
    int NC = 0, lastTick = -1, maxRound = -1, err = 0;
    int jcount[V4_SETS];
#pragma unroll
    for (int s = 0; s < V4_SETS; ++s) jcount[s] = 0;
    int subMax[V4_SETS]; // contact sub-slots per phase = most manifolds on one body whose contact slot has that phase
#pragma unroll
    for (int s = 0; s < V4_SETS; ++s) subMax[s] = 0;
#pragma unroll
    for (int p = 0; p < V4_PASSES; ++p) {
        const int bl = p * WAVE + lane;
        const bool solve = (misc[p] & 0x100) != 0;
        const bool hasJ = solve && parent[p] >= 0;
        const int nT = solve ? (misc[p] & 0xff) : 0;
        const int jr = sched[p] & 0xff, oc = (sched[p] >> 8) & 0xff;
        const int phase = jr % P, cphase = oc % P;
#pragma unroll
        for (int s = 0; s < V4_SETS; ++s) {
            const unsigned long long m = __ballot(hasJ && phase == s);
            if (hasJ && phase == s) {
                const int rank = jcount[s] + __popcll(m & below);
                if (rank < WAVE) sh.jmap[s][rank] = (unsigned short)bl;
                else err = REM2D_ERR_SOLVER_OVERFLOW; // the tile breaks the host's rule (<= 64 joints per phase)
            }
            jcount[s] += __popcll(m);
            if (nT > 0 && cphase == s) subMax[s] = max(subMax[s], nT);
        }
        if (hasJ && phase >= V4_SETS) err = REM2D_ERR_SOLVER_OVERFLOW;
        if (nT > 0 && cphase >= V4_SETS) err = REM2D_ERR_SOLVER_OVERFLOW;
#pragma unroll
        for (int t = 0; t < KT; ++t) { // manifold t of every body: ranks in (pass, t, lane) order
            const unsigned long long cm = __ballot(nT > t);
            if (nT > t) sh.cmap[NC + __popcll(cm & below)] = (unsigned short)(bl | (t << 8));
            NC += __popcll(cm);
        }
        if (iters > 0) {
            if (hasJ) lastTick = max(lastTick, jr + (iters - 1) * P);
            if (nT > 0) lastTick = max(lastTick, oc + (iters - 1) * P);
        }
        if (hasJ) maxRound = max(maxRound, jr);
    }
    const int nTicks = wave_max(lastTick) + 1;
    const int nRounds = wave_max(maxRound) + 1;
    int maxSub = 0;
#pragma unroll
    for (int s = 0; s < V4_SETS; ++s) { subMax[s] = wave_max(subMax[s]); maxSub = max(maxSub, subMax[s]); }
    lds_sync();

    // ---------------- joint role: one joint per phase and lane ----------------
    JointT J[V4_SETS];
#pragma unroll
    for (int s = 0; s < V4_SETS; ++s) {
        J[s].key = 0;
        const int child = sh.jmap[s][lane];
        if (s < P && child != 0xffff) v4_joint_load(S, tb0, K, child, h, J[s]);
    }
    // ---------------- contact role: manifold `lane + 64 cs` of the tile; beyond V4_CSETS * 64 through scratch ----------------
    ContactT C[V4_CSETS];
#pragma unroll
    for (int cs = 0; cs < V4_CSETS; ++cs) {
        C[cs].key = 0;
        C[cs].c.count = 0;
        C[cs].mB = C[cs].iB = 0.0f;
        const int ci = cs * WAVE + lane;
        if (ci < NC) {
            const int e = sh.cmap[ci];
            const int b = e & 0xff, t = e >> 8;
            const unsigned gl = tb0 + (unsigned)b;
            const int oc = (LI(L_JROUND) >> 8) & 0xff;
            C[cs].key = b | (t << 8) | (oc << 16) | ((oc % P) << 24) | (int)0x80000000;
            C[cs].mB = LF(L_INVM); C[cs].iB = LF(L_INVI);
            cc_load(S, (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl, C[cs].c);
        }
    }
    const bool spill = NC > V4_CSETS * WAVE; // wave-uniform
    // ---------------- warm start: contacts (per body in list order), then joints in island rounds ----------------
    for (int t = 0; t < maxSub; ++t) {
#pragma unroll
        for (int cs = 0; cs < V4_CSETS; ++cs) {
            if (V4_VALID(C[cs].key) && V4_CT(C[cs].key) == t) {
                const int b = V4_CBODY(C[cs].key);
                V4Vel v = sh.vel[b];
                contact_warm_start(C[cs].c, C[cs].mB, C[cs].iB, v.x, v.y, v.w);
                sh.vel[b] = v;
            }
        }
        if (spill) {
            for (int ci = V4_CSETS * WAVE + lane; ci < NC; ci += WAVE) {
                const int e = sh.cmap[ci];
                const int b = e & 0xff;
                if ((e >> 8) != t) continue;
                const unsigned gl = tb0 + (unsigned)b;
                ContactC c;
                cc_load(S, (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl, c);
                V4Vel v = sh.vel[b];
                contact_warm_start(c, LF(L_INVM), LF(L_INVI), v.x, v.y, v.w);
                sh.vel[b] = v;
            }
        }
        lds_sync();
    }
    for (int r = 0; r < nRounds; ++r) {
#pragma unroll
        for (int s = 0; s < V4_SETS; ++s) {
            if (V4_VALID(J[s].key) && V4_JROUND(J[s].key) == r) {
                const int a = V4_JA(J[s].key), b = V4_JB(J[s].key);
                V4Vel ra = sh.vel[a], rb = sh.vel[b];
                V2 vA = mk(ra.x, ra.y), vB = mk(rb.x, rb.y);
                float wA = ra.w, wB = rb.w;
                V2 Pw = mk(J[s].impX, J[s].impY);
                vA = vsub(vA, vscale(J[s].mA, Pw));
                wA -= J[s].iA * (vcross(J[s].rA, Pw) + J[s].motorImp + J[s].impZ);
                vB = vadd(vB, vscale(J[s].mB, Pw));
                wB += J[s].iB * (vcross(J[s].rB, Pw) + J[s].motorImp + J[s].impZ);
                ra.x = vA.x; ra.y = vA.y; ra.w = wA;
                rb.x = vB.x; rb.y = vB.y; rb.w = wB;
                sh.vel[a] = ra;
                sh.vel[b] = rb;
            }
        }
        lds_sync();
    }
    // ---------------- velocity iterations ----------------
    {
        const int span = iters * P; // joint k fires at ticks jround + i P, i < iters, i.e. while tick - jround < span
        int ph = 0;
        for (int tick = 0; tick < nTicks; ++tick) {
            // joint slot: the register set of this tick's phase (wave-uniform switch)
#pragma unroll
            for (int s = 0; s < V4_SETS; ++s) {
                if (s == ph) {
                    if (V4_VALID(J[s].key) && (unsigned)(tick - V4_JROUND(J[s].key)) < (unsigned)span) v4_joint_slot(J[s], sh);
                }
            }
            lds_sync();
            // contact sub-slots of the bodies whose contact slot is this tick
            int nsub = 0;
#pragma unroll
            for (int s = 0; s < V4_SETS; ++s) nsub = (s == ph) ? subMax[s] : nsub;
            for (int t = 0; t < nsub; ++t) {
#pragma unroll
                for (int cs = 0; cs < V4_CSETS; ++cs) {
                    // offC = ph (mod P) and offC <= tick < offC + iters P: this tick is the body's contact slot
                    if (V4_VALID(C[cs].key) && V4_CT(C[cs].key) == t && V4_CPHASE(C[cs].key) == ph &&
                        (unsigned)(tick - V4_COFF(C[cs].key)) < (unsigned)span) {
                        const int b = V4_CBODY(C[cs].key);
                        V4Vel v = sh.vel[b];
                        contact_solve(C[cs].c, C[cs].mB, C[cs].iB, mu, v.x, v.y, v.w);
                        sh.vel[b] = v;
                    }
                }
                if (spill) {
                    for (int ci = V4_CSETS * WAVE + lane; ci < NC; ci += WAVE) {
                        const int e = sh.cmap[ci];
                        const int b = e & 0xff;
                        if ((e >> 8) != t) continue;
                        const unsigned gl = tb0 + (unsigned)b;
                        const int d = tick - ((LI(L_JROUND) >> 8) & 0xff);
                        if (!((unsigned)d < (unsigned)span && d % P == 0)) continue;
                        const unsigned cb = (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl;
                        ContactC c;
                        cc_load(S, cb, c);
                        V4Vel v = sh.vel[b];
                        contact_solve(c, LF(L_INVM), LF(L_INVI), mu, v.x, v.y, v.w);
                        sh.vel[b] = v;
                        SW(cb, 10) = c.n0; SW(cb, 11) = c.n1; SW(cb, 12) = c.t0; SW(cb, 13) = c.t1;
                    }
                }
                lds_sync();
            }
            ph = ph + 1 == P ? 0 : ph + 1;
        }
    }
    // ---------------- StoreImpulses, joint impulses, body velocities ----------------
#pragma unroll
    for (int cs = 0; cs < V4_CSETS; ++cs) {
        if (V4_VALID(C[cs].key)) {
            const int b = V4_CBODY(C[cs].key), t = V4_CT(C[cs].key);
            const unsigned gl = tb0 + (unsigned)b;
            const unsigned sp = (unsigned)__float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + gl, 1));
            const unsigned o = ((sp >> (5 * t)) & 0x1f) * Lp + gl;
            CF(C_N0, o) = C[cs].c.n0;
            CF(C_T0, o) = C[cs].c.t0;
            if (C[cs].c.count > 1) {
                CF(C_N1, o) = C[cs].c.n1;
                CF(C_T1, o) = C[cs].c.t1;
            }
        }
    }
    if (spill) {
        for (int ci = V4_CSETS * WAVE + lane; ci < NC; ci += WAVE) {
            const int e = sh.cmap[ci];
            const int b = e & 0xff, t = e >> 8;
            const unsigned gl = tb0 + (unsigned)b;
            const unsigned cb = (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl;
            const unsigned sp = (unsigned)__float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + gl, 1));
            const unsigned o = ((sp >> (5 * t)) & 0x1f) * Lp + gl;
            CF(C_N0, o) = SW(cb, 10);
            CF(C_T0, o) = SW(cb, 12);
            if (__float_as_int(SW(cb, 20)) > 1) {
                CF(C_N1, o) = SW(cb, 11);
                CF(C_T1, o) = SW(cb, 13);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < V4_SETS; ++s) {
        if (V4_VALID(J[s].key)) {
            const unsigned gl = tb0 + (unsigned)V4_JB(J[s].key);
            LF(L_JIMPX) = J[s].impX; LF(L_JIMPY) = J[s].impY; LF(L_JIMPZ) = J[s].impZ; LF(L_JMOTORIMP) = J[s].motorImp;
        }
    }
#pragma unroll
    for (int p = 0; p < V4_PASSES; ++p) {
        const int bl = p * WAVE + lane;
        if (misc[p] & 0x100) {
            const unsigned gl = tb0 + (unsigned)bl;
            const V4Vel v = sh.vel[bl];
            LF(L_VX) = v.x; LF(L_VY) = v.y; LF(L_W) = v.w;
        }
    }
    err = wave_or(err);
    if (err && lane == 0) {
        const unsigned env = (unsigned)c0;
        atomicOr(&EI(E_ERR), err);
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
__global__ __launch_bounds__(WAVE, 2) void rem2d_vel4_kernel(Vel4Batch B, Vel4Args A) {
    __shared__ Vel4Shared sh;
    unsigned tile = blockIdx.x;
    int b = 0;
    while (b + 1 < B.n && tile >= B.tileEnd[b]) ++b;
    if (b > 0) tile -= B.tileEnd[b - 1];
    vel4_body(B.S[b], B.friction[b], A, tile, B.lanes[b], sh);
}

#endif
