// rem2d_vel3.h -- velocity kernel of the split pipeline, third form: ONE wavefront per workgroup, several
// joints per lane.  Part of the single translation unit rem2d.hip; not a stand-alone header.
//
// rem2d_step_kernel runs a joint slot with 1/period of its lanes active: a lane owns one joint and that
// joint fires once per period.  rem2d_vel_kernel (4-wave workgroups, lanes re-dealt to constraints) fills the
// lanes but serialises on s_barrier: one busy wavefront per workgroup.  Here a wavefront owns `cw`
// consecutive creatures (cw * K <= 256 bodies; their velocities in an LDS mailbox) and a lane owns up to
// V3_SETS joints, ONE PER PHASE of the modulo schedule, in separate register sets: in the tick of phase p
// every lane that holds a phase-p joint works, so the joint slot runs at up to full occupancy with nothing
// but wave-local LDS hand-offs (lds_sync, no s_barrier).  Creature i of the wavefront gets the lanes
// [i*Q, (i+1)*Q); the joints of one phase of one creature are pairwise body-disjoint (they are a matching of
// its tree), so there are at most K/2 of them and Q = K/2 always suffices; the host may pass a smaller Q
// (more creatures per wavefront) when it knows the population's largest phase class.  Touching bodies are
// compacted into contact lanes (one body per lane, KR constraints in registers, the rest and any body beyond
// 64 through scratch).  Same arithmetic and the same ordering argument as the other two forms.
#ifndef REM2D_VEL3_H
#define REM2D_VEL3_H

#define V3_SETS 4          // register sets = largest schedule period supported (reference modules: <= 4)
#define V3_MAX_BODIES 256  // bodies per wavefront
#define V3_PASSES (V3_MAX_BODIES / WAVE)

struct Vel3Args { int K, Q, cw; int velIters; float dt, friction; };

struct Vel3Shared {
    float vel[3][V3_MAX_BODIES];
    unsigned char jmap[V3_SETS][WAVE]; // (set, lane) -> wave-local body id of the joint's child, 0xff = none
    unsigned char clist[V3_MAX_BODIES]; // contact lane -> wave-local body id
};

DEV void joint_load(const State &S, unsigned wb0, int K, int child, float h, JointV &J, int &jA, int &jround) {
    const unsigned Lp = S.Lp;
    const unsigned gl = wb0 + (unsigned)child;
    jA = (child & ~(K - 1)) + LI(L_PARENT);
    jround = LI(L_JROUND) & 0xff;
    const unsigned jb = (unsigned)SCR_JREC_BASE * Lp + gl;
    J.rA = mk(SW(jb, 0), SW(jb, 1));
    J.rB = mk(SW(jb, 2), SW(jb, 3));
    J.mB = LF(L_INVM); J.iB = LF(L_INVI);
    J.impX = LF(L_JIMPX); J.impY = LF(L_JIMPY); J.impZ = LF(L_JIMPZ); J.motorImp = LF(L_JMOTORIMP);
    J.motorSpeed = LF(L_JMOTORSPEED);
    J.limitState = LI(L_JLIMIT);
    J.maxMotorImpulse = h * LF(L_JTORQUE);
    {
        const unsigned gl = wb0 + (unsigned)jA;
        J.mA = LF(L_INVM); J.iA = LF(L_INVI);
    }
    const float mA = J.mA, iA = J.iA, mB = J.mB, iB = J.iB;
    const V2 rA = J.rA, rB = J.rB;
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

DEV void vel3_body(const State &S, const Vel3Args &A, unsigned block, Vel3Shared &sh) {
    const int lane = threadIdx.x;
    const unsigned Lp = S.Lp;
    const int K = A.K, Q = A.Q;
    const int NB = A.cw * K;                 // bodies of this wavefront (<= V3_MAX_BODIES)
    const unsigned wb0 = block * (unsigned)NB;
    const int iters = A.velIters;
    const float h = A.dt, mu = A.friction;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (WAVE - lane));
    const unsigned long long group = (K >= WAVE ? ~0ull : ((1ull << K) - 1ull)) << (lane & ~(K - 1));

#pragma unroll
    for (int s = 0; s < V3_SETS; ++s) sh.jmap[s][lane] = 0xff;
    // ---------------- body role (NB / 64 passes): publish velocities, deal joints and contacts to lanes ----------------
    int sched[V3_PASSES], misc[V3_PASSES], parent[V3_PASSES];
    int P = 1;
#pragma unroll
    for (int p = 0; p < V3_PASSES; ++p) {
        const int bl = p * WAVE + lane;
        const unsigned gl = wb0 + (unsigned)bl;
        sched[p] = 0; misc[p] = 0; parent[p] = -1;
        if (bl < NB && gl < Lp) {
            misc[p] = __float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + gl, 0));
            sched[p] = LI(L_JROUND);
            parent[p] = LI(L_PARENT);
            sh.vel[0][bl] = LF(L_VX); sh.vel[1][bl] = LF(L_VY); sh.vel[2][bl] = LF(L_W);
        }
        const int per = (misc[p] & 0x100) ? ((sched[p] >> 16) & 0xff) : 0;
        P = max(P, per);
    }
    P = wave_max(P);
    lds_sync();
    int NC = 0, lastTick = -1, maxRound = -1, err = 0;
    int cPhases = 0;
#pragma unroll
    for (int p = 0; p < V3_PASSES; ++p) {
        const int bl = p * WAVE + lane;
        const bool solve = (misc[p] & 0x100) != 0;
        const bool hasJ = solve && parent[p] >= 0;
        const bool touching = solve && (misc[p] & 0xff) > 0;
        const int jr = sched[p] & 0xff, oc = (sched[p] >> 8) & 0xff;
        const int phase = jr % P;
        const int ci = bl / K; // creature within the wavefront
#pragma unroll
        for (int s = 0; s < V3_SETS; ++s) {
            const unsigned long long m = __ballot(hasJ && phase == s);
            if (hasJ && phase == s) {
                const int rank = __popcll(m & group & below);
                if (rank < Q) sh.jmap[s][ci * Q + rank] = (unsigned char)bl;
                else err = REM2D_ERR_SOLVER_OVERFLOW; // the host chose Q too small for this creature
            }
        }
        if (hasJ && phase >= V3_SETS) err = REM2D_ERR_SOLVER_OVERFLOW;
        const unsigned long long cm = __ballot(touching);
        if (touching) sh.clist[NC + __popcll(cm & below)] = (unsigned char)bl;
        NC += __popcll(cm);
        if (iters > 0) {
            if (hasJ) lastTick = max(lastTick, jr + (iters - 1) * P);
            if (touching) lastTick = max(lastTick, oc + (iters - 1) * P);
        }
        if (hasJ) maxRound = max(maxRound, jr);
        if (touching) cPhases |= P <= 32 ? (1 << (oc % P)) : -1;
    }
    const int nTicks = wave_max(lastTick) + 1;
    const int nRounds = wave_max(maxRound) + 1;
    cPhases = wave_or(cPhases);
    lds_sync();

    // ---------------- joint role: one joint per phase and lane ----------------
    JointV J[V3_SETS];
    int jA[V3_SETS], jB[V3_SETS], jround[V3_SETS];
#pragma unroll
    for (int s = 0; s < V3_SETS; ++s) {
        jB[s] = sh.jmap[s][lane];
        jA[s] = 0; jround[s] = -1;
        J[s].limitState = LIM_INACTIVE;
        if (s < P && jB[s] != 0xff) joint_load(S, wb0, K, jB[s], h, J[s], jA[s], jround[s]);
    }
    // ---------------- contact role: lane i takes touching body clist[i]; bodies beyond 64 go through scratch ----------------
    const bool crole = lane < NC;
    ContactC cc[KR];
#pragma unroll
    for (int t = 0; t < KR; ++t) cc[t].count = 0;
    int cBody = 0, nTouch = 0, offC = 0;
    unsigned glC = 0, slotPack = 0u;
    float cmB = 0.0f, ciB = 0.0f;
    if (crole) {
        cBody = sh.clist[lane];
        const unsigned gl = wb0 + (unsigned)cBody;
        glC = gl;
        const unsigned mb = (unsigned)SCR_MISC_BASE * Lp + gl;
        nTouch = __float_as_int(SW(mb, 0)) & 0xff;
        slotPack = (unsigned)__float_as_int(SW(mb, 1));
        offC = (LI(L_JROUND) >> 8) & 0xff;
        cmB = LF(L_INVM); ciB = LF(L_INVI);
#pragma unroll
        for (int t = 0; t < KR; ++t)
            if (t < nTouch) cc_load(S, (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl, cc[t]);
    }
    // ---------------- warm start: contacts, then joints in island rounds ----------------
    if (crole) {
        float cvx = sh.vel[0][cBody], cvy = sh.vel[1][cBody], cw = sh.vel[2][cBody];
#pragma unroll
        for (int t = 0; t < KR; ++t)
            if (t < nTouch) contact_warm_start(cc[t], cmB, ciB, cvx, cvy, cw);
        for (int t = KR; t < nTouch; ++t) {
            ContactC c;
            cc_load(S, (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + glC, c);
            contact_warm_start(c, cmB, ciB, cvx, cvy, cw);
        }
        sh.vel[0][cBody] = cvx; sh.vel[1][cBody] = cvy; sh.vel[2][cBody] = cw;
    }
    for (int e = WAVE + lane; e < NC; e += WAVE) { // overflow bodies: everything through scratch
        const int b = sh.clist[e];
        const unsigned gl = wb0 + (unsigned)b;
        const int nt = __float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + gl, 0)) & 0xff;
        const float mB = LF(L_INVM), iB = LF(L_INVI);
        float cvx = sh.vel[0][b], cvy = sh.vel[1][b], cw = sh.vel[2][b];
        for (int t = 0; t < nt; ++t) {
            ContactC c;
            cc_load(S, (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl, c);
            contact_warm_start(c, mB, iB, cvx, cvy, cw);
        }
        sh.vel[0][b] = cvx; sh.vel[1][b] = cvy; sh.vel[2][b] = cw;
    }
    lds_sync();
    for (int r = 0; r < nRounds; ++r) {
#pragma unroll
        for (int s = 0; s < V3_SETS; ++s) {
            if (jround[s] == r) {
                const int a = jA[s], b = jB[s];
                V2 vA = mk(sh.vel[0][a], sh.vel[1][a]);
                float wA = sh.vel[2][a];
                V2 vB = mk(sh.vel[0][b], sh.vel[1][b]);
                float wB = sh.vel[2][b];
                V2 Pw = mk(J[s].impX, J[s].impY);
                vA = vsub(vA, vscale(J[s].mA, Pw));
                wA -= J[s].iA * (vcross(J[s].rA, Pw) + J[s].motorImp + J[s].impZ);
                vB = vadd(vB, vscale(J[s].mB, Pw));
                wB += J[s].iB * (vcross(J[s].rB, Pw) + J[s].motorImp + J[s].impZ);
                sh.vel[0][a] = vA.x; sh.vel[1][a] = vA.y; sh.vel[2][a] = wA;
                sh.vel[0][b] = vB.x; sh.vel[1][b] = vB.y; sh.vel[2][b] = wB;
            }
        }
        lds_sync();
    }
    // ---------------- velocity iterations ----------------
    {
        int leftJ[V3_SETS];
#pragma unroll
        for (int s = 0; s < V3_SETS; ++s) leftJ[s] = (jround[s] >= 0) ? iters : 0;
        int nextC = (crole && iters > 0) ? offC : 0x7fffffff, leftC = iters;
        int ph = 0;
        for (int tick = 0; tick < nTicks; ++tick) {
            // joint slot: the register set of this tick's phase (wave-uniform switch)
#pragma unroll
            for (int s = 0; s < V3_SETS; ++s) {
                if (s == ph) {
                    if (leftJ[s] > 0 && tick >= jround[s]) { // jround[s] = phase s (mod P): fires every P ticks from there
                        --leftJ[s];
                        const int a = jA[s], b = jB[s];
                        V2 vA = mk(sh.vel[0][a], sh.vel[1][a]);
                        float wA = sh.vel[2][a];
                        V2 vB = mk(sh.vel[0][b], sh.vel[1][b]);
                        float wB = sh.vel[2][b];
                        joint_solve_velocity(J[s], vA, wA, vB, wB);
                        sh.vel[0][a] = vA.x; sh.vel[1][a] = vA.y; sh.vel[2][a] = wA;
                        sh.vel[0][b] = vB.x; sh.vel[1][b] = vB.y; sh.vel[2][b] = wB;
                    }
                }
            }
            lds_sync();
            if (P > 32 || ((cPhases >> ph) & 1)) {
                if (tick == nextC) {
                    nextC = (--leftC > 0) ? nextC + P : 0x7fffffff;
                    float cvx = sh.vel[0][cBody], cvy = sh.vel[1][cBody], cw = sh.vel[2][cBody];
#pragma unroll
                    for (int t = 0; t < KR; ++t)
                        if (t < nTouch) contact_solve(cc[t], cmB, ciB, mu, cvx, cvy, cw);
                    if (nTouch > KR) {
                        for (int t = KR; t < nTouch; ++t) {
                            const unsigned cb = (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + glC;
                            ContactC c;
                            cc_load(S, cb, c);
                            contact_solve(c, cmB, ciB, mu, cvx, cvy, cw);
                            SW(cb, 10) = c.n0; SW(cb, 11) = c.n1; SW(cb, 12) = c.t0; SW(cb, 13) = c.t1;
                        }
                    }
                    sh.vel[0][cBody] = cvx; sh.vel[1][cBody] = cvy; sh.vel[2][cBody] = cw;
                }
                if (NC > WAVE) {
                    for (int e = WAVE + lane; e < NC; e += WAVE) {
                        const int b = sh.clist[e];
                        const unsigned gl = wb0 + (unsigned)b;
                        const int oc = (LI(L_JROUND) >> 8) & 0xff;
                        const int d = tick - oc;
                        if (d < 0 || d % P != 0 || d / P >= iters) continue;
                        const int nt = __float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + gl, 0)) & 0xff;
                        const float mB = LF(L_INVM), iB = LF(L_INVI);
                        float cvx = sh.vel[0][b], cvy = sh.vel[1][b], cw = sh.vel[2][b];
                        for (int t = 0; t < nt; ++t) {
                            const unsigned cb = (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl;
                            ContactC c;
                            cc_load(S, cb, c);
                            contact_solve(c, mB, iB, mu, cvx, cvy, cw);
                            SW(cb, 10) = c.n0; SW(cb, 11) = c.n1; SW(cb, 12) = c.t0; SW(cb, 13) = c.t1;
                        }
                        sh.vel[0][b] = cvx; sh.vel[1][b] = cvy; sh.vel[2][b] = cw;
                    }
                }
                lds_sync();
            }
            ph = ph + 1 == P ? 0 : ph + 1;
        }
    }
    // ---------------- StoreImpulses, joint impulses, body velocities ----------------
    if (crole) {
#pragma unroll
        for (int t = 0; t < KR; ++t) {
            if (t < nTouch) {
                unsigned o = ((slotPack >> (5 * t)) & 0x1f) * Lp + glC;
                CF(C_N0, o) = cc[t].n0;
                CF(C_T0, o) = cc[t].t0;
                if (cc[t].count > 1) {
                    CF(C_N1, o) = cc[t].n1;
                    CF(C_T1, o) = cc[t].t1;
                }
            }
        }
        for (int t = KR; t < nTouch; ++t) {
            const unsigned cb = (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + glC;
            unsigned o = ((slotPack >> (5 * t)) & 0x1f) * Lp + glC;
            CF(C_N0, o) = SW(cb, 10);
            CF(C_T0, o) = SW(cb, 12);
            if (__float_as_int(SW(cb, 20)) > 1) {
                CF(C_N1, o) = SW(cb, 11);
                CF(C_T1, o) = SW(cb, 13);
            }
        }
    }
    for (int e = WAVE + lane; e < NC; e += WAVE) {
        const int b = sh.clist[e];
        const unsigned gl = wb0 + (unsigned)b;
        const unsigned mb = (unsigned)SCR_MISC_BASE * Lp + gl;
        const int nt = __float_as_int(SW(mb, 0)) & 0xff;
        const unsigned sp = (unsigned)__float_as_int(SW(mb, 1));
        for (int t = 0; t < nt; ++t) {
            const unsigned cb = (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl;
            unsigned o = ((sp >> (5 * t)) & 0x1f) * Lp + gl;
            CF(C_N0, o) = SW(cb, 10);
            CF(C_T0, o) = SW(cb, 12);
            if (__float_as_int(SW(cb, 20)) > 1) {
                CF(C_N1, o) = SW(cb, 11);
                CF(C_T1, o) = SW(cb, 13);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < V3_SETS; ++s) {
        if (jround[s] >= 0) {
            const unsigned gl = wb0 + (unsigned)jB[s];
            LF(L_JIMPX) = J[s].impX; LF(L_JIMPY) = J[s].impY; LF(L_JIMPZ) = J[s].impZ; LF(L_JMOTORIMP) = J[s].motorImp;
        }
    }
#pragma unroll
    for (int p = 0; p < V3_PASSES; ++p) {
        const int bl = p * WAVE + lane;
        if (misc[p] & 0x100) {
            const unsigned gl = wb0 + (unsigned)bl;
            LF(L_VX) = sh.vel[0][bl]; LF(L_VY) = sh.vel[1][bl]; LF(L_W) = sh.vel[2][bl];
        }
    }
    err = wave_or(err);
    if (err && lane == 0) {
        const unsigned env = wb0 / (unsigned)K;
        atomicOr(&EI(E_ERR), err);
    }
}

__global__ __launch_bounds__(WAVE, 2) void rem2d_vel3_kernel(State S, Vel3Args A) {
    __shared__ Vel3Shared sh;
    vel3_body(S, A, blockIdx.x, sh);
}

#endif
