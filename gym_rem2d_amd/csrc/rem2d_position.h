// rem2d_position.h -- software-pipelined position iterations of b2Island::Solve (shared by rem2d_step_multi_kernel
// and rem2d_post_multi_kernel).  Part of the single translation unit rem2d.hip; not a stand-alone header.
#ifndef REM2D_POSITION_H
#define REM2D_POSITION_H

// ---------------------------------------------------------------------------------------------------
// b2Island::Solve position iterations for the K-lane creatures of one wavefront, software-pipelined.
//
// Box2D: for it < posIters { contacts (per body, list order); joints (island order); if all within
// tolerance: break }.  The joints of one iteration form a chain of `rounds` dependent steps, but joint k
// of iteration n+1 only has to wait for the operations of iteration n that share one of its two bodies,
// so iteration n+1 can start `period` ticks after iteration n (same modulo schedule as the velocity
// iterations; one tick = a contact slot, then a joint slot; body b's contacts fire in a tick of its window
// between its last joint of the previous iteration and its first joint of this one).  The catch is the exit test: whether iteration n was the last one is only known when its
// last joint has run, by which time early joints of iterations n+1.. have already moved bodies.  Every
// body therefore snapshots its position after its last operation of each iteration into a ring in LDS;
// when iteration n passes the test the creature's bodies are rolled back to snapshot n -- exactly the
// state Box2D leaves.  The effective period is raised so that at most POS_RING iterations are in flight.
// A 60-iteration creature costs 60*period ticks instead of 60*rounds rounds (3x fewer for 16 modules).
// ---------------------------------------------------------------------------------------------------
#define POS_RING 8
static_assert(POS_RING <= 32, "the failure bits of the iterations in flight live in one 32-bit word");
// max { y : sqrtf(y) <= b2_linearSlop (0.005f) } (tests/test_host_golden.py checks the literal against numpy's sqrt)
#define POS_SLOP_SQ_MAX 0x1.a36e3p-16f
struct __attribute__((aligned(16))) PosRec { float x, y, a, pad; }; // one body's position: one 16-byte LDS access
struct PosShared {
    union {
        float mbox[3][WAVE];  // (the fused step kernel's velocity mailbox during its velocity iterations)
        PosRec pos[WAVE];     // the position solver's mailbox
    };
    PosRec snap[POS_RING][WAVE];
    int firstR[WAVE], lastR[WAVE];
};
// manifold geometry as the position solver reads it (b2PositionSolverManifold): type | count << 8, local normal,
// local point, two manifold points
#define POS_KR 3
struct PosManifold { int tc; V2 ln, lp, p0, p1; };
DEV void pos_manifold_load(const State &S, unsigned gl, int t, PosManifold &m) {
    const unsigned sb = (unsigned)(t * SCR_WORDS) * S.Lp + gl;
    m.tc = __float_as_int(SW(sb, 0));
    m.ln = mk(SW(sb, 1), SW(sb, 2)); m.lp = mk(SW(sb, 3), SW(sb, 4));
    m.p0 = mk(SW(sb, 5), SW(sb, 6)); m.p1 = mk(SW(sb, 7), SW(sb, 8));
}
// b2ContactSolver::SolvePositionConstraints for one manifold between the static ground (A) and this body (B)
DEV void pos_solve_manifold(const PosManifold &m, float mB, float iB, float radiusB, float &cx, float &cy, float &ca,
                            float &minSeparation) {
    const int mtype = m.tc & 0xff, mcount = m.tc >> 8;
    const float radiusA = B2_POLYGON_RADIUS;
    for (int j = 0; j < mcount; ++j) {
        const V2 pj = j == 0 ? m.p0 : m.p1;
        V2 cB = mk(cx, cy);
        V2 normal, point;
        float separation;
        Rot qB = rot_set(ca);
        if (mtype == MF_CIRCLES) {
            V2 pointA = m.lp;
            V2 pointB = xmul(qB, cB, m.p0);
            normal = vsub(pointB, pointA);
            vnormalize(normal);
            point = vscale(0.5f, vadd(pointA, pointB));
            separation = vdot(vsub(pointB, pointA), normal) - radiusA - radiusB;
        } else if (mtype == MF_FACE_A) {
            normal = m.ln;
            V2 planePoint = m.lp;
            V2 clipPoint = xmul(qB, cB, pj);
            separation = vdot(vsub(clipPoint, planePoint), normal) - radiusA - radiusB;
            point = clipPoint;
        } else {
            normal = rmul(qB, m.ln);
            V2 planePoint = xmul(qB, cB, m.lp);
            V2 clipPoint = pj;
            separation = vdot(vsub(clipPoint, planePoint), normal) - radiusA - radiusB;
            point = clipPoint;
            normal = vneg(normal);
        }
        V2 rBp = vsub(point, cB);
        minSeparation = fmin32(minSeparation, separation);
        float C = fclamp(B2_BAUMGARTE * (separation + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
        float rnB = vcross(rBp, normal);
        float Kn = mB + iB * rnB * rnB;
        float impulse = Kn > 0.0f ? -C / Kn : 0.0f;
        V2 Pc = vscale(impulse, normal);
        cx = cx + mB * Pc.x;
        cy = cy + mB * Pc.y;
        ca += iB * vcross(rBp, Pc);
    }
}
template <int K> DEV int group_max(int v) {
#pragma unroll
    for (int o = 1; o < K; o <<= 1) {
        int t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}
template <int K>
DEV void solve_positions_pipelined(const State &S, PosShared &sh, unsigned gl, int lane, int pl, bool active, bool hasJoint,
                                   int jround, int period, int nTouch, float mA, float iA, float mB, float iB, float radiusB,
                                   int limitState, float motorMass, int posIters, float &px, float &py, float &ang,
                                   bool &envSolved, int &itersUsed) {
    const V2 anchorA = mk(LF(L_JAX), LF(L_JAY)), anchorB = mk(LF(L_JBX), LF(L_JBY));
    const float jLower = LF(L_JLOWER), jUpper = LF(L_JUPPER);
    // the limit case of this lane's joint as constants of the iteration loop (see the joint slot)
    const float limRef = limitState == LIM_AT_UPPER ? jUpper : jLower;
    const float limOff = limitState == LIM_AT_LOWER ? B2_ANGULAR_SLOP : (limitState == LIM_AT_UPPER ? -B2_ANGULAR_SLOP : 0.0f);
    const float limLo = limitState == LIM_AT_UPPER ? 0.0f : -B2_MAX_ANGULAR_CORRECTION;
    const float limHi = limitState == LIM_AT_LOWER ? 0.0f : B2_MAX_ANGULAR_CORRECTION;
    PosManifold pm[POS_KR];
#pragma unroll
    for (int t = 0; t < POS_KR; ++t) {
        pm[t].tc = 0; pm[t].ln = pm[t].lp = pm[t].p0 = pm[t].p1 = mk(0.0f, 0.0f);
        if (active && t < nTouch) pos_manifold_load(S, gl, t, pm[t]);
    }
    // first / last joint round of every body (its own joint and those of its children)
    sh.firstR[lane] = 0x7fffffff;
    sh.lastR[lane] = -1;
    lds_sync();
    if (hasJoint) {
        atomicMin(&sh.firstR[lane], jround); atomicMax(&sh.lastR[lane], jround);
        atomicMin(&sh.firstR[pl], jround); atomicMax(&sh.lastR[pl], jround);
    }
    lds_sync();
    const int lastB = sh.lastR[lane];
    const bool anyJoint = lastB >= 0;
    const bool isLastA = hasJoint && jround == sh.lastR[pl], isLastB = hasJoint && jround == lastB;
    const int maxR = group_max<K>(hasJoint ? jround : 0);
    const int P = max(group_max<K>(period), maxR / POS_RING + 1); // >= 1; keeps <= POS_RING iterations in flight
    const bool touching = active && nTouch > 0;
    const bool hasOps = active && (anyJoint || nTouch > 0);
    const bool run = posIters > 0;
    // Contact slot of body b in iteration n: any tick after its last joint of iteration n - 1 and not after its first
    // joint of iteration n (the contact section of a tick runs before its joint section), i.e. offC + n P with offC in
    // [lastR + 1 - P, firstR].  The section costs the wavefront the same whether one lane or all are due, so (1) the
    // touching bodies of a creature are steered to one common phase, the one inside the window of the most of them (a
    // body whose window misses it keeps the latest tick, its first joint's), and (2) -- creatures are independent -- every
    // creature runs its whole schedule `rot` ticks late so that this phase is phase 0 of the wavefront.  Iteration 0 has no
    // earlier joints: max(offC, 0).
    const int hiC = anyJoint ? sh.firstR[lane] : 0, loC = anyJoint ? lastB + 1 - P : 1 - P;
    const unsigned long long groupLanes = (K == WAVE ? ~0ull : ((1ull << (K & 63)) - 1ull)) << (lane & ~(K - 1));
    int offC = hiC, rot = 0;
    {
        const int Pw = wave_max_active(P, 8); // (this function runs under `if (envAwake)`: not every lane is here)
        int best = 0, bestPhase = 0;
        for (int c = 0; c < Pw; ++c) { // (per creature: the ballots are masked with its lanes)
            int d = (c - loC) % P;
            d = d < 0 ? d + P : d;
            const int n = __popcll(__ballot(touching && c < P && d <= hiC - loC) & groupLanes);
            if (n > best) { best = n; bestPhase = c; }
        }
        int d = (bestPhase - loC) % P;
        d = d < 0 ? d + P : d;
        if (d <= hiC - loC) offC = loC + d;
        rot = best > 0 ? (P - bestPhase) % P : 0;
    }
    offC += rot;
    int nextC = (run && touching) ? (offC > 0 ? offC : 0) : 0x7fffffff, leftC = posIters, itC = 0;
    int nextJ = (run && hasJoint) ? jround + rot : 0x7fffffff, leftJ = posIters, itJ = 0;
    int nextD = run ? maxR + rot : 0x7fffffff, itD = 0;
    unsigned failBits = 0u; // a ring of POS_RING (<= 32) iterations in flight: one 32-bit word
    const int lastTick = wave_max_active(run ? maxR + rot + (posIters - 1) * P + 1 : 0, 16) - 1;
    { PosRec r0; r0.x = px; r0.y = py; r0.a = ang; r0.pad = 0.0f; sh.pos[lane] = r0; }
    lds_sync();
#ifdef REM2D_POS_STAMPS // diagnostic build (tools/pos_stamps_probe.py): where the cycles of a wavefront's tick loop go
    unsigned long long pT0 = __builtin_amdgcn_s_memtime(), pTa = pT0, pC = 0, pJ = 0, pV = 0;
    int pTicks = 0, pCsec = 0, pJsec = 0;
#define POS_STAMP(acc) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - pTa; pTa = t_; }
#else
#define POS_STAMP(acc)
#endif
    for (int tick = 0; tick <= lastTick; ++tick) {
#ifdef REM2D_POS_STAMPS
        pTicks += 1; pCsec += __any(tick == nextC ? 1 : 0) ? 1 : 0; pJsec += __any(tick == nextJ ? 1 : 0) ? 1 : 0;
        pTa = __builtin_amdgcn_s_memtime();
#endif
        // ---- contact slot: b2ContactSolver::SolvePositionConstraints for this body's manifolds ----
        if (tick == nextC) {
            const PosRec mine = sh.pos[lane];
            float cx = mine.x, cy = mine.y, ca = mine.a;
            float minSeparation = 0.0f;
#pragma unroll
            for (int t = 0; t < POS_KR; ++t) // the first manifolds of the body: geometry preloaded into registers
                if (t < nTouch) pos_solve_manifold(pm[t], mB, iB, radiusB, cx, cy, ca, minSeparation);
            for (int t = POS_KR; t < nTouch; ++t) { // further ones (rare): re-read from scratch in every iteration
                PosManifold m;
                pos_manifold_load(S, gl, t, m);
                pos_solve_manifold(m, mB, iB, radiusB, cx, cy, ca, minSeparation);
            }
            if (!(minSeparation >= -3.0f * B2_LINEAR_SLOP)) failBits |= 1u << (itC & 31);
            PosRec out; out.x = cx; out.y = cy; out.a = ca; out.pad = 0.0f;
            sh.pos[lane] = out;
            if (!anyJoint) { // the contact slot is this body's last operation of the iteration
                const int r = itC & (POS_RING - 1);
                sh.snap[r][lane] = out;
            }
            ++itC;
            nextC = (--leftC > 0) ? offC + itC * P : 0x7fffffff;
        }
        lds_sync();
        POS_STAMP(pC)
        // ---- joint slot: b2RevoluteJoint::SolvePositionConstraints ----
        if (tick == nextJ) {
            const PosRec recA = sh.pos[pl], recB = sh.pos[lane];
            V2 cA = mk(recA.x, recA.y);
            float aA = recA.a;
            V2 cB = mk(recB.x, recB.y);
            float aB = recB.a;
            float angularError = 0.0f, positionError = 0.0f;
            if (limitState != LIM_INACTIVE) {
                // b2RevoluteJoint::SolvePositionConstraints' three limit cases as ONE instruction sequence (a wavefront
                // usually holds joints at their lower and at their upper limit and would run the cases one after the other):
                //   equal:  C = clamp(angle - lower,        -maxCorr, maxCorr), error = |C|
                //   lower:  C = clamp(angle - lower + slop, -maxCorr, 0),       error = -(angle - lower)
                //   upper:  C = clamp(angle - upper - slop,  0, maxCorr),       error =   angle - upper
                // with the per-joint constants (reference angle, slop term, clamp bounds) picked before the loop.  Same
                // operations on the same operands; the equal case adds a literal 0 (x + 0 is x bar the sign of a zero).
                float angle = aB - aA - 0.0f;
                const float C0 = angle - limRef;
                const float C = fclamp(C0 + limOff, limLo, limHi);
                const float limitImpulse = -motorMass * C;
                angularError = limitState == LIM_EQUAL ? fabs32(C) : (limitState == LIM_AT_LOWER ? -C0 : C0);
                aA -= iA * limitImpulse;
                aB += iB * limitImpulse;
            }
            {
                Rot qA = rot_set(aA), qB = rot_set(aB);
                V2 prA = rmul(qA, vsub(anchorA, mk(0.0f, 0.0f)));
                V2 prB = rmul(qB, vsub(anchorB, mk(0.0f, 0.0f)));
                V2 C = vsub(vsub(vadd(cB, prB), cA), prA);
                // (only compared with b2_linearSlop below: |C| <= slop <=> C.C <= the largest binary32 whose correctly rounded
                // square root is <= slop -- sqrt is monotone -- which spares the square root's ten instructions)
                positionError = vdot(C, C);
                float Kexx = mA + mB + iA * prA.y * prA.y + iB * prB.y * prB.y;
                float Kexy = -iA * prA.x * prA.y - iB * prB.x * prB.y;
                float Keyy = mA + mB + iA * prA.x * prA.x + iB * prB.x * prB.x;
                float det = Kexx * Keyy - Kexy * Kexy;
                if (det != 0.0f) det = 1.0f / det;
                V2 sol = mk(det * (Keyy * C.x - Kexy * C.y), det * (Kexx * C.y - Kexy * C.x));
                V2 impulse = vneg(sol);
                cA = vsub(cA, vscale(mA, impulse));
                aA -= iA * vcross(prA, impulse);
                cB = vadd(cB, vscale(mB, impulse));
                aB += iB * vcross(prB, impulse);
            }
            PosRec outA, outB;
            outA.x = cA.x; outA.y = cA.y; outA.a = aA; outA.pad = 0.0f;
            outB.x = cB.x; outB.y = cB.y; outB.a = aB; outB.pad = 0.0f;
            sh.pos[pl] = outA;
            sh.pos[lane] = outB;
            if (!(positionError <= POS_SLOP_SQ_MAX && angularError <= B2_ANGULAR_SLOP)) failBits |= 1u << (itJ & 31);
            const int r = itJ & (POS_RING - 1);
            if (isLastA) sh.snap[r][pl] = outA;
            if (isLastB) sh.snap[r][lane] = outB;
            ++itJ;
            nextJ = (--leftJ > 0) ? nextJ + P : 0x7fffffff;
        }
        lds_sync();
        POS_STAMP(pJ)
        // ---- verdict on the iteration whose last joint has just run (creature-uniform) ----
        const bool due = tick == nextD;
        if (__any(due ? 1 : 0)) { // (a wavefront whose only unfinished creature has period P is due once in P ticks)
            // (one ballot and the creature's lane mask instead of a K-lane shuffle reduction in every tick)
            const unsigned long long failedLanes = __ballot(due && ((failBits >> (itD & 31)) & 1u) != 0u);
            const unsigned long long groupLanes = (K == WAVE ? ~0ull : ((1ull << (K & 63)) - 1ull)) << (lane & ~(K - 1));
            const bool failed = (failedLanes & groupLanes) != 0ull;
            if (due) failBits &= ~(1u << (itD & 31)); // the mask is a ring: at most POS_RING iterations are in flight
            bool restored = false;
            if (due) {
                if (!failed) { // Box2D breaks here: drop whatever later iterations have already done
                    envSolved = true;
                    itersUsed = itD + 1;
                    nextC = nextJ = nextD = 0x7fffffff;
                    if (hasOps) {
                        const int r = itD & (POS_RING - 1);
                        sh.pos[lane] = sh.snap[r][lane];
                        restored = true;
                    }
                } else {
                    ++itD;
                    nextD = itD < posIters ? nextD + P : 0x7fffffff;
                }
            }
            if (__any(restored ? 1 : 0)) lds_sync();
            if (__all(nextD == 0x7fffffff ? 1 : 0)) break;
        }
        POS_STAMP(pV)
    }
#ifdef REM2D_POS_STAMPS
    if (lane == 0 && 4 * K <= WAVE) { // per block (tools/pos_blocks_probe.py): in the TOI event counters of its first four creatures
        unsigned env = gl / K;
        EI(E_TOIEVENTS) = (int)((__builtin_amdgcn_s_memtime() - pT0) >> 6);
        env += 1; EI(E_TOIEVENTS) = pTicks | (pCsec << 16);
        env += 1; EI(E_TOIEVENTS) = (int)(pC >> 6);
        env += 1; EI(E_TOIEVENTS) = (int)(pJ >> 6);
    }
    if (lane == 0 && pTicks >= 40 * P) { // the wavefronts that iterate to the end only
        const unsigned long long all = __builtin_amdgcn_s_memtime() - pT0;
        atomicAdd(&S.toiWork[2], (int)(pC >> 6)); atomicAdd(&S.toiWork[3], (int)(pJ >> 6)); atomicAdd(&S.toiWork[4], (int)(pV >> 6));
        atomicAdd(&S.toiWork[5], (int)(all >> 6)); atomicAdd(&S.toiWork[6], pTicks); atomicAdd(&S.toiWork[7], pCsec);
        atomicAdd(&S.toiWork[8], pJsec); atomicAdd(&S.toiWork[9], 1); atomicMax(&S.toiWork[10], (int)(all >> 6));
    }
#endif
    { const PosRec fin = sh.pos[lane]; px = fin.x; py = fin.y; ang = fin.a; }
}

#endif
