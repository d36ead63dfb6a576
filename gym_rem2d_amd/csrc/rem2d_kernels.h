// rem2d_kernels.h -- rem2d_step_multi_kernel, rem2d_toi_kernel, rem2d_reset_kernel.
// Part of the single translation unit rem2d.hip (see its header comment); not a stand-alone header.
#ifndef REM2D_KERNELS_H
#define REM2D_KERNELS_H

// wod / reward / done (Modular2DEnv.py:613-614,642-649) and evaluate()'s fitness rule
// (REM2D_main.py:362-377); executed by one lane per creature, once per env step.
DEV void env_bookkeeping(const State &S, unsigned env, int sub, float rootx) {
    if (sub != 0) return;
    double wod = ED(E_WOD) + 0.04;
    ED(E_WOD) = wod;
    double r = (double)rootx;
    double rew = r;
    int d = 0;
    if (r < 0.0) { rew = -100.0; d = 1; }
    if (wod > r) { rew = -100.0; d = 1; }
    EF(E_REWARD) = (float)rew;
    EI(E_DONE) = d;
    if (S.outIndex && env < S.nEnvs) { // population-order outputs: no gather kernels on the host side
        const int g = S.outIndex[env];
        S.outReward[g] = (float)rew;
        S.outDone[g] = (unsigned char)d;
    }
    if (d) EI(E_EVERDONE) = 1;
    int stepIdx = EI(E_STEPS);
    if (!EI(E_FROZEN)) {
        if (rew < -10.0) EI(E_FROZEN) = 1;
        else if (rew > 100.0) { ED(E_FITNESS) = rew + (double)(10000 - stepIdx) / 10000.0; EI(E_FROZEN) = 1; }
        else if (rew > 0.0) ED(E_FITNESS) = rew;
    }
    EI(E_STEPS) = stepIdx + 1;
}

// =====================================================================================
// the step kernel
// =====================================================================================
struct StepArgs { int nSteps; float dt; int velIters, posIters; int defer; /* TOI kernel finishes the step */ int heavyPerWave; int prio; /* rem2d.hip: prio_mode */ };

// Register budget: what the 180-iteration velocity loop touches stays in VGPRs (body velocity,
// joint effective-mass terms and impulses, KR contact constraints); everything else (pose
// history, shape, fat AABB, anchors, controller, per-creature bookkeeping) is re-read from
// HBM/L2 at its point of use once per step, so that the kernel fits two waves per SIMD.
// Two waves per SIMD (238 VGPRs, no spills); a 3-waves build (168 VGPRs) spilled ~95 registers and was not faster.
template <int K>
DEV void step_body(const State &S, const Terrain &T, const StepArgs &A, unsigned block, PosShared &psh) {
    // psh.mbox: velocity mailbox of the joint slots; the rest of psh: position solver
    const int lane = threadIdx.x;
    const unsigned gl0 = block * WAVE + lane;
    const unsigned env0 = gl0 / K;
    if (gl0 == 0) S.toiWork[0] = 0; // work list of the TOI kernels that follow
    if (S.flags & REM2D_FLAG_SKIP_FROZEN) { // evaluate() has left its loop for every creature of this wavefront
        const unsigned env = env0;
        if (__all(EI(E_FROZEN) != 0 ? 1 : 0)) return;
    }
    unsigned gl = gl0, env = env0;
    const int base = lane & ~(K - 1);
    const int sub = lane & (K - 1);
    const unsigned Lp = S.Lp;
    // scratch: manifolds [KT][SCR_WORDS][Lp], then overflow constraints [KT-KR][CC_WORDS][Lp]

    const int shape = LI(L_SHAPE);
    const bool active = shape != SHAPE_NONE;
    const float mB = LF(L_INVM), iB = LF(L_INVI);
    float px = LF(L_PX), py = LF(L_PY), ang = LF(L_ANG), vx = LF(L_VX), vy = LF(L_VY), w = LF(L_W);
    float sleepT = LF(L_SLEEPT);
    int awake = LI(L_AWAKE);
    int cCount = LI(L_CCOUNT);
    const int parent = LI(L_PARENT);
    const bool hasJoint = active && parent >= 0;
    const int pl = base + (parent >= 0 ? parent : 0);
    // packed schedule (compiler.pipeline_schedule): joint round | contact slot << 8 | period << 16
    const int sched = LI(L_JROUND);
    const int jround = hasJoint ? (sched & 0xff) : -1;
    const int offC = (sched >> 8) & 0xff, period = (sched >> 16) & 0xff;
    float impX = LF(L_JIMPX), impY = LF(L_JIMPY), impZ = LF(L_JIMPZ), motorImp = LF(L_JMOTORIMP);
    int limitState = LI(L_JLIMIT);
    float motorSpeed = LF(L_JMOTORSPEED);
    const float mA = __shfl(mB, pl), iA = __shfl(iB, pl);
    const int nRounds = wave_max(jround) + 1;
    // bit b of childMask: some lane of this creature hangs off body b (K <= 64 -> two 32-bit halves)
    const int childLo = group_or<K>((hasJoint && parent < 32) ? (1 << parent) : 0);
    const int childHi = group_or<K>((hasJoint && parent >= 32) ? (1 << (parent - 32)) : 0);
    const bool jointed = hasJoint || (((sub < 32 ? childLo >> sub : childHi >> (sub - 32))) & 1); // SetMotorSpeed wakes both bodies
    float invDt0 = EF(E_INVDT0);
    int newFix = EI(E_NEWFIX), err = 0, lastPosIters = EI(E_POSITERS);

    const float h = A.dt;
    const float inv_dt = h > 0.0f ? 1.0f / h : 0.0f;
    const float friction = T.friction;
    const bool sleepResetAlways = (S.flags & REM2D_FLAG_SLEEP_RESET_ALWAYS) != 0;
    const bool allowSleep = (S.flags & REM2D_FLAG_NO_SLEEP) == 0;

    for (int step = 0; step < A.nSteps; ++step) {
        // Launder the lane / creature offsets once per step: otherwise LLVM hoists every field's
        // 64-bit address out of the step loop (60+ VGPR pairs) and then spills them.  With the
        // offsets opaque per step, accesses select as global_load/store vdst, voffset, saddr.
        asm volatile("" : "+v"(gl), "+v"(env));
        // =============== Modular2D.step: controllers, PID -> motorSpeed ===============
        {
            float angParent = __shfl(ang, pl);
            if (hasJoint) {
                double ist = LD(D_CISTATE) + LD(D_CFREQ);
                LD(D_CISTATE) = ist;
                double target = (LD(D_CAMP) * dev_sin(ist + LD(D_CPHASE))) + LD(D_COFFSET);
                float jointAngle = ang - angParent - 0.0f;
                double speed = (target - (double)jointAngle) * 1.9;
                motorSpeed = (float)speed;
            }
            if (active && jointed) { // b2RevoluteJoint::SetMotorSpeed -> SetAwake(true) on both bodies
                if (sleepResetAlways || !awake) sleepT = 0.0f;
                awake = 1;
            }
        }
        // =============== b2World::Step ===============
        const float c0x = px, c0y = py, a0 = ang; // sweep start (b2Island::Solve: c0 = c, a0 = a)
        const float dtRatio = invDt0 * h;
        const float hx = LF(L_HX), hy = LF(L_HY);
        const float radiusB = shape == SHAPE_CIRCLE ? hx : B2_POLYGON_RADIUS;
        V2 fatLo = mk(LF(L_FATLX), LF(L_FATLY)), fatHi = mk(LF(L_FATUX), LF(L_FATUY));
        if (newFix) { // FindNewContacts for freshly created fixtures
            if (active && find_new_pairs(S, T, gl, cCount, fatLo, fatHi, err)) {
                if (sleepResetAlways || !awake) sleepT = 0.0f;
                awake = 1;
            }
            newFix = 0;
        }
        Rot q = rot_set(ang); // body transform m_xf (q from sweep.a, p = c since localCenter = 0)
        // ---- b2ContactManager::Collide: destroy separated pairs, update manifolds ----
        int nTouch = 0;
        slotpack_t slotPack = 0;
        if (active && awake) {
            int s = 0;
            while (s < cCount) {
                unsigned o = (unsigned)s * Lp + gl;
                int e = CI(C_EDGE, o);
                if (!aabb_overlap(mk(T.flx[e], T.fly[e]), mk(T.fux[e], T.fuy[e]), fatLo, fatHi)) {
                    // b2ContactManager::Destroy wakes the bodies of a touching contact
                    if ((CI(C_INFO, o) & 0xff) > 0 && sleepResetAlways) sleepT = 0.0f;
                    pairs_remove(S, gl, cCount, s);
                    continue;
                }
                Manifold m;
                contact_update_slot(S, T, o, shape, hx, hy, mk(px, py), q, m, sleepResetAlways, sleepT);
                if (m.count > 0) {
                    if (nTouch < KT) {
                        manifold_store(S, gl, nTouch, m);
                        slotPack |= SP_PUT(s, nTouch);
                        ++nTouch;
                    } else {
                        err |= REM2D_ERR_SOLVER_OVERFLOW;
                    }
                }
                ++s;
            }
        }
        // =============== b2World::Solve ===============
        // the creature is one island; it is simulated iff any of its bodies is awake
        const int envAwake = group_or<K>(active && awake ? 1 : 0);
        if (envAwake) {
            if (active && (!awake || sleepResetAlways)) { awake = 1; sleepT = 0.0f; } // island.Add -> SetAwake(true)
            // ---- integrate velocities (gravity (0,-10), no forces, no damping) ----
            if (active) {
                V2 acc = vadd(vscale(1.0f, mk(0.0f, -10.0f)), vscale(mB, mk(0.0f, 0.0f)));
                V2 v = vadd(mk(vx, vy), vscale(h, acc));
                float wz = w + h * iB * 0.0f;
                v = vscale(1.0f / (1.0f + h * 0.0f), v);
                wz *= 1.0f / (1.0f + h * 0.0f);
                vx = v.x; vy = v.y; w = wz;
            }
            // ---- contact constraints: b2ContactSolver ctor + InitializeVelocityConstraints + WarmStart ----
            ContactC cc[KR];
            const bool anyOverflow = __any(nTouch > KR ? 1 : 0);
#pragma unroll
            for (int t = 0; t < KR; ++t) {
                cc[t].count = 0;
                if (t < nTouch) {
                    const unsigned sb = (unsigned)(t * SCR_WORDS) * S.Lp + gl;
                    int tc = __float_as_int(SW(sb, 0));
                    unsigned o = SP_GET(slotPack, t) * Lp + gl;
                    contact_setup(cc[t], tc & 0xff, tc >> 8, mk(SW(sb, 1), SW(sb, 2)), mk(SW(sb, 3), SW(sb, 4)),
                                  mk(SW(sb, 5), SW(sb, 6)), mk(SW(sb, 7), SW(sb, 8)), mk(px, py), q, mB, iB, radiusB,
                                  dtRatio * CF(C_N0, o), dtRatio * CF(C_T0, o), dtRatio * CF(C_N1, o), dtRatio * CF(C_T1, o));
                    contact_warm_start(cc[t], mB, iB, vx, vy, w);
                }
            }
            if (anyOverflow) {
                for (int t = KR; t < nTouch; ++t) {
                    const unsigned sb = (unsigned)(t * SCR_WORDS) * S.Lp + gl;
                    int tc = __float_as_int(SW(sb, 0));
                    unsigned o = SP_GET(slotPack, t) * Lp + gl;
                    ContactC c;
                    contact_setup(c, tc & 0xff, tc >> 8, mk(SW(sb, 1), SW(sb, 2)), mk(SW(sb, 3), SW(sb, 4)),
                                  mk(SW(sb, 5), SW(sb, 6)), mk(SW(sb, 7), SW(sb, 8)), mk(px, py), q, mB, iB, radiusB,
                                  dtRatio * CF(C_N0, o), dtRatio * CF(C_T0, o), dtRatio * CF(C_N1, o), dtRatio * CF(C_T1, o));
                    contact_warm_start(c, mB, iB, vx, vy, w);
                    cc_store(S, (unsigned)(KT * SCR_WORDS + (t - KR) * CC_WORDS) * S.Lp + gl, c);
                }
            }
            // ---- joints: InitVelocityConstraints (warm start), in island rounds ----
            V2 rA = mk(0.0f, 0.0f), rB = mk(0.0f, 0.0f);
            float m_exx = 0.0f, m_eyx = 0.0f, m_ezx = 0.0f, m_eyy = 0.0f, m_ezy = 0.0f, m_ezz = 0.0f, motorMass = 0.0f;
            float det33 = 0.0f, det22 = 0.0f, cyz_x = 0.0f, cyz_y = 0.0f, cyz_z = 0.0f;
            float maxMotorImpulse = 0.0f;
            {
                float sA = __shfl(q.s, pl), cA = __shfl(q.c, pl);
                float aA = __shfl(ang, pl);
                if (hasJoint) {
                    const V2 anchorA = mk(LF(L_JAX), LF(L_JAY)), anchorB = mk(LF(L_JBX), LF(L_JBY));
                    const float jLower = LF(L_JLOWER), jUpper = LF(L_JUPPER);
                    maxMotorImpulse = h * LF(L_JTORQUE);
                    Rot qA; qA.s = sA; qA.c = cA;
                    rA = rmul(qA, vsub(anchorA, mk(0.0f, 0.0f)));
                    rB = rmul(q, vsub(anchorB, mk(0.0f, 0.0f)));
                    m_exx = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
                    m_eyx = -rA.y * rA.x * iA - rB.y * rB.x * iB;
                    m_ezx = -rA.y * iA - rB.y * iB;
                    m_eyy = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
                    m_ezy = rA.x * iA + rB.x * iB;
                    m_ezz = iA + iB;
                    motorMass = iA + iB;
                    if (motorMass > 0.0f) motorMass = 1.0f / motorMass;
                    float jointAngle = ang - aA - 0.0f;
                    if (fabs32(jUpper - jLower) < 2.0f * B2_ANGULAR_SLOP) {
                        limitState = LIM_EQUAL;
                    } else if (jointAngle <= jLower) {
                        if (limitState != LIM_AT_LOWER) impZ = 0.0f;
                        limitState = LIM_AT_LOWER;
                    } else if (jointAngle >= jUpper) {
                        if (limitState != LIM_AT_UPPER) impZ = 0.0f;
                        limitState = LIM_AT_UPPER;
                    } else {
                        limitState = LIM_INACTIVE;
                        impZ = 0.0f;
                    }
                    impX *= dtRatio; impY *= dtRatio; impZ *= dtRatio; motorImp *= dtRatio;
                    // loop invariants of b2Mat33::Solve33 / Solve22 (same expressions, evaluated once)
                    // ex = (m_exx, m_eyx, m_ezx), ey = (m_eyx, m_eyy, m_ezy), ez = (m_ezx, m_ezy, m_ezz)
                    cyz_x = m_eyy * m_ezz - m_ezy * m_ezy;
                    cyz_y = m_ezy * m_ezx - m_eyx * m_ezz;
                    cyz_z = m_eyx * m_ezy - m_eyy * m_ezx;
                    det33 = m_exx * cyz_x + m_eyx * cyz_y + m_ezx * cyz_z;
                    if (det33 != 0.0f) det33 = 1.0f / det33;
                    det22 = m_exx * m_eyy - m_eyx * m_eyx;
                    if (det22 != 0.0f) det22 = 1.0f / det22;
                }
            }
            if (nRounds > 0) {
                psh.mbox[0][lane] = vx; psh.mbox[1][lane] = vy; psh.mbox[2][lane] = w;
                lds_sync();
                for (int r = 0; r < nRounds; ++r) {
                    if (jround == r) {
                        V2 vA = mk(psh.mbox[0][pl], psh.mbox[1][pl]);
                        float wA = psh.mbox[2][pl];
                        V2 vB = mk(psh.mbox[0][lane], psh.mbox[1][lane]);
                        float wB = psh.mbox[2][lane];
                        V2 P = mk(impX, impY);
                        vA = vsub(vA, vscale(mA, P));
                        wA -= iA * (vcross(rA, P) + motorImp + impZ);
                        vB = vadd(vB, vscale(mB, P));
                        wB += iB * (vcross(rB, P) + motorImp + impZ);
                        psh.mbox[0][pl] = vA.x; psh.mbox[1][pl] = vA.y; psh.mbox[2][pl] = wA;
                        psh.mbox[0][lane] = vB.x; psh.mbox[1][lane] = vB.y; psh.mbox[2][lane] = wB;
                    }
                    lds_sync();
                }
                vx = psh.mbox[0][lane]; vy = psh.mbox[1][lane]; w = psh.mbox[2][lane];
            }
            // ---- velocity iterations, software-pipelined across iterations ----
            // One tick = a joint slot then a contact slot.  Joint (parent, this body) of iteration t fires
            // at tick jround + t*period, this body's contacts at offC + t*period; the host proves that any
            // two operations sharing a body keep Box2D's sequential order, so the result is bit-identical
            // to "for it: all joints in island order; all contacts" while a chain of J joints costs
            // `period` (2..4) slots per iteration instead of J.  Velocities live in the LDS mailbox.
            {
                const int iters = A.velIters;
                psh.mbox[0][lane] = vx; psh.mbox[1][lane] = vy; psh.mbox[2][lane] = w;
                lds_sync();
                int nextJ = (hasJoint && iters > 0) ? jround : 0x7fffffff, leftJ = iters;
                int nextC = (active && nTouch > 0 && iters > 0) ? offC : 0x7fffffff, leftC = iters;
                const int nTicks = wave_max_active((active && iters > 0) ? offC + (iters - 1) * period + 1 : 0, 16); // (under `if (envAwake)`)
                for (int tick = 0; tick < nTicks; ++tick) {
                    if (tick == nextJ) {
                        nextJ = (--leftJ > 0) ? nextJ + period : 0x7fffffff;
                        V2 vA = mk(psh.mbox[0][pl], psh.mbox[1][pl]);
                        float wA = psh.mbox[2][pl];
                        V2 vB = mk(psh.mbox[0][lane], psh.mbox[1][lane]);
                        float wB = psh.mbox[2][lane];
                        // motor
                        if (limitState != LIM_EQUAL) {
                            float Cdot = wB - wA - motorSpeed;
                            float impulse = -motorMass * Cdot;
                            float oldImpulse = motorImp;
                            motorImp = fclamp(oldImpulse + impulse, -maxMotorImpulse, maxMotorImpulse);
                            impulse = motorImp - oldImpulse;
                            wA -= iA * impulse;
                            wB += iB * impulse;
                        }
                        if (limitState != LIM_INACTIVE) {
                            V2 Cdot1 = vsub(vsub(vadd(vB, vcross_sv(wB, rB)), vA), vcross_sv(wA, rA));
                            float Cdot2 = wB - wA;
                            // impulse = -m_mass.Solve33(Cdot)
                            float bx = Cdot1.x, by = Cdot1.y, bz = Cdot2;
                            float sx = det33 * (bx * cyz_x + by * cyz_y + bz * cyz_z);
                            float cbx = by * m_ezz - bz * m_ezy, cby = bz * m_ezx - bx * m_ezz, cbz = bx * m_ezy - by * m_ezx;
                            float sy = det33 * (m_exx * cbx + m_eyx * cby + m_ezx * cbz);
                            float ebx = m_eyy * bz - m_ezy * by, eby = m_ezy * bx - m_eyx * bz, ebz = m_eyx * by - m_eyy * bx;
                            float sz = det33 * (m_exx * ebx + m_eyx * eby + m_ezx * ebz);
                            float ix = -sx, iy = -sy, iz = -sz;
                            if (limitState == LIM_EQUAL) {
                                impX += ix; impY += iy; impZ += iz;
                            } else {
                                float newImpulse = impZ + iz;
                                bool reduce = limitState == LIM_AT_LOWER ? newImpulse < 0.0f : newImpulse > 0.0f;
                                if (reduce) {
                                    V2 rhs = vadd(vneg(Cdot1), vscale(impZ, mk(m_ezx, m_ezy)));
                                    float rx = det22 * (m_eyy * rhs.x - m_eyx * rhs.y);
                                    float ry = det22 * (m_exx * rhs.y - m_eyx * rhs.x);
                                    ix = rx; iy = ry; iz = -impZ;
                                    impX += rx; impY += ry; impZ = 0.0f;
                                } else {
                                    impX += ix; impY += iy; impZ += iz;
                                }
                            }
                            V2 P = mk(ix, iy);
                            vA = vsub(vA, vscale(mA, P));
                            wA -= iA * (vcross(rA, P) + iz);
                            vB = vadd(vB, vscale(mB, P));
                            wB += iB * (vcross(rB, P) + iz);
                        } else {
                            V2 Cdot = vsub(vsub(vadd(vB, vcross_sv(wB, rB)), vA), vcross_sv(wA, rA));
                            V2 b = vneg(Cdot);
                            V2 impulse = mk(det22 * (m_eyy * b.x - m_eyx * b.y), det22 * (m_exx * b.y - m_eyx * b.x));
                            impX += impulse.x; impY += impulse.y;
                            vA = vsub(vA, vscale(mA, impulse));
                            wA -= iA * vcross(rA, impulse);
                            vB = vadd(vB, vscale(mB, impulse));
                            wB += iB * vcross(rB, impulse);
                        }
                        psh.mbox[0][pl] = vA.x; psh.mbox[1][pl] = vA.y; psh.mbox[2][pl] = wA;
                        psh.mbox[0][lane] = vB.x; psh.mbox[1][lane] = vB.y; psh.mbox[2][lane] = wB;
                    }
                    lds_sync();
                    if (tick == nextC) { // contacts of this body, in list order
                        nextC = (--leftC > 0) ? nextC + period : 0x7fffffff;
                        float cvx = psh.mbox[0][lane], cvy = psh.mbox[1][lane], cw = psh.mbox[2][lane];
#pragma unroll
                        for (int t = 0; t < KR; ++t)
                            if (t < nTouch) contact_solve(cc[t], mB, iB, friction, cvx, cvy, cw);
                        if (nTouch > KR) {
                            for (int t = KR; t < nTouch; ++t) {
                                const unsigned cb = (unsigned)(KT * SCR_WORDS + (t - KR) * CC_WORDS) * S.Lp + gl;
                                ContactC c;
                                cc_load(S, cb, c);
                                contact_solve(c, mB, iB, friction, cvx, cvy, cw);
                                SW(cb, 10) = c.n0; SW(cb, 11) = c.n1; SW(cb, 12) = c.t0; SW(cb, 13) = c.t1;
                            }
                        }
                        psh.mbox[0][lane] = cvx; psh.mbox[1][lane] = cvy; psh.mbox[2][lane] = cw;
                    }
                    lds_sync();
                }
                vx = psh.mbox[0][lane]; vy = psh.mbox[1][lane]; w = psh.mbox[2][lane];
            }
            // ---- StoreImpulses ----
#pragma unroll
            for (int t = 0; t < KR; ++t) {
                if (t < nTouch) {
                    unsigned o = SP_GET(slotPack, t) * Lp + gl;
                    CF(C_N0, o) = cc[t].n0;
                    CF(C_T0, o) = cc[t].t0;
                    if (cc[t].count > 1) {
                        CF(C_N1, o) = cc[t].n1;
                        CF(C_T1, o) = cc[t].t1;
                    }
                }
            }
            if (anyOverflow) {
                for (int t = KR; t < nTouch; ++t) {
                    const unsigned cb = (unsigned)(KT * SCR_WORDS + (t - KR) * CC_WORDS) * S.Lp + gl;
                    unsigned o = SP_GET(slotPack, t) * Lp + gl;
                    CF(C_N0, o) = SW(cb, 10);
                    CF(C_T0, o) = SW(cb, 12);
                    if (__float_as_int(SW(cb, 20)) > 1) {
                        CF(C_N1, o) = SW(cb, 11);
                        CF(C_T1, o) = SW(cb, 13);
                    }
                }
            }
            // ---- integrate positions ----
            if (active) {
                V2 v = mk(vx, vy);
                V2 translation = vscale(h, v);
                if (vdot(translation, translation) > B2_MAX_TRANSLATION_SQ) {
                    float ratio = B2_MAX_TRANSLATION / vlen(translation);
                    v = vscale(ratio, v);
                }
                float rotation = h * w;
                if (rotation * rotation > B2_MAX_ROTATION_SQ) {
                    float ratio = B2_MAX_ROTATION / fabs32(rotation);
                    w *= ratio;
                }
                px = px + h * v.x;
                py = py + h * v.y;
                ang += h * w;
                vx = v.x; vy = v.y;
            }
            // ---- position iterations (per creature early exit), software-pipelined with roll-back ----
            bool envSolved = false;
            int itersUsed = A.posIters;
            solve_positions_pipelined<K>(S, psh, gl, lane, pl, active, hasJoint, jround, period, nTouch, mA, iA, mB, iB, radiusB,
                                         limitState, motorMass, A.posIters, px, py, ang, envSolved, itersUsed);
            lastPosIters = itersUsed;
            // ---- sleep ----
            if (allowSleep) {
                float myT = FLT_MAX;
                if (active) {
                    const float linTolSqr = B2_LINEAR_SLEEP_TOL * B2_LINEAR_SLEEP_TOL;
                    const float angTolSqr = B2_ANGULAR_SLEEP_TOL * B2_ANGULAR_SLEEP_TOL;
                    if (w * w > angTolSqr || vdot(mk(vx, vy), mk(vx, vy)) > linTolSqr) {
                        sleepT = 0.0f;
                        myT = 0.0f;
                    } else {
                        sleepT += h;
                        myT = sleepT;
                    }
                }
                float minSleepTime = group_min<K>(myT);
                if (minSleepTime >= B2_TIME_TO_SLEEP && envSolved && active) {
                    awake = 0; sleepT = 0.0f; vx = 0.0f; vy = 0.0f; w = 0.0f;
                }
            }
            // ---- SynchronizeFixtures: swept AABB vs fat AABB (b2DynamicTree::MoveProxy) ----
            if (active) {
                Rot q0 = rot_set(a0);
                Rot q1 = rot_set(ang);
                V2 p0 = vsub(mk(c0x, c0y), rmul(q0, mk(0.0f, 0.0f)));
                V2 p1 = vsub(mk(px, py), rmul(q1, mk(0.0f, 0.0f)));
                AABB b1 = body_aabb(shape, hx, hy, p0, q0), b2 = body_aabb(shape, hx, hy, p1, q1);
                V2 lo = vmin2(b1.lo, b2.lo), hi = vmax2(b1.hi, b2.hi);
                V2 displacement = vsub(p1, p0);
                bool contains = fatLo.x <= lo.x && fatLo.y <= lo.y && hi.x <= fatHi.x && hi.y <= fatHi.y;
                if (!contains) {
                    V2 r = mk(B2_AABB_EXTENSION, B2_AABB_EXTENSION);
                    V2 flo = vsub(lo, r), fhi = vadd(hi, r);
                    V2 d = vscale(B2_AABB_MULTIPLIER, displacement);
                    if (d.x < 0.0f) flo.x += d.x; else fhi.x += d.x;
                    if (d.y < 0.0f) flo.y += d.y; else fhi.y += d.y;
                    LF(L_FATLX) = flo.x; LF(L_FATLY) = flo.y; LF(L_FATUX) = fhi.x; LF(L_FATUY) = fhi.y;
                    if (find_new_pairs(S, T, gl, cCount, flo, fhi, err)) {
                        if (sleepResetAlways || !awake) sleepT = 0.0f; // AddPair -> SetAwake(true)
                        awake = 1;
                    }
                }
            }
        }
        if (A.defer) { // continuous physics: the TOI kernel needs the sweep start and finishes the step
            const unsigned wb = (unsigned)SCR_SWEEP_BASE * S.Lp + gl;
            SW(wb, 0) = c0x; SW(wb, 1) = c0y; SW(wb, 2) = a0;
        }
        if (h > 0.0f) invDt0 = inv_dt;
        if (!A.defer) env_bookkeeping(S, env, sub, __shfl(px, base));
    }
    // ---- store ----
    gl = gl0; env = env0;
    LF(L_PX) = px; LF(L_PY) = py; LF(L_ANG) = ang; LF(L_VX) = vx; LF(L_VY) = vy; LF(L_W) = w;
    LF(L_SLEEPT) = sleepT; LI(L_AWAKE) = awake;
    LI(L_CCOUNT) = cCount;
    LF(L_JIMPX) = impX; LF(L_JIMPY) = impY; LF(L_JIMPZ) = impZ; LF(L_JMOTORIMP) = motorImp;
    LI(L_JLIMIT) = limitState; LF(L_JMOTORSPEED) = motorSpeed;
    int envErr = group_or<K>(err);
    if (sub == 0) {
        EF(E_INVDT0) = invDt0;
        EI(E_NEWFIX) = newFix; EI(E_ERR) = EI(E_ERR) | envErr; EI(E_POSITERS) = lastPosIters;
    }
}


// ---- several worlds (lane buckets of one population) in ONE launch ----
// Kernels of different worlds launched on different streams barely overlap: a big bucket's grid owns every
// wave slot (the register file is the limit) and the small buckets then run almost alone, latency bound.
// One grid over the blocks of all worlds lets the dispatcher pack them: block -> world by prefix sums.
#define REM2D_MAX_BATCH 8
struct Batch {
    State S[REM2D_MAX_BATCH];
    Terrain T[REM2D_MAX_BATCH];
    unsigned blockEnd[REM2D_MAX_BATCH]; // exclusive prefix sums of the worlds' block counts
    int lanes[REM2D_MAX_BATCH];
    int n;
};
DEV int batch_find(const Batch &B, unsigned &block) {
    int b = 0;
    while (b + 1 < B.n && block >= B.blockEnd[b]) ++b;
    if (b > 0) block -= B.blockEnd[b - 1];
    return b;
}
#define BATCH_DISPATCH(BODY, ...)                                 \
    switch (B.lanes[b]) {                                         \
    case 2: BODY<2>(B.S[b], B.T[b], A, block, ##__VA_ARGS__); break;   \
    case 4: BODY<4>(B.S[b], B.T[b], A, block, ##__VA_ARGS__); break;   \
    case 8: BODY<8>(B.S[b], B.T[b], A, block, ##__VA_ARGS__); break;   \
    case 16: BODY<16>(B.S[b], B.T[b], A, block, ##__VA_ARGS__); break; \
    case 32: BODY<32>(B.S[b], B.T[b], A, block, ##__VA_ARGS__); break; \
    default: BODY<64>(B.S[b], B.T[b], A, block, ##__VA_ARGS__); break; \
    }
__global__ __launch_bounds__(WAVE, 2) void rem2d_step_multi_kernel(Batch B, StepArgs A) {
    __shared__ PosShared psh;
    unsigned block = blockIdx.x;
    const int b = batch_find(B, block);
    BATCH_DISPATCH(step_body, psh)
}

// =====================================================================================
// TOI kernels: b2World::SolveTOI, then the per-step bookkeeping.  Launched after the step (or post)
// kernel when REM2D_FLAG_CONTINUOUS is set; kept out of the step kernel so that the branchy GJK /
// root-finder code does not share a register allocation with the velocity loop.
//
//   rem2d_toi_scan_multi_kernel   lane = body.  Invalidates last step's TOI flags and applies the two exact
//                           early-outs (toi_far_apart) to every pair.  A body all of whose pairs are far
//                           apart has alpha = 1 everywhere: SolveTOI leaves it untouched.  The others
//                           (a few per cent: bodies that are landing or sliding into an edge) are appended
//                           to a work list.  Light (no GJK), so it runs at high occupancy over its
//                           memory latency.
//   rem2d_toi_heavy_multi_kernel  lane = work-list entry: the full solve_toi_lane for that body, i.e. dense
//                           wavefronts of bodies that really need b2TimeOfImpact / TOI sub-steps instead
//                           of one such lane per wavefront.  Bodies are independent (terrain sweeps are
//                           time-transparent), so the list order does not matter.
// reward / done / fitness only read the root body: whichever kernel finalises the root does the bookkeeping.
// =====================================================================================
// the scan for one body (its final pose of the discrete step in registers or freshly loaded): invalidate the TOI flags of
// its pairs, apply the exact early-outs, then either queue the body for the TOI solve or -- for a root -- finish the step
// true: the body has a pair the exact early-outs cannot rule out -> it needs the TOI solve (solve_toi_lane)
DEV bool toi_scan_heavy(const State &S, const Terrain &T, float dt, unsigned gl, int shape, float px, float py, float ang,
                        float c0x, float c0y, float a0, float hx, float hy, int awake, int cCount) {
    const unsigned Lp = S.Lp;
    bool heavy = false;
    if (shape != SHAPE_NONE && dt > 0.0f) {
        Sweep sw;
        sw.c0 = mk(c0x, c0y); sw.a0 = a0;
        sw.c = mk(px, py); sw.a = ang;
        const Proxy pB = proxy_body(shape, hx, hy);
        const float coreR = shape == SHAPE_BOX ? sqrtf(hx * hx + hy * hy) : 0.0f;
        for (int s = 0; s < cCount; ++s) {
            unsigned o = (unsigned)s * Lp + gl;
            const int old = CI(C_INFO, o);
            const int info = old & (0xffff | CI_ENABLED); // m_stepComplete: invalidate TOIs
            if (info != old) CI(C_INFO, o) = info;
            if (!(info & CI_ENABLED) || !awake || heavy) continue;
            int e = CI(C_EDGE, o);
            Proxy pA = proxy_edge(static_vert(T, e, 0), static_vert(T, e, 1));
            if (e < T.nPoly) { pA.v[2] = static_vert(T, e, 2); pA.v[3] = static_vert(T, e, 3); pA.count = 4; }
            heavy = !toi_far_apart(pA, pB, sw, shape, hx, hy, coreR);
        }
    }
    return heavy;
}
DEV void toi_scan_lane(const State &S, const Terrain &T, float dt, unsigned gl, unsigned env, int sub, int shape, float px,
                       float py, float ang, float c0x, float c0y, float a0, float hx, float hy, int awake, int cCount) {
    const bool heavy = toi_scan_heavy(S, T, dt, gl, shape, px, py, ang, c0x, c0y, a0, hx, hy, awake, cCount);
    if (heavy) {
        int slot = atomicAdd(&S.toiWork[0], 1);
        S.toiWork[16 + slot] = (int)gl;
    } else if (sub == 0) {
        env_bookkeeping(S, env, 0, px);
    }
}
template <int K>
DEV void toi_scan_body(const State &S, const Terrain &T, const StepArgs &A, unsigned block) {
    const int lane = threadIdx.x;
    const unsigned gl = block * WAVE + lane;
    const unsigned env = gl / K;
    const int sub = lane & (K - 1);
    const unsigned Lp = S.Lp;
    if (S.flags & REM2D_FLAG_SKIP_FROZEN) {
        if (__all(EI(E_FROZEN) != 0 ? 1 : 0)) return; // the step kernel skipped this wavefront too
    }
    const unsigned wb = (unsigned)SCR_SWEEP_BASE * Lp + gl;
    toi_scan_lane(S, T, A.dt, gl, env, sub, LI(L_SHAPE), LF(L_PX), LF(L_PY), LF(L_ANG), SW(wb, 0), SW(wb, 1), SW(wb, 2),
                  LF(L_HX), LF(L_HY), LI(L_AWAKE), LI(L_CCOUNT));
}
// the TOI solve of one body (lane gl of the world) by the G lanes that carry it, then the step's bookkeeping if it is a root
template <int K, int COLS = WAVE>
DEV void toi_heavy_one(const State &S, const Terrain &T, const StepArgs &A, unsigned gl, ToiSharedT<COLS> &ts, int sub, int G) {
    const unsigned env = gl / K;
    const int shape = LI(L_SHAPE);
    const unsigned wb = (unsigned)SCR_SWEEP_BASE * S.Lp + gl;
    LaneBody B;
    B.px = LF(L_PX); B.py = LF(L_PY); B.ang = LF(L_ANG); B.vx = LF(L_VX); B.vy = LF(L_VY); B.w = LF(L_W);
    B.sleepT = LF(L_SLEEPT); B.awake = LI(L_AWAKE); B.cCount = LI(L_CCOUNT); B.err = 0; B.events = 0;
    B = solve_toi_lane(S, T, gl, shape, LF(L_HX), LF(L_HY), LF(L_INVM), LF(L_INVI), A.dt, A.velIters, SW(wb, 0), SW(wb, 1),
                       SW(wb, 2), B, ts, (int)threadIdx.x, sub, G);
    if (sub != 0) return; // the body's other lanes hold the same result
    LF(L_PX) = B.px; LF(L_PY) = B.py; LF(L_ANG) = B.ang; LF(L_VX) = B.vx; LF(L_VY) = B.vy; LF(L_W) = B.w;
    LF(L_SLEEPT) = B.sleepT; LI(L_AWAKE) = B.awake; LI(L_CCOUNT) = B.cCount;
    if (B.events > 0) atomicAdd(&EI(E_TOIEVENTS), B.events);
    if (B.err) atomicOr(&EI(E_ERR), B.err);
    if ((gl & (K - 1)) == 0) env_bookkeeping(S, env, 0, B.px);
}
template <int K>
DEV void toi_heavy_body(const State &S, const Terrain &T, const StepArgs &A, unsigned block, ToiShared &ts) {
    // Bodies per wavefront: the lanes of a wavefront run the event loop in lockstep -- every b2TimeOfImpact any lane needs at
    // any pair slot, the 180 sweeps of a sub-step as soon as one lane's does not reach its fixed point, as many rounds as
    // the lane with the most events -- so a full wavefront costs about twice its slowest body.  The list is short (about
    // 1 % of the bodies): it is dealt out REM2D_HEAVY_PER_WAVE bodies (default: one) to a wavefront, more only when it is so long that
    // the grid would not cover it (the landing after a reset).
    const unsigned queued = (unsigned)S.toiWork[0], blocks = S.Lp / WAVE;
    unsigned per = (queued + blocks - 1) / blocks;
    per = per < (unsigned)A.heavyPerWave ? (unsigned)A.heavyPerWave : per;
    if (block * per >= queued) return;
    unsigned G = 1; // lanes per body: the largest power of two that fits (rem2d_toi.h: the alpha pass runs a body's pairs side by side)
    while (G * 2 * per <= WAVE) G *= 2;
    const unsigned sub = threadIdx.x & (G - 1), slot = threadIdx.x / G;
    const unsigned idx = block * per + slot;
    if (slot >= per || idx >= queued) return;
    if (A.prio & 4) __builtin_amdgcn_s_setprio(3); // (a few hundred wavefronts, each one long dependent chain that the group's next step waits for)
    toi_heavy_one<K>(S, T, A, (unsigned)S.toiWork[16 + idx], ts, (int)sub, (int)G);
}

__global__ __launch_bounds__(WAVE) void rem2d_toi_scan_multi_kernel(Batch B, StepArgs A) {
    unsigned block = blockIdx.x;
    const int b = batch_find(B, block);
    BATCH_DISPATCH(toi_scan_body)
}
__global__ __launch_bounds__(WAVE, 2) void rem2d_toi_heavy_multi_kernel(Batch B, StepArgs A) {
    __shared__ ToiShared ts;
    unsigned block = blockIdx.x;
    const int b = batch_find(B, block);
    BATCH_DISPATCH(toi_heavy_body, ts)
}

// =====================================================================================
// reset kernel: Modular2D.reset -> b2World() + create_robot
// =====================================================================================
DEV void box_mass(float hx, float hy, float &mass, float &I) { // b2PolygonShape::ComputeMass, density 1
    const V2 vs[4] = {mk(-hx, -hy), mk(hx, -hy), mk(hx, hy), mk(-hx, hy)};
    V2 center = mk(0.0f, 0.0f);
    float area = 0.0f, In = 0.0f;
    V2 s = mk(0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) s = vadd(s, vs[i]);
    s = vscale(1.0f / 4.0f, s);
    const float k_inv3 = 1.0f / 3.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        V2 e1 = vsub(vs[i], s);
        V2 e2 = vsub(vs[(i + 1) & 3], s);
        float D = vcross(e1, e2);
        float triangleArea = 0.5f * D;
        area += triangleArea;
        center = vadd(center, vscale(triangleArea * k_inv3, vadd(e1, e2)));
        float ex1 = e1.x, ey1 = e1.y, ex2 = e2.x, ey2 = e2.y;
        float intx2 = ex1 * ex1 + ex2 * ex1 + ex2 * ex2;
        float inty2 = ey1 * ey1 + ey2 * ey1 + ey2 * ey2;
        In += (0.25f * k_inv3 * D) * (intx2 + inty2);
    }
    const float density = 1.0f;
    mass = density * area;
    center = vscale(1.0f / area, center);
    V2 mc = vadd(center, s);
    float Iout = density * In;
    Iout += mass * (vdot(mc, mc) - vdot(center, center));
    // b2Body::ResetMassData with a single fixture
    float m = 0.0f + mass;
    V2 lc = vadd(mk(0.0f, 0.0f), vscale(mass, mc));
    float Ib = 0.0f + Iout;
    float invM = 1.0f / m;
    lc = vscale(invM, lc);
    Ib -= m * vdot(lc, lc);
    mass = m;
    I = Ib;
}
__global__ void rem2d_reset_kernel(State S, rem2d_morph M, int K) {
    unsigned gl = blockIdx.x * blockDim.x + threadIdx.x;
    if (gl >= S.Lp) return;
    unsigned env = gl / (unsigned)K, sub = gl % (unsigned)K;
    bool real = env < S.nEnvs;
    int shape = real ? M.shape[gl] : 0;
    float hx = real ? M.hx[gl] : 0.0f, hy = real ? M.hy[gl] : 0.0f;
    float x = real ? M.x[gl] : 0.0f, y = real ? M.y[gl] : 0.0f, a = real ? M.angle[gl] : 0.0f;
    float invM = 0.0f, invI = 0.0f;
    if (shape == SHAPE_BOX) {
        float m, I;
        box_mass(hx, hy, m, I);
        invM = 1.0f / m;
        invI = I > 0.0f ? 1.0f / I : 0.0f;
    } else if (shape == SHAPE_CIRCLE) {
        const float density = 1.0f;
        float mass = density * B2_PI * hx * hx;
        float I = mass * (0.5f * hx * hx + vdot(mk(0.0f, 0.0f), mk(0.0f, 0.0f)));
        float m = 0.0f + mass;
        V2 lc = vadd(mk(0.0f, 0.0f), vscale(mass, mk(0.0f, 0.0f)));
        float Ib = 0.0f + I;
        invM = 1.0f / m;
        lc = vscale(invM, lc);
        Ib -= m * vdot(lc, lc);
        invI = Ib > 0.0f ? 1.0f / Ib : 0.0f;
    }
    LI(L_SHAPE) = shape;
    LF(L_HX) = hx; LF(L_HY) = hy; LF(L_INVM) = invM; LF(L_INVI) = invI;
    // b2Body ctor + ResetMassData: sweep.c = b2Mul(xf, localCenter = 0)
    Rot q = rot_set(a);
    V2 c = xmul(q, mk(x, y), mk(0.0f, 0.0f));
    LF(L_PX) = c.x; LF(L_PY) = c.y; LF(L_ANG) = a;
    LF(L_VX) = 0.0f; LF(L_VY) = 0.0f; LF(L_W) = 0.0f; LF(L_SLEEPT) = 0.0f;
    LI(L_AWAKE) = shape != SHAPE_NONE ? 1 : 0;
    // b2Fixture::CreateProxies: fat AABB of the initial transform
    AABB bb = body_aabb(shape == SHAPE_NONE ? SHAPE_CIRCLE : shape, hx, hy, mk(x, y), q);
    V2 r = mk(B2_AABB_EXTENSION, B2_AABB_EXTENSION);
    V2 lo = vsub(bb.lo, r), hi = vadd(bb.hi, r);
    LF(L_FATLX) = lo.x; LF(L_FATLY) = lo.y; LF(L_FATUX) = hi.x; LF(L_FATUY) = hi.y;
    int parent = real && shape != SHAPE_NONE ? M.parent[gl] : -1;
    LI(L_PARENT) = parent;
    LI(L_JROUND) = real ? M.jround[gl] : 0;
    LF(L_JAX) = real ? M.ax[gl] : 0.0f; LF(L_JAY) = real ? M.ay[gl] : 0.0f;
    LF(L_JBX) = real ? M.bx[gl] : 0.0f; LF(L_JBY) = real ? M.by[gl] : 0.0f;
    LF(L_JTORQUE) = real ? M.torque[gl] : 0.0f; LF(L_JLOWER) = real ? M.lower[gl] : 0.0f;
    LF(L_JUPPER) = real ? M.upper[gl] : 0.0f;
    LF(L_JIMPX) = 0.0f; LF(L_JIMPY) = 0.0f; LF(L_JIMPZ) = 0.0f; LF(L_JMOTORIMP) = 0.0f; LF(L_JMOTORSPEED) = 0.0f;
    LI(L_JLIMIT) = LIM_INACTIVE;
    LD(D_CAMP) = real ? M.amp[gl] : 0.0; LD(D_CPHASE) = real ? M.phase[gl] : 0.0; LD(D_CFREQ) = real ? M.freq[gl] : 0.0;
    LD(D_COFFSET) = real ? M.offset[gl] : 0.0; LD(D_CISTATE) = real ? M.istate[gl] : 0.0;
    LI(L_CCOUNT) = 0;
    for (int s = 0; s < KC; ++s) {
        unsigned o = (unsigned)s * S.Lp + gl;
        CI(C_EDGE, o) = -1; CI(C_INFO, o) = 0; CU(C_KEY0, o) = 0u; CU(C_KEY1, o) = 0u;
        CF(C_N0, o) = 0.0f; CF(C_N1, o) = 0.0f; CF(C_T0, o) = 0.0f; CF(C_T1, o) = 0.0f;
    }
    if (sub == 0) {
        ED(E_WOD) = 0.0; ED(E_FITNESS) = 0.0; EF(E_REWARD) = 0.0f; EI(E_DONE) = 0; EI(E_EVERDONE) = 0;
        EI(E_FROZEN) = 0; EI(E_STEPS) = 0; EF(E_INVDT0) = 0.0f; EI(E_NEWFIX) = 1; EI(E_ERR) = 0;
        EI(E_POSITERS) = 0; EI(E_TOIEVENTS) = 0;
    }
}

#endif
