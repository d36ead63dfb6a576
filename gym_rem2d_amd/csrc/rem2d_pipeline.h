// rem2d_pipeline.h -- the split step pipeline: rem2d_pre_kernel -> rem2d_vel_kernel -> rem2d_post_kernel.
// Part of the single translation unit rem2d.hip (see its header comment); not a stand-alone header.
//
// Same arithmetic as rem2d_step_kernel, cut at the two ends of the velocity phase of b2Island::Solve so
// that the 180 velocity iterations run in a kernel of their own with a different work mapping:
//
//   pre  (lane = body, workgroup = wave64): controller / PID, Collide, velocity integration, contact and
//        joint constraint set-up.  Hands over through HBM: integrated velocities (L_VX..), dtRatio-scaled
//        warm-start impulses (L_JIMP*, contact records), limit state, one contact record per touching
//        manifold (scratch, cc_store layout), joint lever arms, nTouch / pair-slot map.
//   vel  (workgroup = 4 waves = 256 bodies): lanes are re-assigned to CONSTRAINTS.  The joints of the
//        256 bodies are counting-sorted by their phase in the modulo schedule (round mod period) and
//        dealt to consecutive lanes, so the joints that fire in one tick sit in one or two wavefronts
//        that run at full lane occupancy while the other wavefronts of the workgroup wait at the barrier
//        (an idle wave costs no issue slots; an idle lane does).  Touching bodies are compacted the same
//        way into contact lanes.  Body velocities live in a 3 KB LDS mailbox; one s_barrier per slot.
//        In rem2d_step_kernel the same slots run at 1/period (joints) and ~5 % (contacts) lane occupancy.
//   post (lane = body, workgroup = wave64): position integration, position iterations, sleep,
//        SynchronizeFixtures / FindNewContacts, per-step bookkeeping or hand-over to the TOI kernel.
//
// Any two operations that share a body keep the order proven by compiler.pipeline_schedule (any period
// >= the creature's own is valid, so the workgroup runs all of its creatures at the largest one), hence
// the result is bit-identical to rem2d_step_kernel and to the sequential oracle.
#ifndef REM2D_PIPELINE_H
#define REM2D_PIPELINE_H

#define VEL_THREADS 256
#define VEL_WAVES (VEL_THREADS / WAVE)
#define VEL_PHASES 8 // phase buckets of the joint sort (period <= 4 for the reference's modules)

struct VelArgs { int K; int velIters; float dt; float friction; };

// ---------------------------------------------------------------------------------------------------
// pre: Modular2D.step's controller sweep, b2World::Step up to (not including) the warm start
// ---------------------------------------------------------------------------------------------------
template <int K>
DEV void pre_body(const State &S, const Terrain &T, const StepArgs &A, unsigned block) {
    const int lane = threadIdx.x;
    const unsigned gl = block * WAVE + lane;
    const unsigned env = gl / K;
    const int base = lane & ~(K - 1);
    const int sub = lane & (K - 1);
    const unsigned Lp = S.Lp;
    if (gl == 0) S.toiWork[0] = 0; // work list of the TOI kernels that follow
    if (S.flags & REM2D_FLAG_SKIP_FROZEN) { // evaluate() has left its loop for every creature of this wavefront
        if (__all(EI(E_FROZEN) != 0 ? 1 : 0)) {
            const unsigned mb = (unsigned)SCR_MISC_BASE * Lp + gl;
            SW(mb, 0) = __int_as_float(0); // nothing to solve for the velocity kernel; post and the TOI kernels skip too
            return;
        }
    }

    const int shape = LI(L_SHAPE);
    const bool active = shape != SHAPE_NONE;
    const float mB = LF(L_INVM), iB = LF(L_INVI);
    const float px = LF(L_PX), py = LF(L_PY), ang = LF(L_ANG);
    float vx = LF(L_VX), vy = LF(L_VY), w = LF(L_W);
    float sleepT = LF(L_SLEEPT);
    int awake = LI(L_AWAKE);
    int cCount = LI(L_CCOUNT);
    const int parent = LI(L_PARENT);
    const bool hasJoint = active && parent >= 0;
    const int pl = base + (parent >= 0 ? parent : 0);
    float impX = LF(L_JIMPX), impY = LF(L_JIMPY), impZ = LF(L_JIMPZ), motorImp = LF(L_JMOTORIMP);
    int limitState = LI(L_JLIMIT);
    float motorSpeed = LF(L_JMOTORSPEED);
    const int childLo = group_or<K>((hasJoint && parent < 32) ? (1 << parent) : 0);
    const int childHi = group_or<K>((hasJoint && parent >= 32) ? (1 << (parent - 32)) : 0);
    const bool jointed = hasJoint || (((sub < 32 ? childLo >> sub : childHi >> (sub - 32))) & 1);
    const float invDt0 = EF(E_INVDT0);
    int err = 0;
    const float h = A.dt;
    const bool sleepResetAlways = (S.flags & REM2D_FLAG_SLEEP_RESET_ALWAYS) != 0;

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
    const float dtRatio = invDt0 * h;
    const float hx = LF(L_HX), hy = LF(L_HY);
    const float radiusB = shape == SHAPE_CIRCLE ? hx : B2_POLYGON_RADIUS;
    V2 fatLo = mk(LF(L_FATLX), LF(L_FATLY)), fatHi = mk(LF(L_FATUX), LF(L_FATUY));
    if (EI(E_NEWFIX)) { // FindNewContacts for freshly created fixtures
        if (active && find_new_pairs(S, T, gl, cCount, fatLo, fatHi, err)) {
            if (sleepResetAlways || !awake) sleepT = 0.0f;
            awake = 1;
        }
    }
    Rot q = rot_set(ang);
    // ---- b2ContactManager::Collide: destroy separated pairs, update manifolds ----
    int nTouch = 0;
    unsigned slotPack = 0u;
    if (active && awake) {
        int s = 0;
        while (s < cCount) {
            unsigned o = (unsigned)s * Lp + gl;
            int e = CI(C_EDGE, o);
            if (!aabb_overlap(mk(T.flx[e], T.fly[e]), mk(T.fux[e], T.fuy[e]), fatLo, fatHi)) {
                if ((CI(C_INFO, o) & 0xff) > 0 && sleepResetAlways) sleepT = 0.0f;
                pairs_remove(S, gl, cCount, s);
                continue;
            }
            Manifold m;
            contact_update_slot(S, T, o, shape, hx, hy, mk(px, py), q, m, sleepResetAlways, sleepT);
            if (m.count > 0) {
                if (nTouch < KT) {
                    manifold_store(S, gl, nTouch, m);
                    slotPack |= (unsigned)s << (5 * nTouch);
                    ++nTouch;
                } else {
                    err |= REM2D_ERR_SOLVER_OVERFLOW;
                }
            }
            ++s;
        }
    }
    // =============== b2World::Solve (first part) ===============
    const int envAwake = group_or<K>(active && awake ? 1 : 0);
    if (envAwake) {
        if (active && (!awake || sleepResetAlways)) { awake = 1; sleepT = 0.0f; } // island.Add -> SetAwake(true)
        if (active) { // integrate velocities (gravity (0,-10), no forces, no damping)
            V2 acc = vadd(vscale(1.0f, mk(0.0f, -10.0f)), vscale(mB, mk(0.0f, 0.0f)));
            V2 v = vadd(mk(vx, vy), vscale(h, acc));
            float wz = w + h * iB * 0.0f;
            v = vscale(1.0f / (1.0f + h * 0.0f), v);
            wz *= 1.0f / (1.0f + h * 0.0f);
            vx = v.x; vy = v.y; w = wz;
        }
        // contact constraints: b2ContactSolver ctor + InitializeVelocityConstraints -> records
        for (int t = 0; t < nTouch; ++t) {
            const unsigned sb = (unsigned)(t * SCR_WORDS) * Lp + gl;
            int tc = __float_as_int(SW(sb, 0));
            unsigned o = ((slotPack >> (5 * t)) & 0x1f) * Lp + gl;
            ContactC c;
            contact_setup(c, tc & 0xff, tc >> 8, mk(SW(sb, 1), SW(sb, 2)), mk(SW(sb, 3), SW(sb, 4)), mk(SW(sb, 5), SW(sb, 6)),
                          mk(SW(sb, 7), SW(sb, 8)), mk(px, py), q, mB, iB, radiusB, dtRatio * CF(C_N0, o), dtRatio * CF(C_T0, o),
                          dtRatio * CF(C_N1, o), dtRatio * CF(C_T1, o));
            cc_store(S, (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl, c);
        }
        // joints: the state-dependent part of InitVelocityConstraints (lever arms, limit state)
        {
            float sA = __shfl(q.s, pl), cA = __shfl(q.c, pl);
            float aA = __shfl(ang, pl);
            if (hasJoint) {
                const V2 anchorA = mk(LF(L_JAX), LF(L_JAY)), anchorB = mk(LF(L_JBX), LF(L_JBY));
                const float jLower = LF(L_JLOWER), jUpper = LF(L_JUPPER);
                Rot qA; qA.s = sA; qA.c = cA;
                V2 rA = rmul(qA, vsub(anchorA, mk(0.0f, 0.0f)));
                V2 rB = rmul(q, vsub(anchorB, mk(0.0f, 0.0f)));
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
                const unsigned jb = (unsigned)SCR_JREC_BASE * Lp + gl;
                SW(jb, 0) = rA.x; SW(jb, 1) = rA.y; SW(jb, 2) = rB.x; SW(jb, 3) = rB.y;
            }
        }
    }
    // ---- hand-over ----
    LF(L_VX) = vx; LF(L_VY) = vy; LF(L_W) = w;
    LF(L_SLEEPT) = sleepT; LI(L_AWAKE) = awake;
    LI(L_CCOUNT) = cCount;
    LF(L_JIMPX) = impX; LF(L_JIMPY) = impY; LF(L_JIMPZ) = impZ; LF(L_JMOTORIMP) = motorImp;
    LI(L_JLIMIT) = limitState; LF(L_JMOTORSPEED) = motorSpeed;
    {
        const unsigned mb = (unsigned)SCR_MISC_BASE * Lp + gl;
        SW(mb, 0) = __int_as_float(nTouch | (envAwake ? 0x100 : 0));
        SW(mb, 1) = __int_as_float((int)slotPack);
    }
    int envErr = group_or<K>(err);
    if (sub == 0) {
        EI(E_NEWFIX) = 0;
        if (envErr) EI(E_ERR) = EI(E_ERR) | envErr;
    }
}

template <int K>
__global__ __launch_bounds__(WAVE) void rem2d_pre_kernel(State S, Terrain T, StepArgs A) {
    pre_body<K>(S, T, A, blockIdx.x);
}
__global__ __launch_bounds__(WAVE) void rem2d_pre_multi_kernel(Batch B, StepArgs A) {
    unsigned block = blockIdx.x;
    const int b = batch_find(B, block);
    BATCH_DISPATCH(pre_body)
}

// ---------------------------------------------------------------------------------------------------
// vel: warm start + velocity iterations + StoreImpulses, lanes = constraints
// ---------------------------------------------------------------------------------------------------
struct JointV {
    V2 rA, rB;
    float mA, iA, mB, iB;
    float exx, eyx, ezx, eyy, ezy, ezz, motorMass, det33, det22, cyzx, cyzy, cyzz;
    float impX, impY, impZ, motorImp, motorSpeed, maxMotorImpulse;
    int limitState;
};
// b2RevoluteJoint::SolveVelocityConstraints (motor, then limit 3x3 / point 2x2); same expression
// sequence as the joint slot of rem2d_step_kernel
DEV void joint_solve_velocity(JointV &j, V2 &vA, float &wA, V2 &vB, float &wB) {
    if (j.limitState != LIM_EQUAL) {
        float Cdot = wB - wA - j.motorSpeed;
        float impulse = -j.motorMass * Cdot;
        float oldImpulse = j.motorImp;
        j.motorImp = fclamp(oldImpulse + impulse, -j.maxMotorImpulse, j.maxMotorImpulse);
        impulse = j.motorImp - oldImpulse;
        wA -= j.iA * impulse;
        wB += j.iB * impulse;
    }
    if (j.limitState != LIM_INACTIVE) {
        V2 Cdot1 = vsub(vsub(vadd(vB, vcross_sv(wB, j.rB)), vA), vcross_sv(wA, j.rA));
        float Cdot2 = wB - wA;
        float bx = Cdot1.x, by = Cdot1.y, bz = Cdot2;
        float sx = j.det33 * (bx * j.cyzx + by * j.cyzy + bz * j.cyzz);
        float cbx = by * j.ezz - bz * j.ezy, cby = bz * j.ezx - bx * j.ezz, cbz = bx * j.ezy - by * j.ezx;
        float sy = j.det33 * (j.exx * cbx + j.eyx * cby + j.ezx * cbz);
        float ebx = j.eyy * bz - j.ezy * by, eby = j.ezy * bx - j.eyx * bz, ebz = j.eyx * by - j.eyy * bx;
        float sz = j.det33 * (j.exx * ebx + j.eyx * eby + j.ezx * ebz);
        float ix = -sx, iy = -sy, iz = -sz;
        if (j.limitState == LIM_EQUAL) {
            j.impX += ix; j.impY += iy; j.impZ += iz;
        } else {
            float newImpulse = j.impZ + iz;
            bool reduce = j.limitState == LIM_AT_LOWER ? newImpulse < 0.0f : newImpulse > 0.0f;
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
        V2 b = vneg(Cdot);
        V2 impulse = mk(j.det22 * (j.eyy * b.x - j.eyx * b.y), j.det22 * (j.exx * b.y - j.eyx * b.x));
        j.impX += impulse.x; j.impY += impulse.y;
        vA = vsub(vA, vscale(j.mA, impulse));
        wA -= j.iA * vcross(j.rA, impulse);
        vB = vadd(vB, vscale(j.mB, impulse));
        wB += j.iB * vcross(j.rB, impulse);
    }
}

DEV int wave_sum(int v) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) v += __shfl_xor(v, o);
    return v;
}
DEV int wave_or(int v) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) v |= __shfl_xor(v, o);
    return v;
}

struct VelShared {
    float vel[3][VEL_THREADS];            // body velocity mailbox (vx, vy, w)
    int cntJ[VEL_PHASES][VEL_WAVES];      // joints per (phase, wave)
    int cntC[VEL_WAVES];                  // touching bodies per wave
    int redA[VEL_WAVES], redB[VEL_WAVES], redC[VEL_WAVES], redD[VEL_WAVES];
    unsigned char jlist[VEL_THREADS];     // joint lane -> child body (workgroup-local id)
    unsigned char clist[VEL_THREADS];     // contact lane -> body
};
DEV void vel_body(const State &S, const VelArgs &A, unsigned block, VelShared &sh) {
    float (&vel)[3][VEL_THREADS] = sh.vel;
    int (&cntJ)[VEL_PHASES][VEL_WAVES] = sh.cntJ;
    int (&cntC)[VEL_WAVES] = sh.cntC;
    int (&redA)[VEL_WAVES] = sh.redA, (&redB)[VEL_WAVES] = sh.redB, (&redC)[VEL_WAVES] = sh.redC, (&redD)[VEL_WAVES] = sh.redD;
    unsigned char (&jlist)[VEL_THREADS] = sh.jlist, (&clist)[VEL_THREADS] = sh.clist;

    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & (WAVE - 1);
    const unsigned wg0 = block * VEL_THREADS;
    const unsigned Lp = S.Lp;
    const int K = A.K;
    const int iters = A.velIters;
    const float h = A.dt;

    // ---------------- body role: publish velocity, schedule words ----------------
    int misc = 0, sched = 0, parent = -1;
    {
        const unsigned gl = wg0 + tid;
        if (gl < Lp) {
            misc = __float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + gl, 0));
            sched = LI(L_JROUND);
            parent = LI(L_PARENT);
            vel[0][tid] = LF(L_VX); vel[1][tid] = LF(L_VY); vel[2][tid] = LF(L_W);
        }
    }
    const bool solve = (misc & 0x100) != 0;
    const bool hasJ = solve && parent >= 0;
    const bool touching = solve && (misc & 0xff) > 0;
    // one period for the whole workgroup (any period >= a creature's own keeps its order)
    {
        int p = wave_max(solve ? ((sched >> 16) & 0xff) : 0);
        if (ln == 0) redA[wv] = p;
    }
    __syncthreads();
    int P = max(max(redA[0], redA[1]), max(redA[2], redA[3]));
    if (P < 1) P = 1;
    const int jr = sched & 0xff, oc = (sched >> 8) & 0xff;
    const int jphase = (jr % P) & (VEL_PHASES - 1);
    // ---------------- counting sort of the joints by phase; compaction of touching bodies ----------------
    unsigned long long jmask[VEL_PHASES];
#pragma unroll
    for (int p = 0; p < VEL_PHASES; ++p) {
        jmask[p] = __ballot(hasJ && jphase == p);
        if (ln == 0) cntJ[p][wv] = __popcll(jmask[p]);
    }
    const unsigned long long cmaskW = __ballot(touching);
    if (ln == 0) cntC[wv] = __popcll(cmaskW);
    {
        // last tick any of this wave's bodies needs; joint rounds of the warm start; contact phases in use
        int lastTick = -1;
        if (iters > 0) {
            if (hasJ) lastTick = jr + (iters - 1) * P;
            if (touching) lastTick = max(lastTick, oc + (iters - 1) * P);
        }
        int lt = wave_max(lastTick), mr = wave_max(hasJ ? jr : -1);
        int cph = wave_or(touching ? (P <= 32 ? (1 << (oc % P)) : -1) : 0);
        if (ln == 0) { redB[wv] = lt; redC[wv] = mr; redD[wv] = cph; }
    }
    __syncthreads();
    const int nTicks = max(max(redB[0], redB[1]), max(redB[2], redB[3])) + 1;
    const int nRounds = max(max(redC[0], redC[1]), max(redC[2], redC[3])) + 1;
    const int cPhases = redD[0] | redD[1] | redD[2] | redD[3];
    int NJ = 0, NC = 0;
    {
        const unsigned long long below = ln == 0 ? 0ull : (~0ull >> (WAVE - ln));
        int startMine = 0;
#pragma unroll
        for (int p = 0; p < VEL_PHASES; ++p) {
#pragma unroll
            for (int x = 0; x < VEL_WAVES; ++x) {
                int c = cntJ[p][x];
                if (p < jphase || (p == jphase && x < wv)) startMine += c;
                NJ += c;
            }
        }
        if (hasJ) {
            unsigned long long mine = 0ull;
#pragma unroll
            for (int p = 0; p < VEL_PHASES; ++p) mine = (p == jphase) ? jmask[p] : mine;
            jlist[startMine + __popcll(mine & below)] = (unsigned char)tid;
        }
        int startC = 0;
#pragma unroll
        for (int x = 0; x < VEL_WAVES; ++x) {
            int c = cntC[x];
            if (x < wv) startC += c;
            NC += c;
        }
        if (touching) clist[startC + __popcll(cmaskW & below)] = (unsigned char)tid;
    }
    __syncthreads();

    // ---------------- joint role: lane tid solves the joint of child body jlist[tid] ----------------
    const bool jrole = tid < NJ;
    JointV J;
    int jA = 0, jB = 0, jround = -1;
    unsigned glJ = 0;
    J.limitState = LIM_INACTIVE;
    if (jrole) {
        jB = jlist[tid];
        const unsigned gl = wg0 + (unsigned)jB;
        glJ = gl;
        jA = (jB & ~(K - 1)) + LI(L_PARENT);
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
            const unsigned gl = wg0 + (unsigned)jA;
            J.mA = LF(L_INVM); J.iA = LF(L_INVI);
        }
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
    // ---------------- contact role: lane VEL_THREADS-1-i solves the contacts of body clist[i] ----------------
    // contact lanes fill one wavefront from the top; which wavefront rotates with the workgroup so that the
    // contact work of the workgroups resident on one CU does not pile up on the same SIMD
    const int crot = (int)((block >> 8) + (block >> 10)) & (VEL_WAVES - 1);
    const int ci = VEL_THREADS - 1 - ((tid + crot * WAVE) & (VEL_THREADS - 1));
    const bool crole = ci < NC;
    ContactC cc[KR];
#pragma unroll
    for (int t = 0; t < KR; ++t) cc[t].count = 0;
    int cBody = 0, nTouch = 0, offC = 0;
    unsigned glC = 0, slotPack = 0u;
    float cmB = 0.0f, ciB = 0.0f;
    if (crole) {
        cBody = clist[ci];
        const unsigned gl = wg0 + (unsigned)cBody;
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
    __syncthreads();
    // ---------------- warm start: contacts (per body, list order), then joints in island rounds ----------------
    const float mu = A.friction; // b2MixFriction(terrain, module)
    if (crole) {
        float cvx = vel[0][cBody], cvy = vel[1][cBody], cw = vel[2][cBody];
#pragma unroll
        for (int t = 0; t < KR; ++t)
            if (t < nTouch) contact_warm_start(cc[t], cmB, ciB, cvx, cvy, cw);
        for (int t = KR; t < nTouch; ++t) {
            ContactC c;
            cc_load(S, (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + glC, c);
            contact_warm_start(c, cmB, ciB, cvx, cvy, cw);
        }
        vel[0][cBody] = cvx; vel[1][cBody] = cvy; vel[2][cBody] = cw;
    }
    __syncthreads();
    for (int r = 0; r < nRounds; ++r) {
        if (jround == r) {
            V2 vA = mk(vel[0][jA], vel[1][jA]);
            float wA = vel[2][jA];
            V2 vB = mk(vel[0][jB], vel[1][jB]);
            float wB = vel[2][jB];
            V2 Pw = mk(J.impX, J.impY);
            vA = vsub(vA, vscale(J.mA, Pw));
            wA -= J.iA * (vcross(J.rA, Pw) + J.motorImp + J.impZ);
            vB = vadd(vB, vscale(J.mB, Pw));
            wB += J.iB * (vcross(J.rB, Pw) + J.motorImp + J.impZ);
            vel[0][jA] = vA.x; vel[1][jA] = vA.y; vel[2][jA] = wA;
            vel[0][jB] = vB.x; vel[1][jB] = vB.y; vel[2][jB] = wB;
        }
        __syncthreads();
    }
    // ---------------- velocity iterations: modulo schedule, one s_barrier per slot ----------------
    {
        int nextJ = (jrole && iters > 0) ? jround : 0x7fffffff, leftJ = iters;
        int nextC = (crole && iters > 0) ? offC : 0x7fffffff, leftC = iters;
        int ph = 0;
        for (int tick = 0; tick < nTicks; ++tick) {
            if (tick == nextJ) {
                nextJ = (--leftJ > 0) ? nextJ + P : 0x7fffffff;
                V2 vA = mk(vel[0][jA], vel[1][jA]);
                float wA = vel[2][jA];
                V2 vB = mk(vel[0][jB], vel[1][jB]);
                float wB = vel[2][jB];
                joint_solve_velocity(J, vA, wA, vB, wB);
                vel[0][jA] = vA.x; vel[1][jA] = vA.y; vel[2][jA] = wA;
                vel[0][jB] = vB.x; vel[1][jB] = vB.y; vel[2][jB] = wB;
            }
            __syncthreads();
            if (P > 32 || ((cPhases >> ph) & 1)) { // workgroup-uniform: some body has its contact slot at this phase
                if (tick == nextC) {
                    nextC = (--leftC > 0) ? nextC + P : 0x7fffffff;
                    float cvx = vel[0][cBody], cvy = vel[1][cBody], cw = vel[2][cBody];
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
                    vel[0][cBody] = cvx; vel[1][cBody] = cvy; vel[2][cBody] = cw;
                }
                __syncthreads();
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
    if (jrole) {
        const unsigned gl = glJ;
        LF(L_JIMPX) = J.impX; LF(L_JIMPY) = J.impY; LF(L_JIMPZ) = J.impZ; LF(L_JMOTORIMP) = J.motorImp;
    }
    if (solve) { // the last slot ended with a barrier
        const unsigned gl = wg0 + tid;
        LF(L_VX) = vel[0][tid]; LF(L_VY) = vel[1][tid]; LF(L_W) = vel[2][tid];
    }
}

__global__ __launch_bounds__(VEL_THREADS) void rem2d_vel_kernel(State S, VelArgs A) {
    __shared__ VelShared sh;
    vel_body(S, A, blockIdx.x, sh);
}
// several worlds in one launch: blockEnd counts VEL_THREADS-wide blocks here
struct VelBatch {
    State S[REM2D_MAX_BATCH];
    unsigned blockEnd[REM2D_MAX_BATCH];
    int lanes[REM2D_MAX_BATCH];
    int n;
};
__global__ __launch_bounds__(VEL_THREADS) void rem2d_vel_multi_kernel(VelBatch B, VelArgs A) {
    __shared__ VelShared sh;
    unsigned block = blockIdx.x;
    int b = 0;
    while (b + 1 < B.n && block >= B.blockEnd[b]) ++b;
    if (b > 0) block -= B.blockEnd[b - 1];
    VelArgs Ab = A;
    Ab.K = B.lanes[b];
    vel_body(B.S[b], Ab, block, sh);
}

// ---------------------------------------------------------------------------------------------------
// post: position integration / iterations, sleep, broadphase refresh, bookkeeping
// ---------------------------------------------------------------------------------------------------
template <int K>
DEV void post_body(const State &S, const Terrain &T, const StepArgs &A, unsigned block, PosShared &psh) {
    const int lane = threadIdx.x;
    const unsigned gl = block * WAVE + lane;
    const unsigned env = gl / K;
    const int base = lane & ~(K - 1);
    const int sub = lane & (K - 1);
    const unsigned Lp = S.Lp;
    if (S.flags & REM2D_FLAG_SKIP_FROZEN) {
        if (__all(EI(E_FROZEN) != 0 ? 1 : 0)) return; // pre skipped this wavefront too
    }

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
    const int jround = hasJoint ? (LI(L_JROUND) & 0xff) : -1;
    const int limitState = LI(L_JLIMIT);
    const float mA = __shfl(mB, pl), iA = __shfl(iB, pl);
    const int nRounds = wave_max(jround) + 1;
    const int misc = __float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + gl, 0));
    const int nTouch = misc & 0xff;
    const int envAwake = (misc >> 8) & 1;
    int err = 0, lastPosIters = EI(E_POSITERS);
    const float h = A.dt;
    const float inv_dt = h > 0.0f ? 1.0f / h : 0.0f;
    const bool sleepResetAlways = (S.flags & REM2D_FLAG_SLEEP_RESET_ALWAYS) != 0;
    const bool allowSleep = (S.flags & REM2D_FLAG_NO_SLEEP) == 0;
    const float c0x = px, c0y = py, a0 = ang; // sweep start (b2Island::Solve: c0 = c, a0 = a)
    const float hx = LF(L_HX), hy = LF(L_HY);
    const float radiusB = shape == SHAPE_CIRCLE ? hx : B2_POLYGON_RADIUS;

    if (envAwake) {
        float motorMass = iA + iB;
        if (motorMass > 0.0f) motorMass = 1.0f / motorMass;
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
        solve_positions_pipelined<K>(S, psh, gl, lane, pl, active, hasJoint, jround, (LI(L_JROUND) >> 16) & 0xff, nTouch, mA, iA, mB,
                                     iB, radiusB, limitState, motorMass, A.posIters, px, py, ang, envSolved, itersUsed);
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
            V2 fatLo = mk(LF(L_FATLX), LF(L_FATLY)), fatHi = mk(LF(L_FATUX), LF(L_FATUY));
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
        const unsigned wb = (unsigned)SCR_SWEEP_BASE * Lp + gl;
        SW(wb, 0) = c0x; SW(wb, 1) = c0y; SW(wb, 2) = a0;
    }
    LF(L_PX) = px; LF(L_PY) = py; LF(L_ANG) = ang; LF(L_VX) = vx; LF(L_VY) = vy; LF(L_W) = w;
    LF(L_SLEEPT) = sleepT; LI(L_AWAKE) = awake;
    LI(L_CCOUNT) = cCount;
    int envErr = group_or<K>(err);
    if (sub == 0) {
        if (h > 0.0f) EF(E_INVDT0) = inv_dt;
        if (envErr) EI(E_ERR) = EI(E_ERR) | envErr;
        EI(E_POSITERS) = lastPosIters;
    }
    if (!A.defer) env_bookkeeping(S, env, sub, __shfl(px, base));
}

template <int K>
__global__ __launch_bounds__(WAVE) void rem2d_post_kernel(State S, Terrain T, StepArgs A) {
    __shared__ PosShared psh;
    post_body<K>(S, T, A, blockIdx.x, psh);
}
__global__ __launch_bounds__(WAVE) void rem2d_post_multi_kernel(Batch B, StepArgs A) {
    __shared__ PosShared psh;
    unsigned block = blockIdx.x;
    const int b = batch_find(B, block);
    BATCH_DISPATCH(post_body, psh)
}

#endif
