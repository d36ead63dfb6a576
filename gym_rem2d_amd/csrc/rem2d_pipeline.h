// rem2d_pipeline.h -- the two body-per-lane ends of the tile pipeline: rem2d_pre_multi_kernel -> (rem2d_vel4_kernel,
// rem2d_vel4.h) -> rem2d_post_multi_kernel.  Part of the single translation unit rem2d.hip; not a stand-alone header.
//
// Same arithmetic as rem2d_step_multi_kernel, cut at the two ends of the velocity phase of b2Island::Solve so that the 180
// velocity iterations run in a kernel of their own whose lanes are constraints:
//
//   pre  (lane = body, workgroup = wave64): controller / PID, Collide, velocity integration, contact and
//        joint constraint set-up.  Hands over through HBM: integrated velocities (L_VX..), dtRatio-scaled
//        warm-start impulses (L_JIMP*, contact records), limit state, one contact record per touching
//        manifold (scratch, cc_store layout), joint lever arms, nTouch / pair-slot map.
//   post (lane = body, workgroup = wave64): position integration, position iterations, sleep,
//        SynchronizeFixtures / FindNewContacts, per-step bookkeeping or hand-over to the TOI kernels.
//
// Cutting the step also cuts the register footprint: pre runs at 3, post at 5 waves per SIMD where the fused kernel
// (232 VGPRs) runs at 2 -- this path is bound by dependent-instruction latency, so resident wavefronts are what counts.
#ifndef REM2D_PIPELINE_H
#define REM2D_PIPELINE_H


// ---------------------------------------------------------------------------------------------------
// pre: Modular2D.step's controller sweep, b2World::Step up to (not including) the warm start
// ---------------------------------------------------------------------------------------------------
// INTILE (rem2d_step_train_kernel: pre, the velocity tile, the position block and the TOI solve of the same 64 lanes by one
// workgroup): the lane group takes the creature the velocity tile and the position block of this workgroup take -- slot s of the
// grid is creature order[s] -- and has no grid-wide chores (no TOI work list; REM2D_FLAG_RETILE worlds, whose order is copied by
// this kernel for the whole grid, keep the launches of their own).
template <int K, bool INTILE = false>
DEV void pre_body(const State &S, const Terrain &T, const StepArgs &A, unsigned block) {
    const int lane = threadIdx.x;
    const int base = lane & ~(K - 1);
    const int sub = lane & (K - 1);
    const unsigned slot = (block * WAVE + lane) / K;
    const unsigned env = (INTILE && (S.flags & REM2D_STATE_ORDERED)) ? (unsigned)S.order[slot] : slot;
    const unsigned gl = env * K + sub;
    const unsigned Lp = S.Lp;
    if (!INTILE) {
        if (gl == 0) S.toiWork[0] = 0; // work list of the TOI kernels that follow
        // creature order of the post kernel: what post wrote in the last step becomes what it reads in this one
        if (S.flags & REM2D_FLAG_RETILE) {
            if (gl < S.Np) S.order[gl] = S.order[S.Np + gl];
            if (gl == 0) { S.order[2 * S.Np] = 0; S.order[2 * S.Np + 1] = 0; }
        }
    }
    if (S.flags & REM2D_FLAG_SKIP_FROZEN) { // evaluate() has left its loop for every creature of this wavefront
        if (__all(EI(E_FROZEN) != 0 ? 1 : 0)) {
            const unsigned mb = (unsigned)SCR_MISC_BASE * Lp + gl;
            SW(mb, 0) = __int_as_float(0x200); // nothing to solve for the velocity kernel; 0x200: post and the TOI kernels skip this creature too
            return;
        }
    }

    const int shape = LI(L_SHAPE);
    const bool active = shape != SHAPE_NONE;
    const float mB = LF(L_INVM), iB = LF(L_INVI);
    const float px = LF(L_PX), py = LF(L_PY), ang = LF(L_ANG);
    // (the velocities, the joint's accumulated impulses and its limit state are loaded where they are first needed, after
    // the narrowphase: fewer values alive across it)
    float sleepT = LF(L_SLEEPT);
    int awake = LI(L_AWAKE);
    int cCount = LI(L_CCOUNT);
    const int parent = LI(L_PARENT);
    const bool hasJoint = active && parent >= 0;
    const int pl = base + (parent >= 0 ? parent : 0);
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
    slotpack_t slotPack = 0;
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
                    slotPack |= SP_PUT(s, nTouch);
                    ++nTouch;
                } else {
                    err |= REM2D_ERR_SOLVER_OVERFLOW;
                }
            }
            ++s;
        }
    }
    // =============== b2World::Solve (first part) ===============
    float vx = LF(L_VX), vy = LF(L_VY), w = LF(L_W);
    float impX = LF(L_JIMPX), impY = LF(L_JIMPY), impZ = LF(L_JIMPZ), motorImp = LF(L_JMOTORIMP);
    int limitState = LI(L_JLIMIT);
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
            unsigned o = SP_GET(slotPack, t) * Lp + gl;
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
        sp_store(S, mb, slotPack);
    }
    int envErr = group_or<K>(err);
    if (sub == 0) {
        EI(E_NEWFIX) = 0;
        if (envErr) EI(E_ERR) = EI(E_ERR) | envErr;
    }
}

// WPS = 4: four wavefronts per SIMD (128 VGPRs, no spills since the velocities and the joint's impulses are loaded after the
// narrowphase; with them alive across it the kernel took 137).  The kernel is short and every step of its group waits for
// it; at three per SIMD the wavefronts of a mixed population's launch queue behind the other groups' kernels: +2.0 % on
// config 3 (49.9 -> 50.9 M), +0.8 % on config 4.  WPS = 3 for the wider tile shapes: the 65 536 8-module chains are 1.4 %
// faster with it (186.6 vs 184.0 M) -- fewer of its wavefronts at once beside the 3-per-SIMD velocity kernel of that shape.
template <int WPS>
__global__ __launch_bounds__(WAVE, WPS) void rem2d_pre_multi_kernel(Batch B, StepArgs A) {
    unsigned block = blockIdx.x;
    const int b = batch_find(B, block);
    BATCH_DISPATCH(pre_body)
}

DEV int wave_sum(int v) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) v += __shfl_xor(v, o);
    return __builtin_amdgcn_readfirstlane(v);
}
DEV int wave_or(int v) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) v |= __shfl_xor(v, o);
    return __builtin_amdgcn_readfirstlane(v);
}

// ---------------------------------------------------------------------------------------------------
// post: position integration / iterations, sleep, broadphase refresh, bookkeeping
// ---------------------------------------------------------------------------------------------------
// Dynamic re-tiling.  The position iterations are bimodal: most creatures pass Box2D's tolerance test after one or two
// iterations, 10-15 % (a joint at its limit pressed against the ground) never pass it and use all of them -- and they are
// the same creatures from step to step.  A wavefront runs as many iterations as its slowest creature, and with the
// creatures in morphology order 60-78 % of the wavefronts hold one.  Creatures are independent and every per-lane access
// goes through `gl`, so the wavefronts of this kernel may take the creatures in ANY order: slot s of the launch (lane
// group s of the grid) handles creature order[s].  Every step the kernel deals the creatures out again for the next one
// -- those that used every iteration from the FRONT of the order (their wavefronts are the long ones: they must be
// dispatched first), the others from the back (one atomic per wavefront and class) -- so that ~12 % of the wavefronts
// run 60 iterations instead of ~70 %.  Same arithmetic on the same operands: same bits, whatever order the atomics
// produce.  Measured (profiles/archive/r03_retile.txt): -34 % VALU wave-instructions in this kernel (61.9 -> 41.1 M per launch),
// but a wavefront of eight such creatures with different periods runs a contact AND a joint section in every tick, so
// the slowest wavefront -- the kernel's duration -- gets longer: +4.4 % env-steps/s where the chip's instruction issue
// is the limit (131 072 creatures per GPU, config 5's share), -5 % where the chain of kernels is (65 536).  Hence a
// property of the world (REM2D_FLAG_RETILE) that the host sets for large populations.
DEV void post_place(const State &S, int lane, int K, unsigned env, bool leader, bool slow) {
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (WAVE - lane));
    const unsigned long long mf = __ballot(leader && !slow), ms = __ballot(leader && slow);
    int bf = 0, bs = 0;
    if (lane == 0) {
        if (mf) bf = atomicAdd(&S.order[2 * S.Np], __popcll(mf));
        if (ms) bs = atomicAdd(&S.order[2 * S.Np + 1], __popcll(ms));
    }
    bf = __shfl(bf, 0); bs = __shfl(bs, 0);
    if (leader) {
        const int pos = slow ? bs + __popcll(ms & below) : (int)S.Np - 1 - (bf + __popcll(mf & below));
        S.order[S.Np + pos] = (int)env;
    }
}
// FUSED (rem2d_rest_multi_kernel): instead of queueing a body that needs the TOI solve on the world's work list, return
// true for its lane -- the caller solves it in this very wavefront.
template <int K, bool FUSED>
DEV bool post_body(const State &S, const Terrain &T, const StepArgs &A, unsigned block, PosShared &psh) {
    const int lane = threadIdx.x;
    const int base = lane & ~(K - 1);
    const int sub = lane & (K - 1);
    const unsigned slot = (block * WAVE + lane) / K;
    const bool retile = (S.flags & REM2D_FLAG_RETILE) != 0;       // the kernel deals the creatures anew for the next step
    const bool ordered = retile || (S.flags & REM2D_STATE_ORDERED) != 0;
    const unsigned env = ordered ? (unsigned)S.order[slot] : slot; // the creature this lane group handles in this step
    const unsigned gl = env * K + sub;
    const unsigned Lp = S.Lp;
    const int misc = __float_as_int(SW((unsigned)SCR_MISC_BASE * Lp + gl, 0));
    // REM2D_FLAG_SKIP_FROZEN: pre skips a wavefront (in ITS order) whose creatures have all finished and marks them;
    // those creatures are nobody's business in this step, whichever wavefront of this kernel they ride in
    const bool skipped = (misc & 0x200) != 0;
    if (__all(skipped ? 1 : 0)) {
        if (retile) post_place(S, lane, K, env, sub == 0, false);
        return false;
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
    int envErr = group_or<K>(err);
    const float rootx = __shfl(px, base);
    if (retile) post_place(S, lane, K, env, sub == 0, envAwake && lastPosIters >= A.posIters && A.posIters > 2);
    if (skipped) return false; // (a creature pre left alone: nothing of it changes in this step)
    if (A.defer) { // continuous physics: the TOI kernel needs the sweep start and finishes the step
        const unsigned wb = (unsigned)SCR_SWEEP_BASE * Lp + gl;
        SW(wb, 0) = c0x; SW(wb, 1) = c0y; SW(wb, 2) = a0;
    }
    LF(L_PX) = px; LF(L_PY) = py; LF(L_ANG) = ang; LF(L_VX) = vx; LF(L_VY) = vy; LF(L_W) = w;
    LF(L_SLEEPT) = sleepT; LI(L_AWAKE) = awake;
    LI(L_CCOUNT) = cCount;
    if (sub == 0) {
        if (h > 0.0f) EF(E_INVDT0) = inv_dt;
        if (envErr) EI(E_ERR) = EI(E_ERR) | envErr;
        EI(E_POSITERS) = lastPosIters;
    }
    if (!A.defer) env_bookkeeping(S, env, sub, rootx);
    // continuous physics: the TOI scan of this body, with its pose and sweep start still in registers (it was a kernel
    // of its own: one launch, one grid of early-exits and a round trip of the sweep start through HBM less per step)
    else if (A.defer == 2) {
        if (!FUSED) {
            toi_scan_lane(S, T, A.dt, gl, env, sub, shape, px, py, ang, c0x, c0y, a0, hx, hy, awake, cCount);
        } else {
            const bool heavy = toi_scan_heavy(S, T, A.dt, gl, shape, px, py, ang, c0x, c0y, a0, hx, hy, awake, cCount);
            if (!heavy && sub == 0) env_bookkeeping(S, env, 0, px);
            return heavy;
        }
    }
    return false;
}
template <int K> DEV void post_only_body(const State &S, const Terrain &T, const StepArgs &A, unsigned block, PosShared &psh) {
#ifdef REM2D_V4_PROBES // diagnostic (tools/chain_probe.py): this wavefront's time in the kernel, in the SECOND creature of its block
    const unsigned long long rEntry = __builtin_amdgcn_s_memrealtime();
#endif
    (void)post_body<K, false>(S, T, A, block, psh);
#ifdef REM2D_V4_PROBES
    if (threadIdx.x == 0 && 2 * K <= WAVE && !(S.flags & (REM2D_FLAG_RETILE | REM2D_STATE_ORDERED))) {
        const unsigned env = block * (WAVE / K) + 1;
        if (env < S.Np) EI(E_TOIEVENTS) = (int)(__builtin_amdgcn_s_memrealtime() - rEntry);
    }
#endif
}

__global__ __launch_bounds__(WAVE) void rem2d_post_multi_kernel(Batch B, StepArgs A) {
    __shared__ PosShared psh;
    unsigned block = blockIdx.x;
    const int b = batch_find(B, block);
    BATCH_DISPATCH(post_only_body, psh)
}

// ---------------------------------------------------------------------------------------------------
// Creature order by current cost, made on the device (REM2D_OPT_REBALANCE): a STABLE counting sort of the world's creatures
// by cost class -- those that used every position iteration in the last step first, in their static order, then those that used
// three or more, then the rest, each class in its static order -- written to both halves of State::order (what rem2d_world_set_order would install).  One workgroup per
// world: every thread counts the classes of its contiguous chunk, the counts are scanned in LDS, every thread writes its
// chunk's creatures to their places; launched every N env-steps in front of `pre`.
// ---------------------------------------------------------------------------------------------------
// One wavefront for a world of up to ~16 000 creatures (it starts as soon as any wavefront slot is free; 64 / 256 / 1 024 threads
// measured the same at config 3's world sizes), more for bigger worlds so that a thread's chunk stays at <= 256 creatures
// (a 262 144-creature world of a 1 M-individual generation: 1 024 threads).
#define REBALANCE_MAX_THREADS 1024
#ifdef REBALANCE_VEL
#define REBALANCE_CLASSES (3 * REBALANCE_VEL)
#endif
#ifndef REBALANCE_CLASSES
#define REBALANCE_CLASSES 3 // (three classes under the step train: +0.7 % on config 3 over two, config 4 unchanged; profiles/r05_step_train.txt)
#endif
// cost class of a creature from the position iterations of its last step: 0 = used all of them (the long blocks), ...
DEV int rebalance_class(int positers, int posIters) {
#if REBALANCE_CLASSES == 2
    return positers >= posIters ? 0 : 1;

#elif REBALANCE_CLASSES == 3
    return positers >= posIters ? 0 : (positers >= 3 ? 1 : 2);
#else
    return positers >= posIters ? 0 : (positers >= 8 ? 1 : (positers >= 3 ? 2 : 3));
#endif
}
__global__ __launch_bounds__(REBALANCE_MAX_THREADS) void rem2d_rebalance_kernel(State S, int posIters) {
    __shared__ int cnt[REBALANCE_CLASSES][REBALANCE_MAX_THREADS];
    const int n = (int)S.nEnvs, T = (int)blockDim.x, t = threadIdx.x;
    const int per = (n + T - 1) / T;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    int c[REBALANCE_CLASSES];
#pragma unroll
    for (int k = 0; k < REBALANCE_CLASSES; ++k) c[k] = 0;
#ifdef REBALANCE_VEL
    // (experiment, round 6: the cost of the VELOCITY tile as a second key -- a tile pays, per iteration, as many contact sub-slots as
    // its most loaded creature: the most manifolds on one body (phase 0 after the rotation) plus one tick per further touching body.
    // Read from the last step's `pre` hand-over words of the creature's lanes.)
    const int K = (int)(S.Lp / S.Np);
    auto key_of = [&](int e) -> int {
        const unsigned env = (unsigned)e;
        int maxT = 0, nTouching = 0;
        for (int l = 0; l < K; ++l) {
            const int m = __float_as_int(SW((unsigned)SCR_MISC_BASE * S.Lp + (unsigned)(e * K + l), 0));
            const int nT = (m & 0x100) ? (m & 0xff) : 0;
            maxT = max(maxT, nT);
            nTouching += nT > 0 ? 1 : 0;
        }
        const int vel = min(REBALANCE_VEL - 1, maxT + min(nTouching > 0 ? nTouching - 1 : 0, 2)); // 0 .. REBALANCE_VEL - 1
        const int pos = EI(E_POSITERS) >= posIters ? 0 : (EI(E_POSITERS) >= 3 ? 1 : 2);
        return pos * REBALANCE_VEL + (REBALANCE_VEL - 1 - vel); // (the expensive ones first inside a position class)
    };
#define rebalance_class(key, posIters) (key)
#else
    auto key_of = [&](int e) -> int {
        const unsigned env = (unsigned)e;
        return EI(E_POSITERS);
    };
#endif
    for (int e = lo; e < hi; ++e) {
        const int cls = rebalance_class(key_of(e), posIters);
#pragma unroll
        for (int k = 0; k < REBALANCE_CLASSES; ++k) c[k] += cls == k ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < REBALANCE_CLASSES; ++k) cnt[k][t] = c[k];
    __syncthreads();
    for (int o = 1; o < T; o <<= 1) { // inclusive scans (Hillis-Steele), one per class
        int v[REBALANCE_CLASSES];
#pragma unroll
        for (int k = 0; k < REBALANCE_CLASSES; ++k) v[k] = t >= o ? cnt[k][t - o] : 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < REBALANCE_CLASSES; ++k) cnt[k][t] += v[k];
        __syncthreads();
    }
    // a stable counting sort: class k starts behind all creatures of the classes before it; inside a class, chunk by chunk
    int pos[REBALANCE_CLASSES], base = 0;
#pragma unroll
    for (int k = 0; k < REBALANCE_CLASSES; ++k) {
        pos[k] = base + cnt[k][t] - c[k];
        base += cnt[k][T - 1];
    }
    for (int e = lo; e < hi; ++e) {
        const int cls = rebalance_class(key_of(e), posIters);
        int p = 0;
#pragma unroll
        for (int k = 0; k < REBALANCE_CLASSES; ++k)
            if (cls == k) p = pos[k]++;
        S.order[p] = e;
        S.order[S.Np + p] = e;
    }
    for (int e = n + t; e < (int)S.Np; e += T) { S.order[e] = e; S.order[S.Np + e] = e; } // padding creatures keep their slots
#ifdef REBALANCE_VEL
#undef rebalance_class
#endif
}

// (Round 3 built "rest": post + the TOI solve of the wavefront's own bodies + the next step's pre in one launch, two
// launches per step instead of four.  Bit-exact, but it needs the TOI solve's 256 VGPRs and was slower on every workload:
// profiles/archive/r03_fused_rest.txt.  Removed; post_body keeps the FUSED hook it used.)

#endif
