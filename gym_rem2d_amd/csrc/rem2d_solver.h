// rem2d_solver.h -- pair lists (b2ContactManager) and the per-contact pieces of b2ContactSolver.
// Part of the single translation unit rem2d.hip (see its header comment); not a stand-alone header.
#ifndef REM2D_SOLVER_H
#define REM2D_SOLVER_H

// =====================================================================================
// per-lane solver structures (registers)
// =====================================================================================
struct ContactC { // one touching contact (terrain A static, body B = this lane)
    V2 normal;
    V2 rB0, rB1;
    float nm0, nm1, tm0, tm1; // normalMass / tangentMass per point
    float n0, n1, t0, t1;     // accumulated impulses
    float k11, k12, k22;      // K
    float i11, i12, i22;      // normalMass = K^-1 (symmetric)
    int count;                // solver point count (block solver may drop to 1)
};
#define CC_WORDS 21

// b2ContactManager::AddPair: head-insert into the body's pair list
DEV void pairs_insert_front(const State &S, unsigned gl, int &count, int edge, int &err) {
    if (count >= KC) { err |= REM2D_ERR_PAIR_OVERFLOW; return; }
    for (int s = count; s > 0; --s) {
        unsigned d = (unsigned)s * S.Lp + gl, f = (unsigned)(s - 1) * S.Lp + gl;
        CI(C_EDGE, d) = CI(C_EDGE, f);
        CI(C_INFO, d) = CI(C_INFO, f);
        CU(C_KEY0, d) = CU(C_KEY0, f);
        CU(C_KEY1, d) = CU(C_KEY1, f);
        CF(C_N0, d) = CF(C_N0, f);
        CF(C_N1, d) = CF(C_N1, f);
        CF(C_T0, d) = CF(C_T0, f);
        CF(C_T1, d) = CF(C_T1, f);
    }
    CI(C_EDGE, gl) = edge;
    CI(C_INFO, gl) = 1 << 16; // e_enabledFlag (CI_ENABLED)
    CU(C_KEY0, gl) = 0u;
    CU(C_KEY1, gl) = 0u;
    CF(C_N0, gl) = 0.0f;
    CF(C_N1, gl) = 0.0f;
    CF(C_T0, gl) = 0.0f;
    CF(C_T1, gl) = 0.0f;
    ++count;
}
DEV void pairs_remove(const State &S, unsigned gl, int &count, int s) {
    for (int k = s; k + 1 < count; ++k) {
        unsigned d = (unsigned)k * S.Lp + gl, f = (unsigned)(k + 1) * S.Lp + gl;
        CI(C_EDGE, d) = CI(C_EDGE, f);
        CI(C_INFO, d) = CI(C_INFO, f);
        CU(C_KEY0, d) = CU(C_KEY0, f);
        CU(C_KEY1, d) = CU(C_KEY1, f);
        CF(C_N0, d) = CF(C_N0, f);
        CF(C_N1, d) = CF(C_N1, f);
        CF(C_T0, d) = CF(C_T0, f);
        CF(C_T1, d) = CF(C_T1, f);
    }
    --count;
    CI(C_EDGE, (unsigned)count * S.Lp + gl) = -1;
}
// b2BroadPhase::UpdatePairs for one moved body proxy: new pairs in ascending edge (= proxy id) order
DEV bool find_new_pairs(const State &S, const Terrain &T, unsigned gl, int &count, V2 flo, V2 fhi, int &err) {
    bool added = false;
    for (int st = 0; st < T.nPoly; ++st) { // hardcore boxes have the lowest proxy ids
        if (!aabb_overlap(mk(T.flx[st], T.fly[st]), mk(T.fux[st], T.fuy[st]), flo, fhi)) continue;
        bool exists = false;
        for (int s = 0; s < count; ++s) exists |= (CI(C_EDGE, (unsigned)s * S.Lp + gl) == st);
        if (!exists) {
            pairs_insert_front(S, gl, count, st, err);
            added = true;
        }
    }
    int lo = (int)floorf((flo.x - 0.25f - T.x0) * T.invPitch) - 1;
    int hi = (int)floorf((fhi.x + 0.25f - T.x0) * T.invPitch) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > T.nEdge - 1 ? T.nEdge - 1 : hi;
    for (int e = lo; e <= hi; ++e) {
        const int st = T.nPoly + e;
        if (!aabb_overlap(mk(T.flx[st], T.fly[st]), mk(T.fux[st], T.fuy[st]), flo, fhi)) continue;
        bool exists = false;
        for (int s = 0; s < count; ++s) exists |= (CI(C_EDGE, (unsigned)s * S.Lp + gl) == st);
        if (!exists) {
            pairs_insert_front(S, gl, count, st, err);
            added = true; // b2ContactManager::AddPair wakes both bodies
        }
    }
    return added;
}

// ---- b2ContactSolver pieces for one (static terrain, this body) constraint ----
// ctor + InitializeVelocityConstraints: b2WorldManifold with xfA = identity, radiusA = polygonRadius
DEV void contact_setup(ContactC &c, int mtype, int mcount, V2 ln, V2 lp, V2 p0, V2 p1, V2 cB, Rot q, float mB, float iB,
                       float radiusB, float n0, float t0, float n1, float t1) {
    c.count = mcount;
    c.n0 = n0; c.t0 = t0; c.n1 = n1; c.t1 = t1;
    V2 normal, w0, w1 = mk(0.0f, 0.0f);
    const float radiusA = B2_POLYGON_RADIUS;
    if (mtype == MF_CIRCLES) {
        normal = mk(1.0f, 0.0f);
        V2 pointA = lp;
        V2 pointB = xmul(q, cB, p0);
        if (vdist2(pointA, pointB) > B2_EPSILON * B2_EPSILON) {
            normal = vsub(pointB, pointA);
            vnormalize(normal);
        }
        V2 cA = vadd(pointA, vscale(radiusA, normal));
        V2 cBp = vsub(pointB, vscale(radiusB, normal));
        w0 = vscale(0.5f, vadd(cA, cBp));
    } else if (mtype == MF_FACE_A) {
        normal = ln;
        V2 planePoint = lp;
        V2 clip = xmul(q, cB, p0);
        V2 cA = vadd(clip, vscale(radiusA - vdot(vsub(clip, planePoint), normal), normal));
        V2 cBp = vsub(clip, vscale(radiusB, normal));
        w0 = vscale(0.5f, vadd(cA, cBp));
        if (mcount > 1) {
            clip = xmul(q, cB, p1);
            cA = vadd(clip, vscale(radiusA - vdot(vsub(clip, planePoint), normal), normal));
            cBp = vsub(clip, vscale(radiusB, normal));
            w1 = vscale(0.5f, vadd(cA, cBp));
        }
    } else {
        V2 nB = rmul(q, ln);
        V2 planePoint = xmul(q, cB, lp);
        V2 clip = p0;
        V2 cBp = vadd(clip, vscale(radiusB - vdot(vsub(clip, planePoint), nB), nB));
        V2 cA = vsub(clip, vscale(radiusA, nB));
        w0 = vscale(0.5f, vadd(cA, cBp));
        if (mcount > 1) {
            clip = p1;
            cBp = vadd(clip, vscale(radiusB - vdot(vsub(clip, planePoint), nB), nB));
            cA = vsub(clip, vscale(radiusA, nB));
            w1 = vscale(0.5f, vadd(cA, cBp));
        }
        normal = vneg(nB);
    }
    c.normal = normal;
    V2 tangent = vcross_vs(normal, 1.0f);
    c.rB0 = vsub(w0, cB);
    c.rB1 = vsub(w1, cB);
    {
        float rnB = vcross(c.rB0, normal);
        float kNormal = mB + iB * rnB * rnB;
        c.nm0 = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
        float rtB = vcross(c.rB0, tangent);
        float kTangent = mB + iB * rtB * rtB;
        c.tm0 = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
    }
    c.nm1 = c.tm1 = 0.0f;
    c.k11 = c.k12 = c.k22 = c.i11 = c.i12 = c.i22 = 0.0f;
    if (mcount > 1) {
        float rnB = vcross(c.rB1, normal);
        float kNormal = mB + iB * rnB * rnB;
        c.nm1 = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
        float rtB = vcross(c.rB1, tangent);
        float kTangent = mB + iB * rtB * rtB;
        c.tm1 = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
        float rn1B = vcross(c.rB0, normal), rn2B = vcross(c.rB1, normal);
        float k11 = mB + iB * rn1B * rn1B;
        float k22 = mB + iB * rn2B * rn2B;
        float k12 = mB + iB * rn1B * rn2B;
        const float k_maxConditionNumber = 1000.0f;
        if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12)) {
            c.k11 = k11; c.k12 = k12; c.k22 = k22;
            float det = k11 * k22 - k12 * k12;
            if (det != 0.0f) det = 1.0f / det;
            c.i11 = det * k22;
            c.i12 = -det * k12;
            c.i22 = det * k11;
        } else {
            c.count = 1;
        }
    }
}
// WarmStart
DEV void contact_warm_start(const ContactC &c, float mB, float iB, float &vx, float &vy, float &w) {
    V2 normal = c.normal, tangent = vcross_vs(normal, 1.0f);
    V2 P = vadd(vscale(c.n0, normal), vscale(c.t0, tangent));
    w += iB * vcross(c.rB0, P);
    vx += mB * P.x; vy += mB * P.y;
    if (c.count > 1) {
        P = vadd(vscale(c.n1, normal), vscale(c.t1, tangent));
        w += iB * vcross(c.rB1, P);
        vx += mB * P.x; vy += mB * P.y;
    }
}
// SolveVelocityConstraints for one contact (friction first, then normal / 2-point block LCP)
DEV void contact_solve(ContactC &c, float mB, float iB, float friction, float &vx, float &vy, float &w) {
    V2 normal = c.normal, tangent = vcross_vs(normal, 1.0f);
    V2 vB = mk(vx, vy);
    float wB = w;
    {
        V2 dv = vadd(vB, vcross_sv(wB, c.rB0));
        float vt = vdot(dv, tangent) - 0.0f;
        float lambda = c.tm0 * (-vt);
        float maxFriction = friction * c.n0;
        float newImpulse = fclamp(c.t0 + lambda, -maxFriction, maxFriction);
        lambda = newImpulse - c.t0;
        c.t0 = newImpulse;
        V2 P = vscale(lambda, tangent);
        vB = vadd(vB, vscale(mB, P));
        wB += iB * vcross(c.rB0, P);
    }
    if (c.count > 1) {
        V2 dv = vadd(vB, vcross_sv(wB, c.rB1));
        float vt = vdot(dv, tangent) - 0.0f;
        float lambda = c.tm1 * (-vt);
        float maxFriction = friction * c.n1;
        float newImpulse = fclamp(c.t1 + lambda, -maxFriction, maxFriction);
        lambda = newImpulse - c.t1;
        c.t1 = newImpulse;
        V2 P = vscale(lambda, tangent);
        vB = vadd(vB, vscale(mB, P));
        wB += iB * vcross(c.rB1, P);
    }
    if (c.count == 1) {
        V2 dv = vadd(vB, vcross_sv(wB, c.rB0));
        float vn = vdot(dv, normal);
        float lambda = -c.nm0 * (vn - 0.0f);
        float newImpulse = fmax32(c.n0 + lambda, 0.0f);
        lambda = newImpulse - c.n0;
        c.n0 = newImpulse;
        V2 P = vscale(lambda, normal);
        vB = vadd(vB, vscale(mB, P));
        wB += iB * vcross(c.rB0, P);
    } else {
        V2 a = mk(c.n0, c.n1);
        V2 dv1 = vadd(vB, vcross_sv(wB, c.rB0));
        V2 dv2 = vadd(vB, vcross_sv(wB, c.rB1));
        float vn1 = vdot(dv1, normal), vn2 = vdot(dv2, normal);
        V2 b = mk(vn1 - 0.0f, vn2 - 0.0f);
        b = vsub(b, mk(c.k11 * a.x + c.k12 * a.y, c.k12 * a.x + c.k22 * a.y));
        V2 x;
        bool solved = false;
        x = vneg(mk(c.i11 * b.x + c.i12 * b.y, c.i12 * b.x + c.i22 * b.y)); // case 1
        solved = x.x >= 0.0f && x.y >= 0.0f;
        if (!solved) { // case 2
            x.x = -c.nm0 * b.x;
            x.y = 0.0f;
            vn2 = c.k12 * x.x + b.y;
            solved = x.x >= 0.0f && vn2 >= 0.0f;
        }
        if (!solved) { // case 3
            x.x = 0.0f;
            x.y = -c.nm1 * b.y;
            vn1 = c.k12 * x.y + b.x;
            solved = x.y >= 0.0f && vn1 >= 0.0f;
        }
        if (!solved) { // case 4
            x.x = 0.0f;
            x.y = 0.0f;
            solved = b.x >= 0.0f && b.y >= 0.0f;
        }
        if (solved) {
            V2 d = vsub(x, a);
            V2 P1 = vscale(d.x, normal), P2 = vscale(d.y, normal);
            vB = vadd(vB, vscale(mB, vadd(P1, P2)));
            wB += iB * (vcross(c.rB0, P1) + vcross(c.rB1, P2));
            c.n0 = x.x;
            c.n1 = x.y;
        }
    }
    vx = vB.x; vy = vB.y; w = wB;
}
// ---- the same solve by FOUR lanes (a quad: lanes 4k .. 4k+3, all of them active and holding the same constraint, body
// constants and angular velocity).  One wavefront issues one instruction per ~6 cycles whatever the dependencies, so a
// chain of solves costs its instruction COUNT; the lanes of a body's group are idle copies anyway (TOI kernel) -- here
// they split the 2-vector arithmetic: lane q works on component (q & 1) (x / y) and, in the 2-point block solve, on point
// (q >> 1).  Every scalar below is the serial form's expression on the same operands (a - b c == a + b (-c), the two
// products of a dot / cross product added in either order: exact identities of IEEE arithmetic), so accumulated impulses
// and velocities come out with the same bits.  v: this lane's velocity component; w, the impulses: the same in all four.
struct QuadRole { float tq, nq, r0c, r1c; bool isY, pt1; };
DEV float quad_swap1(float x) { // the value of lane ^ 1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false)); // quad_perm [1,0,3,2]
}
DEV float quad_swap2(float x) { // the value of lane ^ 2
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false)); // quad_perm [2,3,0,1]
}
DEV QuadRole quad_role(const ContactC &c, int lane) {
    QuadRole r;
    r.isY = (lane & 1) != 0;
    r.pt1 = (lane & 2) != 0;
    r.nq = r.isY ? c.normal.y : c.normal.x;
    r.tq = r.isY ? -c.normal.x : c.normal.y;   // tangent = cross(normal, 1) = (n.y, -n.x)
    r.r0c = r.isY ? c.rB0.x : -c.rB0.y;        // cross(w, r) = (-w r.y, w r.x);  cross(r, P) = r.x P.y - r.y P.x
    r.r1c = r.isY ? c.rB1.x : -c.rB1.y;
    return r;
}
DEV void contact_solve_quad(ContactC &c, const QuadRole &r, float mB, float iB, float friction, float &v, float &w) {
    {
        float dv = v + w * r.r0c;
        float p = dv * r.tq;
        float vt = p + quad_swap1(p);
        float lambda = c.tm0 * (-vt);
        float maxFriction = friction * c.n0;
        float newImpulse = fclamp(c.t0 + lambda, -maxFriction, maxFriction);
        lambda = newImpulse - c.t0;
        c.t0 = newImpulse;
        float P = lambda * r.tq;
        v = v + mB * P;
        float cq = r.r0c * P;
        w += iB * (cq + quad_swap1(cq));
    }
    if (c.count > 1) {
        float dv = v + w * r.r1c;
        float p = dv * r.tq;
        float vt = p + quad_swap1(p);
        float lambda = c.tm1 * (-vt);
        float maxFriction = friction * c.n1;
        float newImpulse = fclamp(c.t1 + lambda, -maxFriction, maxFriction);
        lambda = newImpulse - c.t1;
        c.t1 = newImpulse;
        float P = lambda * r.tq;
        v = v + mB * P;
        float cq = r.r1c * P;
        w += iB * (cq + quad_swap1(cq));
    }
    if (c.count == 1) {
        float dv = v + w * r.r0c;
        float p = dv * r.nq;
        float vn = p + quad_swap1(p);
        float lambda = -c.nm0 * vn;
        float newImpulse = fmax32(c.n0 + lambda, 0.0f);
        lambda = newImpulse - c.n0;
        c.n0 = newImpulse;
        float P = lambda * r.nq;
        v = v + mB * P;
        float cq = r.r0c * P;
        w += iB * (cq + quad_swap1(cq));
    } else {
        V2 a = mk(c.n0, c.n1);
        // lanes 0,1: point 0; lanes 2,3: point 1
        float dv = v + w * (r.pt1 ? r.r1c : r.r0c);
        float p = dv * r.nq;
        float vnOwn = p + quad_swap1(p);
        float vnOther = quad_swap2(vnOwn);
        float vn1 = r.pt1 ? vnOther : vnOwn, vn2 = r.pt1 ? vnOwn : vnOther;
        V2 b = mk(vn1, vn2);
        b = vsub(b, mk(c.k11 * a.x + c.k12 * a.y, c.k12 * a.x + c.k22 * a.y));
        V2 x;
        bool solved = false;
        x = vneg(mk(c.i11 * b.x + c.i12 * b.y, c.i12 * b.x + c.i22 * b.y)); // case 1
        solved = x.x >= 0.0f && x.y >= 0.0f;
        if (!solved) { // case 2
            x.x = -c.nm0 * b.x;
            x.y = 0.0f;
            vn2 = c.k12 * x.x + b.y;
            solved = x.x >= 0.0f && vn2 >= 0.0f;
        }
        if (!solved) { // case 3
            x.x = 0.0f;
            x.y = -c.nm1 * b.y;
            vn1 = c.k12 * x.y + b.x;
            solved = x.y >= 0.0f && vn1 >= 0.0f;
        }
        if (!solved) { // case 4
            x.x = 0.0f;
            x.y = 0.0f;
            solved = b.x >= 0.0f && b.y >= 0.0f;
        }
        if (solved) {
            V2 d = vsub(x, a);
            float P1 = d.x * r.nq, P2 = d.y * r.nq;
            v = v + mB * (P1 + P2);
            float c1 = r.r0c * P1, c2 = r.r1c * P2;
            w += iB * ((c1 + quad_swap1(c1)) + (c2 + quad_swap1(c2)));
            c.n0 = x.x;
            c.n1 = x.y;
        }
    }
}
// ---- the same solve by TWO lanes (a pair: lanes 2k, 2k + 1, both active, holding the same constraint; rem2d_vel4_kernel's
// contact sub-slots).  Lane q works on component (q & 1) of the 2-vectors, both points of the block solve; the scalar
// chain (impulse clamps, the 2 x 2 LCP) runs in both.  Same identities as contact_solve_quad: same bits.
DEV void contact_solve_pair(ContactC &c, const QuadRole &r, float mB, float iB, float friction, float &v, float &w) {
    {
        float dv = v + w * r.r0c;
        float p = dv * r.tq;
        float vt = p + quad_swap1(p);
        float lambda = c.tm0 * (-vt);
        float maxFriction = friction * c.n0;
        float newImpulse = fclamp(c.t0 + lambda, -maxFriction, maxFriction);
        lambda = newImpulse - c.t0;
        c.t0 = newImpulse;
        float P = lambda * r.tq;
        v = v + mB * P;
        float cq = r.r0c * P;
        w += iB * (cq + quad_swap1(cq));
    }
    if (c.count > 1) {
        float dv = v + w * r.r1c;
        float p = dv * r.tq;
        float vt = p + quad_swap1(p);
        float lambda = c.tm1 * (-vt);
        float maxFriction = friction * c.n1;
        float newImpulse = fclamp(c.t1 + lambda, -maxFriction, maxFriction);
        lambda = newImpulse - c.t1;
        c.t1 = newImpulse;
        float P = lambda * r.tq;
        v = v + mB * P;
        float cq = r.r1c * P;
        w += iB * (cq + quad_swap1(cq));
    }
    if (c.count == 1) {
        float dv = v + w * r.r0c;
        float p = dv * r.nq;
        float vn = p + quad_swap1(p);
        float lambda = -c.nm0 * vn;
        float newImpulse = fmax32(c.n0 + lambda, 0.0f);
        lambda = newImpulse - c.n0;
        c.n0 = newImpulse;
        float P = lambda * r.nq;
        v = v + mB * P;
        float cq = r.r0c * P;
        w += iB * (cq + quad_swap1(cq));
    } else {
        V2 a = mk(c.n0, c.n1);
        float p1 = (v + w * r.r0c) * r.nq, p2 = (v + w * r.r1c) * r.nq;
        float vn1 = p1 + quad_swap1(p1), vn2 = p2 + quad_swap1(p2);
        V2 b = mk(vn1, vn2);
        b = vsub(b, mk(c.k11 * a.x + c.k12 * a.y, c.k12 * a.x + c.k22 * a.y));
        V2 x;
        bool solved = false;
        x = vneg(mk(c.i11 * b.x + c.i12 * b.y, c.i12 * b.x + c.i22 * b.y)); // case 1
        solved = x.x >= 0.0f && x.y >= 0.0f;
        if (!solved) { // case 2
            x.x = -c.nm0 * b.x;
            x.y = 0.0f;
            vn2 = c.k12 * x.x + b.y;
            solved = x.x >= 0.0f && vn2 >= 0.0f;
        }
        if (!solved) { // case 3
            x.x = 0.0f;
            x.y = -c.nm1 * b.y;
            vn1 = c.k12 * x.y + b.x;
            solved = x.y >= 0.0f && vn1 >= 0.0f;
        }
        if (!solved) { // case 4
            x.x = 0.0f;
            x.y = 0.0f;
            solved = b.x >= 0.0f && b.y >= 0.0f;
        }
        if (solved) {
            V2 d = vsub(x, a);
            float P1 = d.x * r.nq, P2 = d.y * r.nq;
            v = v + mB * (P1 + P2);
            float c1 = r.r0c * P1, c2 = r.r1c * P2;
            w += iB * ((c1 + quad_swap1(c1)) + (c2 + quad_swap1(c2)));
            c.n0 = x.x;
            c.n1 = x.y;
        }
    }
}
// constraints beyond the register-resident ones live in handle scratch ([word][lane], coalesced)
DEV void cc_store(const State &S, unsigned cb, const ContactC &c) {
    SW(cb, 0) = c.normal.x; SW(cb, 1) = c.normal.y; SW(cb, 2) = c.rB0.x; SW(cb, 3) = c.rB0.y;
    SW(cb, 4) = c.rB1.x; SW(cb, 5) = c.rB1.y; SW(cb, 6) = c.nm0; SW(cb, 7) = c.nm1; SW(cb, 8) = c.tm0;
    SW(cb, 9) = c.tm1; SW(cb, 10) = c.n0; SW(cb, 11) = c.n1; SW(cb, 12) = c.t0; SW(cb, 13) = c.t1;
    SW(cb, 14) = c.k11; SW(cb, 15) = c.k12; SW(cb, 16) = c.k22; SW(cb, 17) = c.i11; SW(cb, 18) = c.i12;
    SW(cb, 19) = c.i22; SW(cb, 20) = __int_as_float(c.count);
}
DEV void cc_load(const State &S, unsigned cb, ContactC &c) {
    c.normal = mk(SW(cb, 0), SW(cb, 1)); c.rB0 = mk(SW(cb, 2), SW(cb, 3)); c.rB1 = mk(SW(cb, 4), SW(cb, 5));
    c.nm0 = SW(cb, 6); c.nm1 = SW(cb, 7); c.tm0 = SW(cb, 8); c.tm1 = SW(cb, 9);
    c.n0 = SW(cb, 10); c.n1 = SW(cb, 11); c.t0 = SW(cb, 12); c.t1 = SW(cb, 13);
    c.k11 = SW(cb, 14); c.k12 = SW(cb, 15); c.k22 = SW(cb, 16); c.i11 = SW(cb, 17); c.i12 = SW(cb, 18);
    c.i22 = SW(cb, 19); c.count = __float_as_int(SW(cb, 20));
}

#endif
