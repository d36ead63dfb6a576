// rem2d_toi.h -- b2Distance, b2TimeOfImpact and the per-lane b2World::SolveTOI loop.
// Part of the single translation unit rem2d.hip (see its header comment); not a stand-alone header.
#ifndef REM2D_TOI_H
#define REM2D_TOI_H

// =====================================================================================
// continuous collision: b2Distance (GJK), b2TimeOfImpact, and the per-body TOI event loop of
// b2World::SolveTOI.  Body A is always a static terrain shape (identity transform, time-
// transparent sweep), body B this lane's module, so every body's TOI sequence is independent.
// =====================================================================================
struct XF { V2 p; Rot q; };
DEV V2 xfmul(const XF &T, V2 v) { return xmul(T.q, T.p, v); }
DEV XF xf_identity() { XF T; T.p = mk(0.0f, 0.0f); T.q.s = 0.0f; T.q.c = 1.0f; return T; }

struct Proxy { V2 v[4]; int count; float radius; };
DEV V2 pget(const Proxy &p, int i) { return sel4(p.v, i); }
DEV int psupport(const Proxy &p, V2 d) {
    int best = 0;
    float bv = vdot(p.v[0], d);
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        if (i < p.count) {
            float val = vdot(p.v[i], d);
            if (val > bv) { best = i; bv = val; }
        }
    }
    return best;
}
DEV Proxy proxy_edge(V2 a, V2 b) {
    Proxy p; p.v[0] = a; p.v[1] = b; p.v[2] = mk(0.0f, 0.0f); p.v[3] = mk(0.0f, 0.0f); p.count = 2; p.radius = B2_POLYGON_RADIUS; return p;
}
DEV Proxy proxy_body(int shape, float hx, float hy) {
    Proxy p;
    if (shape == SHAPE_BOX) {
        p.v[0] = mk(-hx, -hy); p.v[1] = mk(hx, -hy); p.v[2] = mk(hx, hy); p.v[3] = mk(-hx, hy); p.count = 4; p.radius = B2_POLYGON_RADIUS;
    } else {
        p.v[0] = mk(0.0f, 0.0f); p.v[1] = p.v[0]; p.v[2] = p.v[0]; p.v[3] = p.v[0]; p.count = 1; p.radius = hx;
    }
    return p;
}
struct SVtx { V2 wA, wB, w; float a; int iA, iB; };
struct Simplex { SVtx v0, v1, v2; int count; };
struct SCache { float metric; int count; int iA0, iA1, iA2, iB0, iB1, iB2; };

DEV float simplex_metric(const Simplex &s) {
    if (s.count == 2) return sqrtf(vdist2(s.v0.w, s.v1.w));
    if (s.count == 3) return vcross(vsub(s.v1.w, s.v0.w), vsub(s.v2.w, s.v0.w));
    return 0.0f;
}
DEV SVtx svtx_make(const Proxy &pA, const XF &xfA, const Proxy &pB, const XF &xfB, int iA, int iB, float a) {
    SVtx v;
    v.iA = iA; v.iB = iB;
    v.wA = xfmul(xfA, pget(pA, iA));
    v.wB = xfmul(xfB, pget(pB, iB));
    v.w = vsub(v.wB, v.wA);
    v.a = a;
    return v;
}
DEV void simplex_read_cache(Simplex &s, const SCache &c, const Proxy &pA, const XF &xfA, const Proxy &pB, const XF &xfB) {
    s.count = c.count;
    s.v0 = svtx_make(pA, xfA, pB, xfB, c.count > 0 ? c.iA0 : 0, c.count > 0 ? c.iB0 : 0, 0.0f);
    s.v1 = svtx_make(pA, xfA, pB, xfB, c.count > 1 ? c.iA1 : 0, c.count > 1 ? c.iB1 : 0, 0.0f);
    s.v2 = svtx_make(pA, xfA, pB, xfB, c.count > 2 ? c.iA2 : 0, c.count > 2 ? c.iB2 : 0, 0.0f);
    if (s.count > 1) {
        float metric1 = c.metric;
        float metric2 = simplex_metric(s);
        if (metric2 < 0.5f * metric1 || 2.0f * metric1 < metric2 || metric2 < B2_EPSILON) s.count = 0;
    }
    if (s.count == 0) {
        s.v0 = svtx_make(pA, xfA, pB, xfB, 0, 0, 1.0f);
        s.count = 1;
    }
}
DEV void simplex_write_cache(const Simplex &s, SCache &c) {
    c.metric = simplex_metric(s);
    c.count = s.count;
    c.iA0 = s.v0.iA; c.iB0 = s.v0.iB;
    c.iA1 = s.v1.iA; c.iB1 = s.v1.iB;
    c.iA2 = s.v2.iA; c.iB2 = s.v2.iB;
}
DEV void simplex_solve2(Simplex &s) {
    V2 w1 = s.v0.w, w2 = s.v1.w;
    V2 e12 = vsub(w2, w1);
    float d12_2 = -vdot(w1, e12);
    if (d12_2 <= 0.0f) { s.v0.a = 1.0f; s.count = 1; return; }
    float d12_1 = vdot(w2, e12);
    if (d12_1 <= 0.0f) { s.v1.a = 1.0f; s.count = 1; s.v0 = s.v1; return; }
    float inv_d12 = 1.0f / (d12_1 + d12_2);
    s.v0.a = d12_1 * inv_d12;
    s.v1.a = d12_2 * inv_d12;
    s.count = 2;
}
DEV void simplex_solve3(Simplex &s) {
    V2 w1 = s.v0.w, w2 = s.v1.w, w3 = s.v2.w;
    V2 e12 = vsub(w2, w1);
    float w1e12 = vdot(w1, e12), w2e12 = vdot(w2, e12);
    float d12_1 = w2e12, d12_2 = -w1e12;
    V2 e13 = vsub(w3, w1);
    float w1e13 = vdot(w1, e13), w3e13 = vdot(w3, e13);
    float d13_1 = w3e13, d13_2 = -w1e13;
    V2 e23 = vsub(w3, w2);
    float w2e23 = vdot(w2, e23), w3e23 = vdot(w3, e23);
    float d23_1 = w3e23, d23_2 = -w2e23;
    float n123 = vcross(e12, e13);
    float d123_1 = n123 * vcross(w2, w3);
    float d123_2 = n123 * vcross(w3, w1);
    float d123_3 = n123 * vcross(w1, w2);
    if (d12_2 <= 0.0f && d13_2 <= 0.0f) { s.v0.a = 1.0f; s.count = 1; return; }
    if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f) {
        float inv_d12 = 1.0f / (d12_1 + d12_2);
        s.v0.a = d12_1 * inv_d12; s.v1.a = d12_2 * inv_d12; s.count = 2; return;
    }
    if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f) {
        float inv_d13 = 1.0f / (d13_1 + d13_2);
        s.v0.a = d13_1 * inv_d13; s.v2.a = d13_2 * inv_d13; s.count = 2; s.v1 = s.v2; return;
    }
    if (d12_1 <= 0.0f && d23_2 <= 0.0f) { s.v1.a = 1.0f; s.count = 1; s.v0 = s.v1; return; }
    if (d13_1 <= 0.0f && d23_1 <= 0.0f) { s.v2.a = 1.0f; s.count = 1; s.v0 = s.v2; return; }
    if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f) {
        float inv_d23 = 1.0f / (d23_1 + d23_2);
        s.v1.a = d23_1 * inv_d23; s.v2.a = d23_2 * inv_d23; s.count = 2; s.v0 = s.v2; return;
    }
    float inv_d123 = 1.0f / (d123_1 + d123_2 + d123_3);
    s.v0.a = d123_1 * inv_d123; s.v1.a = d123_2 * inv_d123; s.v2.a = d123_3 * inv_d123; s.count = 3;
}
// b2Distance with useRadii = false; returns the distance between the core shapes
DEV float gjk_distance(SCache &cache, const Proxy &pA, const XF &xfA, const Proxy &pB, const XF &xfB) {
    Simplex s;
    simplex_read_cache(s, cache, pA, xfA, pB, xfB);
    int iter = 0;
    while (iter < 20) {
        int saveCount = s.count;
        int sA0 = s.v0.iA, sB0 = s.v0.iB, sA1 = s.v1.iA, sB1 = s.v1.iB, sA2 = s.v2.iA, sB2 = s.v2.iB;
        if (s.count == 2) simplex_solve2(s);
        else if (s.count == 3) simplex_solve3(s);
        if (s.count == 3) break;
        // search direction
        V2 d;
        if (s.count == 1) d = vneg(s.v0.w);
        else {
            V2 e12 = vsub(s.v1.w, s.v0.w);
            float sgn = vcross(e12, vneg(s.v0.w));
            d = sgn > 0.0f ? vcross_sv(1.0f, e12) : vcross_vs(e12, 1.0f);
        }
        if (vdot(d, d) < B2_EPSILON * B2_EPSILON) break;
        int iA = psupport(pA, rmulT(xfA.q, vneg(d)));
        int iB = psupport(pB, rmulT(xfB.q, d));
        SVtx nv = svtx_make(pA, xfA, pB, xfB, iA, iB, 0.0f);
        // vertices[count] = nv (a is left as it was in Box2D: stale; it is overwritten by the next Solve)
        if (s.count == 1) { nv.a = s.v1.a; s.v1 = nv; } else { nv.a = s.v2.a; s.v2 = nv; }
        ++iter;
        bool duplicate = (saveCount > 0 && iA == sA0 && iB == sB0) || (saveCount > 1 && iA == sA1 && iB == sB1) ||
                         (saveCount > 2 && iA == sA2 && iB == sB2);
        if (duplicate) break;
        ++s.count;
    }
    V2 pointA, pointB;
    if (s.count == 1) { pointA = s.v0.wA; pointB = s.v0.wB; }
    else if (s.count == 2) {
        pointA = vadd(vscale(s.v0.a, s.v0.wA), vscale(s.v1.a, s.v1.wA));
        pointB = vadd(vscale(s.v0.a, s.v0.wB), vscale(s.v1.a, s.v1.wB));
    } else {
        pointA = vadd(vadd(vscale(s.v0.a, s.v0.wA), vscale(s.v1.a, s.v1.wA)), vscale(s.v2.a, s.v2.wA));
        pointB = pointA;
    }
    simplex_write_cache(s, cache);
    return sqrtf(vdist2(pointA, pointB));
}

struct Sweep { V2 c0, c; float a0, a; };
DEV XF sweep_xf(const Sweep &s, float beta) {
    XF T;
    T.p = vadd(vscale(1.0f - beta, s.c0), vscale(beta, s.c));
    float angle = (1.0f - beta) * s.a0 + beta * s.a;
    T.q = rot_set(angle);
    T.p = vsub(T.p, rmul(T.q, mk(0.0f, 0.0f)));
    return T;
}
enum { SEP_POINTS = 0, SEP_FACE_A = 1, SEP_FACE_B = 2 };
struct SepFn { int type; V2 localPoint, axis; };
DEV void sepfn_init(SepFn &f, const SCache &cache, const Proxy &pA, const Proxy &pB, const Sweep &sB, float t1) {
    XF xfA = xf_identity(), xfB = sweep_xf(sB, t1);
    f.localPoint = mk(0.0f, 0.0f);
    if (cache.count == 1) {
        f.type = SEP_POINTS;
        V2 pointA = xfmul(xfA, pget(pA, cache.iA0)), pointB = xfmul(xfB, pget(pB, cache.iB0));
        f.axis = vsub(pointB, pointA);
        vnormalize(f.axis);
    } else if (cache.iA0 == cache.iA1) {
        f.type = SEP_FACE_B;
        V2 lB1 = pget(pB, cache.iB0), lB2 = pget(pB, cache.iB1);
        f.axis = vcross_vs(vsub(lB2, lB1), 1.0f);
        vnormalize(f.axis);
        V2 normal = rmul(xfB.q, f.axis);
        f.localPoint = vscale(0.5f, vadd(lB1, lB2));
        V2 pointB = xfmul(xfB, f.localPoint);
        V2 pointA = xfmul(xfA, pget(pA, cache.iA0));
        float s = vdot(vsub(pointA, pointB), normal);
        if (s < 0.0f) f.axis = vneg(f.axis);
    } else {
        f.type = SEP_FACE_A;
        V2 lA1 = pget(pA, cache.iA0), lA2 = pget(pA, cache.iA1);
        f.axis = vcross_vs(vsub(lA2, lA1), 1.0f);
        vnormalize(f.axis);
        V2 normal = rmul(xfA.q, f.axis);
        f.localPoint = vscale(0.5f, vadd(lA1, lA2));
        V2 pointA = xfmul(xfA, f.localPoint);
        V2 pointB = xfmul(xfB, pget(pB, cache.iB0));
        float s = vdot(vsub(pointB, pointA), normal);
        if (s < 0.0f) f.axis = vneg(f.axis);
    }
}
DEV float sepfn_find_min(const SepFn &f, const Proxy &pA, const Proxy &pB, const Sweep &sB, int &indexA, int &indexB, float t) {
    XF xfA = xf_identity(), xfB = sweep_xf(sB, t);
    if (f.type == SEP_POINTS) {
        V2 axisA = rmulT(xfA.q, f.axis);
        V2 axisB = rmulT(xfB.q, vneg(f.axis));
        indexA = psupport(pA, axisA);
        indexB = psupport(pB, axisB);
        V2 pointA = xfmul(xfA, pget(pA, indexA)), pointB = xfmul(xfB, pget(pB, indexB));
        return vdot(vsub(pointB, pointA), f.axis);
    } else if (f.type == SEP_FACE_A) {
        V2 normal = rmul(xfA.q, f.axis);
        V2 pointA = xfmul(xfA, f.localPoint);
        V2 axisB = rmulT(xfB.q, vneg(normal));
        indexA = -1;
        indexB = psupport(pB, axisB);
        V2 pointB = xfmul(xfB, pget(pB, indexB));
        return vdot(vsub(pointB, pointA), normal);
    } else {
        V2 normal = rmul(xfB.q, f.axis);
        V2 pointB = xfmul(xfB, f.localPoint);
        V2 axisA = rmulT(xfA.q, vneg(normal));
        indexB = -1;
        indexA = psupport(pA, axisA);
        V2 pointA = xfmul(xfA, pget(pA, indexA));
        return vdot(vsub(pointA, pointB), normal);
    }
}
DEV float sepfn_evaluate(const SepFn &f, const Proxy &pA, const Proxy &pB, const Sweep &sB, int indexA, int indexB, float t) {
    XF xfA = xf_identity(), xfB = sweep_xf(sB, t);
    if (f.type == SEP_POINTS) {
        V2 pointA = xfmul(xfA, pget(pA, indexA)), pointB = xfmul(xfB, pget(pB, indexB));
        return vdot(vsub(pointB, pointA), f.axis);
    } else if (f.type == SEP_FACE_A) {
        V2 normal = rmul(xfA.q, f.axis);
        V2 pointA = xfmul(xfA, f.localPoint);
        V2 pointB = xfmul(xfB, pget(pB, indexB));
        return vdot(vsub(pointB, pointA), normal);
    } else {
        V2 normal = rmul(xfB.q, f.axis);
        V2 pointB = xfmul(xfB, f.localPoint);
        V2 pointA = xfmul(xfA, pget(pA, indexA));
        return vdot(vsub(pointA, pointB), normal);
    }
}
enum { TOI_UNKNOWN = 0, TOI_FAILED, TOI_OVERLAPPED, TOI_TOUCHING, TOI_SEPARATED };
// b2TimeOfImpact(static A, swept B, tMax = 1)
DEV void time_of_impact(int &state, float &tOut, const Proxy &pA, const Proxy &pB, Sweep sB) {
    state = TOI_UNKNOWN;
    const float tMax = 1.0f;
    tOut = tMax;
    { // b2Sweep::Normalize
        float twoPi = 2.0f * B2_PI;
        float d = twoPi * floorf(sB.a0 / twoPi);
        sB.a0 -= d;
        sB.a -= d;
    }
    float totalRadius = pA.radius + pB.radius;
    float target = fmax32(B2_LINEAR_SLOP, totalRadius - 3.0f * B2_LINEAR_SLOP);
    float tolerance = 0.25f * B2_LINEAR_SLOP;
    float t1 = 0.0f;
    int iter = 0;
    SCache cache;
    cache.metric = 0.0f; cache.count = 0; cache.iA0 = cache.iA1 = cache.iA2 = cache.iB0 = cache.iB1 = cache.iB2 = 0;
    for (;;) {
        XF xfA = xf_identity(), xfB = sweep_xf(sB, t1);
        float distance = gjk_distance(cache, pA, xfA, pB, xfB);
        if (distance <= 0.0f) { state = TOI_OVERLAPPED; tOut = 0.0f; break; }
        if (distance < target + tolerance) { state = TOI_TOUCHING; tOut = t1; break; }
        SepFn fcn;
        sepfn_init(fcn, cache, pA, pB, sB, t1);
        bool done = false;
        float t2 = tMax;
        int pushBackIter = 0;
        for (;;) {
            int indexA, indexB;
            float s2 = sepfn_find_min(fcn, pA, pB, sB, indexA, indexB, t2);
            if (s2 > target + tolerance) { state = TOI_SEPARATED; tOut = tMax; done = true; break; }
            if (s2 > target - tolerance) { t1 = t2; break; }
            float s1 = sepfn_evaluate(fcn, pA, pB, sB, indexA, indexB, t1);
            if (s1 < target - tolerance) { state = TOI_FAILED; tOut = t1; done = true; break; }
            if (s1 <= target + tolerance) { state = TOI_TOUCHING; tOut = t1; done = true; break; }
            int rootIterCount = 0;
            float a1 = t1, a2 = t2;
            for (;;) {
                float t;
                if (rootIterCount & 1) t = a1 + (target - s1) * (a2 - a1) / (s2 - s1);
                else t = 0.5f * (a1 + a2);
                ++rootIterCount;
                float s = sepfn_evaluate(fcn, pA, pB, sB, indexA, indexB, t);
                if (fabs32(s - target) < tolerance) { t2 = t; break; }
                if (s > target) { a1 = t; s1 = s; } else { a2 = t; s2 = s; }
                if (rootIterCount == 50) break;
            }
            ++pushBackIter;
            if (pushBackIter == 8) break;
        }
        ++iter;
        if (done) break;
        if (iter == 20) { state = TOI_FAILED; tOut = t1; break; }
    }
}

// pair-slot info word: bits 0-7 manifold point count, 8-15 manifold type, then b2Contact flags
#define CI_COUNT(i) ((i) & 0xff)
#define CI_TYPE(i) (((i) >> 8) & 0xff)
#define CI_ENABLED (1 << 16)
#define CI_TOIFLAG (1 << 17)
#define CI_ISLAND (1 << 18)
#define CI_TOICOUNT_SHIFT 20
#define CI_TOICOUNT(i) (((i) >> CI_TOICOUNT_SHIFT) & 0x1f)
#define CI_KEEP_MASK (~0xffff) // flag bits survive a manifold update

// b2Contact::Update for pair slot o32 of this lane at body transform (p, q): narrowphase, carry the
// warm-start impulses over by feature id, store.  Returns the new manifold.
DEV void contact_update_slot(const State &S, const Terrain &T, unsigned o, int shape, float hx, float hy, V2 p, Rot q,
                             Manifold &m, bool sleepResetAlways, float &sleepT) {
    int e = CI(C_EDGE, o); // static proxy index: hardcore boxes first, then edges
    if (e < T.nPoly) {
        Poly4 PA = static_poly(T, e);
        if (shape == SHAPE_BOX) collide_polygons(m, PA, box_poly(hx, hy), p, q);
        else collide_polygon_circle(m, PA, hx, p);
    } else {
        V2 e1 = static_vert(T, e, 0), e2 = static_vert(T, e, 1);
        if (shape == SHAPE_BOX) collide_edge_box(m, e1, e2, hx, hy, p, q);
        else collide_edge_circle(m, e1, e2, hx, p);
    }
    int info = CI(C_INFO, o);
    int oldCount = CI_COUNT(info);
    if (((m.count > 0) != (oldCount > 0)) && sleepResetAlways) sleepT = 0.0f; // touching changed -> SetAwake
    unsigned ok0 = CU(C_KEY0, o), ok1 = CU(C_KEY1, o);
    float on0 = CF(C_N0, o), on1 = CF(C_N1, o), ot0 = CF(C_T0, o), ot1 = CF(C_T1, o);
    float n0 = 0.0f, t0 = 0.0f, n1 = 0.0f, t1 = 0.0f;
    if (m.count > 0) {
        if (oldCount > 0 && ok0 == m.k0) { n0 = on0; t0 = ot0; }
        else if (oldCount > 1 && ok1 == m.k0) { n0 = on1; t0 = ot1; }
    }
    if (m.count > 1) {
        if (oldCount > 0 && ok0 == m.k1) { n1 = on0; t1 = ot0; }
        else if (oldCount > 1 && ok1 == m.k1) { n1 = on1; t1 = ot1; }
    }
    // Most pairs of a body are broadphase overlaps that do not touch: nothing about them changes from step to step.
    // Store only words whose BITS change (so the arena holds exactly what an unconditional store would leave): the
    // pair lists were a third of the step's HBM write traffic.
    const int newInfo = (info & CI_KEEP_MASK) | CI_ENABLED | m.count | (m.type << 8);
    if (newInfo != info) CI(C_INFO, o) = newInfo;
    if (m.k0 != ok0) CU(C_KEY0, o) = m.k0;
    if (m.k1 != ok1) CU(C_KEY1, o) = m.k1;
    if (__float_as_uint(n0) != __float_as_uint(on0)) CF(C_N0, o) = n0;
    if (__float_as_uint(n1) != __float_as_uint(on1)) CF(C_N1, o) = n1;
    if (__float_as_uint(t0) != __float_as_uint(ot0)) CF(C_T0, o) = t0;
    if (__float_as_uint(t1) != __float_as_uint(ot1)) CF(C_T1, o) = t1;
}
DEV void manifold_store(const State &S, unsigned gl, int t, const Manifold &m) {
    const unsigned sb = (unsigned)(t * SCR_WORDS) * S.Lp + gl;
    SW(sb, 0) = __int_as_float(m.type | (m.count << 8));
    SW(sb, 1) = m.ln.x; SW(sb, 2) = m.ln.y; SW(sb, 3) = m.lp.x; SW(sb, 4) = m.lp.y;
    SW(sb, 5) = m.p0.x; SW(sb, 6) = m.p0.y; SW(sb, 7) = m.p1.x; SW(sb, 8) = m.p1.y;
}
// scratch word offsets (per lane): manifolds [KT][SCR_WORDS], constraints [KT][CC_WORDS], TOI alphas [KC]
#define SCR_CC_BASE (KT * SCR_WORDS)
#define SCR_TOI_BASE (KT * SCR_WORDS + KT * CC_WORDS)
#define SCR_SWEEP_BASE (SCR_TOI_BASE + KC) // c0.x, c0.y, a0 handed from the step kernel to the TOI kernel
#define SCR_MISC_BASE (SCR_SWEEP_BASE + 3)  // split pipeline: [0] nTouch | solve << 8, [1] pair-slot map of the touching contacts
#define SCR_JREC_BASE (SCR_MISC_BASE + 1 + SP_WORDS)   // split pipeline: joint lever arms rA.x rA.y rB.x rB.y
#define SCR_TOTAL_WORDS (SCR_JREC_BASE + 4)
// the pair-slot map in the misc record (words 1 .. SP_WORDS)
DEV void sp_store(const State &S, unsigned mb, slotpack_t p) {
    SW(mb, 1) = __int_as_float((int)(unsigned)p);
#if SP_WORDS == 2
    SW(mb, 2) = __int_as_float((int)(unsigned)(p >> 32));
#endif
}
DEV slotpack_t sp_load(const State &S, unsigned mb) {
    slotpack_t p = (unsigned)__float_as_int(SW(mb, 1));
#if SP_WORDS == 2
    p |= (slotpack_t)(unsigned)__float_as_int(SW(mb, 2)) << 32;
#endif
    return p;
}

// Exact early-outs of the TOI query for (static proxy pA, this body's proxy pB swept by sw): true means
// b2TimeOfImpact would answer "separated" (alpha = 1) without it having to run.
DEV bool toi_far_apart(const Proxy &pA, const Proxy &pB, const Sweep &sw, int shape, float hx, float hy, float coreR) {
    // Conservative exact skip.  b2TimeOfImpact can only answer e_touching if some core-shape
    // distance / separation it evaluates for t in [0,1] falls below target + tolerance, and every
    // such value is >= the true distance of the core shapes at that t.  The body's core stays inside
    // the disk of radius coreR around its centre, which moves on the segment c0 -> c: if that
    // capsule's bounding box keeps more than target + tolerance (+ 5 mm for rounding) away from
    // the static shape's bounding box, the answer is alpha = 1 without running GJK.
    bool farApart;
    {
        V2 slo = pA.v[0], shi = pA.v[0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (k < pA.count) { slo = vmin2(slo, pA.v[k]); shi = vmax2(shi, pA.v[k]); }
        V2 blo = vsub(vmin2(sw.c0, sw.c), mk(coreR, coreR)), bhi = vadd(vmax2(sw.c0, sw.c), mk(coreR, coreR));
        float gap = fmax32(fmax32(blo.x - shi.x, slo.x - bhi.x), fmax32(blo.y - shi.y, slo.y - bhi.y));
        float totalRadius = pA.radius + pB.radius;
        float target = fmax32(B2_LINEAR_SLOP, totalRadius - 3.0f * B2_LINEAR_SLOP);
        const float need = target + 0.25f * B2_LINEAR_SLOP;
        farApart = gap > need + 0.005f;
        if (!farApart) {
            // Second bound (catches resting contacts): any separating axis gives a lower bound lb0 of the
            // core distance at the sweep start, and no point of the body moves further than
            // |c - c0| + coreR * |a - a0| during the sweep, so distance(t) >= lb0 - that.  Axes tried:
            // the static shape's face normals and the body's own axes.
            // Tighter for the static shape's own axes u, which do not move.  A body vertex follows
            // c(t) + R(a(t)) r with c and a interpolated linearly (b2Sweep), so u.vertex(t) is a linear function of
            // t plus a sinusoid of amplitude <= |r| sampled over an angle |a - a0|: it stays within
            // |r| (1 - cos(|a - a0| / 2)) <= coreR (a - a0)^2 / 8 of the chord between its values at t = 0 and t = 1.
            // Hence separation_u(t) >= min(separation_u(0), separation_u(1)) - coreR (a - a0)^2 / 8 for all t: a body
            // that slides or turns next to an edge without coming closer is skipped however far it moves.
            Rot q0 = rot_set(sw.a0), q1 = rot_set(sw.a);
            float lb0 = -FLT_MAX, lbTight = -FLT_MAX;
            const V2 dc = vsub(sw.c, sw.c0);
            const float da = sw.a - sw.a0;
            const float rot = coreR * fabs32(da);
            const float sag = coreR * (da * da) * 0.125f;
            V2 bv[4], bw[4];
            const int nb = pB.count;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bv[k] = xmul(q0, sw.c0, pB.v[k < nb ? k : 0]);
                bw[k] = xmul(q1, sw.c, pB.v[k < nb ? k : 0]);
            }
            if (pA.count == 2) { // edge: +-normal, and the edge direction beyond either end
                V2 e = vsub(pA.v[1], pA.v[0]);
                vnormalize(e);
                V2 n = mk(e.y, -e.x);
                float lo = FLT_MAX, hi = -FLT_MAX, tlo = FLT_MAX, thi = -FLT_MAX;
                float lo1 = FLT_MAX, hi1 = -FLT_MAX, tlo1 = FLT_MAX, thi1 = -FLT_MAX;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float d = vdot(n, vsub(bv[k], pA.v[0]));
                    lo = fmin32(lo, d); hi = fmax32(hi, d);
                    float td = vdot(e, vsub(bv[k], pA.v[0]));
                    tlo = fmin32(tlo, td); thi = fmax32(thi, td);
                    float d1 = vdot(n, vsub(bw[k], pA.v[0]));
                    lo1 = fmin32(lo1, d1); hi1 = fmax32(hi1, d1);
                    float td1 = vdot(e, vsub(bw[k], pA.v[0]));
                    tlo1 = fmin32(tlo1, td1); thi1 = fmax32(thi1, td1);
                }
                float elen = vdot(e, vsub(pA.v[1], pA.v[0]));
                lb0 = fmax32(fmax32(lo, -hi), fmax32(tlo - elen, -thi));
                lbTight = fmax32(fmax32(fmin32(lo, lo1), -fmax32(hi, hi1)),
                                 fmax32(fmin32(tlo, tlo1) - elen, -fmax32(thi, thi1))) - sag;
            } else { // static box: its four face normals
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    V2 a = pA.v[i], b2 = pA.v[(i + 1) & 3];
                    V2 ed = vsub(b2, a);
                    vnormalize(ed);
                    V2 n = mk(ed.y, -ed.x);
                    float lo = FLT_MAX, lo1 = FLT_MAX;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lo = fmin32(lo, vdot(n, vsub(bv[k], a)));
                        lo1 = fmin32(lo1, vdot(n, vsub(bw[k], a)));
                    }
                    lb0 = fmax32(lb0, lo);
                    lbTight = fmax32(lbTight, fmin32(lo, lo1) - sag);
                }
            }
            if (shape == SHAPE_BOX) { // the body's axes against the static shape's vertices
                float xlo = FLT_MAX, xhi = -FLT_MAX, ylo = FLT_MAX, yhi = -FLT_MAX;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < pA.count) {
                        V2 l = rmulT(q0, vsub(pA.v[k], sw.c0));
                        xlo = fmin32(xlo, l.x); xhi = fmax32(xhi, l.x);
                        ylo = fmin32(ylo, l.y); yhi = fmax32(yhi, l.y);
                    }
                }
                lb0 = fmax32(lb0, fmax32(fmax32(xlo - hx, -xhi - hx), fmax32(ylo - hy, -yhi - hy)));
            }
            float maxDisp = vlen(dc) + rot;
            farApart = (lb0 - maxDisp > need + 0.002f) || (lbTight > need + 0.002f);
        }
    }
    return farApart;
}

struct LaneBody { float px, py, ang, vx, vy, w, sleepT; int awake, cCount, err, events; };

// b2World::SolveTOI restricted to this lane's body (see the section comment above).
// The TOI island's manifolds (<= KT per body) live in LDS while the sub-step is solved: the 20 TOI position
// iterations and the constraint set-up read them again and again, and a round trip to the scratch arena in HBM/L2
// per word was most of this kernel's latency chain.  [word][lane]: conflict-free for a wavefront.
// COLS = WAVE: a column per lane (a wavefront may carry several bodies); COLS = 1: the whole wavefront carries ONE body, its lanes
// hold copies of each other and share one column (the in-wavefront solve of rem2d_step_queue_kernel: 216 B instead of 13.8 KB).
template <int COLS> struct ToiSharedT { float m[KT * SCR_WORDS][COLS]; };
typedef ToiSharedT<WAVE> ToiShared;
template <int COLS> DEV void island_store(ToiSharedT<COLS> &ts, int lane, int t, const Manifold &m) {
    float (*w)[COLS] = ts.m + t * SCR_WORDS;
    const int c = lane & (COLS - 1);
    w[0][c] = __int_as_float(m.type | (m.count << 8));
    w[1][c] = m.ln.x; w[2][c] = m.ln.y; w[3][c] = m.lp.x; w[4][c] = m.lp.y;
    w[5][c] = m.p0.x; w[6][c] = m.p0.y; w[7][c] = m.p1.x; w[8][c] = m.p1.y;
}
#define IW(t, k) (ts.m[(t) * SCR_WORDS + (k)][lane & (COLS - 1)])
template <int COLS>
DEV LaneBody solve_toi_lane(const State &S, const Terrain &T, unsigned gl, int shape, float hx, float hy, float mB, float iB,
                                                float h, int velIters, float c0x, float c0y, float a0, LaneBody B,
                                                ToiSharedT<COLS> &ts, int lane, int sub, int G) {
    // sub / G: the G lanes (a power of two, G-aligned) that carry this body.  All of them run this function on the same
    // values -- so every store below writes the same word G times, harmless -- except in the alpha pass, where lane `sub`
    // takes the pair slots sub, sub + G, ...: the b2TimeOfImpact calls of a body run side by side instead of one after
    // the other, and a reduction hands every lane the same (smallest alpha, first slot) the sequential scan finds.
    const bool sleepResetAlways = (S.flags & REM2D_FLAG_SLEEP_RESET_ALWAYS) != 0;
    const float radiusB = shape == SHAPE_CIRCLE ? hx : B2_POLYGON_RADIUS;
    const float friction = T.friction;
    const unsigned Lp = S.Lp;
    // m_stepComplete is always true at this point: invalidate TOIs
    float alpha0 = 0.0f;
    Sweep sw;
    sw.c0 = mk(c0x, c0y); sw.c = mk(B.px, B.py); sw.a0 = a0; sw.a = B.ang;
    for (int s = 0; s < B.cCount; ++s) {
        unsigned o = (unsigned)s * Lp + gl;
        CI(C_INFO, o) = CI(C_INFO, o) & (0xffff | CI_ENABLED);
    }
    const Proxy pB = proxy_body(shape, hx, hy);
    const float coreR = shape == SHAPE_BOX ? sqrtf(hx * hx + hy * hy) : 0.0f; // circumradius of the core shape
#ifdef REM2D_TOI_STAMPS
    unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tk, cyc[5] = {0, 0, 0, 0, 0};
    int nSweeps = 0, nAlpha = 0, islMax = 0;
#define TOI_STAMP(i) do { tk = __builtin_amdgcn_s_memtime(); cyc[i] += tk - tk0; tk0 = tk; } while (0)
#define TOI_COUNT(x) do { x; } while (0)
#else
#define TOI_STAMP(i) do {} while (0)
#define TOI_COUNT(x) do {} while (0)
#endif
    for (;;) {
        int minSlot = -1;
        float minAlpha = 1.0f;
        for (int s = sub; s < B.cCount; s += G) {
            unsigned o = (unsigned)s * Lp + gl;
            int info = CI(C_INFO, o);
            if (!(info & CI_ENABLED)) continue;
            if (CI_TOICOUNT(info) > 8) continue;
            float alpha = 1.0f;
            if (info & CI_TOIFLAG) {
                alpha = SW((unsigned)(SCR_TOI_BASE + s) * Lp + gl, 0);
            } else {
                if (!B.awake) continue;
                int e = CI(C_EDGE, o);
                Proxy pA = proxy_edge(static_vert(T, e, 0), static_vert(T, e, 1));
                if (e < T.nPoly) { pA.v[2] = static_vert(T, e, 2); pA.v[3] = static_vert(T, e, 3); pA.count = 4; }
                const bool farApart = toi_far_apart(pA, pB, sw, shape, hx, hy, coreR);
                if (farApart) {
                    alpha = 1.0f;
                } else {
                    int state;
                    float t;
                    time_of_impact(state, t, pA, pB, sw);
                    TOI_COUNT(nAlpha += 1);
                    float beta = t;
                    if (state == TOI_TOUCHING) alpha = fmin32(alpha0 + (1.0f - alpha0) * beta, 1.0f);
                    else alpha = 1.0f;
                }
                SW((unsigned)(SCR_TOI_BASE + s) * Lp + gl, 0) = alpha;
                CI(C_INFO, o) = info | CI_TOIFLAG;
            }
            if (alpha < minAlpha) { minSlot = s; minAlpha = alpha; }
        }
        if (G > 1) __threadfence_block(); // the flags and alphas a lane stored for its slots are read by the body's other lanes below
        for (int o = 1; o < G; o <<= 1) { // smallest alpha of the body's lanes; among equal ones the first slot (the scan's choice)
            const float a2 = __shfl_xor(minAlpha, o);
            const int s2 = __shfl_xor(minSlot, o);
            if (s2 >= 0 && (a2 < minAlpha || (a2 == minAlpha && (minSlot < 0 || s2 < minSlot)))) { minAlpha = a2; minSlot = s2; }
        }
        TOI_STAMP(0); // alpha pass: b2TimeOfImpact of every pair whose TOI is not cached
        if (minSlot < 0 || 1.0f - 10.0f * B2_EPSILON < minAlpha) break;
        // ---- advance the body to the TOI (b2Body::Advance) ----
        const Sweep backup = sw;
        const float backupAlpha0 = alpha0;
        {
            float beta = (minAlpha - alpha0) / (1.0f - alpha0);
            sw.c0 = vadd(sw.c0, vscale(beta, vsub(sw.c, sw.c0)));
            sw.a0 += beta * (sw.a - sw.a0);
            alpha0 = minAlpha;
            sw.c = sw.c0;
            sw.a = sw.a0;
        }
        Rot q = rot_set(sw.a);
        V2 p = vsub(sw.c, rmul(q, mk(0.0f, 0.0f)));
        const unsigned om = (unsigned)minSlot * Lp + gl;
        Manifold m;
        contact_update_slot(S, T, om, shape, hx, hy, p, q, m, sleepResetAlways, B.sleepT);
        {
            int info = CI(C_INFO, om);
            int cnt = CI_TOICOUNT(info) + 1;
            info = (info & ~(CI_TOIFLAG | (0x1f << CI_TOICOUNT_SHIFT))) | (cnt << CI_TOICOUNT_SHIFT);
            if (m.count == 0) info &= ~CI_ENABLED; // not solid: SetEnabled(false), restore the sweep
            CI(C_INFO, om) = info;
        }
        if (m.count == 0) {
            sw = backup;
            alpha0 = backupAlpha0;
            continue;
        }
        if (sleepResetAlways || !B.awake) B.sleepT = 0.0f;
        B.awake = 1;
        B.events += 1;
        // ---- TOI island: this body, the TOI contact, then its other touching contacts (list order) ----
        int nIsl = 0;
        island_store(ts, lane, 0, m);
        nIsl = 1;
        CI(C_INFO, om) = CI(C_INFO, om) | CI_ISLAND;
        for (int s = 0; s < B.cCount; ++s) {
            if (s == minSlot) continue;
            unsigned o = (unsigned)s * Lp + gl;
            Manifold mo;
            contact_update_slot(S, T, o, shape, hx, hy, p, q, mo, sleepResetAlways, B.sleepT);
            if (mo.count == 0) continue;
            if (nIsl >= KT) { B.err |= REM2D_ERR_SOLVER_OVERFLOW; continue; }
            island_store(ts, lane, nIsl, mo);
            ++nIsl;
        }
        TOI_COUNT(islMax = max(islMax, nIsl));
        TOI_STAMP(1); // contact updates (TOI contact + the body's other pairs)
        // ---- b2Island::SolveTOI ----
        float cx = sw.c.x, cy = sw.c.y, ca = sw.a;
        for (int it = 0; it < 20; ++it) { // SolveTOIPositionConstraints: only this body moves
            float minSeparation = 0.0f;
            for (int t = 0; t < nIsl; ++t) {
                int tc = __float_as_int(IW(t, 0));
                int mtype = tc & 0xff, mcount = tc >> 8;
                V2 ln = mk(IW(t, 1), IW(t, 2)), lp = mk(IW(t, 3), IW(t, 4));
                const float radiusA = B2_POLYGON_RADIUS;
                for (int j = 0; j < mcount; ++j) {
                    V2 pj = mk(IW(t, 5 + 2 * j), IW(t, 6 + 2 * j));
                    V2 cB = mk(cx, cy);
                    V2 normal, point;
                    float separation;
                    Rot qB = rot_set(ca);
                    if (mtype == MF_CIRCLES) {
                        V2 pointA = lp;
                        V2 pointB = xmul(qB, cB, mk(IW(t, 5), IW(t, 6)));
                        normal = vsub(pointB, pointA);
                        vnormalize(normal);
                        point = vscale(0.5f, vadd(pointA, pointB));
                        separation = vdot(vsub(pointB, pointA), normal) - radiusA - radiusB;
                    } else if (mtype == MF_FACE_A) {
                        normal = ln;
                        V2 clipPoint = xmul(qB, cB, pj);
                        separation = vdot(vsub(clipPoint, lp), normal) - radiusA - radiusB;
                        point = clipPoint;
                    } else {
                        normal = rmul(qB, ln);
                        V2 planePoint = xmul(qB, cB, lp);
                        V2 clipPoint = pj;
                        separation = vdot(vsub(clipPoint, planePoint), normal) - radiusA - radiusB;
                        point = clipPoint;
                        normal = vneg(normal);
                    }
                    V2 rBp = vsub(point, cB);
                    minSeparation = fmin32(minSeparation, separation);
                    float C = fclamp(0.75f * (separation + B2_LINEAR_SLOP), -B2_MAX_LINEAR_CORRECTION, 0.0f);
                    float rnB = vcross(rBp, normal);
                    float Kn = mB + iB * rnB * rnB;
                    float impulse = Kn > 0.0f ? -C / Kn : 0.0f;
                    V2 P = vscale(impulse, normal);
                    cx = cx + mB * P.x;
                    cy = cy + mB * P.y;
                    ca += iB * vcross(rBp, P);
                }
            }
            if (minSeparation >= -1.5f * B2_LINEAR_SLOP) break;
        }
        TOI_STAMP(2); // TOI position iterations
        // leap of faith to the new safe state
        sw.c0 = mk(cx, cy);
        sw.a0 = ca;
        // InitializeVelocityConstraints without warm starting, then velIters sweeps over the island contacts
        // (the first KR constraints stay in registers, further ones -- rare -- go through scratch)
        {
            Rot qn = rot_set(ca);
            ContactC tcc[KR];
#pragma unroll
            for (int t = 0; t < KR; ++t) {
                tcc[t].count = 0;
                if (t < nIsl) {
                    int tc = __float_as_int(IW(t, 0));
                    contact_setup(tcc[t], tc & 0xff, tc >> 8, mk(IW(t, 1), IW(t, 2)), mk(IW(t, 3), IW(t, 4)),
                                  mk(IW(t, 5), IW(t, 6)), mk(IW(t, 7), IW(t, 8)), mk(cx, cy), qn, mB, iB, radiusB, 0.0f, 0.0f,
                                  0.0f, 0.0f);
                }
            }
            for (int t = KR; t < nIsl; ++t) {
                int tc = __float_as_int(IW(t, 0));
                ContactC c;
                contact_setup(c, tc & 0xff, tc >> 8, mk(IW(t, 1), IW(t, 2)), mk(IW(t, 3), IW(t, 4)), mk(IW(t, 5), IW(t, 6)),
                              mk(IW(t, 7), IW(t, 8)), mk(cx, cy), qn, mB, iB, radiusB, 0.0f, 0.0f, 0.0f, 0.0f);
                cc_store(S, (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl, c);
            }
            if (G >= 4) {
                // the lanes of this body's group are copies of each other: every quad of them splits the 2-vector arithmetic
                // of a contact solve four ways (contact_solve_quad; same bits), this lane keeping component `lane & 1`
                QuadRole role[KR];
#pragma unroll
                for (int t = 0; t < KR; ++t) role[t] = quad_role(tcc[t], lane);
                const bool isY = (lane & 1) != 0;
                float vq = isY ? B.vy : B.vx;
                for (int it = 0; it < velIters; ++it) {
                    TOI_COUNT(nSweeps += 1);
                    // Exact early exit: one sweep is a deterministic function of (velocity, impulses); a sweep
                    // that changes no bit is a fixed point, so every later sweep is the identity.  (Single-body
                    // contact-only systems reach it after ~10 sweeps; the full 180 are never needed.)
                    const unsigned hq = __float_as_uint(vq), h2 = __float_as_uint(B.w);
                    bool changed = false;
#pragma unroll
                    for (int t = 0; t < KR; ++t) {
                        if (t < nIsl) {
                            const unsigned a0 = __float_as_uint(tcc[t].n0), a1 = __float_as_uint(tcc[t].n1);
                            const unsigned a2 = __float_as_uint(tcc[t].t0), a3 = __float_as_uint(tcc[t].t1);
                            contact_solve_quad(tcc[t], role[t], mB, iB, friction, vq, B.w);
                            changed |= a0 != __float_as_uint(tcc[t].n0) || a1 != __float_as_uint(tcc[t].n1) ||
                                       a2 != __float_as_uint(tcc[t].t0) || a3 != __float_as_uint(tcc[t].t1);
                        }
                    }
                    if (nIsl > KR) { // (rare) further constraints go through scratch, solved by one lane's serial form
                        float vo = quad_swap1(vq);
                        float vx = isY ? vo : vq, vy = isY ? vq : vo;
                        for (int t = KR; t < nIsl; ++t) {
                            const unsigned cb = (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl;
                            ContactC c;
                            cc_load(S, cb, c);
                            const unsigned a0 = __float_as_uint(c.n0), a1 = __float_as_uint(c.n1);
                            const unsigned a2 = __float_as_uint(c.t0), a3 = __float_as_uint(c.t1);
                            contact_solve(c, mB, iB, friction, vx, vy, B.w);
                            changed |= a0 != __float_as_uint(c.n0) || a1 != __float_as_uint(c.n1) || a2 != __float_as_uint(c.t0) ||
                                       a3 != __float_as_uint(c.t1);
                            SW(cb, 10) = c.n0; SW(cb, 11) = c.n1; SW(cb, 12) = c.t0; SW(cb, 13) = c.t1;
                        }
                        vq = isY ? vy : vx;
                    }
                    int ch = (changed || hq != __float_as_uint(vq) || h2 != __float_as_uint(B.w)) ? 1 : 0;
                    ch |= __builtin_amdgcn_update_dpp(0, ch, 0xB1, 0xf, 0xf, false); // the other component's lane
                    if (!ch) break;
                }
                const float vo = quad_swap1(vq);
                B.vx = isY ? vo : vq;
                B.vy = isY ? vq : vo;
            } else
            for (int it = 0; it < velIters; ++it) {
                // (a group of fewer than four lanes per body -- the long work list after a reset: the serial form)
                unsigned h0 = __float_as_uint(B.vx), h1 = __float_as_uint(B.vy), h2 = __float_as_uint(B.w);
                bool changed = false;
#pragma unroll
                for (int t = 0; t < KR; ++t) {
                    if (t < nIsl) {
                        const unsigned a0 = __float_as_uint(tcc[t].n0), a1 = __float_as_uint(tcc[t].n1);
                        const unsigned a2 = __float_as_uint(tcc[t].t0), a3 = __float_as_uint(tcc[t].t1);
                        contact_solve(tcc[t], mB, iB, friction, B.vx, B.vy, B.w);
                        changed |= a0 != __float_as_uint(tcc[t].n0) || a1 != __float_as_uint(tcc[t].n1) ||
                                   a2 != __float_as_uint(tcc[t].t0) || a3 != __float_as_uint(tcc[t].t1);
                    }
                }
                for (int t = KR; t < nIsl; ++t) {
                    const unsigned cb = (unsigned)(SCR_CC_BASE + t * CC_WORDS) * Lp + gl;
                    ContactC c;
                    cc_load(S, cb, c);
                    const unsigned a0 = __float_as_uint(c.n0), a1 = __float_as_uint(c.n1);
                    const unsigned a2 = __float_as_uint(c.t0), a3 = __float_as_uint(c.t1);
                    contact_solve(c, mB, iB, friction, B.vx, B.vy, B.w);
                    changed |= a0 != __float_as_uint(c.n0) || a1 != __float_as_uint(c.n1) || a2 != __float_as_uint(c.t0) ||
                               a3 != __float_as_uint(c.t1);
                    SW(cb, 10) = c.n0; SW(cb, 11) = c.n1; SW(cb, 12) = c.t0; SW(cb, 13) = c.t1;
                }
                changed |= h0 != __float_as_uint(B.vx) || h1 != __float_as_uint(B.vy) || h2 != __float_as_uint(B.w);
                if (!changed) break;
            }
        }
        TOI_STAMP(3); // constraint set-up + velocity iterations
        // integrate the remaining (1 - minAlpha) * dt; TOI impulses are not stored
        {
            float hs = (1.0f - minAlpha) * h;
            V2 v = mk(B.vx, B.vy);
            V2 translation = vscale(hs, v);
            if (vdot(translation, translation) > B2_MAX_TRANSLATION_SQ) {
                float ratio = B2_MAX_TRANSLATION / vlen(translation);
                v = vscale(ratio, v);
            }
            float rotation = hs * B.w;
            if (rotation * rotation > B2_MAX_ROTATION_SQ) {
                float ratio = B2_MAX_ROTATION / fabs32(rotation);
                B.w *= ratio;
            }
            cx = cx + hs * v.x;
            cy = cy + hs * v.y;
            ca += hs * B.w;
            B.vx = v.x; B.vy = v.y;
        }
        sw.c = mk(cx, cy);
        sw.a = ca;
        // ---- reset flags, SynchronizeFixtures, FindNewContacts ----
        for (int s = 0; s < B.cCount; ++s) {
            unsigned o = (unsigned)s * Lp + gl;
            CI(C_INFO, o) = CI(C_INFO, o) & ~(CI_TOIFLAG | CI_ISLAND);
        }
        {
            Rot q0 = rot_set(sw.a0), q1 = rot_set(sw.a);
            V2 p0 = vsub(sw.c0, rmul(q0, mk(0.0f, 0.0f)));
            V2 p1 = vsub(sw.c, rmul(q1, mk(0.0f, 0.0f)));
            AABB b1 = body_aabb(shape, hx, hy, p0, q0), b2 = body_aabb(shape, hx, hy, p1, q1);
            V2 lo = vmin2(b1.lo, b2.lo), hi = vmax2(b1.hi, b2.hi);
            V2 displacement = vsub(p1, p0);
            V2 fatLo = mk(LF(L_FATLX), LF(L_FATLY)), fatHi = mk(LF(L_FATUX), LF(L_FATUY));
            bool contains = fatLo.x <= lo.x && fatLo.y <= lo.y && hi.x <= fatHi.x && hi.y <= fatHi.y;
            if (!contains) {
                V2 r = mk(B2_AABB_EXTENSION, B2_AABB_EXTENSION);
                V2 flo = vsub(lo, r), fhi = vadd(hi, r);
                V2 d = vscale(B2_AABB_MULTIPLIER, displacement);
                if (d.x < 0.0f) flo.x += d.x; else fhi.x += d.x;
                if (d.y < 0.0f) flo.y += d.y; else fhi.y += d.y;
                LF(L_FATLX) = flo.x; LF(L_FATLY) = flo.y; LF(L_FATUX) = fhi.x; LF(L_FATUY) = fhi.y;
                if (find_new_pairs(S, T, gl, B.cCount, flo, fhi, B.err)) {
                    if (sleepResetAlways || !B.awake) B.sleepT = 0.0f;
                    B.awake = 1;
                }
            }
        }
#ifdef REM2D_TOI_STAMPS
        TOI_STAMP(4); // integrate, flags, SynchronizeFixtures, FindNewContacts
    }
    {
        unsigned long long tot = 0;
        for (int i = 0; i < 5; ++i) { atomicAdd(&S.toiWork[2 + i], (int)(cyc[i] >> 6)); tot += cyc[i]; }
        if (atomicMax(&S.toiWork[8], (int)(tot >> 6)) < (int)(tot >> 6)) { // (diagnostic, racy: the split of the longest lane)
            S.toiWork[12] = (int)(cyc[0] >> 6); S.toiWork[13] = (int)(cyc[3] >> 6);
            S.toiWork[14] = (int)((cyc[1] + cyc[2] + cyc[4]) >> 6);
            S.toiWork[15] = nSweeps | (nAlpha << 12) | (B.events << 20) | (islMax << 24) | (G << 27);
        }
        atomicAdd(&S.toiWork[9], B.events);
        atomicMax(&S.toiWork[10], B.events);
        atomicAdd(&S.toiWork[11], 1);
#else
    }
    {
#endif
    }
    B.px = sw.c.x; B.py = sw.c.y; B.ang = sw.a;
    return B;
}

#endif
