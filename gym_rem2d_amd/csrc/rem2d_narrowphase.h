// rem2d_narrowphase.h -- b2CollideEdgeAndCircle, b2EPCollider (edge-box), b2CollidePolygons, b2CollidePolygonAndCircle, shape AABBs.
// Part of the single translation unit rem2d.hip (see its header comment); not a stand-alone header.
#ifndef REM2D_NARROWPHASE_H
#define REM2D_NARROWPHASE_H

// =====================================================================================
// narrowphase (terrain body A is static at the origin: xfA = identity)
// =====================================================================================
struct Manifold {
    int type, count;
    V2 ln, lp, p0, p1;
    unsigned k0, k1;
};
DEV unsigned mkkey(int indexA, int indexB, int typeA, int typeB) {
    return (unsigned)indexA | ((unsigned)indexB << 8) | ((unsigned)typeA << 16) | ((unsigned)typeB << 24);
}

// b2CollideEdgeAndCircle (circle m_p = 0)
DEV void collide_edge_circle(Manifold &m, V2 A, V2 B, float rB, V2 center) {
    m.count = 0;
    m.type = MF_CIRCLES;
    m.ln = mk(0.0f, 0.0f);
    m.lp = mk(0.0f, 0.0f);
    m.p0 = mk(0.0f, 0.0f);
    m.p1 = mk(0.0f, 0.0f);
    m.k0 = m.k1 = 0u;
    V2 Q = center;
    V2 e = vsub(B, A);
    float u = vdot(e, vsub(B, Q));
    float v = vdot(e, vsub(Q, A));
    float radius = B2_POLYGON_RADIUS + rB;
    if (v <= 0.0f) {
        V2 d = vsub(Q, A);
        float dd = vdot(d, d);
        if (dd > radius * radius) return;
        m.count = 1;
        m.type = MF_CIRCLES;
        m.lp = A;
        m.k0 = mkkey(0, 0, CF_VERTEX, CF_VERTEX);
        return;
    }
    if (u <= 0.0f) {
        V2 d = vsub(Q, B);
        float dd = vdot(d, d);
        if (dd > radius * radius) return;
        m.count = 1;
        m.type = MF_CIRCLES;
        m.lp = B;
        m.k0 = mkkey(1, 0, CF_VERTEX, CF_VERTEX);
        return;
    }
    float den = vdot(e, e);
    V2 P = vscale(1.0f / den, vadd(vscale(u, A), vscale(v, B)));
    V2 d = vsub(Q, P);
    float dd = vdot(d, d);
    if (dd > radius * radius) return;
    V2 n = mk(-e.y, e.x);
    if (vdot(n, vsub(Q, A)) < 0.0f) n = mk(-n.x, -n.y);
    vnormalize(n);
    m.count = 1;
    m.type = MF_FACE_A;
    m.ln = n;
    m.lp = A;
    m.k0 = mkkey(0, 0, CF_FACE, CF_VERTEX);
}

struct ClipV { V2 v; int iA, iB, tA, tB; };
DEV int clip_segment(ClipV (&out)[2], const ClipV (&in)[2], V2 normal, float offset, int vertexIndexA) {
    int numOut = 0;
    float distance0 = vdot(normal, in[0].v) - offset;
    float distance1 = vdot(normal, in[1].v) - offset;
    ClipV o0 = in[0], o1 = in[1];
    bool k0 = distance0 <= 0.0f, k1 = distance1 <= 0.0f;
    // compact without dynamic indexing
    if (k0 && k1) { out[0] = o0; out[1] = o1; numOut = 2; }
    else if (k0) { out[0] = o0; out[1] = o0; numOut = 1; }
    else if (k1) { out[0] = o1; out[1] = o1; numOut = 1; }
    else { out[0] = o0; out[1] = o0; numOut = 0; }
    if (distance0 * distance1 < 0.0f) {
        float interp = distance0 / (distance0 - distance1);
        ClipV x;
        x.v = vadd(in[0].v, vscale(interp, vsub(in[1].v, in[0].v)));
        x.iA = vertexIndexA & 0xff;
        x.iB = in[0].iB;
        x.tA = CF_VERTEX;
        x.tB = CF_FACE;
        if (numOut == 0) out[0] = x; else out[1] = x; // numOut is 0 or 1 here
        ++numOut;
    }
    return numOut;
}
DEV V2 sel4(const V2 (&a)[4], int i) {
    V2 r = a[0];
    r = i == 1 ? a[1] : r;
    r = i == 2 ? a[2] : r;
    r = i == 3 ? a[3] : r;
    return r;
}
// b2EPCollider::Collide for an isolated edge and a SetAsBox polygon (centroid 0, radius 0.01)
DEV void collide_edge_box(Manifold &m, V2 v1, V2 v2, float hx, float hy, V2 p, Rot q) {
    m.count = 0;
    m.type = MF_FACE_A;
    m.ln = mk(0.0f, 0.0f);
    m.lp = mk(0.0f, 0.0f);
    m.p0 = mk(0.0f, 0.0f);
    m.p1 = mk(0.0f, 0.0f);
    m.k0 = m.k1 = 0u;
    const V2 vloc[4] = {mk(-hx, -hy), mk(hx, -hy), mk(hx, hy), mk(-hx, hy)};
    const V2 nloc[4] = {mk(0.0f, -1.0f), mk(1.0f, 0.0f), mk(0.0f, 1.0f), mk(-1.0f, 0.0f)};
    V2 centroidB = xmul(q, p, mk(0.0f, 0.0f));
    V2 edge1 = vsub(v2, v1);
    vnormalize(edge1);
    V2 normal1 = mk(edge1.y, -edge1.x);
    float offset1 = vdot(normal1, vsub(centroidB, v1));
    bool front = offset1 >= 0.0f;
    V2 normal = front ? normal1 : vneg(normal1);
    V2 limit = front ? vneg(normal1) : normal1; // lowerLimit == upperLimit for an isolated edge
    V2 pv[4], pn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pv[i] = xmul(q, p, vloc[i]);
        pn[i] = rmul(q, nloc[i]);
    }
    const float radius = 2.0f * B2_POLYGON_RADIUS;
    float edgeSep = FLT_MAX;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s = vdot(normal, vsub(pv[i], v1));
        if (s < edgeSep) edgeSep = s;
    }
    if (edgeSep > radius) return;
    // ComputePolygonSeparation
    int polyIndex = -1;
    float polySep = -FLT_MAX;
    bool separated = false;
    V2 perp = mk(-normal.y, normal.x);
    (void)perp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        V2 n = vneg(pn[i]);
        float s1 = vdot(n, vsub(pv[i], v1));
        float s2 = vdot(n, vsub(pv[i], v2));
        float s = fmin32(s1, s2);
        if (s > radius) separated = true; // first such axis ends the search with "no collision"
        // adjacency filter: lower == upper limit, so both branches test the same expression
        bool skip = vdot(vsub(n, limit), normal) < -B2_ANGULAR_SLOP;
        if (!separated && !skip && s > polySep) {
            polyIndex = i;
            polySep = s;
        }
    }
    if (separated) return;
    bool polyValid = polyIndex >= 0;
    const float k_relativeTol = 0.98f, k_absoluteTol = 0.001f;
    bool primaryIsPoly = polyValid && (polySep > k_relativeTol * edgeSep + k_absoluteTol);
    ClipV ie[2];
    int rf_i1, rf_i2;
    V2 rf_v1, rf_v2, rf_normal;
    if (!primaryIsPoly) {
        m.type = MF_FACE_A;
        int bestIndex = 0;
        float bestValue = vdot(normal, pn[0]);
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            float value = vdot(normal, pn[i]);
            if (value < bestValue) { bestValue = value; bestIndex = i; }
        }
        int i1 = bestIndex, i2 = i1 + 1 < 4 ? i1 + 1 : 0;
        ie[0].v = sel4(pv, i1); ie[0].iA = 0; ie[0].iB = i1; ie[0].tA = CF_FACE; ie[0].tB = CF_VERTEX;
        ie[1].v = sel4(pv, i2); ie[1].iA = 0; ie[1].iB = i2; ie[1].tA = CF_FACE; ie[1].tB = CF_VERTEX;
        if (front) { rf_i1 = 0; rf_i2 = 1; rf_v1 = v1; rf_v2 = v2; rf_normal = normal1; }
        else { rf_i1 = 1; rf_i2 = 0; rf_v1 = v2; rf_v2 = v1; rf_normal = vneg(normal1); }
    } else {
        m.type = MF_FACE_B;
        ie[0].v = v1; ie[0].iA = 0; ie[0].iB = polyIndex; ie[0].tA = CF_VERTEX; ie[0].tB = CF_FACE;
        ie[1].v = v2; ie[1].iA = 0; ie[1].iB = polyIndex; ie[1].tA = CF_VERTEX; ie[1].tB = CF_FACE;
        rf_i1 = polyIndex;
        rf_i2 = rf_i1 + 1 < 4 ? rf_i1 + 1 : 0;
        rf_v1 = sel4(pv, rf_i1);
        rf_v2 = sel4(pv, rf_i2);
        rf_normal = sel4(pn, rf_i1);
    }
    V2 sideNormal1 = mk(rf_normal.y, -rf_normal.x);
    V2 sideNormal2 = vneg(sideNormal1);
    float sideOffset1 = vdot(sideNormal1, rf_v1);
    float sideOffset2 = vdot(sideNormal2, rf_v2);
    ClipV c1[2], c2[2];
    int np = clip_segment(c1, ie, sideNormal1, sideOffset1, rf_i1);
    if (np < 2) return;
    np = clip_segment(c2, c1, sideNormal2, sideOffset2, rf_i2);
    if (np < 2) return;
    if (!primaryIsPoly) {
        m.ln = rf_normal;
        m.lp = rf_v1;
    } else {
        m.ln = sel4(nloc, rf_i1);
        m.lp = sel4(vloc, rf_i1);
    }
    int pointCount = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float separation = vdot(rf_normal, vsub(c2[i].v, rf_v1));
        if (separation <= radius) {
            V2 lp;
            unsigned key;
            if (!primaryIsPoly) {
                lp = xmulT(q, p, c2[i].v);
                key = mkkey(c2[i].iA, c2[i].iB, c2[i].tA, c2[i].tB);
            } else {
                lp = c2[i].v;
                key = mkkey(c2[i].iB, c2[i].iA, c2[i].tB, c2[i].tA);
            }
            if (pointCount == 0) { m.p0 = lp; m.k0 = key; }
            else { m.p1 = lp; m.k1 = key; }
            ++pointCount;
        }
    }
    m.count = pointCount;
}

// ---- hardcore terrain: static convex boxes (b2CollidePolygons, b2CollidePolygonAndCircle) ----
DEV V2 static_vert(const Terrain &T, int s, int k) { return mk(T.vx[k * T.nStatic + s], T.vy[k * T.nStatic + s]); }
DEV V2 static_normal(const Terrain &T, int s, int k) { return mk(T.nx[k * T.nStatic + s], T.ny[k * T.nStatic + s]); }
struct Poly4 { V2 v[4]; V2 n[4]; };
struct XFq { V2 p; Rot q; };
DEV V2 xq_mul(const XFq &T, V2 v) { return xmul(T.q, T.p, v); }
DEV V2 xq_mulT(const XFq &T, V2 v) { return xmulT(T.q, T.p, v); }
DEV XFq xq_mulT_xf(const XFq &A, const XFq &B) { // b2MulT(A, B)
    XFq C;
    C.q.s = A.q.c * B.q.s - A.q.s * B.q.c;
    C.q.c = A.q.c * B.q.c + A.q.s * B.q.s;
    C.p = rmulT(A.q, vsub(B.p, A.p));
    return C;
}
// b2FindMaxSeparation (2.3.1, exhaustive)
DEV float find_max_separation(int &edgeIndex, const Poly4 &poly1, const XFq &xf1, const Poly4 &poly2, const XFq &xf2) {
    XFq xf = xq_mulT_xf(xf2, xf1);
    int bestIndex = 0;
    float maxSeparation = -FLT_MAX;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        V2 n = rmul(xf.q, poly1.n[i]);
        V2 v1 = xq_mul(xf, poly1.v[i]);
        float si = FLT_MAX;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sij = vdot(n, vsub(poly2.v[j], v1));
            if (sij < si) si = sij;
        }
        if (si > maxSeparation) { maxSeparation = si; bestIndex = i; }
    }
    edgeIndex = bestIndex;
    return maxSeparation;
}
// b2CollidePolygons(static box A at identity, module box B); flip rule of 2.3.1
DEV void collide_polygons(Manifold &m, const Poly4 &polyA, const Poly4 &polyB, V2 pB, Rot qB) {
    m.count = 0;
    m.type = MF_FACE_A;
    m.ln = mk(0.0f, 0.0f); m.lp = mk(0.0f, 0.0f); m.p0 = mk(0.0f, 0.0f); m.p1 = mk(0.0f, 0.0f);
    m.k0 = m.k1 = 0u;
    XFq xfA; xfA.p = mk(0.0f, 0.0f); xfA.q.s = 0.0f; xfA.q.c = 1.0f;
    XFq xfB; xfB.p = pB; xfB.q = qB;
    const float totalRadius = B2_POLYGON_RADIUS + B2_POLYGON_RADIUS;
    int edgeA = 0;
    float separationA = find_max_separation(edgeA, polyA, xfA, polyB, xfB);
    if (separationA > totalRadius) return;
    int edgeB = 0;
    float separationB = find_max_separation(edgeB, polyB, xfB, polyA, xfA);
    if (separationB > totalRadius) return;
    const float k_tol = 0.1f * B2_LINEAR_SLOP;
    const bool flip = separationB > separationA + k_tol;
    const Poly4 &poly1 = flip ? polyB : polyA;
    const Poly4 &poly2 = flip ? polyA : polyB;
    const XFq xf1 = flip ? xfB : xfA, xf2 = flip ? xfA : xfB;
    const int edge1 = flip ? edgeB : edgeA;
    m.type = flip ? MF_FACE_B : MF_FACE_A;
    // b2FindIncidentEdge
    ClipV incident[2];
    {
        V2 normal1 = rmulT(xf2.q, rmul(xf1.q, sel4(poly1.n, edge1)));
        int index = 0;
        float minDot = FLT_MAX;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float dot = vdot(normal1, poly2.n[i]);
            if (dot < minDot) { minDot = dot; index = i; }
        }
        int i1 = index, i2 = i1 + 1 < 4 ? i1 + 1 : 0;
        incident[0].v = xq_mul(xf2, sel4(poly2.v, i1)); incident[0].iA = edge1; incident[0].iB = i1; incident[0].tA = CF_FACE; incident[0].tB = CF_VERTEX;
        incident[1].v = xq_mul(xf2, sel4(poly2.v, i2)); incident[1].iA = edge1; incident[1].iB = i2; incident[1].tA = CF_FACE; incident[1].tB = CF_VERTEX;
    }
    int iv1 = edge1, iv2 = edge1 + 1 < 4 ? edge1 + 1 : 0;
    V2 v11 = sel4(poly1.v, iv1), v12 = sel4(poly1.v, iv2);
    V2 localTangent = vsub(v12, v11);
    vnormalize(localTangent);
    V2 localNormal = vcross_vs(localTangent, 1.0f);
    V2 planePoint = vscale(0.5f, vadd(v11, v12));
    V2 tangent = rmul(xf1.q, localTangent);
    V2 normal = vcross_vs(tangent, 1.0f);
    v11 = xq_mul(xf1, v11);
    v12 = xq_mul(xf1, v12);
    float frontOffset = vdot(normal, v11);
    float sideOffset1 = -vdot(tangent, v11) + totalRadius;
    float sideOffset2 = vdot(tangent, v12) + totalRadius;
    ClipV c1[2], c2[2];
    int np = clip_segment(c1, incident, vneg(tangent), sideOffset1, iv1);
    if (np < 2) return;
    np = clip_segment(c2, c1, tangent, sideOffset2, iv2);
    if (np < 2) return;
    m.ln = localNormal;
    m.lp = planePoint;
    int pointCount = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float separation = vdot(normal, c2[i].v) - frontOffset;
        if (separation <= totalRadius) {
            V2 lp = xq_mulT(xf2, c2[i].v);
            unsigned key = flip ? mkkey(c2[i].iB, c2[i].iA, c2[i].tB, c2[i].tA) : mkkey(c2[i].iA, c2[i].iB, c2[i].tA, c2[i].tB);
            if (pointCount == 0) { m.p0 = lp; m.k0 = key; }
            else { m.p1 = lp; m.k1 = key; }
            ++pointCount;
        }
    }
    m.count = pointCount;
}
// b2CollidePolygonAndCircle(static box A at identity, circle B with m_p = 0)
DEV void collide_polygon_circle(Manifold &m, const Poly4 &polyA, float rB, V2 center) {
    m.count = 0;
    m.type = MF_FACE_A;
    m.ln = mk(0.0f, 0.0f); m.lp = mk(0.0f, 0.0f); m.p0 = mk(0.0f, 0.0f); m.p1 = mk(0.0f, 0.0f);
    m.k0 = m.k1 = 0u;
    V2 cLocal = center;
    int normalIndex = 0;
    float separation = -FLT_MAX;
    float radius = B2_POLYGON_RADIUS + rB;
    bool out = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s = vdot(polyA.n[i], vsub(cLocal, polyA.v[i]));
        if (s > radius) out = true;
        if (!out && s > separation) { separation = s; normalIndex = i; }
    }
    if (out) return;
    int vertIndex1 = normalIndex, vertIndex2 = vertIndex1 + 1 < 4 ? vertIndex1 + 1 : 0;
    V2 v1 = sel4(polyA.v, vertIndex1), v2 = sel4(polyA.v, vertIndex2);
    if (separation < B2_EPSILON) {
        m.count = 1;
        m.ln = sel4(polyA.n, normalIndex);
        m.lp = vscale(0.5f, vadd(v1, v2));
        return;
    }
    float u1 = vdot(vsub(cLocal, v1), vsub(v2, v1));
    float u2 = vdot(vsub(cLocal, v2), vsub(v1, v2));
    if (u1 <= 0.0f) {
        if (vdist2(cLocal, v1) > radius * radius) return;
        m.count = 1;
        m.ln = vsub(cLocal, v1);
        vnormalize(m.ln);
        m.lp = v1;
    } else if (u2 <= 0.0f) {
        if (vdist2(cLocal, v2) > radius * radius) return;
        m.count = 1;
        m.ln = vsub(cLocal, v2);
        vnormalize(m.ln);
        m.lp = v2;
    } else {
        V2 faceCenter = vscale(0.5f, vadd(v1, v2));
        float sep = vdot(vsub(cLocal, faceCenter), sel4(polyA.n, vertIndex1));
        if (sep > radius) return;
        m.count = 1;
        m.ln = sel4(polyA.n, vertIndex1);
        m.lp = faceCenter;
    }
}
DEV Poly4 static_poly(const Terrain &T, int s) {
    Poly4 P;
#pragma unroll
    for (int k = 0; k < 4; ++k) { P.v[k] = static_vert(T, s, k); P.n[k] = static_normal(T, s, k); }
    return P;
}
DEV Poly4 box_poly(float hx, float hy) {
    Poly4 P;
    P.v[0] = mk(-hx, -hy); P.v[1] = mk(hx, -hy); P.v[2] = mk(hx, hy); P.v[3] = mk(-hx, hy);
    P.n[0] = mk(0.0f, -1.0f); P.n[1] = mk(1.0f, 0.0f); P.n[2] = mk(0.0f, 1.0f); P.n[3] = mk(-1.0f, 0.0f);
    return P;
}

// =====================================================================================
// shape AABBs (b2PolygonShape/b2CircleShape::ComputeAABB)
// =====================================================================================
struct AABB { V2 lo, hi; };
DEV AABB body_aabb(int shape, float hx, float hy, V2 p, Rot q) {
    AABB bb;
    if (shape == SHAPE_BOX) {
        V2 lower = xmul(q, p, mk(-hx, -hy)), upper = lower;
        V2 v = xmul(q, p, mk(hx, -hy));
        lower = vmin2(lower, v); upper = vmax2(upper, v);
        v = xmul(q, p, mk(hx, hy));
        lower = vmin2(lower, v); upper = vmax2(upper, v);
        v = xmul(q, p, mk(-hx, hy));
        lower = vmin2(lower, v); upper = vmax2(upper, v);
        V2 r = mk(B2_POLYGON_RADIUS, B2_POLYGON_RADIUS);
        bb.lo = vsub(lower, r);
        bb.hi = vadd(upper, r);
    } else {
        V2 c = vadd(p, rmul(q, mk(0.0f, 0.0f)));
        bb.lo = mk(c.x - hx, c.y - hx);
        bb.hi = mk(c.x + hx, c.y + hx);
    }
    return bb;
}
DEV bool aabb_overlap(V2 alo, V2 ahi, V2 blo, V2 bhi) {
    V2 d1 = vsub(blo, ahi), d2 = vsub(alo, bhi);
    if (d1.x > 0.0f || d1.y > 0.0f) return false;
    if (d2.x > 0.0f || d2.y > 0.0f) return false;
    return true;
}

#endif
