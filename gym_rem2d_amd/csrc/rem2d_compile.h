// rem2d_compile.h -- host-side genotype -> phenotype -> SoA morphology for L-system genomes, native and
// multi-threaded.  Part of the single translation unit rem2d.hip; host code only.
//
// Restates, in binary64 with the same libm calls in the same order, what the Python layer does per
// individual and what is pinned bit-for-bit against the reference's fixtures there:
//   LSystem.create          Encodings/LSystem.py:144-199   (gym_rem2d_amd/encodings/lsystem.py)
//   create_robot            Modular2DEnv.py:517-563        (compiler.build_creature)
//   Standard2D / Circular2D simple_module.py:147-199,231-313, circular_module.py:138-221 (modules.py)
//   create_joint            module_utility.py:7-33         (modules._site_to_anchor_frame)
//   island order, joint rounds, modulo schedule            (compiler.island_joint_order / joint_rounds /
//                                                           pipeline_schedule)
// and writes the [creature][lane] arrays rem2d_world_reset uploads.  tests/test_compile_native.py compares
// every output word with the Python path over thousands of seeded genomes.
#ifndef REM2D_COMPILE_H
#define REM2D_COMPILE_H

#include <thread>

namespace rem2d_host {

static const int MAXN = 64;     // nodes / bodies per creature
static const double PI = 3.141592653589793; // math.pi

static inline double f32r(double v) { return (double)(float)v; } // the SWIG boundary: python double -> float32 -> double

struct TreeNode { int index, parent, type, site; }; // site: -1 none, 0 left, 1 right, 2 top

struct Genome {
    int nTypes;
    const int32_t *shape;
    const double *width, *height, *radius, *angle, *torque, *amp, *phase, *freq, *offset;
    const int32_t *ruleN, *ruleSite, *ruleRef;
};

// ---- LSystem.create ----
struct Sym { int index, parent, ref, site; bool handled; int nChild; int child[3]; };
struct Expander {
    const Genome &g;
    int maxModules;
    Sym pool[4 * MAXN];
    int nPool;
    bool overflow;
    Expander(const Genome &gg, int mm) : g(gg), maxModules(mm), nPool(0), overflow(false) {}
    int rewrite(int s, int index) {
        if (index > maxModules) return index;
        Sym &sym = pool[s];
        if (!sym.handled) {
            sym.handled = true;
            const int r = sym.ref;
            for (int c = 0; c < g.ruleN[r]; ++c) { // Rule.update
                index += 1;
                if (nPool >= 4 * MAXN) { overflow = true; return index; }
                Sym &f = pool[nPool];
                f.index = index; f.parent = pool[s].index; f.ref = g.ruleRef[r * 3 + c]; f.site = g.ruleSite[r * 3 + c];
                f.handled = false; f.nChild = 0;
                pool[s].child[pool[s].nChild++] = nPool++;
            }
        } else {
            for (int c = 0; c < sym.nChild; ++c) index = rewrite(pool[s].child[c], index);
        }
        return index;
    }
    int emit(int parentIndex, int s, TreeNode *nodes, int &nNodes, int counter) {
        if (counter > maxModules) return counter;
        if (nNodes >= MAXN) { overflow = true; return counter; }
        nodes[nNodes].index = pool[s].index; nodes[nNodes].parent = parentIndex;
        nodes[nNodes].type = pool[s].ref; nodes[nNodes].site = pool[s].site;
        ++nNodes;
        for (int c = 0; c < pool[s].nChild; ++c) {
            counter += 1;
            counter = emit(pool[pool[s].child[c]].parent, pool[s].child[c], nodes, nNodes, counter);
        }
        return counter;
    }
    int create(int treeDepth, TreeNode *nodes) {
        nPool = 1;
        pool[0].index = 0; pool[0].parent = -1; pool[0].ref = 0; pool[0].site = -1; pool[0].handled = false; pool[0].nChild = 0;
        int index = 0;
        for (int d = 0; d < treeDepth; ++d) index = rewrite(0, index);
        int nNodes = 0;
        emit(-1, 0, nodes, nNodes, 0);
        return nNodes;
    }
};

// ---- create_robot ----
struct Body { int shape; double hx, hy, x, y, angle; }; // values already rounded to binary32
struct Joint { int parent, child; double ax, ay, bx, by, torque; int node; };
struct Creature { int nBodies, nJoints; Body bodies[MAXN]; Joint joints[MAXN]; bool overflow; };

static inline double site_value(int site) { return site == 0 ? -1.0 : (site == 1 ? 1.0 : 0.0); }

static void connection_site(const Genome &g, int ptype, int site, const Body &pb, double &sx, double &sy, double &sth) {
    const double con = site_value(site);
    if (g.shape[ptype] == 1) { // Standard2D.get_global_position_of_connection_site
        const double height = g.height[ptype], width = g.width[ptype];
        double ray = con * g.angle[ptype] + PI / 2;
        while (ray > 2 * PI) ray -= 2 * PI;
        const double flip_x = (0.5 * PI < ray && ray < 1.5 * PI) ? -1.0 : 1.0;
        const double flip_y = (PI < ray && ray < 2 * PI) ? -1.0 : 1.0;
        const double s = sin(ray), c = cos(ray);
        double tb0, tb1, lr0, lr1;
        if (2 * s == 0) { tb0 = 10000; tb1 = 10000; }
        else { tb0 = (height * c) / (2 * s) * flip_y; tb1 = height / 2 * flip_y; }
        if (2 * c == 0) { lr0 = 10000; lr1 = 10000; }
        else { lr0 = width / 2 * flip_x; lr1 = (width * s) / (2 * c) * flip_x; }
        const double d_tb = sqrt(pow(tb0, 2) + pow(tb1, 2));
        const double d_lr = sqrt(pow(lr0, 2) + pow(lr1, 2));
        const double reach = d_lr < d_tb ? d_lr : d_tb;
        const double pangle = pb.angle;
        sx = (cos(pangle + ray) * reach) + pb.x;
        sy = (sin(pangle + ray) * reach) + pb.y;
        sth = pangle + ray - PI / 2;
    } else { // Circular2D
        double turn = con * g.angle[ptype];
        turn += pb.angle;
        const double r0 = cos(turn + PI / 2) * g.radius[ptype], r1 = sin(turn + PI / 2) * g.radius[ptype];
        sx = r0 + pb.x; sy = r1 + pb.y; sth = turn;
    }
}

static void build_creature(const Genome &g, const TreeNode *nodes, int nNodes, double terrainHeight, Creature &cr) {
    cr.nBodies = cr.nJoints = 0;
    cr.overflow = false;
    bool expressed[MAXN];
    int component[MAXN]; // body slot or -1
    int handled[MAXN], nHandled = 0;
    for (int i = 0; i < nNodes; ++i) { expressed[i] = false; component[i] = -1; }
    auto emit = [&](int ni, int parentBody, bool haveSite, double sx, double sy, double sth) {
        const int t = nodes[ni].type;
        double pos0 = 5, pos1 = terrainHeight + 2; // SPAWN (Modular2DEnv.py:429-432)
        double angle = 0;
        expressed[ni] = true;
        int shape; double hx, hy;
        if (g.shape[t] == 1) {
            const double n_height = g.height[t], n_width = g.width[t];
            if (parentBody >= 0) {
                const double up = sth + angle + PI / 2;
                pos0 = (cos(up) * n_height / 2) + sx;
                pos1 = (sin(up) * n_height / 2) + sy;
            }
            if (pos1 - sqrt(pow(n_width, 2) + pow(n_height, 2)) < terrainHeight) return;
            if (haveSite) angle += sth;
            shape = 1; hx = n_width / 2; hy = n_height / 2;
        } else {
            const double r = g.radius[t];
            if (parentBody >= 0) {
                const double up = sth + PI / 2;
                pos0 = (cos(up) * r) + sx;
                pos1 = (sin(up) * r) + sy;
            }
            if (haveSite) angle += sth;
            if (pos1 - r < terrainHeight) return;
            shape = 2; hx = r; hy = 0.0;
        }
        if (cr.nBodies >= MAXN) { cr.overflow = true; return; }
        const int slot = cr.nBodies++;
        Body &b = cr.bodies[slot];
        b.shape = shape; b.hx = f32r(hx); b.hy = f32r(hy); b.x = f32r(pos0); b.y = f32r(pos1); b.angle = f32r(angle);
        component[ni] = slot;
        if (haveSite) { // module_utility.create_joint
            const Body &pa = cr.bodies[parentBody];
            const double dis_a = sqrt(pow(sx - pa.x, 2) + pow(sy - pa.y, 2));
            const double ang_a = sth - pa.angle + PI / 2;
            const double dis_b = sqrt(pow(sx - b.x, 2) + pow(sy - b.y, 2));
            const double ang_b = b.angle - sth - PI / 2;
            Joint &j = cr.joints[cr.nJoints++];
            j.parent = parentBody; j.child = slot;
            j.ax = f32r(cos(ang_a) * dis_a); j.ay = f32r(sin(ang_a) * dis_a);
            j.bx = f32r(cos(ang_b) * dis_b); j.by = f32r(sin(ang_b) * dis_b);
            j.torque = f32r(g.torque[t]);
            j.node = ni;
        }
    };
    for (int i = 0; i < nNodes; ++i)
        if (nodes[i].parent == -1) { emit(i, -1, false, 0, 0, 0); handled[nHandled++] = i; }
    for (int i = 0; i < nNodes; ++i) {
        if (expressed[i]) continue;
        int p = -1;
        for (int h = 0; h < nHandled; ++h)
            if (nodes[handled[h]].index == nodes[i].parent && expressed[handled[h]]) { p = handled[h]; break; }
        if (p < 0 || component[p] < 0) continue;
        double sx, sy, sth;
        connection_site(g, nodes[p].type, nodes[i].site, cr.bodies[component[p]], sx, sy, sth);
        emit(i, component[p], true, sx, sy, sth);
        handled[nHandled++] = i;
    }
}

// ---- island order, rounds, modulo schedule ----
static void schedule(const Creature &cr, int *rounds, int *offC, int &period) {
    const int nb = cr.nBodies, nj = cr.nJoints;
    // joint edges head-inserted per body
    int edges[MAXN][MAXN], nEdges[MAXN];
    for (int b = 0; b < nb; ++b) nEdges[b] = 0;
    for (int k = 0; k < nj; ++k) {
        const int ab[2] = {cr.joints[k].parent, cr.joints[k].child};
        for (int e = 0; e < 2; ++e) {
            const int b = ab[e];
            for (int i = nEdges[b]; i > 0; --i) edges[b][i] = edges[b][i - 1];
            edges[b][0] = k;
            ++nEdges[b];
        }
    }
    int order[MAXN], nOrder = 0, stack[MAXN];
    bool jflag[MAXN], bflag[MAXN];
    for (int k = 0; k < nj; ++k) jflag[k] = false;
    for (int b = 0; b < nb; ++b) bflag[b] = false;
    for (int seed = nb - 1; seed >= 0; --seed) {
        if (bflag[seed]) continue;
        int sp = 0;
        stack[sp++] = seed;
        bflag[seed] = true;
        while (sp > 0) {
            const int b = stack[--sp];
            for (int i = 0; i < nEdges[b]; ++i) {
                const int k = edges[b][i];
                if (jflag[k]) continue;
                const int a = cr.joints[k].parent, c = cr.joints[k].child;
                const int other = a == b ? c : a;
                order[nOrder++] = k;
                jflag[k] = true;
                if (bflag[other]) continue;
                stack[sp++] = other;
                bflag[other] = true;
            }
        }
    }
    int last[MAXN], first[MAXN], lastTouch[MAXN];
    for (int b = 0; b < nb; ++b) { lastTouch[b] = -1; last[b] = 0; first[b] = -1; }
    for (int o = 0; o < nOrder; ++o) {
        const int k = order[o], a = cr.joints[k].parent, b = cr.joints[k].child;
        const int r = (lastTouch[a] > lastTouch[b] ? lastTouch[a] : lastTouch[b]) + 1;
        rounds[k] = r;
        lastTouch[a] = lastTouch[b] = r;
    }
    for (int k = 0; k < nj; ++k) {
        const int ab[2] = {cr.joints[k].parent, cr.joints[k].child};
        for (int e = 0; e < 2; ++e) {
            const int x = ab[e];
            if (rounds[k] > last[x]) last[x] = rounds[k];
            if (first[x] < 0 || rounds[k] < first[x]) first[x] = rounds[k];
        }
    }
    period = 1;
    for (int k = 0; k < nj; ++k) {
        const int a = cr.joints[k].parent, b = cr.joints[k].child;
        const int p = (last[a] > last[b] ? last[a] : last[b]) + 1 - rounds[k];
        if (p > period) period = p;
    }
    for (int b = 0; b < nb; ++b) {
        offC[b] = last[b];
        if (first[b] < 0) continue;
        int m = (period - 1 - last[b]) % period; // python modulo: non-negative for positive period
        if (m < 0) m += period;
        const int v = last[b] + m;
        if (v <= first[b] + period - 1) offC[b] = v;
    }
}

// one creature -> the [creature][lane] arrays rem2d_world_reset uploads (lane = body slot; joint data on the child's lane)
static void write_creature(const rem2d_morph *out, size_t lo, int lanes, const Creature &cr, const TreeNode *nodes, const Genome &g,
                           float lower, float upper) {
    int rounds[MAXN], offC[MAXN], period;
    schedule(cr, rounds, offC, period);
    for (int l = 0; l < lanes; ++l) {
        const size_t i = lo + l;
        ((int32_t *)out->shape)[i] = 0; ((int32_t *)out->parent)[i] = -1; ((int32_t *)out->jround)[i] = 0;
        ((float *)out->hx)[i] = 0; ((float *)out->hy)[i] = 0; ((float *)out->x)[i] = 0; ((float *)out->y)[i] = 0;
        ((float *)out->angle)[i] = 0; ((float *)out->ax)[i] = 0; ((float *)out->ay)[i] = 0;
        ((float *)out->bx)[i] = 0; ((float *)out->by)[i] = 0; ((float *)out->torque)[i] = 0;
        ((float *)out->lower)[i] = 0; ((float *)out->upper)[i] = 0;
        ((double *)out->amp)[i] = 0; ((double *)out->phase)[i] = 0; ((double *)out->freq)[i] = 0;
        ((double *)out->offset)[i] = 0; ((double *)out->istate)[i] = 0;
    }
    for (int b = 0; b < cr.nBodies; ++b) {
        const size_t i = lo + b;
        ((int32_t *)out->jround)[i] = (offC[b] << 8) | (period << 16);
        ((int32_t *)out->shape)[i] = cr.bodies[b].shape;
        ((float *)out->hx)[i] = (float)cr.bodies[b].hx; ((float *)out->hy)[i] = (float)cr.bodies[b].hy;
        ((float *)out->x)[i] = (float)cr.bodies[b].x; ((float *)out->y)[i] = (float)cr.bodies[b].y;
        ((float *)out->angle)[i] = (float)cr.bodies[b].angle;
    }
    for (int k = 0; k < cr.nJoints; ++k) {
        const Joint &j = cr.joints[k];
        const size_t i = lo + j.child;
        const int t = nodes[j.node].type;
        ((int32_t *)out->parent)[i] = j.parent;
        ((int32_t *)out->jround)[i] |= rounds[k];
        ((float *)out->ax)[i] = (float)j.ax; ((float *)out->ay)[i] = (float)j.ay;
        ((float *)out->bx)[i] = (float)j.bx; ((float *)out->by)[i] = (float)j.by;
        ((float *)out->torque)[i] = (float)j.torque;
        ((float *)out->lower)[i] = lower; ((float *)out->upper)[i] = upper;
        ((double *)out->amp)[i] = g.amp[t]; ((double *)out->phase)[i] = g.phase[t];
        ((double *)out->freq)[i] = g.freq[t]; ((double *)out->offset)[i] = g.offset[t];
        ((double *)out->istate)[i] = 0.0;
    }
}

} // namespace rem2d_host

extern "C" int rem2d_compile_lsystem(const rem2d_lsystem_genomes *G, int32_t tree_depth, int32_t max_modules,
                                     double terrain_height, int32_t lanes, const rem2d_morph *out, int32_t *n_bodies,
                                     int32_t n_threads) {
    using namespace rem2d_host;
    if (!G || !n_bodies) return fail(REM2D_E_INVALID, "NULL argument"); // (out == NULL: the body counts only)
    if (G->n_types <= 0 || G->n_types > 64) return fail(REM2D_E_INVALID, "n_types must be 1..64");
    if (lanes <= 0 || lanes > MAXN) return fail(REM2D_E_INVALID, "lanes must be 1..64");
    const int n = G->n, T = G->n_types;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads > n) n_threads = n > 0 ? n : 1;
    const float lower = (float)(-PI / 2), upper = (float)(PI / 2);
    std::vector<int> status((size_t)(n_threads > 0 ? n_threads : 1), 0);
    auto work = [&](int tid) {
        // contiguous blocks per thread: neighbouring creatures share cache lines of the output arrays
        const int per = (n + n_threads - 1) / n_threads;
        const int e0 = tid * per, e1 = e0 + per < n ? e0 + per : n;
        for (int e = e0; e < e1; ++e) {
            Genome g;
            g.nTypes = T;
            g.shape = G->mod_shape + (size_t)e * T;
            g.width = G->mod_width + (size_t)e * T; g.height = G->mod_height + (size_t)e * T;
            g.radius = G->mod_radius + (size_t)e * T; g.angle = G->mod_angle + (size_t)e * T;
            g.torque = G->mod_torque + (size_t)e * T;
            g.amp = G->ctl_amp + (size_t)e * T; g.phase = G->ctl_phase + (size_t)e * T;
            g.freq = G->ctl_freq + (size_t)e * T; g.offset = G->ctl_offset + (size_t)e * T;
            g.ruleN = G->rule_n + (size_t)e * T; g.ruleSite = G->rule_site + (size_t)e * T * 3;
            g.ruleRef = G->rule_ref + (size_t)e * T * 3;
            TreeNode nodes[MAXN];
            Expander ex(g, max_modules);
            const int nNodes = ex.create(tree_depth, nodes);
            Creature cr;
            build_creature(g, nodes, nNodes, terrain_height, cr);
            if (ex.overflow || cr.overflow || cr.nBodies > lanes) { status[tid] = 1; n_bodies[e] = -1; continue; }
            if (out) write_creature(out, (size_t)e * lanes, lanes, cr, nodes, g, lower, upper); // (out == NULL: count only)
            n_bodies[e] = cr.nBodies;
        }
    };
    if (n_threads <= 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
        for (auto &t : th) t.join();
    }
    for (int s : status)
        if (s) return fail(REM2D_E_INVALID, "a creature has more bodies than `lanes` (n_bodies = -1 marks it)");
    return REM2D_OK;
}

// Phenotype trees of ANY encoding (direct, network, cellular ...: whatever genome.create() returned) -> morphology.
// The tree is given node by node in Tree.getNodes() order with each node's own module and controller parameters
// (Direct_Encoding.py:18-27 keeps a module object per node; Network_Encoding.py:97-129 sets one per query), so the
// "type" table of build_creature is simply the node list of that creature.
extern "C" int rem2d_compile_trees(const rem2d_tree_batch *B, double terrain_height, int32_t lanes, const rem2d_morph *out,
                                   int32_t *n_bodies, int32_t n_threads) {
    using namespace rem2d_host;
    if (!B || !n_bodies) return fail(REM2D_E_INVALID, "NULL argument"); // (out == NULL: the body counts only)
    if (B->max_nodes <= 0 || B->max_nodes > MAXN) return fail(REM2D_E_INVALID, "max_nodes must be 1..64");
    if (lanes <= 0 || lanes > MAXN) return fail(REM2D_E_INVALID, "lanes must be 1..64");
    const int n = B->n, M = B->max_nodes;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads > n) n_threads = n > 0 ? n : 1;
    const float lower = (float)(-PI / 2), upper = (float)(PI / 2);
    std::vector<int> status((size_t)(n_threads > 0 ? n_threads : 1), 0);
    auto work = [&](int tid) {
        const int per = (n + n_threads - 1) / n_threads;
        const int e0 = tid * per, e1 = e0 + per < n ? e0 + per : n;
        for (int e = e0; e < e1; ++e) {
            const size_t r = (size_t)e * M;
            const int nNodes = B->node_count[e];
            if (nNodes < 0 || nNodes > M) { status[tid] = 2; n_bodies[e] = -1; continue; }
            Genome g;
            g.nTypes = nNodes;
            g.shape = B->shape + r;
            g.width = B->width + r; g.height = B->height + r; g.radius = B->radius + r; g.angle = B->angle + r;
            g.torque = B->torque + r;
            g.amp = B->ctl_amp + r; g.phase = B->ctl_phase + r; g.freq = B->ctl_freq + r; g.offset = B->ctl_offset + r;
            g.ruleN = nullptr; g.ruleSite = nullptr; g.ruleRef = nullptr;
            TreeNode nodes[MAXN];
            for (int i = 0; i < nNodes; ++i) {
                nodes[i].index = B->index[r + i];
                nodes[i].parent = B->parent[r + i];
                nodes[i].site = B->site[r + i];
                nodes[i].type = i;
            }
            Creature cr;
            build_creature(g, nodes, nNodes, terrain_height, cr);
            if (cr.overflow || cr.nBodies > lanes) { status[tid] = 1; n_bodies[e] = -1; continue; }
            if (out) write_creature(out, (size_t)e * lanes, lanes, cr, nodes, g, lower, upper); // (out == NULL: count only)
            n_bodies[e] = cr.nBodies;
        }
    };
    if (n_threads <= 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
        for (auto &t : th) t.join();
    }
    for (int s : status) {
        if (s == 2) return fail(REM2D_E_INVALID, "node_count out of range");
        if (s) return fail(REM2D_E_INVALID, "a creature has more bodies than `lanes` (n_bodies = -1 marks it)");
    }
    return REM2D_OK;
}

// ---- Network encoding (Network_Encoding.py:42-222): a function queried per connection site grows the tree ----
// Restates NNEncoding.create / _grow / _query / _emit (gym_rem2d_amd/encodings/network.py, the mirror of the reference's
// update / iterate / create / recursiveNodeGen) and Standard2D / Circular2D.setMorph + limitWH, Controller.setControl +
// minMax, in binary64 with the same libm calls in the same order as the Python floats.  The genome of this build is the
// synthetic feed-forward CPPN (neat-python is absent): per hidden node one of four activation functions.
namespace rem2d_host {

static inline double pymax(double a, double b) { return b > a ? b : a; } // python's max(a, b): first maximal element
static inline double pymin(double a, double b) { return b < a ? b : a; }

struct NetSym { int index, parent, ref, site; bool handled; int nChild; int child[3]; int node; };

struct NetExpander {
    const rem2d_network_genomes &G;
    const int e, T, H, maxDepth, maxModules;
    NetSym pool[4 * MAXN];
    int nPool;
    bool overflow;
    // per-symbol module / controller parameters (the child's own copies)
    int shape[4 * MAXN];
    double width[4 * MAXN], height[4 * MAXN], radius[4 * MAXN], angle[4 * MAXN], torque[4 * MAXN];
    double amp[4 * MAXN], phase[4 * MAXN], freq[4 * MAXN], offset[4 * MAXN];
    NetExpander(const rem2d_network_genomes &g, int ee, int depth)
        : G(g), e(ee), T(g.n_types), H(g.n_hidden), maxDepth(depth), maxModules(g.max_modules), nPool(0), overflow(false) {}
    double proto(const double *a, int t) const { return a[(size_t)e * T + t]; }
    void activate(const double x[3], double out[10]) const {
        double h[64];
        const double *w1 = G.w1 + (size_t)e * H * 4;
        const int32_t *a1 = G.a1 + (size_t)e * H;
        const double *w2 = G.w2 + (size_t)e * 10 * (H + 1);
        for (int k = 0; k < H; ++k) {
            double s = 0.0; // python sum(): 0 + w0 x0 + w1 x1 + w2 x2 + w3 * 1.0, left to right
            s = s + w1[k * 4 + 0] * x[0];
            s = s + w1[k * 4 + 1] * x[1];
            s = s + w1[k * 4 + 2] * x[2];
            s = s + w1[k * 4 + 3] * 1.0;
            switch (a1[k]) {
            case 0: h[k] = tanh(s); break;
            case 1: h[k] = sin(s); break;
            case 2: h[k] = exp(-s * s) * 2 - 1; break;
            default: h[k] = pymax(-1.0, pymin(1.0, s)); break;
            }
        }
        for (int o = 0; o < 10; ++o) {
            double s = 0.0;
            for (int k = 0; k < H; ++k) s = s + w2[o * (H + 1) + k] * h[k];
            s = s + w2[o * (H + 1) + H] * 1.0;
            out[o] = tanh(s);
        }
    }
    int query(int index, int s, int depth) { // NNEncoding._query: one network query per free connection site
        if (depth > maxDepth || index > maxModules) return index;
        const NetSym parent = pool[s];
        if (shape[s] != 1) return index; // circles offer no connection sites (circular_module.py:31-53)
        static const double siteValue[3] = {-1.0, 1.0, 0.0}; // BoxConnection left, right, top: con.value[0]
        for (int c = 0; c < 3; ++c) {
            double x[3], out[10];
            x[0] = 1.0 - (2.0 * ((double)depth / (double)maxDepth));
            x[1] = 1.0 - (2.0 * ((double)(parent.ref + 1) / (double)T));
            x[2] = siteValue[c];
            activate(x, out);
            if (out[0] > 0.5) {
                out[1] = pymax(-1., pymin(1., out[1]));
                int ref = (int)(((out[1] * 0.5) + 0.5) * (double)(T - 1)); // int(): towards zero
                ref = ref < 0 ? 0 : (ref > T - 1 ? T - 1 : ref);
                if (nPool >= 4 * MAXN) { overflow = true; return index; }
                const int k = nPool++;
                NetSym &f = pool[k];
                f.index = index; f.parent = parent.index; f.ref = ref; f.site = c; f.handled = false; f.nChild = 0; f.node = -1;
                // child.module = deepcopy(moduleList[ref]); child.module.setMorph(out[2], out[3], out[4])
                shape[k] = G.mod_shape[(size_t)e * T + ref];
                torque[k] = proto(G.mod_torque, ref);
                width[k] = proto(G.mod_width, ref); height[k] = proto(G.mod_height, ref); radius[k] = proto(G.mod_radius, ref);
                if (shape[k] == 1) {
                    width[k] = (out[2] * 0.5 * (G.box_max_width - G.box_min_width)) + 0.5 * (G.box_max_width - G.box_min_width);
                    height[k] = (out[2] * 0.5 * (G.box_max_height - G.box_min_height)) + 0.5 * (G.box_max_height - G.box_min_height);
                    angle[k] = G.box_min_angle + (((out[4] + 1.0) * 0.5) * (G.box_max_angle - G.box_min_angle));
                    height[k] = pymin(pymax(height[k], G.box_min_height), G.box_max_height);
                    width[k] = pymin(pymax(width[k], G.box_min_width), G.box_max_width);
                    angle[k] = pymin(pymax(angle[k], G.box_min_angle), G.box_max_angle);
                } else {
                    radius[k] = out[2] + 1.5;
                    angle[k] = G.circle_min_angle + (((out[4] + 1.0) * 0.5) * (G.circle_max_angle - G.circle_min_angle));
                    radius[k] = pymin(pymax(radius[k], G.circle_min_radius), G.circle_max_radius);
                    angle[k] = pymin(pymax(angle[k], G.circle_min_angle), G.circle_max_angle);
                }
                // ctrl = deepcopy(moduleList[ref].controller); ctrl.setControl(out[5..8], moduleList[ref].angle)
                const double pangle = proto(G.mod_angle, ref);
                amp[k] = ((out[5] + 1.0) * 0.5) * G.ctl_max_amp;
                phase[k] = out[6] * G.ctl_max_phase;
                offset[k] = out[7] * G.ctl_max_offset;
                freq[k] = out[8] * G.ctl_max_freq;
                amp[k] = pymin(pymax(amp[k], 0.0), G.ctl_max_amp);
                phase[k] = pymin(pymax(phase[k], -G.ctl_max_phase), G.ctl_max_phase);
                freq[k] = pymin(pymax(freq[k], -G.ctl_max_freq), G.ctl_max_freq);
                if (offset[k] > pangle / 2) offset[k] = pangle / 2;
                else if (offset[k] < -pangle / 2) offset[k] = -pangle / 2;
                pool[s].child[pool[s].nChild++] = k;
                index += 1;
            }
        }
        return index;
    }
    int grow(int s, int index, int depth) {
        if (!pool[s].handled) {
            pool[s].handled = true;
            index = query(index, s, depth);
        } else {
            for (int c = 0; c < pool[s].nChild; ++c) index = grow(pool[s].child[c], index, depth + 1);
        }
        return index;
    }
    int emit(int parentIndex, int s, TreeNode *nodes, int *symOf, int &nNodes, int counter) {
        if (counter > 20) return counter; // MAX_MODULES of the emitter (Network_Encoding.py / encodings/network.py)
        if (nNodes >= MAXN) { overflow = true; return counter; }
        nodes[nNodes].index = pool[s].index; nodes[nNodes].parent = parentIndex;
        nodes[nNodes].type = nNodes; nodes[nNodes].site = pool[s].site;
        symOf[nNodes] = s;
        ++nNodes;
        for (int c = 0; c < pool[s].nChild; ++c) {
            counter += 1;
            counter = emit(pool[pool[s].child[c]].parent, pool[s].child[c], nodes, symOf, nNodes, counter);
        }
        return counter;
    }
    int create(TreeNode *nodes, int *symOf) {
        // base = _Symbol(0, moduleList[0], -1): a copy of prototype 0 (no setMorph), its prototype controller
        nPool = 1;
        NetSym &b = pool[0];
        b.index = 0; b.parent = -1; b.ref = -1; b.site = -1; b.handled = false; b.nChild = 0; b.node = -1;
        shape[0] = G.mod_shape[(size_t)e * T + 0];
        width[0] = proto(G.mod_width, 0); height[0] = proto(G.mod_height, 0); radius[0] = proto(G.mod_radius, 0);
        angle[0] = proto(G.mod_angle, 0); torque[0] = proto(G.mod_torque, 0);
        amp[0] = proto(G.ctl_amp, 0); phase[0] = proto(G.ctl_phase, 0); freq[0] = proto(G.ctl_freq, 0); offset[0] = proto(G.ctl_offset, 0);
        int index = 1;
        for (int d = 0; d < maxDepth; ++d) index = grow(0, index, 0);
        int nNodes = 0;
        emit(-1, 0, nodes, symOf, nNodes, 0);
        return nNodes;
    }
};

} // namespace rem2d_host

extern "C" int rem2d_compile_network(const rem2d_network_genomes *G, int32_t tree_depth, double terrain_height, int32_t lanes,
                                     const rem2d_morph *out, int32_t *n_bodies, int32_t n_threads) {
    using namespace rem2d_host;
    if (!G || !n_bodies) return fail(REM2D_E_INVALID, "NULL argument"); // (out == NULL: the body counts only)
    if (G->n_types <= 0 || G->n_types > 64) return fail(REM2D_E_INVALID, "n_types must be 1..64");
    if (G->n_hidden <= 0 || G->n_hidden > 64) return fail(REM2D_E_INVALID, "n_hidden must be 1..64");
    if (lanes <= 0 || lanes > MAXN) return fail(REM2D_E_INVALID, "lanes must be 1..64");
    if (tree_depth <= 0) return fail(REM2D_E_INVALID, "tree_depth must be positive");
    const int n = G->n;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads > n) n_threads = n > 0 ? n : 1;
    const float lower = (float)(-PI / 2), upper = (float)(PI / 2);
    std::vector<int> status((size_t)(n_threads > 0 ? n_threads : 1), 0);
    auto work = [&](int tid) {
        const int per = (n + n_threads - 1) / n_threads;
        const int e0 = tid * per, e1 = e0 + per < n ? e0 + per : n;
        std::vector<NetExpander> holder; // the expander is large: keep it off the thread's stack
        for (int e = e0; e < e1; ++e) {
            holder.clear();
            holder.emplace_back(*G, e, tree_depth);
            NetExpander &ex = holder[0];
            TreeNode nodes[MAXN];
            int symOf[MAXN];
            const int nNodes = ex.create(nodes, symOf);
            // per-node parameter tables for build_creature (type = node position)
            int32_t nshape[MAXN];
            double nw[MAXN], nh[MAXN], nr[MAXN], na[MAXN], nt[MAXN], camp[MAXN], cph[MAXN], cfr[MAXN], cof[MAXN];
            for (int i = 0; i < nNodes; ++i) {
                const int s = symOf[i];
                nshape[i] = ex.shape[s]; nw[i] = ex.width[s]; nh[i] = ex.height[s]; nr[i] = ex.radius[s]; na[i] = ex.angle[s];
                nt[i] = ex.torque[s]; camp[i] = ex.amp[s]; cph[i] = ex.phase[s]; cfr[i] = ex.freq[s]; cof[i] = ex.offset[s];
            }
            Genome g;
            g.nTypes = nNodes;
            g.shape = nshape; g.width = nw; g.height = nh; g.radius = nr; g.angle = na; g.torque = nt;
            g.amp = camp; g.phase = cph; g.freq = cfr; g.offset = cof;
            g.ruleN = nullptr; g.ruleSite = nullptr; g.ruleRef = nullptr;
            Creature cr;
            build_creature(g, nodes, nNodes, terrain_height, cr);
            if (ex.overflow || cr.overflow || cr.nBodies > lanes) { status[tid] = 1; n_bodies[e] = -1; continue; }
            if (out) write_creature(out, (size_t)e * lanes, lanes, cr, nodes, g, lower, upper); // (out == NULL: count only)
            n_bodies[e] = cr.nBodies;
        }
    };
    if (n_threads <= 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
        for (auto &t : th) t.join();
    }
    for (int s : status)
        if (s) return fail(REM2D_E_INVALID, "a creature has more bodies than `lanes` (n_bodies = -1 marks it)");
    return REM2D_OK;
}

// ---------------------------------------------------------------------------------------------------
// DirectEncoding.mutate (Encodings/Direct_Encoding.py:82-139) for a whole array population, in place, native and
// multi-threaded (round 5; REM2D_main.py:280-298 mutates every offspring of every generation: at 1 M individuals the numpy
// form of gym_rem2d_amd.population.DirectPopulation costs ~8 s per generation, this ~0.1 s).  One individual after the other
// per thread, the reference's algorithm statement by statement -- the children walk with its list-iterator semantics (the
// sibling that slides into a removed child's slot is skipped; a hit on a child of the ROOT removes nothing, but the child is
// not descended into), countModules() recounts, the growth loop over the free sites with the same iterator semantics, then
// module.mutate / limitWH and controller.mutate / minMax of every visited node.  Not the reference's random stream (python's
// Mersenne twister): a xoshiro256** per individual seeded from (seed, individual) -- the same distributions, the same result
// whatever the thread count.  Left out: the order in which freed sites are re-appended to availableConnections (here always
// left, right, top).
// ---------------------------------------------------------------------------------------------------
namespace rem2d_host {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    Rng(uint64_t seed, uint64_t stream) {
        uint64_t x = seed ^ (stream * 0xd1342543de82ef95ull + 0x632be59bd9b4e019ull);
        for (int i = 0; i < 4; ++i) s[i] = splitmix(x);
    }
    static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * 0x1.0p-53; }            // [0, 1)
    double uniform(double lo, double hi) { return lo + (hi - lo) * uniform(); }
    int randint(int n) { return (int)(uniform() * n); }                        // 0 .. n - 1
    double gauss(double mu, double sigma) {                                     // Marsaglia's polar method
        double u, v, q;
        do { u = 2.0 * uniform() - 1.0; v = 2.0 * uniform() - 1.0; q = u * u + v * v; } while (q >= 1.0 || q == 0.0);
        return mu + sigma * u * std::sqrt(-2.0 * std::log(q) / q);
    }
};

struct MutNode {
    int shape, site, nChild, child[3];
    bool freeSite[3];
    double width, height, radius, angle, torque, amp, phase, freq, offset;
};
struct TreeMutator {
    const rem2d_tree_population &P;
    Rng rng;
    MutNode nd[2 * MAXN];
    int nNodes;          // pool entries handed out
    int nModules;        // countModules()
    double morph, rate, sigma;
    TreeMutator(const rem2d_tree_population &p, uint64_t seed, uint64_t e, double m, double r, double sg)
        : P(p), rng(seed, e), nNodes(0), nModules(0), morph(m), rate(r), sigma(sg) {}
    int count(int n) const {
        int c = 1;
        for (int k = 0; k < nd[n].nChild; ++k) c += count(nd[n].child[k]);
        return c;
    }
    void mutate_params(MutNode &x) {
        if (x.shape == 1) { // Standard2D.mutate, limitWH (simple_module.py:55-85)
            if (rng.uniform() < morph) x.width = rng.gauss(x.width, sigma);
            if (rng.uniform() < morph) x.height = rng.gauss(x.height, sigma);
            if (rng.uniform() < morph) x.angle = rng.gauss(x.angle, sigma * PI);
            x.height = std::min(std::max(x.height, P.box_min_height), P.box_max_height);
            x.width = std::min(std::max(x.width, P.box_min_width), P.box_max_width);
            x.angle = std::min(std::max(x.angle, P.box_min_angle), P.box_max_angle);
        } else {            // Circular2D.mutate, limitWH (circular_module.py:55-80)
            if (rng.uniform() < morph) x.radius = rng.gauss(x.radius, sigma);
            if (rng.uniform() < morph) x.angle = rng.gauss(x.angle, sigma * PI);
            x.radius = std::min(std::max(x.radius, P.circle_min_radius), P.circle_max_radius);
            x.angle = std::min(std::max(x.angle, P.circle_min_angle), P.circle_max_angle);
        }
        // Controller.mutate: value += gauss(value, sigma) (m_controller.py:50-58), then minMax(module.angle)
        if (rng.uniform() < rate) x.amp += rng.gauss(x.amp, sigma);
        if (rng.uniform() < rate) x.phase += rng.gauss(x.phase, sigma);
        if (rng.uniform() < rate) x.freq += rng.gauss(x.freq, sigma * 0.1);
        if (rng.uniform() < rate) x.offset += rng.gauss(x.offset, sigma);
        x.amp = std::min(std::max(x.amp, 0.0), P.ctl_max_amp);
        x.phase = std::min(std::max(x.phase, -P.ctl_max_phase), P.ctl_max_phase);
        x.freq = std::min(std::max(x.freq, -P.ctl_max_freq), P.ctl_max_freq);
        if (x.offset > x.angle / 2) x.offset = x.angle / 2;
        else if (x.offset < -x.angle / 2) x.offset = -x.angle / 2;
    }
    void mutate_node(int n, int depth) {
        nModules = count(0);
        // for mod in node.children: (a list iterator: the index advances whatever happens to the list)
        for (int i = 0; i < nd[n].nChild; ++i) {
            const int mod = nd[n].child[i];
            if (rng.uniform() < morph / 2.0 / (double)nModules) {
                if (depth != 0) {
                    nd[n].freeSite[nd[mod].site] = true;                 // availableConnections.append(...)
                    for (int k = i; k + 1 < nd[n].nChild; ++k) nd[n].child[k] = nd[n].child[k + 1];
                    nd[n].nChild -= 1;                                   // children.remove(mod): the next sibling slides into slot i
                    nModules = count(0);
                }
            } else {
                mutate_node(mod, depth + 1);
            }
        }
        // for con in node.availableConnections: (the same iterator semantics over the free sites, in the order left, right, top)
        if (nd[n].shape == 1) {
            int freeList[3], nFree = 0;
            for (int sidx = 0; sidx < 3; ++sidx)
                if (nd[n].freeSite[sidx]) freeList[nFree++] = sidx;
            for (int k = 0; k < nFree; ++k) {
                nModules = count(0);
                if (nModules < P.max_modules && depth < P.max_depth && rng.uniform() < morph / (double)nModules) {
                    const int ref = rng.randint(P.n_box + P.n_circle);
                    if (nNodes >= 2 * MAXN || nd[n].nChild >= 3) continue;
                    MutNode &c = nd[nNodes];
                    c.shape = ref < P.n_box ? 1 : 2;                     // copy.deepcopy(self.moduleList[ref]): the prototypes keep their defaults
                    c.width = c.shape == 1 ? P.proto_box_width : 0.0; c.height = c.shape == 1 ? P.proto_box_height : 0.0;
                    c.radius = c.shape == 1 ? 0.0 : P.proto_circle_radius;
                    c.angle = P.proto_angle; c.torque = P.proto_torque;
                    c.amp = rng.uniform(0.0, P.ctl_max_amp);             // Controller()
                    c.phase = rng.uniform(-P.ctl_max_phase, P.ctl_max_phase);
                    c.freq = rng.uniform(-P.ctl_max_freq, P.ctl_max_freq);
                    c.offset = rng.uniform(-P.ctl_max_offset, P.ctl_max_offset);
                    c.site = freeList[k]; c.nChild = 0;
                    for (int q = 0; q < 3; ++q) c.freeSite[q] = c.shape == 1;
                    nd[n].child[nd[n].nChild++] = nNodes++;
                    nd[n].freeSite[freeList[k]] = false;
                    for (int q = k; q + 1 < nFree; ++q) freeList[q] = freeList[q + 1];   // availableConnections.remove(con) ...
                    nFree -= 1;                                                          // ... and the iterator moves on past the next one
                }
            }
        }
        mutate_params(nd[n]);
    }
};

} // namespace rem2d_host

extern "C" int rem2d_mutate_trees(const rem2d_tree_population *P, double morph_rate, double rate, double sigma, uint64_t seed,
                                  int32_t n_threads) {
    using namespace rem2d_host;
    if (!P || !P->node_count || !P->parent || !P->site || !P->shape || !P->width || !P->height || !P->radius || !P->angle ||
        !P->torque || !P->ctl_amp || !P->ctl_phase || !P->ctl_freq || !P->ctl_offset)
        return fail(REM2D_E_INVALID, "NULL argument");
    const int n = P->n, M = P->max_nodes;
    if (M <= 0 || M > MAXN || P->max_modules > M || P->max_modules <= 0) return fail(REM2D_E_INVALID, "max_nodes must be 1..64 and >= max_modules");
    if (P->n_box < 0 || P->n_circle < 0 || P->n_box + P->n_circle <= 0) return fail(REM2D_E_INVALID, "no module prototypes");
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads > n) n_threads = n > 0 ? n : 1;
    std::vector<int> status((size_t)(n_threads > 0 ? n_threads : 1), 0);
    auto work = [&](int tid) {
        const int per = (n + n_threads - 1) / n_threads;
        const int e0 = tid * per, e1 = e0 + per < n ? e0 + per : n;
        std::vector<TreeMutator> holder;
        for (int e = e0; e < e1; ++e) {
            holder.clear();
            holder.emplace_back(*P, seed, (uint64_t)e, morph_rate, rate, sigma);
            TreeMutator &T = holder[0];
            const size_t o = (size_t)e * M;
            const int cnt = P->node_count[e];
            if (cnt <= 0 || cnt > M) { status[tid] = 1; continue; }
            for (int i = 0; i < cnt; ++i) {     // the node table (Tree.getNodes() order) -> child lists in that order
                MutNode &x = T.nd[i];
                x.shape = P->shape[o + i]; x.site = P->site[o + i]; x.nChild = 0;
                x.width = P->width[o + i]; x.height = P->height[o + i]; x.radius = P->radius[o + i]; x.angle = P->angle[o + i];
                x.torque = P->torque[o + i]; x.amp = P->ctl_amp[o + i]; x.phase = P->ctl_phase[o + i]; x.freq = P->ctl_freq[o + i];
                x.offset = P->ctl_offset[o + i];
                for (int q = 0; q < 3; ++q) x.freeSite[q] = x.shape == 1;
                if (i > 0) {
                    const int p = P->parent[o + i];
                    if (p < 0 || p >= i || x.site < 0 || x.site > 2 || T.nd[p].nChild >= 3) { status[tid] = 1; break; }
                    T.nd[p].child[T.nd[p].nChild++] = i;
                    T.nd[p].freeSite[x.site] = false;
                }
            }
            if (status[tid]) continue;
            T.nNodes = cnt;
            T.mutate_node(0, 0);
            // reassignIndices: back into Tree.getNodes() order
            int stackN[2 * MAXN], stackP[2 * MAXN], top = 0, outN = 0;
            stackN[top] = 0; stackP[top++] = -1;
            while (top > 0) {
                const int cur = stackN[--top], par = stackP[top];
                if (outN >= M) { status[tid] = 1; break; }
                const MutNode &x = T.nd[cur];
                const int me = outN++;
                P->parent[o + me] = par; P->site[o + me] = cur == 0 ? -1 : x.site; P->shape[o + me] = x.shape;
                P->width[o + me] = x.width; P->height[o + me] = x.height; P->radius[o + me] = x.radius; P->angle[o + me] = x.angle;
                P->torque[o + me] = x.torque; P->ctl_amp[o + me] = x.amp; P->ctl_phase[o + me] = x.phase;
                P->ctl_freq[o + me] = x.freq; P->ctl_offset[o + me] = x.offset;
                for (int k = x.nChild - 1; k >= 0; --k) { stackN[top] = x.child[k]; stackP[top++] = me; } // (first child on top)
            }
            for (int i = outN; i < M; ++i) {     // padding
                P->parent[o + i] = -1; P->site[o + i] = -1; P->shape[o + i] = 0;
                P->width[o + i] = P->height[o + i] = P->radius[o + i] = P->angle[o + i] = P->torque[o + i] = 0.0;
                P->ctl_amp[o + i] = P->ctl_phase[o + i] = P->ctl_freq[o + i] = P->ctl_offset[o + i] = 0.0;
            }
            P->node_count[e] = outN;
        }
    };
    if (n_threads <= 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
        for (auto &t : th) t.join();
    }
    for (int st : status)
        if (st) return fail(REM2D_E_INVALID, "mutate_trees: a node table is not a tree in Tree.getNodes() order (or outgrew max_nodes)");
    return REM2D_OK;
}

#endif
