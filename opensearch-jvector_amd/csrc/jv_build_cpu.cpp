// jv_build_cpu.cpp — CPU index construction helpers (see jv_build.h: write side, not the hot path).
#include "jv_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Scored {
    float score;
    int32_t node;
};
// total order (score desc, node asc), the same order the search path uses
inline bool better(const Scored& a, const Scored& b) {
    return a.score > b.score || (a.score == b.score && a.node < b.node);
}

inline float dotf(const float* a, const float* b, int d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= d; i += 8)
        for (int j = 0; j < 8; j++) acc[j] += a[i + j] * b[i + j];
    float s = 0;
    for (; i < d; i++) s += a[i] * b[i];
    for (int j = 0; j < 8; j++) s += acc[j];
    return s;
}
inline float l2f(const float* a, const float* b, int d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= d; i += 8)
        for (int j = 0; j < 8; j++) {
            float t = a[i + j] - b[i + j];
            acc[j] += t * t;
        }
    float s = 0;
    for (; i < d; i++) {
        float t = a[i] - b[i];
        s += t * t;
    }
    for (int j = 0; j < 8; j++) s += acc[j];
    return s;
}

struct Space {
    const float* v;
    int d;
    int sim;
    std::vector<float> norm2;  // cosine
    const float* row(int i) const { return v + (size_t)i * d; }
    float score(const float* a, const float* b, float na, float nb) const {
        if (sim == 0) return 1.0f / (1.0f + l2f(a, b, d));
        if (sim == 1) return (1.0f + dotf(a, b, d)) * 0.5f;
        return (1.0f + dotf(a, b, d) / std::sqrt(na * nb)) * 0.5f;
    }
    float score_nodes(int a, int b) const {
        return score(row(a), row(b), sim == 2 ? norm2[a] : 0.f, sim == 2 ? norm2[b] : 0.f);
    }
};

struct Graph {
    int n, cap;  // cap = row capacity (R * overflow)
    std::vector<int32_t> adj;
    std::vector<int32_t> deg;
    Graph(int n_, int cap_) : n(n_), cap(cap_), adj((size_t)n_ * cap_, -1), deg(n_, 0) {}
    int32_t* row(int i) { return adj.data() + (size_t)i * cap; }
    const int32_t* row(int i) const { return adj.data() + (size_t)i * cap; }
};

// best-first beam search over the nodes inserted so far; returns the top-L scored nodes (desc)
void beam_search(const Space& sp, const Graph& g, int entry, int target, int L, std::vector<Scored>& out,
                 std::vector<uint32_t>& visited_stamp, uint32_t stamp) {
    const float* q = sp.row(target);
    float qn = sp.sim == 2 ? sp.norm2[target] : 0.f;
    auto sc = [&](int node) { return sp.score(q, sp.row(node), qn, sp.sim == 2 ? sp.norm2[node] : 0.f); };
    std::vector<Scored> cand;  // max-heap by better()
    auto cand_cmp = [](const Scored& a, const Scored& b) { return better(b, a); };
    std::vector<Scored> res;  // min-heap: worst on top
    auto res_cmp = [](const Scored& a, const Scored& b) { return better(a, b); };
    visited_stamp[entry] = stamp;
    cand.push_back({sc(entry), entry});
    std::vector<Scored> pool;  // every scored node (Vamana prunes from the visited set, not just the beam)
    pool.push_back(cand[0]);
    while (!cand.empty()) {
        std::pop_heap(cand.begin(), cand.end(), cand_cmp);
        Scored c = cand.back();
        cand.pop_back();
        if ((int)res.size() >= L && better(res.front(), c)) break;
        if ((int)res.size() < L) {
            res.push_back(c);
            std::push_heap(res.begin(), res.end(), res_cmp);
        } else if (better(c, res.front())) {
            std::pop_heap(res.begin(), res.end(), res_cmp);
            res.back() = c;
            std::push_heap(res.begin(), res.end(), res_cmp);
        }
        const int32_t* nb = g.row(c.node);
        int dg = g.deg[c.node];
        for (int i = 0; i < dg; i++) {
            int nn = nb[i];
            if (visited_stamp[nn] == stamp) continue;
            visited_stamp[nn] = stamp;
            Scored s{sc(nn), nn};
            pool.push_back(s);
            if ((int)res.size() >= L && better(res.front(), s)) continue;
            cand.push_back(s);
            std::push_heap(cand.begin(), cand.end(), cand_cmp);
        }
    }
    std::sort(pool.begin(), pool.end(), better);
    if ((int)pool.size() > 4 * L) pool.resize(4 * L);
    out.swap(pool);
}

// jvector-style diversity selection: alpha sweeps 1.0, 1.2, ... <= alpha; a candidate is kept if no
// already-selected neighbour is more similar to it than (its similarity to the base) * alpha.
void robust_prune(const Space& sp, int base, std::vector<Scored>& cands, float alpha, int R,
                  std::vector<int32_t>& out) {
    std::sort(cands.begin(), cands.end(), better);
    cands.erase(std::unique(cands.begin(), cands.end(),
                            [](const Scored& a, const Scored& b) { return a.node == b.node; }),
                cands.end());
    out.clear();
    std::vector<char> taken(cands.size(), 0);
    for (float a = 1.0f; a <= alpha + 1e-6f && (int)out.size() < R; a += 0.2f) {
        for (size_t i = 0; i < cands.size() && (int)out.size() < R; i++) {
            if (taken[i] || cands[i].node == base) continue;
            bool diverse = true;
            for (int s : out) {
                if (sp.score_nodes(cands[i].node, s) > cands[i].score * a) {
                    diverse = false;
                    break;
                }
            }
            if (diverse) {
                taken[i] = 1;
                out.push_back(cands[i].node);
            }
        }
    }
}

void set_row(Graph& g, int node, const std::vector<int32_t>& nb) {
    int32_t* r = g.row(node);
    int k = std::min<int>((int)nb.size(), g.cap);
    for (int i = 0; i < k; i++) r[i] = nb[i];
    for (int i = k; i < g.cap; i++) r[i] = -1;
    g.deg[node] = k;
}

void add_backlink(const Space& sp, Graph& g, int s, int u, float alpha, int R) {
    int32_t* r = g.row(s);
    for (int i = 0; i < g.deg[s]; i++)
        if (r[i] == u) return;
    if (g.deg[s] < g.cap) {
        r[g.deg[s]++] = u;
        return;
    }
    std::vector<Scored> c;
    c.reserve(g.deg[s] + 1);
    for (int i = 0; i < g.deg[s]; i++) c.push_back({sp.score_nodes(s, r[i]), r[i]});
    c.push_back({sp.score_nodes(s, u), u});
    std::vector<int32_t> sel;
    robust_prune(sp, s, c, alpha, R, sel);
    set_row(g, s, sel);
}

int approx_medoid(const Space& sp, const std::vector<int32_t>& ids) {
    int d = sp.d;
    std::vector<double> c(d, 0.0);
    for (int id : ids) {
        const float* r = sp.row(id);
        for (int j = 0; j < d; j++) c[j] += r[j];
    }
    std::vector<float> cf(d);
    for (int j = 0; j < d; j++) cf[j] = (float)(c[j] / (double)ids.size());
    float cn = dotf(cf.data(), cf.data(), d);
    int best = ids[0];
    float bs = -std::numeric_limits<float>::infinity();
    for (int id : ids) {
        float s = sp.score(cf.data(), sp.row(id), cn, sp.sim == 2 ? sp.norm2[id] : 0.f);
        if (s > bs) {
            bs = s;
            best = id;
        }
    }
    return best;
}

// Build over the node subset `ids` (ascending ordinals). out rows are in `ids` order, stride R.
void build_subset(const Space& sp, const std::vector<int32_t>& ids, int R, int L, float alpha, float overflow,
                  int max_batch, int threads, int32_t* out_adj, int32_t* out_entry, bool refine = true) {
    int m = (int)ids.size();
    int cap = std::max(R, (int)std::ceil(R * overflow));
    if (m == 0) {
        *out_entry = -1;
        return;
    }
    // adjacency rows hold global ordinals; the working graph is indexed by global ordinal too
    Graph g((int)(*std::max_element(ids.begin(), ids.end())) + 1, cap);
    int entry = ids[0];
    int inserted = 1;
    std::vector<uint32_t> stamp_all;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = threads > 0 ? threads : omp_get_max_threads();
#endif
    std::vector<std::vector<uint32_t>> stamps(nthreads, std::vector<uint32_t>(g.n, 0));
    std::vector<uint32_t> stamp_ctr(nthreads, 0);
    while (inserted < m) {
        int B = std::max(1, std::min(max_batch, inserted / 8));
        B = std::min(B, m - inserted);
        std::vector<std::vector<Scored>> cands(B);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
        for (int b = 0; b < B; b++) {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            uint32_t st = ++stamp_ctr[tid];
            beam_search(sp, g, entry, ids[inserted + b], L, cands[b], stamps[tid], st);
        }
        std::vector<int32_t> sel;
        for (int b = 0; b < B; b++) {
            int u = ids[inserted + b];
            robust_prune(sp, u, cands[b], alpha, R, sel);
            set_row(g, u, sel);
            for (int s : sel) add_backlink(sp, g, s, u, alpha, R);
        }
        inserted += B;
    }
    // second pass over the finished graph (Vamana's refinement / jvector cleanup's improveConnections):
    // re-search every node and re-prune its neighbourhood with the union of old and new candidates
    for (int start = 0; start < m && refine; start += max_batch) {
        int B = std::min(max_batch, m - start);
        std::vector<std::vector<Scored>> cands(B);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
        for (int b = 0; b < B; b++) {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            uint32_t st = ++stamp_ctr[tid];
            beam_search(sp, g, entry, ids[start + b], L, cands[b], stamps[tid], st);
        }
        std::vector<int32_t> sel;
        for (int b = 0; b < B; b++) {
            int u = ids[start + b];
            const int32_t* r = g.row(u);
            for (int k = 0; k < g.deg[u]; k++) cands[b].push_back({sp.score_nodes(u, r[k]), r[k]});
            robust_prune(sp, u, cands[b], alpha, R, sel);
            set_row(g, u, sel);
            for (int s : sel) add_backlink(sp, g, s, u, alpha, R);
        }
    }
    // cleanup: enforce degree <= R
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
    for (int i = 0; i < m; i++) {
        int u = ids[i];
        if (g.deg[u] > R) {
            std::vector<Scored> c;
            const int32_t* r = g.row(u);
            for (int k = 0; k < g.deg[u]; k++) c.push_back({sp.score_nodes(u, r[k]), r[k]});
            std::vector<int32_t> sel;
            robust_prune(sp, u, c, alpha, R, sel);
            set_row(g, u, sel);
        }
    }
    for (int i = 0; i < m; i++) {
        const int32_t* r = g.row(ids[i]);
        for (int k = 0; k < R; k++) out_adj[(size_t)i * R + k] = k < g.deg[ids[i]] ? r[k] : -1;
    }
    *out_entry = approx_medoid(sp, ids);
}

Space make_space(const float* vectors, int n, int d, int sim) {
    Space sp;
    sp.v = vectors;
    sp.d = d;
    sp.sim = sim;
    if (sim == 2) {
        sp.norm2.resize(n);
        for (int i = 0; i < n; i++) sp.norm2[i] = dotf(sp.row(i), sp.row(i), d);
    }
    return sp;
}

uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace

extern "C" int jvb_build_graph_cpu(const float* vectors, int32_t n, int32_t d, int32_t sim, int32_t R,
                                   int32_t L, float alpha, float overflow, int32_t max_batch,
                                   int32_t threads, int32_t* out_adj, int32_t* out_entry) {
    if (!vectors || n < 0 || d <= 0 || R <= 0 || L <= 0 || !out_adj || !out_entry) return -1;
    if (n == 0) {
        *out_entry = -1;
        return 0;
    }
    Space sp = make_space(vectors, n, d, sim);
    std::vector<int32_t> ids(n);
    std::iota(ids.begin(), ids.end(), 0);
    build_subset(sp, ids, R, L, alpha, overflow, max_batch > 0 ? max_batch : 256, threads, out_adj, out_entry);
    return 0;
}

extern "C" int jvb_build_upper_layers_cpu(const float* vectors, int32_t n, int32_t d, int32_t sim,
                                          int32_t R, int32_t L, float alpha, int32_t num_layers,
                                          uint64_t seed, int32_t* counts, int32_t** nodes,
                                          int32_t** adj, int32_t* out_entry) {
    if (!vectors || n <= 0 || num_layers <= 0 || !counts) return -1;
    // deterministic geometric levels: P(level >= l) = (1/R)^l, at least one node per layer
    std::vector<int> level(n, 0);
    uint64_t st = seed;
    double ml = 1.0 / std::log((double)std::max(2, R));
    for (int i = 0; i < n; i++) {
        double u = ((splitmix64(st) >> 11) + 1) * (1.0 / 9007199254740993.0);
        level[i] = std::min(num_layers, (int)(-std::log(u) * ml));
    }
    for (int l = 1; l <= num_layers; l++) {
        bool any = false;
        for (int i = 0; i < n; i++) any |= level[i] >= l;
        if (!any) level[(int)(splitmix64(st) % (uint64_t)n)] = num_layers;
    }
    Space sp = make_space(vectors, n, d, sim);
    int top_entry = -1;
    for (int l = 1; l <= num_layers; l++) {
        std::vector<int32_t> ids;
        for (int i = 0; i < n; i++)
            if (level[i] >= l) ids.push_back(i);
        counts[l - 1] = (int32_t)ids.size();
        if (nodes && adj) {
            std::memcpy(nodes[l - 1], ids.data(), ids.size() * sizeof(int32_t));
            int32_t e;
            build_subset(sp, ids, R, L, alpha, 1.2f, 64, 1, adj[l - 1], &e);
            if (l == num_layers) top_entry = e;
        }
    }
    if (out_entry) *out_entry = top_entry;
    return 0;
}

extern "C" int jvb_pq_train_cpu(const float* vectors, int32_t n, int32_t d, int32_t M, int32_t K,
                                int32_t center, int32_t iters, int32_t max_train, uint64_t seed,
                                int32_t threads, float* out_codebooks, float* out_centroid) {
    if (!vectors || n <= 0 || d <= 0 || M <= 0 || M > d || K <= 0 || K > 256 || !out_codebooks) return -1;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = threads > 0 ? threads : omp_get_max_threads();
#endif
    int nt = std::min(n, max_train > 0 ? max_train : n);
    std::vector<int32_t> rows(nt);
    for (int i = 0; i < nt; i++) rows[i] = (int32_t)(((int64_t)i * n) / nt);
    std::vector<float> cen(d, 0.f);
    if (center) {
        std::vector<double> acc(d, 0.0);
        for (int i = 0; i < n; i++) {
            const float* r = vectors + (size_t)i * d;
            for (int j = 0; j < d; j++) acc[j] += r[j];
        }
        for (int j = 0; j < d; j++) cen[j] = (float)(acc[j] / n);
        if (out_centroid) std::memcpy(out_centroid, cen.data(), sizeof(float) * d);
    }
    std::vector<int> sizes(M), offs(M);
    int off = 0;
    for (int m = 0; m < M; m++) {
        sizes[m] = d / M + (m < d % M ? 1 : 0);
        offs[m] = off;
        off += sizes[m];
    }
    std::vector<size_t> cb_off(M);
    size_t co = 0;
    for (int m = 0; m < M; m++) {
        cb_off[m] = co;
        co += (size_t)K * sizes[m];
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int m = 0; m < M; m++) {
        int s = sizes[m];
        std::vector<float> pts((size_t)nt * s);
        for (int i = 0; i < nt; i++) {
            const float* r = vectors + (size_t)rows[i] * d + offs[m];
            for (int j = 0; j < s; j++) pts[(size_t)i * s + j] = r[j] - cen[offs[m] + j];
        }
        float* cb = out_codebooks + cb_off[m];
        uint64_t st = seed * 0x9E3779B97F4A7C15ull + (uint64_t)m * 1315423911ull + 7;
        // k-means++ seeding
        std::vector<float> mind(nt, std::numeric_limits<float>::max());
        int first = (int)(splitmix64(st) % (uint64_t)nt);
        std::memcpy(cb, &pts[(size_t)first * s], sizeof(float) * s);
        int kk = std::min(K, nt);
        for (int c = 1; c < K; c++) {
            if (c >= kk) {  // fewer points than clusters: duplicate
                std::memcpy(cb + (size_t)c * s, cb + (size_t)(c % kk) * s, sizeof(float) * s);
                continue;
            }
            double tot = 0;
            for (int i = 0; i < nt; i++) {
                float dd = l2f(&pts[(size_t)i * s], cb + (size_t)(c - 1) * s, s);
                if (dd < mind[i]) mind[i] = dd;
                tot += mind[i];
            }
            double r = ((splitmix64(st) >> 11) * (1.0 / 9007199254740992.0)) * tot;
            int pick = nt - 1;
            double run = 0;
            for (int i = 0; i < nt; i++) {
                run += mind[i];
                if (run >= r) {
                    pick = i;
                    break;
                }
            }
            std::memcpy(cb + (size_t)c * s, &pts[(size_t)pick * s], sizeof(float) * s);
        }
        // Lloyd
        std::vector<int> assign(nt, 0);
        std::vector<double> sum((size_t)K * s);
        std::vector<int> cnt(K);
        for (int it = 0; it < iters; it++) {
            for (int i = 0; i < nt; i++) {
                float bd = std::numeric_limits<float>::max();
                int bc = 0;
                for (int c = 0; c < K; c++) {
                    float dd = l2f(&pts[(size_t)i * s], cb + (size_t)c * s, s);
                    if (dd < bd) {
                        bd = dd;
                        bc = c;
                    }
                }
                assign[i] = bc;
            }
            std::fill(sum.begin(), sum.end(), 0.0);
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int i = 0; i < nt; i++) {
                cnt[assign[i]]++;
                for (int j = 0; j < s; j++) sum[(size_t)assign[i] * s + j] += pts[(size_t)i * s + j];
            }
            for (int c = 0; c < K; c++)
                if (cnt[c] > 0)
                    for (int j = 0; j < s; j++) cb[(size_t)c * s + j] = (float)(sum[(size_t)c * s + j] / cnt[c]);
        }
    }
    return 0;
}

extern "C" int jvb_pq_encode_cpu(const float* vectors, int32_t n, int32_t d, int32_t M, int32_t K,
                                 const float* codebooks, const float* centroid, int32_t threads,
                                 uint8_t* out_codes) {
    if (!vectors || !codebooks || !out_codes || n < 0 || M <= 0 || K <= 0 || K > 256) return -1;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = threads > 0 ? threads : omp_get_max_threads();
#endif
    std::vector<int> sizes(M), offs(M);
    std::vector<size_t> cb_off(M);
    int off = 0;
    size_t co = 0;
    for (int m = 0; m < M; m++) {
        sizes[m] = d / M + (m < d % M ? 1 : 0);
        offs[m] = off;
        off += sizes[m];
        cb_off[m] = co;
        co += (size_t)K * sizes[m];
    }
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int i = 0; i < n; i++) {
        std::vector<float> tmp(d);
        const float* r = vectors + (size_t)i * d;
        for (int j = 0; j < d; j++) tmp[j] = centroid ? r[j] - centroid[j] : r[j];
        for (int m = 0; m < M; m++) {
            const float* cb = codebooks + cb_off[m];
            float bd = std::numeric_limits<float>::max();
            int bc = 0;
            for (int c = 0; c < K; c++) {
                // the canonical PQ distance: the sequential fmaf chain over the subspace's dimensions that the search
                // kernels use for a look-up table entry (so encode(x) == argmin of x's own LUT row), ties -> lowest c
                const float* xs = &tmp[offs[m]];
                const float* cv = cb + (size_t)c * sizes[m];
                float dd = 0.0f;
                for (int j = 0; j < sizes[m]; j++) {
                    const float df = xs[j] - cv[j];
                    dd = std::fmaf(df, df, dd);
                }
                if (dd < bd) {
                    bd = dd;
                    bc = c;
                }
            }
            out_codes[(size_t)i * M + m] = (uint8_t)bc;
        }
    }
    return 0;
}
