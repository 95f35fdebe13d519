// Step chains: several captured steps as ONE executable hipGraph (host code only, no kernel here).
//
// A replayed step is ~40 kernel nodes that follow each other without a gap, but between two hipGraphLaunch calls the queue idles
// for ~20 us (end-of-graph signal, the next launch's first dispatch): 4 % of a 0.5 ms step.  The main.py:157 loop is launch-bound
// at that granularity, so the loop body is chained: the nodes of the captured step graphs are re-added, in their own order, to a
// new graph — segment after segment, the roots of a segment depending on the leaves of the one before — and the whole sequence
// `repeat` times.  Nothing is re-captured: the source graphs keep their buffers (they must outlive the chain) and stay
// launchable on their own, a chain launch is exactly the same kernels with the same arguments as the launches it replaces.
#include "common.h"
#include <vector>
#include <unordered_map>

namespace {
struct Chain {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int32_t nodes = 0;
};

// nodes of g in a topological order (Kahn over hipGraphNodeGetDependencies; a single-stream capture is already a chain)
int topo_nodes(hipGraph_t g, std::vector<hipGraphNode_t>& order,
               std::unordered_map<hipGraphNode_t, std::vector<hipGraphNode_t>>& deps) {
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(g, nullptr, &n);
    if (e != hipSuccess) return (int)e;
    std::vector<hipGraphNode_t> all(n);
    if (n) { e = hipGraphGetNodes(g, all.data(), &n); if (e != hipSuccess) return (int)e; }
    std::unordered_map<hipGraphNode_t, int> indeg;
    std::unordered_map<hipGraphNode_t, std::vector<hipGraphNode_t>> outs;
    for (hipGraphNode_t v : all) {
        size_t nd = 0;
        e = hipGraphNodeGetDependencies(v, nullptr, &nd);
        if (e != hipSuccess) return (int)e;
        std::vector<hipGraphNode_t> d(nd);
        if (nd) { e = hipGraphNodeGetDependencies(v, d.data(), &nd); if (e != hipSuccess) return (int)e; }
        indeg[v] = (int)nd;
        for (hipGraphNode_t u : d) outs[u].push_back(v);
        deps[v] = std::move(d);
    }
    order.clear();
    std::vector<hipGraphNode_t> ready;
    for (hipGraphNode_t v : all) if (indeg[v] == 0) ready.push_back(v);      // (insertion order kept among the ready ones)
    size_t head = 0;
    while (head < ready.size()) {
        hipGraphNode_t v = ready[head++];
        order.push_back(v);
        for (hipGraphNode_t w : outs[v]) if (--indeg[w] == 0) ready.push_back(w);
    }
    return order.size() == all.size() ? 0 : GRAPES_EINVAL;
}
}  // namespace

extern "C" int grapes_graph_chain_create(void* const* graphs, int32_t n, int32_t repeat, void** chain_out, int32_t* nodes_out) {
    if (!graphs || n <= 0 || repeat <= 0 || !chain_out) return GRAPES_EINVAL;
    for (int i = 0; i < n; i++) if (!graphs[i]) return GRAPES_EINVAL;
    Chain* c = new Chain();
    hipError_t e = hipGraphCreate(&c->graph, 0);
    if (e != hipSuccess) { delete c; return (int)e; }
    int rc = 0;
    std::vector<hipGraphNode_t> prev_leaves;
    // per source graph: its order and dependencies, read once
    std::vector<std::vector<hipGraphNode_t>> orders(n);
    std::vector<std::unordered_map<hipGraphNode_t, std::vector<hipGraphNode_t>>> depss(n);
    for (int i = 0; i < n && !rc; i++) rc = topo_nodes((hipGraph_t)graphs[i], orders[i], depss[i]);
    for (int r = 0; r < repeat && !rc; r++) {
        for (int i = 0; i < n && !rc; i++) {
            std::unordered_map<hipGraphNode_t, hipGraphNode_t> image;       // source node -> its copy in this segment
            std::unordered_map<hipGraphNode_t, bool> has_child;
            for (hipGraphNode_t v : orders[i]) {
                std::vector<hipGraphNode_t> d;
                const auto& dv = depss[i][v];
                if (dv.empty()) d = prev_leaves;                             // a root waits for the whole previous segment
                else for (hipGraphNode_t u : dv) { d.push_back(image[u]); has_child[u] = true; }
                hipGraphNodeType ty;
                e = hipGraphNodeGetType(v, &ty);
                if (e != hipSuccess) { rc = (int)e; break; }
                hipGraphNode_t w = nullptr;
                if (ty == hipGraphNodeTypeKernel) {
                    hipKernelNodeParams p;
                    e = hipGraphKernelNodeGetParams(v, &p);
                    if (e == hipSuccess) e = hipGraphAddKernelNode(&w, c->graph, d.data(), d.size(), &p);
                } else if (ty == hipGraphNodeTypeEmpty) {
                    e = hipGraphAddEmptyNode(&w, c->graph, d.data(), d.size());
                } else if (ty == hipGraphNodeTypeMemset) {
                    hipMemsetParams p;
                    e = hipGraphMemsetNodeGetParams(v, &p);
                    if (e == hipSuccess) e = hipGraphAddMemsetNode(&w, c->graph, d.data(), d.size(), &p);
                } else if (ty == hipGraphNodeTypeMemcpy) {                   // (a contiguous device copy captured as a copy node: ADVICE r04)
                    hipMemcpy3DParms p;
                    e = hipGraphMemcpyNodeGetParams(v, &p);
                    if (e == hipSuccess) e = hipGraphAddMemcpyNode(&w, c->graph, d.data(), d.size(), &p);
                } else {
                    rc = GRAPES_EINVAL;                                      // (host functions, events, child graphs: not in a captured step;
                    break;                                                   //  the caller falls back to one launch per step)
                }
                if (e != hipSuccess) { rc = (int)e; break; }
                image[v] = w;
                c->nodes++;
            }
            if (rc) break;
            prev_leaves.clear();
            for (hipGraphNode_t v : orders[i]) if (!has_child.count(v)) prev_leaves.push_back(image[v]);
        }
    }
    if (!rc) {
        e = hipGraphInstantiate(&c->exec, c->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) rc = (int)e;
    }
    if (rc) {
        if (c->graph) (void)hipGraphDestroy(c->graph);
        delete c;
        return rc;
    }
    *chain_out = c;
    if (nodes_out) *nodes_out = c->nodes;
    return 0;
}

extern "C" int grapes_graph_chain_launch(void* chain, grapes_stream_t stream) {
    if (!chain) return GRAPES_EINVAL;
    return (int)hipGraphLaunch(((Chain*)chain)->exec, (hipStream_t)stream);
}

extern "C" int grapes_graph_chain_destroy(void* chain) {
    if (!chain) return GRAPES_EINVAL;
    Chain* c = (Chain*)chain;
    hipError_t e1 = c->exec ? hipGraphExecDestroy(c->exec) : hipSuccess;
    hipError_t e2 = c->graph ? hipGraphDestroy(c->graph) : hipSuccess;
    delete c;
    return (int)(e1 != hipSuccess ? e1 : e2);
}
