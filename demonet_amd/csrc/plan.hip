// C-ABI implementation: plan construction, workspace layout, launch sequence, hipGraph caching, profiling hook.
// Host-side only (no kernels here). See include/demonet_hip.h for the contract.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "common.h"

// ---------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static thread_local char g_kernel[96] = "";
void dn_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
const char* dn_last_kernel() { return g_kernel; }

void dn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dn_last_error(void) { return g_err; }
extern "C" int dn_abi_version(void) { return DN_ABI_VERSION; }

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int dn_knob(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

hipError_t dn_allow_big_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({dev, kernel});
    return e;
}

// XCD grouping policy (common.h). Read per call, never latched: DN_XCD=0 switches it off (A/B runs, tests of the plain mapping).
int xcd_images_per_group(int n) {
    const char* v = getenv("DN_XCD");
    if (v && atoi(v) == 0) return 0;
    return n >= 8 ? (n + 7) / 8 : 0;
}

struct Layout {
    int n = 0;
    std::vector<size_t> toff;       // per tensor byte offset in workspace (SIZE_MAX: not materialised)
    std::vector<size_t> tbytes;
    size_t resized_off = 0, logits_off = 0, reg_off = 0, scale_off = 0, post_off = 0, post_bytes = 0, total = 0;
    // liveness reuse: every sub-batch chain owns one arena of `arena` bytes (chains run concurrently at different points of the
    // op list, so blocks that different tensors share must not be shared between chains); toff / tbytes then describe chain 0
    // (chain_n images), chain k adds k * arena. arena == 0: one contiguous [n]-image block per tensor.
    size_t arena = 0;
    int chain_n = 0;
    size_t secnt_off = 0;       // [n_se_in_dw][n] unsigned: last-workgroup counters of the squeeze-excitation tails
};

struct GraphKey {
    const void* img; int n, h, w; void* ws; void* boxes; void* scores; void* labels; void* counts; int heads_only; void* packed;
    int chain;          // sub-batch chain index (one single-chain graph per sub-batch), -1: the whole forward in one graph
    bool operator<(const GraphKey& o) const {
        return std::tie(img, n, h, w, ws, boxes, scores, labels, counts, heads_only, packed, chain) <
               std::tie(o.img, o.n, o.h, o.w, o.ws, o.boxes, o.scores, o.labels, o.counts, o.heads_only, o.packed, o.chain);
    }
};

struct dn_plan {
    dn_model_desc d;
    std::vector<dn_tensor_desc> tensors;
    std::vector<dn_op_desc> ops;
    std::vector<int> level_off;         // anchor offset per level
    std::vector<int> pool_blocks;       // per DN_T_POOL tensor: partial-sum rows per image (= dw workgroups per image)
    unsigned char* weights_dev = nullptr;
    size_t zeros_off = 0;
    size_t weight_bytes = 0;
    float* anchors_dev = nullptr;
    std::map<int, Layout> layouts;
    // workspaces whose last forward was dn_forward (not dn_forward_heads): their head arrays may be incomplete -- from DN_HEAD_SOFTMAX_MINN images
    // per chain up the fused head launch writes scores / boxes instead of the large levels' logits -- so dn_head_outputs refuses them
    std::map<const void*, bool> heads_partial;
    bool graph_mode = true;
    std::map<GraphKey, hipGraphExec_t> graphs;
    hipStream_t capture_stream = nullptr;   // capture never happens on the caller's stream (may be the null stream)
    std::vector<int> op_stream;             // 0 backbone, 1 class-head chain, 2 box-head chain (head ops and the depthwise ops feeding them)
    std::vector<int> op_wait_level;         // head-chain op reading a feature map: its level, else -1
    int split = 2;                          // sub-batch branches per forward (see batch_split)
    bool ws_reuse = true;                   // DN_WS_REUSE=0: one private block per tensor (every intermediate stays readable after the forward)
    int chain_graphs = -1;                  // DN_CHAIN_GRAPHS: 1 = one single-chain graph per sub-batch on its own stream, 0 = branches of ONE graph, -1 = by batch size
    bool xcd = true;                        // XCD grouping of every kernel's workgroups by image (common.h; DN_XCD, read in dn_create)
    hipStream_t branch_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_branch[3] = {nullptr, nullptr, nullptr};
    std::map<std::pair<int, int>, Layout> sub_layouts;
    int chains_override = 0;                // dn_set_chains: > 0 = that many sub-batch chains per forward whatever the batch size
    float* packed_out = nullptr;
    bool input_u8 = false;                  // the current call's images are [n][h][w][3] uint8 (dn_forward_u8)
    // head ops (dw -> 1x1 / dense 3x3 per level, both heads) run as grouped launches once the backbone is done
    int head_first = -1;                    // index of the first head-chain op (all later ops are head-chain ops), -1: off
    std::vector<int> head_dw, head_cls, head_reg;
    // run of tiny backbone layers [tail_first, tail_end) executed by one per-image workgroup (tail.hip); -1: none
    int tail_first = -1, tail_end = -1;
    std::vector<int> se_in_dw;              // per op: DW op -> index of the SE op whose FCs run in its tail (depthwise.hip dw_se_tail), SE op -> -2, else -1
    int n_se_in_dw = 0;                     // such pairs; slot q of the counter block belongs to the q-th
    std::vector<int> se_slot;               // per op (DW op of a pair): q
    int post_ticket_slot = -1;              // slot of the counter block lent to launch_postprocess (PostArgs::tickets); needs the stem launch that zeroes the block
    std::vector<char> stem_split_ok;        // per op: STEM op that may run on the split-fp16 matrix kernel (dn_create checks the host copy of the weights)
    std::vector<int> stem_scale_log2;       // per op: its power-of-two weight scale
    std::vector<int> se_fold;               // per op: PW op -> index of the SE op whose FCs run in its prologue (pointwise.hip SEF), SE op -> -2, else -1
    std::vector<char> tail_materialise;     // per op of the run: its output is read outside the run (pyramid feature) -> also to HBM            // optional extra output of the merge kernel (dn_set_packed_output)
    // inverted-residual stages that run as one launch (expdw.hip): at the first op of a group, fused_len = number of ops and
    // fused_kind bit0 = has expand (1x1), bit1 = has project (1x1 [+ residual]); the depthwise op is always part of it
    std::vector<int> fused_len, fused_kind;
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> events;
    std::vector<double> prof_ms;
    std::vector<std::string> prof_kernel;   // label of the launch each op took part in
    std::vector<int> prof_owner;            // op index whose event segment holds that launch's time
    int prof_runs = 0;
    // packed_out / input_u8 / the graph cache are per-plan mutable state set around a call: a plan serves ONE host thread at a time
    // (several forwards in flight are several STREAMS fed by one thread, pipeline.py). A second thread entering gets DN_E_INVALID.
    std::atomic<int> in_call{0};
};

// Graph executables may still be replaying on other streams (the slots of a ForwardPipeline) when the cache is cleared: drain the
// device first. Clearing is rare (cache bound reached, dn_set_chains, dn_destroy), so the host wait does not matter.
static void drop_graphs(dn_plan* p) {
    if (p->graphs.empty()) return;
    (void)hipDeviceSynchronize();
    for (auto& kv : p->graphs) (void)hipGraphExecDestroy(kv.second);
    p->graphs.clear();
}

// The launch chain is latency-bound (~70 dependent launches, most of them far from filling 256 CUs), so a forward of n images
// is issued as `split` independent sub-batch chains that the hipGraph runs as parallel branches (measured +6 % at n = 64 with
// two branches; more branches lose again, and below 32 images there is nothing to gain). Every workspace tensor is
// image-major, so a sub-batch simply addresses rows [n0, n0 + ns) of the same layout.
static int batch_split(const dn_plan* p, int n) {
    if (p->chains_override > 0) return std::min(p->chains_override, n);
    if (p->split <= 1 || n < 32) return 1;
    return p->split;
}
static int sub_count(int n, int S, int k) { const int base = n / S, rem = n % S; return base + (k < rem ? 1 : 0); }

static const Layout& get_layout(dn_plan* p, int n) {
    auto it = p->layouts.find(n);
    if (it != p->layouts.end()) return it->second;
    Layout L;
    L.n = n;
    size_t off = 0;
    const size_t T = p->tensors.size();
    L.toff.assign(T, (size_t)-1);
    L.tbytes.assign(T, 0);
    const int S_chains = batch_split(p, n);
    const int nn = p->ws_reuse ? sub_count(n, S_chains, 0) : n;      // images per block: one chain's share with reuse, else all
    L.chain_n = nn;
    for (size_t i = 0; i < T; ++i) {
        const dn_tensor_desc& t = p->tensors[i];
        size_t b = 0;
        if (t.kind == DN_T_ACT) b = (size_t)nn * t.h * t.w * t.c * 2;
        else if (t.kind == DN_T_VEC) b = (size_t)nn * t.c * 4;
        else if (t.kind == DN_T_POOL) b = (size_t)nn * p->pool_blocks[i] * t.c * 4;
        else continue;      // image: caller's buffer (or the resized copy below)
        L.tbytes[i] = b;
    }
    if (!p->ws_reuse) {
        for (size_t i = 0; i < T; ++i)
            if (L.tbytes[i]) { L.toff[i] = off; off += align256(L.tbytes[i]); }
    } else {
        // Liveness reuse: a tensor occupies its block from the launch that writes it to the last launch that reads it (launch
        // time = op index, with the ops of one fused / grouped launch sharing a time step); blocks are placed first-fit at the
        // lowest offset that is free over the whole interval. Tensors that only exist inside a fused launch get no block. The
        // pyramid features live until the head launches. Every block still holds all n images, so the sub-batch chains keep
        // addressing disjoint rows of the same blocks. 2.6 GB -> ~0.4 GB at batch 64: producer -> consumer pairs of the large maps
        // now rewrite lines that are already resident in the Infinity Cache instead of streaming through fresh memory.
        const int NO = (int)p->ops.size();
        std::vector<int> when(NO);                 // launch time step of every op
        std::vector<char> inner(T, 0);             // tensor produced AND consumed inside one fused launch (never materialised)
        {
            int tstep = 0;
            for (int i = 0; i < NO;) {
                int len = 1;
                if (p->head_first >= 0 && i >= p->head_first) {
                    // head launches: the depthwise group, then the 1x1 / dense group(s)
                    for (int q = i; q < NO; ++q) when[q] = tstep + (p->ops[q].type == DN_OP_DW ? 0 : 1);
                    break;
                }
                if (i == p->tail_first) len = p->tail_end - p->tail_first;
                else if (p->fused_len[i] > 0) len = p->fused_len[i];
                for (int q = 0; q < len; ++q) when[i + q] = tstep;
                if (len > 1) {
                    for (int q = 0; q + 1 < len; ++q) {
                        const int tid = p->ops[i + q].out;
                        bool keep = false;
                        if (i == p->tail_first) keep = p->tail_materialise[q] != 0;
                        for (int l = 0; l < p->d.n_levels; ++l) keep |= p->d.level_tensor[l] == tid;
                        for (int u = 0; u < NO; ++u) {
                            if (u >= i && u < i + len) continue;
                            const dn_op_desc& o = p->ops[u];
                            keep |= (o.in == tid || o.residual == tid || o.se == tid);
                        }
                        if (!keep) inner[tid] = 1;
                    }
                }
                ++tstep;
                i += len;
            }
        }
        std::vector<int> born(T, -1), dies(T, -1);
        for (int i = 0; i < NO; ++i) {
            const dn_op_desc& o = p->ops[i];
            auto use = [&](int tid) { if (tid >= 0) dies[tid] = std::max(dies[tid], when[i]); };
            use(o.in); use(o.residual); use(o.se);
            if (p->se_fold[i] >= 0) use(p->ops[p->se_fold[i]].in);       // folded squeeze-excitation: the projection reads the pooled partial sums
            if (born[o.out] < 0) born[o.out] = when[i];
            dies[o.out] = std::max(dies[o.out], when[i]);
            if (o.pool >= 0) { if (born[o.pool] < 0) born[o.pool] = when[i]; dies[o.pool] = std::max(dies[o.pool], when[i]); }
        }
        for (int i = 0; i < NO; ++i)
            if (p->se_in_dw[i] >= 0) {      // the scale vector is written by the depthwise launch itself (dw_se_tail)
                const int t = p->ops[p->se_in_dw[i]].out;
                if (born[t] >= 0) born[t] = std::min(born[t], when[i]);
            }
        const int t_end = NO + 2;
        for (int l = 0; l < p->d.n_levels; ++l) dies[p->d.level_tensor[l]] = t_end;      // read back by tests / callers after the forward
        struct Blk { size_t off, bytes; int born, dies; };
        std::vector<Blk> placed;
        std::vector<size_t> order;
        for (size_t i = 0; i < T; ++i)
            if (L.tbytes[i] && !inner[i] && born[i] >= 0) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return born[a] < born[b]; });
        for (size_t i : order) {
            const size_t need = align256(L.tbytes[i]);
            std::vector<std::pair<size_t, size_t>> busy;       // blocks alive at some point of [born, dies]
            for (const Blk& b : placed)
                if (!(b.dies < born[i] || b.born > dies[i])) busy.push_back({b.off, b.off + b.bytes});
            std::sort(busy.begin(), busy.end());
            size_t at = 0;
            for (const auto& iv : busy) {
                if (iv.first >= at + need) break;
                at = std::max(at, iv.second);
            }
            L.toff[i] = at;
            placed.push_back({at, need, born[i], dies[i]});
            off = std::max(off, at + need);
        }
        L.arena = align256(off);
        off = L.arena * (size_t)S_chains;
    }
    L.resized_off = off;
    off += align256((size_t)n * 3 * p->d.image_h * p->d.image_w * 4);
    L.logits_off = off;
    off += align256((size_t)n * p->d.num_anchors * p->d.num_classes * 4);
    L.reg_off = off;
    off += align256((size_t)n * p->d.num_anchors * 4 * 4);
    L.scale_off = off;
    off += align256((size_t)n * 2 * 4);
    L.secnt_off = off;
    off += align256((size_t)n * p->n_se_in_dw * 4 + 4);
    L.post_off = off;
    {
        const int S = batch_split(p, n);          // one private post-process scratch slice per sub-batch branch
        L.post_bytes = (size_t)S * align256(postprocess_ws_bytes(sub_count(n, S, 0), p->d.num_anchors, p->d.num_classes,
                                                                 p->d.topk_candidates, p->d.detections_per_img));
    }
    off += align256(L.post_bytes);
    L.total = off;
    return p->layouts.emplace(n, std::move(L)).first->second;
}

// ---------------------------------------------------------------------------------------------------------
extern "C" int dn_create(const dn_model_desc* desc, const void* weights, size_t weight_bytes, dn_plan** out) {
    DN_REQUIRE(desc && weights && out, "dn_create: null argument");
    DN_REQUIRE(desc->abi_version == DN_ABI_VERSION, "dn_create: ABI version %d != library %d", desc->abi_version, DN_ABI_VERSION);
    DN_REQUIRE(desc->n_tensors > 0 && desc->n_ops > 0 && desc->tensors && desc->ops, "dn_create: empty graph");
    DN_REQUIRE(desc->n_levels >= 1 && desc->n_levels <= 8, "dn_create: n_levels=%d outside [1,8]", desc->n_levels);
    DN_REQUIRE(desc->num_classes >= 2, "dn_create: num_classes must include background (>= 2)");
    DN_REQUIRE(desc->anchors && desc->num_anchors > 0, "dn_create: anchors missing");
    DN_REQUIRE(desc->score_thresh >= 0.f, "dn_create: score_thresh must be >= 0");
    dn_plan* p = new dn_plan();
    p->d = *desc;
    p->tensors.assign(desc->tensors, desc->tensors + desc->n_tensors);
    p->ops.assign(desc->ops, desc->ops + desc->n_ops);
    p->d.tensors = p->tensors.data();
    p->d.ops = p->ops.data();
    // validate ops
    auto fail = [&](int rc) { delete p; return rc; };
    for (int i = 0; i < desc->n_ops; ++i) {
        const dn_op_desc& o = p->ops[i];
        auto ok_t = [&](int t) { return t >= 0 && t < desc->n_tensors; };
        if (!ok_t(o.in) || !ok_t(o.out)) { dn_set_error("dn_create: op %d has bad tensor ids", i); return fail(DN_E_INVALID); }
        if (o.w_off >= 0 && (size_t)o.w_off >= weight_bytes) { dn_set_error("dn_create: op %d weight offset out of range", i); return fail(DN_E_INVALID); }
        if (o.type < DN_OP_STEM || o.type > DN_OP_L2NORM) { dn_set_error("dn_create: op %d unknown type %d", i, o.type); return fail(DN_E_INVALID); }
        if (o.head && (o.level < 0 || o.level >= desc->n_levels)) { dn_set_error("dn_create: head op %d bad level", i); return fail(DN_E_INVALID); }
    }
    // ---- expand (1x1) -> depthwise pairs whose expanded tensor has no other reader: one launch, the tensor never leaves LDS
    p->fused_len.assign(desc->n_ops, 0);
    p->fused_kind.assign(desc->n_ops, 0);
    {
        const bool enabled = getenv("DN_EXPDW") ? atoi(getenv("DN_EXPDW")) != 0 : true;
        const int max_hw = getenv("DN_EXPDW_MAXHW") ? atoi(getenv("DN_EXPDW_MAXHW")) : (1 << 30);
        // measured per layer (bench.py --per-op, tools/probe_expdw.py): the fused kernel currently wins on the large maps only
        const int min_hw = getenv("DN_EXPDW_MINHW") ? atoi(getenv("DN_EXPDW_MINHW")) : 6400;
        std::vector<int> uses(desc->n_tensors, 0);
        for (int i = 0; i < desc->n_ops; ++i) {
            const dn_op_desc& o = p->ops[i];
            uses[o.in]++;
            if (o.residual >= 0) uses[o.residual]++;
            if (o.se >= 0) uses[o.se]++;
        }
        for (int l = 0; l < desc->n_levels; ++l) uses[desc->level_tensor[l]] += 100;    // features must be materialised
        // project stage (whole inverted-residual block in one launch) down to the 20 x 20 maps: one workgroup owns all expanded channels
        // of its pixel tile (chunk loop), the expand's A fragments fit the register variants up to cin = 88, and the (pixel tile x
        // channel tile) units of the projection must fit the 8 waves (5 x 10 pixel tiles there). Measured: the 160^2 / 80^2 blocks
        // -20 % each (round 1); the four blocks without squeeze-excitation on the 20 x 20 maps (40->240->80 stride 2, 80->200->80,
        // 2 x 80->184->80: three launches of 5 - 11 us each become one of 11 - 17 us) batch 64 1.10 -> 1.055 ms, batch 32 0.765 -> 0.74
        const int proj_min_hw = getenv("DN_EXPDW_PROJ_MINHW") ? atoi(getenv("DN_EXPDW_PROJ_MINHW")) : 300;      // (19 x 19 maps of the 300-pixel models included)
        auto plain_pw = [&](const dn_op_desc& o) { return o.type == DN_OP_PW && !o.head && o.se < 0; };
        auto dw_ok = [&](const dn_op_desc& o) { return o.type == DN_OP_DW && !o.head && o.dil == 1; };
        auto proj_ok = [&](const dn_op_desc& pj, const dn_op_desc& d, int block_in) {
            const dn_tensor_desc& to = p->tensors[d.out];
            return plain_pw(pj) && pj.in == d.out && uses[d.out] == 1 && d.pool < 0 && pj.act == DN_ACT_NONE && d.cin <= 640 &&
                   to.h * to.w >= proj_min_hw && expdw_project_supported(d.cin, pj.cout, to.h, to.w, d.stride) &&
                   (pj.residual < 0 || (pj.residual == block_in && d.stride == 1 && pj.cout == p->tensors[block_in].c));
        };
        for (int i = 0; enabled && i + 1 < desc->n_ops; ++i) {
            const dn_op_desc& a = p->ops[i];
            const dn_op_desc& d = p->ops[i + 1];
            if (plain_pw(a) && a.residual < 0 && uses[a.out] == 1 && dw_ok(d) && d.in == a.out && p->tensors[a.in].kind == DN_T_ACT &&
                expdw_supported(a.cin, a.cout, d.k, d.stride)) {
                const dn_tensor_desc& to = p->tensors[d.out];
                if (to.h * to.w > max_hw) continue;
                const bool can_proj = i + 2 < desc->n_ops && a.cin <= 96 && proj_ok(p->ops[i + 2], d, a.in);
                if (to.h * to.w < min_hw && !can_proj) continue;      // below the expand+depthwise threshold only whole blocks fuse
                if (can_proj) {
                    p->fused_len[i] = 3; p->fused_kind[i] = 3; i += 2;
                } else {
                    p->fused_len[i] = 2; p->fused_kind[i] = 1; i += 1;
                }
                continue;
            }
            // depthwise 3x3 -> project without an expand (first block of the MobileNets, 16 / 32 channels on the largest map): the
            // depthwise is computed straight into the projection's B fragments (pwdirect.hip pw_dw_direct_kernel), fused_kind 8
            // (measured: V3's 16-channel block batch 64 0.782 -> 0.765 ms in flight, 1.035 -> 1.02 one at a time; the 32-channel block of the V2
            // model at 300 x 300 level with the two launches -- two K steps of nine taps per lane for a projection that fills half a tile: DN_PW_DW=2 only)
            if (dn_knob("DN_PW_DW", 1) && dw_ok(a) && a.k == 3 && a.stride == 1 && a.pad == 1 && a.pool < 0 && (a.cin == 16 || (a.cin == 32 && dn_knob("DN_PW_DW", 1) == 2)) &&
                p->tensors[a.in].kind == DN_T_ACT && plain_pw(d) && d.in == a.out && uses[a.out] == 1 && d.cout <= 32 && d.cout % 8 == 0 &&
                (d.residual < 0 || (d.residual == a.in && d.cout == a.cin)) && p->tensors[a.in].h * p->tensors[a.in].w >= 32 &&
                fd_ok((unsigned long long)p->tensors[a.in].h * p->tensors[a.in].w, (unsigned)p->tensors[a.in].w)) {      // (the kernel's x = pixel % w)
                p->fused_len[i] = 2; p->fused_kind[i] = 8; i += 1;
                continue;
            }
            // (the same pair through the LDS-tiled block kernel was correct but slower than the two launches -- 16 channels leave half of the
            //  workgroup idle in the depthwise stage -- and is not dispatched)
        }
    }
    // ---- dense 3x3 conv -> MaxPool2d(2, 2) pairs (VGG conv1_2 / conv2_2): one launch, fused_kind 4 (convbig.hip conv_patch_kernel)
    {
        std::vector<int> uses(desc->n_tensors, 0);
        for (int i = 0; i < desc->n_ops; ++i) {
            const dn_op_desc& o = p->ops[i];
            uses[o.in]++;
            if (o.residual >= 0) uses[o.residual]++;
            if (o.se >= 0) uses[o.se]++;
        }
        for (int l = 0; l < desc->n_levels; ++l) uses[desc->level_tensor[l]] += 100;
        for (int i = 0; i + 1 < desc->n_ops; ++i) {
            const dn_op_desc& c = p->ops[i];
            const dn_op_desc& m = p->ops[i + 1];
            const dn_tensor_desc& tc = p->tensors[c.out];
            if (c.type == DN_OP_CONV && !c.head && c.k == 3 && c.stride == 1 && c.pad == 1 && c.dil == 1 && m.type == DN_OP_MAXPOOL && m.in == c.out &&
                m.k == 2 && m.stride == 2 && m.pad == 0 && p->fused_len[i] == 0 && p->tensors[c.in].h == tc.h &&
                ((uses[c.out] == 1 && conv_pool_ok(c.cin, c.cout, tc.h, tc.w)) ||
                 (!conv_patch_pool_ok(c.cin, c.cout, tc.h, tc.w) && conv_halo_pool_ok(c.cin, c.cout, tc.h, tc.w)))) {      // (the run-staged tile can write both maps)
                p->fused_len[i] = 2;
                p->fused_kind[i] = 4;
                ++i;
            }
        }
    }
    // ---- small squeeze-excitations (c <= 128, squeeze <= 32: the 40 x 40 blocks of MobileNetV3) are computed in the prologue of the
    //      projection that consumes them: one dependent launch (~10 us of pure latency) less per block
    // ---- stems: the split-fp16 matrix kernel (depthwise.hip stem_split_kernel) takes weights and bias scaled by a power of two such that the
    //      largest magnitude lands in [2^13, 2^15) (the low halves of the split then stay normal fp16 numbers), and a normalised image that fp16 holds
    //      (pixels in [0, 1]: |x| <= max(mean, 1 - mean) / std)
    p->stem_split_ok.assign(desc->n_ops, 0);
    p->stem_scale_log2.assign(desc->n_ops, 0);
    for (int i = 0; i < desc->n_ops; ++i) {
        const dn_op_desc& so = p->ops[i];
        if (so.type != DN_OP_STEM) continue;
        const float* hw = reinterpret_cast<const float*>(static_cast<const unsigned char*>(weights) + so.w_off);
        const float* hb = reinterpret_cast<const float*>(static_cast<const unsigned char*>(weights) + so.b_off);
        bool ok = true;
        float big = 0.f;
        for (int q = 0; q < so.k * so.k * 3 * so.cout; ++q) { ok = ok && std::isfinite(hw[q]); big = std::max(big, std::fabs(hw[q])); }
        for (int q = 0; q < so.cout; ++q) { ok = ok && std::isfinite(hb[q]); big = std::max(big, std::fabs(hb[q])); }
        for (int c = 0; c < 3; ++c) ok = ok && desc->std[c] > 0.f && std::max(std::fabs(desc->mean[c]), std::fabs(1.f - desc->mean[c])) / desc->std[c] < 3.0e4f;
        int e = 0;
        if (ok && big > 0.f) {
            (void)std::frexp(big, &e);           // big = m 2^e, m in [0.5, 1)
            e = 15 - e;                          // big 2^e in [2^14, 2^15)
            e = std::max(-100, std::min(100, e));
        }
        p->stem_split_ok[i] = ok ? 1 : 0;
        p->stem_scale_log2[i] = e;
    }
    p->se_fold.assign(desc->n_ops, -1);
    for (int i = 0; i + 1 < desc->n_ops; ++i) {
        const dn_op_desc& so = p->ops[i];
        const dn_op_desc& pj = p->ops[i + 1];
        if (so.type != DN_OP_SE || pj.type != DN_OP_PW || pj.se != so.out || pj.head) continue;
        const dn_tensor_desc& ti = p->tensors[pj.in];
        int users = 0;
        for (int q = 0; q < desc->n_ops; ++q) users += p->ops[q].se == so.out;
        if (users == 1 && pw_se_fold_supported(pj.cin, pj.cout, so.squeeze, ti.h * ti.w)) {
            // DN_SE_SMALL (default 1, see below): these small FCs go to the tail of the pooling depthwise launch instead and the projection runs
            // on the register-direct kernel with the scale applied to its x fragments
            if (dn_knob("DN_SE_SMALL", 1) && (ti.h * ti.w) % 32 == 0 && pj.cin <= 128 && depthwise_se_tail_supported(so.cin, so.squeeze)) continue;
            p->se_fold[i] = -2;
            p->se_fold[i + 1] = i;
        }
    }
    // ---- the other squeeze-excitations (opt-in, DN_SE_IN_DW=1): their FCs run in the tail of the depthwise launch that pools for
    //      them (the last workgroup of an image to finish; depthwise.hip dw_se_tail) instead of a 32-workgroup launch of their own.
    //      Needs the plain depthwise launch (not the fused expand+depthwise or tail runs) and the stem launch, which clears
    //      the counters. MEASURED and left off. With a device-scope fence per workgroup the 20 x 20 depthwise launches went from
    //      10 to 50 us (a release at agent scope writes back the XCD's L2: batch 64 1.12 -> 1.32 ms). With the fence-free publish
    //      (device-scope atomic stores / loads of the partial sums, relaxed ticket) the launch overhead is gone, but the FCs of the
    //      large blocks stream 230 - 450 KB of weights into ONE compute unit per image on 256 threads: batch 64 1.063 -> 1.068 ms.
    //      DN_SE_SMALL=1 does the same for the small squeeze-excitations only (instead of folding them into the projection's
    //      prologue), the projection then runs on the register-direct kernel with the scale applied to its x fragments:
    //      1.063 -> 1.054 ms at batch 64, no change at 32 / 16 one forward at a time; with three forwards in flight (pipeline.py)
    //      0.840 -> 0.825 ms at batch 64 and 0.463 -> 0.458 ms at 32: on by default.
    p->se_in_dw.assign(desc->n_ops, -1);
    p->se_slot.assign(desc->n_ops, -1);
    p->n_se_in_dw = 0;
    if ((dn_knob("DN_SE_IN_DW", 0) != 0 || dn_knob("DN_SE_SMALL", 1) != 0) && p->ops[0].type == DN_OP_STEM) {
        for (int i = 1; i < desc->n_ops; ++i) {
            const dn_op_desc& so = p->ops[i];
            if (so.type != DN_OP_SE || p->se_fold[i] == -2) continue;
            if (!dn_knob("DN_SE_IN_DW", 0) && !(so.cin <= 128 && so.squeeze <= 32)) continue;      // DN_SE_SMALL alone: the small ones only
            int j = -1;
            for (int q = 0; q < i; ++q) if (p->ops[q].type == DN_OP_DW && p->ops[q].pool == so.in) j = q;
            if (j < 1 || !depthwise_se_tail_supported(so.cin, so.squeeze)) continue;
            bool plain = p->fused_len[j] == 0 && !(p->fused_len[j - 1] >= 2);
            if (j >= 2 && p->fused_len[j - 2] >= 3) plain = false;
            if (!plain) continue;
            p->se_in_dw[j] = i;
            p->se_in_dw[i] = -2;
            p->se_slot[j] = p->n_se_in_dw++;
        }
    }
    p->post_ticket_slot = -1;
    for (int i = 0; i < desc->n_ops; ++i)
        if (p->ops[i].type == DN_OP_STEM) { p->post_ticket_slot = p->n_se_in_dw++; break; }
    // partial-sum rows of every pooled tensor = workgroups per image of its producing depthwise op
    p->pool_blocks.assign(desc->n_tensors, 0);
    for (int i = 0; i < desc->n_ops; ++i) {
        const dn_op_desc& o = p->ops[i];
        if (o.type == DN_OP_DW && o.pool >= 0 && i > 0 && p->fused_len[i - 1] >= 2 && (p->fused_kind[i - 1] & 1)) {
            const dn_tensor_desc& to = p->tensors[o.out];
            p->pool_blocks[o.pool] = expdw_tiles_per_image(to.h, to.w, o.stride);
            continue;
        }
        if (o.type == DN_OP_DW && o.pool >= 0) {
            const dn_tensor_desc& ti = p->tensors[o.in];
            const dn_tensor_desc& to = p->tensors[o.out];
            DwArgs a{};
            a.n = 1; a.h = ti.h; a.w_ = ti.w; a.c = o.cin; a.k = o.k; a.stride = o.stride; a.pad = o.pad; a.ho = to.h; a.wo = to.w;
            a.pool = reinterpret_cast<float*>(1);      // (geometry only: a pooling launch)
            p->pool_blocks[o.pool] = depthwise_pool_blocks(a);
        }
    }
    // stream assignment
    p->op_stream.assign(desc->n_ops, 0);
    p->op_wait_level.assign(desc->n_ops, -1);
    auto level_of = [&](int tensor) { for (int l = 0; l < desc->n_levels; ++l) if (desc->level_tensor[l] == tensor) return l; return -1; };
    for (int i = 0; i < desc->n_ops; ++i) {
        const dn_op_desc& o = p->ops[i];
        if (o.head) p->op_stream[i] = o.head;
        else
            for (int j = 0; j < desc->n_ops; ++j)
                if (p->ops[j].head && p->ops[j].in == o.out && level_of(o.out) < 0) p->op_stream[i] = p->ops[j].head;
    }
    for (int i = 0; i < desc->n_ops; ++i) {
        const dn_op_desc& o = p->ops[i];
        if (p->op_stream[i]) p->op_wait_level[i] = level_of(o.in);
    }
    {
        const bool enabled = getenv("DN_HEAD_GROUPS") ? atoi(getenv("DN_HEAD_GROUPS")) != 0 : true;
        int first = -1, last_main = -1;
        for (int i = 0; i < desc->n_ops; ++i) {
            if (p->op_stream[i]) { if (first < 0) first = i; }
            else last_main = i;
        }
        bool ok = enabled && first > last_main && first >= 0;
        int kinds = 0;
        for (int i = first; ok && i < desc->n_ops; ++i) {
            const dn_op_desc& o = p->ops[i];
            if (o.type == DN_OP_DW) {
                if (o.pool >= 0 || o.k != p->ops[first].k || o.stride != p->ops[first].stride || p->ops[first].type != DN_OP_DW) { ok = false; break; }
                p->head_dw.push_back(i);
            } else if ((o.type == DN_OP_PW || o.type == DN_OP_CONV) && o.head) {
                kinds |= (o.type == DN_OP_PW) ? 1 : 2;
                (o.head == 1 ? p->head_cls : p->head_reg).push_back(i);
            } else ok = false;
        }
        if (ok && (kinds == 3 || p->head_dw.size() > 12 || p->head_cls.size() > 8 || p->head_reg.size() > 8 || p->head_cls.empty())) ok = false;
        if (ok) p->head_first = first;
        else { p->head_dw.clear(); p->head_cls.clear(); p->head_reg.clear(); }
    }
    // ---- tail run: the longest suffix of the backbone (ops before the heads) that is a linear chain of tiny layers
    if ((getenv("DN_TAIL") ? atoi(getenv("DN_TAIL")) : 1) != 0) {
        const int end = p->head_first >= 0 ? p->head_first : desc->n_ops;
        int first = end;
        while (first > 0) {
            const int i = first - 1;
            const dn_op_desc& o = p->ops[i];
            const dn_tensor_desc& ti = p->tensors[o.in];
            const dn_tensor_desc& to = p->tensors[o.out];
            bool fused = p->fused_len[i] > 0;
            for (int q = 1; q <= 2 && i - q >= 0; ++q) fused |= p->fused_len[i - q] > q;
            if (fused || ti.kind != DN_T_ACT || !tail_op_supported(o, ti.h, ti.w, to.h, to.w)) break;
            if (first < end && p->ops[first].in != o.out) break;         // must feed the next op of the run
            --first;
        }
        if (end - first > TAIL_MAX_OPS) first = end - TAIL_MAX_OPS;
        if (end - first >= 2) {
            p->tail_first = first;
            p->tail_end = end;
            p->tail_materialise.assign(end - first, 0);
            for (int i = first; i < end; ++i) {
                const int t = p->ops[i].out;
                bool outside = false;
                for (int l = 0; l < desc->n_levels; ++l) outside |= desc->level_tensor[l] == t;
                for (int j = 0; j < desc->n_ops; ++j) {
                    if (j >= first && j < end) continue;
                    const dn_op_desc& u = p->ops[j];
                    outside |= (u.in == t || u.residual == t || u.se == t);
                }
                p->tail_materialise[i - first] = outside ? 1 : 0;
            }
        }
    }
    p->graph_mode = dn_knob("DN_GRAPH", 1) != 0;      // DN_GRAPH=0: plain launches (diagnostics)
    p->xcd = getenv("DN_XCD") ? atoi(getenv("DN_XCD")) != 0 : true;
    p->chain_graphs = dn_knob("DN_CHAIN_GRAPHS", -1);      // -1: auto (forward_impl)
    p->ws_reuse = dn_knob("DN_WS_REUSE", 1) != 0;
    p->split = getenv("DN_SPLIT") ? atoi(getenv("DN_SPLIT")) : 2;
    if (p->split < 1) p->split = 1;
    if (p->split > 4) p->split = 4;
    // anchor offsets per level
    int acc = 0;
    for (int l = 0; l < desc->n_levels; ++l) {
        const dn_tensor_desc& t = p->tensors[desc->level_tensor[l]];
        p->level_off.push_back(acc);
        acc += t.h * t.w * desc->anchors_per_loc[l];
    }
    if (acc != desc->num_anchors) {
        dn_set_error("dn_create: anchors per level sum to %d, desc says %d", acc, desc->num_anchors);
        return fail(DN_E_INVALID);
    }
    // the arena ends with 256 zero bytes: the source of out-of-image taps in convbig.hip
    p->zeros_off = (weight_bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc((void**)&p->weights_dev, p->zeros_off + 256);
    if (e == hipSuccess) e = hipMemset(p->weights_dev + p->zeros_off, 0, 256);
    if (e == hipSuccess) e = hipMemcpy(p->weights_dev, weights, weight_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&p->anchors_dev, (size_t)desc->num_anchors * 16);
    if (e == hipSuccess) e = hipMemcpy(p->anchors_dev, desc->anchors, (size_t)desc->num_anchors * 16, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        dn_set_error("dn_create: device allocation/upload failed: %s", hipGetErrorString(e));
        if (p->weights_dev) (void)hipFree(p->weights_dev);
        if (p->anchors_dev) (void)hipFree(p->anchors_dev);
        return fail(DN_E_HIP);
    }
    for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipStreamCreateWithFlags(&p->branch_stream[i], hipStreamNonBlocking);
    for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&p->ev_branch[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming);
    if (e != hipSuccess) {
        dn_set_error("dn_create: stream/event creation failed: %s", hipGetErrorString(e));
        return fail(DN_E_HIP);
    }
    p->weight_bytes = weight_bytes;
    p->d.anchors = nullptr;
    *out = p;
    return DN_OK;
}

extern "C" void dn_destroy(dn_plan* p) {
    if (!p) return;
    (void)hipDeviceSynchronize();       // forwards of this plan may still be in flight on the caller's streams: the weights go away below
    drop_graphs(p);
    if (p->capture_stream) (void)hipStreamDestroy(p->capture_stream);
    for (int i = 0; i < 3; ++i) if (p->branch_stream[i]) (void)hipStreamDestroy(p->branch_stream[i]);
    for (int i = 0; i < 3; ++i) if (p->ev_branch[i]) (void)hipEventDestroy(p->ev_branch[i]);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    for (auto ev : p->events) (void)hipEventDestroy(ev);
    if (p->weights_dev) (void)hipFree(p->weights_dev);
    if (p->anchors_dev) (void)hipFree(p->anchors_dev);
    delete p;
}

extern "C" size_t dn_workspace_bytes(const dn_plan* p, int n) {
    if (!p || n <= 0) return 0;
    return get_layout(const_cast<dn_plan*>(p), n).total;
}

// The in-forward split trades launch-bound kernels for two half-size chains that run in lockstep. A caller that keeps several
// FORWARDS in flight (one stream, workspace and output set each; demonet_amd/pipeline.py) overlaps chains that sit in different
// phases instead, and is better served by one whole-batch chain per forward: batch 64, three in flight: 0.84 ms per forward
// against 1.06 ms for one forward at a time as two chains (tools/pipeline_probe.py).
extern "C" int dn_set_chains(dn_plan* p, int chains) {
    DN_REQUIRE(p, "null plan");
    DN_REQUIRE(chains >= 0 && chains <= 4, "dn_set_chains: %d outside [0, 4] (0 = automatic)", chains);
    if (chains == p->chains_override) return DN_OK;
    p->chains_override = chains;
    // layouts, workspace sizes and captured graphs all depend on the split
    drop_graphs(p);
    p->layouts.clear();
    p->sub_layouts.clear();
    return DN_OK;
}

// dev hook: forget the captured graphs (tools that install debug hooks after the first forward)
extern "C" __attribute__((visibility("default"))) int dn_debug_clear_graphs(dn_plan* p) {
    DN_REQUIRE(p, "null plan");
    drop_graphs(p);
    return DN_OK;
}

extern "C" int dn_set_graph_mode(dn_plan* p, int enabled) {
    DN_REQUIRE(p, "null plan");
    p->graph_mode = enabled != 0;
    return DN_OK;
}

// view of rows [n0, n0 + ns) of the n-image layout, as a layout of its own (branch k of S)
static const Layout& get_sub_layout(dn_plan* p, int n, int S, int k) {
    auto key = std::make_pair(n, k);
    auto it = p->sub_layouts.find(key);
    if (it != p->sub_layouts.end()) return it->second;
    const Layout& L = get_layout(p, n);
    int n0 = 0;
    for (int q = 0; q < k; ++q) n0 += sub_count(n, S, q);
    const int ns = sub_count(n, S, k);
    Layout V;
    V.n = ns;
    V.toff = L.toff;
    V.tbytes = L.tbytes;
    for (size_t t = 0; t < L.toff.size(); ++t) {
        if (L.toff[t] == (size_t)-1) continue;
        const size_t per = L.tbytes[t] / (size_t)L.chain_n;
        V.toff[t] = L.arena ? L.toff[t] + (size_t)k * L.arena : L.toff[t] + (size_t)n0 * per;
        V.tbytes[t] = (size_t)ns * per;
    }
    V.resized_off = L.resized_off + (size_t)n0 * 3 * p->d.image_h * p->d.image_w * 4;
    V.logits_off = L.logits_off + (size_t)n0 * p->d.num_anchors * p->d.num_classes * 4;
    V.reg_off = L.reg_off + (size_t)n0 * p->d.num_anchors * 16;
    V.scale_off = L.scale_off + (size_t)n0 * 8;
    V.secnt_off = L.secnt_off + (size_t)n0 * p->n_se_in_dw * 4;
    const size_t slice = L.post_bytes / (size_t)S;
    V.post_off = L.post_off + (size_t)k * slice;
    V.post_bytes = slice;
    V.total = L.total;
    return p->sub_layouts.emplace(key, std::move(V)).first->second;
}

// ---------------------------------------------------------------------------------------------------------
// the launch sequence
// ---------------------------------------------------------------------------------------------------------
static int enqueue(dn_plan* p, const float* images, int n, int h, int w, float* boxes, float* scores, int64_t* labels,
                   int32_t* counts, unsigned char* ws, const Layout& L, bool heads_only, hipStream_t s, bool record,
                   float* packed, int ev0 = 0, int chain = 0) {
    const dn_model_desc& d = p->d;
    auto tptr = [&](int tid) -> void* { return ws + L.toff[tid]; };
    const float* net_in = images;
    const bool resize = (h != d.image_h || w != d.image_w);
    float* scale_xy = nullptr;
    if (p->input_u8) {
        // uint8 HWC decoder output: /255, bilinear resize and HWC -> planar in one pass (the stem then normalises on load)
        float* rz = reinterpret_cast<float*>(ws + L.resized_off);
        if (resize) scale_xy = reinterpret_cast<float*>(ws + L.scale_off);
        int rc = launch_u8hwc_to_planar(reinterpret_cast<const unsigned char*>(images), rz, scale_xy, n, h, w, d.image_h, d.image_w, s);
        if (rc) return rc;
        net_in = rz;
    } else if (resize) {
        // normalisation commutes with bilinear interpolation (affine per channel), so resizing the raw image first and
        // normalising on load in the stem equals transform.py:113-114 (normalize then resize) up to fp32 rounding.
        float* rz = reinterpret_cast<float*>(ws + L.resized_off);
        scale_xy = reinterpret_cast<float*>(ws + L.scale_off);
        int rc = launch_resize_bilinear(images, rz, scale_xy, n, h, w, d.image_h, d.image_w, s);
        if (rc) return rc;
        net_in = rz;
    }
    float* logits = reinterpret_cast<float*>(ws + L.logits_off);
    float* reg = reinterpret_cast<float*>(ws + L.reg_off);
    const unsigned char* const Wb = p->weights_dev;
    const int xq = (p->xcd && n >= 8) ? (n + 7) / 8 : 0;        // images per XCD group of this (sub-)batch, 0: plain mapping
    auto make_pw = [&](const dn_op_desc& o) {
        const dn_tensor_desc& ti = p->tensors[o.in];
        PwArgs a;
        a.x = reinterpret_cast<const half_t*>(tptr(o.in));
        a.w = reinterpret_cast<const half_t*>(Wb + o.w_off);
        a.wfrag = (o.type == DN_OP_PW && o.w2_off >= 0) ? reinterpret_cast<const half_t*>(Wb + o.w2_off) : nullptr;
        a.bias = reinterpret_cast<const float*>(Wb + o.b_off);
        a.residual = o.residual >= 0 ? reinterpret_cast<const half_t*>(tptr(o.residual)) : nullptr;
        a.se = o.se >= 0 ? reinterpret_cast<const float*>(tptr(o.se)) : nullptr;
        a.hw = ti.h * ti.w;
        a.m = n * a.hw;
        a.cin = o.cin; a.cout = o.cout; a.act = o.act;
        a.xq = xq;
        const int oi = (int)(&o - p->ops.data());
        if (oi >= 0 && oi < (int)p->ops.size() && p->se_fold[oi] >= 0) {
            // the squeeze-excitation FCs run in this projection's prologue: hand over the pooled partial sums and the FC weights
            const dn_op_desc& so = p->ops[p->se_fold[oi]];
            a.se = nullptr;
            a.sef_part = reinterpret_cast<const float*>(tptr(so.in));
            a.sef_nblk = p->pool_blocks[so.in];
            a.sef_sq = so.squeeze;
            a.sef_inv = 1.0f / (float)so.pool_pixels;
            a.sef_w1t = reinterpret_cast<const half_t*>(Wb + so.w_off); a.sef_b1 = reinterpret_cast<const float*>(Wb + so.b_off);
            a.sef_w2t = reinterpret_cast<const half_t*>(Wb + so.w2_off); a.sef_b2 = reinterpret_cast<const float*>(Wb + so.b2_off);
        }
        if (o.head) {
            const int cols = (o.head == 1) ? d.num_classes : 4;
            a.out = (o.head == 1) ? (void*)logits : (void*)reg;
            a.out_fp32 = 1;
            a.out_img_stride = (long)d.num_anchors * cols;
            a.out_base = (long)p->level_off[o.level] * cols;
        } else {
            a.out = tptr(o.out);
            a.out_fp32 = 0; a.out_img_stride = 0; a.out_base = 0;
        }
        return a;
    };
    auto make_dw = [&](const dn_op_desc& o) {
        const dn_tensor_desc& ti = p->tensors[o.in];
        const dn_tensor_desc& to = p->tensors[o.out];
        DwArgs a;
        a.x = reinterpret_cast<const half_t*>(tptr(o.in));
        a.w = reinterpret_cast<const half_t*>(Wb + o.w_off);
        a.bias = reinterpret_cast<const float*>(Wb + o.b_off);
        a.out = reinterpret_cast<half_t*>(tptr(o.out));
        a.n = n; a.h = ti.h; a.w_ = ti.w; a.c = o.cin; a.k = o.k; a.stride = o.stride; a.pad = o.pad; a.act = o.act;
        a.ho = to.h; a.wo = to.w;
        a.pool = o.pool >= 0 ? reinterpret_cast<float*>(tptr(o.pool)) : nullptr;
        a.pool_rows = o.pool >= 0 ? p->pool_blocks[o.pool] : 0;
        a.xq = xq;
        const int oi = (int)(&o - p->ops.data());
        if (oi >= 0 && oi < (int)p->ops.size() && p->se_in_dw[oi] >= 0) {
            const dn_op_desc& so = p->ops[p->se_in_dw[oi]];
            a.se_w1t = reinterpret_cast<const half_t*>(Wb + so.w_off); a.se_b1 = reinterpret_cast<const float*>(Wb + so.b_off);
            a.se_w2t = reinterpret_cast<const half_t*>(Wb + so.w2_off); a.se_b2 = reinterpret_cast<const float*>(Wb + so.b2_off);
            a.se_scale = reinterpret_cast<float*>(tptr(so.out));
            a.se_counter = reinterpret_cast<unsigned*>(ws + L.secnt_off) + (size_t)p->se_slot[oi] * n;
            a.se_sq = so.squeeze;
            a.se_inv = 1.0f / (float)so.pool_pixels;
        }
        return a;
    };
    auto make_conv = [&](const dn_op_desc& o) {
        const dn_tensor_desc& ti = p->tensors[o.in];
        const dn_tensor_desc& to = p->tensors[o.out];
        ConvArgs a;
        a.x = reinterpret_cast<const half_t*>(tptr(o.in));
        a.w = reinterpret_cast<const half_t*>(Wb + o.w_off);
        a.bias = reinterpret_cast<const float*>(Wb + o.b_off);
        a.zeros = reinterpret_cast<const half_t*>(Wb + p->zeros_off);
        a.n = n; a.h = ti.h; a.w_ = ti.w; a.cin = o.cin; a.cout = o.cout; a.k = o.k; a.stride = o.stride;
        a.pad = o.pad; a.dil = o.dil; a.act = o.act; a.ho = to.h; a.wo = to.w;
        a.xq = xq;
        if (o.head) {
            const int cols = (o.head == 1) ? d.num_classes : 4;
            a.out = (o.head == 1) ? (void*)logits : (void*)reg;
            a.out_fp32 = 1;
            a.out_img_stride = (long)d.num_anchors * cols;
            a.out_base = (long)p->level_off[o.level] * cols;
        } else {
            a.out = tptr(o.out);
            a.out_fp32 = 0; a.out_img_stride = 0; a.out_base = 0;
        }
        return a;
    };
    int ev = ev0;
    hipStream_t const main_stream = s;
    // set by launch_heads when the fused head launch also ran softmax + decode (headfuse.hip, SM): the post-process starts at the cut-off
    bool scores_ready = false;
    HistRows fused_rows;
    int small_first = -1;
    // head launches of the pyramid levels [lv0, lv1): the depthwise group, then the 1x1 / dense group(s), on stream hs
    auto launch_heads = [&](int lv0, int lv1, hipStream_t hs, bool rec, size_t& seg) -> int {
        int rc = DN_OK;
        auto hnote = [&](size_t op, size_t owner) {
            if (!rec) return;
            p->prof_kernel[op] = dn_last_kernel();
            p->prof_owner[op] = (int)owner;
        };
        std::vector<int> h_dw, h_cls, h_reg;
        for (int q : p->head_dw) if (p->op_wait_level[q] >= lv0 && p->op_wait_level[q] < lv1) h_dw.push_back(q);
        for (int q : p->head_cls) if (p->ops[q].level >= lv0 && p->ops[q].level < lv1) h_cls.push_back(q);
        for (int q : p->head_reg) if (p->ops[q].level >= lv0 && p->ops[q].level < lv1) h_reg.push_back(q);
        // SSDLite heads (depthwise 3x3 -> 1x1, class and box head per level): ONE launch with the depthwise computed inside the 1x1 GEMM's
        // operand staging (headfuse.hip) -- no depthwise output in HBM, no second launch. DN_HEAD_FUSE=0 keeps the two grouped launches
        // (the reference path of the bit-identity test).
        if (dn_knob("DN_HEAD_FUSE", 1) != 0 && !h_dw.empty() && !h_cls.empty() && !h_reg.empty()) {
            HeadFuseLevel fl[8];
            int nl = 0;
            std::vector<int> members;
            for (size_t q = 0; q < h_cls.size() && nl < 8; ++q) {
                // a level joins when its class and box head are both (depthwise 3x3 stride 1 -> 1x1) chains on the level's feature map
                const dn_op_desc& oc = p->ops[h_cls[q]];
                int qr = -1;
                for (int u : h_reg) if (p->ops[u].level == oc.level) qr = u;
                if (qr < 0 || oc.type != DN_OP_PW || p->ops[qr].type != DN_OP_PW) continue;
                const dn_op_desc& orr = p->ops[qr];
                int dc = -1, dr = -1;
                for (int u : h_dw) { if (p->ops[u].out == oc.in) dc = u; if (p->ops[u].out == orr.in) dr = u; }
                if (dc < 0 || dr < 0) continue;
                const dn_op_desc &odc = p->ops[dc], &odr = p->ops[dr];
                if (odc.in != odr.in || odc.k != 3 || odr.k != 3 || odc.stride != 1 || odr.stride != 1 || odc.pad != 1 || odr.pad != 1 || odc.dil != 1 ||
                    odr.dil != 1 || odc.act != odr.act || odc.cin != odr.cin || oc.cin != odc.cin || orr.cin != odc.cin || oc.act != DN_ACT_NONE ||
                    orr.act != DN_ACT_NONE || oc.se >= 0 || orr.se >= 0 || oc.residual >= 0 || orr.residual >= 0 || oc.w2_off < 0 || orr.w2_off < 0 ||
                    odc.pool >= 0 || odr.pool >= 0) continue;
                const dn_tensor_desc& ti = p->tensors[odc.in];
                HeadFuseLevel& f = fl[nl];
                f.x = reinterpret_cast<const half_t*>(tptr(odc.in));
                const dn_op_desc* op[2] = {&oc, &orr};
                for (int hsel = 0; hsel < 2; ++hsel) {
                    const PwArgs pa = make_pw(*op[hsel]);
                    f.wf[hsel] = pa.wfrag; f.bias[hsel] = pa.bias;
                    f.out[hsel] = reinterpret_cast<float*>(pa.out); f.out_img_stride[hsel] = pa.out_img_stride; f.out_base[hsel] = pa.out_base;
                    f.nc[hsel] = pa.cout;
                }
                f.wdg = odc.w2_off >= 0 ? reinterpret_cast<const half_t*>(Wb + odc.w2_off) : nullptr;
                f.wslot = odc.b2_off >= 0 ? Wb + odc.b2_off : nullptr;
                f.n = n; f.H = ti.h; f.W = ti.w; f.C = odc.cin; f.act = odc.act;
                if (!head_fused_level_supported(f)) continue;
                // DN_HEAD_FUSE_MINHW: levels with fewer pixels per image stay on the grouped launches. The fused workgroups take 512 residency
                // slots (2 per CU); at batch 64 levels 0 - 1 are 504 of them, every further workgroup starts a second round of the whole launch
                if (ti.h * ti.w < dn_knob("DN_HEAD_FUSE_MINHW", 0)) continue;
                if (nl > 0 && (dn_cdiv(f.nc[0], 32) + 4) / 4 != (dn_cdiv(fl[0].nc[0], 32) + 4) / 4) continue;      // (one instantiation per launch: channel tiles per wave)
                ++nl;
                members.push_back(dc); members.push_back(dr); members.push_back(h_cls[q]); members.push_back(qr);
            }
            if (nl > 0) {
                // softmax + decode + histogram rows in the same launch (DN_HEAD_SOFTMAX, default 1) when EVERY level's heads are in it and the
                // post-process follows (dn_forward_heads wants the logits themselves)
                HeadPost hp;
                bool with_post = false;
                // From 32 images per chain up (DN_HEAD_SOFTMAX_MINN): below that the launch is far from filling the chip and the epilogue is pure
                // latency on the chain (batch 32 as two chains of 16: 0.630 -> 0.648 ms one forward at a time; from 32 per chain up it gains).
                if (!heads_only && dn_knob("DN_HEAD_SOFTMAX", 1) != 0 && n >= dn_knob("DN_HEAD_SOFTMAX_MINN", 32) && lv0 == 0 && lv1 >= d.n_levels) {
                    const PostBuffers pb = post_buffers(ws + L.post_off, n, d.num_anchors, d.num_classes, d.topk_candidates);
                    int clamped = 0;
                    post_hist_range(d.score_thresh, &hp.hb0, &hp.nb, &clamped);
                    hp.scoresT = pb.scoresT; hp.boxes = pb.boxes; hp.hrows = pb.phist; hp.anchors = p->anchors_dev;
                    hp.A = d.num_anchors; hp.K = d.num_classes;
                    hp.img_w = (float)d.image_w; hp.img_h = (float)d.image_h; hp.score_thr = d.score_thresh;
                    // the levels with >= 32 pixels per image take the epilogue; they must be a prefix of the anchor axis (the rest -- anchors
                    // [small_first, A) -- gets its softmax in the cut-off launch, from the logits this launch writes for them)
                    HistRows hr;
                    int rows = 0, nsm = 0;
                    bool prefix = true;
                    for (int q = 0; q < nl; ++q) {
                        const int level = p->ops[members[4 * q + 2]].level;
                        fl[q].aoff = p->level_off[level];
                        fl[q].aloc = fl[q].nc[0] / d.num_classes;
                        fl[q].sm = fl[q].H * fl[q].W >= std::max(32, dn_knob("DN_HEAD_SM_MINHW", 32)) ? 1 : 0;
                        if (fl[q].sm) {
                            prefix = prefix && nsm == q && (q == 0 ? fl[q].aoff == 0 : fl[q].aoff == fl[q - 1].aoff + fl[q - 1].H * fl[q - 1].W * fl[q - 1].aloc);
                            fl[q].sbase = rows;
                            hr.hw[nsm] = fl[q].H * fl[q].W; hr.sbase[nsm] = rows; hr.grouped[nsm] = head_fused_grouped(xq, hr.hw[nsm]) ? 1 : 0;
                            rows += hist_rows_slots(hr.hw[nsm]);
                            ++nsm;
                        }
                    }
                    hr.levels = nsm;
                    hr.rows_per_image = rows;
                    // the epilogue levels must be pyramid levels 0 .. nsm - 1: everything behind them -- small levels of this launch and levels that did
                    // not join it (the V2 model's last level is a plain 1x1 conv on the grouped launches) -- stays in logit form
                    for (int q = 0; q < nsm; ++q) prefix = prefix && p->ops[members[4 * q + 2]].level == q;
                    // (the softmax tiles of those levels put their histogram rows behind these: the row stride of an image covers both)
                    const int sfirst = nsm < d.n_levels ? p->level_off[nsm] : d.num_anchors;
                    hp.rows_per_image = rows + dn_cdiv(d.num_anchors - sfirst, 64);
                    with_post = prefix && nsm > 0 && hp.rows_per_image <= pb.tiles && head_fused_post_supported(fl, nl, hp);
                    if (with_post) {
                        scores_ready = true; fused_rows = hr;
                        small_first = sfirst;
                    }
                }
                rc = launch_head_fused(fl, nl, xq, hs, with_post ? &hp : nullptr);
                if (rc != DN_OK) return rc;
                for (int q : members) hnote(q, seg);
                ++seg;
                if (rec && seg < p->ops.size()) (void)hipEventRecord(p->events[ev++], hs);
                // what did not join (the V2 model's last level is a plain 1x1 conv) takes the grouped launches below
                auto drop = [&](std::vector<int>& v) {
                    std::vector<int> keep;
                    for (int q : v) if (std::find(members.begin(), members.end(), q) == members.end()) keep.push_back(q);
                    v.swap(keep);
                };
                drop(h_dw); drop(h_cls); drop(h_reg);
                if (h_cls.empty() && h_reg.empty() && h_dw.empty()) return DN_OK;
            }
        }
        if (!h_dw.empty()) {
            DwArgs arr[12];
            for (size_t q = 0; q < h_dw.size(); ++q) arr[q] = make_dw(p->ops[h_dw[q]]);
            rc = launch_depthwise_group(arr, (int)h_dw.size(), hs);
            if (rc != DN_OK) return rc;
            for (int q : h_dw) hnote(q, seg);
            ++seg;
            if (rec) (void)hipEventRecord(p->events[ev++], hs);
        }
        // box and class heads of all levels in ONE launch when they fit (a dependent launch costs ~4.5 us even when empty, and
        // the narrow box heads then share the class heads' tile instead of running as a launch of their own)
        const bool merge_heads = dn_knob("DN_HEAD_MERGE", 1) != 0;
        // (dense-conv heads leave in a narrow and a wide launch of at most 12 problems each, so 7 levels x 2 heads still go together --
        // the box heads of the large levels must meet their class heads to ride in their tiles: 0.45 ms per 16 images on ssd512)
        const bool conv_heads = !h_reg.empty() && p->ops[h_reg[0]].type == DN_OP_CONV;
        const bool one = merge_heads && !h_reg.empty() && !h_cls.empty() && p->ops[h_reg[0]].type == p->ops[h_cls[0]].type &&
                         (h_reg.size() + h_cls.size() <= 12 || (conv_heads && h_reg.size() <= 12 && h_cls.size() <= 12));
        for (int kind = 0; kind < 2; ++kind) {
            std::vector<int> lst = kind ? h_cls : h_reg;
            if (one) {
                if (kind == 1) break;
                lst.insert(lst.end(), h_cls.begin(), h_cls.end());
            }
            if (lst.empty()) continue;
            std::vector<PwArgs> arr(lst.size());
            const bool conv = p->ops[lst[0]].type == DN_OP_CONV;
            int cnt = 0;
            std::vector<int> grouped;
            std::set<int> taken;
            // wide heads first, so that a box head that rides along is known before the groups are formed
            std::stable_sort(lst.begin(), lst.end(), [&](int x, int y) { return p->ops[x].cout > p->ops[y].cout; });
            for (size_t q = 0; q < lst.size(); ++q) {
                PwArgs pa = conv ? conv_to_pw(make_conv(p->ops[lst[q]])) : make_pw(p->ops[lst[q]]);
                if (conv && taken.count(lst[q])) continue;          // rides in another head's launch
                if (conv && conv_head_big_supported(pa)) {
                    // the wide dense heads of the large levels: MFMA-bound, each on the run-staged 256x256 tile. The box head of the
                    // same level (same input, 16 / 24 channels) fits in the idle part of its last channel tile.
                    int rider = -1;
                    for (size_t u = 0; u < lst.size() && rider < 0; ++u) {
                        const dn_op_desc &ou = p->ops[lst[u]], &oq = p->ops[lst[q]];
                        if (u != q && !taken.count(lst[u]) && ou.in == oq.in && ou.type == oq.type && ou.k == oq.k && ou.stride == oq.stride &&
                            ou.pad == oq.pad && ou.dil == oq.dil && ou.act == oq.act && ou.cout < oq.cout &&
                            dn_cdiv(oq.cout + ou.cout, 256) == dn_cdiv(oq.cout, 256))
                            rider = (int)u;
                    }
                    if (rider >= 0) {
                        // the tile is chosen from cout + cout_b: the rider joins only if the launch still has a tile WITH it (21 classes x 6 anchors =
                        // 126 + 24 channels cross a 128-channel tile boundary and fail the narrow-tile test the class head passed alone)
                        const PwArgs pb = conv_to_pw(make_conv(p->ops[lst[rider]]));
                        PwArgs with = pa;
                        with.w_b = pb.w; with.bias_b = pb.bias; with.out_b = pb.out; with.cout_b = pb.cout;
                        with.out_b_img_stride = pb.out_img_stride; with.out_b_base = pb.out_base;
                        if (conv_head_big_supported(with)) { pa = with; taken.insert(lst[rider]); }
                        else rider = -1;
                    }
                    rc = launch_conv_head_big(pa, hs);
                    if (rc != DN_OK) return rc;
                    hnote(lst[q], seg);
                    if (rider >= 0) hnote(lst[rider], seg);
                    ++seg;
                    if (rec && seg < p->ops.size()) (void)hipEventRecord(p->events[ev++], hs);
                    continue;
                }
                arr[cnt++] = pa;
                grouped.push_back(lst[q]);
            }
            if (cnt == 0) continue;
            // once the wide heads have left, the narrow box heads (16 / 24 channels) would pay the wide tile of the remaining
            // class heads: dense-conv groups are split into a narrow and a wide launch
            const bool split_narrow = conv && (cnt < (int)lst.size() || cnt > 12);
            for (int pass = 0; pass < (split_narrow ? 2 : 1); ++pass) {
                PwArgs sub[12];
                std::vector<int> ids;
                int nsub = 0;
                for (int q = 0; q < cnt; ++q) {
                    const bool narrow = arr[q].cout <= 32;
                    if (split_narrow && narrow != (pass == 0)) continue;
                    DN_REQUIRE(nsub < 12, "head group: more than 12 problems in one launch");
                    sub[nsub++] = arr[q];
                    ids.push_back(grouped[q]);
                }
                if (nsub == 0) continue;
                rc = launch_pointwise_group(sub, nsub, conv, hs);
                if (rc != DN_OK) return rc;
                for (int q : ids) hnote(q, seg);
                ++seg;
                if (rec && seg < p->ops.size()) (void)hipEventRecord(p->events[ev++], hs);
            }
        }
        return DN_OK;
    };
    // DN_POISON=1 (correctness tooling): a launch that fills every LDS byte and vector register with NaN patterns in front of every launch of
    // the forward -- results must not change (no kernel may read LDS or registers it has not written)
    const bool poison = !record && dn_knob("DN_POISON", 0) != 0;
    for (size_t i = 0; i < p->ops.size(); ++i) {
        if (poison) { int prc = launch_poison(main_stream); if (prc != DN_OK) return prc; }
        const dn_op_desc& o = p->ops[i];
        const dn_tensor_desc& ti = p->tensors[o.in];
        const dn_tensor_desc& to = p->tensors[o.out];
        s = main_stream;
        if (record) (void)hipEventRecord(p->events[ev++], s);
        int rc = DN_OK;
        const unsigned char* W = p->weights_dev;
        auto note = [&](size_t op, size_t owner) {
            if (!record) return;
            p->prof_kernel[op] = dn_last_kernel();
            p->prof_owner[op] = (int)owner;
        };
        if ((int)i == p->head_first) {
            // all remaining ops are head ops of the pyramid levels: three grouped launches instead of up to 28.
            // Profiling: the three launches take the event segments of ops i, i+1, i+2 (prof_owner maps members to them).
            size_t seg = i;
            rc = launch_heads(0, 1 << 20, s, record, seg);
            if (rc != DN_OK) return rc;
            for (size_t q = seg + 1; q < p->ops.size(); ++q)
                if (record) (void)hipEventRecord(p->events[ev++], s);
            break;
        }
        if ((int)i == p->tail_first) {
            TailArgs ta{};
            ta.count = p->tail_end - p->tail_first;
            ta.weights = reinterpret_cast<const half_t*>(W);
            const dn_op_desc& o0 = p->ops[i];
            ta.in0 = reinterpret_cast<const half_t*>(tptr(o0.in));
            ta.in0_stride = (long)(L.tbytes[o0.in] / (size_t)L.n / 2);
            ta.xq = xq;
            for (int q = 0; q < ta.count; ++q) {
                const dn_op_desc& oq = p->ops[i + q];
                const dn_tensor_desc& tq = p->tensors[oq.in];
                const dn_tensor_desc& uq = p->tensors[oq.out];
                TailOp& t = ta.op[q];
                t.type = oq.type; t.cin = oq.cin; t.cout = oq.type == DN_OP_DW ? oq.cin : oq.cout; t.k = oq.k; t.stride = oq.stride; t.pad = oq.pad;
                t.hin = tq.h; t.win = tq.w; t.hout = uq.h; t.wout = uq.w; t.act = oq.act;
                t.w_off = (long)((oq.type == DN_OP_PW ? oq.w2_off : oq.w_off) / 2); t.b_off = (long)oq.b_off;
                t.out = p->tail_materialise[q] ? reinterpret_cast<half_t*>(tptr(oq.out)) : nullptr;
                t.out_stride = (long)(L.tbytes[oq.out] / (size_t)L.n / 2);
            }
            rc = launch_tail(ta, n, s);
            if (rc != DN_OK) return rc;
            for (int q = 0; q < ta.count; ++q) note(i + q, i);
            for (int q = 1; q < ta.count; ++q)
                if (record) (void)hipEventRecord(p->events[ev++], s);
            i += ta.count - 1;
            continue;
        }
        if (p->fused_len[i] > 0 && p->fused_kind[i] == 8) {
            PwArgs pa = make_pw(p->ops[i + 1]);
            const DwArgs da = make_dw(o);
            pa.x = da.x;                                    // (the depthwise output is not materialised)
            if (pa.residual) pa.residual = da.x;
            rc = launch_pw_dw_direct(pa, da, s);
            if (rc != DN_OK) return rc;
            note(i, i); note(i + 1, i);
            if (record) (void)hipEventRecord(p->events[ev++], s);
            i += 1;
            continue;
        }
        if (p->fused_len[i] > 0 && p->fused_kind[i] == 4) {
            PwArgs pa = conv_to_pw(make_conv(o));
            if (L.toff[o.out] == (size_t)-1) pa.out = nullptr;      // (materialised only when something else reads the full-resolution map)
            pa.pool_out = reinterpret_cast<half_t*>(tptr(p->ops[i + 1].out));
            rc = launch_conv_pool(pa, s);
            if (rc != DN_OK) return rc;
            note(i, i); note(i + 1, i);
            if (record) (void)hipEventRecord(p->events[ev++], s);
            i += 1;
            continue;
        }
        if (p->fused_len[i] > 0) {
            const int kind = p->fused_kind[i], len = p->fused_len[i];
            const dn_op_desc* e = (kind & 1) ? &p->ops[i] : nullptr;
            const dn_op_desc& dwo = p->ops[i + ((kind & 1) ? 1 : 0)];
            const dn_op_desc* pj = (kind & 2) ? &p->ops[i + len - 1] : nullptr;
            const dn_tensor_desc& tin = p->tensors[p->ops[i].in];
            const dn_tensor_desc& tdo = p->tensors[dwo.out];
            ExpDwArgs a{};
            a.x = reinterpret_cast<const half_t*>(tptr(p->ops[i].in));
            a.out = reinterpret_cast<half_t*>(tptr(pj ? pj->out : dwo.out));
            a.pool = dwo.pool >= 0 ? reinterpret_cast<float*>(tptr(dwo.pool)) : nullptr;
            if (e) { a.w1 = reinterpret_cast<const half_t*>(W + e->w_off); a.b1 = reinterpret_cast<const float*>(W + e->b_off); a.act1 = e->act; }
            a.wd = reinterpret_cast<const half_t*>(W + dwo.w_off); a.bd = reinterpret_cast<const float*>(W + dwo.b_off); a.act2 = dwo.act;
            if (pj) { a.w3 = reinterpret_cast<const half_t*>(W + pj->w_off); a.b3 = reinterpret_cast<const float*>(W + pj->b_off); }
            a.n = n; a.H = tin.h; a.W = tin.w; a.Ho = tdo.h; a.Wo = tdo.w;
            a.cin = tin.c; a.cexp = dwo.cin; a.cout = pj ? pj->cout : dwo.cin;
            a.k = dwo.k; a.stride = dwo.stride; a.pad = dwo.pad;
            a.has_res = (pj && pj->residual >= 0) ? 1 : 0;
            a.xq = xq;
            rc = launch_expdw(a, s);
            if (rc != DN_OK) return rc;
            for (int q = 0; q < len; ++q) note(i + q, i);
            for (int q = 1; q < len; ++q)
                if (record) (void)hipEventRecord(p->events[ev++], s);
            i += len - 1;
            continue;
        }
        switch (o.type) {
            case DN_OP_STEM: {
                StemArgs a;
                a.img = net_in;
                a.w = reinterpret_cast<const float*>(W + o.w_off);
                a.bias = reinterpret_cast<const float*>(W + o.b_off);
                a.out = reinterpret_cast<half_t*>(tptr(o.out));
                a.n = n; a.h = ti.h; a.w_ = ti.w; a.cout = o.cout; a.k = o.k; a.stride = o.stride; a.pad = o.pad; a.act = o.act;
                a.ho = to.h; a.wo = to.w;
                for (int c = 0; c < 3; ++c) { a.mean[c] = d.mean[c]; a.inv_std[c] = 1.0f / d.std[c]; }
                a.xq = xq;
                a.split_ok = p->stem_split_ok[i];
                a.w_scale = std::ldexp(1.0f, p->stem_scale_log2[i]);
                a.w_unscale = std::ldexp(1.0f, -p->stem_scale_log2[i]);
                if (p->n_se_in_dw > 0) { a.zero_u32 = reinterpret_cast<unsigned*>(ws + L.secnt_off); a.zero_count = p->n_se_in_dw * n; }
                rc = launch_stem(a, s);
                break;
            }
            case DN_OP_PW:
                rc = launch_pointwise(make_pw(o), s);
                break;
            case DN_OP_DW:
                rc = launch_depthwise(make_dw(o), s);
                break;
            case DN_OP_SE: {
                if (p->se_fold[i] == -2) { dn_note_kernel("(se folded into the projection)"); break; }
                if (p->se_in_dw[i] == -2) { dn_note_kernel("(se in the tail of the depthwise launch)"); break; }
                rc = launch_se_fc(reinterpret_cast<const float*>(tptr(o.in)), p->pool_blocks[o.in], W + o.w_off,
                                  reinterpret_cast<const float*>(W + o.b_off), W + o.w2_off,
                                  reinterpret_cast<const float*>(W + o.b2_off), reinterpret_cast<float*>(tptr(o.out)), n,
                                  o.cin, o.squeeze, o.pool_pixels, s, xq);
                break;
            }
            case DN_OP_CONV:
                rc = launch_conv(make_conv(o), s);
                break;
            case DN_OP_MAXPOOL:
                rc = launch_maxpool(reinterpret_cast<const half_t*>(tptr(o.in)), reinterpret_cast<half_t*>(tptr(o.out)), n,
                                    ti.h, ti.w, ti.c, o.k, o.stride, o.pad, to.h, to.w, s);
                break;
            case DN_OP_L2NORM:
                rc = launch_l2norm(reinterpret_cast<const half_t*>(tptr(o.in)), reinterpret_cast<const float*>(W + o.w_off),
                                   reinterpret_cast<half_t*>(tptr(o.out)), (long)n * ti.h * ti.w, ti.c, s);
                break;
        }
        if (rc != DN_OK) return rc;
        note(i, i);
    }
    s = main_stream;
    if (!heads_only) {
        PostArgs a;
        a.logits = logits; a.reg = reg; a.anchors = p->anchors_dev;
        a.n = n; a.A = d.num_anchors; a.K = d.num_classes;
        a.img_h = (float)d.image_h; a.img_w = (float)d.image_w;
        a.scale_xy = nullptr;
        // ratio = original / network size in fp32 (transform.py:280-285), written by the resize kernel
        if (resize) a.scale_xy = scale_xy;
        a.score_thresh = d.score_thresh; a.nms_thresh = d.nms_thresh; a.topk = d.topk_candidates; a.dets = d.detections_per_img;
        a.boxes = boxes; a.scores = scores; a.labels = labels; a.counts = counts; a.kept_anchor = nullptr;
        a.packed = packed;
        a.ws = ws + L.post_off; a.ws_bytes = L.post_bytes;
        a.xq = xq;
        a.scores_ready = scores_ready; a.hrows = fused_rows; a.small_first = small_first;
        if (scores_ready && p->post_ticket_slot >= 0) a.tickets = reinterpret_cast<unsigned*>(ws + L.secnt_off) + (size_t)p->post_ticket_slot * n;
        hipEvent_t* pe = record ? &p->events[ev] : nullptr;
        int rc = launch_postprocess(a, s, pe);
        if (rc) return rc;
        ev += 5;
    } else if (record) {
        (void)hipEventRecord(p->events[ev++], s);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dn_set_error("kernel launch failed: %s", hipGetErrorString(e));
        return DN_E_HIP;
    }
    return DN_OK;
}

// the whole forward: one chain, or batch_split() sub-batch chains forked onto branch streams (parallel graph branches when
// captured). record = profiling: the sub-batches run back to back on `s`, each with its own block of events.
static int enqueue_all(dn_plan* p, const float* images, int n, int h, int w, float* boxes, float* scores, int64_t* labels,
                       int32_t* counts, unsigned char* ws, bool heads_only, hipStream_t s, bool record) {
    const int S = batch_split(p, n);
    if (S == 1) return enqueue(p, images, n, h, w, boxes, scores, labels, counts, ws, get_layout(p, n), heads_only, s, record, p->packed_out, 0);
    const size_t D = (size_t)p->d.detections_per_img;
    const int ev_stride = (int)p->ops.size() + 6;
    if (!record) DN_HIP_CHECK(hipEventRecord(p->ev_fork, s));
    size_t n0 = 0;
    for (int k = 0; k < S; ++k) {
        const int ns = sub_count(n, S, k);
        const Layout& V = get_sub_layout(p, n, S, k);
        hipStream_t bs = s;
        if (!record && k > 0) {
            bs = p->branch_stream[k - 1];
            DN_HIP_CHECK(hipStreamWaitEvent(bs, p->ev_fork, 0));
        }
        const float* sub_images = p->input_u8 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(images) + n0 * 3 * (size_t)h * w)
                                              : images + n0 * 3 * (size_t)h * w;
        int rc = enqueue(p, sub_images, ns, h, w, boxes ? boxes + n0 * D * 4 : nullptr, scores ? scores + n0 * D : nullptr,
                         labels ? labels + n0 * D : nullptr, counts ? counts + n0 : nullptr, ws, V, heads_only, bs, record,
                         p->packed_out ? p->packed_out + n0 * (D + 1) * 6 : nullptr, k * ev_stride, k);
        if (rc) return rc;
        if (!record && k > 0) DN_HIP_CHECK(hipEventRecord(p->ev_branch[k - 1], bs));
        n0 += ns;
    }
    if (!record)
        for (int k = 1; k < S; ++k) DN_HIP_CHECK(hipStreamWaitEvent(s, p->ev_branch[k - 1], 0));
    return DN_OK;
}

// chain k of S on stream s (no fork / join): what one per-chain graph captures
static int enqueue_chain(dn_plan* p, const float* images, int n, int h, int w, float* boxes, float* scores, int64_t* labels,
                         int32_t* counts, unsigned char* ws, bool heads_only, hipStream_t s, int S, int k) {
    const size_t D = (size_t)p->d.detections_per_img;
    size_t n0 = 0;
    for (int q = 0; q < k; ++q) n0 += sub_count(n, S, q);
    const int ns = sub_count(n, S, k);
    const Layout& V = get_sub_layout(p, n, S, k);
    const float* sub_images = p->input_u8 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(images) + n0 * 3 * (size_t)h * w)
                                          : images + n0 * 3 * (size_t)h * w;
    return enqueue(p, sub_images, ns, h, w, boxes ? boxes + n0 * D * 4 : nullptr, scores ? scores + n0 * D : nullptr,
                   labels ? labels + n0 * D : nullptr, counts ? counts + n0 : nullptr, ws, V, heads_only, s, false,
                   p->packed_out ? p->packed_out + n0 * (D + 1) * 6 : nullptr, 0, k);
}

static int forward_impl(dn_plan* p, const float* images, int n, int h, int w, float* boxes, float* scores, int64_t* labels,
                        int32_t* counts, void* workspace, size_t ws_bytes, void* stream, bool heads_only, bool u8 = false) {
    DN_REQUIRE(p && images && workspace, "dn_forward: null argument");
    DN_REQUIRE(n > 0 && h > 0 && w > 0, "dn_forward: bad shape n=%d h=%d w=%d", n, h, w);
    struct Busy {
        std::atomic<int>& f; bool ok;
        explicit Busy(std::atomic<int>& x) : f(x), ok(x.exchange(1) == 0) {}
        ~Busy() { if (ok) f.store(0); }
    } busy(p->in_call);
    DN_REQUIRE(busy.ok, "dn_forward: the plan is in use by another host thread (one plan serves one thread at a time; use one plan per thread)");
    // per-call state of the plan: set and cleared INSIDE the guard (a second thread that the guard rejects must not have touched it)
    struct U8 { dn_plan* p; U8(dn_plan* q, bool v) : p(q) { p->input_u8 = v; } ~U8() { p->input_u8 = false; } } u8_state(p, u8);
    DN_REQUIRE(heads_only || (boxes && scores && labels && counts), "dn_forward: null output buffer");
    const Layout& L = get_layout(p, n);
    if (ws_bytes < L.total) {
        dn_set_error("dn_forward: workspace %zu B < required %zu B for n=%d", ws_bytes, L.total, n);
        return DN_E_WORKSPACE;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
    p->heads_partial[workspace] = !heads_only;
    if (p->profiling) {
        const int S = batch_split(p, n);
        const size_t stride = p->ops.size() + 6;
        const size_t need = stride * S;
        while (p->events.size() < need) {
            hipEvent_t e;
            DN_HIP_CHECK(hipEventCreate(&e));
            p->events.push_back(e);
        }
        p->prof_kernel.assign(p->ops.size(), "");
        p->prof_owner.assign(p->ops.size(), -1);
        int rc = enqueue_all(p, images, n, h, w, boxes, scores, labels, counts, ws, heads_only, s, true);
        if (rc) return rc;
        DN_HIP_CHECK(hipStreamSynchronize(s));
        const size_t nseg = heads_only ? p->ops.size() : p->ops.size() + 4;      // + softmax/decode | cut-off + selection | merge | fallback
        if (p->prof_ms.size() < p->ops.size() + 4) p->prof_ms.assign(p->ops.size() + 4, 0.0);
        for (int k = 0; k < S; ++k)
            for (size_t i = 0; i < nseg; ++i) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, p->events[k * stride + i], p->events[k * stride + i + 1]);
                p->prof_ms[i] += ms;        // per op: summed over the sub-batch launches of one forward
            }
        p->prof_runs++;
        return DN_OK;
    }
    if (!p->graph_mode) return enqueue_all(p, images, n, h, w, boxes, scores, labels, counts, ws, heads_only, s, false);

    // Default: the sub-batch chains are parallel branches of ONE graph. DN_CHAIN_GRAPHS=1: one single-chain hipGraph per
    // sub-batch, each replayed on a stream of its own (the caller's and the plan's branch streams, forked / joined with events
    // around the launches). In isolation (tools/queue_probe.hip) single-chain graphs on different streams overlap completely and
    // each queue pays its own 1.7 us per dependent launch, while the branches of one graph pay ~2.9 us per launch one after
    // another; on the real chain the two forms measure the same at two chains (1.25 vs 1.24 ms) and per-chain graphs lose
    // badly at three or four (1.9 ms): kept as an opt-in for that measurement only.
    const int S = batch_split(p, n);
    // Measured (batch 32 = 2 x 16, 12 runs each): per-chain graphs 0.78 ms every time, one graph with two branches 0.785 ms in
    // two runs of three and 0.82 - 0.85 ms in the third (the placement of the branches differs from process to process); at batch
    // 64 one graph is 0.7 % faster (1.127 vs 1.135 ms). Default: per-chain graphs below 64 images.
    const bool want_chains = p->chain_graphs < 0 ? n < 64 : p->chain_graphs != 0;
    const bool per_chain = S > 1 && want_chains;
    const int flags = (heads_only ? 1 : 0) | (p->input_u8 ? 2 : 0);
    auto capture = [&](const GraphKey& key, hipGraphExec_t* out) -> int {
        hipGraph_t g = nullptr;
        if (!p->capture_stream) DN_HIP_CHECK(hipStreamCreateWithFlags(&p->capture_stream, hipStreamNonBlocking));
        hipStream_t cs = p->capture_stream;
        DN_HIP_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int rc;
        if (key.chain < 0) rc = enqueue_all(p, images, n, h, w, boxes, scores, labels, counts, ws, heads_only, cs, false);
        else rc = enqueue_chain(p, images, n, h, w, boxes, scores, labels, counts, ws, heads_only, cs, S, key.chain);
        hipError_t e = hipStreamEndCapture(cs, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) { dn_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return DN_E_HIP; }
        e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) { dn_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return DN_E_HIP; }
        return DN_OK;
    };
    GraphKey key{images, n, h, w, workspace, boxes, scores, labels, counts, flags, p->packed_out, per_chain ? 0 : -1};
    auto it = p->graphs.find(key);
    if (it == p->graphs.end()) {
        // first call with this signature: run once eagerly (sets function attributes, validates), then capture
        int rc = enqueue_all(p, images, n, h, w, boxes, scores, labels, counts, ws, heads_only, s, false);
        if (rc) return rc;
        if (p->graphs.size() + (per_chain ? S : 1) > 64)         // bound the cache (a pipeline of forwards holds one entry per slot and output set)
            drop_graphs(p);                                      // (drains the device first: other slots may be replaying these executables)
        for (int k = 0; k < (per_chain ? S : 1); ++k) {
            GraphKey kk = key;
            kk.chain = per_chain ? k : -1;
            hipGraphExec_t ge = nullptr;
            rc = capture(kk, &ge);
            if (rc) return rc;
            p->graphs.emplace(kk, ge);
        }
        return DN_OK;       // the eager run above already produced this call's results
    }
    if (!per_chain) {
        DN_HIP_CHECK(hipGraphLaunch(it->second, s));
        return DN_OK;
    }
    DN_HIP_CHECK(hipEventRecord(p->ev_fork, s));
    for (int k = 1; k < S; ++k) {
        GraphKey kk = key;
        kk.chain = k;
        auto itk = p->graphs.find(kk);
        DN_REQUIRE(itk != p->graphs.end(), "dn_forward: chain graph %d missing", k);
        hipStream_t bs = p->branch_stream[k - 1];
        DN_HIP_CHECK(hipStreamWaitEvent(bs, p->ev_fork, 0));
        DN_HIP_CHECK(hipGraphLaunch(itk->second, bs));
        DN_HIP_CHECK(hipEventRecord(p->ev_branch[k - 1], bs));
    }
    DN_HIP_CHECK(hipGraphLaunch(it->second, s));
    for (int k = 1; k < S; ++k) DN_HIP_CHECK(hipStreamWaitEvent(s, p->ev_branch[k - 1], 0));
    return DN_OK;
}

extern "C" int dn_forward(dn_plan* plan, const float* images_dev, int n, int h, int w, float* boxes_dev, float* scores_dev,
                          int64_t* labels_dev, int32_t* counts_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
    return forward_impl(plan, images_dev, n, h, w, boxes_dev, scores_dev, labels_dev, counts_dev, workspace_dev,
                        workspace_bytes, stream, false);
}

extern "C" int dn_forward_u8(dn_plan* plan, const uint8_t* images_dev, int n, int h, int w, float* boxes_dev, float* scores_dev,
                             int64_t* labels_dev, int32_t* counts_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
    DN_REQUIRE(plan, "dn_forward_u8: null plan");
    return forward_impl(plan, reinterpret_cast<const float*>(images_dev), n, h, w, boxes_dev, scores_dev, labels_dev, counts_dev,
                        workspace_dev, workspace_bytes, stream, false, true);
}

extern "C" int dn_forward_heads(dn_plan* plan, const float* images_dev, int n, int h, int w, void* workspace_dev,
                                size_t workspace_bytes, void* stream) {
    return forward_impl(plan, images_dev, n, h, w, nullptr, nullptr, nullptr, nullptr, workspace_dev, workspace_bytes, stream,
                        true);
}

extern "C" int dn_head_outputs(const dn_plan* p, void* workspace, int n, float** logits, float** reg) {
    DN_REQUIRE(p && workspace && n > 0, "dn_head_outputs: bad argument");
    {
        const auto it = p->heads_partial.find(workspace);
        DN_REQUIRE(it == p->heads_partial.end() || !it->second, "dn_head_outputs: the last forward on this workspace was dn_forward, which may compute softmax / box "
                   "decode inside the head launch and never writes the large levels' logits: run dn_forward_heads first");
    }
    const Layout& L = get_layout(const_cast<dn_plan*>(p), n);
    if (logits) *logits = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + L.logits_off);
    if (reg) *reg = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + L.reg_off);
    return DN_OK;
}

extern "C" int dn_tensor_ptr(const dn_plan* p, void* workspace, int n, int tensor_id, void** ptr, size_t* bytes) {
    DN_REQUIRE(p && workspace && n > 0 && ptr, "dn_tensor_ptr: bad argument");
    DN_REQUIRE(tensor_id >= 0 && tensor_id < (int)p->tensors.size(), "dn_tensor_ptr: tensor id %d out of range", tensor_id);
    const Layout& L = get_layout(const_cast<dn_plan*>(p), n);
    DN_REQUIRE(L.toff[tensor_id] != (size_t)-1, "dn_tensor_ptr: tensor %d is not materialised in the workspace", tensor_id);
    DN_REQUIRE(L.chain_n == n, "dn_tensor_ptr: with workspace reuse the rows of a tensor are not contiguous across the sub-batch chains of a "
               "%d-image forward (DN_WS_REUSE=0 keeps one block per tensor)", n);
    *ptr = reinterpret_cast<unsigned char*>(workspace) + L.toff[tensor_id];
    if (bytes) *bytes = L.tbytes[tensor_id];
    return DN_OK;
}

extern "C" int dn_batch_split(const dn_plan* p, int n) {
    DN_REQUIRE(p && n > 0, "dn_batch_split: bad argument");
    return batch_split(p, n);
}

extern "C" int dn_set_packed_output(dn_plan* p, float* packed_dev) {
    DN_REQUIRE(p, "null plan");
    DN_REQUIRE(p->in_call.load() == 0, "dn_set_packed_output: the plan is inside a forward of another host thread");
    p->packed_out = packed_dev;
    return DN_OK;
}

extern "C" int dn_profile_begin(dn_plan* p) {
    DN_REQUIRE(p, "null plan");
    p->profiling = true;
    p->prof_ms.assign(p->ops.size() + 4, 0.0);
    p->prof_runs = 0;
    return DN_OK;
}

extern "C" int dn_profile_op_info(const dn_plan* p, int op_index, char* kernel, int capacity, int32_t* owner) {
    DN_REQUIRE(p && kernel && owner && capacity > 0, "dn_profile_op_info: null argument");
    DN_REQUIRE(op_index >= 0 && (size_t)op_index < p->prof_kernel.size(), "dn_profile_op_info: op %d not profiled", op_index);
    snprintf(kernel, (size_t)capacity, "%s", p->prof_kernel[op_index].c_str());
    *owner = p->prof_owner[op_index];
    return DN_OK;
}

extern "C" int dn_profile_end(dn_plan* p, float* ms_per_op, int capacity) {
    DN_REQUIRE(p && ms_per_op, "null argument");
    p->profiling = false;
    const int nseg = (int)p->ops.size() + 4;
    DN_REQUIRE(capacity >= nseg, "dn_profile_end: capacity %d < %d", capacity, nseg);
    for (int i = 0; i < nseg; ++i) ms_per_op[i] = p->prof_runs ? (float)(p->prof_ms[i] / p->prof_runs) : 0.f;
    return p->prof_runs;
}

// ---------------------------------------------------------------------------------------------------------
// stand-alone entry points
// ---------------------------------------------------------------------------------------------------------
extern "C" size_t dn_postprocess_workspace_bytes(int n, int num_anchors, int num_classes, int topk, int dets) {
    return postprocess_ws_bytes(n, num_anchors, num_classes, topk, dets);
}

extern "C" int dn_postprocess(const float* logits, const float* reg, const float* anchors, int n, int A, int K, float image_h,
                              float image_w, const float* scale_xy, float score_thresh, float nms_thresh, int topk, int dets,
                              float* boxes, float* scores, int64_t* labels, int32_t* counts, int32_t* kept_anchor, void* ws,
                              size_t ws_bytes, void* stream) {
    DN_REQUIRE(logits && reg && anchors && boxes && scores && labels && counts && ws, "dn_postprocess: null argument");
    DN_REQUIRE(score_thresh >= 0.f, "dn_postprocess: score_thresh must be >= 0");
    PostArgs a;
    a.logits = logits; a.reg = reg; a.anchors = anchors; a.n = n; a.A = A; a.K = K;
    a.img_h = image_h; a.img_w = image_w; a.scale_xy = scale_xy;
    a.score_thresh = score_thresh; a.nms_thresh = nms_thresh; a.topk = topk; a.dets = dets;
    a.boxes = boxes; a.scores = scores; a.labels = labels; a.counts = counts; a.kept_anchor = kept_anchor;
    a.ws = ws; a.ws_bytes = ws_bytes;
    a.xq = xcd_images_per_group(n);
    return launch_postprocess(a, reinterpret_cast<hipStream_t>(stream), nullptr);
}

extern "C" int dn_pointwise_conv(const void* x, const void* w, const void* w_frag, const float* bias, const void* residual,
                                 const float* se, void* out, int m, int cin, int cout, int hw, int act, int out_fp32,
                                 int64_t out_img_stride, void* stream) {
    DN_REQUIRE(x && w && bias && out, "dn_pointwise_conv: null argument");
    PwArgs a;
    a.x = reinterpret_cast<const half_t*>(x); a.w = reinterpret_cast<const half_t*>(w); a.bias = bias;
    a.wfrag = reinterpret_cast<const half_t*>(w_frag);
    a.residual = reinterpret_cast<const half_t*>(residual); a.se = se; a.out = out;
    a.m = m; a.cin = cin; a.cout = cout; a.hw = hw; a.act = act; a.out_fp32 = out_fp32;
    a.out_img_stride = out_fp32 ? (long)out_img_stride : 0; a.out_base = 0;
    a.xq = (hw > 0 && m % hw == 0) ? xcd_images_per_group(m / hw) : 0;
    int rc = launch_pointwise(a, reinterpret_cast<hipStream_t>(stream));
    if (rc) return rc;
    DN_HIP_CHECK(hipGetLastError());
    return DN_OK;
}

extern "C" int dn_expand_depthwise(const void* x, const void* w1, const float* b1, const void* wd, const float* bd, const void* w3,
                                   const float* b3, void* out, float* pool_partial, int n, int h, int w, int cin, int cexp, int cout,
                                   int k, int stride, int act1, int act2, int has_res, void* stream) {
    DN_REQUIRE(x && wd && bd && out, "dn_expand_depthwise: null argument");
    DN_REQUIRE((w1 == nullptr) == (b1 == nullptr) && (w3 == nullptr) == (b3 == nullptr), "dn_expand_depthwise: weight without bias");
    ExpDwArgs a{};
    a.x = reinterpret_cast<const half_t*>(x); a.out = reinterpret_cast<half_t*>(out); a.pool = pool_partial;
    a.w1 = reinterpret_cast<const half_t*>(w1); a.b1 = b1; a.wd = reinterpret_cast<const half_t*>(wd); a.bd = bd;
    a.w3 = reinterpret_cast<const half_t*>(w3); a.b3 = b3;
    a.n = n; a.H = h; a.W = w; a.k = k; a.stride = stride; a.pad = (k - 1) / 2;
    a.Ho = (h + 2 * a.pad - k) / stride + 1; a.Wo = (w + 2 * a.pad - k) / stride + 1;
    a.cin = cin; a.cexp = cexp; a.cout = w3 ? cout : cexp; a.act1 = act1; a.act2 = act2; a.has_res = has_res;
    a.xq = xcd_images_per_group(n);
    int rc = launch_expdw(a, reinterpret_cast<hipStream_t>(stream));
    if (rc) return rc;
    DN_HIP_CHECK(hipGetLastError());
    return DN_OK;
}

extern "C" int dn_expand_depthwise_tiles(int ho, int wo, int stride) { return expdw_tiles_per_image(ho, wo, stride); }

extern "C" int dn_dense_conv(const void* x, const void* w, const float* bias, const void* zeros, void* out, int n, int h, int wd,
                             int cin, int cout, int k, int stride, int pad, int dil, int act, void* stream) {
    DN_REQUIRE(x && w && bias && out, "dn_dense_conv: null argument");
    DN_REQUIRE(n > 0 && h > 0 && wd > 0 && k >= 1 && stride >= 1 && dil >= 1 && pad >= 0, "dn_dense_conv: bad geometry");
    ConvArgs a;
    a.x = reinterpret_cast<const half_t*>(x); a.w = reinterpret_cast<const half_t*>(w); a.bias = bias; a.out = out;
    a.zeros = reinterpret_cast<const half_t*>(zeros);
    a.n = n; a.h = h; a.w_ = wd; a.cin = cin; a.cout = cout; a.k = k; a.stride = stride; a.pad = pad; a.dil = dil; a.act = act;
    a.ho = (h + 2 * pad - dil * (k - 1) - 1) / stride + 1;
    a.wo = (wd + 2 * pad - dil * (k - 1) - 1) / stride + 1;
    DN_REQUIRE(a.ho > 0 && a.wo > 0, "dn_dense_conv: empty output");
    a.out_fp32 = 0; a.out_img_stride = 0; a.out_base = 0;
    a.xq = xcd_images_per_group(n);
    int rc = launch_conv(a, reinterpret_cast<hipStream_t>(stream));
    if (rc) return rc;
    DN_HIP_CHECK(hipGetLastError());
    return DN_OK;
}

extern "C" int dn_depthwise_conv(const void* x, const void* w, const float* bias, void* out, int n, int h, int wd, int c, int k,
                                 int stride, int pad, int act, void* stream) {
    DN_REQUIRE(x && w && bias && out, "dn_depthwise_conv: null argument");
    DwArgs a;
    a.x = reinterpret_cast<const half_t*>(x); a.w = reinterpret_cast<const half_t*>(w); a.bias = bias;
    a.out = reinterpret_cast<half_t*>(out);
    a.n = n; a.h = h; a.w_ = wd; a.c = c; a.k = k; a.stride = stride; a.pad = pad; a.act = act;
    a.ho = (h + 2 * pad - k) / stride + 1;
    a.wo = (wd + 2 * pad - k) / stride + 1;
    a.xq = xcd_images_per_group(n);
    int rc = launch_depthwise(a, reinterpret_cast<hipStream_t>(stream));
    if (rc) return rc;
    DN_HIP_CHECK(hipGetLastError());
    return DN_OK;
}
