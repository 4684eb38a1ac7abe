// plan.hip — host runtime behind the C-ABI (include/esahrnet.h): stage table -> static plan
// (list of fused kernels over SB tensors), weight packing, workspace planning, forward.
//
// Reference being replaced: models/seg_hrnet.py:260-423 (module construction) and :425-473
// (forward).  The plan is NOT a transcription of the module tree:
//   * every BatchNorm2d is folded into its convolution by the host before esahrnet_set_conv;
//   * conv + bias + residual + ReLU are one kernel (conv_mfma.hip);
//   * the cross-resolution sums are one kernel per output branch (fuse.hip);
//   * last_layer[0] (1x1, 480->480 on the concatenated, up-sampled branches, 26 % of the
//     reference's MACs) is evaluated per branch at the branch's own resolution and summed by the
//     fuse kernel — a 1x1 convolution commutes with bilinear interpolation (both linear, the
//     interpolation weights act per channel), so conv(cat(up(x_b))) = sum_b up(conv_b(x_b))
//     exactly in real arithmetic; it needs 8x fewer MACs and never builds the 480-channel concat;
//   * last_layer[6] + concat + output_layer are one kernel (head.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/esahrnet.h"
#include "devstate.h"
#include "kernels.h"

namespace esa {
int final_kt(int K);
}

namespace {

thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return 1;
}

#define HIP_OK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

inline int pad32(int c) { return (c + 31) & ~31; }
constexpr int STEM_POOL_SLABS_MAX = 1024;   // pooled partials of the raw stem tensor: slabs reserved per image
inline int pad64(int c) { return (c + 63) & ~63; }

struct ConvSpec {               // one Conv2d of the reference module tree
    std::string name, bn;
    int cin, cout, k, stride, level;   // level: output resolution level (0 = crop, 1 = /2, ...)
    bool has_bias, relu;
    std::vector<float> w, b;    // folded weights as handed over by the host
    bool set = false;
    // derived spec (not visible to the host; filled at commit): the nine taps of input channels [pc0, pc1) of 3x3 spec `parent`
    // as ONE 1x1 convolution with 9 * cout outputs, output channel (co / 8) * 72 + tap * 8 + co % 8 (head_gather.hip)
    int parent = -1, pc0 = 0, pc1 = 0;
};

struct DevConv {                // one MFMA convolution launch (a ConvSpec or a cin-slice of one)
    int spec;                   // index into specs
    int c0, c1;                 // input-channel slice of the spec
    bool use_bias;
    bool out_f32 = false;       // output plain f32 NHWC instead of SB (head terms)
    std::vector<int> perm;      // optional input-channel permutation: packed ci -> spec ci
    int cinp, coutp;
    void* w = nullptr;          // device, packed
    float* bias = nullptr;      // device, f32 [coutp]
};

enum OpKind { OP_STEM, OP_STEMF, OP_CONV, OP_BLOCK, OP_FUSE, OP_HEAD, OP_HEADT, OP_HEAD2, OP_FINAL, OP_HEADBF,
              // seg_hrnet3 (CBAM) variant only:
              OP_STEMRAW, OP_POOL, OP_MLP, OP_MAPS, OP_APPLY, OP_RESAMPLE, OP_ZERO, OP_TONCHW, OP_GATHER };

struct AuxSpec {                // a non-conv parameter tensor (CBAM weights)
    std::string name;
    int shape[4];
    std::vector<float> data;
    bool set = false;
    float* dev = nullptr;
};

struct Tensor {
    int C, Cp, level;
    int flat = 0;               // > 0: plain f32 scratch of `flat` floats per sample (no spatial extent)
    std::string tap;            // name for esahrnet_tap_read, "" if anonymous
    bool tlayout = false;       // head term in the transposed T layout (head_t.hip): row pitch head_t_xp(w)
    int alt = 0;                // 0: always; 1 / 2: only when the first / second generation head runs
    int def = -1, last = -1;    // op indices
    size_t off = 0;             // per-shape plan
};

struct Op {
    OpKind kind;
    int dconv = -1;             // OP_CONV / OP_STEMF / OP_BLOCK (conv1)
    int dconv2 = -1;            // OP_BLOCK (conv2)
    int in = -1, out = -1, res = -1;
    int out2 = -1;              // second tensor this op writes (OP_STEMRAW: the pooled partials of its output), allocated with it
    int terms[4] = {-1, -1, -1, -1};
    int nterms = 0;
    bool relu = false;
    int lane = 0;               // execution lane (stream): 0 = caller's stream, 1..3 = side streams (schedule_waves)
    int wave = 0;               // waves run one after another; inside a wave the lanes run beside each other
    int wait0 = -1;             // side-lane launch: lane-0 op whose event it waits for first (-1: none)
    bool wait_entry = false;    // side-lane launch: waits for the wave's entry event first
    bool record = false;        // lane-0 op: an event is recorded behind it (a side lane forks from here)
    unsigned join = 0;          // lane-0 launch: side lanes (bit l) lane 0 waits for first
    int aux[3] = {-1, -1, -1};  // CBAM ops: fc.0, fc.2, sa.conv1
    int c0 = 0;                 // channel offset inside `out` (OP_APPLY / OP_RESAMPLE / OP_ZERO)
    int nchan = 0;              // OP_ZERO: channels to clear;  OP_RESAMPLE/OP_APPLY/OP_MAPS: real channels
    int align = 0;              // OP_RESAMPLE: align_corners
    int alt = 0;                // 0: always; 1: head_fused path only; 2: head_fused2 path only (chosen per shape)
    int multi = -1, mpos = 0;   // OP_CONV: member mpos of multi-head group `multi` (mpos 0 launches for all members)
    int jkey = -1;              // OP_CONV of an HRModule branch: (module id << 8) | position along the branch's conv chain
    int job = -1, jpos = 0;     // member jpos of job group `job` (independent same-depth convs of several branches, one launch)
};

// stride-2 3x3 convolutions of the SAME input tensor (the first links of the fuse-down chains of a stage), evaluated
// by one multi-head launch of the stream kernel: the input is read once instead of once per chain
struct Multi {
    int n = 0;
    int op[3] = {-1, -1, -1};   // op indices, consecutive: op[0] = leader
    void* w = nullptr;          // device: the members' packed weights, concatenated along cout
    float* bias = nullptr;
    int coutp = 0;
};

// the same-depth 3x3 convolutions of the branches of one HRModule: independent, evaluated by one launch of the stream
// kernel's multi-convolution form where every member is a 64-cout stream launch at the shape at hand
struct JobGroup {
    int n = 0;
    int op[6] = {-1, -1, -1, -1, -1, -1};   // op indices, consecutive: op[0] = leader
};

struct ShapePlan {
    int n = 0, h = 0, w = 0;
    bool keep = false;
    size_t bytes = 0;
    std::vector<int> lh, lw;    // resolution per level
    bool head2 = false;         // this shape runs the second-generation head (ops with alt == 2)
    bool head2_ulo = false;     // ... with the lo part of the interpolation weights
    std::vector<char> multi_on; // per Multi: evaluated as one launch at this shape
    std::vector<char> job_on;   // per JobGroup: evaluated as one launch at this shape
};

}  // namespace

int esa::set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return 1;
}

struct esahrnet_ctx {
    esahrnet_cfg cfg;
    int device;
    std::vector<ConvSpec> specs;
    std::map<std::string, int> spec_by_name;
    std::vector<DevConv> dconvs;
    std::vector<AuxSpec> aux;
    float *stemraw_w = nullptr, *stemraw_b = nullptr;     // variant 1: un-normalised conv1 for the skip
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    int spec_stem = -1, spec_final = -1;
    float *stem_w = nullptr, *stem_b = nullptr, *final_w = nullptr, *final_b = nullptr;
    void* final_wpk = nullptr;  // output_layer weights as MFMA fragments (head.hip, final_mfma_kernel)
    int spec_l0 = -1, spec_l3 = -1, head_c0 = 0;     // fused head (OP_HEAD)
    void *head_w0 = nullptr, *head_w3 = nullptr;
    float *head_b0 = nullptr, *head_b3 = nullptr;
    bool committed = false;
    bool keep = false;
    bool bf = false;            // cfg.precision == 1: tensors are single bf16 (sb.h "BF"), channels padded to 64
    int fmt = esa::FMT_SB;      // tensor format of the plan: FMT_SB (precision 0), FMT_BF (1), FMT_F32 (2: bf16x6 arithmetic)
    bool x6() const { return fmt == esa::FMT_F32; }
    int padc(int ch) const { return bf ? pad64(ch) : pad32(ch); }
    int eb() const { return bf ? 2 : 4; }       // bytes per stored channel
    size_t wbytes(int coutp, int cinp, int k) const {
        return bf ? esa::packed_weight_bytes_bf(coutp, cinp, k) : x6() ? esa::packed_weight_bytes_x6(coutp, cinp, k)
                                                                       : esa::packed_weight_bytes(coutp, cinp, k);
    }
    bool fuse_big = true;       // fused stem + fused head (ESAHRNET_UNFUSED=1 selects the op-by-op plan)
    bool head2_enabled = true;  // ESAHRNET_HEAD_V1=1 keeps the first-generation fused head for every shape
    bool x6_stemf = false;      // fp32-grade mode: conv1 inside conv2's staging (stem_x6_kernel), cin == 1
    bool cbam_unfused = false;  // ESAHRNET_CBAM_UNFUSED=1: cbam_maps + cbam_apply instead of cbam_spatial
    int head2_op = -1;          // index of the OP_HEAD2 op, -1 if the plan has none
    // wave executor (schedule_waves): the launches of a wave that do not depend on each other run on up to four lanes
    // (the caller's stream + three side streams), fork/join through the caller's stream at every wave boundary
    int nlanes = 1;             // 1: everything on the caller's stream;  4: waves (ESAHRNET_STREAMS)
    int nwaves = 1;
    std::vector<int> wave_last;                       // per wave: index of its last op
    std::vector<unsigned char> wave_mask;             // per wave: bit l set = lane l has work
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> wave_entry;               // per wave: recorded on the caller's stream when the wave opens
    std::vector<hipEvent_t> wave_end;                 // per op x side lane (Op::join) + the final join
    std::vector<hipEvent_t> op_event;                 // per op with Op::record
    ShapePlan sp;
    int max_level = 0;
    int nhost_specs = -1;       // specs [nhost_specs, end) are derived (ConvSpec::parent); -1: none
    int stemraw_partial = -1;   // seg_hrnet3: tensor of the pooled partials of the raw stem output when the stem kernel makes them
    std::vector<Multi> multis;
    std::vector<JobGroup> jobs;
};

namespace {

struct Builder {
    esahrnet_ctx& c;
    int jkey = -1;              // job key given to the next stride-1 convolutions (HRModule branches), -1: none
    int jkey2 = -1;             // job key given to the next stride-2 convolution (fuse-down chain link), -1: none
    explicit Builder(esahrnet_ctx& ctx) : c(ctx) {}

    int spec(const std::string& name, const std::string& bn, int cin, int cout, int k, int stride,
             int level, bool bias, bool relu) {
        ConvSpec s;
        s.name = name; s.bn = bn; s.cin = cin; s.cout = cout; s.k = k; s.stride = stride;
        s.level = level; s.has_bias = bias; s.relu = relu;
        c.specs.push_back(s);
        c.spec_by_name[name] = (int)c.specs.size() - 1;
        c.max_level = std::max(c.max_level, level);
        return (int)c.specs.size() - 1;
    }
    int taps_spec(int parent, int c0, int c1, int level) {
        const std::string name = c.specs[parent].name + "#taps[" + std::to_string(c0) + ":" + std::to_string(c1) + "]";
        const int cout9 = 9 * ((c.specs[parent].cout + 7) & ~7);
        const int sp = spec(name, "", c1 - c0, cout9, 1, 1, level, false, false);
        c.specs[sp].parent = parent; c.specs[sp].pc0 = c0; c.specs[sp].pc1 = c1;
        if (c.nhost_specs < 0) c.nhost_specs = sp;
        return sp;
    }
    int tensor(int C, int level, const std::string& tap = "") {
        Tensor t;
        t.C = C; t.Cp = c.padc(C); t.level = level; t.tap = tap;
        c.tensors.push_back(t);
        return (int)c.tensors.size() - 1;
    }
    void use(int t, int op) {
        if (t < 0) return;
        c.tensors[t].last = std::max(c.tensors[t].last, op);
    }
    // conv op on a spec (optionally a cin slice, optionally plain-f32 output); returns the output tensor
    int conv(int sp, int in, int res, bool relu, const std::string& tap = "", int c0 = 0, int c1 = -1,
             bool use_bias = true, bool out_f32 = false) {
        const ConvSpec& s = c.specs[sp];
        DevConv d;
        d.spec = sp; d.c0 = c0; d.c1 = c1 < 0 ? s.cin : c1; d.use_bias = use_bias; d.out_f32 = out_f32;
        d.cinp = c.padc(d.c1 - d.c0); d.coutp = c.padc(s.cout);
        c.dconvs.push_back(d);
        Op o;
        o.kind = OP_CONV; o.dconv = (int)c.dconvs.size() - 1; o.in = in; o.res = res; o.relu = relu;
        if (jkey >= 0 && s.k == 3 && s.stride == 1) o.jkey = jkey++;
        if (jkey2 >= 0 && ((s.k == 3 && s.stride == 2) || s.k == 1)) o.jkey = jkey2;
        // ESAHRNET_TAP_ALL=1 (debugging): every convolution output becomes a named tap
        o.out = tensor(s.cout, s.level, tap.empty() && !out_f32 && getenv("ESAHRNET_TAP_ALL") ? "conv:" + s.name : tap);
        const int idx = (int)c.ops.size();
        c.tensors[o.out].def = idx;
        use(in, idx); use(res, idx);
        c.ops.push_back(o);
        return o.out;
    }
    int fuse(const std::vector<int>& terms, int C, int level, bool relu, const std::string& tap = "") {
        Op o;
        o.kind = OP_FUSE; o.nterms = (int)terms.size(); o.relu = relu;
        o.out = tensor(C, level, tap);
        const int idx = (int)c.ops.size();
        c.tensors[o.out].def = idx;
        for (int i = 0; i < o.nterms; ++i) { o.terms[i] = terms[i]; use(terms[i], idx); }
        c.ops.push_back(o);
        return o.out;
    }
    int aux(const std::string& name, int a, int b, int kh, int kw) {
        AuxSpec s;
        s.name = name; s.shape[0] = a; s.shape[1] = b; s.shape[2] = kh; s.shape[3] = kw;
        c.aux.push_back(s);
        return (int)c.aux.size() - 1;
    }
    int push(Op& o) {
        const int idx = (int)c.ops.size();
        if (o.out >= 0 && c.tensors[o.out].def < 0) c.tensors[o.out].def = idx;
        use(o.in, idx); use(o.res, idx);
        for (int i = 0; i < 4; ++i) use(o.terms[i], idx);
        c.ops.push_back(o);
        return idx;
    }
    int flat_tensor(int floats_per_sample) {
        Tensor t;
        t.C = t.Cp = 0; t.level = 0; t.flat = floats_per_sample;
        c.tensors.push_back(t);
        return (int)c.tensors.size() - 1;
    }
    // CBAM (seg_hrnet3.py:32-61, 90-91): y[:, c0:c0+C] = [relu]( sa(ca*x) * (ca*x) [+ res] ); prefix p
    // names the owner of .ca / .sa ("" = the network's own).  Returns nothing: writes into `y`.
    static constexpr int POOL_SLABS = 64;
    void cbam(const std::string& p, int x, int C, int res, bool relu, int y, int y_c0) {
        const std::string q = p.empty() ? "" : p + ".";
        const int cr = C / 16;
        const int a0 = aux(q + "ca.fc.0.weight", cr, C, 1, 1), a1 = aux(q + "ca.fc.2.weight", C, cr, 1, 1);
        const int a2 = aux(q + "sa.conv1.weight", 1, 2, 7, 7);
        const int level = c.tensors[x].level, Cp = c.tensors[x].Cp;
        const int partial = flat_tensor(POOL_SLABS * Cp * 2), cav = flat_tensor(Cp);
        Tensor mt; mt.C = 2; mt.Cp = 2; mt.level = level;
        c.tensors.push_back(mt);
        const int maps = (int)c.tensors.size() - 1;
        // inside an HRModule branch (jkey >= 0) the four CBAM launches take the next depth keys behind the block's two
        // convolutions, so that group_jobs can put the module in depth-major order and merge the convolutions of its branches
        auto key = [&]() { return jkey >= 0 ? jkey++ : -1; };
        { Op o; o.kind = OP_POOL; o.in = x; o.out = partial; o.jkey = key(); push(o); }
        { Op o; o.kind = OP_MLP; o.in = partial; o.out = cav; o.aux[0] = a0; o.aux[1] = a1; o.nchan = C; o.terms[0] = x; o.jkey = key(); push(o); }
        { Op o; o.kind = OP_MAPS; o.in = x; o.terms[1] = cav; o.out = maps; o.nchan = C; o.jkey = key(); push(o); }
        { Op o; o.kind = OP_APPLY; o.in = x; o.res = res; o.terms[1] = cav; o.terms[2] = maps; o.aux[2] = a2;
          o.out = y; o.c0 = y_c0; o.relu = relu; o.nchan = C; o.jkey = key(); push(o); }
    }
    // BasicBlock of seg_hrnet3.py:64-103: conv-bn-relu-conv-bn, CBAM, (+res), relu
    int basic_block_cbam(const std::string& p, int x, int cin, int cout, int level, const std::string& tap) {
        int res = x;
        const int c1 = spec(p + ".conv1", p + ".bn1", cin, cout, 3, 1, level, false, true);
        const int c2 = spec(p + ".conv2", p + ".bn2", cout, cout, 3, 1, level, false, false);
        if (cin != cout) {
            const int d = spec(p + ".downsample.0", p + ".downsample.1", cin, cout, 1, 1, level, false, false);
            res = conv(d, x, -1, false);
        }
        const int o1 = conv(c1, x, -1, true);
        const int o2 = conv(c2, o1, -1, false);
        const int y = tensor(cout, level, tap);
        cbam(p, o2, cout, res, true, y, 0);
        return y;
    }
    // BasicBlock (seg_hrnet.py:32-61): conv-bn-relu-conv-bn (+res) relu
    int basic_block(const std::string& p, int x, int cin, int cout, int level, const std::string& tap) {
        if (c.cfg.variant == 1) return basic_block_cbam(p, x, cin, cout, level, tap);
        int res = x;
        const int c1 = spec(p + ".conv1", p + ".bn1", cin, cout, 3, 1, level, false, true);
        const int c2 = spec(p + ".conv2", p + ".bn2", cout, cout, 3, 1, level, false, true);
        // whole block in one kernel (bblock32.hip) — for layer1; inside the HRModules the block's two convolutions join the
        // same-depth convolutions of the other branches in one launch instead (group_jobs; ESAHRNET_NO_JOBS=1: fused block)
        const bool in_job = jkey >= 0 && !getenv("ESAHRNET_NO_JOBS");
        if (c.fuse_big && cin == cout && pad32(cin) == 32 && !getenv("ESAHRNET_NO_BBLOCK") && !in_job) {
            Op o;
            o.kind = OP_BLOCK; o.in = x; o.relu = true;
            for (int k = 0; k < 2; ++k) {
                DevConv d;
                d.spec = k ? c2 : c1; d.c0 = 0; d.c1 = cin; d.use_bias = true; d.cinp = 32; d.coutp = 32;
                c.dconvs.push_back(d);
                (k ? o.dconv2 : o.dconv) = (int)c.dconvs.size() - 1;
            }
            o.out = tensor(cout, level, tap);
            const int idx = (int)c.ops.size();
            c.tensors[o.out].def = idx;
            use(x, idx);
            c.ops.push_back(o);
            return o.out;
        }
        if (cin != cout) {
            const int d = spec(p + ".downsample.0", p + ".downsample.1", cin, cout, 1, 1, level, false, false);
            res = conv(d, x, -1, false);
        }
        const int o = conv(c1, x, -1, true);
        return conv(c2, o, res, true, tap);
    }
};

// Post-pass over the op list: find the stride-2 3x3 convolutions that share their input and make each such group
// consecutive (moving a member EARLIER is always legal: its only input is defined before the group's first member).
void group_multihead(esahrnet_ctx& c) {
    if (getenv("ESAHRNET_NO_MULTIHEAD") || c.fmt != esa::FMT_SB) return;
    auto eligible = [&](const Op& o) {
        if (o.kind != OP_CONV || o.res >= 0 || o.alt != 0 || o.multi >= 0) return false;
        const DevConv& d = c.dconvs[o.dconv];
        const ConvSpec& s = c.specs[d.spec];
        return s.k == 3 && s.stride == 2 && !d.out_f32 && d.c0 == 0 && d.c1 == s.cin && d.perm.empty() && d.use_bias;
    };
    std::vector<Op> ops = c.ops;
    for (size_t i = 0; i < ops.size(); ++i) {
        if (!eligible(ops[i])) continue;
        std::vector<size_t> members{i};
        for (size_t j = i + 1; j < ops.size() && members.size() < 3; ++j)
            if (eligible(ops[j]) && ops[j].in == ops[i].in) members.push_back(j);
        if (members.size() < 2) continue;
        // measured (W32, batch 32): worth it only where the shared input is the big 32/48-channel branch AND the
        // heads add up to whole 64-cout workgroup slices (stage 4: 53.4 -> 40.8 us); 96 couts on 32-cout slices
        // (stage 3: 40.1 -> 45.2 us) and the 64-channel input (31.8 -> 32.6 us) are not
        {
            int tot = 0;
            for (size_t k : members) tot += c.dconvs[ops[k].dconv].coutp;
            if (tot % 64 != 0 || c.dconvs[ops[i].dconv].cinp > 64 || c.tensors[ops[i].in].level > 1) continue;
        }
        const int mi = (int)c.multis.size();
        Multi m;
        m.n = (int)members.size();
        // pull the members up behind the leader (back to front so that the indices stay valid)
        std::vector<Op> pulled;
        for (size_t k = members.size(); k-- > 1;) {
            pulled.insert(pulled.begin(), ops[members[k]]);
            ops.erase(ops.begin() + members[k]);
        }
        ops.insert(ops.begin() + i + 1, pulled.begin(), pulled.end());
        for (int k = 0; k < m.n; ++k) {
            ops[i + k].multi = mi;
            ops[i + k].mpos = k;
            m.op[k] = (int)(i + k);
            m.coutp += c.dconvs[ops[i + k].dconv].coutp;
        }
        c.multis.push_back(m);
    }
    c.ops = ops;
    // op indices moved: recompute first definition / last use of every tensor and the head2 op index
    for (Tensor& t : c.tensors) { t.def = -1; t.last = -1; }
    c.head2_op = -1;
    for (size_t k = 0; k < c.ops.size(); ++k) {
        const Op& o = c.ops[k];
        if (o.kind == OP_HEAD2) c.head2_op = (int)k;
        if (o.out >= 0 && c.tensors[o.out].def < 0) c.tensors[o.out].def = (int)k;
        if (o.out2 >= 0 && c.tensors[o.out2].def < 0) c.tensors[o.out2].def = (int)k;
        auto use = [&](int t) { if (t >= 0) c.tensors[t].last = std::max(c.tensors[t].last, (int)k); };
        use(o.in); use(o.res);
        for (int t = 0; t < 4; ++t) use(o.terms[t]);
    }
}

// Post-pass: inside one HRModule the branches are independent chains of 3x3 convolutions.  The builder emits them
// branch by branch; here the convolutions of a module are put in depth-major order (stable, so every branch keeps its own
// order) and the same-depth ones of up to three branches become a JobGroup.
void group_jobs(esahrnet_ctx& c) {
    if (getenv("ESAHRNET_NO_JOBS")) return;
    std::vector<Op> ops = c.ops;
    auto eligible = [&](const Op& o) {
        // the CBAM launches of a depth (seg_hrnet3): same kind on every branch, one launch (cbam.hip: cbam_jobs_kernel)
        if ((o.kind == OP_POOL || o.kind == OP_MLP || o.kind == OP_MAPS || o.kind == OP_APPLY) && o.jkey >= 0)
            return !getenv("ESAHRNET_NO_CBAM_JOBS");
        if (o.kind != OP_CONV || o.jkey < 0 || o.alt != 0 || o.multi >= 0) return false;
        const DevConv& d = c.dconvs[o.dconv];
        const ConvSpec& s = c.specs[d.spec];
        if (s.k == 1) return !c.bf && o.res < 0 && !d.out_f32 && d.c0 == 0 && d.c1 == s.cin && d.perm.empty() && d.use_bias;
        return s.k == 3 && !d.out_f32 && d.c0 == 0 && d.c1 == s.cin && d.perm.empty() && d.use_bias &&
               d.coutp % (c.bf ? 64 : 32) == 0;
    };
    size_t i = 0;
    while (i < ops.size()) {
        if (ops[i].jkey < 0) { ++i; continue; }
        const int mod = ops[i].jkey >> 8;
        size_t j = i;
        while (j < ops.size() && ops[j].jkey >= 0 && (ops[j].jkey >> 8) == mod) ++j;      // [i, j): this module's branch convs
        std::stable_sort(ops.begin() + i, ops.begin() + j, [](const Op& a, const Op& b) { return (a.jkey & 255) < (b.jkey & 255); });
        for (size_t a = i; a < j;) {
            size_t b = a;
            std::vector<size_t> mem;
            const size_t cap = ops[a].kind == OP_CONV && c.specs[c.dconvs[ops[a].dconv].spec].k == 1 ? 6 : 4;
            while (b < j && ops[b].jkey == ops[a].jkey) { if (eligible(ops[b]) && mem.size() < cap) mem.push_back(b); ++b; }
            // members must be consecutive for the leader to stand for them: move the ineligible ones of this depth behind
            if (mem.size() >= 2) {
                std::vector<Op> grp, rest;
                for (size_t k = a; k < b; ++k) (std::find(mem.begin(), mem.end(), k) != mem.end() ? grp : rest).push_back(ops[k]);
                JobGroup g;
                g.n = (int)grp.size();
                for (int k = 0; k < g.n; ++k) { grp[k].job = (int)c.jobs.size(); grp[k].jpos = k; g.op[k] = (int)(a + k); }
                std::copy(grp.begin(), grp.end(), ops.begin() + a);
                std::copy(rest.begin(), rest.end(), ops.begin() + a + grp.size());
                c.jobs.push_back(g);
            }
            a = b;
        }
        i = j;
    }
    c.ops = ops;
    for (Tensor& t : c.tensors) { t.def = -1; t.last = -1; }
    c.head2_op = -1;
    for (size_t k = 0; k < c.ops.size(); ++k) {
        const Op& o = c.ops[k];
        if (o.kind == OP_HEAD2 || o.kind == OP_HEADBF) c.head2_op = (int)k;
        if (o.out >= 0 && c.tensors[o.out].def < 0) c.tensors[o.out].def = (int)k;
        if (o.out2 >= 0 && c.tensors[o.out2].def < 0) c.tensors[o.out2].def = (int)k;
        auto use = [&](int t) { if (t >= 0) c.tensors[t].last = std::max(c.tensors[t].last, (int)k); };
        use(o.in); use(o.res);
        for (int t = 0; t < 4; ++t) use(o.terms[t]);
    }
    // the multi-head groups refer to op indices too
    for (Multi& m : c.multis) m.n = 0;
    for (size_t k = 0; k < c.ops.size(); ++k)
        if (c.ops[k].multi >= 0) { Multi& m = c.multis[c.ops[k].multi]; m.op[c.ops[k].mpos] = (int)k; m.n = std::max(m.n, c.ops[k].mpos + 1); }
}

// Post-pass: lanes and waves.  Lane 0 is the caller's stream (the trunk), lanes 1..3 are side streams.  A launch unit (one
// op, or the consecutive members of a job / multi-head group) is placed by what it depends on among the not yet joined
// launches of the current wave — earlier writers of what it reads, earlier readers and writers of what it writes:
//   lane 0 only, incl. its tail   -> lane 0: it continues the trunk;
//   lane 0 only, not the tail     -> it forks onto the least loaded side lane behind the event of its latest lane-0
//                                    dependency (or the wave's entry event): what lane 0 has queued since runs beside it;
//   one side lane, not the trunk tail -> that side lane (behind the event of its latest lane-0 dependency, if newer);
//   one side lane + the trunk tail, or two and more side lanes -> lane 0, which first waits for those side lanes (join).
// A join that covers every side lane with work in flight ends the wave: all earlier launches are complete when the joining
// unit starts.  plan_shape() keeps a tensor alive to the end of the wave of its last reader, so recycled scratch adds no
// dependency inside a wave, and the schedule — true dependencies only — does not depend on the shape.
// Side lanes only ever wait on events of lane 0, and lane 0 on theirs: ROCm 7.x overflows its stack in
// hipStreamEndCapture when two non-origin streams of one capture have waited on each other (found with
// tools/ubench/segv_bt.c); fork/join through the origin stream captures fine (tools/ubench/graph_branches.hip).
void schedule_waves(esahrnet_ctx& c) {
    const size_t nops = c.ops.size();
    for (Op& o : c.ops) { o.lane = 0; o.wave = 0; o.wait0 = -1; o.wait_entry = false; o.record = false; o.join = 0; }
    c.nwaves = 1;
    if (c.nlanes > 1) {
        std::vector<std::vector<int>> readers(c.tensors.size()), writers(c.tensors.size());
        std::vector<int> unit_end(nops, 0);       // op -> last op of its launch unit
        int wave = 0, wave_start = 0, tail0 = -1;
        int load[4], waited0[4], tail[4], joined_tail[4], joined_op[4];
        bool active[4];
        auto reset = [&]() {
            for (int l = 0; l < 4; ++l) { load[l] = 0; waited0[l] = -1; tail[l] = -1; joined_tail[l] = -1; joined_op[l] = -1; active[l] = false; }
        };
        reset();
        for (size_t k = 0; k < nops;) {
            size_t e = k + 1;
            while (e < nops && ((c.ops[k].job >= 0 && c.ops[e].job == c.ops[k].job) ||
                                (c.ops[k].multi >= 0 && c.ops[e].multi == c.ops[k].multi))) ++e;
            unsigned side = 0;
            int zmax = -1;
            auto dep = [&](int d) {
                if (d < wave_start || d >= (int)k) return;
                const int l = c.ops[d].lane;
                if (l == 0) zmax = std::max(zmax, unit_end[d]);
                else if (d <= joined_tail[l]) zmax = std::max(zmax, joined_op[l]);      // already joined into the trunk
                else side |= 1u << l;
            };
            for (size_t m = k; m < e; ++m) {
                const Op& o = c.ops[m];
                for (int t : {o.in, o.res, o.terms[0], o.terms[1], o.terms[2], o.terms[3]})
                    if (t >= 0) for (int w : writers[t]) dep(w);
                if (o.out >= 0) {
                    for (int r : readers[o.out]) dep(r);
                    for (int w : writers[o.out]) dep(w);
                }
            }
            Op& lead = c.ops[k];
            int lane = 0;
            const int nside = __builtin_popcount(side);
            if (nside >= 2 || (nside == 1 && zmax == tail0)) {
                lead.join = side;
                bool all = true;
                for (int l = 1; l < 4; ++l) if (active[l] && !(side >> l & 1)) all = false;
                if (all) {
                    ++wave; wave_start = (int)k; tail0 = -1;
                    reset();
                } else {
                    for (int l = 1; l < 4; ++l)
                        if (side >> l & 1) { active[l] = false; joined_tail[l] = tail[l]; joined_op[l] = (int)e - 1; }
                }
            } else if (nside == 1) {
                lane = __builtin_ctz(side);
            } else if (zmax != tail0) {
                lane = 1;
                for (int l = 2; l < 4; ++l) if (load[l] < load[lane]) lane = l;
            }
            if (lane) {
                if (zmax > waited0[lane]) { lead.wait0 = zmax; c.ops[zmax].record = true; waited0[lane] = zmax; }
                else if (waited0[lane] < 0 && tail[lane] < 0) lead.wait_entry = true;
                active[lane] = true;
            }
            ++load[lane];
            for (size_t m = k; m < e; ++m) {
                Op& o = c.ops[m];
                o.lane = lane; o.wave = wave;
                unit_end[m] = (int)e - 1;
                for (int t : {o.in, o.res, o.terms[0], o.terms[1], o.terms[2], o.terms[3]}) if (t >= 0) readers[t].push_back((int)m);
                if (o.out >= 0) writers[o.out].push_back((int)m);
            }
            tail[lane] = (int)e - 1;
            if (!lane) tail0 = (int)e - 1;
            k = e;
        }
        c.nwaves = wave + 1;
    }
    c.wave_last.assign(c.nwaves, 0);
    c.wave_mask.assign(c.nwaves, 0);
    for (size_t k = 0; k < nops; ++k) {
        c.wave_last[c.ops[k].wave] = (int)k;
        c.wave_mask[c.ops[k].wave] |= (unsigned char)(1u << c.ops[k].lane);
    }
}

int build_plan_ops(esahrnet_ctx& c);
int build_plan(esahrnet_ctx& c) {
    if (build_plan_ops(c)) return 1;
    group_multihead(c);
    group_jobs(c);
    schedule_waves(c);
    return 0;
}

int build_plan_ops(esahrnet_ctx& c) {
    const esahrnet_cfg& g = c.cfg;
    Builder B(c);
    const int sw = g.stem_width;
    // ---- stem (seg_hrnet.py:265-270, 426-431) ----
    c.spec_stem = B.spec("conv1", "bn1", g.cin, sw, 3, 1, 0, false, true);
    const int spec_conv2 = B.spec("conv2", "bn2", sw, sw, 3, 2, 1, false, true);
    int x;
    int stem_raw = -1;
    if (g.variant == 1) {   // seg_hrnet3.py:473-475: x0 = conv1(x0) is kept (pre-BN) for the CBAM skip
        c.fuse_big = false;
        { Op o; o.kind = OP_STEMRAW; o.out = B.tensor(sw, 0, "stem_raw"); stem_raw = o.out;
          o.aux[0] = B.aux("conv1.weight", sw, g.cin, 3, 3); B.push(o); }
        // bn1 + ReLU + conv2 + bn2 + ReLU: the fused stem kernel (conv1 is evaluated a second time inside it,
        // with bn1 folded; the raw copy above exists only for the skip)
        DevConv d;
        d.spec = spec_conv2; d.c0 = 0; d.c1 = sw; d.use_bias = true;
        d.cinp = pad32(sw); d.coutp = pad32(sw);
        c.dconvs.push_back(d);
        Op o; o.kind = OP_STEMF; o.dconv = (int)c.dconvs.size() - 1; o.out = B.tensor(sw, 1, "stem2");
        B.push(o);
        x = o.out;
    } else if (c.fuse_big || c.x6_stemf) {       // conv1 recomputed per tile inside the conv2 kernel (stem_fused.hip / conv_x6.hip)
        DevConv d;
        d.spec = spec_conv2; d.c0 = 0; d.c1 = sw; d.use_bias = true;
        d.cinp = pad32(sw); d.coutp = pad32(sw);
        c.dconvs.push_back(d);
        Op o; o.kind = OP_STEMF; o.dconv = (int)c.dconvs.size() - 1; o.out = B.tensor(sw, 1, "stem2");
        c.tensors[o.out].def = 0;
        c.ops.push_back(o);
        x = o.out;
    } else {
        Op o; o.kind = OP_STEM; o.out = B.tensor(sw, 0, "stem1");
        c.tensors[o.out].def = 0;
        c.ops.push_back(o);
        x = B.conv(spec_conv2, c.ops[0].out, -1, true, "stem2");
    }
    // ---- layer1 (:277, :432) ----
    int cin = sw;
    const int nb1 = g.blocks[0][0];
    for (int k = 0; k < nb1; ++k) {
        x = B.basic_block("layer1." + std::to_string(k), x, cin, g.widths[0], 1, k == nb1 - 1 ? "layer1" : "");
        cin = g.widths[0];
    }
    std::vector<int> ys{x};
    std::vector<int> pre{g.widths[0]};
    for (int s = 2; s <= 4; ++s) {
        int nb = 0;
        while (nb < ESAHRNET_MAX_BRANCHES && g.blocks[s - 1][nb] > 0) ++nb;
        if (nb < (int)pre.size() || nb > (int)pre.size() + 1)
            return fail("stage %d: %d branches after %zu (must grow by at most one)", s, nb, pre.size());
        std::vector<int> cur(g.widths, g.widths + nb);
        const std::string t = "transition" + std::to_string(s - 1);
        // ---- transition (:343-377, wiring :434-457: new branch from ys.back()) ----
        std::vector<int> xs;
        for (int i = 0; i < nb; ++i) {
            if (i < (int)pre.size()) {
                if (pre[i] != cur[i]) {
                    const std::string q = t + "." + std::to_string(i);
                    xs.push_back(B.conv(B.spec(q + ".0", q + ".1", pre[i], cur[i], 3, 1, 1 + i, false, true), ys[i], -1, true));
                } else {
                    xs.push_back(ys[i]);
                }
            } else {
                int tt = ys.back();
                const int nconv = i + 1 - (int)pre.size();
                for (int j = 0; j < nconv; ++j) {
                    const std::string q = t + "." + std::to_string(i) + "." + std::to_string(j);
                    const int co = j == nconv - 1 ? cur[i] : pre.back();
                    tt = B.conv(B.spec(q + ".0", q + ".1", pre.back(), co, 3, 2, (int)pre.size() + j + 1, false, true), tt, -1, true);
                }
                xs.push_back(tt);
            }
        }
        // ---- HighResolutionModule x NUM_MODULES (:105-249) ----
        for (int m = 0; m < g.modules[s - 1]; ++m) {
            const std::string p = "stage" + std::to_string(s) + "." + std::to_string(m);
            for (int b = 0; b < nb; ++b) {
                for (int k = 0; k < g.blocks[s - 1][b]; ++k) {
                    // conv1 / conv2 of block k take keys 2k / 2k + 1 (seg_hrnet3: 6k .. 6k + 5 with the block's CBAM launches)
                    B.jkey = (((s << 4) | m) << 8) | ((g.variant == 1 ? 6 : 2) * k);
                    xs[b] = B.basic_block(p + ".branches." + std::to_string(b) + "." + std::to_string(k),
                                          xs[b], cur[b], cur[b], 1 + b, "");
                    B.jkey = -1;
                }
            }
            // fuse layers (:176-220, :232-247).  Emitted family by family, not output branch by output branch, so that
            // independent launches of one kernel family stand next to each other (group_jobs / group_multihead merge them):
            // all 1x1 fuse-up convolutions, then the stride-2 fuse-down chains link by link, then the sums.
            std::vector<std::vector<int>> terms(nb, std::vector<int>(nb, -1));
            const int modkey = ((s << 4) | m) << 8;
            for (int i = 0; i < nb; ++i) {
                terms[i][i] = xs[i];
                for (int j = i + 1; j < nb; ++j) {      // 1x1 + BN on the low-res grid; up-sampled inside fuse
                    const std::string q = p + ".fuse_layers." + std::to_string(i) + "." + std::to_string(j);
                    B.jkey2 = modkey | 100;             // all fuse-up 1x1 convolutions of the module are independent (before the chains)
                    terms[i][j] = B.conv(B.spec(q + ".0", q + ".1", cur[j], cur[i], 1, 1, 1 + j, false, false), xs[j], -1, false);
                    B.jkey2 = -1;
                }
            }
            for (int k = 0; k + 1 < nb; ++k)            // link k of every chain of 3x3 s2 (:198-217) that has one
                for (int i = k + 1; i < nb; ++i)
                    for (int j = 0; j + k < i; ++j) {
                        const bool last = k == i - j - 1;
                        const std::string qq = p + ".fuse_layers." + std::to_string(i) + "." + std::to_string(j) + "." + std::to_string(k);
                        const int sp_ = B.spec(qq + ".0", qq + ".1", cur[j], last ? cur[i] : cur[j], 3, 2, 1 + j + k + 1, false, !last);
                        B.jkey2 = modkey | (128 + k);
                        terms[i][j] = B.conv(sp_, k == 0 ? xs[j] : terms[i][j], -1, !last);
                        B.jkey2 = -1;
                    }
            std::vector<int> outs;
            for (int i = 0; i < nb; ++i) {
                const bool final_module = m == g.modules[s - 1] - 1;
                outs.push_back(B.fuse(terms[i], cur[i], 1 + i, true,
                                      final_module ? "stage" + std::to_string(s) + "." + std::to_string(i) : ""));
            }
            xs = outs;
        }
        ys = xs;
        pre = cur;
    }
    // ---- head (:313-340, :461-469) ----
    int tot = 0;
    for (int v : pre) tot += v;
    const int K = g.num_keypoints;
    if (g.variant == 1) {
        // seg_hrnet3.py:363-383, 506-520: cat of the up-sampled branches (materialised: last_layer[0] is a
        // 3x3 conv here), 3x3 480->480, 1x1 480->K, up x2 (align_corners=True), cat with CBAM(stem skip),
        // 3x3 (K+64)->K.  The second concat is laid out [skip | heat-maps] so that both slices start on
        // an 8-channel group; output_layer's input channels are permuted accordingly when packed.
        const int l0 = B.spec("last_layer.0", "last_layer.1", tot, tot, 3, 1, 1, true, true);
        const int l3 = B.spec("last_layer.3", "last_layer.4", tot, K, 1, 1, 1, true, true);
        c.spec_final = B.spec("output_layer.0", "", K + sw, K, 3, 1, 0, true, false);
        int h0, wide_h0 = 0;
        if (ys.size() == 4 && !c.bf && !getenv("ESAHRNET_HEAD3_DIRECT")) {
            // last_layer[0] by linearity (head_gather.hip): branches 2, 3 as nine 1x1 products on their own grids + a gather,
            // branch 0 and the up-sampled branch 1 as a direct 3x3 that takes the gather's result as its residual
            const int cd = pre[0] + pre[1];
            const int cat = B.tensor(cd, 1, "head_cat");
            int off = 0;
            for (int b = 0; b < 2; ++b) {
                Op o; o.kind = OP_RESAMPLE; o.in = ys[b]; o.out = cat; o.c0 = off; o.nchan = pre[b]; o.align = 0;
                B.push(o);
                off += pre[b];
            }
            if (pad32(cd) > ((cd + 7) & ~7)) {
                Op o; o.kind = OP_ZERO; o.out = cat; o.terms[0] = cat; o.c0 = (cd + 7) & ~7; o.nchan = pad32(cd) - ((cd + 7) & ~7);
                B.push(o);
            }
            int z[2];
            for (int b = 2; b < 4; ++b) {
                const int zs = B.taps_spec(l0, off, off + pre[b], 1 + b);
                z[b - 2] = B.conv(zs, ys[b], -1, false, "", 0, -1, false, true);      // plain f32: the gather needs no join
                off += pre[b];
            }
            Op gop; gop.kind = OP_GATHER; gop.in = z[0]; gop.nterms = 1; gop.terms[0] = z[1];
            gop.out = B.tensor(tot, 1, "head_gather");
            B.push(gop);
            h0 = B.conv(l0, cat, gop.out, true, "head0", 0, cd);
            // 64-cout workgroup slices for the direct convolution: the stream kernel stages an input tile once per 64 instead of
            // once per 32 couts (480 = 7.5 x 64: the output, its residual and the reader's input are padded to 512 channels,
            // the padding is exact zeros end to end: zero weights, zero bias, zero residual)
            if (pad64(tot) != pad32(tot) && !getenv("ESAHRNET_HEAD3_COUT32")) {
                const int cp = pad64(tot);
                c.tensors[gop.out].Cp = cp;
                c.tensors[h0].Cp = cp;
                c.dconvs[c.ops.back().dconv].coutp = cp;
                wide_h0 = cp;
            }
        } else {
            const int cat = B.tensor(tot, 1, "head_cat");
            int off = 0;
            for (size_t b = 0; b < ys.size(); ++b) {
                Op o; o.kind = OP_RESAMPLE; o.in = ys[b]; o.out = cat; o.c0 = off; o.nchan = pre[b]; o.align = 0;
                B.push(o);
                off += pre[b];
            }
            if (pad32(tot) > ((tot + 7) & ~7)) {
                Op o; o.kind = OP_ZERO; o.out = cat; o.terms[0] = cat; o.c0 = (tot + 7) & ~7; o.nchan = pad32(tot) - ((tot + 7) & ~7);
                B.push(o);
            }
            h0 = B.conv(l0, cat, -1, true, "head0");
        }
        const int h3 = B.conv(l3, h0, -1, true, "head3");
        if (wide_h0) c.dconvs[c.ops.back().dconv].cinp = wide_h0;
        const int cat2 = B.tensor(sw + K, 0, "head_cat2");
        const int first_cbam_op = (int)c.ops.size();
        B.cbam("", stem_raw, sw, -1, false, cat2, 0);
        // the pooling over the raw stem tensor rides in the kernel that writes it (stem.hip: launch_stem_pool): that op also
        // owns the partials, sized for the slabs that kernel makes (one per row piece) instead of pool_partial's 64
        if (c.ops[first_cbam_op].kind == OP_POOL && c.ops[first_cbam_op].in == stem_raw && !getenv("ESAHRNET_STEM_POOL_SEPARATE"))
            for (Op& so : c.ops)
                if (so.kind == OP_STEMRAW) {
                    const int partial = c.ops[first_cbam_op].out;
                    so.out2 = partial;
                    c.tensors[partial].flat = STEM_POOL_SLABS_MAX * c.tensors[stem_raw].Cp * 2;
                    c.stemraw_partial = partial;
                }
        { Op o; o.kind = OP_RESAMPLE; o.in = h3; o.out = cat2; o.terms[0] = cat2; o.c0 = sw; o.nchan = K; o.align = 1; B.push(o); }
        if (pad32(sw + K) > sw + ((K + 7) & ~7)) {
            Op o; o.kind = OP_ZERO; o.out = cat2; o.terms[0] = cat2; o.c0 = sw + ((K + 7) & ~7); o.nchan = pad32(sw + K) - o.c0;
            B.push(o);
        }
        const int oc = B.conv(c.spec_final, cat2, -1, false, "out_sb");
        std::vector<int>& perm = c.dconvs[c.ops.back().dconv].perm;     // packed ci -> reference ci
        for (int i = 0; i < sw; ++i) perm.push_back(K + i);              // skip channels come second in the reference
        for (int i = 0; i < K; ++i) perm.push_back(i);
        { Op o; o.kind = OP_TONCHW; o.in = oc; B.push(o); }
        return 0;
    }
    const int l0 = B.spec("last_layer.0", "last_layer.1", tot, tot, 1, 1, 1, true, true);
    const int l3 = B.spec("last_layer.3", "last_layer.4", tot, K, 1, 1, 1, true, true);
    c.spec_final = B.spec("output_layer.0", "", K + g.cin, K, 3, 1, 0, true, false);
    int h3;
    const bool fused_head = c.fuse_big && ys.size() == 4 && (pad32(pre[0]) == 32 || pad32(pre[0]) == 64);
    if (fused_head) {
        // t_b = W_b x_b on branch b's grid (f32 NHWC), b = 1..3; W_0, bias, ReLU, last_layer[3..5]
        // and the up-sampling of the t_b all happen inside head_fused.hip
        c.spec_l0 = l0; c.spec_l3 = l3; c.head_c0 = pre[0];
        // Two alternatives, chosen per input shape (plan_shape): alt 2 = head_t.hip + head_fused2.hip
        // (interpolation on the matrix cores, t_1 never materialised) when its geometry checks pass,
        // alt 1 = f32 NHWC terms + head_fused.hip otherwise.  Both read the same packed W_b slices.
        const bool have2 = c.head2_enabled && (pad32(pre[1]) == 64 || pad32(pre[1]) == 96) &&
                           esa::head_t_supported(pad32(pre[2])) && esa::head_t_supported(pad32(pre[3]));
        Op o; o.kind = OP_HEAD; o.in = ys[0]; o.nterms = 3; o.alt = have2 ? 1 : 0;
        int off = pre[0];
        int dslice[4] = {-1, -1, -1, -1};
        for (int b = 1; b < 4; ++b) {
            const int save = c.specs[l0].level;
            c.specs[l0].level = 1 + b;
            o.terms[b - 1] = B.conv(l0, ys[b], -1, false, "", off, off + pre[b], false, true);
            c.ops.back().alt = o.alt;
            c.tensors[o.terms[b - 1]].alt = o.alt;
            dslice[b] = c.ops.back().dconv;
            c.specs[l0].level = save;
            off += pre[b];
        }
        o.out = B.tensor(K, 1, "head3");
        c.tensors[o.out].Cp = (K + 15) & ~15;     // read only by head.hip: 16-channel pitch halves its traffic for K <= 16
        const int idx = (int)c.ops.size();
        c.tensors[o.out].def = idx;
        B.use(o.in, idx);
        for (int i = 0; i < 3; ++i) B.use(o.terms[i], idx);
        c.ops.push_back(o);
        h3 = o.out;
        if (have2) {
            int tt[4] = {-1, -1, -1, -1};
            for (int b = 2; b < 4; ++b) {
                Op t; t.kind = OP_HEADT; t.in = ys[b]; t.dconv = dslice[b]; t.alt = 2;
                t.out = B.tensor(tot, 1 + b);
                c.tensors[t.out].tlayout = true;
                c.tensors[t.out].alt = 2;
                B.push(t);
                tt[b] = t.out;
            }
            Op q; q.kind = OP_HEAD2; q.in = ys[0]; q.nterms = 3; q.alt = 2; q.dconv = dslice[1];
            q.terms[0] = ys[1]; q.terms[1] = tt[2]; q.terms[2] = tt[3];
            q.out = h3;
            c.head2_op = B.push(q);
        }
    } else {
        // bf16 mode: the slices t_1..t_3 are shared by two alternatives, chosen per input shape (plan_shape):
        // alt 2 = head_fused_bf.hip (W0, interpolation, ReLU, last_layer[3] in one kernel) where its source-region
        // geometry holds, alt 1 = slice 0 + fuse + 1x1 (the 720-channel tensors materialised) otherwise
        // (fp32-grade mode: the same arrangement with head_x6.hip; f32 NHWC slices, ESAHRNET_X6_UNFUSED_HEAD=1 keeps alt 1)
        const bool bf_head = ys.size() == 4 &&
                             ((c.bf && (c.padc(pre[0]) == 64 || c.padc(pre[0]) == 128) && !getenv("ESAHRNET_BF_UNFUSED_HEAD")) ||
                              (c.x6() && (c.padc(pre[0]) == 32 || c.padc(pre[0]) == 64) && !getenv("ESAHRNET_X6_UNFUSED_HEAD")));
        std::vector<int> hterms(ys.size(), -1);
        int off = 0;
        std::vector<int> offs;
        for (size_t b = 0; b < ys.size(); ++b) { offs.push_back(off); off += pre[b]; }
        for (size_t b = bf_head ? 1 : 0; b < ys.size(); ++b) {
            // the slice runs at branch b's own resolution: fix the output level of the slice conv
            const int save = c.specs[l0].level;
            c.specs[l0].level = 1 + (int)b;
            hterms[b] = B.conv(l0, ys[b], -1, false, "", offs[b], offs[b] + pre[b], b == 0);
            c.specs[l0].level = save;
        }
        int h3b = -1;
        if (bf_head) {
            const int first_alt1 = (int)c.ops.size();
            const int save = c.specs[l0].level;
            hterms[0] = B.conv(l0, ys[0], -1, false, "", 0, pre[0], true);
            c.specs[l0].level = save;
            c.spec_l0 = l0; c.spec_l3 = l3; c.head_c0 = pre[0];
            Op q; q.kind = OP_HEADBF; q.in = ys[0]; q.nterms = 3; q.alt = 2;
            for (int b = 1; b < 4; ++b) q.terms[b - 1] = hterms[b];
            q.out = B.tensor(K, 1, "head3_fused");
            c.tensors[q.out].Cp = K <= 16 ? 16 : 32;      // read only by the output-layer kernel
            c.tensors[q.out].alt = 2;
            h3b = q.out;
            // alternative 1 first (its ops and tensors carry alt = 1), then the fused op
            const int h0 = B.fuse(hterms, tot, 1, true, "head0");
            h3 = B.conv(l3, h0, -1, true, "head3");
            for (int k = first_alt1; k < (int)c.ops.size(); ++k) {
                c.ops[k].alt = 1;
                if (c.ops[k].out >= 0) c.tensors[c.ops[k].out].alt = 1;
            }
            c.head2_op = B.push(q);
        } else {
            const int h0 = B.fuse(hterms, tot, 1, true, "head0");
            h3 = B.conv(l3, h0, -1, true, "head3");
        }
        if (h3b >= 0) {
            Op o; o.kind = OP_FINAL; o.in = h3b; o.alt = 2;
            B.push(o);
            Op o1; o1.kind = OP_FINAL; o1.in = h3; o1.alt = 1;
            B.push(o1);
            return 0;
        }
    }
    {
        Op o; o.kind = OP_FINAL; o.in = h3;
        B.use(h3, (int)c.ops.size());
        c.ops.push_back(o);
    }
    return 0;
}

void level_dims(const esahrnet_ctx& c, int h, int w, std::vector<int>& lh, std::vector<int>& lw) {
    lh.assign(c.max_level + 1, 0);
    lw.assign(c.max_level + 1, 0);
    lh[0] = h; lw[0] = w;
    for (int l = 1; l <= c.max_level; ++l) { lh[l] = (lh[l - 1] + 1) / 2; lw[l] = (lw[l - 1] + 1) / 2; }
}

int check_shape(const esahrnet_ctx& c, int n, int h, int w) {
    if (n <= 0) return fail("batch must be positive (got %d)", n);
    if (h < 16 || w < 16 || (h & 1) || (w & 1))
        return fail("crop %dx%d: height and width must be even and >= 16 "
                    "(UpsamplingBilinear2d(x2) output must match the crop, seg_hrnet.py:330,469)", h, w);
    (void)c;
    return 0;
}

// does this input shape run the second-generation head (ops with alt == 2)?
bool head2_for_shape(const esahrnet_ctx& c, const std::vector<int>& lh, const std::vector<int>& lw, bool* ulo) {
    if (c.head2_op < 0 || !c.head2_enabled) return false;
    const Op& o = c.ops[c.head2_op];
    int th[3], tw[3];
    for (int i = 0; i < 3; ++i) {
        const int lv = c.tensors[o.terms[i]].level;
        th[i] = lh[lv]; tw[i] = lw[lv];
    }
    const Tensor& t0 = c.tensors[o.in];
    if (o.kind == OP_HEADBF && c.x6()) {
        if (ulo) *ulo = false;
        return esa::head_x6_supported(lh[t0.level], lw[t0.level], th, tw, t0.Cp, c.cfg.num_keypoints);
    }
    if (o.kind == OP_HEADBF) {
        if (ulo) *ulo = false;
        return esa::head_fused_bf_supported(lh[t0.level], lw[t0.level], th, tw, t0.Cp, c.cfg.num_keypoints);
    }
    return esa::head_fused2_supported(lh[t0.level], lw[t0.level], th, tw, t0.Cp, c.tensors[o.terms[0]].Cp,
                                      c.cfg.num_keypoints, ulo);
}

// first-fit interval allocator over op order; tensors die after their last use
// is this multi-head group evaluated as ONE launch at this shape?  (depends on the shape only through the kernel's limits)
bool multi_on_for(const esahrnet_ctx& c, const Multi& m, int n, const std::vector<int>& lh, const std::vector<int>& lw) {
    esa::ConvParams q{};
    const Op& o0 = c.ops[m.op[0]];
    const Tensor& ti = c.tensors[o0.in];
    const Tensor& to = c.tensors[o0.out];
    q.N = n; q.H = lh[ti.level]; q.W = lw[ti.level]; q.OH = lh[to.level]; q.OW = lw[to.level];
    q.Cinp = ti.Cp; q.Coutp = m.coutp; q.nheads = m.n;
    for (int k = 0; k < m.n; ++k) {
        q.hb[k + 1] = q.hb[k] + c.dconvs[c.ops[m.op[k]].dconv].coutp;
        q.yh[k] = reinterpret_cast<char*>(const_cast<esahrnet_ctx*>(&c));      // only tested against nullptr
    }
    return esa::conv_s2c32_multi_supported(q);
}

// ConvParams of a convolution op at a shape; `ws` == nullptr: shapes and flags only (pointers that are merely tested
// against nullptr get a non-null dummy)
esa::ConvParams conv_params_of(const esahrnet_ctx& c, const Op& o, int n, const std::vector<int>& lh, const std::vector<int>& lw,
                               char* ws) {
    const DevConv& d = c.dconvs[o.dconv];
    const Tensor& ti = c.tensors[o.in];
    const Tensor& to = c.tensors[o.out];
    esa::ConvParams p{};
    char* dummy = reinterpret_cast<char*>(const_cast<esahrnet_ctx*>(&c));
    p.x = ws ? ws + ti.off : dummy;
    p.y = ws ? ws + to.off : dummy;
    p.res = o.res >= 0 ? (ws ? ws + c.tensors[o.res].off : dummy) : nullptr;
    p.w = static_cast<const uint4*>(d.w); p.bias = d.bias;
    p.N = n; p.H = lh[ti.level]; p.W = lw[ti.level]; p.OH = lh[to.level]; p.OW = lw[to.level];
    p.Cinp = d.cinp; p.Coutp = d.coutp; p.relu = o.relu; p.out_f32 = d.out_f32 && !c.x6(); p.fmt = c.fmt;      // (fp32-grade: every tensor is plain f32)
    return p;
}

// is this job group evaluated as ONE launch at this shape?  Every member must be a stream-kernel launch of its own
// there (the kernel serving a layer depends on the shape only, never on the grouping)
bool job_on_for(const esahrnet_ctx& c, const JobGroup& g, int n, const std::vector<int>& lh, const std::vector<int>& lw) {
    if (c.ops[g.op[0]].kind != OP_CONV) return true;        // CBAM groups: the merged kernel runs every shape its members run
    esa::ConvParams ps[6];
    if (c.x6()) {       // fp32-grade mode: conv_x6_jobs_kernel serves every kernel size / stride of the module
        const ConvSpec& s0 = c.specs[c.dconvs[c.ops[g.op[0]].dconv].spec];
        for (int k = 0; k < g.n; ++k) {
            ps[k] = conv_params_of(c, c.ops[g.op[k]], n, lh, lw, nullptr);
            const ConvSpec& sk = c.specs[c.dconvs[c.ops[g.op[k]].dconv].spec];
            if (sk.k != s0.k || sk.stride != s0.stride) return false;
        }
        return esa::conv_x6_jobs_supported(ps, g.n, s0.k, s0.stride);
    }
    if (c.specs[c.dconvs[c.ops[g.op[0]].dconv].spec].k == 1) {
        for (int k = 0; k < g.n; ++k) ps[k] = conv_params_of(c, c.ops[g.op[k]], n, lh, lw, nullptr);
        return esa::conv1x1_jobs_supported(ps, g.n);
    }
    for (int k = 0; k < g.n; ++k) {
        ps[k] = conv_params_of(c, c.ops[g.op[k]], n, lh, lw, nullptr);
        const int stride = c.specs[c.dconvs[c.ops[g.op[k]].dconv].spec].stride;
        if (stride != c.specs[c.dconvs[c.ops[g.op[0]].dconv].spec].stride) return false;
        if (stride == 1 && !esa::conv_is_stream_s1(ps[k])) return false;
    }
    return esa::conv_jobs_supported(ps, g.n, c.specs[c.dconvs[c.ops[g.op[0]].dconv].spec].stride);
}

int plan_shape(esahrnet_ctx& c, int n, int h, int w) {
    if (c.sp.n == n && c.sp.h == h && c.sp.w == w && c.sp.keep == c.keep) return 0;
    if (check_shape(c, n, h, w)) return 1;
    ShapePlan sp;
    sp.n = n; sp.h = h; sp.w = w; sp.keep = c.keep;
    level_dims(c, h, w, sp.lh, sp.lw);
    // seg_hrnet3's head by linearity (head_gather.hip) stages the tap-product windows of two branches in LDS: a shape it
    // cannot serve is refused HERE with a message, not by a failed launch in the middle of a forward
    for (const Op& o : c.ops)
        if (o.kind == OP_GATHER) {
            const Tensor& to = c.tensors[o.out];
            const int zt[2] = {o.in, o.terms[0]};
            int zh[2], zw[2];
            for (int b = 0; b < 2; ++b) { zh[b] = sp.lh[c.tensors[zt[b]].level]; zw[b] = sp.lw[c.tensors[zt[b]].level]; }
            if (!esa::head_gather_supported(sp.lh[to.level], sp.lw[to.level], zh, zw, to.Cp))
                return fail("crop %dx%d: the interpolation windows of seg_hrnet3's last_layer[0] do not fit the gather kernel's LDS "
                            "budget (set ESAHRNET_HEAD3_DIRECT=1 before creating the net for the direct 480-channel 3x3)", h, w);
        }
    sp.head2 = head2_for_shape(c, sp.lh, sp.lw, &sp.head2_ulo);
    const int active_alt = sp.head2 ? 2 : 1;
    sp.multi_on.assign(c.multis.size(), 0);
    for (size_t mi = 0; mi < c.multis.size(); ++mi) sp.multi_on[mi] = multi_on_for(c, c.multis[mi], n, sp.lh, sp.lw) ? 1 : 0;
    sp.job_on.assign(c.jobs.size(), 0);
    for (size_t ji = 0; ji < c.jobs.size(); ++ji) sp.job_on[ji] = job_on_for(c, c.jobs[ji], n, sp.lh, sp.lw) ? 1 : 0;
    struct Free { size_t off, len; };
    std::vector<Free> free_list;
    size_t top = 0;
    auto bytes_of = [&](const Tensor& t) {
        const size_t wpix = t.tlayout ? (size_t)esa::head_t_xp(sp.lw[t.level]) : (size_t)sp.lw[t.level];
        size_t b = t.flat ? (size_t)n * t.flat * 4 : (size_t)n * sp.lh[t.level] * wpix * t.Cp * c.eb();
        return (b + 255) & ~(size_t)255;
    };
    auto alloc = [&](size_t len) {
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].len >= len) {
                const size_t off = free_list[i].off;
                free_list[i].off += len; free_list[i].len -= len;
                if (!free_list[i].len) free_list.erase(free_list.begin() + i);
                return off;
            }
        const size_t off = top;
        top += len;
        return off;
    };
    auto release = [&](size_t off, size_t len) {
        free_list.push_back({off, len});
        std::sort(free_list.begin(), free_list.end(), [](const Free& a, const Free& b) { return a.off < b.off; });
        for (size_t i = 0; i + 1 < free_list.size();)
            if (free_list[i].off + free_list[i].len == free_list[i + 1].off) {
                free_list[i].len += free_list[i + 1].len;
                free_list.erase(free_list.begin() + i + 1);
            } else ++i;
        if (!free_list.empty() && free_list.back().off + free_list.back().len == top) {
            top = free_list.back().off;
            free_list.pop_back();
        }
    };
    const size_t nops = c.ops.size();
    size_t high = 0;
    std::vector<char> allocated(c.tensors.size(), 0);
    for (size_t oi = 0; oi < nops; ++oi) {
        const Op& o = c.ops[oi];
        const bool active = o.alt == 0 || o.alt == active_alt;
        if (active && o.out >= 0 && !allocated[o.out]) {         // (slice writers re-use the allocation)
            allocated[o.out] = 1;
            const size_t len = bytes_of(c.tensors[o.out]);
            const size_t off = alloc(len);
            c.tensors[o.out].off = off;
            high = std::max(high, std::max(top, off + len));
        }
        if (active && o.out2 >= 0 && !allocated[o.out2]) {
            allocated[o.out2] = 1;
            const size_t len = bytes_of(c.tensors[o.out2]);
            const size_t off = alloc(len);
            c.tensors[o.out2].off = off;
            high = std::max(high, std::max(top, off + len));
        }
        // a job group runs as ONE launch: what one member reads last must not be handed to another member's output, so a
        // tensor whose last reader sits inside a group stays alive until the group's last member
        // (and with lanes: until the last op of the wave, whose launches run concurrently)
        auto last_of = [&](const Tensor& t) {
            if (t.last < 0) return t.last;
            if (c.nlanes > 1) return c.wave_last[c.ops[t.last].wave];
            if (c.ops[t.last].job >= 0) {
                const JobGroup& g = c.jobs[c.ops[t.last].job];
                return g.op[g.n - 1];
            }
            return t.last;
        };
        if (!c.keep)
            for (size_t ti = 0; ti < c.tensors.size(); ++ti) {
                Tensor& t = c.tensors[ti];
                if (allocated[ti] && last_of(t) == (int)oi) release(t.off, bytes_of(t));
            }
        high = std::max(high, top);
    }
    sp.bytes = high;
    c.sp = sp;
    return 0;
}

void free_weights(esahrnet_ctx& c) {
    for (DevConv& d : c.dconvs) {
        if (d.w) (void)hipFree(d.w);
        if (d.bias) (void)hipFree(d.bias);
        d.w = nullptr; d.bias = nullptr;
    }
    for (float** p : {&c.stem_w, &c.stem_b, &c.final_w, &c.final_b, &c.head_b0, &c.head_b3})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    for (void** p : {&c.head_w0, &c.head_w3, &c.final_wpk})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    for (AuxSpec& a : c.aux) if (a.dev) { (void)hipFree(a.dev); a.dev = nullptr; }
    for (Multi& m : c.multis) {
        if (m.w) (void)hipFree(m.w);
        if (m.bias) (void)hipFree(m.bias);
        m.w = nullptr; m.bias = nullptr;
    }
    for (float** p : {&c.stemraw_w, &c.stemraw_b}) if (*p) { (void)hipFree(*p); *p = nullptr; }
    for (std::vector<hipEvent_t>* v : {&c.wave_entry, &c.wave_end, &c.op_event}) {
        for (hipEvent_t& e : *v) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        v->clear();
    }
    for (int i = 0; i < 3; ++i) {
        if (c.side[i]) { (void)hipStreamDestroy(c.side[i]); c.side[i] = nullptr; }
    }
    c.committed = false;
}

template <typename T>
int upload(const std::vector<T>& host, void** dev) {
    HIP_OK(hipMalloc(dev, host.size() * sizeof(T)));
    HIP_OK(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

}  // namespace

// ============================================ C ABI ============================================
extern "C" {

const char* esahrnet_last_error(void) { return g_err; }
int esahrnet_abi_version(void) { return ESAHRNET_ABI_VERSION; }

int esahrnet_create(const esahrnet_cfg* cfg, int device, esahrnet_handle* out) {
    if (!cfg || !out) return fail("esahrnet_create: null argument");
    if (cfg->final_conv_kernel != 1) return fail("FINAL_CONV_KERNEL=%d unsupported (reference default 1)", cfg->final_conv_kernel);
    if (cfg->cin < 1 || cfg->cin > 4) return fail("cin=%d unsupported (1..4)", cfg->cin);
    if (cfg->num_keypoints < 1 || esa::final_kt(cfg->num_keypoints) < 0) return fail("num_keypoints=%d unsupported (1..32)", cfg->num_keypoints);
    if (cfg->stem_width < 1 || cfg->blocks[0][0] < 1) return fail("bad stem_width/blocks");
    if (cfg->variant != 0 && cfg->variant != 1) return fail("variant=%d unsupported (0: seg_hrnet/2, 1: seg_hrnet3)", cfg->variant);
    if (cfg->precision < 0 || cfg->precision > 2)
        return fail("precision=%d unsupported (0: split-bf16 'bf16x3', 1: bf16, 2: fp32-grade 'bf16x6')", cfg->precision);
    if (cfg->precision == 1 && cfg->variant != 0) return fail("precision 1 (bf16) is built for variant 0 only");
    if (cfg->variant == 1) {
        if (cfg->stem_width % 16) return fail("variant 1: stem_width must be a multiple of 16 (ChannelAttention ratio)");
        for (int b = 0; b < ESAHRNET_MAX_BRANCHES; ++b)
            if (cfg->blocks[3][b] > 0 && (cfg->widths[b] < 16 || cfg->widths[b] % 8))
                return fail("variant 1: branch widths must be >= 16 and multiples of 8 (got %d)", cfg->widths[b]);
    }
    for (int s = 1; s < 4; ++s)
        if (cfg->modules[s] < 1) return fail("NUM_MODULES of stage %d must be >= 1", s + 1);
    for (int b = 0; b < ESAHRNET_MAX_BRANCHES; ++b)
        if (cfg->blocks[3][b] > 0 && cfg->widths[b] < 1) return fail("width of branch %d must be positive", b);
    // the job keys of group_jobs pack (stage << 4 | module) above an 8-bit depth field in which the branch convolutions take
    // (2 or 6) * block + 0..5, the fuse-up 1x1s 100 and the fuse-down links 128 + k: reject stage tables that overflow either
    for (int s = 0; s < 4; ++s) {
        if (cfg->modules[s] > 15) return fail("NUM_MODULES of stage %d is %d (at most 15)", s + 1, cfg->modules[s]);
        for (int b = 0; b < ESAHRNET_MAX_BRANCHES; ++b)
            if ((cfg->variant == 1 ? 6 : 2) * cfg->blocks[s][b] + (cfg->variant == 1 ? 5 : 1) >= 100)
                return fail("NUM_BLOCKS of stage %d branch %d is %d (at most %d)", s + 1, b, cfg->blocks[s][b], cfg->variant == 1 ? 15 : 49);
    }
    esahrnet_ctx* c = new esahrnet_ctx();
    c->cfg = *cfg;
    c->device = device;
    if (const char* e = getenv("ESAHRNET_UNFUSED")) c->fuse_big = !(e[0] && e[0] != '0');
    if (cfg->precision == 1) {      // bf16 mode: op-by-op plan on the stream / 1x1 / fuse kernels (the fused stem, block
        c->bf = true;               // and head kernels are built for the split format only)
        c->fuse_big = false;
        c->fmt = esa::FMT_BF;
    }
    if (cfg->precision == 2) {      // fp32-grade mode: f32 NHWC tensors, bf16x6 arithmetic (conv_x6.hip); op-by-op plan
        c->fmt = esa::FMT_F32;
        c->fuse_big = false;
        c->x6_stemf = esa::stem_fused_x6_supported(cfg->cin, pad32(cfg->stem_width), pad32(cfg->stem_width)) && cfg->stem_width == 64 &&
                      !getenv("ESAHRNET_X6_UNFUSED_STEM");
    }
    if (const char* e = getenv("ESAHRNET_STREAMS")) c->nlanes = atoi(e) > 1 ? 4 : 1;
    if (const char* e = getenv("ESAHRNET_HEAD_V1")) c->head2_enabled = !(e[0] && e[0] != '0');
    c->cbam_unfused = getenv("ESAHRNET_CBAM_UNFUSED") != nullptr;
    if (build_plan(*c)) { delete c; return 1; }
    *out = c;
    return 0;
}

int esahrnet_destroy(esahrnet_handle h) {
    if (!h) return 0;
    if (h->committed) { (void)hipSetDevice(h->device); }
    free_weights(*h);
    delete h;
    return 0;
}

int esahrnet_handle_device(esahrnet_handle h) { return h ? h->device : -1; }

int esahrnet_debug_devstate(int* kernel_device_entries, int* devices) {
    esa::dev_state_counts(kernel_device_entries, devices);
    return 0;
}

int esahrnet_debug_set_launch_limit(long long bytes) {
    esa::set_stream_launch_limit(bytes);
    return 0;
}

int esahrnet_debug_op_schedule(esahrnet_handle h, int index, int* wave, int* lane) {
    if (!h || !wave || !lane || index < 0 || index >= (int)h->ops.size()) return fail("debug_op_schedule: bad argument");
    *wave = h->ops[index].wave;
    *lane = h->ops[index].lane;
    return 0;
}

// the convolutions the host fills: derived specs (ConvSpec::parent) sit behind them and are filled at commit
static int host_specs(const esahrnet_ctx* h) { return h->nhost_specs >= 0 ? h->nhost_specs : (int)h->specs.size(); }
int esahrnet_conv_count(esahrnet_handle h) { return h ? host_specs(h) : -1; }

int esahrnet_conv_desc_get(esahrnet_handle h, int i, esahrnet_conv_desc* out) {
    if (!h || !out || i < 0 || i >= host_specs(h)) return fail("conv_desc_get: bad index %d", i);
    const ConvSpec& s = h->specs[i];
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", s.name.c_str());
    snprintf(out->bn, sizeof out->bn, "%s", s.bn.c_str());
    out->cin = s.cin; out->cout = s.cout; out->k = s.k; out->stride = s.stride;
    out->has_bias = s.has_bias; out->relu = s.relu;
    return 0;
}

int esahrnet_set_conv(esahrnet_handle h, int i, const float* w, const float* b) {
    if (!h || !w || !b || i < 0 || i >= host_specs(h)) return fail("set_conv: bad argument (index %d)", i);
    ConvSpec& s = h->specs[i];
    const size_t nw = (size_t)s.cout * s.cin * s.k * s.k;
    s.w.assign(w, w + nw);
    s.b.assign(b, b + s.cout);
    for (size_t j = 0; j < nw; ++j)
        if (!std::isfinite(s.w[j])) return fail("set_conv(%s): non-finite weight", s.name.c_str());
    s.set = true;
    return 0;
}

int esahrnet_aux_count(esahrnet_handle h) { return h ? (int)h->aux.size() : -1; }

int esahrnet_aux_desc_get(esahrnet_handle h, int i, esahrnet_aux_desc* out) {
    if (!h || !out || i < 0 || i >= (int)h->aux.size()) return fail("aux_desc_get: bad index %d", i);
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", h->aux[i].name.c_str());
    for (int k = 0; k < 4; ++k) out->shape[k] = h->aux[i].shape[k];
    return 0;
}

int esahrnet_set_aux(esahrnet_handle h, int i, const float* w) {
    if (!h || !w || i < 0 || i >= (int)h->aux.size()) return fail("set_aux: bad argument (index %d)", i);
    AuxSpec& a = h->aux[i];
    const size_t n = (size_t)a.shape[0] * a.shape[1] * a.shape[2] * a.shape[3];
    a.data.assign(w, w + n);
    for (float v : a.data) if (!std::isfinite(v)) return fail("set_aux(%s): non-finite value", a.name.c_str());
    a.set = true;
    return 0;
}

int esahrnet_commit(esahrnet_handle h) {
    if (!h) return fail("commit: null handle");
    for (ConvSpec& s : h->specs) {
        if (s.parent < 0) continue;
        const ConvSpec& ps = h->specs[s.parent];
        if (!ps.set) return fail("commit: weights of '%s' were never set", ps.name.c_str());
        s.w.assign((size_t)s.cout * s.cin, 0.f);
        s.b.assign(s.cout, 0.f);
        for (int co = 0; co < ps.cout; ++co)
            for (int t = 0; t < 9; ++t)
                for (int ci = 0; ci < s.cin; ++ci)
                    s.w[(size_t)((co >> 3) * 72 + t * 8 + (co & 7)) * s.cin + ci] = ps.w[((size_t)co * ps.cin + s.pc0 + ci) * 9 + t];
        s.set = true;
    }
    for (const ConvSpec& s : h->specs)
        if (!s.set) return fail("commit: weights of '%s' were never set", s.name.c_str());
    for (const AuxSpec& a : h->aux)
        if (!a.set) return fail("commit: tensor '%s' was never set", a.name.c_str());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("commit: no HIP device visible — the MI355X kernels cannot run (no CPU fallback exists)");
    HIP_OK(hipSetDevice(h->device));
    free_weights(*h);
    std::vector<char> packed;
    for (DevConv& d : h->dconvs) {
        const ConvSpec& s = h->specs[d.spec];
        const int cin = d.c1 - d.c0, taps = s.k * s.k;
        std::vector<float> w((size_t)s.cout * cin * taps);
        for (int co = 0; co < s.cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < taps; ++t) {
                    const int src_ci = d.perm.empty() ? d.c0 + ci : d.perm[ci];
                    w[((size_t)co * cin + ci) * taps + t] = s.w[((size_t)co * s.cin + src_ci) * taps + t];
                }
        if (h->bf) {
            packed.assign(esa::packed_weight_bytes_bf(d.coutp, d.cinp, s.k), 0);
            esa::pack_conv_weights_bf(w.data(), s.cout, cin, s.k, d.coutp, d.cinp, packed.data());
        } else if (h->x6()) {
            packed.assign(esa::packed_weight_bytes_x6(d.coutp, d.cinp, s.k), 0);
            esa::pack_conv_weights_x6(w.data(), s.cout, cin, s.k, d.coutp, d.cinp, packed.data());
        } else {
            packed.assign(esa::packed_weight_bytes(d.coutp, d.cinp, s.k), 0);
            esa::pack_conv_weights(w.data(), s.cout, cin, s.k, d.coutp, d.cinp, packed.data());
        }
        if (upload(packed, &d.w)) return 1;
        std::vector<float> bias(d.coutp, 0.f);
        if (d.use_bias) std::copy(s.b.begin(), s.b.end(), bias.begin());
        if (upload(bias, reinterpret_cast<void**>(&d.bias))) return 1;
    }
    for (Multi& m : h->multis) {     // members' packed weights / biases back to back (cout-tile major packing)
        size_t wbytes = 0;
        for (int k = 0; k < m.n; ++k) {
            const DevConv& d = h->dconvs[h->ops[m.op[k]].dconv];
            wbytes += esa::packed_weight_bytes(d.coutp, d.cinp, 3);
        }
        HIP_OK(hipMalloc(&m.w, wbytes));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&m.bias), (size_t)m.coutp * sizeof(float)));
        size_t wo = 0, bo = 0;
        for (int k = 0; k < m.n; ++k) {
            const DevConv& d = h->dconvs[h->ops[m.op[k]].dconv];
            const size_t wb = esa::packed_weight_bytes(d.coutp, d.cinp, 3);
            HIP_OK(hipMemcpy(static_cast<char*>(m.w) + wo, d.w, wb, hipMemcpyDeviceToDevice));
            HIP_OK(hipMemcpy(m.bias + bo, d.bias, (size_t)d.coutp * sizeof(float), hipMemcpyDeviceToDevice));
            wo += wb; bo += d.coutp;
        }
    }
    {   // stem: [cout/8][cin][9][8]
        const ConvSpec& s = h->specs[h->spec_stem];
        const int coutp = h->padc(s.cout);
        std::vector<float> w((size_t)coutp * s.cin * 9, 0.f), b(coutp, 0.f);
        for (int co = 0; co < s.cout; ++co) {
            b[co] = s.b[co];
            for (int ci = 0; ci < s.cin; ++ci)
                for (int t = 0; t < 9; ++t)
                    w[(((size_t)(co >> 3) * s.cin + ci) * 9 + t) * 8 + (co & 7)] = s.w[((size_t)co * s.cin + ci) * 9 + t];
        }
        if (h->x6_stemf) {      // stem_x6_kernel reads conv1 in its own layout (bias inside)
            std::vector<float> wx(2 * 4 * 10 * 8, 0.f);
            esa::pack_stem_w1_x6(s.w.data(), s.b.data(), wx.data());
            w = wx;
        }
        if (upload(w, reinterpret_cast<void**>(&h->stem_w)) || upload(b, reinterpret_cast<void**>(&h->stem_b))) return 1;
    }
    if (h->cfg.variant == 0) {   // final: [K+cin][9][KT]
        const ConvSpec& s = h->specs[h->spec_final];
        const int kt = esa::final_kt(s.cout);
        std::vector<float> w((size_t)s.cin * 9 * kt, 0.f), b(std::max(kt, 32), 0.f);
        if (esa::final_mfma_supported(s.cout, s.cin - s.cout) && !getenv("ESAHRNET_FINAL_VALU") && !h->x6()) {
            packed.assign(esa::final_mfma_bytes(s.cout, s.cin - s.cout), 0);
            esa::pack_final_mfma(s.w.data(), s.cout, s.cin - s.cout, packed.data());
            if (upload(packed, &h->final_wpk)) return 1;
        }
        for (int co = 0; co < s.cout; ++co) {
            b[co] = s.b[co];
            for (int ci = 0; ci < s.cin; ++ci)
                for (int t = 0; t < 9; ++t)
                    w[((size_t)ci * 9 + t) * kt + co] = s.w[((size_t)co * s.cin + ci) * 9 + t];
        }
        if (upload(w, reinterpret_cast<void**>(&h->final_w)) || upload(b, reinterpret_cast<void**>(&h->final_b))) return 1;
    }
    for (AuxSpec& a : h->aux)
        if (upload(a.data, reinterpret_cast<void**>(&a.dev))) return 1;
    for (const Op& o : h->ops)
        if (o.kind == OP_STEMRAW) {   // un-normalised conv1 in the stem kernel's [cout/8][cin][9][8] layout, zero bias
            const AuxSpec& a = h->aux[o.aux[0]];
            const int cout = a.shape[0], cin = a.shape[1], coutp = pad32(cout);
            std::vector<float> w((size_t)coutp * cin * 9, 0.f), b(coutp, 0.f);
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int t = 0; t < 9; ++t)
                        w[(((size_t)(co >> 3) * cin + ci) * 9 + t) * 8 + (co & 7)] = a.data[((size_t)co * cin + ci) * 9 + t];
            if (upload(w, reinterpret_cast<void**>(&h->stemraw_w)) || upload(b, reinterpret_cast<void**>(&h->stemraw_b))) return 1;
        }
    if (h->spec_l0 >= 0) {   // fused head: W0 slice (standard pack), W3 (permuted-K pack), biases
        const ConvSpec& s0 = h->specs[h->spec_l0];
        const ConvSpec& s3 = h->specs[h->spec_l3];
        const int ct = s0.cout, ctp = h->padc(ct), c0 = h->head_c0, c0p = h->padc(c0), c3p = pad32(s3.cout);
        std::vector<float> w((size_t)ct * c0);
        for (int co = 0; co < ct; ++co)
            for (int ci = 0; ci < c0; ++ci) w[(size_t)co * c0 + ci] = s0.w[(size_t)co * s0.cin + ci];
        if (h->bf) {
            packed.assign(esa::packed_weight_bytes_bf(ctp, c0p, 1), 0);
            esa::pack_conv_weights_bf(w.data(), ct, c0, 1, ctp, c0p, packed.data());
            if (upload(packed, &h->head_w0)) return 1;
            packed.assign(esa::head_w3_bf_bytes(s3.cout, ctp), 0);
            esa::pack_head_w3_bf(s3.w.data(), s3.cout, ct, ctp, packed.data());
            if (upload(packed, &h->head_w3)) return 1;
        } else if (h->x6()) {      // head_x6.hip: both in conv_x6's three-term fragments (its K order is the h0 fragment's)
            packed.assign(esa::packed_weight_bytes_x6(ctp, c0p, 1), 0);
            esa::pack_conv_weights_x6(w.data(), ct, c0, 1, ctp, c0p, packed.data());
            if (upload(packed, &h->head_w0)) return 1;
            const int m3p = s3.cout <= 16 ? 16 : 32;
            packed.assign(esa::packed_weight_bytes_x6(m3p, ctp, 1), 0);
            esa::pack_conv_weights_x6(s3.w.data(), s3.cout, ct, 1, m3p, ctp, packed.data());
            if (upload(packed, &h->head_w3)) return 1;
        } else {
            packed.assign(esa::packed_weight_bytes(ctp, c0p, 1), 0);
            esa::pack_conv_weights(w.data(), ct, c0, 1, ctp, c0p, packed.data());
            if (upload(packed, &h->head_w0)) return 1;
            packed.assign(esa::head_w3_bytes(s3.cout, ctp), 0);
            esa::pack_head_w3(s3.w.data(), s3.cout, ct, ctp, packed.data());
            if (upload(packed, &h->head_w3)) return 1;
        }
        std::vector<float> b0(ctp, 0.f), b3(c3p, 0.f);
        std::copy(s0.b.begin(), s0.b.end(), b0.begin());
        std::copy(s3.b.begin(), s3.b.end(), b3.begin());
        if (upload(b0, reinterpret_cast<void**>(&h->head_b0)) || upload(b3, reinterpret_cast<void**>(&h->head_b3))) return 1;
    }
    if (h->nlanes > 1) {     // side streams + events for the wave executor
        for (int i = 0; i < 3; ++i) HIP_OK(hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking));
        h->wave_entry.assign(h->nwaves, nullptr);
        h->wave_end.assign((h->ops.size() + 1) * 3, nullptr);
        for (hipEvent_t& e : h->wave_entry) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (size_t k = 0; k <= h->ops.size(); ++k)
            for (int l = 1; l < 4; ++l)
                if (k == h->ops.size() || (h->ops[k].join >> l & 1))
                    HIP_OK(hipEventCreateWithFlags(&h->wave_end[k * 3 + l - 1], hipEventDisableTiming));
        h->op_event.assign(h->ops.size(), nullptr);
        for (size_t k = 0; k < h->ops.size(); ++k)
            if (h->ops[k].record) HIP_OK(hipEventCreateWithFlags(&h->op_event[k], hipEventDisableTiming));
    }
    h->committed = true;
    return 0;
}

int esahrnet_set_debug_keep(esahrnet_handle h, int keep) {
    if (!h) return fail("null handle");
    h->keep = keep != 0;
    return 0;
}

int esahrnet_workspace_bytes(esahrnet_handle h, int n, int height, int width, size_t* bytes) {
    if (!h || !bytes) return fail("workspace_bytes: null argument");
    if (plan_shape(*h, n, height, width)) return 1;
    *bytes = h->sp.bytes;
    return 0;
}

// CBAM's per-pixel maps and their 7x7 attention in one kernel (cbam.hip: cbam_spatial) where the channel-group count allows
// and the image has at least 32 of its 16 x 32 tiles (measured, batch 32: 128x128x32ch 69 -> 47 us, 256x256x64ch 448 -> 323 us,
// but 64x64 and smaller lose: too few workgroups, each walking a halo that is mostly padding).  The batch size is
// deliberately not part of the rule: the kernel serving a layer must not depend on it.
// slabs per image the stem kernel pools the raw seg_hrnet3 skip tensor into; 0: pool_partial does it (plan without the
// arrangement, or a crop whose rows make more slabs than the partials tensor reserves)
static int stem_pools(const esahrnet_ctx& c, int height, int width) {
    if (c.stemraw_partial < 0) return 0;
    const int slabs = esa::stem_pool_slabs(c.cfg.cin, pad32(c.cfg.stem_width), height, width);
    return slabs <= STEM_POOL_SLABS_MAX ? slabs : 0;
}

// (ESAHRNET_CBAM_UNFUSED is read once, at esahrnet_create, like the other plan switches — not per launch)
static bool cbam_fused(const esahrnet_ctx& c, int Cp, int hh, int ww) {
    return esa::cbam_spatial_supported(Cp) && ((hh + 15) / 16) * ((ww + 31) / 32) >= 32 && !c.cbam_unfused;
}

static int run_forward(esahrnet_handle h, const void* x_dev, int n, int height, int width,
                       void* heat_dev, void* ws_dev, size_t ws_bytes, esahrnet_stream stream_,
                       hipEvent_t* events, void* part_dev = nullptr) {
    if (!h || !x_dev || !heat_dev || !ws_dev) return fail("forward: null argument");
    if (!h->committed) return fail("forward: esahrnet_commit has not been called");
    if (plan_shape(*h, n, height, width)) return 1;
    if (ws_bytes < h->sp.bytes) return fail("forward: workspace too small (%zu < %zu)", ws_bytes, h->sp.bytes);
    if (reinterpret_cast<uintptr_t>(ws_dev) & 255) return fail("forward: workspace must be 256-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* ws = static_cast<char*>(ws_dev);
    const ShapePlan& sp = h->sp;
    auto T = [&](int t) { return ws + h->tensors[t].off; };
    int op_index = 0;
    if (events && hipEventRecord(events[0], stream) != hipSuccess) return fail("forward: hipEventRecord failed");
    // wave executor: lane 0 is the caller's stream; the other lanes of a wave are side streams that fork from and join
    // into it (also valid under stream capture: the graph gets parallel branches).  The timed and the
    // keep-intermediates modes stay on one stream.
    const bool multi = !events && !h->keep && h->nlanes > 1 && h->side[0] != nullptr;
    const hipStream_t caller = stream;
    bool lane_used[4] = {true, false, false, false};
    int cur_wave = -1;
    auto join = [&](size_t slot, unsigned lanes) -> int {      // the caller's stream continues behind those side lanes
        for (int l = 1; l < 4; ++l)
            if ((lanes >> l & 1) && lane_used[l]) {
                hipEvent_t e = h->wave_end[slot * 3 + l - 1];
                if (hipEventRecord(e, h->side[l - 1]) != hipSuccess || hipStreamWaitEvent(caller, e, 0) != hipSuccess) return 1;
                lane_used[l] = false;
            }
        return 0;
    };
    const int active_alt = sp.head2 ? 2 : 1;
    // one CBAM launch as a job description (cbam.hip); kind < 0: nothing to launch (maps formed inside cbam_spatial)
    auto cbam_job = [&](const Op& o) {
        esa::CbamJob q{};
        q.kind = -1;
        const Tensor& tx = h->tensors[o.kind == OP_MLP ? o.terms[0] : o.in];
        const int hh = sp.lh[tx.level], ww = sp.lw[tx.level];
        q.ap.N = n; q.ap.H = hh; q.ap.W = ww; q.ap.Cp = tx.Cp; q.ap.C = o.nchan; q.ap.fmt = h->fmt;
        q.HW = hh * ww; q.P = std::min(Builder::POOL_SLABS, q.HW); q.Cr = o.nchan / 16;
        switch (o.kind) {
            case OP_POOL: q.kind = esa::CBAM_POOL; q.ap.x = T(o.in); q.partial = reinterpret_cast<float*>(T(o.out)); q.ap.C = tx.C; break;
            case OP_MLP:
                q.kind = esa::CBAM_MLP; q.partial = reinterpret_cast<float*>(T(o.in)); q.ca = reinterpret_cast<float*>(T(o.out));
                q.w0 = h->aux[o.aux[0]].dev; q.w2 = h->aux[o.aux[1]].dev;
                break;
            case OP_MAPS:
                if (cbam_fused(*h, tx.Cp, hh, ww)) break;
                q.kind = esa::CBAM_MAPS; q.ap.x = T(o.in); q.ap.ca = reinterpret_cast<const float*>(T(o.terms[1]));
                q.maps = reinterpret_cast<float*>(T(o.out));
                break;
            default: {
                const Tensor& to = h->tensors[o.out];
                q.kind = cbam_fused(*h, tx.Cp, hh, ww) ? esa::CBAM_SPATIAL : esa::CBAM_APPLY;
                q.ap.x = T(o.in); q.ap.res = o.res >= 0 ? T(o.res) : nullptr;
                q.ap.ca = reinterpret_cast<const float*>(T(o.terms[1])); q.ap.maps = reinterpret_cast<const float*>(T(o.terms[2]));
                q.ap.w_sa = h->aux[o.aux[2]].dev; q.ap.y = T(o.out);
                q.ap.y_pix_bytes = to.Cp * 4; q.ap.y_c0 = o.c0; q.ap.relu = o.relu;
            }
        }
        return q;
    };
    for (const Op& o : h->ops) {
        int rc = 0;
        if (multi) {
            if (o.join && join((size_t)op_index, o.join)) return fail("forward: joining side lanes failed");
            if (o.wave != cur_wave) {
                cur_wave = o.wave;
                if (h->wave_mask[cur_wave] & ~1u) HIP_OK(hipEventRecord(h->wave_entry[cur_wave], caller));
            }
            stream = o.lane == 0 ? caller : h->side[o.lane - 1];
            lane_used[o.lane] = true;
            if (o.wait_entry) HIP_OK(hipStreamWaitEvent(stream, h->wave_entry[cur_wave], 0));
            if (o.wait0 >= 0) HIP_OK(hipStreamWaitEvent(stream, h->op_event[o.wait0], 0));
        }
        if (o.alt != 0 && o.alt != active_alt) {         // the head alternative not used at this shape
            if (multi && o.record) HIP_OK(hipEventRecord(h->op_event[op_index], caller));
            ++op_index;
            if (events && hipEventRecord(events[op_index], stream) != hipSuccess) return fail("forward: hipEventRecord failed");
            continue;
        }
        switch (o.kind) {
            case OP_STEM: {
                const Tensor& t = h->tensors[o.out];
                esa::StemParams p{static_cast<const float*>(x_dev), T(o.out), h->stem_w, h->stem_b,
                                  n, height, width, h->cfg.cin, t.Cp, 1, h->fmt};
                rc = esa::launch_stem(p, stream);
                break;
            }
            case OP_STEMRAW: {
                const Tensor& t = h->tensors[o.out];
                esa::StemParams p{static_cast<const float*>(x_dev), T(o.out), h->stemraw_w, h->stemraw_b,
                                  n, height, width, h->cfg.cin, t.Cp, 0, h->fmt};
                rc = stem_pools(*h, height, width) ? esa::launch_stem_pool(p, reinterpret_cast<float*>(T(o.out2)), stream)
                                                   : esa::launch_stem(p, stream);
                break;
            }
            case OP_POOL: case OP_MLP: case OP_MAPS: case OP_APPLY:
                if (o.job >= 0 && sp.job_on[o.job]) {           // the same step of every branch of the module in one launch
                    if (o.jpos > 0) break;
                    const JobGroup& g = h->jobs[o.job];
                    esa::CbamJob js[esa::CBAM_MAXJOBS];
                    int nj = 0;
                    for (int k = 0; k < g.n; ++k) {
                        const esa::CbamJob q = cbam_job(h->ops[g.op[k]]);
                        if (q.kind >= 0) js[nj++] = q;
                    }
                    if (nj) rc = esa::launch_cbam_jobs(js, nj, stream);
                    break;
                }
                switch (o.kind) {
            case OP_POOL: {
                const Tensor& ti = h->tensors[o.in];
                if (o.out == h->stemraw_partial && stem_pools(*h, height, width)) break;       // made by the stem kernel
                const int HW = sp.lh[ti.level] * sp.lw[ti.level];
                rc = esa::launch_pool_partial(T(o.in), reinterpret_cast<float*>(T(o.out)), n, HW, ti.Cp,
                                              std::min(Builder::POOL_SLABS, HW), stream, h->fmt);
                break;
            }
            case OP_MLP: {
                const Tensor& tx = h->tensors[o.terms[0]];
                const int HW = sp.lh[tx.level] * sp.lw[tx.level];
                const int slabs = o.in == h->stemraw_partial ? stem_pools(*h, height, width) : 0;
                rc = esa::launch_ca_mlp(reinterpret_cast<const float*>(T(o.in)), h->aux[o.aux[0]].dev, h->aux[o.aux[1]].dev,
                                        reinterpret_cast<float*>(T(o.out)), n, HW, o.nchan, tx.Cp, o.nchan / 16,
                                        slabs ? slabs : std::min(Builder::POOL_SLABS, HW), stream);
                break;
            }
            case OP_MAPS: {
                const Tensor& ti = h->tensors[o.in];
                if (cbam_fused(*h, ti.Cp, sp.lh[ti.level], sp.lw[ti.level])) break;      // formed inside cbam_spatial (the OP_APPLY that follows)
                rc = esa::launch_cbam_maps(T(o.in), reinterpret_cast<const float*>(T(o.terms[1])),
                                           reinterpret_cast<float*>(T(o.out)), n, sp.lh[ti.level] * sp.lw[ti.level],
                                           o.nchan, ti.Cp, stream, h->fmt);
                break;
            }
            case OP_APPLY: {
                const Tensor& ti = h->tensors[o.in];
                const Tensor& to = h->tensors[o.out];
                esa::CbamApplyParams p{};
                p.x = T(o.in); p.res = o.res >= 0 ? T(o.res) : nullptr;
                p.ca = reinterpret_cast<const float*>(T(o.terms[1])); p.maps = reinterpret_cast<const float*>(T(o.terms[2]));
                p.w_sa = h->aux[o.aux[2]].dev; p.y = T(o.out);
                p.N = n; p.H = sp.lh[ti.level]; p.W = sp.lw[ti.level]; p.Cp = ti.Cp;
                p.y_pix_bytes = to.Cp * 4; p.y_c0 = o.c0; p.relu = o.relu; p.C = o.nchan; p.fmt = h->fmt;
                rc = cbam_fused(*h, ti.Cp, p.H, p.W) ? esa::launch_cbam_spatial(p, stream) : esa::launch_cbam_apply(p, stream);
                break;
            }
            default: break;
                }                   // (inner switch: the single-tensor launches)
                break;
            case OP_RESAMPLE: {
                const Tensor& ti = h->tensors[o.in];
                const Tensor& to = h->tensors[o.out];
                esa::ResampleParams p{};
                p.x = T(o.in); p.y = T(o.out); p.N = n;
                p.h = sp.lh[ti.level]; p.w = sp.lw[ti.level]; p.H = sp.lh[to.level]; p.W = sp.lw[to.level];
                p.C = o.nchan; p.Cp_src = ti.Cp; p.y_pix_bytes = to.Cp * 4; p.y_c0 = o.c0; p.align = o.align; p.fmt = h->fmt;
                rc = esa::launch_resample_slice(p, stream);
                break;
            }
            case OP_GATHER: {
                const Tensor& to = h->tensors[o.out];
                esa::GatherParams p{};
                const int zt[2] = {o.in, o.terms[0]};
                for (int b = 0; b < 2; ++b) {
                    const Tensor& tz = h->tensors[zt[b]];
                    p.z[b] = T(zt[b]); p.h[b] = sp.lh[tz.level]; p.w[b] = sp.lw[tz.level]; p.zpix[b] = tz.Cp * 4;
                }
                p.y = T(o.out); p.N = n; p.H = sp.lh[to.level]; p.W = sp.lw[to.level]; p.C = to.C; p.Cp = to.Cp;
                rc = esa::launch_head_gather(p, stream, h->fmt);
                break;
            }
            case OP_ZERO: {
                const Tensor& to = h->tensors[o.out];
                rc = esa::launch_zero_slice(T(o.out), (long long)n * sp.lh[to.level] * sp.lw[to.level], to.Cp * 4,
                                            o.c0, o.nchan, stream);
                break;
            }
            case OP_TONCHW: {
                const Tensor& ti = h->tensors[o.in];
                rc = esa::launch_fmt_to_nchw(h->fmt, T(o.in), n, h->cfg.num_keypoints, height, width, ti.Cp,
                                             static_cast<float*>(heat_dev), stream);
                break;
            }
            case OP_STEMF: {
                const DevConv& d = h->dconvs[o.dconv];
                const Tensor& to = h->tensors[o.out];
                esa::StemFusedParams p{};
                p.x = static_cast<const float*>(x_dev); p.y = T(o.out);
                p.w1 = h->stem_w; p.bias1 = h->stem_b;
                p.w2 = static_cast<const uint4*>(d.w); p.bias2 = d.bias;
                p.N = n; p.H = height; p.W = width; p.OH = sp.lh[to.level]; p.OW = sp.lw[to.level];
                p.cin = h->cfg.cin; p.Cmid = d.cinp; p.Coutp = d.coutp;
                rc = h->x6() ? esa::launch_stem_fused_x6(p, stream) : esa::launch_stem_fused(p, stream);
                break;
            }
            case OP_CONV: {
                const DevConv& d = h->dconvs[o.dconv];
                const ConvSpec& s = h->specs[d.spec];
                const Tensor& ti = h->tensors[o.in];
                const Tensor& to = h->tensors[o.out];
                if (ti.Cp != d.cinp || to.Cp != d.coutp) return fail("plan bug: channel mismatch at %s", s.name.c_str());
                if (o.multi >= 0 && sp.multi_on[o.multi]) {
                    if (o.mpos > 0) break;                      // evaluated by the group's leader
                    const Multi& m = h->multis[o.multi];
                    esa::ConvParams p{};
                    p.x = T(o.in);
                    p.w = static_cast<const uint4*>(m.w); p.bias = m.bias;
                    p.N = n; p.H = sp.lh[ti.level]; p.W = sp.lw[ti.level];
                    p.OH = sp.lh[to.level]; p.OW = sp.lw[to.level];
                    p.Cinp = d.cinp; p.Coutp = m.coutp; p.nheads = m.n;
                    for (int k = 0; k < m.n; ++k) {
                        const Op& ok = h->ops[m.op[k]];
                        p.yh[k] = T(ok.out);
                        p.hb[k + 1] = p.hb[k] + h->dconvs[ok.dconv].coutp;
                        p.hrelu[k] = ok.relu;
                    }
                    rc = esa::launch_conv_s2c32_multi(p, stream);
                    break;
                }
                if (o.job >= 0 && sp.job_on[o.job]) {
                    if (o.jpos > 0) break;                      // evaluated by the group's leader
                    const JobGroup& g = h->jobs[o.job];
                    esa::ConvParams ps[6];
                    for (int k = 0; k < g.n; ++k) ps[k] = conv_params_of(*h, h->ops[g.op[k]], n, sp.lh, sp.lw, ws);
                    rc = h->x6() ? esa::launch_conv_x6_jobs(ps, g.n, s.k, s.stride, stream)
                       : s.k == 1 ? esa::launch_conv1x1_jobs(ps, g.n, stream) : esa::launch_conv_jobs(ps, g.n, s.stride, stream);
                    break;
                }
                const esa::ConvParams p = conv_params_of(*h, o, n, sp.lh, sp.lw, ws);
                rc = esa::launch_conv(p, s.k, s.stride, stream);
                break;
            }
            case OP_BLOCK: {
                const DevConv& d1 = h->dconvs[o.dconv];
                const DevConv& d2 = h->dconvs[o.dconv2];
                const Tensor& ti = h->tensors[o.in];
                if (ti.Cp != 32 || h->tensors[o.out].Cp != 32) return fail("plan bug: bblock32 on a non-32-channel tensor");
                esa::BlockParams p{};
                p.x = T(o.in); p.y = T(o.out);
                p.w1 = static_cast<const uint4*>(d1.w); p.w2 = static_cast<const uint4*>(d2.w);
                p.bias1 = d1.bias; p.bias2 = d2.bias;
                p.N = n; p.H = sp.lh[ti.level]; p.W = sp.lw[ti.level];
                rc = esa::launch_bblock32(p, stream);
                break;
            }
            case OP_HEAD: {
                const Tensor& ti = h->tensors[o.in];
                const Tensor& to = h->tensors[o.out];
                esa::HeadParams p{};
                p.x0 = T(o.in); p.y = T(o.out);
                p.w0 = static_cast<const uint4*>(h->head_w0); p.w3 = static_cast<const uint4*>(h->head_w3);
                p.bias0 = h->head_b0; p.bias3 = h->head_b3;
                p.N = n; p.H = sp.lh[ti.level]; p.W = sp.lw[ti.level];
                for (int i = 0; i < 3; ++i) {
                    const Tensor& tt = h->tensors[o.terms[i]];
                    p.t[i] = T(o.terms[i]); p.th[i] = sp.lh[tt.level]; p.tw[i] = sp.lw[tt.level];
                    p.Ctp = tt.Cp;
                }
                p.C0p = ti.Cp; p.C3p = to.Cp; p.K = h->cfg.num_keypoints;
                if (!esa::head_fused_supported(p.H, p.W, p.th, p.tw, p.C0p, p.K))
                    return fail("forward: fused head does not support this shape (%dx%d)", p.H, p.W);
                rc = esa::launch_head(p, stream);
                break;
            }
            case OP_HEADBF: {
                const Tensor& ti = h->tensors[o.in];
                const Tensor& to = h->tensors[o.out];
                esa::HeadParams p{};
                p.x0 = T(o.in); p.y = T(o.out);
                p.w0 = static_cast<const uint4*>(h->head_w0); p.w3 = static_cast<const uint4*>(h->head_w3);
                p.bias0 = h->head_b0; p.bias3 = h->head_b3;
                p.N = n; p.H = sp.lh[ti.level]; p.W = sp.lw[ti.level];
                for (int i = 0; i < 3; ++i) {
                    const Tensor& tt = h->tensors[o.terms[i]];
                    p.t[i] = T(o.terms[i]); p.th[i] = sp.lh[tt.level]; p.tw[i] = sp.lw[tt.level];
                    p.Ctp = tt.Cp;
                }
                p.C0p = ti.Cp; p.C3p = to.Cp; p.K = h->cfg.num_keypoints;
                rc = h->x6() ? esa::launch_head_x6(p, stream) : esa::launch_head_bf(p, stream);
                break;
            }
            case OP_HEADT: {
                const Tensor& ti = h->tensors[o.in];
                const Tensor& to = h->tensors[o.out];
                const DevConv& d = h->dconvs[o.dconv];
                esa::HeadTParams p{};
                p.x = T(o.in); p.t = T(o.out); p.wt = static_cast<const uint4*>(d.w);
                p.N = n; p.h = sp.lh[ti.level]; p.w = sp.lw[ti.level];
                p.Cinp = ti.Cp; p.Ctp = to.Cp; p.XP = esa::head_t_xp(p.w);
                rc = esa::launch_head_t(p, stream);
                break;
            }
            case OP_HEAD2: {
                const Tensor& ti = h->tensors[o.in];
                const Tensor& to = h->tensors[o.out];
                esa::Head2Params p{};
                p.x0 = T(o.in); p.x1 = T(o.terms[0]); p.t2 = T(o.terms[1]); p.t3 = T(o.terms[2]); p.y = T(o.out);
                p.w0 = static_cast<const uint4*>(h->head_w0); p.w3 = static_cast<const uint4*>(h->head_w3);
                p.w1 = static_cast<const uint4*>(h->dconvs[o.dconv].w);
                p.bias0 = h->head_b0; p.bias3 = h->head_b3;
                p.N = n; p.H = sp.lh[ti.level]; p.W = sp.lw[ti.level];
                for (int i = 0; i < 3; ++i) {
                    const Tensor& tt = h->tensors[o.terms[i]];
                    p.th[i] = sp.lh[tt.level]; p.tw[i] = sp.lw[tt.level];
                }
                p.xp2 = esa::head_t_xp(p.tw[1]); p.xp3 = esa::head_t_xp(p.tw[2]);
                p.C0p = ti.Cp; p.C1p = h->tensors[o.terms[0]].Cp; p.Ctp = h->tensors[o.terms[1]].Cp;
                p.C3p = to.Cp; p.K = h->cfg.num_keypoints;
                rc = esa::launch_head2(p, sp.head2_ulo, stream);
                break;
            }
            case OP_FUSE: {
                const Tensor& to = h->tensors[o.out];
                esa::FuseParams p{};
                p.nterms = o.nterms;
                for (int i = 0; i < o.nterms; ++i) {
                    const Tensor& ti = h->tensors[o.terms[i]];
                    if (ti.Cp != to.Cp) return fail("plan bug: fuse channel mismatch");
                    p.x[i] = T(o.terms[i]); p.h[i] = sp.lh[ti.level]; p.w[i] = sp.lw[ti.level];
                }
                p.y = T(o.out); p.N = n; p.H = sp.lh[to.level]; p.W = sp.lw[to.level]; p.Cp = to.Cp;
                p.relu = o.relu; p.fmt = h->fmt;
                rc = esa::launch_fuse(p, stream);
                break;
            }
            case OP_FINAL: {
                const Tensor& ti = h->tensors[o.in];
                esa::FinalParams p{};
                p.h3 = T(o.in); p.x0 = static_cast<const float*>(x_dev); p.out = static_cast<float*>(heat_dev);
                p.w = h->final_w; p.bias = h->final_b; p.wpk = static_cast<const uint4*>(h->final_wpk);
                p.N = n; p.H = height; p.W = width; p.h = sp.lh[ti.level]; p.wd = sp.lw[ti.level];
                p.K = h->cfg.num_keypoints; p.cin = h->cfg.cin; p.Cp = ti.Cp; p.fmt = h->fmt;
                p.part = static_cast<float2*>(part_dev);
                rc = esa::launch_final(p, stream);
                break;
            }
        }
        if (rc) return fail("forward: kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        if (multi && o.record) HIP_OK(hipEventRecord(h->op_event[op_index], caller));
        ++op_index;
        if (events && hipEventRecord(events[op_index], stream) != hipSuccess) return fail("forward: hipEventRecord failed");
    }
    if (multi && join(h->ops.size(), 0xeu)) return fail("forward: joining the side lanes failed");
    return 0;
}

int esahrnet_forward(esahrnet_handle h, const void* x_dev, int n, int height, int width,
                     void* heat_dev, void* ws_dev, size_t ws_bytes, esahrnet_stream stream) {
    return run_forward(h, x_dev, n, height, width, heat_dev, ws_dev, ws_bytes, stream, nullptr);
}

int esahrnet_forward_timed(esahrnet_handle h, const void* x_dev, int n, int height, int width,
                           void* heat_dev, void* ws_dev, size_t ws_bytes, esahrnet_stream stream,
                           float* ms_out) {
    if (!h || !ms_out) return fail("forward_timed: null argument");
    const size_t nops = h->ops.size();
    std::vector<hipEvent_t> ev(nops + 1, nullptr);
    hipEvent_t nul[2] = {nullptr, nullptr};
    int rc = 0;
    for (size_t i = 0; i <= nops && !rc; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) rc = fail("forward_timed: hipEventCreate failed");
    for (int i = 0; i < 2 && !rc; ++i)
        if (hipEventCreate(&nul[i]) != hipSuccess) rc = fail("forward_timed: hipEventCreate failed");
    // the cost of an empty event bracket on this stream is measured and subtracted, so that the
    // per-launch figures are kernel time (comparable with rocprofv3 --kernel-trace), not kernel + marker
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!rc && (hipEventRecord(nul[0], st) != hipSuccess || hipEventRecord(nul[1], st) != hipSuccess))
        rc = fail("forward_timed: hipEventRecord failed");
    if (!rc) rc = run_forward(h, x_dev, n, height, width, heat_dev, ws_dev, ws_bytes, stream, ev.data());
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = fail("forward_timed: stream sync failed");
    float null_ms = 0.f;
    if (!rc && hipEventElapsedTime(&null_ms, nul[0], nul[1]) != hipSuccess) rc = fail("forward_timed: hipEventElapsedTime failed");
    for (size_t i = 0; i < nops && !rc; ++i) {
        if (hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]) != hipSuccess) rc = fail("forward_timed: hipEventElapsedTime failed");
        ms_out[i] = ms_out[i] > null_ms ? ms_out[i] - null_ms : 0.f;
    }
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : nul) if (e) (void)hipEventDestroy(e);
    return rc;
}

int esahrnet_op_desc_get(esahrnet_handle h, int index, int n, int height, int width, esahrnet_op_desc* out) {
    if (!h || !out || index < 0 || index >= (int)h->ops.size()) return fail("op_desc_get: bad argument");
    if (check_shape(*h, n, height, width)) return 1;
    std::vector<int> lh, lw;
    level_dims(*h, height, width, lh, lw);
    memset(out, 0, sizeof *out);
    const Op& o = h->ops[index];
    if (o.alt != 0) {        // a head alternative: described only for the shapes that run it (else kernel = "")
        bool ulo = false;
        if (o.alt != (head2_for_shape(*h, lh, lw, &ulo) ? 2 : 1)) {
            snprintf(out->label, sizeof out->label, "(not used at this shape)");
            return 0;
        }
    }
    auto tbytes = [&](int t) {
        const Tensor& x = h->tensors[t];
        const double wpix = x.tlayout ? (double)esa::head_t_xp(lw[x.level]) : (double)lw[x.level];
        return x.flat ? (double)n * x.flat * 4.0 : (double)n * lh[x.level] * wpix * x.Cp * (double)h->eb();
    };
    switch (o.kind) {
        case OP_STEM: {
            const ConvSpec& s = h->specs[h->spec_stem];
            snprintf(out->kernel, sizeof out->kernel, "stem_kernel");
            snprintf(out->label, sizeof out->label, "%s", s.name.c_str());
            out->flops = 2.0 * n * height * width * s.cout * s.cin * 9;
            out->bytes = (double)n * height * width * s.cin * 4 + tbytes(o.out);
            break;
        }
        case OP_STEMF: {
            const ConvSpec& s1 = h->specs[h->spec_stem];
            const ConvSpec& s2 = h->specs[h->dconvs[o.dconv].spec];
            const Tensor& to = h->tensors[o.out];
            snprintf(out->kernel, sizeof out->kernel, h->x6() ? "stem_x6_kernel" : "stem_fused");
            snprintf(out->label, sizeof out->label, "conv1 + conv2");
            out->flops = 2.0 * n * height * width * s1.cout * s1.cin * 9 +
                         2.0 * n * lh[to.level] * lw[to.level] * s2.cout * s2.cin * 9;
            out->bytes = (double)n * height * width * s1.cin * 4 + tbytes(o.out) +
                         (double)h->wbytes(pad32(s2.cout), pad32(s2.cin), 3);
            break;
        }
        case OP_CONV: {
            const DevConv& d = h->dconvs[o.dconv];
            const ConvSpec& s = h->specs[d.spec];
            const Tensor& to = h->tensors[o.out];
            if (o.multi >= 0 && multi_on_for(*h, h->multis[o.multi], n, lh, lw)) {
                const Multi& m = h->multis[o.multi];
                if (o.mpos > 0) {        // no launch of its own
                    snprintf(out->label, sizeof out->label, "%s (in the multi-head launch above)", s.name.c_str());
                    break;
                }
                snprintf(out->kernel, sizeof out->kernel, m.coutp % 64 == 0 ? "conv_s2c32_kernel<2, 4, 4, true>" : "conv_s2c32_kernel<2, 4, 2, true>");
                std::string lab;
                out->bytes = tbytes(o.in);
                for (int k = 0; k < m.n; ++k) {
                    const Op& ok = h->ops[m.op[k]];
                    const DevConv& dk = h->dconvs[ok.dconv];
                    const ConvSpec& sk = h->specs[dk.spec];
                    const Tensor& tk = h->tensors[ok.out];
                    lab += (k ? " + " : "") + sk.name;
                    out->flops += 2.0 * n * lh[tk.level] * lw[tk.level] * sk.cout * sk.cin * 9.0;
                    out->bytes += tbytes(ok.out) + (double)esa::packed_weight_bytes(dk.coutp, dk.cinp, 3);
                }
                snprintf(out->label, sizeof out->label, "%s", lab.c_str());
                break;
            }
            if (o.job >= 0 && job_on_for(*h, h->jobs[o.job], n, lh, lw)) {
                const JobGroup& g = h->jobs[o.job];
                if (o.jpos > 0) {        // no launch of its own
                    snprintf(out->label, sizeof out->label, "%s (in the merged launch above)", s.name.c_str());
                    break;
                }
                if (h->x6()) {
                    esa::ConvParams qs[6];
                    for (int k = 0; k < g.n; ++k) qs[k] = conv_params_of(*h, h->ops[g.op[k]], n, lh, lw, nullptr);
                    if (s.k == 1 && esa::conv1x1_x6_jobs_supported(qs, g.n)) snprintf(out->kernel, sizeof out->kernel, "conv1x1_x6_jobs_kernel");
                    else snprintf(out->kernel, sizeof out->kernel, "conv_x6_jobs_kernel<%d, %d>", s.k, s.stride);
                }
                else if (s.k == 1) snprintf(out->kernel, sizeof out->kernel, "conv1x1_jobs_kernel");
                else snprintf(out->kernel, sizeof out->kernel, "conv_s2c32_jobs_kernel<%d, %d, %s>", s.stride, s.stride == 1 ? 8 : 4, h->bf ? "true" : "false");
                std::string lab;
                for (int k = 0; k < g.n; ++k) {
                    const Op& ok = h->ops[g.op[k]];
                    const DevConv& dk = h->dconvs[ok.dconv];
                    const ConvSpec& sk = h->specs[dk.spec];
                    const Tensor& tk = h->tensors[ok.out];
                    lab += (k ? " + " : "") + sk.name;
                    out->flops += 2.0 * n * lh[tk.level] * lw[tk.level] * sk.cout * sk.cin * (double)(sk.k * sk.k);
                    out->bytes += tbytes(ok.in) + tbytes(ok.out) + (ok.res >= 0 ? tbytes(ok.res) : 0.0) +
                                  (double)h->wbytes(dk.coutp, dk.cinp, sk.k);
                }
                snprintf(out->label, sizeof out->label, "%s", lab.c_str());
                break;
            }
            {
                const Tensor& ti = h->tensors[o.in];
                esa::ConvParams q{};
                q.N = n; q.H = lh[ti.level]; q.W = lw[ti.level]; q.OH = lh[to.level]; q.OW = lw[to.level];
                q.Cinp = d.cinp; q.Coutp = d.coutp; q.out_f32 = d.out_f32 && !h->x6(); q.fmt = h->fmt;
                q.res = o.res >= 0 ? reinterpret_cast<const char*>(h) : nullptr;     // only tested against nullptr
                snprintf(out->kernel, sizeof out->kernel, "%s", esa::conv_kernel_name(q, s.k, s.stride));
            }
            if (d.c0 != 0 || d.c1 != s.cin) snprintf(out->label, sizeof out->label, "%s[:, %d:%d]", s.name.c_str(), d.c0, d.c1);
            else snprintf(out->label, sizeof out->label, "%s", s.name.c_str());
            out->flops = 2.0 * n * lh[to.level] * lw[to.level] * s.cout * (d.c1 - d.c0) * s.k * s.k;
            out->bytes = tbytes(o.in) + tbytes(o.out) + (o.res >= 0 ? tbytes(o.res) : 0.0) +
                         (double)h->wbytes(d.coutp, d.cinp, s.k);
            break;
        }
        case OP_BLOCK: {
            const ConvSpec& s1 = h->specs[h->dconvs[o.dconv].spec];
            const ConvSpec& s2 = h->specs[h->dconvs[o.dconv2].spec];
            const Tensor& to = h->tensors[o.out];
            snprintf(out->kernel, sizeof out->kernel, "bblock32");
            snprintf(out->label, sizeof out->label, "%s + conv2", s1.name.c_str());
            out->flops = 2.0 * n * lh[to.level] * lw[to.level] * 9.0 * ((double)s1.cout * s1.cin + (double)s2.cout * s2.cin);
            out->bytes = tbytes(o.in) + tbytes(o.out) + 2.0 * (double)esa::packed_weight_bytes(32, 32, 3);
            break;
        }
        case OP_HEAD: {
            const ConvSpec& s0 = h->specs[h->spec_l0];
            const ConvSpec& s3 = h->specs[h->spec_l3];
            const Tensor& to = h->tensors[o.out];
            snprintf(out->kernel, sizeof out->kernel, "head_fused");
            snprintf(out->label, sizeof out->label, "last_layer.0[:, 0:%d] + up + last_layer.3", h->head_c0);
            out->flops = 2.0 * n * lh[to.level] * lw[to.level] * ((double)s0.cout * h->head_c0 + (double)s3.cout * s3.cin);
            out->bytes = tbytes(o.in) + tbytes(o.out);
            for (int i = 0; i < 3; ++i) out->bytes += tbytes(o.terms[i]);
            break;
        }
        case OP_HEADBF: {
            const ConvSpec& s0 = h->specs[h->spec_l0];
            const ConvSpec& s3 = h->specs[h->spec_l3];
            const Tensor& to = h->tensors[o.out];
            snprintf(out->kernel, sizeof out->kernel, h->x6() ? "head_x6" : "head_fused_bf");
            snprintf(out->label, sizeof out->label, "last_layer.0[:, 0:%d] + up + last_layer.3", h->head_c0);
            out->flops = 2.0 * n * lh[to.level] * lw[to.level] * ((double)s0.cout * h->head_c0 + (double)s3.cout * s3.cin);
            out->bytes = tbytes(o.in) + tbytes(o.out);
            for (int i = 0; i < 3; ++i) out->bytes += tbytes(o.terms[i]);
            break;
        }
        case OP_HEADT: {
            const DevConv& d = h->dconvs[o.dconv];
            const ConvSpec& s = h->specs[d.spec];
            const Tensor& ti = h->tensors[o.in];
            snprintf(out->kernel, sizeof out->kernel, "head_t");
            snprintf(out->label, sizeof out->label, "%s[:, %d:%d] (T layout)", s.name.c_str(), d.c0, d.c1);
            out->flops = 2.0 * n * lh[ti.level] * lw[ti.level] * s.cout * (d.c1 - d.c0);
            out->bytes = tbytes(o.in) + tbytes(o.out) + (double)esa::packed_weight_bytes(d.coutp, d.cinp, 1);
            break;
        }
        case OP_HEAD2: {
            const ConvSpec& s0 = h->specs[h->spec_l0];
            const ConvSpec& s3 = h->specs[h->spec_l3];
            const DevConv& d1 = h->dconvs[o.dconv];
            const Tensor& to = h->tensors[o.out];
            const Tensor& t1 = h->tensors[o.terms[0]];
            snprintf(out->kernel, sizeof out->kernel, "head_fused2");
            snprintf(out->label, sizeof out->label, "last_layer.0[:, 0:%d] + up (MFMA) + last_layer.3", d1.c1);
            out->flops = 2.0 * n * lh[to.level] * lw[to.level] * ((double)s0.cout * h->head_c0 + (double)s3.cout * s3.cin) +
                         2.0 * n * lh[t1.level] * lw[t1.level] * (double)s0.cout * (d1.c1 - d1.c0);
            out->bytes = tbytes(o.in) + tbytes(o.out);
            for (int i = 0; i < 3; ++i) out->bytes += tbytes(o.terms[i]);
            break;
        }
        case OP_FUSE: {
            snprintf(out->kernel, sizeof out->kernel, "fuse_kernel");
            snprintf(out->label, sizeof out->label, "fuse -> %s", h->tensors[o.out].tap.c_str());
            out->bytes = tbytes(o.out);
            for (int i = 0; i < o.nterms; ++i) out->bytes += tbytes(o.terms[i]);
            break;
        }
        case OP_STEMRAW: case OP_POOL: case OP_MLP: case OP_MAPS: case OP_APPLY: case OP_RESAMPLE: case OP_ZERO:
        case OP_TONCHW: case OP_GATHER: {
            static const char* names[] = {"stem_kernel(raw)", "pool_partial", "ca_mlp", "cbam_maps", "cbam_apply",
                                          "resample_slice", "zero_slice", "sb_to_nchw", "head_gather"};
            snprintf(out->kernel, sizeof out->kernel, "%s", names[o.kind - OP_STEMRAW]);
            snprintf(out->label, sizeof out->label, "seg_hrnet3");
            if (o.kind == OP_POOL && o.out == h->stemraw_partial && stem_pools(*h, height, width)) {
                out->kernel[0] = 0;
                snprintf(out->label, sizeof out->label, "seg_hrnet3 (inside stem_kernel(raw))");
                break;
            }
            if (o.job >= 0 && (o.kind == OP_POOL || o.kind == OP_MLP || o.kind == OP_MAPS || o.kind == OP_APPLY)) {
                // merged launch of the group (run_forward): issued at the first member, for every member that launches at all
                const JobGroup& g = h->jobs[o.job];
                auto folded = [&](const Op& ok) {
                    return ok.kind == OP_MAPS && cbam_fused(*h, h->tensors[ok.in].Cp, lh[h->tensors[ok.in].level], lw[h->tensors[ok.in].level]);
                };
                int launching = 0;
                for (int k = 0; k < g.n; ++k) launching += folded(h->ops[g.op[k]]) ? 0 : 1;
                if (o.jpos > 0 || launching == 0) {
                    out->kernel[0] = 0;
                    snprintf(out->label, sizeof out->label, "seg_hrnet3 (%s)", launching ? "in the merged launch" : "inside cbam_spatial");
                    break;
                }
                static const char* jn[] = {"cbam_jobs(pool)", "cbam_jobs(mlp)", "cbam_jobs(maps)", "cbam_jobs(apply)"};
                snprintf(out->kernel, sizeof out->kernel, "%s", jn[o.kind - OP_POOL]);
                snprintf(out->label, sizeof out->label, "seg_hrnet3: %d branches", launching);
                for (int k = 0; k < g.n; ++k) {
                    const Op& ok = h->ops[g.op[k]];
                    if (!folded(ok)) out->bytes += (ok.in >= 0 ? tbytes(ok.in) : 0.0) + (ok.out >= 0 ? tbytes(ok.out) : 0.0);
                }
                break;
            }
            if ((o.kind == OP_MAPS || o.kind == OP_APPLY) &&
                cbam_fused(*h, h->tensors[o.in].Cp, lh[h->tensors[o.in].level], lw[h->tensors[o.in].level])) {
                if (o.kind == OP_MAPS) { out->kernel[0] = 0; snprintf(out->label, sizeof out->label, "(inside cbam_spatial)"); break; }
                snprintf(out->kernel, sizeof out->kernel, "cbam_spatial");
            }
            out->bytes += (o.in >= 0 ? tbytes(o.in) : 0.0) + (o.out >= 0 ? tbytes(o.out) : 0.0);
            if (o.kind == OP_GATHER) out->bytes += tbytes(o.terms[0]);
            break;
        }
        case OP_FINAL: {
            const ConvSpec& s = h->specs[h->spec_final];
            snprintf(out->kernel, sizeof out->kernel, "final_kernel");
            snprintf(out->label, sizeof out->label, "%s", s.name.c_str());
            out->flops = 2.0 * n * height * width * s.cout * s.cin * 9;
            out->bytes = tbytes(o.in) + (double)n * height * width * (h->cfg.cin + s.cout) * 4;
            break;
        }
    }
    return 0;
}

int esahrnet_partial_tiles(esahrnet_handle h, int height, int width, int* ntiles) {
    if (!h || !ntiles) return fail("partial_tiles: null argument");
    if (!h->committed) return fail("partial_tiles: esahrnet_commit has not been called");
    *ntiles = 0;
    // only the matrix-core output-layer kernel of the seg_hrnet / seg_hrnet2 plans reports per-tile maxima
    if (h->cfg.variant == 0 && h->final_wpk)
        *ntiles = esa::final_part_tiles(h->cfg.num_keypoints, h->cfg.cin, height, width);
    return 0;
}

int esahrnet_forward_partials(esahrnet_handle h, const void* x_dev, int n, int height, int width, void* heat_dev,
                              void* part_dev, void* ws_dev, size_t ws_bytes, esahrnet_stream stream) {
    int nt = 0;
    if (part_dev) {
        if (esahrnet_partial_tiles(h, height, width, &nt)) return 1;
        if (nt <= 0) return fail("forward_partials: this handle's output layer does not report per-tile maxima (esahrnet_partial_tiles = 0)");
    }
    return run_forward(h, x_dev, n, height, width, heat_dev, ws_dev, ws_bytes, stream, nullptr, part_dev);
}

int esahrnet_keypoints_finish(const void* heat_dev, const void* part_dev, int ntiles, int n, int k, int height, int width,
                              void* kp_dev, void* idx_dev, esahrnet_stream stream) {
    if (!heat_dev || !part_dev || !kp_dev || n <= 0 || k <= 0 || ntiles <= 0) return fail("keypoints_finish: bad argument");
    const int rc = esa::launch_keypoints_finish(static_cast<const float*>(heat_dev), static_cast<const float2*>(part_dev), ntiles,
                                                n * k, height, width, static_cast<float*>(kp_dev), static_cast<int*>(idx_dev),
                                                static_cast<hipStream_t>(stream));
    if (rc) return fail("keypoints_finish: %s", hipGetErrorString((hipError_t)rc));
    return 0;
}

int esahrnet_keypoints_ex(const void* heat_dev, int n, int k, int height, int width, void* kp_dev, void* idx_dev,
                          esahrnet_stream stream) {
    if (!heat_dev || !kp_dev || n <= 0 || k <= 0) return fail("keypoints: bad argument");
    const int rc = esa::launch_keypoints(static_cast<const float*>(heat_dev), n * k, height, width,
                                         static_cast<float*>(kp_dev), static_cast<int*>(idx_dev),
                                         static_cast<hipStream_t>(stream));
    if (rc) return fail("keypoints: kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return 0;
}

int esahrnet_keypoints(const void* heat_dev, int n, int k, int height, int width, void* kp_dev,
                       esahrnet_stream stream) {
    return esahrnet_keypoints_ex(heat_dev, n, k, height, width, kp_dev, nullptr, stream);
}

int esahrnet_crops(const void* frames_dev, int n, int frame_h, int frame_w, const void* boxes_dev, int scale,
                   float mean, float stdv, void* out_dev, esahrnet_stream stream) {
    if (!frames_dev || !boxes_dev || !out_dev || n <= 0 || scale <= 0 || !(stdv > 0.f)) return fail("crops: bad argument");
    const int rc = esa::launch_crops(static_cast<const unsigned char*>(frames_dev), static_cast<const int*>(boxes_dev),
                                     static_cast<float*>(out_dev), n, frame_h, frame_w, scale, mean, stdv,
                                     static_cast<hipStream_t>(stream));
    if (rc) return fail("crops: kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return 0;
}

int esahrnet_flops_per_crop(esahrnet_handle h, int height, int width, double* flops) {
    if (!h || !flops) return fail("flops_per_crop: null argument");
    std::vector<int> lh, lw;
    level_dims(*h, height, width, lh, lw);
    double f = 0;
    for (const ConvSpec& s : h->specs)        // the reference's direct form: derived specs restate work already counted
        if (s.parent < 0) f += 2.0 * lh[s.level] * lw[s.level] * s.cout * s.cin * s.k * s.k;
    *flops = f;
    return 0;
}

int esahrnet_launch_count(esahrnet_handle h) { return h ? (int)h->ops.size() : -1; }

int esahrnet_tap_count(esahrnet_handle h) {
    if (!h) return -1;
    int n = 0;
    for (const Tensor& t : h->tensors) n += !t.tap.empty();
    return n;
}

int esahrnet_tap_name(esahrnet_handle h, int index, char* out, size_t cap) {
    if (!h || !out) return fail("tap_name: null argument");
    int n = 0;
    for (const Tensor& t : h->tensors)
        if (!t.tap.empty() && n++ == index) { snprintf(out, cap, "%s", t.tap.c_str()); return 0; }
    return fail("tap_name: index %d out of range", index);
}

static const Tensor* find_tap(esahrnet_handle h, const char* name) {
    for (const Tensor& t : h->tensors)
        if (t.tap == name) return &t;
    return nullptr;
}

int esahrnet_tap_shape(esahrnet_handle h, const char* name, int height, int width, int* c, int* th, int* tw) {
    if (!h || !name || !c || !th || !tw) return fail("tap_shape: null argument");
    const Tensor* t = find_tap(h, name);
    if (!t) return fail("tap_shape: no tensor named '%s'", name);
    std::vector<int> lh, lw;
    level_dims(*h, height, width, lh, lw);
    *c = t->C; *th = lh[t->level]; *tw = lw[t->level];
    return 0;
}

int esahrnet_tap_read(esahrnet_handle h, const char* name, int n, int height, int width,
                      const void* ws_dev, void* out_dev, esahrnet_stream stream) {
    if (!h || !name || !ws_dev || !out_dev) return fail("tap_read: null argument");
    if (!h->keep) return fail("tap_read: call esahrnet_set_debug_keep(h, 1) before the forward (buffers are recycled otherwise)");
    const Tensor* t = find_tap(h, name);
    if (!t) return fail("tap_read: no tensor named '%s'", name);
    if (plan_shape(*h, n, height, width)) return 1;
    if (t->alt != 0 && t->alt != (h->sp.head2 ? 2 : 1))
        return fail("tap_read: '%s' belongs to the head alternative that does not run at this shape", name);
    const int rc = esa::launch_fmt_to_nchw(h->fmt,
        static_cast<const char*>(ws_dev) + t->off, n, t->C, h->sp.lh[t->level], h->sp.lw[t->level], t->Cp,
        static_cast<float*>(out_dev), static_cast<hipStream_t>(stream));
    if (rc) return fail("tap_read: %s", hipGetErrorString((hipError_t)rc));
    return 0;
}

// ---- stand-alone operators on f32 NCHW tensors (tests) ---------------------------------------
int esahrnet_op_conv(const void* x_dev, int n, int cin, int height, int width, const float* w,
                     const float* b, int cout, int k, int stride, int relu, const void* res_dev,
                     void* y_dev, esahrnet_stream stream_) {
    return esahrnet_op_conv_ex(x_dev, n, cin, height, width, w, b, cout, k, stride, relu, res_dev, y_dev, 0, stream_);
}

int esahrnet_op_conv_ex(const void* x_dev, int n, int cin, int height, int width, const float* w,
                        const float* b, int cout, int k, int stride, int relu, const void* res_dev,
                        void* y_dev, int precision, esahrnet_stream stream_) {
    if (!x_dev || !w || !b || !y_dev) return fail("op_conv: null argument");
    if (precision < 0 || precision > 2) return fail("op_conv: precision %d", precision);
    const bool bf = precision == 1, x6 = precision == 2;
    const int fmt = bf ? esa::FMT_BF : x6 ? esa::FMT_F32 : esa::FMT_SB;
    const int eb = bf ? 2 : 4;
    if (!((k == 1 && stride == 1) || (k == 3 && (stride == 1 || stride == 2)))) return fail("op_conv: k=%d stride=%d unsupported", k, stride);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int cinp = bf ? pad64(cin) : pad32(cin), coutp = bf ? pad64(cout) : pad32(cout);
    const int oh = stride == 2 ? (height + 1) / 2 : height, ow = stride == 2 ? (width + 1) / 2 : width;
    std::vector<char> packed(bf ? esa::packed_weight_bytes_bf(coutp, cinp, k) : x6 ? esa::packed_weight_bytes_x6(coutp, cinp, k)
                                                                                    : esa::packed_weight_bytes(coutp, cinp, k), 0);
    if (bf) esa::pack_conv_weights_bf(w, cout, cin, k, coutp, cinp, packed.data());
    else if (x6) esa::pack_conv_weights_x6(w, cout, cin, k, coutp, cinp, packed.data());
    else esa::pack_conv_weights(w, cout, cin, k, coutp, cinp, packed.data());
    std::vector<float> bias(coutp, 0.f);
    std::copy(b, b + cout, bias.begin());
    void *dw = nullptr, *db = nullptr, *xs = nullptr, *ys = nullptr, *rs = nullptr;
    int rc = 0;
    auto cleanup = [&]() { for (void* p : {dw, db, xs, ys, rs}) if (p) (void)hipFree(p); };
    if (upload(packed, &dw) || upload(bias, &db)) { cleanup(); return 1; }
    const size_t xb = (size_t)n * height * width * cinp * eb, yb = (size_t)n * oh * ow * coutp * eb;
    if (hipMalloc(&xs, xb) != hipSuccess || hipMalloc(&ys, yb) != hipSuccess ||
        (res_dev && hipMalloc(&rs, yb) != hipSuccess)) { cleanup(); return fail("op_conv: hipMalloc failed"); }
    rc = esa::launch_nchw_to_fmt(fmt, static_cast<const float*>(x_dev), n, cin, height, width, static_cast<char*>(xs), cinp, stream);
    if (!rc && res_dev) rc = esa::launch_nchw_to_fmt(fmt, static_cast<const float*>(res_dev), n, cout, oh, ow, static_cast<char*>(rs), coutp, stream);
    if (!rc) {
        esa::ConvParams p{};
        p.x = static_cast<const char*>(xs); p.y = static_cast<char*>(ys); p.res = static_cast<const char*>(rs);
        p.w = static_cast<const uint4*>(dw); p.bias = static_cast<const float*>(db);
        p.N = n; p.H = height; p.W = width; p.OH = oh; p.OW = ow; p.Cinp = cinp; p.Coutp = coutp; p.relu = relu;
        p.fmt = fmt;
        rc = esa::launch_conv(p, k, stride, stream);
    }
    if (!rc) rc = esa::launch_fmt_to_nchw(fmt, static_cast<const char*>(ys), n, cout, oh, ow, coutp, static_cast<float*>(y_dev), stream);
    hipError_t se = hipStreamSynchronize(stream);
    cleanup();
    if (rc) return fail("op_conv: launch failed: %s", hipGetErrorString((hipError_t)rc));
    if (se != hipSuccess) return fail("op_conv: %s", hipGetErrorString(se));
    return 0;
}

int esahrnet_op_fuse(const void* const* xs_dev, const int* hs, const int* ws, int nterms, int n, int c,
                     int height, int width, int relu, void* y_dev, esahrnet_stream stream_) {
    return esahrnet_op_fuse_ex(xs_dev, hs, ws, nterms, n, c, height, width, relu, y_dev, 0, stream_);
}

int esahrnet_op_fuse_ex(const void* const* xs_dev, const int* hs, const int* ws, int nterms, int n, int c,
                        int height, int width, int relu, void* y_dev, int precision, esahrnet_stream stream_) {
    if (!xs_dev || !hs || !ws || !y_dev || nterms < 1 || nterms > 4) return fail("op_fuse: bad argument");
    if (precision < 0 || precision > 2) return fail("op_fuse: precision %d", precision);
    const bool bf = precision == 1;
    const int fmt = bf ? esa::FMT_BF : precision == 2 ? esa::FMT_F32 : esa::FMT_SB;
    const int eb = bf ? 2 : 4;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int cp = bf ? pad64(c) : pad32(c);
    void* bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    auto cleanup = [&]() { for (void* p : bufs) if (p) (void)hipFree(p); };
    int rc = 0;
    esa::FuseParams p{};
    p.nterms = nterms; p.N = n; p.H = height; p.W = width; p.Cp = cp; p.relu = relu; p.fmt = fmt;
    for (int i = 0; i < nterms && !rc; ++i) {
        if (hipMalloc(&bufs[i], (size_t)n * hs[i] * ws[i] * cp * eb) != hipSuccess) { cleanup(); return fail("op_fuse: hipMalloc failed"); }
        rc = esa::launch_nchw_to_fmt(fmt, static_cast<const float*>(xs_dev[i]), n, c, hs[i], ws[i], static_cast<char*>(bufs[i]), cp, stream);
        p.x[i] = static_cast<const char*>(bufs[i]); p.h[i] = hs[i]; p.w[i] = ws[i];
    }
    if (!rc && hipMalloc(&bufs[4], (size_t)n * height * width * cp * eb) != hipSuccess) { cleanup(); return fail("op_fuse: hipMalloc failed"); }
    p.y = static_cast<char*>(bufs[4]);
    if (!rc) rc = esa::launch_fuse(p, stream);
    if (!rc) rc = esa::launch_fmt_to_nchw(fmt, p.y, n, c, height, width, cp, static_cast<float*>(y_dev), stream);
    hipError_t se = hipStreamSynchronize(stream);
    cleanup();
    if (rc) return fail("op_fuse: launch failed: %s", hipGetErrorString((hipError_t)rc));
    if (se != hipSuccess) return fail("op_fuse: %s", hipGetErrorString(se));
    return 0;
}

}  // extern "C"
