// Thin PyTorch-ROCm extension over the C ABI of libdoda_hip.so.
//
// The arithmetic lives behind include/doda_hip.h; this file is host-side glue only (no device code):
// C++ autograd functions for the sparse convolutions and the fused BatchNorm(+ReLU), so that a U-Net
// step does not pay Python + ctypes overhead on each of its ~800 native launches (measured: the
// Python glue alone needs ~11.5 ms per step, as much as all GPU kernels together).
// doda_amd/spconv/functional.py and doda_amd/nn.py use it when it is built and fall back to the
// equivalent ctypes glue (same kernels) otherwise.
#include <torch/extension.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/functions/utils.h>
#include <torch/csrc/autograd/graph_task.h>
#include <torch/csrc/autograd/saved_variable.h>
#include <c10/hip/HIPGuard.h>
#include <unordered_map>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include "../../include/doda_hip.h"
#include <chrono>
#include <cstdlib>

// Debug aid (DODA_POISON=1): every tensor this file allocates uninitialised is filled with 0xFF bytes (NaN for fp32 / bf16 /
// fp64, -1 for int32) right after the allocation, on the current stream.  A kernel that reads something it (or a predecessor) did
// not write then yields NaN instead of whatever the caching allocator left there — the stale values that make such a read an
// intermittent, plausible-looking error (tools/traindet.py, tools/stepdet.py).
static const bool g_poison = [] { const char *e = getenv("DODA_POISON"); return e && e[0] == '1'; }();
static inline at::Tensor poisoned(at::Tensor t) {
    if (g_poison && t.defined() && t.is_cuda() && t.numel() > 0)
        (void)hipMemsetAsync(t.data_ptr(), 0xFF, (size_t)t.numel() * t.element_size(), c10::hip::getCurrentHIPStream(t.device().index()).stream());
    return t;
}
template <class... A> static inline at::Tensor pempty(A &&...a) { return poisoned(at::empty(std::forward<A>(a)...)); }
static inline at::Tensor pempty(at::IntArrayRef sizes, const at::TensorOptions &o) { return poisoned(at::empty(sizes, o)); }
template <class... A> static inline at::Tensor pempty_like(A &&...a) { return poisoned(at::empty_like(std::forward<A>(a)...)); }
// Debug aid (DODA_NANCHECK=1, with DODA_POISON=1): after every operator of this file its outputs are searched for NaN (a
// synchronisation each: slow) and the first hit is reported with the operator and its shapes — where a poisoned byte was read.
static const bool g_nancheck = [] { const char *e = getenv("DODA_NANCHECK"); return e && e[0] == '1'; }();
static int g_nan_reports = 0;
// Debug aid (DODA_FPLOG=1): a fingerprint (int64 sum of the raw 32-bit words) of every operator output, in call order;
// fp_log_take() hands the list over and clears it.  Two passes over the same inputs must log the same list: the first entry that
// differs names the operator whose output changed (tools/fwddet.py).
static const bool g_fplog = [] { const char *e = getenv("DODA_FPLOG"); return e && e[0] == '1'; }();
static std::vector<std::tuple<std::string, int64_t>> g_fp;
static std::mutex g_fp_mu;
static inline void fp_log(const at::Tensor &t, const char *what, const char *which, long long a, long long b, long long c) {
    if (!t.defined() || t.numel() == 0 || !t.is_cuda()) return;
    at::NoGradGuard ng;
    at::Tensor flat = t.detach().contiguous().reshape({-1});
    const int64_t bytes = flat.numel() * flat.element_size();
    at::Tensor words = bytes % 4 == 0 ? flat.view(at::kInt) : flat.view(at::kByte);
    const int64_t v = at::sum(words, at::kLong).item<int64_t>();
    char name[160];
    snprintf(name, sizeof(name), "%s/%s %lld %lld %lld [%lld]", what, which, a, b, c, (long long)flat.numel());
    std::lock_guard<std::mutex> lock(g_fp_mu);
    g_fp.emplace_back(std::string(name), v);
}
static inline void nan_check(const at::Tensor &t, const char *what, const char *which, long long a = 0, long long b = 0, long long c = 0) {
    if (g_fplog) fp_log(t, what, which, a, b, c);
    if (!g_nancheck || !t.defined() || !t.is_floating_point() || t.numel() == 0 || g_nan_reports >= 20) return;
    at::NoGradGuard ng;
    at::Tensor bad = at::isnan(t.detach());
    const long long n = bad.sum().item<int64_t>();
    if (n > 0) {
        ++g_nan_reports;
        at::Tensor pos = bad.reshape({-1}).nonzero().reshape({-1});
        fprintf(stderr, "[doda nancheck] %s: %lld NaN in %s %s, first at flat index %lld (of %lld); args %lld %lld %lld\n", what, n, which,
                c10::str(t.sizes()).c_str(), (long long)pos[0].item<int64_t>(), (long long)t.numel(), a, b, c);
    }
}

// Host-side time of the extension's own entry points (DODA_HOST_TIMING=1; tools/hostcount.py prints it): where the issuing
// thread spends a step — the step sits at the host / GPU crossover.
namespace host_timing {
struct Slot { const char *name; long long ns = 0, calls = 0; };
static Slot g_slots[] = {{"residual_block"}, {"conv_backward"}, {"bn_backward"}, {"flush_wgrads"}, {"wgrad_multi (library)"},
                         {"coarse_backward"}, {"coarse_forward"}};
static const bool g_on = getenv("DODA_HOST_TIMING") && getenv("DODA_HOST_TIMING")[0] == '1';
struct Scope {
    int k;
    std::chrono::steady_clock::time_point t0;
    explicit Scope(int k_) : k(k_) { if (g_on) t0 = std::chrono::steady_clock::now(); }
    ~Scope() {
        if (!g_on) return;
        g_slots[k].ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        g_slots[k].calls += 1;
    }
};
}  // namespace host_timing

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

inline void *stream_of(const at::Tensor &t) {
    return (void *)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

inline void check(int status, const char *what) {
    TORCH_CHECK(status == 0, what, " failed: ", doda_strerror(status), " (", status, ")");
}

inline int elem_bytes(const at::Tensor &t) {
    if (t.scalar_type() == at::kFloat) return 4;
    TORCH_CHECK(t.scalar_type() == at::kBFloat16, "doda: features must be float32 or bfloat16");
    return 2;
}

// BatchNorm statistics that ride in a conv epilogue (doda_spconv_gather_ex).  Forward: `want` asks for
// (sum y, sum y^2) of the output; data-grad: bn_* describe the BatchNorm(+ReLU) in FRONT of the conv and
// the sums are (sum dz, sum dz*xhat).  On return `stats` is [rows, 2, nc] or undefined (generic kernel).
struct GatherEpi {
    bool want = false;
    at::Tensor bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta;
    bool bn_relu = false;
    at::Tensor stats;
};

// A SubM table allocated by table_with_tilebook() carries its tilebook (doda_tilebook_build) in the same
// storage, 256-byte aligned behind the K x ld entries: the table tensor is the one handle every layer of
// the rulebook already passes around (forward, data-grad, saved for backward), so the tilebook reaches
// the kernels without a second argument through the Python and autograd layers.  Recognised by the exact
// storage size; any other table has exactly K * ld * 4 bytes.
inline size_t tilebook_offset(int64_t K, int64_t ld) { return ((size_t)(K * ld * 4) + 255) / 256 * 256; }

const void *tilebook_behind(const at::Tensor &tbl, int64_t n_out) {
    if (tbl.dim() != 2 || !tbl.is_contiguous() || tbl.storage_offset() != 0 || tbl.size(1) != n_out) return nullptr;
    const size_t tb = doda_tilebook_bytes((int32_t)n_out, (int32_t)tbl.size(0));
    if (tb == 0 || tbl.storage().nbytes() != tilebook_offset(tbl.size(0), tbl.size(1)) + tb) return nullptr;
    return (const char *)tbl.data_ptr() + tilebook_offset(tbl.size(0), tbl.size(1));
}

// int32 [K, m] table whose storage has room for the tilebook (written later by build_tilebook)
at::Tensor table_with_tilebook(int64_t K, int64_t m, const at::TensorOptions &iopt) {
    const size_t tb = doda_tilebook_bytes((int32_t)m, (int32_t)K);
    if (tb == 0) return pempty({K, m}, iopt);
    at::Tensor buf = pempty({(int64_t)((tilebook_offset(K, m) + tb) / 4)}, iopt);
    return buf.narrow(0, 0, K * m).view({K, m});
}

void build_tilebook(const at::Tensor &tbl, void *st) {
    const void *tb = tilebook_behind(tbl, tbl.size(1));
    TORCH_CHECK(tb, "doda: the table has no room for a tilebook");
    check(doda_tilebook_build((const int32_t *)tbl.data_ptr(), (int32_t)tbl.size(1), (int32_t)tbl.size(0),
                              (int32_t)tbl.size(1), const_cast<void *>(tb),
                              doda_tilebook_bytes((int32_t)tbl.size(1), (int32_t)tbl.size(0)), st), "doda_tilebook_build");
}

// BatchNorm statistics as fp64 TOTALS (ABI 9, doda_conv_epilogue.stats_totals): the conv epilogues add their workgroups' sums
// into DODA_STATS_TOTALS_DOUBLES(nc) doubles and the BatchNorm that follows derives its vectors from them inside its own sweep — the `final`
// reduction launches (52 per U-Net step) are gone.  The totals of a pass come out of ONE zeroed arena (one memset per forward
// pass instead of one per conv): a slice per statistics-producing call, forward and backward.  A slice is handed out once:
// the arena only grows, and a full one is replaced by a fresh zeroed buffer (one 8 MB memset every two or three steps; slices still
// referenced keep the old storage alive).
bool g_stats_totals = [] { const char *e = getenv("DODA_STATS_TOTALS"); return !(e && e[0] == '0'); }();
constexpr int64_t TOT_ARENA_DOUBLES = 1024 * 1024;     // 8 MB: 64 x nc doubles per slice (512 slices of 32 channels)
std::mutex g_tot_mu;
at::Tensor g_tot_buf;
int64_t g_tot_used = 0;
void stats_totals_begin_pass() {
    std::lock_guard<std::mutex> lock(g_tot_mu);
    g_tot_buf = at::Tensor();
    g_tot_used = 0;
}
at::Tensor stats_totals_take(int64_t nc, const at::TensorOptions &like) {
    const int64_t n = (int64_t)DODA_STATS_TOTALS_DOUBLES(nc);
    std::lock_guard<std::mutex> lock(g_tot_mu);
    if (!g_tot_buf.defined() || g_tot_buf.device() != like.device() || g_tot_used + n > g_tot_buf.numel()) {
        g_tot_buf = at::zeros({n > TOT_ARENA_DOUBLES ? n : TOT_ARENA_DOUBLES}, like.dtype(at::kDouble));
        g_tot_used = 0;
    }
    at::Tensor t = g_tot_buf.narrow(0, g_tot_used, n).view({(int64_t)DODA_STATS_SLOTS, 2, nc / 4, 16});
    g_tot_used += n;
    return t;
}
inline bool is_totals(const at::Tensor &st, int64_t c) {
    return st.defined() && st.scalar_type() == at::kDouble && st.dim() == 4 && st.size(0) == DODA_STATS_SLOTS && st.size(1) == 2 &&
           st.size(2) * 4 == c && st.size(3) == 16 && st.is_contiguous();
}

// y[t] = sum_o x[tbl[o][t]] . B_o   (include/doda_hip.h: doda_spconv_gather_ex)
at::Tensor gather(const at::Tensor &x_in, const at::Tensor &w, const c10::optional<at::Tensor> &packed,
                  const at::Tensor &tbl, int64_t n_out, int64_t layout, int64_t nc, bool out_f32,
                  const c10::optional<at::Tensor> &residual = c10::nullopt, GatherEpi *epi = nullptr) {
    const at::Tensor x = x_in.contiguous();
    TORCH_CHECK(x.is_cuda() && x.dim() == 2 && tbl.is_cuda() && tbl.dim() == 2, "doda gather: bad inputs");
    const int esz = elem_bytes(x);
    const int64_t K = tbl.size(0), ld = tbl.size(1), kc = x.size(1);
    void *st = stream_of(x);
    const bool f32_out = esz == 4 || out_f32;
    at::Tensor y = pempty({n_out, nc}, x.options().dtype(f32_out ? at::kFloat : at::kBFloat16));
    at::Tensor res;   // y = conv + res
    if (residual.has_value() && residual->defined()) {
        res = residual->contiguous();
        TORCH_CHECK(res.is_cuda() && res.scalar_type() == y.scalar_type() && res.dim() == 2 &&
                    res.size(0) == n_out && res.size(1) == nc, "doda gather: residual must be [n_out, nc] in the output dtype");
    }
    doda_conv_epilogue ep;
    memset(&ep, 0, sizeof(ep));
    int32_t stats_rows = 0;
    at::Tensor stats;
    ep.residual = res.defined() ? res.data_ptr() : nullptr;
    bool with_stats = epi && epi->want && n_out > 0;
    const bool totals = with_stats && g_stats_totals && nc % 4 == 0;
    if (with_stats) {
        if (totals) {
            stats = stats_totals_take(nc, x.options());
            ep.stats_totals = (double *)stats.data_ptr();
            ep.stats = (float *)stats.data_ptr();      // (non-NULL selects the statistics epilogue; nothing is written through it)
        } else {
            stats = pempty({(int64_t)doda_spconv_stats_capacity((int32_t)n_out), 2, nc}, x.options().dtype(at::kFloat));
            ep.stats = (float *)stats.data_ptr();
        }
        ep.stats_rows_h = &stats_rows;
        if (epi->bn_x.defined()) {
            TORCH_CHECK(epi->bn_x.scalar_type() == y.scalar_type() && epi->bn_x.is_contiguous() &&
                        epi->bn_x.size(0) == n_out && epi->bn_x.size(1) == nc, "doda gather: bn_x must match the output");
            ep.bn_x = epi->bn_x.data_ptr();
            ep.bn_mean = (const float *)epi->bn_mean.data_ptr();
            ep.bn_invstd = (const float *)epi->bn_invstd.data_ptr();
            ep.bn_gamma = (const float *)epi->bn_gamma.data_ptr();
            ep.bn_beta = (const float *)epi->bn_beta.data_ptr();
            ep.bn_relu = epi->bn_relu ? 1 : 0;
        }
    }
    ep.tilebook = tilebook_behind(tbl, n_out);
    ep.tilebook_rows = ep.tilebook ? (int32_t)n_out : 0;
    const void *wptr;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    at::Tensor ws_t, wc;
    int lay = (int)layout;
    bool use_packed = packed.has_value() && packed->defined();
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (use_packed) {
            wptr = packed->data_ptr();
            lay = (int)layout | 0x100;
        } else {
            wc = w.contiguous();
            TORCH_CHECK(wc.scalar_type() == at::kFloat && wc.numel() == K * kc * nc, "doda gather: weight shape");
            wptr = wc.data_ptr();
            ws_bytes = doda_spconv_gather_workspace_bytes((int)K, (int)kc, (int)nc, esz);
            ws_t = pempty({(int64_t)ws_bytes}, x.options().dtype(at::kByte));
            ws = ws_t.data_ptr();
            lay = (int)layout;
        }
        const int status = doda_spconv_gather_ex(x.data_ptr(), (int)x.size(0), (int)kc, esz, (const float *)wptr,
                                                 (int)nc, (const int32_t *)tbl.data_ptr(), (int)ld, (int)K, (int)n_out,
                                                 y.data_ptr(), out_f32 ? 1 : 0, lay, ws, ws_bytes, &ep, st);
        if (status == DODA_ERR_UNSUPPORTED && use_packed) {  // fast path refused the pre-packed weights
            use_packed = false;
            continue;
        }
        if (status == DODA_ERR_UNSUPPORTED && with_stats) {  // generic kernel: no statistics in its epilogue
            with_stats = false;
            ep.stats = nullptr;
            ep.stats_totals = nullptr;
            ep.stats_rows_h = nullptr;
            ep.bn_x = nullptr;
            use_packed = packed.has_value() && packed->defined();
            continue;
        }
        check(status, "doda_spconv_gather");
        break;
    }
    if (epi) epi->stats = (with_stats && stats_rows > 0) ? (totals ? stats : stats.narrow(0, 0, stats_rows)) : at::Tensor();
    if (g_nancheck || g_fplog) {
        nan_check(x, "gather", "INPUT x", K, n_out, layout);
        nan_check(y, "gather", "y", K, n_out, layout);
        if (epi && epi->stats.defined()) nan_check(epi->stats, "gather", "statistics", K, n_out, stats_rows);
    }
    return y;
}

// pair lists of a rulebook for the pair-list weight-gradient kernel (all optional)
struct PairLists {
    at::Tensor in, out, num;   // int32 [K, ld] (unit inner stride), int32 [K] or undefined (full lists)
    at::Tensor seg;            // int32 [K, nt] segment prefix of the lists (with num)
    bool defined() const { return in.defined() && out.defined(); }
    int64_t ld() const { return in.size(0) > 1 ? in.stride(0) : in.size(1); }
};

inline bool pairs_usable(const at::Tensor &a, const at::Tensor &b, const PairLists &pl) {
    return pl.defined() && a.scalar_type() == at::kBFloat16 && b.scalar_type() == at::kBFloat16 &&
           a.size(1) % 16 == 0 && b.size(1) % 16 == 0;
}

at::Tensor wgrad(const at::Tensor &a_in, const at::Tensor &b_in, const at::Tensor &tbl, int64_t n_rows,
                 const PairLists &pl = PairLists()) {
    const at::Tensor a = a_in.contiguous(), b = b_in.contiguous();
    TORCH_CHECK(a.scalar_type() == b.scalar_type(), "doda wgrad: dtype mismatch");
    const int esz = elem_bytes(a);
    const int64_t K = tbl.size(0), ld = tbl.size(1), ca = a.size(1), cb = b.size(1);
    at::Tensor dw = pempty({K, ca, cb}, a.options().dtype(at::kFloat));
    // one job of the multi-layer entry point (ABI 7: the only weight-gradient entry point); the library picks the
    // kernel: LDS-staged over the tilebook, pair lists, or the gather table
    doda_wgrad_job j;
    memset(&j, 0, sizeof(j));
    j.a = a.data_ptr(); j.b = b.data_ptr(); j.tbl = (const int32_t *)tbl.data_ptr(); j.dw = (float *)dw.data_ptr();
    j.ca = (int32_t)ca; j.cb = (int32_t)cb; j.ld = (int32_t)ld; j.K = (int32_t)K; j.n_rows = (int32_t)n_rows;
    j.elem_bytes = esz; j.n_a = (int32_t)a.size(0);
    j.tilebook = tilebook_behind(tbl, n_rows);
    if (pairs_usable(a, b, pl)) {
        j.pair_in = (const int32_t *)pl.in.data_ptr();
        j.pair_out = (const int32_t *)pl.out.data_ptr();
        j.pair_num = pl.num.defined() ? (const int32_t *)pl.num.data_ptr() : nullptr;
        j.pair_ld = (int32_t)pl.ld();
        j.pair_seg = pl.seg.defined() ? (const int32_t *)pl.seg.data_ptr() : nullptr;
        j.pair_seg_nt = pl.seg.defined() ? (int32_t)pl.seg.size(1) : 0;
    }
    const size_t wsb = doda_spconv_wgrad_multi_workspace_bytes(&j, 1), dsb = doda_spconv_wgrad_multi_desc_bytes(1);
    at::Tensor ws = pempty({(int64_t)wsb}, a.options().dtype(at::kByte));
    at::Tensor desc = pempty({(int64_t)dsb}, a.options().dtype(at::kByte));
    check(doda_spconv_wgrad_multi(&j, 1, ws.data_ptr(), wsb, desc.data_ptr(), dsb, stream_of(a)), "doda_spconv_wgrad_multi");
    return dw;
}

// spconv-format pair lists of a gather table without the -1 fill (doda_rulebook_pairs, flip | 2), their
// counts and their segment prefix [K, nt] (the head of the export's workspace)
std::tuple<at::Tensor, at::Tensor, at::Tensor> export_pairs(const at::Tensor &tbl, int64_t n_rows, bool flip, void *st) {
    const int64_t K = tbl.size(0);
    const auto iopt = tbl.options();
    at::Tensor pairs = pempty({2, K, n_rows > 0 ? n_rows : 1}, iopt), num = pempty({K}, iopt);
    // the workspace is an ordinary int32 tensor: its head IS the segment prefix handed on to the weight
    // gradient, and a view of it takes part in record_stream like any other allocation
    const size_t wsb = doda_rulebook_pairs_workspace_bytes((int32_t)n_rows, (int32_t)K);
    at::Tensor ws = pempty({(int64_t)((wsb > 256 ? wsb : 256) / 4)}, iopt);
    check(doda_rulebook_pairs((const int32_t *)tbl.data_ptr(), (int32_t)tbl.size(1), (int32_t)K, (int32_t)n_rows,
                              (flip ? 1 : 0) | 2, (int32_t *)pairs.data_ptr(), (int32_t)pairs.size(2),
                              (int32_t *)num.data_ptr(), ws.data_ptr(), (size_t)ws.numel() * 4, st),
          "doda_rulebook_pairs");
    const int64_t tile = doda_rulebook_pairs_tile(), nt = n_rows > 0 ? (n_rows + tile - 1) / tile : 1;
    return {pairs, num, ws.narrow(0, 0, K * nt).view({K, nt})};
}

// One int32 from the device to the host WITHOUT a spinning wait: copy into pinned memory, then sleep on an event created
// with hipEventBlockingSync.  Tensor::item() waits in hipStreamSynchronize, which spins by default: the rulebook thread
// makes six such waits per step (~1.5 ms of a core) next to the issuing thread, which is what paces the step on a busy
// host.  DODA_SPIN_READBACK=1 restores item().
bool read_back_blocking(const void *dev_ptr, int32_t *out, int n, void *st, int dev) {   // n <= 16; st: a stream of device `dev`; false: the caller spins
    static const bool spin = getenv("DODA_SPIN_READBACK") && getenv("DODA_SPIN_READBACK")[0] == '1';
    if (spin) return false;
    struct Slot { int32_t *host = nullptr; hipEvent_t ev = nullptr; int dev = -1; };
    static thread_local Slot slot;
    // the event belongs to the device it was created on: a thread that later builds a pyramid on another device gets a
    // new one under that device's guard (recording a foreign-device event on the stream fails); the pinned buffer is shared
    const int cur = dev;
    c10::hip::HIPGuard dev_guard((c10::DeviceIndex)dev);
    if (!slot.host)
        TORCH_CHECK(hipHostMalloc((void **)&slot.host, 64, hipHostMallocDefault) == hipSuccess, "doda: hipHostMalloc");
    if (!slot.ev || slot.dev != cur) {
        if (slot.ev) (void)hipEventDestroy(slot.ev);
        slot.ev = nullptr;
        TORCH_CHECK(hipEventCreateWithFlags(&slot.ev, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess, "doda: hipEventCreate");
        slot.dev = cur;
    }
    TORCH_CHECK(hipMemcpyAsync(slot.host, dev_ptr, (size_t)n * 4, hipMemcpyDeviceToHost, (hipStream_t)st) == hipSuccess, "doda: hipMemcpyAsync");
    TORCH_CHECK(hipEventRecord(slot.ev, (hipStream_t)st) == hipSuccess, "doda: hipEventRecord");
    TORCH_CHECK(hipEventSynchronize(slot.ev) == hipSuccess, "doda: hipEventSynchronize");
    for (int k = 0; k < n; ++k) out[k] = slot.host[k];
    return true;
}
int32_t read_back_i32(const at::Tensor &t, void *st) {
    int32_t v = 0;
    if (read_back_blocking(t.data_ptr(), &v, 1, st, (int)t.device().index())) return v;
    return t.item<int32_t>();
}

// with_pairs: also export every rulebook's pair lists (SubM: [2,27,M] from nbr; k2s2: [2,8,M] from par_off)
// for the pair-list weight gradient — on the same (side) stream, off the critical path.
typedef std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, std::vector<int64_t>, at::Tensor, at::Tensor,
                   at::Tensor, at::Tensor, at::Tensor, at::Tensor> PyramidLevel;
// pairs_min_rows < 0: no lists; else lists for rulebooks of at least that many rows.
std::vector<PyramidLevel> build_pyramid(const at::Tensor &indices_in, std::vector<int64_t> shape, int64_t batch,
                                        int64_t n_levels, int64_t pairs_min_rows, int64_t tile_min_rows,
                                        int64_t tile_levels) {
    TORCH_CHECK(indices_in.is_cuda() && indices_in.scalar_type() == at::kInt && indices_in.dim() == 2 &&
                indices_in.size(1) == 4 && shape.size() == 3, "doda build_pyramid: indices must be int32 [M,4] on the GPU");
    std::vector<PyramidLevel> out;
    at::Tensor indices = indices_in.contiguous();
    const auto iopt = indices.options();
    void *st = stream_of(indices);
    for (int64_t lvl = 0; lvl < n_levels; ++lvl) {
        const int32_t m = (int32_t)indices.size(0);
        int32_t shp[3] = {(int32_t)shape[0], (int32_t)shape[1], (int32_t)shape[2]};
        size_t wsb = doda_rulebook_workspace_bytes(m);
        {   // room for the direct-address grid behind it (doda_hip.h: grids of <= 2^28 cells; larger ones keep the hash)
            const long double cells = (long double)batch * shape[0] * shape[1] * shape[2];
            static const long long grid_max = [] {      // (csrc/rulebook.hip GRID_MAX_CELLS: the same switch)
                const char *e = getenv("DODA_RULEBOOK_GRID_MAX_LOG2");
                const int b = e ? atoi(e) : 28;
                return 1ll << (b < 0 ? 0 : b > 30 ? 30 : b);
            }();
            if (m > 0 && cells > 0 && cells <= (long double)grid_max) wsb = (wsb + 255) / 256 * 256 + (size_t)cells * 4;
        }
        at::Tensor ws = pempty({(int64_t)(wsb > 256 ? wsb : 256)}, iopt.dtype(at::kByte));
        // tilebooks for the finest `tile_levels` levels: DODA's 16- and 32-channel bf16 layers (2) or the 16-channel
        // fp32 layers (1) — rows of 32 or 64 bytes, what the tile kernel stages
        const bool tiled = lvl < tile_levels && tile_min_rows >= 0 && m >= tile_min_rows;
        at::Tensor nbr = tiled ? table_with_tilebook(27, m, iopt) : pempty({27, m}, iopt);
        check(doda_rulebook_subm((const int32_t *)indices.data_ptr(), m, shp, (int32_t)batch, 3,
                                 (int32_t *)nbr.data_ptr(), m, ws.data_ptr(), (size_t)ws.numel(), st),
              "doda_rulebook_subm");
        if (tiled) build_tilebook(nbr, st);
        at::Tensor sp, sn, sh, dp, dn, dh;
        const bool with_pairs = pairs_min_rows >= 0 && m >= pairs_min_rows;
        // (round 4) a SubM rulebook with a tilebook needs no pair lists: every bf16 layer of 16 / 32 channels on either side
        // takes the LDS-staged weight gradient over the tilebook (csrc/spconv_wgrad.hip classify(): from 32 768 rows, the
        // smallest rulebook that gets a tilebook; a layer it does not take falls back to the gather table)
        const bool subm_pairs = with_pairs && !tiled;
        if (subm_pairs) std::tie(sp, sn, sh) = export_pairs(nbr, m, true, st);
        if (lvl == n_levels - 1) {
            out.emplace_back(nbr, at::Tensor(), at::Tensor(), at::Tensor(), std::vector<int64_t>(), sp, sn, sh, dp, dn, dh);
            break;
        }
        at::Tensor parent = pempty({m > 0 ? m : 1}, iopt), off = pempty({m > 0 ? m : 1}, iopt);
        at::Tensor out_idx = pempty({m > 0 ? m : 1, 4}, iopt), count = pempty({1}, iopt);
        check(doda_rulebook_down2_assign((const int32_t *)indices.data_ptr(), m, shp, (int32_t)batch,
                                         (int32_t *)parent.data_ptr(), (int32_t *)off.data_ptr(),
                                         (int32_t *)out_idx.data_ptr(), (int32_t *)count.data_ptr(), ws.data_ptr(),
                                         (size_t)ws.numel(), st), "doda_rulebook_down2_assign");
        const int32_t m_out = read_back_i32(count, st);   // the level's one size read-back
        at::Tensor child = pempty({8, m_out}, iopt), par_off = pempty({8, m}, iopt);
        check(doda_rulebook_down2_tables((const int32_t *)parent.data_ptr(), (const int32_t *)off.data_ptr(), m, m_out,
                                         (int32_t *)child.data_ptr(), m_out, (int32_t *)par_off.data_ptr(), m, st),
              "doda_rulebook_down2_tables");
        // (round 5) the strided rulebooks' lists are no longer exported by default: since the SubM layers of levels 1-2 take the
        // tile weight gradient, the only consumers left were the six k2 s2 / inverse layers of levels 1-3, and the export (three
        // launches per rulebook on the rulebook stream, 230 us per step) cost four times what the pair-list kernels saved over
        // the gather-table kernel (54 us): 5.2-5.6 -> 5.0 ms per step in alternating runs (tools/pairs_ab.sh).
        static const bool down_pairs = [] { const char *e = getenv("DODA_WGRAD_PAIRS_DOWN"); return e && e[0] == '1'; }();
        if (with_pairs && down_pairs) std::tie(dp, dn, dh) = export_pairs(par_off, m, false, st);
        std::vector<int64_t> oshape = {(shape[0] - 2) / 2 + 1, (shape[1] - 2) / 2 + 1, (shape[2] - 2) / 2 + 1};
        at::Tensor outids = out_idx.narrow(0, 0, m_out);
        out.emplace_back(nbr, outids, child, par_off, oshape, sp, sn, sh, dp, dn, dh);
        indices = outids;
        shape = oshape;
    }
    return out;
}

// ---- deferred weight gradients ---------------------------------------------------------------
// The weight gradient of a layer feeds nothing else in the backward pass.  With deferral switched on
// (set_defer_wgrad) the conv backward only queues (features, dy, table / pair lists, weight); an engine
// callback that runs when the backward pass is complete issues ALL queued jobs through
// doda_spconv_wgrad_multi (~8 launches instead of 2 per layer) and only THEN binds the results:
//   * weight.grad undefined at flush time -> a fresh tensor is written and bound as .grad;
//   * weight.grad defined (an earlier backward pass of the same optimizer step — the two passes of
//     tool/st.py:136-198 —, zero_grad(set_to_none=False), or another consumer of the weight whose
//     gradient went through AccumulateGrad during this pass) -> the job ACCUMULATES into it.
// Nothing uninitialised is ever visible as .grad.  A weight that is queued twice in one pass (shared
// module, two forward passes summed into one loss) gets its second job in a follow-up call, after the
// first has written.  Preconditions per layer: fp32 contiguous leaf weight (otherwise the layer computes
// its gradient on the spot).  AccumulateGrad hooks do not see these gradients, so torch DDP must not be
// combined with deferral (doda_amd.dist reduces gradients itself).  If a backward pass aborts, its
// queued jobs are dropped when the next pass starts (different graph-task id).
// ---- gradient homes (doda_amd.dist.GradAllReduce) -------------------------------------------------------------
// Data-parallel training all-reduces the gradients in a few flat buckets (reference tool/train.py:360-361: DDP's
// bucket views).  A parameter may be given a HOME: a view of its bucket.  The kernels that produce the parameter's
// gradient — the deferred weight-gradient launch for conv weights, the BatchNorm backward for gamma / beta — then write
// straight into the home and .grad becomes an alias of it: the reducer finds every byte in place (no torch.cat before
// the collective, no copy back after it).  A home is used only for the FIRST gradient of a parameter in an optimizer
// step (.grad undefined, and for BatchNorm not handed out yet in this backward pass); anything else — a second
// backward pass, a shared module — takes the ordinary path and is accumulated by autograd / DODA_WGRAD_ACCUMULATE.
struct GradHome {
    c10::weak_intrusive_ptr<c10::TensorImpl> owner;
    at::Tensor view;
    int task = -2;       // graph task that last took it (BatchNorm nodes)
    GradHome(const at::Tensor &param, const at::Tensor &v)
        : owner(c10::weak_intrusive_ptr<c10::TensorImpl>(param.getIntrusivePtr())), view(v) {}
};
std::mutex g_home_mu;
std::unordered_map<c10::TensorImpl *, GradHome> g_homes;

void set_grad_home(const at::Tensor &param, const c10::optional<at::Tensor> &view) {
    std::lock_guard<std::mutex> lock(g_home_mu);
    c10::TensorImpl *key = param.unsafeGetTensorImpl();
    if (!view.has_value() || !view->defined()) {
        g_homes.erase(key);
        return;
    }
    TORCH_CHECK(view->sizes() == param.sizes() && view->scalar_type() == param.scalar_type() && view->device() == param.device() &&
                view->is_contiguous() && !view->requires_grad(), "doda set_grad_home: the view must be a contiguous tensor shaped like the parameter");
    g_homes.erase(key);
    g_homes.emplace(key, GradHome(param, *view));
}
// forget the parameter's home only if it still is `view` (a reducer built later over the same module has replaced it
// otherwise, and the older reducer's close() / __del__ must leave that one alone).  True when a home was dropped.
bool drop_grad_home(const at::Tensor &param, const at::Tensor &view) {
    std::lock_guard<std::mutex> lock(g_home_mu);
    auto it = g_homes.find(param.unsafeGetTensorImpl());
    if (it == g_homes.end() || !view.defined() || it->second.view.data_ptr() != view.data_ptr()) return false;
    g_homes.erase(it);
    return true;
}
void clear_grad_homes() {
    std::lock_guard<std::mutex> lock(g_home_mu);
    g_homes.clear();
}
// an alias of the parameter's home, or undefined.  task >= 0: not twice within one graph task.
at::Tensor take_grad_home(const at::Tensor &param, int task = -1) {
    if (!param.defined() || param.grad().defined()) return at::Tensor();
    std::lock_guard<std::mutex> lock(g_home_mu);
    if (g_homes.empty()) return at::Tensor();
    auto it = g_homes.find(param.unsafeGetTensorImpl());
    if (it == g_homes.end()) return at::Tensor();
    auto alive = it->second.owner.lock();
    if (!alive || alive.get() != param.unsafeGetTensorImpl()) {   // the parameter died and its address was reused
        g_homes.erase(it);
        return at::Tensor();
    }
    if (task >= 0) {
        if (it->second.task == task) return at::Tensor();
        it->second.task = task;
    }
    return it->second.view.alias();
}

struct PendingWgrad {
    at::Tensor a, b, tbl, weight;
    PairLists pl;
    int64_t n_rows;
};
std::mutex g_wq_mu;
std::vector<PendingWgrad> g_wq;
bool g_wq_callback = false, g_defer_wgrad = false;
// (set_direct_grads, with set_defer_wgrad) parameter gradients of the extension's own nodes are written to .grad by the
// nodes themselves — conv weights by the deferred launch, BatchNorm gamma / beta by the BatchNorm backward — and the nodes keep
// NO autograd edge to those leaves: the engine then has 203 AccumulateGrad nodes less to schedule per U-Net step (~0.5 ms of
// the issuing thread, which paces the step as much as the GPU does).  Same contract as the deferred weight gradient: no
// AccumulateGrad hooks for these parameters, torch.autograd.grad(..., params) does not see them.
bool g_direct_grads = false;

inline bool direct_leaf(const at::Tensor &p) {
    return g_direct_grads && g_defer_wgrad && p.defined() && p.requires_grad() && p.is_leaf() &&
           p.scalar_type() == at::kFloat && p.is_contiguous();
}
// A backward pass that was asked for SPECIFIC gradients (torch.autograd.grad(loss, x), loss.backward(inputs=[...])) carries a
// non-empty exec_info map; a plain loss.backward() — every leaf accumulates — an empty one.  The direct path has no edge the
// engine could consult per parameter, so in a restricted pass the nodes neither compute nor deposit parameter gradients
// (such passes used to mutate .grad and pay for a full weight gradient).  torch.autograd.grad(loss, params) still
// does not see these parameters: documented in INTEGRATION.md next to set_direct_grads.
inline bool plain_accumulating_backward() {
    const auto *info = torch::autograd::get_current_graph_task_exec_info();
    return info == nullptr || info->empty();
}

// what AccumulateGrad does for a fresh, contiguous gradient: bind it, or add to an existing one
inline void deposit_grad(const at::Tensor &param, const at::Tensor &g) {
    at::Tensor &slot = const_cast<at::Tensor &>(param).mutable_grad();
    if (!slot.defined()) {
        slot = g.sizes() == param.sizes() ? g : g.reshape(param.sizes());
    } else {
        at::NoGradGuard no_grad;
        slot.add_(g.reshape(slot.sizes()));
    }
}
int g_wq_task = -2;
c10::optional<c10::hip::HIPStream> g_wq_stream;

void issue_wgrads(std::vector<PendingWgrad> &q, const c10::hip::HIPStream &st) {
    if (q.empty()) return;
    c10::hip::HIPStreamGuard guard(st);
    std::vector<doda_wgrad_job> jobs(q.size());
    std::vector<at::Tensor> fresh(q.size());
    for (size_t k = 0; k < q.size(); ++k) {
        const PendingWgrad &p = q[k];
        at::Tensor target = p.weight.grad();
        int flags = 0;
        if (target.defined() && target.scalar_type() == at::kFloat && target.is_contiguous() &&
            target.sizes() == p.weight.sizes() && target.device() == p.weight.device()) {
            flags = DODA_WGRAD_ACCUMULATE;
        } else {
            TORCH_CHECK(!target.defined(), "doda deferred wgrad: existing .grad of a conv weight is not a contiguous fp32 tensor");
            target = take_grad_home(p.weight);     // the reducer's bucket view, when the weight has one
            if (!target.defined()) target = pempty(p.weight.sizes(), p.weight.options());
            fresh[k] = target;
        }
        doda_wgrad_job j;
        memset(&j, 0, sizeof(j));
        j.a = p.a.data_ptr(); j.b = p.b.data_ptr();
        j.tbl = (const int32_t *)p.tbl.data_ptr();
        j.dw = (float *)target.data_ptr();
        j.ca = (int32_t)p.a.size(1); j.cb = (int32_t)p.b.size(1);
        j.ld = (int32_t)p.tbl.size(1); j.K = (int32_t)p.tbl.size(0);
        j.n_rows = (int32_t)p.n_rows; j.elem_bytes = (int32_t)elem_bytes(p.a);
        j.n_a = (int32_t)p.a.size(0);
        j.flags = flags;
        j.tilebook = tilebook_behind(p.tbl, p.n_rows);   // SubM rulebooks of the fine levels carry one behind the table
        if (pairs_usable(p.a, p.b, p.pl)) {
            j.pair_in = (const int32_t *)p.pl.in.data_ptr();
            j.pair_out = (const int32_t *)p.pl.out.data_ptr();
            j.pair_num = p.pl.num.defined() ? (const int32_t *)p.pl.num.data_ptr() : nullptr;
            j.pair_seg = p.pl.seg.defined() ? (const int32_t *)p.pl.seg.data_ptr() : nullptr;
            j.pair_seg_nt = p.pl.seg.defined() ? (int32_t)p.pl.seg.size(1) : 0;
            j.pair_ld = (int32_t)p.pl.ld();
        }
        jobs[k] = j;
    }
    const auto opt = q[0].a.options().dtype(at::kByte);
    const size_t wsb = doda_spconv_wgrad_multi_workspace_bytes(jobs.data(), (int32_t)jobs.size());
    const size_t dsb = doda_spconv_wgrad_multi_desc_bytes((int32_t)jobs.size());
    at::Tensor ws = pempty({(int64_t)wsb}, opt), desc = pempty({(int64_t)dsb}, opt);
    {
        host_timing::Scope lib_scope(4);
        check(doda_spconv_wgrad_multi(jobs.data(), (int32_t)jobs.size(), ws.data_ptr(), wsb, desc.data_ptr(), dsb,
                                      (void *)st.stream()), "doda_spconv_wgrad_multi");
    }
    for (size_t k = 0; k < q.size(); ++k)
        if (fresh[k].defined()) q[k].weight.mutable_grad() = fresh[k];
    if (g_nancheck || g_fplog)
        for (size_t k = 0; k < q.size(); ++k) {
            nan_check(q[k].a, "weight gradient", "INPUT a", (long long)k, q[k].n_rows, q[k].a.size(1));
            nan_check(q[k].b, "weight gradient", "INPUT b", (long long)k, q[k].n_rows, q[k].b.size(1));
            nan_check(q[k].weight.grad(), "weight gradient", "dw", (long long)k, q[k].n_rows, (long long)jobs[k].flags);
        }
}

// jobs whose weight already appeared earlier in the queue wait for a follow-up call
void issue_in_rounds(std::vector<PendingWgrad> &q, const c10::hip::HIPStream &st) {
    while (!q.empty()) {
        std::vector<PendingWgrad> now, later;
        for (PendingWgrad &p : q) {
            bool dup = false;
            for (const PendingWgrad &e : now) dup |= e.weight.unsafeGetTensorImpl() == p.weight.unsafeGetTensorImpl();
            (dup ? later : now).push_back(std::move(p));
        }
        issue_wgrads(now, st);
        q.swap(later);
    }
}

// Split flush for data-parallel training (doda_amd.dist.GradAllReduce, world size > 1).  Nearly all gradient
// BYTES belong to the coarse levels (64..224 channels: 28 of the U-Net's 30 MB), nearly all weight-gradient
// TIME to levels 1-2 (16 / 32 channels, 600k / 180k rows).  The wide layers are issued first and an event is
// recorded behind them: the all-reduce of their gradients (and of everything AccumulateGrad delivered during
// backward) can start on another stream while the narrow layers' kernels still run.  The rule is static
// (channel counts of the weight), so every rank cuts its gradients the same way whatever its batch looks like.
bool g_wq_split = false, g_ev_early_valid = false;
hipEvent_t g_ev_early = nullptr;
int g_ev_early_dev = -1;   // device the event lives on (one training process drives one device; re-made if that changes)

inline bool narrow_weight(const at::Tensor &w) {   // [kD, kH, kW, Cin, Cout]
    return w.dim() == 5 && w.size(3) <= 32 && w.size(4) <= 32;
}

// Side flush (round 5): the weight gradients queued when the backward pass ENTERS its coarse levels — the whole decoder of
// levels 1-3, about half of the step's weight-gradient time — are issued on a second stream, behind an event on the backward
// pass's stream, and run under the coarse levels' chain of launch-floor kernels (few rows: most CUs idle).  The backward
// pass's stream waits for them in flush_wgrads(), before it issues the rest; the queued tensors are kept until then (the
// caching allocator would otherwise hand their memory to the main stream while the side stream still reads it).
std::vector<PendingWgrad> g_side_keep;
c10::optional<c10::hip::HIPStream> g_side_stream, g_side_main;
hipEvent_t g_ev_fork = nullptr, g_ev_join = nullptr;
int g_ev_side_dev = -1;
bool g_side_pending = false;

int64_t flush_wgrads_side() {
    std::vector<PendingWgrad> q;
    c10::optional<c10::hip::HIPStream> st;
    {
        std::lock_guard<std::mutex> lock(g_wq_mu);
        q.swap(g_wq);
        st = g_wq_stream;
    }
    if (q.empty() || !st) return 0;
    host_timing::Scope host_scope(3);
    const int dev = (int)st->device_index();
    c10::hip::HIPGuard dev_guard(dev);
    if (g_ev_side_dev != dev) {
        if (g_ev_fork) hipEventDestroy(g_ev_fork);
        if (g_ev_join) hipEventDestroy(g_ev_join);
        TORCH_CHECK(hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&g_ev_join, hipEventDisableTiming) == hipSuccess, "doda: hipEventCreate");
        g_ev_side_dev = dev;
        g_side_stream = c10::hip::getStreamFromPool(false, (c10::DeviceIndex)dev);
    }
    const int64_t n = (int64_t)q.size();
    std::vector<PendingWgrad> keep(q);   // (references: issue_in_rounds consumes its argument)
    if (g_side_pending) {                // a second side flush in one backward pass: it runs behind the first
        g_side_keep.insert(g_side_keep.end(), keep.begin(), keep.end());
    } else {
        g_side_keep.swap(keep);
    }
    TORCH_CHECK(hipEventRecord(g_ev_fork, st->stream()) == hipSuccess &&
                hipStreamWaitEvent(g_side_stream->stream(), g_ev_fork, 0) == hipSuccess, "doda: side flush fork");
    issue_in_rounds(q, *g_side_stream);
    TORCH_CHECK(hipEventRecord(g_ev_join, g_side_stream->stream()) == hipSuccess, "doda: side flush join event");
    g_side_main = st;
    g_side_pending = true;
    return n;
}
// the backward pass's stream waits for the side flush; true when there was one
bool join_side_wgrads(const c10::optional<c10::hip::HIPStream> &st) {
    if (!g_side_pending) return false;
    const c10::hip::HIPStream &main = st ? *st : *g_side_main;
    TORCH_CHECK(hipStreamWaitEvent(main.stream(), g_ev_join, 0) == hipSuccess, "doda: side flush join");
    g_side_pending = false;
    g_side_keep.clear();   // (freed memory is re-used on `main`, behind the wait)
    return true;
}

void flush_wgrads() {
    host_timing::Scope host_scope(3);
    std::vector<PendingWgrad> q;
    c10::optional<c10::hip::HIPStream> st;
    {
        std::lock_guard<std::mutex> lock(g_wq_mu);
        q.swap(g_wq);
        st = g_wq_stream;
        g_wq_callback = false;
        g_wq_task = -2;
    }
    join_side_wgrads(st);
    if (q.empty() || !st) return;   // nothing queued (or no backward pass has run yet: no stream recorded)
    if (!g_wq_split) {
        issue_in_rounds(q, *st);
        return;
    }
    std::vector<PendingWgrad> wide, narrow;
    for (PendingWgrad &p : q) (narrow_weight(p.weight) ? narrow : wide).push_back(std::move(p));
    issue_in_rounds(wide, *st);
    if (g_ev_early && g_ev_early_dev != (int)st->device_index()) {
        hipEventDestroy(g_ev_early);
        g_ev_early = nullptr;
    }
    if (!g_ev_early) {
        c10::hip::HIPGuard dev_guard(st->device_index());
        TORCH_CHECK(hipEventCreateWithFlags(&g_ev_early, hipEventDisableTiming) == hipSuccess, "doda: hipEventCreate");
        g_ev_early_dev = (int)st->device_index();
    }
    TORCH_CHECK(hipEventRecord(g_ev_early, st->stream()) == hipSuccess, "doda: hipEventRecord");
    g_ev_early_valid = true;
    issue_in_rounds(narrow, *st);
}

// Early flush (data-parallel overlap): issue every job queued SO FAR — called from a tensor hook when the
// backward pass leaves the deep levels, whose layers carry 28 of the 30 MB of gradients — on the backward pass's own stream,
// and record the event the reducer's side stream waits for.  The engine callback stays registered: jobs queued afterwards
// (the encoder of levels 1-2) are issued by the ordinary flush when backward ends.  Returns the number of jobs issued.
int64_t flush_wgrads_early() {
    std::vector<PendingWgrad> q;
    c10::optional<c10::hip::HIPStream> st;
    {
        std::lock_guard<std::mutex> lock(g_wq_mu);
        q.swap(g_wq);
        st = g_wq_stream;
    }
    if (q.empty() || !st) return 0;
    const int64_t n = (int64_t)q.size();
    join_side_wgrads(st);
    issue_in_rounds(q, *st);
    if (g_ev_early && g_ev_early_dev != (int)st->device_index()) {
        hipEventDestroy(g_ev_early);
        g_ev_early = nullptr;
    }
    if (!g_ev_early) {
        c10::hip::HIPGuard dev_guard(st->device_index());
        TORCH_CHECK(hipEventCreateWithFlags(&g_ev_early, hipEventDisableTiming) == hipSuccess, "doda: hipEventCreate");
        g_ev_early_dev = (int)st->device_index();
    }
    TORCH_CHECK(hipEventRecord(g_ev_early, st->stream()) == hipSuccess, "doda: hipEventRecord");
    g_ev_early_valid = true;
    return n;
}

// true when the job was queued (the caller then returns no gradient for the weight)
bool try_defer_wgrad(const at::Tensor &features, const at::Tensor &dy, const at::Tensor &tbl, int64_t n_rows,
                     const at::Tensor &weight, const PairLists &pl) {
    if (!g_defer_wgrad || !weight.is_leaf() || weight.scalar_type() != at::kFloat || !weight.is_contiguous() ||
        n_rows <= 0)
        return false;
    const int task = torch::autograd::get_current_graph_task_id();
    if (task < 0) return false;   // not inside an engine run (e.g. a functional call of the node)
    std::lock_guard<std::mutex> lock(g_wq_mu);
    if (g_wq_task != task) {      // leftovers of a backward pass that never completed
        g_wq.clear();
        g_wq_callback = false;
        g_wq_task = task;
    }
    g_wq.push_back(PendingWgrad{features.contiguous(), dy, tbl, weight, pl, n_rows});
    g_wq_stream = c10::hip::getCurrentHIPStream(dy.device().index());
    if (!g_wq_callback) {
        g_wq_callback = true;
        torch::autograd::Engine::get_default_engine().queue_callback([]() { flush_wgrads(); });
    }
    return true;
}

// ---- autograd nodes --------------------------------------------------------------------------
// Hand-written torch::autograd::Node subclasses instead of torch::autograd::Function<T>: measured on
// the MI355X host, Function<T>::apply costs ~9 us per call on top of the launches (13-15 us for a
// one-launch BatchNorm call against 2.7 us per raw launch), and a U-Net step makes 136 such calls
// forward plus the same number of backward wrappers while being issue-bound on the host.
using torch::autograd::SavedVariable;
using torch::autograd::variable_list;

// Link between a fused BatchNorm(+ReLU) and the convolution that consumes its output (the BN -> ReLU ->
// conv triple of reference model/unet_block.py:23-30,46-49,67-79).  In backward the conv's data-grad
// kernel produces dz = d loss / d (BN output); the two sums the BatchNorm backward needs over dz
// (sum dz*mask, sum dz*mask*xhat) are accumulated in that kernel's store epilogue, and the BatchNorm node
// then runs final + apply only.  The link is made when the conv's input IS the tensor the BatchNorm just
// returned (same TensorImpl, still alive); the BatchNorm node uses the statistics only if the gradient it
// receives IS the tensor the conv produced — had the BN output a second consumer, autograd would hand
// over a sum (a different tensor) and the node falls back to its own statistics pass.
struct BNLink {
    at::Tensor x, mean, invstd, gamma, beta;   // the BatchNorm's input and per-channel tensors
    bool relu = false;
    c10::weak_intrusive_ptr<c10::TensorImpl> y;   // the BatchNorm's output
    at::Tensor stats;                             // [rows, 2, c] from the conv's data-grad epilogue
    // The gradient tensor those statistics belong to, held STRONGLY together with its version counter: with a
    // second consumer of the BatchNorm output the engine's InputBuffer would otherwise accumulate that
    // consumer's gradient IN PLACE into this very TensorImpl (use_count 1, GradMode off: old_var.add_(var)) —
    // same pointer, different values.  The extra reference forces an out-of-place sum (a different tensor), the
    // version check catches any other in-place edit, and a live reference rules out address reuse.
    at::Tensor dz;
    uint32_t dz_version = 0;
    BNLink() : y(c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>())) {}
};
thread_local std::shared_ptr<BNLink> g_last_bn;
bool g_bn_fusion = true;
// below: the one-launch BatchNorm kernels (csrc/bn.hip bn_small_*) instead of conv-epilogue statistics (DODA_STATS_MIN_ROWS, also
// read by spconv/functional.py: the two thresholds move together)
const int64_t BN_SMALL_ROWS = [] { const char *e = getenv("DODA_STATS_MIN_ROWS"); return e ? (int64_t)atoll(e) : (int64_t)4096; }();

// features, weight [k,k,k,Cin,Cout], fwd_tbl, bwd_tbl, n_out, bwd_layout, packed fwd / data-grad,
// optional residual (y = conv + residual; its gradient is the incoming gradient itself).
// Gradient edges: 0 features, 1 weight, 2 residual.
struct ConvNode : public torch::autograd::Node {
    SavedVariable features_, weight_;
    at::Tensor fwd_tbl, bwd_tbl, pk_bwd;
    PairLists pl;
    std::shared_ptr<BNLink> bn;   // the BatchNorm in front of this conv, if linked
    int64_t n_out = 0, bwd_layout = 0;
    bool direct_w = false;   // no edge to the weight: its gradient is deposited here (set_direct_grads)

    variable_list apply(variable_list &&grads) override {
        host_timing::Scope host_scope(1);
        const at::Tensor features = features_.unpack(), weight = weight_.unpack();
        const int64_t cin = weight.size(-2), cout = weight.size(-1), K = fwd_tbl.size(0);
        variable_list out(3);
        if (!grads[0].defined()) return out;
        const at::Tensor dy = grads[0].contiguous();  // reference fork patch llijiang/spconv@740a5b7
        if (task_should_compute_output(0)) {
            GatherEpi epi;
            if (bn && bn->x.defined() && bn->x.scalar_type() == dy.scalar_type() && bn->x.size(0) == features.size(0) &&
                bn->x.size(1) == cin) {
                epi.want = true;
                epi.bn_x = bn->x; epi.bn_mean = bn->mean; epi.bn_invstd = bn->invstd;
                epi.bn_gamma = bn->gamma; epi.bn_beta = bn->beta; epi.bn_relu = bn->relu;
            }
            out[0] = gather(dy, weight.reshape({K, cin, cout}),
                            pk_bwd.defined() ? c10::optional<at::Tensor>(pk_bwd) : c10::nullopt, bwd_tbl,
                            features.size(0), bwd_layout, cin, false, c10::nullopt, &epi);
            if (epi.stats.defined()) {
                bn->stats = epi.stats;
                bn->dz = out[0];
                bn->dz_version = out[0]._version();
            }
        }
        if (direct_w) {
            if (plain_accumulating_backward() && !try_defer_wgrad(features, dy, fwd_tbl, n_out, weight, pl))
                deposit_grad(weight, wgrad(features, dy, fwd_tbl, n_out, pl).reshape(weight.sizes()).to(weight.scalar_type()));
        } else if (task_should_compute_output(1) && !try_defer_wgrad(features, dy, fwd_tbl, n_out, weight, pl))
            out[1] = wgrad(features, dy, fwd_tbl, n_out, pl).reshape(weight.sizes()).to(weight.scalar_type());
        if (task_should_compute_output(2)) out[2] = grads[0];
        return out;
    }
    void release_variables() override {
        features_.reset_data();
        weight_.reset_data();
        fwd_tbl.reset();
        bwd_tbl.reset();
        pk_bwd.reset();
        pl = PairLists();
        bn.reset();
    }
    std::string name() const override { return "DodaIndiceConvBackward"; }
};

// want_stats: also return the (sum y, sum y^2) partials of the output for a BatchNorm that follows
// ([rows, 2, Cout]; undefined when the generic kernel ran).
std::vector<at::Tensor> indice_conv_impl(const at::Tensor &features, const at::Tensor &weight, const at::Tensor &fwd_tbl,
                       const at::Tensor &bwd_tbl, int64_t n_out, int64_t bwd_layout,
                       const c10::optional<at::Tensor> &pk_fwd, const c10::optional<at::Tensor> &pk_bwd,
                       const c10::optional<at::Tensor> &residual, const c10::optional<at::Tensor> &pair_in,
                       const c10::optional<at::Tensor> &pair_out, const c10::optional<at::Tensor> &pair_num,
                       const c10::optional<at::Tensor> &pair_seg, bool want_stats) {
    const at::Tensor res = residual.has_value() ? *residual : at::Tensor();
    const bool need_grad = at::GradMode::is_enabled() &&
                           (features.requires_grad() || weight.requires_grad() || (res.defined() && res.requires_grad()));
    const int64_t cin = weight.size(-2), cout = weight.size(-1), K = fwd_tbl.size(0);
    at::Tensor y;
    GatherEpi epi;
    epi.want = want_stats;
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        y = gather(features, weight.reshape({K, cin, cout}), pk_fwd, fwd_tbl, n_out, 0, cout, false, residual, &epi);
    }
    // the BatchNorm that produced `features` (if it was the last fused BatchNorm and its output is this tensor)
    std::shared_ptr<BNLink> link;
    if (g_last_bn) {
        auto alive = g_last_bn->y.lock();
        if (g_bn_fusion && alive && alive.get() == features.unsafeGetTensorImpl() && features.size(0) > BN_SMALL_ROWS)
            link = g_last_bn;
        g_last_bn.reset();
    }
    if (need_grad) {
        auto node = std::shared_ptr<ConvNode>(new ConvNode(), torch::autograd::deleteNode);
        node->bn = link;
        {
            torch::autograd::edge_list edges = torch::autograd::collect_next_edges(features, weight, res);
            if (direct_leaf(weight)) {
                edges[1] = torch::autograd::Edge();
                node->direct_w = true;
            }
            node->set_next_edges(std::move(edges));
        }
        node->features_ = SavedVariable(features, false);
        node->weight_ = SavedVariable(weight, false);
        node->fwd_tbl = fwd_tbl;
        node->bwd_tbl = bwd_tbl;
        if (pk_bwd.has_value() && pk_bwd->defined()) node->pk_bwd = *pk_bwd;
        node->n_out = n_out;
        node->bwd_layout = bwd_layout;
        if (pair_in.has_value() && pair_in->defined() && pair_out.has_value() && pair_out->defined()) {
            TORCH_CHECK(pair_in->scalar_type() == at::kInt && pair_in->dim() == 2 && pair_in->stride(1) == 1 &&
                        pair_out->scalar_type() == at::kInt && pair_out->sizes() == pair_in->sizes() &&
                        pair_out->stride(1) == 1 && pair_out->stride(0) == pair_in->stride(0) &&
                        pair_in->size(0) == fwd_tbl.size(0), "doda indice_conv: pair lists must be int32 [K, ld]");
            node->pl.in = *pair_in;
            node->pl.out = *pair_out;
            if (pair_num.has_value() && pair_num->defined()) node->pl.num = *pair_num;
            if (pair_seg.has_value() && pair_seg->defined()) {
                TORCH_CHECK(pair_seg->is_cuda() && pair_seg->scalar_type() == at::kInt && pair_seg->is_contiguous() &&
                            pair_seg->dim() == 2 && pair_seg->size(0) == fwd_tbl.size(0),
                            "doda indice_conv: the segment prefix must be device int32 [K, nt]");
                node->pl.seg = *pair_seg;
            }
        }
        torch::autograd::set_history(y, node);
    }
    return {y, epi.stats};
}

at::Tensor indice_conv(const at::Tensor &features, const at::Tensor &weight, const at::Tensor &fwd_tbl,
                       const at::Tensor &bwd_tbl, int64_t n_out, int64_t bwd_layout,
                       const c10::optional<at::Tensor> &pk_fwd, const c10::optional<at::Tensor> &pk_bwd,
                       const c10::optional<at::Tensor> &residual, const c10::optional<at::Tensor> &pair_in,
                       const c10::optional<at::Tensor> &pair_out, const c10::optional<at::Tensor> &pair_num,
                       const c10::optional<at::Tensor> &pair_seg) {
    return indice_conv_impl(features, weight, fwd_tbl, bwd_tbl, n_out, bwd_layout, pk_fwd, pk_bwd, residual, pair_in,
                            pair_out, pair_num, pair_seg, false)[0];
}

// The second gradient of a BatchNorm's input as the backward kernels take it (doda_bn_relu_bwd_stats / _bwd_add, `add_ld`): dense, or — without
// a copy — a column slice of a wider matrix whose rows lie `ld` elements apart: what torch.cat's backward hands to the
// skip connection of a U-Net level (reference model/unet_block.py:93).
at::Tensor add_operand(const at::Tensor &e, int64_t m, int64_t c, int64_t &ld) {
    ld = c;
    if (e.is_contiguous()) return e;
    const int esz = e.scalar_type() == at::kFloat ? 4 : 2;
    if (e.dim() == 2 && e.size(0) == m && e.size(1) == c && e.stride(1) == 1 && e.stride(0) >= c && e.stride(0) % 4 == 0 &&
        ((uintptr_t)e.data_ptr() % (size_t)(4 * esz)) == 0 && (m - 1) * e.stride(0) + c < ((int64_t)1 << 40)) {
        ld = e.stride(0);
        return e;
    }
    return e.contiguous();
}

// ---- fused BatchNorm1d(+ReLU).  Gradient edges: 0 x, 1 weight, 2 bias. ----------------------------
// With `passthrough` the op has a second output: an alias of x whose gradient (the skip connection of a
// pre-activation residual block) is summed into dx by the apply pass (doda_bn_relu_bwd_add) instead of
// by a separate accumulation kernel.
struct BNNode : public torch::autograd::Node {
    SavedVariable x_, weight_, bias_;
    at::Tensor mean, invstd;
    std::shared_ptr<BNLink> link;   // statistics may arrive from the consuming conv's data-grad epilogue
    bool training = true, relu = false;
    bool direct_p = false;   // no edges to gamma / beta: their gradients are deposited here (set_direct_grads)

    variable_list apply(variable_list &&grads) override {
        host_timing::Scope host_scope(2);
        const at::Tensor x = x_.unpack(), weight = weight_.unpack(), bias = bias_.unpack();
        variable_list out(3);
        at::Tensor extra;   // gradient of the pass-through alias
        if (grads.size() > 1 && grads[1].defined()) extra = grads[1];
        if (!grads[0].defined()) {
            if (extra.defined() && task_should_compute_output(0)) out[0] = extra;
            return out;
        }
        const at::Tensor dy = grads[0].contiguous();
        at::Tensor dx, dg, db;
        at::Tensor stats;
        // gamma / beta gradients straight into the reducer's bucket views when the parameters have homes (fp32 leaves
        // whose .grad is still undefined; at most once per backward pass): AccumulateGrad then binds the alias as .grad
        auto grad_out = [&](const at::Tensor &param, int64_t c) {
            at::Tensor h;
            if (param.is_leaf() && param.scalar_type() == at::kFloat && param.numel() == c && (!direct_p || plain_accumulating_backward()))
                h = take_grad_home(param, torch::autograd::get_current_graph_task_id());
            return h.defined() ? h : pempty({c}, x.options().dtype(at::kFloat));
        };
        if (link) {
            if (training && link->stats.defined() && link->dz.defined() &&
                link->dz.unsafeGetTensorImpl() == grads[0].unsafeGetTensorImpl() &&
                grads[0]._version() == link->dz_version && grads[0].is_contiguous() &&
                (!extra.defined() || extra.scalar_type() == x.scalar_type()))
                stats = link->stats;
            link->stats.reset();
            link->dz.reset();
        }
        if (stats.defined() && is_totals(stats, x.size(1)) && x.size(1) <= 256) {
            const int64_t m = x.size(0), c = x.size(1);
            int64_t add_ld = c;
            const at::Tensor add = extra.defined() ? add_operand(extra, m, c, add_ld) : at::Tensor();
            dx = pempty_like(x);
            dg = grad_out(weight, c);
            db = grad_out(bias, c);
            check(doda_bn_relu_bwd_totals(x.data_ptr(), dy.data_ptr(), (int)m, (int)c, elem_bytes(x), (const double *)stats.data_ptr(),
                                          (const float *)mean.data_ptr(), (const float *)invstd.data_ptr(),
                                          (const float *)weight.data_ptr(), (const float *)bias.data_ptr(), relu ? 1 : 0,
                                          add.defined() ? add.data_ptr() : nullptr, (int)add_ld, dx.data_ptr(),
                                          (float *)dg.data_ptr(), (float *)db.data_ptr(), stream_of(x)),
                  "doda_bn_relu_bwd_totals");
            extra = at::Tensor();
        } else if (stats.defined() && stats.scalar_type() == at::kFloat) {
            const int64_t m = x.size(0), c = x.size(1);
            int64_t add_ld = c;
            const at::Tensor add = extra.defined() ? add_operand(extra, m, c, add_ld) : at::Tensor();
            dx = pempty_like(x);
            dg = grad_out(weight, c);
            db = grad_out(bias, c);
            at::Tensor coef = pempty({3 * c}, x.options().dtype(at::kFloat));
            check(doda_bn_relu_bwd_stats(x.data_ptr(), dy.data_ptr(), (int)m, (int)c, elem_bytes(x),
                                            (const float *)stats.data_ptr(), (int)stats.size(0),
                                            (const float *)mean.data_ptr(), (const float *)invstd.data_ptr(),
                                            (const float *)weight.data_ptr(), (const float *)bias.data_ptr(), relu ? 1 : 0,
                                            add.defined() ? add.data_ptr() : nullptr, (int)add_ld, dx.data_ptr(),
                                            (float *)dg.data_ptr(), (float *)db.data_ptr(), (float *)coef.data_ptr(),
                                            stream_of(x)),
                  "doda_bn_relu_bwd_stats");
            extra = at::Tensor();
        } else if (training && extra.defined() && extra.scalar_type() == x.scalar_type()) {
            const int64_t m = x.size(0), c = x.size(1);
            int64_t add_ld = c;
            const at::Tensor add = add_operand(extra, m, c, add_ld);
            dx = pempty_like(x);
            dg = grad_out(weight, c);
            db = grad_out(bias, c);
            const size_t wsb = doda_bn_workspace_bytes((int)m, (int)c);
            at::Tensor ws = pempty({(int64_t)wsb}, x.options().dtype(at::kByte));
            check(doda_bn_relu_bwd_add(x.data_ptr(), dy.data_ptr(), (int)m, (int)c, elem_bytes(x),
                                          (const float *)mean.data_ptr(), (const float *)invstd.data_ptr(),
                                          (const float *)weight.data_ptr(), (const float *)bias.data_ptr(),
                                          relu ? 1 : 0, add.data_ptr(), (int)add_ld, dx.data_ptr(), (float *)dg.data_ptr(),
                                          (float *)db.data_ptr(), ws.data_ptr(), wsb, stream_of(x)),
                  "doda_bn_relu_bwd_add");
            extra = at::Tensor();
        } else if (training) {
            const int64_t m = x.size(0), c = x.size(1);
            dx = pempty_like(x);
            dg = grad_out(weight, c);
            db = grad_out(bias, c);
            const size_t wsb = doda_bn_workspace_bytes((int)m, (int)c);
            at::Tensor ws = pempty({(int64_t)wsb}, x.options().dtype(at::kByte));
            check(doda_bn_relu_bwd(x.data_ptr(), dy.data_ptr(), (int)m, (int)c, elem_bytes(x),
                                   (const float *)mean.data_ptr(), (const float *)invstd.data_ptr(),
                                   (const float *)weight.data_ptr(), (const float *)bias.data_ptr(), relu ? 1 : 0,
                                   dx.data_ptr(), (float *)dg.data_ptr(), (float *)db.data_ptr(), ws.data_ptr(), wsb,
                                   stream_of(x)),
                  "doda_bn_relu_bwd");
        } else {  // running statistics are constants
            at::Tensor xh = (x.to(at::kFloat) - mean) * invstd;
            at::Tensor dz = dy.to(at::kFloat);
            if (relu) dz = dz * ((xh * weight + bias) > 0);
            dx = (dz * (weight * invstd)).to(x.scalar_type());
            dg = (dz * xh).sum(0);
            db = dz.sum(0);
        }
        if (extra.defined()) dx = dx + extra;
        if (g_nancheck || g_fplog) {
            nan_check(dy, "bn backward", "INPUT dy", x.size(0), x.size(1));
            nan_check(dx, "bn backward", "dx", x.size(0), x.size(1), stats.defined() ? (stats.scalar_type() == at::kDouble ? 2 : 1) : 0);
            nan_check(dg, "bn backward", "dgamma", x.size(0), x.size(1));
            nan_check(db, "bn backward", "dbeta", x.size(0), x.size(1));
        }
        if (task_should_compute_output(0)) out[0] = dx;
        if (direct_p) {
            if (plain_accumulating_backward()) {
                deposit_grad(weight, dg);
                deposit_grad(bias, db);
            }
        } else {
            if (task_should_compute_output(1)) out[1] = dg.scalar_type() == weight.scalar_type() ? dg : dg.to(weight.scalar_type());
            if (task_should_compute_output(2)) out[2] = db.scalar_type() == bias.scalar_type() ? db : db.to(bias.scalar_type());
        }
        return out;
    }
    void release_variables() override {
        x_.reset_data();
        weight_.reset_data();
        bias_.reset_data();
        mean.reset();
        invstd.reset();
        link.reset();
    }
    std::string name() const override { return "DodaBNReLUBackward"; }
};

// stats: (sum x, sum x^2) partials of x from the epilogue of the conv that produced it, or undefined
std::vector<at::Tensor> bn_relu_impl(const at::Tensor &x_in, const at::Tensor &weight, const at::Tensor &bias,
                                     const at::Tensor &running_mean, const at::Tensor &running_var,
                                     const at::Tensor &nbt, bool training, double momentum, double eps,
                                     bool relu, bool passthrough, const at::Tensor &stats = at::Tensor(),
                                     const at::Tensor &stats_b = at::Tensor()) {
    const bool need_grad = at::GradMode::is_enabled() &&
                           (x_in.requires_grad() || weight.requires_grad() || bias.requires_grad());
    at::Tensor x, y, mean, invstd, xp;
    {
        at::AutoDispatchBelowADInplaceOrView guard;
        x = x_in.contiguous();
        if (passthrough) xp = x.alias();
        const int esz = elem_bytes(x);
        const int64_t m = x.size(0), c = x.size(1);
        y = pempty_like(x);
        if (training) {
            mean = pempty({c}, x.options().dtype(at::kFloat));
            invstd = pempty({c}, x.options().dtype(at::kFloat));
        } else {
            mean = running_mean.to(at::kFloat).contiguous();
            invstd = at::rsqrt(running_var.to(at::kFloat) + eps).contiguous();
        }
        if (training && m > BN_SMALL_ROWS && c <= 256 && stats.defined() && stats.scalar_type() == at::kDouble &&
            (stats_b.defined() ? (stats.dim() == 4 && stats_b.dim() == 4 && is_totals(stats, stats.size(2) * 4) &&
                                  is_totals(stats_b, c - stats.size(2) * 4) && stats.size(2) > 0 && stats.size(2) * 4 < c)
                               : is_totals(stats, c))) {
            // ABI 9: statistics as totals — ONE launch, also for a channel concatenation [a | b] (two producers)
            check(doda_bn_relu_fwd_totals(x.data_ptr(), (int)m, (int)c, esz, (const double *)stats.data_ptr(),
                                          stats_b.defined() ? (const double *)stats_b.data_ptr() : nullptr, (int)stats.size(2) * 4,
                                          (float)eps, (float)momentum, (const float *)weight.data_ptr(),
                                          (const float *)bias.data_ptr(), (float *)running_mean.data_ptr(),
                                          (float *)running_var.data_ptr(), nbt.defined() ? (int64_t *)nbt.data_ptr() : nullptr,
                                          relu ? 1 : 0, y.data_ptr(), (float *)mean.data_ptr(), (float *)invstd.data_ptr(),
                                          stream_of(x)),
                  "doda_bn_relu_fwd_totals");
        } else if (training && stats.defined() && stats_b.defined() && m > BN_SMALL_ROWS && stats.dim() == 3 &&
            stats_b.dim() == 3 && stats.size(2) + stats_b.size(2) == c && stats.scalar_type() == at::kFloat &&
            stats_b.scalar_type() == at::kFloat && stats.is_contiguous() && stats_b.is_contiguous() && stats.size(2) % 4 == 0) {
            // x is a channel concatenation [a | b] (the U-Net level's skip + upsampled features): BatchNorm statistics are
            // per channel, so the two halves' statistics rows — from the epilogues of the two convs that produced them —
            // are reduced separately into the halves of the per-channel vectors, and one apply pass follows: the
            // standalone statistics sweep over the concatenated tensor is gone
            const int64_t ca = stats.size(2), cb = stats_b.size(2);
            float *mp = (float *)mean.data_ptr(), *ip = (float *)invstd.data_ptr();
            float *rmp = (float *)running_mean.data_ptr(), *rvp = (float *)running_var.data_ptr();
            check(doda_bn_fwd_final((const float *)stats.data_ptr(), (int)stats.size(0), (int)m, (int)ca, (float)eps,
                                    (float)momentum, rmp, rvp, nbt.defined() ? (int64_t *)nbt.data_ptr() : nullptr, mp, ip,
                                    stream_of(x)), "doda_bn_fwd_final");
            check(doda_bn_fwd_final((const float *)stats_b.data_ptr(), (int)stats_b.size(0), (int)m, (int)cb, (float)eps,
                                    (float)momentum, rmp + ca, rvp + ca, nullptr, mp + ca, ip + ca, stream_of(x)),
                  "doda_bn_fwd_final");
            check(doda_bn_relu_apply(x.data_ptr(), (int)m, (int)c, esz, mp, ip, (const float *)weight.data_ptr(),
                                     (const float *)bias.data_ptr(), relu ? 1 : 0, y.data_ptr(), stream_of(x)),
                  "doda_bn_relu_apply");
        } else if (training && stats.defined() && m > BN_SMALL_ROWS && stats.dim() == 3 && stats.size(2) == c &&
            stats.scalar_type() == at::kFloat && stats.is_contiguous()) {
            check(doda_bn_relu_fwd_stats(x.data_ptr(), (int)m, (int)c, esz, (const float *)stats.data_ptr(),
                                         (int)stats.size(0), (float)eps, (float)momentum,
                                         (const float *)weight.data_ptr(), (const float *)bias.data_ptr(),
                                         (float *)running_mean.data_ptr(), (float *)running_var.data_ptr(),
                                         nbt.defined() ? (int64_t *)nbt.data_ptr() : nullptr, relu ? 1 : 0, y.data_ptr(),
                                         (float *)mean.data_ptr(), (float *)invstd.data_ptr(), stream_of(x)),
                  "doda_bn_relu_fwd_stats");
        } else {
        const size_t wsb = doda_bn_workspace_bytes((int)m, (int)c);
        at::Tensor ws = pempty({(int64_t)wsb}, x.options().dtype(at::kByte));
        check(doda_bn_relu_fwd(x.data_ptr(), (int)m, (int)c, esz, (float)eps, (float)momentum,
                               (const float *)weight.data_ptr(), (const float *)bias.data_ptr(),
                               training ? (float *)running_mean.data_ptr() : nullptr,
                               training ? (float *)running_var.data_ptr() : nullptr,
                               training && nbt.defined() ? (int64_t *)nbt.data_ptr() : nullptr, training ? 1 : 0,
                               relu ? 1 : 0, y.data_ptr(), (float *)mean.data_ptr(), (float *)invstd.data_ptr(),
                               ws.data_ptr(), wsb, stream_of(x)),
              "doda_bn_relu_fwd");
        }
        if (g_nancheck || g_fplog) {
            nan_check(x, "bn forward", "INPUT x", x.size(0), x.size(1));
            nan_check(y, "bn forward", "y", x.size(0), x.size(1), stats.defined() ? (stats.scalar_type() == at::kDouble ? 2 : 1) : 0);
            nan_check(mean, "bn forward", "mean", x.size(0), x.size(1));
            nan_check(invstd, "bn forward", "invstd", x.size(0), x.size(1));
        }
    }
    g_last_bn.reset();
    if (need_grad) {
        auto node = std::shared_ptr<BNNode>(new BNNode(), torch::autograd::deleteNode);
        if (training && weight.scalar_type() == at::kFloat && bias.scalar_type() == at::kFloat) {
            auto link = std::make_shared<BNLink>();
            link->x = x;
            link->mean = mean; link->invstd = invstd;
            link->gamma = weight.detach(); link->beta = bias.detach();
            link->relu = relu;
            link->y = c10::weak_intrusive_ptr<c10::TensorImpl>(y.getIntrusivePtr());
            node->link = link;
            g_last_bn = link;
        }
        {
            torch::autograd::edge_list edges = torch::autograd::collect_next_edges(x_in, weight, bias);
            if (training && direct_leaf(weight) && direct_leaf(bias)) {
                edges[1] = torch::autograd::Edge();
                edges[2] = torch::autograd::Edge();
                node->direct_p = true;
            }
            node->set_next_edges(std::move(edges));
        }
        node->x_ = SavedVariable(x_in.is_contiguous() ? x_in : x, false);
        node->weight_ = SavedVariable(weight, false);
        node->bias_ = SavedVariable(bias, false);
        node->mean = mean;
        node->invstd = invstd;
        node->training = training;
        node->relu = relu;
        torch::autograd::set_history(y, node);
        if (passthrough) torch::autograd::set_history(xp, node);   // output 1
    }
    if (passthrough) return {y, xp};
    return {y};
}

at::Tensor bn_relu(const at::Tensor &x, const at::Tensor &weight, const at::Tensor &bias,
                   const at::Tensor &running_mean, const at::Tensor &running_var, const at::Tensor &nbt,
                   bool training, double momentum, double eps, bool relu, const c10::optional<at::Tensor> &stats,
                   const c10::optional<at::Tensor> &stats_b) {
    return bn_relu_impl(x, weight, bias, running_mean, running_var, nbt, training, momentum, eps, relu, false,
                        stats.has_value() ? *stats : at::Tensor(), stats_b.has_value() ? *stats_b : at::Tensor())[0];
}

// (y, x_alias): use x_alias wherever the block needs x again (its gradient is summed inside this op's backward)
std::vector<at::Tensor> bn_relu_pass(const at::Tensor &x, const at::Tensor &weight, const at::Tensor &bias,
                                     const at::Tensor &running_mean, const at::Tensor &running_var,
                                     const at::Tensor &nbt, bool training, double momentum, double eps, bool relu,
                                     const c10::optional<at::Tensor> &stats, const c10::optional<at::Tensor> &stats_b) {
    return bn_relu_impl(x, weight, bias, running_mean, running_var, nbt, training, momentum, eps, relu, true,
                        stats.has_value() ? *stats : at::Tensor(), stats_b.has_value() ? *stats_b : at::Tensor());
}

// ---- a whole pre-activation residual block in ONE extension call ------------------------------------
// reference model/unet_block.py:14-37:  y = conv2(relu(bn2(conv1(relu(bn1(x)))))) + skip,  skip = x or i_branch(x).
// The same four ops (and the same autograd nodes, BatchNorm links, epilogue statistics) the Python modules issue one
// by one — bn_relu_impl / indice_conv_impl — without ~50 us of interpreter work between them: the step is paced by
// the issuing thread as much as by the GPU (tools/modeprobe.py).
//   bn1, bn2 : {gamma, beta, running_mean, running_var, num_batches_tracked}
//   cv1, cv2 : {weight [3,3,3,Cin,Cout], packed forward, packed data-grad}   (packed: may be None)
//   rb       : {table (SubM: forward and data-grad), pair_in, pair_out, pair_num, pair_seg}   (lists: may be None)
//   skip     : features of the 1x1 branch, or None for the identity skip (its gradient is then summed inside bn1's
//              backward kernel: bn_relu_pass)
// Returns {y, statistics of y or undefined}.
std::vector<at::Tensor> residual_block(const at::Tensor &x, const c10::optional<at::Tensor> &stats_in,
                                       const std::vector<at::Tensor> &bn1, const std::vector<at::Tensor> &bn2,
                                       bool training, double momentum1, double eps1, double momentum2, double eps2,
                                       const std::vector<c10::optional<at::Tensor>> &cv1,
                                       const std::vector<c10::optional<at::Tensor>> &cv2,
                                       const std::vector<c10::optional<at::Tensor>> &rb, int64_t n_out,
                                       const c10::optional<at::Tensor> &skip, bool want_stats,
                                       const std::vector<c10::optional<at::Tensor>> &sc = {},
                                       const c10::optional<at::Tensor> &stats_in_b = c10::nullopt) {
    host_timing::Scope host_scope(0);
    TORCH_CHECK(bn1.size() == 5 && bn2.size() == 5 && cv1.size() == 3 && cv2.size() == 3 && rb.size() == 5 &&
                cv1[0].has_value() && cv2[0].has_value() && rb[0].has_value(), "doda residual_block: bad argument lists");
    // sc = {weight [1,1,1,Cin,Cout], packed forward, packed data-grad, identity table, identity table as the pair lists or
    // None}: the block's 1x1 skip convolution (reference model/unet_block.py:18-21), run HERE on the first BatchNorm's
    // pass-through alias of x, so that its data gradient is summed inside that BatchNorm's backward kernel instead of
    // by an autograd accumulation kernel (x has two consumers: the BatchNorm and the skip)
    const bool conv_skip = sc.size() == 5 && sc[0].has_value() && sc[3].has_value();
    TORCH_CHECK(!(conv_skip && skip.has_value() && skip->defined()), "doda residual_block: skip features AND a skip convolution");
    const bool identity = !conv_skip && !(skip.has_value() && skip->defined());
    const at::Tensor &tbl = *rb[0];
    const at::Tensor st1 = stats_in.has_value() ? *stats_in : at::Tensor();
    auto a = bn_relu_impl(x, bn1[0], bn1[1], bn1[2], bn1[3], bn1[4], training, momentum1, eps1, true,
                          (identity || conv_skip) && training, st1, stats_in_b.has_value() ? *stats_in_b : at::Tensor());

    auto z1 = indice_conv_impl(a[0], *cv1[0], tbl, tbl, n_out, 2, cv1[1], cv1[2], c10::nullopt, rb[1], rb[2], rb[3], rb[4],
                               want_stats);
    at::Tensor skip_feats;   // (after conv1: the conv that follows a BatchNorm op picks up its statistics link)
    if (conv_skip)
        skip_feats = indice_conv_impl(training ? a[1] : x, *sc[0], *sc[3], *sc[3], x.size(0), 1, sc[1], sc[2], c10::nullopt,
                                      sc[4], sc[4], c10::nullopt, c10::nullopt, false)[0];
    auto y2 = bn_relu_impl(z1[0], bn2[0], bn2[1], bn2[2], bn2[3], bn2[4], training, momentum2, eps2, true, false, z1[1]);
    const at::Tensor res = conv_skip ? skip_feats : identity ? (training ? a[1] : x) : *skip;
    return indice_conv_impl(y2[0], *cv2[0], tbl, tbl, n_out, 2, cv2[1], cv2[2], res, rb[1], rb[2], rb[3], rb[4], want_stats);
}


// ---- coarse levels as one call: op-list glue (csrc/layers.hip, doda_layers_run) -------------------------------------------
// reference model/unet_block.py:55-100: one UBlock subtree (blocks -> strided conv -> UBlock -> inverse conv ->
// concatenation -> blocks_tail, ResidualBlocks of model/unet_block.py:9-37 inside) whose levels hold a few thousand rows
// and fewer, as ONE extension call forward and ONE autograd node backward.  The Python side (doda_amd/model.py:
// UBlock._forward_coarse) flattens the subtree into STEPS; this file expands them into ops (include/doda_hip.h doda_cx_op), owns
// the intermediate tensors and the autograd node, and hands each list to doda_layers_run, which issues the per-layer kernels back
// to back.  No arithmetic here: every op is a kernel of libdoda_hip.so.
//   step RB   (kind 0): tensors {subm table, bn1 x5, conv1 x3, bn2 x5, conv2 x3, skip conv x3 | None x3, identity table | None}
//                       scalars {eps1, momentum1, eps2, momentum2}
//   step DOWN (kind 1): tensors {child [8, m], par_off [8, n], bn x5, conv x3}        scalars {eps, momentum, m}
//   step UP   (kind 2): tensors {par_off [8, n], child [8, m], bn x5, conv x3}        scalars {eps, momentum, n}
// bn x5 = gamma, beta, running_mean, running_var, num_batches_tracked; conv x3 = weight, packed forward, packed data-grad.
namespace coarse {

typedef std::vector<c10::optional<at::Tensor>> TList;

struct Arena {   // device scratch that is never handed out as a tensor: a few chunks instead of a hundred allocations
    std::vector<at::Tensor> chunks;
    char *cur = nullptr;
    size_t left = 0;
    at::TensorOptions opt;
    void *alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > left) {
            const size_t sz = bytes > ((size_t)2 << 20) ? bytes : ((size_t)2 << 20);
            chunks.push_back(pempty({(int64_t)sz}, opt));
            cur = (char *)chunks.back().data_ptr();
            left = sz;
        }
        void *p = cur;
        cur += bytes;
        left -= bytes;
        return p;
    }
    float *floats(size_t n) { return (float *)alloc(n * 4); }
};

struct Val {   // a bf16 [rows, c] matrix: dense, or a column slice of a wider one (ld > c)
    void *p = nullptr;
    int rows = 0, c = 0, ld = 0;
    at::Tensor t;                 // the dense tensor holding it, when there is one (operands of the weight gradients)
    float *stats = nullptr;       // (sum, sum of squares) partial rows of columns [0, c_split)
    float *stats_b = nullptr;     // ... of columns [c_split, c): a concatenation
    int c_split = 0;
};

struct BNP { at::Tensor gamma, beta, rm, rv, nbt; float eps = 0.f, mom = 0.f; };
struct CVP { at::Tensor weight, pk_fwd, pk_bwd; };

struct Layer {   // BatchNorm -> ReLU -> conv, as the backward pass needs it
    BNP bn;
    CVP cv;
    Val x;                      // the BatchNorm's input
    at::Tensor a;               // its output = the conv's input (dense: operand of the weight gradient)
    float *mean = nullptr, *invstd = nullptr;
    at::Tensor fwd_tbl, bwd_tbl;
    int K = 0, n_in = 0, n_out = 0, c_in = 0, c_out = 0;
};

struct Step {
    int kind = 0;
    Layer l1, l2;               // RB: both; DOWN / UP: l1
    bool has_skip = false;
    CVP skip;
    at::Tensor ident;           // identity table of the 1x1 skip conv (weight-gradient job)
    bool skip_pairs = false;    // ... which doubles as both pair lists of the pair-list weight-gradient kernel (large bf16 levels)
};

inline BNP bn_of(const TList &t, size_t at_, double eps, double mom) {
    BNP b;
    b.gamma = *t[at_]; b.beta = *t[at_ + 1]; b.rm = *t[at_ + 2]; b.rv = *t[at_ + 3];
    if (t[at_ + 4].has_value() && t[at_ + 4]->defined()) b.nbt = *t[at_ + 4];
    b.eps = (float)eps; b.mom = (float)mom;
    TORCH_CHECK(b.gamma.scalar_type() == at::kFloat && b.beta.scalar_type() == at::kFloat && b.rm.scalar_type() == at::kFloat &&
                b.rv.scalar_type() == at::kFloat && b.gamma.is_contiguous() && b.beta.is_contiguous() && b.rm.is_contiguous() &&
                b.rv.is_contiguous(), "doda coarse: BatchNorm vectors must be contiguous fp32");
    return b;
}
inline CVP cv_of(const TList &t, size_t at_) {
    CVP c;
    TORCH_CHECK(t[at_].has_value() && t[at_ + 1].has_value() && t[at_ + 2].has_value(), "doda coarse: conv needs weight + packed copies");
    c.weight = *t[at_]; c.pk_fwd = *t[at_ + 1]; c.pk_bwd = *t[at_ + 2];
    return c;
}

// The op list of a pass (include/doda_hip.h doda_cx_op), issued by doda_layers_run as whole-chip per-layer launches: fp64 totals as
// statistics, BatchNorm ops folded into the consuming convolution's gather.  (ABI 8-10 had a second backend, a persistent
// single-XCD executor; removed in ABI 11.)
struct Builder {
    std::vector<doda_cx_op> ops;
    Arena arena;
    at::TensorOptions bf;
    bool first = true;
    int esz = 2;
    std::vector<at::Tensor> tot_keep;   // totals slices of this pass (zeroed arena of stats_totals_take)
    int launches = 0;

    doda_cx_op &push(int kind, int flags) {
        ops.emplace_back();
        doda_cx_op &o = ops.back();
        memset(&o, 0, sizeof(o));
        o.kind = kind;
        o.flags = flags;
        return o;
    }
    // statistics of a c-channel tensor: zeroed fp64 totals, as the op field's float *
    float *stat_buf(int c) {
        at::Tensor t = stats_totals_take(c, bf);
        tot_keep.push_back(t);
        return (float *)t.data_ptr();
    }
    Val dense(int rows, int c, bool as_tensor) {
        Val v;
        v.rows = rows; v.c = c; v.ld = c;
        if (as_tensor) { v.t = pempty({rows, c}, bf); v.p = v.t.data_ptr(); }
        else v.p = arena.alloc((size_t)rows * c * esz);
        return v;
    }
    // `cut` > 0: the ops in front of it are issued, `between` runs on the host (the rulebook prefetcher's gate: an event recorded
    // where the forward pass enters its launch-floor levels), then the rest
    void run(const at::Tensor &like, size_t cut = 0, const std::function<void()> &between = nullptr) {
        if (ops.empty()) return;
        int32_t n = 0, n2 = 0;
        if (cut > 0 && cut < ops.size() && between) {
            check(doda_layers_run(ops.data(), (int32_t)cut, esz, &n, stream_of(like)), "doda_layers_run");
            between();
            check(doda_layers_run(ops.data() + cut, (int32_t)(ops.size() - cut), esz, &n2, stream_of(like)), "doda_layers_run");
        } else {
            check(doda_layers_run(ops.data(), (int32_t)ops.size(), esz, &n, stream_of(like)), "doda_layers_run");
        }
        launches = n + n2;
    }

    // ---- forward ----
    void stats_of(Val &x) {   // a tensor that arrives without statistics (the executor's input)
        x.stats = stat_buf(x.c);
        x.stats_b = nullptr;
        x.c_split = x.c;
        doda_cx_op &o = push(DODA_CX_STATS, first ? 0 : DODA_CX_F_BARRIER);
        o.rows = x.rows; o.c_in = x.c; o.x_ld = x.ld; o.x = x.p; o.stats = x.stats;
        first = false;
    }
    void bn_fwd(Layer &L, bool training) {
        Val &x = L.x;
        if (training && !x.stats) stats_of(x);
        L.a = pempty({x.rows, x.c}, bf);
        doda_cx_op &o = push(DODA_CX_BNFWD, (first ? 0 : DODA_CX_F_BARRIER) | DODA_CX_F_RELU | (training ? DODA_CX_F_TRAINING : 0));
        first = false;
        o.rows = x.rows; o.c_in = x.c; o.x_ld = x.ld; o.y_ld = x.c; o.x = x.p; o.y = L.a.data_ptr();
        o.eps = L.bn.eps; o.momentum = L.bn.mom;
        o.gamma = (const float *)L.bn.gamma.data_ptr(); o.beta = (const float *)L.bn.beta.data_ptr();
        o.running_mean = (float *)L.bn.rm.data_ptr(); o.running_var = (float *)L.bn.rv.data_ptr();
        if (training) {
            L.mean = arena.floats(x.c); L.invstd = arena.floats(x.c);
            o.mean = L.mean; o.invstd = L.invstd;
            o.stats = x.stats; o.stats_b = x.stats_b; o.c_split = x.c_split > 0 ? x.c_split : x.c;
            o.nbt = L.bn.nbt.defined() ? (int64_t *)L.bn.nbt.data_ptr() : nullptr;
        } else {
            // evaluation: the running statistics; backward (running statistics as constants) is not the executor's business
            o.c_split = x.c;
        }
    }
    // y = conv(L.a) (+ res), statistics of y when `want_stats`; `out` preset (p / ld) = where y goes
    // (identity: the 1x1 convolution = K = 1 over the identity table `tbl`)
    void gemm_fwd(Layer &L, const at::Tensor &tbl, bool identity, Val &out, const Val *res, bool want_stats, bool barrier) {
        doda_cx_op &o = push(DODA_CX_GEMM, (barrier ? DODA_CX_F_BARRIER : 0) | (identity ? DODA_CX_F_IDENTITY : 0));
        first = false;
        o.rows = out.rows; o.rows_in = (int32_t)L.a.size(0); o.c_in = (int32_t)L.a.size(1); o.c_out = out.c;
        o.K = identity ? 1 : (int32_t)tbl.size(0);
        o.tbl = (const int32_t *)tbl.data_ptr();
        o.tbl_ld = identity ? out.rows : (int32_t)tbl.size(1);
        o.x = L.a.data_ptr(); o.x_ld = (int32_t)L.a.size(1);
        o.w = L.cv.pk_fwd.data_ptr();
        o.y = out.p; o.y_ld = out.ld;
        if (!identity) o.tilebook = tilebook_behind(tbl, out.rows);   // (levels 1-2: the LDS-staged kernels)
        if (res) { o.res = res->p; o.res_ld = res->ld; }
        if (want_stats) {
            out.stats = stat_buf(out.c);
            out.stats_b = nullptr;
            out.c_split = out.c;
            o.stats = out.stats;
        }
    }
};

int g_coarse_launches_fwd = 0, g_coarse_launches_bwd = 0;   // launches of the last per-layer forward / backward list (tests, tools)

struct CoarseNode : public torch::autograd::Node {
    std::vector<Step> steps;
    std::vector<at::Tensor> keep;      // arena chunks and tensors the saved pointers refer to
    at::Tensor x_in;                   // (kept for its size / options)

    // gamma / beta gradient targets: the reducer's bucket views when the parameters have homes, existing .grad tensors
    // (accumulate) or fresh tensors deposited afterwards
    struct PGrad { at::Tensor param, buf; bool accum = false, fresh = false; };

    variable_list apply(variable_list &&grads) override {
        host_timing::Scope host_scope(5);
        variable_list out(1);
        if (!grads[0].defined()) return out;
        at::Tensor g = grads[0].contiguous();
        TORCH_CHECK(g.scalar_type() == x_in.scalar_type(), "doda coarse: gradient dtype");
        Builder B;
        B.esz = elem_bytes(g);
        B.bf = g.options();
        B.arena.opt = g.options().dtype(at::kByte);
        const int task = torch::autograd::get_current_graph_task_id();
        std::vector<PGrad> pgrads;
        const bool plain = plain_accumulating_backward();
        auto pgrad = [&](const at::Tensor &param, int64_t c, int &flags) -> float * {
            PGrad pg;
            pg.param = param;
            at::Tensor cur = param.grad();
            if (!plain) {   // a pass restricted to specific inputs: the kernel's gamma / beta sums go to scratch
                pg.buf = pempty({c}, param.options().dtype(at::kFloat));
            } else if (cur.defined() && cur.scalar_type() == at::kFloat && cur.is_contiguous() && cur.numel() == c) {
                pg.buf = cur; pg.accum = true; flags |= DODA_CX_F_ACCUM;
            } else {
                at::Tensor h;
                if (param.is_leaf() && param.scalar_type() == at::kFloat && param.numel() == c) h = take_grad_home(param, task);
                pg.buf = h.defined() ? h : pempty({c}, param.options().dtype(at::kFloat));
                pg.fresh = true;
            }
            pgrads.push_back(pg);
            return (float *)pg.buf.data_ptr();
        };
        struct WJob { at::Tensor a, b, tbl, weight; int64_t n_rows; PairLists pl; };
        std::vector<WJob> wjobs;
        bool first = true;
        // dz = data gradient of L's conv applied to `dy`, masked by L's ReLU; statistics for L's BatchNorm backward
        auto gemm_bwd = [&](const Layer &L, const at::Tensor &dy, bool identity, float *&st, bool barrier) -> void * {
            void *dz = B.arena.alloc((size_t)L.n_in * L.c_in * B.esz);
            st = B.stat_buf(L.c_in);
            doda_cx_op &o = B.push(DODA_CX_GEMM, (barrier && !first ? DODA_CX_F_BARRIER : 0) | DODA_CX_F_RELU | (identity ? DODA_CX_F_IDENTITY : 0));
            first = false;
            o.rows = L.n_in; o.rows_in = L.n_out; o.c_in = L.c_out; o.c_out = L.c_in;
            o.K = identity ? 1 : (int32_t)L.bwd_tbl.size(0);
            o.tbl = identity ? nullptr : (const int32_t *)L.bwd_tbl.data_ptr();
            o.tbl_ld = identity ? L.n_in : (int32_t)L.bwd_tbl.size(1);
            if (!identity) o.tilebook = tilebook_behind(L.bwd_tbl, L.n_in);
            o.x = dy.data_ptr(); o.x_ld = L.c_out;
            o.w = L.cv.pk_bwd.data_ptr();
            o.y = dz; o.y_ld = L.c_in;
            o.aux = L.x.p; o.aux_ld = L.x.ld;
            o.mean = L.mean; o.invstd = L.invstd;
            o.gamma = (const float *)L.bn.gamma.data_ptr(); o.beta = (const float *)L.bn.beta.data_ptr();
            o.stats = st;
            return dz;
        };
        // gradient of L's BatchNorm input from dz (+ add); a concatenated input splits into two dense tensors
        auto bn_bwd = [&](const Layer &L, void *dz, float *st, const at::Tensor &add, at::Tensor &left, at::Tensor &right) {
            int flags = DODA_CX_F_BARRIER;
            const int c = L.c_in;
            const int split = (L.x.c_split > 0 && L.x.c_split < c) ? L.x.c_split : c;
            left = pempty({L.n_in, split}, B.bf);
            keep_alive.push_back(left);   // (every tensor an op points at stays alive until the launch is queued: a freed block
            //                               could be handed out again by the next at::empty of this very pass)
            if (split < c) { right = pempty({L.n_in, c - split}, B.bf); keep_alive.push_back(right); }
            float *dg = nullptr, *db = nullptr;
            {
                int f1 = 0, f2 = 0;
                dg = pgrad(L.bn.gamma, c, f1);
                db = pgrad(L.bn.beta, c, f2);
                if ((f1 != 0) != (f2 != 0)) {   // one accumulates, the other does not: give both fresh buffers and add afterwards
                    PGrad &pa = pgrads[pgrads.size() - 2], &pb = pgrads.back();
                    if (pa.accum) { pa.buf = pempty({c}, pa.param.options().dtype(at::kFloat)); pa.accum = false; pa.fresh = true; dg = (float *)pa.buf.data_ptr(); }
                    if (pb.accum) { pb.buf = pempty({c}, pb.param.options().dtype(at::kFloat)); pb.accum = false; pb.fresh = true; db = (float *)pb.buf.data_ptr(); }
                } else if (f1) flags |= DODA_CX_F_ACCUM;
            }
            flags |= DODA_CX_F_RELU;   // (the data-grad kernels store dz unmasked: the op masks, and needs beta)
            doda_cx_op &o = B.push(DODA_CX_BNBWD, flags);
            o.beta = (const float *)L.bn.beta.data_ptr();
            o.rows = L.n_in; o.c_in = c; o.c_split = split;
            o.x = dz; o.x_ld = c;
            o.aux = L.x.p; o.aux_ld = L.x.ld;
            if (add.defined()) { o.res = add.data_ptr(); o.res_ld = (int32_t)add.size(1); }
            o.y = left.data_ptr(); o.y_ld = split;
            if (split < c) { o.y2 = right.data_ptr(); o.y2_ld = c - split; }
            o.stats = st;
            o.mean = L.mean; o.invstd = L.invstd;
            o.gamma = (const float *)L.bn.gamma.data_ptr();
            o.dgamma = dg; o.dbeta = db;
        };
        std::vector<at::Tensor> skip_grads;
        for (size_t si = steps.size(); si-- > 0;) {
            const Step &S = steps[si];
            if (S.kind == 0) {
                float *st2 = nullptr, *st1 = nullptr;
                void *dz2 = gemm_bwd(S.l2, g, false, st2, true);
                at::Tensor gS;
                if (S.has_skip) {   // data gradient of the 1x1 skip conv: no mask, no statistics; independent of dz2
                    gS = pempty({S.l1.n_in, S.l1.c_in}, B.bf);
                    keep_alive.push_back(gS);
                    doda_cx_op &o = B.push(DODA_CX_GEMM, DODA_CX_F_IDENTITY);
                    o.tbl = (const int32_t *)S.ident.data_ptr();
                    o.rows = S.l1.n_in; o.rows_in = S.l1.n_in; o.c_in = S.l2.c_out; o.c_out = S.l1.c_in; o.K = 1; o.tbl_ld = S.l1.n_in;
                    o.x = g.data_ptr(); o.x_ld = S.l2.c_out; o.w = S.skip.pk_bwd.data_ptr(); o.y = gS.data_ptr(); o.y_ld = S.l1.c_in;
                }
                at::Tensor g1, none;
                bn_bwd(S.l2, dz2, st2, at::Tensor(), g1, none);
                void *dz1 = gemm_bwd(S.l1, g1, false, st1, true);
                at::Tensor left, right;
                bn_bwd(S.l1, dz1, st1, S.has_skip ? gS : g, left, right);
                wjobs.push_back({S.l2.a, g, S.l2.fwd_tbl, S.l2.cv.weight, S.l2.n_out, PairLists()});
                wjobs.push_back({S.l1.a, g1, S.l1.fwd_tbl, S.l1.cv.weight, S.l1.n_out, PairLists()});
                if (S.has_skip) {
                    PairLists pl;
                    if (S.skip_pairs) { pl.in = S.ident; pl.out = S.ident; }
                    wjobs.push_back({S.l1.x.t, g, S.ident, S.skip.weight, S.l1.n_in, pl});
                }
                if (right.defined()) { skip_grads.push_back(left); g = right; }
                else g = left;
            } else {
                float *st = nullptr;
                void *dz = gemm_bwd(S.l1, g, false, st, true);
                at::Tensor add, left, none;
                if (S.kind == 1) {
                    TORCH_CHECK(!skip_grads.empty(), "doda coarse: unbalanced down / up steps");
                    add = skip_grads.back();
                    skip_grads.pop_back();
                }
                bn_bwd(S.l1, dz, st, add, left, none);
                wjobs.push_back({S.l1.a, g, S.l1.fwd_tbl, S.l1.cv.weight, S.l1.n_out, PairLists()});
                g = left;
            }
        }
        B.run(g);
        // parameter gradients: the weight gradients join the step's deferred launch; gamma / beta are bound now (not in a
        // backward pass restricted to specific inputs: see plain_accumulating_backward)
        if (!plain) wjobs.clear();
        for (WJob &w : wjobs) {
            if (!try_defer_wgrad(w.a, w.b, w.tbl, w.n_rows, w.weight, w.pl))
                deposit_grad(w.weight, wgrad(w.a, w.b, w.tbl, w.n_rows, w.pl).reshape(w.weight.sizes()).to(w.weight.scalar_type()));
        }
        for (PGrad &pg : pgrads)
            if (pg.fresh) deposit_grad(pg.param, pg.buf);
        // (the arena of this pass must outlive the launch: the caching allocator re-uses a freed block only for work queued
        // LATER on this stream, which is ordered behind the launch)
        keep_alive.clear();
        g_coarse_launches_bwd = B.launches;
        if (task_should_compute_output(0)) out[0] = g;
        return out;
    }
    std::vector<at::Tensor> keep_alive;
    void release_variables() override {
        steps.clear();
        keep.clear();
        x_in.reset();
    }
    std::string name() const override { return "DodaCoarseUBlockBackward"; }
};

// Returns {y [n, c] in the dtype of x, fp64 totals of y (training) or undefined}.
// gate_step >= 0 with `gate`: the host callback runs when the ops of steps [0, gate_step) have been issued.
std::vector<at::Tensor> coarse_ublock(const at::Tensor &x_in, const c10::optional<at::Tensor> &stats_in, const std::vector<int64_t> &kinds,
                                      const std::vector<TList> &tensors, const std::vector<std::vector<double>> &scalars, bool training,
                                      int64_t gate_step = -1, const std::function<void()> &gate = nullptr) {
    host_timing::Scope host_scope(6);
    TORCH_CHECK(x_in.is_cuda() && x_in.dim() == 2 && x_in.size(0) >= 2 &&
                (x_in.scalar_type() == at::kBFloat16 || x_in.scalar_type() == at::kFloat),
                "doda coarse_ublock: bf16 or fp32 device features");
    TORCH_CHECK(kinds.size() == tensors.size() && kinds.size() == scalars.size() && !kinds.empty(), "doda coarse_ublock: step lists");
    const bool need_grad = at::GradMode::is_enabled() && x_in.requires_grad();
    TORCH_CHECK(!need_grad || (training && g_direct_grads && g_defer_wgrad),
                "doda coarse_ublock: a differentiable call needs training mode with deferred weight gradients and direct parameter gradients");
    at::AutoDispatchBelowADInplaceOrView guard;
    const at::Tensor x = x_in.contiguous();
    Builder B;
    B.esz = elem_bytes(x);
    B.bf = x.options();
    B.arena.opt = x.options().dtype(at::kByte);
    std::vector<Step> steps(kinds.size());
    Val cur;
    cur.p = x.data_ptr(); cur.rows = (int)x.size(0); cur.c = (int)x.size(1); cur.ld = cur.c; cur.t = x;
    at::Tensor stats_keep;
    if (training && stats_in.has_value() && stats_in->defined() && is_totals(*stats_in, cur.c)) {
        stats_keep = *stats_in;
        cur.stats = (float *)stats_keep.data_ptr();
        cur.c_split = cur.c;
    }
    struct Skip { Val left; at::Tensor cat; };
    std::vector<Skip> skips;
    at::Tensor y_out, y_stats;
    size_t cut = 0;
    for (size_t si = 0; si < kinds.size(); ++si) {
        if ((int64_t)si == gate_step) cut = B.ops.size();
        Step &S = steps[si];
        const TList &t = tensors[si];
        const std::vector<double> &sc = scalars[si];
        S.kind = (int)kinds[si];
        const bool last = si + 1 == kinds.size();
        if (S.kind == 0) {
            TORCH_CHECK((t.size() == 21 || t.size() == 22) && sc.size() == 4 && t[0].has_value(), "doda coarse_ublock: RB step");
            const at::Tensor &tbl = *t[0];
            const int n = cur.rows;
            TORCH_CHECK(tbl.dim() == 2 && tbl.size(0) == 27 && tbl.size(1) == n && tbl.scalar_type() == at::kInt, "doda coarse_ublock: SubM table");
            S.l1.bn = bn_of(t, 1, sc[0], sc[1]);
            S.l1.cv = cv_of(t, 6);
            S.l2.bn = bn_of(t, 9, sc[2], sc[3]);
            S.l2.cv = cv_of(t, 14);
            S.has_skip = t[17].has_value() && t[17]->defined();
            const int cin = cur.c, cout = (int)S.l1.cv.weight.size(-1);
            TORCH_CHECK(S.l1.cv.weight.size(-2) == cin && S.l2.cv.weight.size(-2) == cout && S.l2.cv.weight.size(-1) == cout &&
                        (S.has_skip || cin == cout), "doda coarse_ublock: channel counts of a residual block");
            S.l1.x = cur;
            S.l1.fwd_tbl = S.l1.bwd_tbl = S.l2.fwd_tbl = S.l2.bwd_tbl = tbl;
            S.l1.K = S.l2.K = 27;
            S.l1.n_in = S.l1.n_out = S.l2.n_in = S.l2.n_out = n;
            S.l1.c_in = cin; S.l1.c_out = cout; S.l2.c_in = cout; S.l2.c_out = cout;
            B.bn_fwd(S.l1, training);
            Val y1 = B.dense(n, cout, false);
            B.gemm_fwd(S.l1, tbl, false, y1, nullptr, training, true);
            Val skipv = cur;
            if (S.has_skip) {   // 1x1 conv of the block's raw input (no barrier: nothing it reads was written since the last one)
                S.skip = cv_of(t, 17);
                TORCH_CHECK(t[20].has_value(), "doda coarse_ublock: identity table of the skip conv");
                S.ident = *t[20];
                S.skip_pairs = t.size() > 21 && t[21].has_value() && t[21]->defined();
                TORCH_CHECK(cur.t.defined() && cur.ld == cur.c, "doda coarse_ublock: the skip conv's input must be dense");
                skipv = B.dense(n, cout, false);
                Layer sk;
                sk.a = cur.t;
                sk.cv = S.skip;
                B.gemm_fwd(sk, S.ident, true, skipv, nullptr, false, false);
            }
            S.l2.x = y1;
            B.bn_fwd(S.l2, training);
            // where the block's output goes: the left half of the concatenation when a strided conv follows
            Val y;
            if (!last && kinds[si + 1] == 1) {
                at::Tensor cat = pempty({n, 2 * cout}, B.bf);
                y.p = cat.data_ptr(); y.rows = n; y.c = cout; y.ld = 2 * cout;
                skips.push_back({Val(), cat});
            } else {
                y = B.dense(n, cout, last);
                if (last) y_out = y.t;
            }
            if (last && training) y_stats = stats_totals_take(cout, x.options());   // (the caller may feed these totals to the next fused BatchNorm)
            B.gemm_fwd(S.l2, tbl, false, y, &skipv, training, true);
            if (last && training) {   // (gemm_fwd put the partial rows into the arena: point the op at the returned tensor instead)
                B.ops.back().stats = (float *)y_stats.data_ptr();
                y.stats = (float *)y_stats.data_ptr();
            }
            if (!last && kinds[si + 1] == 1) skips.back().left = y;
            cur = y;
        } else if (S.kind == 1 || S.kind == 2) {
            TORCH_CHECK(t.size() == 10 && sc.size() == 3 && t[0].has_value() && t[1].has_value(), "doda coarse_ublock: down / up step");
            S.l1.bn = bn_of(t, 2, sc[0], sc[1]);
            S.l1.cv = cv_of(t, 7);
            S.l1.fwd_tbl = *t[0];
            S.l1.bwd_tbl = *t[1];
            S.l1.K = 8;
            const int n_out = (int)sc[2], cin = cur.c, cout = (int)S.l1.cv.weight.size(-1);
            TORCH_CHECK(S.l1.fwd_tbl.size(0) == 8 && S.l1.fwd_tbl.size(1) == n_out && S.l1.bwd_tbl.size(0) == 8 && S.l1.bwd_tbl.size(1) == cur.rows &&
                        S.l1.cv.weight.size(-2) == cin, "doda coarse_ublock: strided rulebook / channels");
            S.l1.x = cur;
            S.l1.n_in = cur.rows; S.l1.n_out = n_out; S.l1.c_in = cin; S.l1.c_out = cout;
            B.bn_fwd(S.l1, training);
            if (S.kind == 1) {
                Val d = B.dense(n_out, cout, false);
                B.gemm_fwd(S.l1, S.l1.fwd_tbl, false, d, nullptr, training, true);
                cur = d;
            } else {
                TORCH_CHECK(!skips.empty(), "doda coarse_ublock: unbalanced down / up steps");
                Skip sk = skips.back();
                skips.pop_back();
                TORCH_CHECK(sk.left.rows == n_out && sk.left.c == cout, "doda coarse_ublock: the inverse conv must restore the skipped level");
                Val u;
                u.p = (char *)sk.cat.data_ptr() + (size_t)cout * B.esz; u.rows = n_out; u.c = cout; u.ld = 2 * cout;
                B.gemm_fwd(S.l1, S.l1.fwd_tbl, false, u, nullptr, training, true);
                Val cat;
                cat.p = sk.cat.data_ptr(); cat.rows = n_out; cat.c = 2 * cout; cat.ld = 2 * cout; cat.t = sk.cat;
                cat.stats = sk.left.stats; cat.stats_b = u.stats; cat.c_split = cout;
                cur = cat;
            }
        } else {
            TORCH_CHECK(false, "doda coarse_ublock: unknown step kind");
        }
    }
    TORCH_CHECK(y_out.defined() && skips.empty(), "doda coarse_ublock: the last step must be a residual block at the input's level");
    B.run(x, cut, gate);
    g_coarse_launches_fwd = B.launches;
    g_last_bn.reset();
    if (need_grad) {
        auto node = std::shared_ptr<CoarseNode>(new CoarseNode(), torch::autograd::deleteNode);
        node->set_next_edges(torch::autograd::collect_next_edges(x_in));
        node->steps = std::move(steps);
        node->keep = B.arena.chunks;
        if (stats_keep.defined()) node->keep.push_back(stats_keep);
        node->keep.push_back(x);
        node->x_in = x;
        torch::autograd::set_history(y_out, node);
    }
    return {y_out, y_stats};
}

}  // namespace coarse

}  // namespace

// ---- optimizer step: all parameters in one launch (doda_sgd_multi) ------------------------------------
void sgd_step(const std::vector<at::Tensor> &params, const std::vector<at::Tensor> &grads,
              const std::vector<at::Tensor> &bufs, const std::vector<int64_t> &first, double lr, double momentum,
              double dampening, double weight_decay, bool nesterov, bool maximize) {
    const size_t n = params.size();
    if (n == 0) return;
    TORCH_CHECK(grads.size() == n && first.size() == n && (momentum == 0.0 || bufs.size() == n),
                "doda sgd_step: list lengths differ");
    std::vector<doda_sgd_tensor> t(n);
    const auto dev = params[0].device();
    for (size_t k = 0; k < n; ++k) {
        const at::Tensor &p = params[k], &g = grads[k];
        TORCH_CHECK(p.is_cuda() && p.device() == dev && g.device() == dev && p.scalar_type() == at::kFloat &&
                    g.scalar_type() == at::kFloat && p.is_contiguous() && g.is_contiguous() && g.sizes() == p.sizes(),
                    "doda sgd_step: parameters and gradients must be contiguous fp32 tensors on one device");
        t[k].p = (float *)p.data_ptr();
        t[k].g = (const float *)g.data_ptr();
        t[k].buf = nullptr;
        if (momentum != 0.0) {
            const at::Tensor &b = bufs[k];
            TORCH_CHECK(b.defined() && b.device() == dev && b.scalar_type() == at::kFloat && b.is_contiguous() &&
                        b.sizes() == p.sizes(), "doda sgd_step: bad momentum buffer");
            t[k].buf = (float *)b.data_ptr();
        }
        t[k].n = p.numel();
        t[k].first_step = first[k] ? 1 : 0;
        t[k].reserved = 0;
    }
    c10::DeviceGuard guard(dev);
    const size_t nb = doda_sgd_multi_desc_bytes((int32_t)n);
    at::Tensor desc = pempty({(int64_t)nb}, params[0].options().dtype(at::kByte));
    check(doda_sgd_multi(t.data(), (int32_t)n, lr, momentum, dampening, weight_decay, nesterov ? 1 : 0,
                         maximize ? 1 : 0, desc.data_ptr(), nb, stream_of(params[0])), "doda_sgd_multi");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("indice_conv", &indice_conv, "sparse conv (gather table) with autograd",
          py::arg("features"), py::arg("weight"), py::arg("fwd_tbl"), py::arg("bwd_tbl"), py::arg("n_out"),
          py::arg("bwd_layout"), py::arg("pk_fwd"), py::arg("pk_bwd"), py::arg("residual"),
          py::arg("pair_in") = py::none(), py::arg("pair_out") = py::none(), py::arg("pair_num") = py::none(),
          py::arg("pair_seg") = py::none());
    m.def("bn_relu", &bn_relu, "fused BatchNorm1d(+ReLU) with autograd", py::arg("x"), py::arg("weight"),
          py::arg("bias"), py::arg("running_mean"), py::arg("running_var"), py::arg("nbt"), py::arg("training"),
          py::arg("momentum"), py::arg("eps"), py::arg("relu"), py::arg("stats") = py::none(), py::arg("stats_b") = py::none());
    m.def("bn_relu_pass", &bn_relu_pass, "fused BatchNorm1d(+ReLU) returning (y, alias of x) for residual blocks",
          py::arg("x"), py::arg("weight"), py::arg("bias"), py::arg("running_mean"), py::arg("running_var"),
          py::arg("nbt"), py::arg("training"), py::arg("momentum"), py::arg("eps"), py::arg("relu"),
          py::arg("stats") = py::none(), py::arg("stats_b") = py::none());
    m.def("indice_conv_stats", [](const at::Tensor &features, const at::Tensor &weight, const at::Tensor &fwd_tbl,
                                  const at::Tensor &bwd_tbl, int64_t n_out, int64_t bwd_layout,
                                  const c10::optional<at::Tensor> &pk_fwd, const c10::optional<at::Tensor> &pk_bwd,
                                  const c10::optional<at::Tensor> &residual, const c10::optional<at::Tensor> &pair_in,
                                  const c10::optional<at::Tensor> &pair_out, const c10::optional<at::Tensor> &pair_num,
                                  const c10::optional<at::Tensor> &pair_seg) {
        auto r = indice_conv_impl(features, weight, fwd_tbl, bwd_tbl, n_out, bwd_layout, pk_fwd, pk_bwd, residual,
                                  pair_in, pair_out, pair_num, pair_seg, true);
        return std::make_pair(r[0], r[1].defined() ? c10::optional<at::Tensor>(r[1]) : c10::nullopt);
    }, "sparse conv that also returns the BatchNorm statistics partials of its output",
          py::arg("features"), py::arg("weight"), py::arg("fwd_tbl"), py::arg("bwd_tbl"), py::arg("n_out"),
          py::arg("bwd_layout"), py::arg("pk_fwd"), py::arg("pk_bwd"), py::arg("residual"),
          py::arg("pair_in") = py::none(), py::arg("pair_out") = py::none(), py::arg("pair_num") = py::none(),
          py::arg("pair_seg") = py::none());
    m.def("residual_block", [](const at::Tensor &x, const c10::optional<at::Tensor> &stats_in, const std::vector<at::Tensor> &bn1,
                               const std::vector<at::Tensor> &bn2, bool training, double momentum1, double eps1,
                               double momentum2, double eps2, const std::vector<c10::optional<at::Tensor>> &cv1,
                               const std::vector<c10::optional<at::Tensor>> &cv2,
                               const std::vector<c10::optional<at::Tensor>> &rb, int64_t n_out,
                               const c10::optional<at::Tensor> &skip, bool want_stats,
                               const std::vector<c10::optional<at::Tensor>> &sc, const c10::optional<at::Tensor> &stats_in_b) {
        auto r = residual_block(x, stats_in, bn1, bn2, training, momentum1, eps1, momentum2, eps2, cv1, cv2, rb, n_out, skip,
                                want_stats, sc, stats_in_b);
        return std::make_pair(r[0], r[1].defined() ? c10::optional<at::Tensor>(r[1]) : c10::nullopt);
    }, "BatchNorm -> ReLU -> SubM conv -> BatchNorm -> ReLU -> SubM conv (+ skip) in one call",
          py::arg("x"), py::arg("stats_in"), py::arg("bn1"), py::arg("bn2"), py::arg("training"), py::arg("momentum1"),
          py::arg("eps1"), py::arg("momentum2"), py::arg("eps2"), py::arg("cv1"), py::arg("cv2"), py::arg("rb"),
          py::arg("n_out"), py::arg("skip"), py::arg("want_stats"),
          py::arg("sc") = std::vector<c10::optional<at::Tensor>>(), py::arg("stats_in_b") = py::none());
    m.def("set_bn_fusion", [](bool on) { g_bn_fusion = on; }, "BatchNorm statistics in the conv epilogues (default on)");
    m.def("gather", [](const at::Tensor &x, const at::Tensor &w, const c10::optional<at::Tensor> &packed,
                       const at::Tensor &tbl, int64_t n_out, int64_t layout, int64_t nc, bool out_f32) {
        return gather(x, w, packed, tbl, n_out, layout, nc, out_f32);
    }, "raw gather-GEMM");
    m.def("wgrad", [](const at::Tensor &a, const at::Tensor &b, const at::Tensor &tbl, int64_t n_rows) {
        return wgrad(a, b, tbl, n_rows);
    }, "raw weight gradient");
    m.def("build_pyramid", &build_pyramid,
          "all SubM k3 and k2s2 rulebooks of an n-level U-Net in one call (13 native builds, 6 size read-backs)",
          py::arg("indices"), py::arg("shape"), py::arg("batch"), py::arg("n_levels"), py::arg("pairs_min_rows") = -1,
          py::arg("tile_min_rows") = -1, py::arg("tile_levels") = 2,
          py::call_guard<py::gil_scoped_release>());   // its size read-backs block: let other Python threads run
    m.def("build_pyramid_probe", [](const at::Tensor &indices, std::vector<int64_t> shape, int64_t batch, int64_t n_levels,
                                    int64_t pairs_min_rows, int64_t tile_min_rows, int64_t tile_levels) {
              // build_pyramid + the finest tilebook's overflow counters (tiles, above the 64-byte capacity, above the list)
              // in ONE call without the GIL: the rulebook thread's read-back does not stall the issuing thread
              auto levels = build_pyramid(indices, shape, batch, n_levels, pairs_min_rows, tile_min_rows, tile_levels);
              int64_t nt = -1, o64 = 0, o32 = 0;
              if (!levels.empty()) {
                  const at::Tensor &tbl = std::get<0>(levels[0]);
                  const void *tb = tilebook_behind(tbl, tbl.dim() == 2 ? tbl.size(1) : 0);
                  if (tb) {
                      const int64_t T = doda_tilebook_tile(), K = tbl.size(0), UMAX = doda_tilebook_umax();
                      nt = (tbl.size(1) + T - 1) / T;
                      (void)K; (void)UMAX; (void)nt;
                      char *p = (char *)const_cast<void *>(tb) + doda_tilebook_bytes((int32_t)tbl.size(1), (int32_t)tbl.size(0)) - 8;
                      int32_t ov[2] = {0, 0};
                      if (!read_back_blocking(p, ov, 2, stream_of(tbl), (int)tbl.device().index())) {
                          at::Tensor over = at::from_blob(p, {2}, tbl.options()).cpu();
                          ov[0] = over[0].item<int32_t>();
                          ov[1] = over[1].item<int32_t>();
                      }
                      o64 = ov[0];
                      o32 = ov[1];
                  }
              }
              return std::make_tuple(levels, nt, o64, o32);
          }, py::arg("indices"), py::arg("shape"), py::arg("batch"), py::arg("n_levels"), py::arg("pairs_min_rows") = -1,
          py::arg("tile_min_rows") = -1, py::arg("tile_levels") = 2, py::call_guard<py::gil_scoped_release>());
    m.def("with_tilebook", [](const at::Tensor &tbl) {
              TORCH_CHECK(tbl.is_cuda() && tbl.scalar_type() == at::kInt && tbl.dim() == 2, "doda with_tilebook: int32 [K, M] table");
              at::Tensor out = table_with_tilebook(tbl.size(0), tbl.size(1), tbl.options());
              out.copy_(tbl);
              if (tilebook_behind(out, out.size(1))) build_tilebook(out, stream_of(out));
              return out;
          }, "copy of a gather table with its tilebook (doda_tilebook_build) in the same storage");
    m.def("has_tilebook", [](const at::Tensor &tbl) { return tilebook_behind(tbl, tbl.dim() == 2 ? tbl.size(1) : 0) != nullptr; });
    m.def("tilebook_parts", [](const at::Tensor &tbl) {   // (ulist [nt,UMAX] int32, local indices [nt,K,T] int16 decoded, ucount [nt] int32), for tests
              const void *tb = tilebook_behind(tbl, tbl.size(1));
              TORCH_CHECK(tb, "doda tilebook_parts: no tilebook");
              const int64_t T = doda_tilebook_tile(), nt = (tbl.size(1) + T - 1) / T, K = tbl.size(0), UMAX = doda_tilebook_umax();
              TORCH_CHECK(K == 27 && T == 256, "doda tilebook_parts: layout of csrc/tilebook.hpp");
              const int64_t LW = 10;              // planes of packed local indices (csrc/tilebook.hpp: tb_lplane / tb_lshift / tb_lpos)
              char *p = (char *)const_cast<void *>(tb);
              auto o32 = tbl.options();
              // (stored in the DMA kernel's lane order, csrc/tilebook.hpp tb_upos: handed out in list order)
              at::Tensor order = pempty({UMAX}, at::TensorOptions().dtype(at::kLong));
              for (int64_t e = 0; e < UMAX; ++e) order.data_ptr<int64_t>()[e] = (((e >> 5) & 7) * 32 + (e & 31)) * 4 + (e >> 8);
              at::Tensor ulist = at::from_blob(p, {nt, UMAX}, o32).index_select(1, order.to(tbl.device()));
              at::Tensor pos = pempty({T}, at::TensorOptions().dtype(at::kLong));      // word of row t inside a plane
              for (int64_t t = 0; t < T; ++t) pos.data_ptr<int64_t>()[t] = (t & 0xC0) | (((t & 3) | ((t & 4) << 1) | ((t & 8) >> 1)) << 2) | ((t >> 4) & 3);
              at::Tensor words = at::from_blob(p + nt * UMAX * 4, {nt, LW, T}, o32).to(at::kLong).index_select(2, pos.to(tbl.device()));
              std::vector<at::Tensor> per_offset;
              for (int64_t o = 0; o < K; ++o)
                  per_offset.push_back(words.select(1, (o & 1) * 5 + (o >> 1) / 3).bitwise_right_shift(10 * ((o >> 1) % 3)).bitwise_and(0x3ff));
              at::Tensor lidx = at::stack(per_offset, 1).to(at::kShort);          // [nt, K, T]
              at::Tensor ucount = at::from_blob(p + nt * UMAX * 4 + nt * T * LW * 4, {nt}, o32).clone();
              return std::make_tuple(ulist, lidx, ucount);
          });
    m.def("tilebook_overflow", [](const at::Tensor &tbl) {   // (tiles, tiles above the 64-byte-row capacity, above the list capacity); syncs
              const void *tb = tilebook_behind(tbl, tbl.size(1));
              TORCH_CHECK(tb, "doda tilebook_overflow: no tilebook");
              const int64_t T = doda_tilebook_tile(), nt = (tbl.size(1) + T - 1) / T, K = tbl.size(0), UMAX = doda_tilebook_umax();
              (void)K; (void)UMAX; (void)nt;
              char *p = (char *)const_cast<void *>(tb) + doda_tilebook_bytes((int32_t)tbl.size(1), (int32_t)tbl.size(0)) - 8;
              at::Tensor over = at::from_blob(p, {2}, tbl.options()).cpu();
              return std::make_tuple(nt, (int64_t)over[0].item<int32_t>(), (int64_t)over[1].item<int32_t>());
          }, py::call_guard<py::gil_scoped_release>());   // (called on the rulebook thread: its read-back must not hold the GIL)
    m.def("set_tile_kernel", [](bool on) { doda_set_option(DODA_OPT_TILE_KERNEL, on ? 1 : 0); });
    m.def("pending_wgrads", []() { std::lock_guard<std::mutex> lock(g_wq_mu); return (int64_t)g_wq.size(); });
    m.def("set_defer_wgrad", [](bool on) { g_defer_wgrad = on; },
          "queue conv weight gradients during backward and issue them in one multi-layer call at its end");
    m.def("get_defer_wgrad", []() { return g_defer_wgrad; });
    m.def("set_direct_grads", [](bool on) { g_direct_grads = on; },
          "with set_defer_wgrad: the extension's nodes write parameter gradients to .grad themselves and keep no autograd edge to "
          "the parameters (no AccumulateGrad nodes for them)");
    m.def("get_direct_grads", []() { return g_direct_grads; });
    m.def("grads_into_views", [](const std::vector<at::Tensor> &params, const std::vector<at::Tensor> &views) {
              // GradAllReduce's per-step pass over its parameters (doda_amd/dist.py _Bucket.gather_in_place) in one call: a
              // gradient already living in its bucket view costs a pointer compare; a stray one is copied in, a missing one
              // zero-filled; .grad is re-bound to the view.  Returns how many were NOT in place.  (281 parameters x ~1.5 us of
              // Python per step sat in front of the collectives on a host-bound step.)
              TORCH_CHECK(params.size() == views.size(), "doda grads_into_views: list lengths differ");
              at::NoGradGuard no_grad;
              int64_t moved = 0;
              for (size_t k = 0; k < params.size(); ++k) {
                  at::Tensor &slot = const_cast<at::Tensor &>(params[k]).mutable_grad();
                  const at::Tensor &v = views[k];
                  if (!slot.defined()) const_cast<at::Tensor &>(v).zero_();
                  else if (slot.data_ptr() != v.data_ptr() || slot.strides() != v.strides()) {
                      const_cast<at::Tensor &>(v).copy_(slot);
                      // (round 6) the copy runs on the CURRENT stream — the reducer's side stream — while the stray gradient was
                      // allocated on the backward pass's stream: dropping it below hands its block back to THAT stream's pool at
                      // once, and the backward pass still running there (the early exchange) or the next step could overwrite it
                      // before the copy has read it.  Seen as a two-rank test failing once in ~30 runs: linear.bias (a gradient
                      // torch's AccumulateGrad makes, not one deposited in its bucket) off by half.
                      // (the allocator's own entry point: Tensor::record_stream wants a stream of the tensor's masqueraded device type)
                      if (slot.is_cuda() && slot.has_storage())
                          c10::hip::HIPCachingAllocator::recordStream(slot.storage().data_ptr(), c10::hip::getCurrentHIPStream(slot.device().index()));
                  }
                  else continue;
                  slot = v;
                  ++moved;
              }
              return moved;
          }, "re-home every parameter's gradient in its bucket view (copy / zero-fill only where it is not there already)");
    m.def("flush_wgrads", &flush_wgrads);
    m.def("fp_log_take", []() { std::lock_guard<std::mutex> lock(g_fp_mu); auto out = g_fp; g_fp.clear(); return out; },
          "DODA_FPLOG=1: the (operator, fingerprint) list logged since the last call");
    m.def("stats_totals_begin_pass", &stats_totals_begin_pass, "a new forward pass: the next statistics totals come out of a fresh zeroed arena");
    m.def("set_stats_totals", [](bool on) { g_stats_totals = on; return g_stats_totals; },
          "BatchNorm statistics of the conv epilogues as fp64 totals (one-launch BatchNorm) or as per-workgroup rows + a reduction launch");
    m.def("get_stats_totals", []() { return g_stats_totals; });
    m.def("flush_wgrads_side", &flush_wgrads_side,
          "issue the weight gradients queued so far on a second stream (joined by the flush at the end of backward); returns the number of jobs");
    m.def("flush_wgrads_early", &flush_wgrads_early,
          "issue the weight-gradient jobs queued so far (mid-backward) and record the event wait_wide_wgrads() waits for");
    m.def("set_grad_home", &set_grad_home, "view of a flat gradient bucket that receives the parameter's gradient in place (None: forget)");
    m.def("drop_grad_home", &drop_grad_home, "forget the parameter's home if it is this view; returns whether it was");
    m.def("clear_grad_homes", &clear_grad_homes);
    m.def("set_wgrad_split", [](bool on) { g_wq_split = on; if (!on) g_ev_early_valid = false; },
          "issue the wide layers' weight gradients first and record an event behind them (GradAllReduce)");
    m.def("wait_wide_wgrads", [](int64_t stream) {
              if (!g_ev_early_valid) return false;
              g_ev_early_valid = false;
              TORCH_CHECK(hipStreamWaitEvent((hipStream_t)stream, g_ev_early, 0) == hipSuccess, "doda: hipStreamWaitEvent");
              return true;
          }, "make `stream` wait for the wide layers' weight gradients of the last flush; false if there was no split flush");
    m.def("host_timing", []() {   // {name: (microseconds, calls)} since the last call; empty unless DODA_HOST_TIMING=1
              py::dict d;
              for (auto &sl : host_timing::g_slots) {
                  d[py::str(sl.name)] = py::make_tuple(sl.ns / 1e3, sl.calls);
                  sl.ns = 0; sl.calls = 0;
              }
              return d;
          });
    m.def("clear_grads", [](const std::vector<at::Tensor> &params) {   // optimizer.zero_grad(set_to_none=True) in one call
              for (const at::Tensor &p : params)
                  if (p.defined() && p.grad().defined()) const_cast<at::Tensor &>(p).mutable_grad().reset();
          }, "set .grad of every listed parameter to None (203 Python attribute stores per step otherwise)");
    m.def("sgd_step", &sgd_step, "torch.optim.SGD's update of all parameters in one launch",
          py::call_guard<py::gil_scoped_release>());
    m.def("coarse_ublock", [](const at::Tensor &x, const c10::optional<at::Tensor> &stats_in, const std::vector<int64_t> &kinds,
                              const std::vector<std::vector<c10::optional<at::Tensor>>> &tensors,
                              const std::vector<std::vector<double>> &scalars, bool training, int64_t gate_step, py::object gate) {
        std::function<void()> between;
        if (gate_step >= 0 && !gate.is_none()) between = [gate]() { gate(); };   // (the GIL is held throughout this call)
        auto r = coarse::coarse_ublock(x, stats_in, kinds, tensors, scalars, training, gate_step, between);
        return std::make_pair(r[0], r[1].defined() ? c10::optional<at::Tensor>(r[1]) : c10::nullopt);
    }, "a U-Net subtree of coarse levels as one op list per direction: per-layer launches issued inside the library with the "
       "BatchNorm ops folded into the convolutions (doda_layers_run)",
          py::arg("x"), py::arg("stats_in"), py::arg("kinds"), py::arg("tensors"), py::arg("scalars"), py::arg("training"),
          py::arg("gate_step") = -1, py::arg("gate") = py::none());
    m.def("coarse_launches", []() { return std::make_pair((int64_t)coarse::g_coarse_launches_fwd, (int64_t)coarse::g_coarse_launches_bwd); },
          "kernel launches of the last per-layer forward / backward op list");
    m.def("abi_version", []() { return doda_abi_version(); });
    m.def("built_for_abi", []() { return (int)DODA_ABI_VERSION; });
}
