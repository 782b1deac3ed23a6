// Host-side runtime bits of libhh: version, thread-local error string, launch check.
#include "common.h"
#include <cstring>
#include <atomic>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void hh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::atomic<long long> g_calls{0};       // launching entry points that reached their launch check (hh_call_count)

int hh_check_launch(const char* what) {
    g_calls.fetch_add(1, std::memory_order_relaxed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hh_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return HH_ERR_LAUNCH;
    }
    return HH_OK;
}

static int current_device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? d : 0; }
bool hh_attr_needed(const std::atomic<uint64_t>& mask) {
    const int d = current_device();
    return d >= 64 || !((mask.load(std::memory_order_acquire) >> d) & 1u);
}
void hh_attr_done(std::atomic<uint64_t>& mask) {
    const int d = current_device();
    if (d < 64) mask.fetch_or(1ull << d, std::memory_order_release);
}

extern "C" int hh_version(void) { return 100; }
// sizeof of the option structs of the ABI as THIS library was compiled: a binding compares it with its own declaration when it loads the
// library (the structs grow by appended fields from round to round; a stale binding would hand the kernels garbage)
extern "C" int hh_abi_sizeof(const char* name) {
    if (name && !strcmp(name, "hh_gemm_epilogue")) return (int)sizeof(hh_gemm_epilogue);
    if (name && !strcmp(name, "hh_qgemm_opts")) return (int)sizeof(hh_qgemm_opts);
    if (name && !strcmp(name, "hh_qgemm_item")) return (int)sizeof(hh_qgemm_item);
    return -1;
}
extern "C" const char* hh_last_error_string(void) { return g_err; }
extern "C" int64_t hh_call_count(void) { return (int64_t)g_calls.load(std::memory_order_relaxed); }

// ---- per-stream CU budget.  The pipelined training step runs the frozen towers of batch i+1 on one stream while the decoder
// forward/backward of batch i runs on another.  The persistent 256x256 GEMM normally puts one workgroup on every CU (LDS and
// VGPRs of a CU are then full), so kernels of the other stream only start in the gaps between encoder kernels.  A budget of
// n < all CUs makes persistent kernels launched on that stream use n workgroups and leaves the other CUs to concurrent streams.
// (Hardware CU masks -- hipExtStreamCreateWithCUMask -- were measured and rejected: mask bit i is CU (i / 8) of XCD (i % 8), but
// the dispatcher keeps dealing workgroups evenly to the 4 shader engines of an XCD, so any mask that is not a multiple of 32
// CUs runs at the speed of the emptiest shader engine: -15 % on LayerNorm, -35 % on the persistent GEMM at 248 of 256 CUs.)
static hipStream_t g_budget_stream[16];
static int g_budget_cus[16];
static int g_budget_n = 0;
static std::mutex g_budget_mu;               // the table is the library's only mutable global state besides tuning knobs

static int device_cus() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
    }
    return ncu;
}

// Number of workgroups a one-per-CU persistent kernel launches on this stream.
int hh_stream_cu_count(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_budget_mu);
    for (int i = 0; i < g_budget_n; ++i)
        if (g_budget_stream[i] == s) return g_budget_cus[i];
    return device_cus();
}

// Slot (0 .. 31) of a stream in the library's per-stream device state (the dynamic tile counters of the persistent GEMM): kernels on one
// stream run one after the other, so one slot per stream is never shared by two running kernels.  -1: table full (callers fall back).
static hipStream_t g_slot_stream[32];
static int g_slot_n = 0;
int hh_stream_slot(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_budget_mu);
    for (int i = 0; i < g_slot_n; ++i)
        if (g_slot_stream[i] == s) return i;
    if (g_slot_n == 32) return -1;
    g_slot_stream[g_slot_n] = s;
    return g_slot_n++;
}

extern "C" int hh_stream_set_cu_budget(hh_stream_t stream, int n_cus) {
    const int ncu = device_cus();
    if (n_cus == 0) n_cus = ncu;
    if (n_cus < 8 || n_cus > ncu || n_cus % 8) {
        hh_set_error("hh_stream_set_cu_budget: n_cus = %d must be 0 or a multiple of 8 in [8, %d]", n_cus, ncu);
        return HH_ERR_SHAPE;
    }
    std::lock_guard<std::mutex> lk(g_budget_mu);
    for (int i = 0; i < g_budget_n; ++i)
        if (g_budget_stream[i] == (hipStream_t)stream) {
            if (n_cus == ncu) {                                  // back to the default: the entry leaves the table
                --g_budget_n;
                g_budget_stream[i] = g_budget_stream[g_budget_n];
                g_budget_cus[i] = g_budget_cus[g_budget_n];
            } else g_budget_cus[i] = n_cus;
            return HH_OK;
        }
    if (n_cus == ncu) return HH_OK;
    if (g_budget_n == 16) { hh_set_error("hh_stream_set_cu_budget: more than 16 budgeted streams"); return HH_ERR_UNSUPPORTED; }
    g_budget_stream[g_budget_n] = (hipStream_t)stream;
    g_budget_cus[g_budget_n++] = n_cus;
    return HH_OK;
}

extern "C" int hh_stream_get_cu_budget(hh_stream_t stream, int* out) {
    if (!out) { hh_set_error("hh_stream_get_cu_budget: null out"); return HH_ERR_SHAPE; }
    *out = hh_stream_cu_count((hipStream_t)stream);
    return HH_OK;
}

// ---- caller-owned workspaces (kernels never allocate).  Sizes in bytes of the scratch buffers the entry
// points below take as arguments, so that a host binding needs no knowledge of the kernels' internals.
extern "C" int64_t hh_workspace_bytes_gemm_splitk(int64_t M, int N, int splitk) {
    return (M < 0 || N <= 0 || splitk < 1) ? -1 : (int64_t)splitk * M * N * 4;                 // fp32 partial slabs [splitk, M, N]
}
extern "C" int64_t hh_workspace_bytes_gemm_zstats(int64_t M, int N) {
    return (M < 0 || N <= 0 || N % 128) ? -1 : M * (int64_t)(N / 128) * 2 * 4;                 // fp32 [M, N / 128, (sum, sum of squares)]
}
extern "C" int64_t hh_workspace_bytes_gemm_tn(int M, int N, int splits) {
    return (M <= 0 || N <= 0 || splits < 1) ? -1 : (int64_t)splits * M * N * 4;                // fp32 partial tiles [splits, M, N]
}
extern "C" int64_t hh_workspace_bytes_xattn_bwd(int B, int Q, int heads, int dq_splits) {
    return (B < 0 || Q <= 0 || heads <= 0 || dq_splits < 1) ? -1 : (int64_t)dq_splits * B * Q * heads * 64 * 4;   // dq partials
}
extern "C" int64_t hh_workspace_bytes_xattn_fwd(int B, int Q, int heads, int splits) {
    return (B < 0 || Q <= 0 || heads <= 0 || splits < 1) ? -1 : (int64_t)splits * B * Q * heads * (64 + 1) * 4;   // out planes, then lse planes
}
extern "C" int64_t hh_workspace_bytes_attn_cls_partial(int B, int T, int n, int heads, int time_mode) {
    if (B < 0 || T <= 0 || n <= 0 || heads <= 0 || (time_mode && T > 32)) return -1;
    int TP = 1;                                                                                // time kernel: frame slots per tile = next power of two
    while (TP < T) TP *= 2;
    const int64_t G = time_mode ? (n + (128 / TP) - 1) / (128 / TP) : T;                       // key groups per (clip, head)
    return (int64_t)B * heads * G * 68 * 4;                                                    // records {m, l, 0, 0, o[64]} fp32
}

// ---- per-kernel device timing recorded by the library itself (bench.py's `roofline`).  When enabled, every stride-th launch of
// an instrumented kernel class is bracketed by two hipEvents recorded on ITS launch stream, with nothing but that kernel between
// them -- so the sum of the elapsed times is comparable with rocprofv3's per-kernel durations (a torch-side bracket around
// hh_gemm_bf16 also contains the row-tail launch and the host gaps).  Off by default: the launch sites then pay one relaxed load.
struct ProfRec { int klass, role; hipEvent_t e0, e1; double work; };
#define HH_PROF_ROLES 4
static std::atomic<int> g_prof_role{0};         // hh_prof_set_role: which part of the step the host is launching (0 = decoder / loss / optimizer, 1 = vision tower, 2 = text tower)
static std::atomic<int> g_prof_stride{0};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static long long g_prof_seen[HH_PROF_CLASSES][HH_PROF_ROLES];
static std::atomic<const char*> g_prof_name[HH_PROF_CLASSES];     // kernel (template instantiation) last dispatched per class while profiling was on
static int g_prof_gen = 0;                       // bumped by every hh_prof_enable: a scope opened before the call must not touch the new records
static const size_t HH_PROF_MAX_RECS = 1u << 16; // bound on the records (and events) kept while profiling stays enabled; later launches are only counted

static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

HHProfScope::HHProfScope(int klass, double work, hipStream_t s) : rec_(-1), gen_(0), stream_(s) {
    const int stride = g_prof_stride.load(std::memory_order_relaxed);
    if (stride <= 0 || klass < 0 || klass >= HH_PROF_CLASSES) return;
    const int role = g_prof_role.load(std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if ((g_prof_seen[klass][role]++ % stride) != 0 || g_prof_recs.size() >= HH_PROF_MAX_RECS) return;      // every stride-th launch of the class IN THIS ROLE
    ProfRec r{klass, role, prof_event(), prof_event(), work};
    hipEventRecord(r.e0, s);
    g_prof_recs.push_back(r);
    rec_ = (int)g_prof_recs.size() - 1;
    gen_ = g_prof_gen;
}

HHProfScope::~HHProfScope() {
    if (rec_ < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    // hh_prof_enable() between the constructor and here cleared the records (and recycled this scope's events): leave the new ones alone
    if (gen_ == g_prof_gen && rec_ < (int)g_prof_recs.size()) hipEventRecord(g_prof_recs[rec_].e1, stream_);
}

// launch sites report WHICH kernel they dispatched (a string literal naming the template instantiation, as rocprofv3 prints it): bench.py
// labels its roofline records with what ran, not with a name composed from the shapes
void hh_prof_note_kernel(int klass, const char* name) {
    if (klass >= 0 && klass < HH_PROF_CLASSES && g_prof_stride.load(std::memory_order_relaxed) > 0) g_prof_name[klass].store(name, std::memory_order_relaxed);
}

extern "C" const char* hh_prof_kernel_name(int klass) {
    if (klass < 0 || klass >= HH_PROF_CLASSES) return "";
    const char* n = g_prof_name[klass].load(std::memory_order_relaxed);
    return n ? n : "";
}

extern "C" int hh_prof_enable(int stride) {
    HH_REQUIRE(stride >= 0, HH_ERR_SHAPE, "hh_prof_enable: stride must be >= 0");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
    g_prof_recs.clear();
    ++g_prof_gen;
    for (int i = 0; i < HH_PROF_CLASSES; ++i)
        for (int j = 0; j < HH_PROF_ROLES; ++j) g_prof_seen[i][j] = 0;
    if (stride > 0)
        for (int i = 0; i < HH_PROF_CLASSES; ++i) g_prof_name[i].store(nullptr, std::memory_order_relaxed);
    g_prof_stride.store(stride, std::memory_order_relaxed);
    return HH_OK;
}

extern "C" int hh_prof_set_role(int role) {
    HH_REQUIRE(role >= 0 && role < HH_PROF_ROLES, HH_ERR_SHAPE, "hh_prof_set_role: role must be in [0, %d)", HH_PROF_ROLES);
    g_prof_role.store(role, std::memory_order_relaxed);
    return HH_OK;
}

extern "C" int hh_prof_read_role(int klass, int role, int64_t* launches_timed, int64_t* launches_seen, double* total_ms, double* total_work);

extern "C" int hh_prof_read(int klass, int64_t* launches_timed, int64_t* launches_seen, double* total_ms, double* total_work) {
    return hh_prof_read_role(klass, -1, launches_timed, launches_seen, total_ms, total_work);
}

extern "C" int hh_prof_read_role(int klass, int role, int64_t* launches_timed, int64_t* launches_seen, double* total_ms, double* total_work) {
    HH_REQUIRE(klass >= 0 && klass < HH_PROF_CLASSES && role >= -1 && role < HH_PROF_ROLES && launches_timed && launches_seen && total_ms && total_work, HH_ERR_SHAPE,
               "hh_prof_read: bad arguments");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int64_t n = 0;
    double ms = 0.0, work = 0.0;
    for (auto& r : g_prof_recs) {
        if (r.klass != klass || (role >= 0 && r.role != role)) continue;
        hipError_t e = hipEventSynchronize(r.e1);
        float t = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, r.e0, r.e1);
        HH_REQUIRE(e == hipSuccess, HH_ERR_LAUNCH, "hh_prof_read: %s", hipGetErrorString(e));
        ms += t; work += r.work; ++n;
    }
    long long seen = 0;
    for (int j = 0; j < HH_PROF_ROLES; ++j)
        if (role < 0 || role == j) seen += g_prof_seen[klass][j];
    *launches_timed = n; *launches_seen = seen; *total_ms = ms; *total_work = work;
    return HH_OK;
}
