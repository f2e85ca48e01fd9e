// ixr_api.hip -- host side of libixrgpu: the C ABI declared in include/ixr_gpu.h.
//
// There is deliberately no CPU decode path in this library: when no gfx950 device is usable every decode entry point
// fails with IXR_ERR_NO_DEVICE / IXR_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/ixr_gpu.h"
#include "../../include/ixr_synth.h"
#include "ixr_device.h"
#include "ixr_launch.h"

using namespace ixr;

// ---------------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static thread_local int32_t g_code = 0;  // code of the last failure on this thread
static int32_t fail(int32_t code, const char* fmt, ...) {
    g_code = code;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr, ret)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            fail(IXR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            ret;                                                                                    \
        }                                                                                           \
    } while (0)

extern "C" const char* ixr_last_error(void) { return g_err; }
namespace ixr {
int32_t set_error(int32_t code, const char* message) { return fail(code, "%s", message); }
}

extern "C" int32_t ixr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------------------------------
// Working memory of ixr_decode_batch_static, the entry point that allocates nothing: one array per GPU, part of the code
// object, held by every GPU the library runs on until the process ends.  `static_scratch` is only a host-side handle of the
// symbol; the array's device address is resolved once per device in ixr_init (hipGetSymbolAddress) and kept in
// g_static_ptr, and that pointer is what kernels, memsets and copies get.
// ---------------------------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char static_scratch[IXR_STATIC_SCRATCH_BYTES];
static std::atomic<uint8_t*> g_static_ptr[64];

extern "C" int32_t ixr_init(int32_t device) {
    // the architecture check is remembered per device (querying the properties costs milliseconds); racing threads at
    // worst both do the query and store the same answer
    static std::atomic<int8_t> checked[64];
    if (device >= 0 && device < 64 && checked[device].load(std::memory_order_acquire) == 1) {
        HIP_TRY(hipSetDevice(device), return IXR_ERR_HIP);
        return 0;
    }
    int n = ixr_device_count();
    if (n <= 0) return fail(IXR_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(IXR_ERR_INVALID_ARGUMENT, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device), return IXR_ERR_HIP);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device), return IXR_ERR_HIP);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(IXR_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    if (device < 64) {
        // this device's copy of the static scratch array (each GPU holds its own, at its own address)
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(static_scratch)), return IXR_ERR_HIP);
        g_static_ptr[device].store((uint8_t*)p, std::memory_order_release);
        checked[device].store(1, std::memory_order_release);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------------------------------------------
extern "C" void* ixr_dev_alloc(int32_t device, uint64_t bytes) {
    if (hipSetDevice(device) != hipSuccess) { fail(IXR_ERR_HIP, "hipSetDevice(%d) failed", device); return nullptr; }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) { fail(IXR_ERR_OUT_OF_MEMORY, "hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
extern "C" void ixr_dev_free(int32_t device, void* dptr) {
    if (!dptr) return;
    (void)hipSetDevice(device);
    (void)hipFree(dptr);
}
extern "C" void* ixr_host_alloc_pinned(uint64_t bytes) {
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) { fail(IXR_ERR_OUT_OF_MEMORY, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
extern "C" void ixr_host_free_pinned(void* hptr) {
    if (hptr) (void)hipHostFree(hptr);
}
extern "C" int32_t ixr_memcpy_h2d(int32_t device, void* dst, const void* src, uint64_t bytes) {
    HIP_TRY(hipSetDevice(device), return IXR_ERR_HIP);
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), return IXR_ERR_HIP);
    return 0;
}
extern "C" int32_t ixr_memcpy_d2h(int32_t device, void* dst, const void* src, uint64_t bytes) {
    HIP_TRY(hipSetDevice(device), return IXR_ERR_HIP);
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), return IXR_ERR_HIP);
    return 0;
}
extern "C" int32_t ixr_dev_synchronize(int32_t device) {
    HIP_TRY(hipSetDevice(device), return IXR_ERR_HIP);
    HIP_TRY(hipDeviceSynchronize(), return IXR_ERR_HIP);
    return 0;
}

extern "C" void* ixr_stream_create(int32_t device) {
    if (ixr_init(device) != 0) return nullptr;
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) { fail(IXR_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); return nullptr; }
    return (void*)st;
}
extern "C" void ixr_stream_destroy(int32_t device, void* stream) {
    if (!stream) return;
    (void)hipSetDevice(device);
    (void)hipStreamDestroy((hipStream_t)stream);
}
extern "C" int32_t ixr_stream_synchronize(int32_t device, void* stream) {
    HIP_TRY(hipSetDevice(device), return IXR_ERR_HIP);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream), return IXR_ERR_HIP);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------------------------
namespace {

static_assert(kInSlack == IXR_IN_SLACK, "device code and the C ABI must agree on the arena slack");
constexpr uint64_t kDefaultScratchBudget = 16ull << 30;
constexpr int kMaxGroupPacks = 32768;  // grid.y limit is 65535

inline uint64_t align16(uint64_t v) { return (v + 15) & ~15ull; }

enum ListClass { L_ALL = 0, L_N32, L_N64, L_STR, L_RAW, L_DICT, L_COUNT };

struct Group {
    int off[L_COUNT];  // offsets into the device list array
    int cnt[L_COUNT];
    uint64_t scratch;
};

struct KernelStat {
    double ms = 0;
    int launches = 0;
    uint64_t alg_bytes = 0;
};

struct TimedLaunch {
    const char* name;
    hipEvent_t a, b;
};

}  // namespace

struct ixr_plan {
    int device = 0;
    int n = 0;
    std::vector<ixr_pack_desc> descs;
    std::vector<DevPack> hpacks;
    std::vector<Group> groups;
    DevPack* d_packs = nullptr;
    PackMeta* d_metas = nullptr;
    int* d_lists = nullptr;
    uint8_t* d_scratch = nullptr;
    uint64_t scratch_bytes = 0;
    uint64_t in_need = 0, out_need = 0;
    uint64_t tot_in = 0, tot_out = 0, tot_rows = 0;
    bool profiling = false;
    bool overlap = true;  // numeric kernels on the auxiliary stream beside the string kernels (ixr_plan_set_overlap)
    int wave_walk_from = 0;  // see LaunchCtx::wave_walk_from
    bool slice = false;    // see LaunchCtx::slice
    bool fused_str = true; // string rows: index and body copy in one kernel (IXR_SPLIT_STR=1: two kernels with per-row offsets in scratch, for A/B runs)
    std::vector<TimedLaunch> timed;       // launches of the last run (profiling only)
    std::vector<hipEvent_t> event_pool;   // reused across runs
    size_t event_used = 0;
    std::vector<std::pair<std::string, KernelStat>> stats;
    int all_off = 0;  // offset of the identity list (all packs) in d_lists
    std::vector<int> hlists;  // host copy of d_lists
    int group_list_entry(const Group& g, int cls, int k) const { return hlists[(size_t)(g.off[cls] + k)]; }
    // The numeric and the string kernels of a group touch disjoint packs and disjoint scratch; they run on two streams
    // so that the latency-bound string kernels overlap the bandwidth-bound numeric ones.  `aux` forks off the caller's
    // stream at the start of every group and joins it at the end (events), so the caller still sees one ordered stream.
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // ixr_plan_run_host: per group the byte ranges of the two arenas its packs span, two staging buffers per direction,
    // three streams (copy in, decode, copy out) and the events that order them; created by the first call
    struct Span { uint64_t in_lo, in_hi, out_lo, out_hi; };
    std::vector<Span> spans;
    uint8_t* h_din[2] = {nullptr, nullptr};
    uint8_t* h_dout[2] = {nullptr, nullptr};
    hipStream_t s_h2d = nullptr, s_cmp = nullptr, s_d2h = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_dec[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
};

static void host_pipeline_release(ixr_plan* pl);
extern "C" void ixr_plan_destroy(ixr_plan* plan) {
    if (!plan) return;
    (void)hipSetDevice(plan->device);
    if (plan->d_packs) (void)hipFree(plan->d_packs);
    if (plan->d_metas) (void)hipFree(plan->d_metas);
    if (plan->d_lists) (void)hipFree(plan->d_lists);
    if (plan->d_scratch) (void)hipFree(plan->d_scratch);
    for (hipEvent_t e : plan->event_pool) (void)hipEventDestroy(e);
    if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
    if (plan->ev_join) (void)hipEventDestroy(plan->ev_join);
    if (plan->aux) (void)hipStreamDestroy(plan->aux);
    host_pipeline_release(plan);
    delete plan;
}

// Switches for A/B runs and tests (not for production), read from the environment once per process.
namespace {
struct EnvSwitches {
    std::atomic<int> split_str{0};       // IXR_SPLIT_STR=1: string rows by k_str_index + k_str_copy instead of k_str_bases + k_str_fused
    std::atomic<int> slice{1};           // IXR_SLICE=0: sorted numeric dictionaries taken in order go through scratch like the others (default: decoded per tile, the slice path of k_dict_num)
    std::atomic<int> wave_walk_from{0};  // IXR_WAVE_WALK_FROM=n: packs per launch from which every stream is walked by a workgroup of one wave
    EnvSwitches() {
        if (const char* e = getenv("IXR_SPLIT_STR")) split_str = e[0] == '1';
        if (const char* e = getenv("IXR_WAVE_WALK_FROM")) wave_walk_from = atoi(e);
        if (const char* e = getenv("IXR_SLICE")) slice = e[0] != '0';
    }
};
EnvSwitches& env_switches() {
    static EnvSwitches sw;
    return sw;
}
}  // namespace
extern "C" int32_t ixr_debug_switch(const char* name, int32_t value) {
    if (!name) return fail(IXR_ERR_INVALID_ARGUMENT, "null switch name");
    if (!strcmp(name, "split_str")) env_switches().split_str = value != 0;
    else if (!strcmp(name, "wave_walk_from")) env_switches().wave_walk_from = value;
    else if (!strcmp(name, "slice")) env_switches().slice = value != 0;
    else return fail(IXR_ERR_INVALID_ARGUMENT, "unknown switch %s", name);
    return 0;
}

static ixr_plan* plan_create(int32_t device, const ixr_pack_desc* descs, int32_t n_packs, uint64_t scratch_budget,
                             uint64_t dict_cap_override) {
    if (ixr_init(device) != 0) return nullptr;
    if (!descs || n_packs <= 0) { fail(IXR_ERR_INVALID_ARGUMENT, "no packs"); return nullptr; }
    if (scratch_budget == 0) scratch_budget = kDefaultScratchBudget;
    ixr_plan* pl = new ixr_plan();
    pl->device = device;
    pl->fused_str = !env_switches().split_str;
    pl->slice = env_switches().slice != 0;
    pl->n = n_packs;
    pl->descs.assign(descs, descs + n_packs);
    pl->hpacks.resize(n_packs);
    std::vector<int> lists;
    Group g;
    std::vector<int> cls[L_COUNT];
    uint64_t gs = 0, index_total = 0, bulk_max = 0;
    auto flush = [&]() {
        if (cls[L_ALL].empty()) return;
        for (int c = 0; c < L_COUNT; c++) {
            g.off[c] = (int)lists.size();
            g.cnt[c] = (int)cls[c].size();
            lists.insert(lists.end(), cls[c].begin(), cls[c].end());
            cls[c].clear();
        }
        g.scratch = gs;
        pl->groups.push_back(g);
        bulk_max = std::max(bulk_max, gs);
        gs = 0;
    };
    for (int i = 0; i < n_packs; i++) {
        const ixr_pack_desc& d = descs[i];
        if (d.pack_size < 0 || d.data_size < 0 || d.obj_count <= 0 || d.obj_count > IXR_PACK_MAX_ROWS ||
            (d.in_off % IXR_ARENA_ALIGN) || (d.out_off % IXR_ARENA_ALIGN) || d.data_type < IXR_INT || d.data_type > IXR_STRING) {
            fail(IXR_ERR_INVALID_ARGUMENT, "pack %d: bad descriptor (sizes, obj_count 1..65536, 16-byte aligned offsets, data type)", i);
            delete pl;
            return nullptr;
        }
        DevPack& p = pl->hpacks[i];
        p.in_off = d.in_off; p.out_off = d.out_off; p.pack_size = d.pack_size; p.data_size = d.data_size;
        p.obj_count = d.obj_count; p.encode_type = d.encode_type; p.data_type = d.data_type;
        p.width = (d.data_type == IXR_INT || d.data_type == IXR_FLOAT) ? 32 : (d.data_type == IXR_STRING ? 0 : 64);
        p.ds_off = 0;
        ScratchNeed need = pack_scratch_layout(p, dict_cap_override);
        if ((gs + need.bulk > scratch_budget && !cls[L_ALL].empty()) || (int)cls[L_ALL].size() >= kMaxGroupPacks) flush();
        // index offsets are relative to the index area, bulk offsets to the group's bulk area; rebased below
        p.s_idx[0] += index_total; p.s_idx[1] += index_total; p.s_restart += index_total;
        p.s_dict += gs; p.s_lens += gs; p.s_rows += gs;
        index_total += need.index;
        gs += need.bulk;
        cls[L_ALL].push_back(i);
        if (d.encode_type == IXR_ENC_RAW) cls[L_RAW].push_back(i);
        else if (p.width == 32) cls[L_N32].push_back(i);
        else if (p.width == 64) cls[L_N64].push_back(i);
        else cls[L_STR].push_back(i);
        if (d.encode_type == IXR_ENC_DICT_COMPRESS) cls[L_DICT].push_back(i);
        pl->in_need = std::max(pl->in_need, d.in_off + (uint64_t)d.pack_size + IXR_IN_SLACK);
        pl->out_need = std::max(pl->out_need, d.out_off + (uint64_t)d.data_size);
        pl->tot_in += (uint64_t)d.pack_size; pl->tot_out += (uint64_t)d.data_size; pl->tot_rows += (uint64_t)d.obj_count;
    }
    flush();
    if (env_switches().wave_walk_from) pl->wave_walk_from = env_switches().wave_walk_from;
    // one allocation: [index area of all packs][bulk area of the largest group]; make the bulk offsets absolute
    index_total = align16(index_total) + 64;  // the bulk area never starts at offset 0 (k_str_copy may read 15 bytes before a dictionary)
    for (int i = 0; i < n_packs; i++) {
        DevPack& p = pl->hpacks[i];
        if (p.encode_type == IXR_ENC_RAW) continue;
        p.s_dict += index_total; p.s_lens += index_total; p.s_rows += index_total;
    }
    pl->scratch_bytes = index_total + bulk_max + 64;
    // the list of all packs, for the single parse launch
    pl->all_off = (int)lists.size();
    for (int i = 0; i < n_packs; i++) lists.push_back(i);
    hipError_t e;
    pl->hlists = lists;
    if ((e = hipMalloc((void**)&pl->d_packs, sizeof(DevPack) * (size_t)n_packs)) != hipSuccess ||
        (e = hipMalloc((void**)&pl->d_metas, sizeof(PackMeta) * (size_t)n_packs)) != hipSuccess ||
        (e = hipMalloc((void**)&pl->d_lists, sizeof(int) * std::max<size_t>(lists.size(), 1))) != hipSuccess ||
        (e = hipMalloc((void**)&pl->d_scratch, pl->scratch_bytes + 256)) != hipSuccess) {
        fail(IXR_ERR_OUT_OF_MEMORY, "hipMalloc for the plan failed: %s", hipGetErrorString(e));
        ixr_plan_destroy(pl);
        return nullptr;
    }
    if ((e = hipStreamCreateWithFlags(&pl->aux, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&pl->ev_join, hipEventDisableTiming)) != hipSuccess) {
        fail(IXR_ERR_HIP, "creating the plan's auxiliary stream failed: %s", hipGetErrorString(e));
        ixr_plan_destroy(pl);
        return nullptr;
    }
    if ((e = hipMemcpy(pl->d_packs, pl->hpacks.data(), sizeof(DevPack) * (size_t)n_packs, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(pl->d_lists, lists.data(), sizeof(int) * lists.size(), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemset(pl->d_metas, 0, sizeof(PackMeta) * (size_t)n_packs)) != hipSuccess) {
        fail(IXR_ERR_HIP, "uploading the plan failed: %s", hipGetErrorString(e));
        ixr_plan_destroy(pl);
        return nullptr;
    }
    return pl;
}

extern "C" ixr_plan* ixr_plan_create(int32_t device, const ixr_pack_desc* descs, int32_t n_packs, uint64_t scratch_budget) {
    return plan_create(device, descs, n_packs, scratch_budget, 0);
}

// Upper bound of the uncompressed DictStruct of a pack: 4 + dict + 4 + 4n with dict <= dataSize.
static inline uint64_t dict_struct_bound(const ixr_pack_desc& d) { return 8ull + (uint64_t)d.data_size + 4ull * (uint64_t)d.obj_count; }

extern "C" uint64_t ixr_plan_in_bytes(const ixr_plan* p) { return p ? p->in_need : 0; }
extern "C" uint64_t ixr_plan_out_bytes(const ixr_plan* p) { return p ? p->out_need : 0; }
extern "C" uint64_t ixr_plan_scratch_bytes(const ixr_plan* p) { return p ? p->scratch_bytes : 0; }
extern "C" void ixr_plan_totals(const ixr_plan* p, uint64_t* in_bytes, uint64_t* out_bytes, uint64_t* rows) {
    if (!p) return;
    if (in_bytes) *in_bytes = p->tot_in;
    if (out_bytes) *out_bytes = p->tot_out;
    if (rows) *rows = p->tot_rows;
}
extern "C" void ixr_plan_set_profiling(ixr_plan* p, int32_t enabled) {
    if (p) p->profiling = enabled != 0;
}
extern "C" void ixr_plan_set_overlap(ixr_plan* p, int32_t enabled) {
    if (p) p->overlap = enabled != 0;
}

namespace {
// Brackets one launch with events when profiling is on.
struct Timer {
    ixr_plan* pl;
    hipStream_t st;
    hipEvent_t get() {
        if (pl->event_used == pl->event_pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            pl->event_pool.push_back(e);
        }
        return pl->event_pool[pl->event_used++];
    }
    template <typename F> void run(const char* name, int count, F f) {
        if (count <= 0) return;
        if (!pl->profiling) { f(); return; }
        TimedLaunch t{name, get(), get()};
        (void)hipEventRecord(t.a, st);
        f();
        (void)hipEventRecord(t.b, st);
        pl->timed.push_back(t);
    }
};

int32_t check_run_args(ixr_plan* pl, const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes) {
    if (!pl) return fail(IXR_ERR_INVALID_ARGUMENT, "null plan");
    if (!d_in || !d_out) return fail(IXR_ERR_INVALID_ARGUMENT, "null arena");
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return fail(IXR_ERR_INVALID_ARGUMENT, "arenas must be 16-byte aligned");
    if (in_bytes < pl->in_need)
        return fail(IXR_ERR_INVALID_ARGUMENT, "input arena too small: %llu < %llu (packs + %d bytes of slack)",
                    (unsigned long long)in_bytes, (unsigned long long)pl->in_need, IXR_IN_SLACK);
    if (out_bytes < pl->out_need)
        return fail(IXR_ERR_INVALID_ARGUMENT, "output arena too small: %llu < %llu", (unsigned long long)out_bytes,
                    (unsigned long long)pl->out_need);
    return 0;
}

// framing walk of every pack of the plan: one launch, so that the serial chains of all packs overlap
void run_parse_all(ixr_plan* pl, Timer& T, LaunchCtx& c) {
    T.run("k_parse", pl->n, [&] { launch_parse(c, pl->d_lists + pl->all_off, pl->n); });
}
// dictionary construction of one group, numeric packs: DELTA rows and delta-coded dictionary entries
void run_group_front_num(ixr_plan* pl, Timer& T, LaunchCtx& c, const Group& g) {
    const int* L = pl->d_lists;
    T.run("k_delta<u32>", g.cnt[L_N32], [&] { launch_delta(c, L + g.off[L_N32], g.cnt[L_N32], 32, 1); });
    T.run("k_delta<u64>", g.cnt[L_N64], [&] { launch_delta(c, L + g.off[L_N64], g.cnt[L_N64], 64, 1); });
}
// dictionary construction of one group, string packs: length lists, offsets, front-coded bodies
void run_group_front_str(ixr_plan* pl, Timer& T, LaunchCtx& c, const Group& g, int use_ds) {
    const int* L = pl->d_lists;
    T.run("k_delta<u32>(str lens)", g.cnt[L_STR], [&] { launch_delta(c, L + g.off[L_STR], g.cnt[L_STR], 32, 2); });
    T.run("k_str_dict_scan", g.cnt[L_STR], [&] { launch_str_dict_scan(c, L + g.off[L_STR], g.cnt[L_STR]); });
    T.run("k_front_summary", g.cnt[L_STR], [&] { launch_front_summary(c, L + g.off[L_STR], g.cnt[L_STR], use_ds); });
    T.run("k_front_combine", g.cnt[L_STR], [&] { launch_front_combine(c, L + g.off[L_STR], g.cnt[L_STR], use_ds); });
    T.run("k_front_decode", g.cnt[L_STR], [&] { launch_front_decode(c, L + g.off[L_STR], g.cnt[L_STR], use_ds); });
    T.run("k_front_serial", g.cnt[L_STR], [&] { launch_front_serial(c, L + g.off[L_STR], g.cnt[L_STR]); });
}
}  // namespace

// rows of one group: numeric kernels beside the string kernels (two streams), joined at the end
static int32_t run_group_rows(ixr_plan* pl, Timer& T, Timer& TB, LaunchCtx& c, LaunchCtx& cb, const Group& g) {
    const int* L = pl->d_lists;
    hipStream_t st = c.stream;
    const bool fork = pl->overlap && g.cnt[L_STR] > 0 && (g.cnt[L_N32] + g.cnt[L_N64] + g.cnt[L_RAW]) > 0;
    LaunchCtx& cn = fork ? cb : c;  // numeric kernels: on the auxiliary stream when there is string work to overlap
    Timer& TN = fork ? TB : T;
    if (fork) {
        HIP_TRY(hipEventRecord(pl->ev_fork, st), return IXR_ERR_HIP);
        HIP_TRY(hipStreamWaitEvent(pl->aux, pl->ev_fork, 0), return IXR_ERR_HIP);
    }
    run_group_front_num(pl, TN, cn, g);
    TN.run("k_dict_num<u32>", g.cnt[L_N32], [&] { launch_dict_num(cn, L + g.off[L_N32], g.cnt[L_N32], 32); });
    TN.run("k_dict_num<u64>", g.cnt[L_N64], [&] { launch_dict_num(cn, L + g.off[L_N64], g.cnt[L_N64], 64); });
    TN.run("k_copy_raw", g.cnt[L_RAW], [&] { launch_copy_raw(cn, L + g.off[L_RAW], g.cnt[L_RAW]); });
    run_group_front_str(pl, T, c, g, 0);
    if (pl->fused_str) {
        T.run("k_str_bases", g.cnt[L_STR], [&] { launch_str_bases(c, L + g.off[L_STR], g.cnt[L_STR]); });
        T.run("k_str_fused", g.cnt[L_STR], [&] { launch_str_fused(c, L + g.off[L_STR], g.cnt[L_STR]); });
    } else {
        T.run("k_str_index", g.cnt[L_STR], [&] { launch_str_index(c, L + g.off[L_STR], g.cnt[L_STR]); });
        T.run("k_str_copy", g.cnt[L_STR], [&] { launch_str_copy(c, L + g.off[L_STR], g.cnt[L_STR]); });
    }
    if (fork) {  // the next group reuses this group's scratch: both streams must be done with it
        HIP_TRY(hipEventRecord(pl->ev_join, pl->aux), return IXR_ERR_HIP);
        HIP_TRY(hipStreamWaitEvent(st, pl->ev_join, 0), return IXR_ERR_HIP);
    }
    return 0;
}

extern "C" int32_t ixr_plan_run(ixr_plan* pl, const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes, void* stream) {
    int32_t rc = check_run_args(pl, d_in, in_bytes, d_out, out_bytes);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(pl->device), return IXR_ERR_HIP);
    hipStream_t st = (hipStream_t)stream;
    // Timed launches accumulate over successive runs until ixr_plan_status collects them (bounded, in case nobody does).
    if (pl->timed.size() > 200000) { pl->timed.clear(); pl->event_used = 0; }
    Timer T{pl, st}, TB{pl, pl->aux};
    LaunchCtx c{st, pl->d_packs, pl->d_metas, (const uint8_t*)d_in, (uint8_t*)d_out, pl->d_scratch};
    c.wave_walk_from = pl->wave_walk_from;
    c.slice = pl->slice;
    LaunchCtx cb = c;
    cb.stream = pl->aux;
    run_parse_all(pl, T, c);
    for (const Group& g : pl->groups) {
        rc = run_group_rows(pl, T, TB, c, cb, g);
        if (rc) return rc;
    }
    HIP_TRY(hipGetLastError(), return IXR_ERR_HIP);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Host arenas in, host arenas out: the plan's groups travel through two staging buffers per direction, so that the
// copy in of group g+1, the decode of group g and the copy out of group g-1 run at the same time on three streams.
// (What a reader without device-resident segments pays: IntegratedSegment.column(...).pack(...) loads a pack from
// storage into host memory first, StorageColumn.java:78-116.)
// ---------------------------------------------------------------------------------------------------------------
static void host_pipeline_release(ixr_plan* pl) {
    for (int b = 0; b < 2; b++) {
        if (pl->h_din[b]) (void)hipFree(pl->h_din[b]);
        if (pl->h_dout[b]) (void)hipFree(pl->h_dout[b]);
        if (pl->ev_in[b]) (void)hipEventDestroy(pl->ev_in[b]);
        if (pl->ev_dec[b]) (void)hipEventDestroy(pl->ev_dec[b]);
        if (pl->ev_out[b]) (void)hipEventDestroy(pl->ev_out[b]);
        pl->h_din[b] = pl->h_dout[b] = nullptr;
        pl->ev_in[b] = pl->ev_dec[b] = pl->ev_out[b] = nullptr;
    }
    if (pl->s_h2d) (void)hipStreamDestroy(pl->s_h2d);
    if (pl->s_cmp) (void)hipStreamDestroy(pl->s_cmp);
    if (pl->s_d2h) (void)hipStreamDestroy(pl->s_d2h);
    pl->s_h2d = pl->s_cmp = pl->s_d2h = nullptr;
}

static int32_t host_pipeline_setup(ixr_plan* pl) {
    if (!pl->spans.empty()) return 0;
    std::vector<ixr_plan::Span> spans;
    uint64_t max_in = 0, max_out = 0;
    std::vector<std::pair<uint64_t, uint64_t>> outs, ins;  // (offset, bytes) of one group's packs
    for (const Group& g : pl->groups) {
        ixr_plan::Span sp{~0ull, 0, ~0ull, 0};
        outs.clear();
        ins.clear();
        for (int k = 0; k < g.cnt[L_ALL]; k++) {
            const ixr_pack_desc& d = pl->descs[(size_t)pl->group_list_entry(g, L_ALL, k)];
            sp.in_lo = std::min(sp.in_lo, d.in_off);
            sp.in_hi = std::max(sp.in_hi, d.in_off + (uint64_t)d.pack_size + IXR_IN_SLACK);
            sp.out_lo = std::min(sp.out_lo, d.out_off);
            sp.out_hi = std::max(sp.out_hi, align16(d.out_off + (uint64_t)d.data_size));
            outs.push_back({d.out_off, (uint64_t)d.data_size});
            ins.push_back({d.in_off, (uint64_t)d.pack_size});
        }
        // what one group copies out must not cover rows of another group: the packs lie in plan order in the output arena
        if (!spans.empty() && sp.out_lo < spans.back().out_hi)
            return fail(IXR_ERR_INVALID_ARGUMENT, "ixr_plan_run_host needs the packs laid out in plan order in the output arena");
        // A group's rows travel back as ONE copy of its whole span, and the staging buffers are sized by span: the packs of a
        // group must lie back to back (up to the alignment bytes) in the output arena, or the copy would overwrite whatever
        // the caller keeps between them; in the input arena the same rule (plus the slack) bounds the staging memory.
        std::sort(outs.begin(), outs.end());
        for (size_t k = 0; k + 1 < outs.size(); k++) {
            const uint64_t end = outs[k].first + outs[k].second;
            if (outs[k + 1].first < end || outs[k + 1].first - end >= IXR_ARENA_ALIGN)
                return fail(IXR_ERR_INVALID_ARGUMENT,
                            "ixr_plan_run_host needs the packs of a group back to back in the output arena (at most %d alignment bytes "
                            "between two packs' rows): offsets %llu and %llu", IXR_ARENA_ALIGN - 1, (unsigned long long)end,
                            (unsigned long long)outs[k + 1].first);
        }
        std::sort(ins.begin(), ins.end());
        for (size_t k = 0; k + 1 < ins.size(); k++) {
            const uint64_t end = ins[k].first + ins[k].second;
            if (ins[k + 1].first >= end && ins[k + 1].first - end > 4096)
                return fail(IXR_ERR_INVALID_ARGUMENT, "ixr_plan_run_host needs the packs of a group close together in the input arena "
                            "(gaps of at most 4096 bytes): offsets %llu and %llu", (unsigned long long)end, (unsigned long long)ins[k + 1].first);
        }
        max_in = std::max(max_in, sp.in_hi - sp.in_lo);
        max_out = std::max(max_out, sp.out_hi - sp.out_lo);
        spans.push_back(sp);
    }
    hipError_t e = hipSuccess;
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        if ((e = hipMalloc((void**)&pl->h_din[b], align16(max_in) + 16)) != hipSuccess) break;
        if ((e = hipMalloc((void**)&pl->h_dout[b], align16(max_out) + 16)) != hipSuccess) break;
        if ((e = hipEventCreateWithFlags(&pl->ev_in[b], hipEventDisableTiming)) != hipSuccess) break;
        if ((e = hipEventCreateWithFlags(&pl->ev_dec[b], hipEventDisableTiming)) != hipSuccess) break;
        e = hipEventCreateWithFlags(&pl->ev_out[b], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&pl->s_h2d, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&pl->s_cmp, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&pl->s_d2h, hipStreamNonBlocking);
    if (e != hipSuccess) {
        host_pipeline_release(pl);  // nothing half-made stays behind: the next call starts over
        return fail(IXR_ERR_OUT_OF_MEMORY, "staging buffers of the host pipeline: %s", hipGetErrorString(e));
    }
    pl->spans.swap(spans);
    return 0;
}

extern "C" int32_t ixr_plan_run_host(ixr_plan* pl, const void* h_in, uint64_t in_bytes, void* h_out, uint64_t out_bytes) {
    if (!pl) return fail(IXR_ERR_INVALID_ARGUMENT, "null plan");
    if (!h_in || !h_out) return fail(IXR_ERR_INVALID_ARGUMENT, "null arena");
    if (in_bytes < pl->in_need) return fail(IXR_ERR_INVALID_ARGUMENT, "input arena too small: %llu < %llu (packs + %d bytes of slack)",
                                            (unsigned long long)in_bytes, (unsigned long long)pl->in_need, IXR_IN_SLACK);
    if (out_bytes < pl->out_need) return fail(IXR_ERR_INVALID_ARGUMENT, "output arena too small: %llu < %llu",
                                              (unsigned long long)out_bytes, (unsigned long long)pl->out_need);
    HIP_TRY(hipSetDevice(pl->device), return IXR_ERR_HIP);
    int32_t rc = host_pipeline_setup(pl);
    if (rc) return rc;
    if (pl->timed.size() > 200000) { pl->timed.clear(); pl->event_used = 0; }
    hipStream_t st = pl->s_cmp;
    Timer T{pl, st}, TB{pl, pl->aux};
    const uint8_t* hin = (const uint8_t*)h_in;
    uint8_t* hout = (uint8_t*)h_out;
    // A failure in the middle leaves copies in flight that read h_in or write h_out: every exit waits for the four streams
    // first, so that the caller may free or reuse the arenas as soon as the call has returned.
    auto drain = [&]() {
        (void)hipStreamSynchronize(pl->s_d2h);
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(pl->aux);
        (void)hipStreamSynchronize(pl->s_h2d);
    };
#define HOST_TRY(expr) HIP_TRY(expr, { drain(); return IXR_ERR_HIP; })
    for (size_t gi = 0; gi < pl->groups.size(); gi++) {
        const Group& g = pl->groups[gi];
        const ixr_plan::Span& sp = pl->spans[gi];
        const int b = (int)(gi & 1);
        const uint64_t out_len = std::min(sp.out_hi, out_bytes) - sp.out_lo;
        // copy in: the staging buffer is free once the decode two groups back has read it
        if (gi >= 2) HOST_TRY(hipStreamWaitEvent(pl->s_h2d, pl->ev_dec[b], 0));
        HOST_TRY(hipMemcpyAsync(pl->h_din[b], hin + sp.in_lo, sp.in_hi - sp.in_lo, hipMemcpyHostToDevice, pl->s_h2d));
        HOST_TRY(hipEventRecord(pl->ev_in[b], pl->s_h2d));
        // decode: needs this group's input, and the output staging buffer copied out (two groups back)
        HOST_TRY(hipStreamWaitEvent(st, pl->ev_in[b], 0));
        if (gi >= 2) HOST_TRY(hipStreamWaitEvent(st, pl->ev_out[b], 0));
        // the kernels address packs by arena offset: hand them the staging buffers shifted back by the group's first offset
        LaunchCtx c{st, pl->d_packs, pl->d_metas, (const uint8_t*)pl->h_din[b] - sp.in_lo, pl->h_dout[b] - sp.out_lo, pl->d_scratch};
        c.slice = pl->slice;
        LaunchCtx cb = c;
        cb.stream = pl->aux;
        T.run("k_parse", g.cnt[L_ALL], [&] { launch_parse(c, pl->d_lists + g.off[L_ALL], g.cnt[L_ALL]); });
        rc = run_group_rows(pl, T, TB, c, cb, g);
        if (rc) { drain(); return rc; }
        HOST_TRY(hipEventRecord(pl->ev_dec[b], st));
        // copy out
        HOST_TRY(hipStreamWaitEvent(pl->s_d2h, pl->ev_dec[b], 0));
        HOST_TRY(hipMemcpyAsync(hout + sp.out_lo, pl->h_dout[b], out_len, hipMemcpyDeviceToHost, pl->s_d2h));
        HOST_TRY(hipEventRecord(pl->ev_out[b], pl->s_d2h));
    }
#undef HOST_TRY
    HIP_TRY(hipStreamSynchronize(pl->s_d2h), { drain(); return IXR_ERR_HIP; });
    HIP_TRY(hipStreamSynchronize(st), { drain(); return IXR_ERR_HIP; });
    HIP_TRY(hipStreamSynchronize(pl->s_h2d), { drain(); return IXR_ERR_HIP; });
    HIP_TRY(hipGetLastError(), return IXR_ERR_HIP);
    return 0;
}

extern "C" int32_t ixr_plan_run_dict_structs(ixr_plan* pl, const void* d_in, uint64_t in_bytes, void* d_out, uint64_t out_bytes,
                                             const uint64_t* ds_off, void* stream) {
    if (!pl) return fail(IXR_ERR_INVALID_ARGUMENT, "null plan");
    if (!d_in || !d_out || !ds_off) return fail(IXR_ERR_INVALID_ARGUMENT, "null argument");
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return fail(IXR_ERR_INVALID_ARGUMENT, "arenas must be 16-byte aligned");
    if (in_bytes < pl->in_need) return fail(IXR_ERR_INVALID_ARGUMENT, "input arena too small");
    HIP_TRY(hipSetDevice(pl->device), return IXR_ERR_HIP);
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < pl->n; i++) {
        if (ds_off[i] % IXR_ARENA_ALIGN) return fail(IXR_ERR_INVALID_ARGUMENT, "ds_off[%d] not 16-byte aligned", i);
        // The DictStruct size is only known on the device (first 4 bytes of the pack); the kernel checks it against
        // the pack's own promise, the host checks the upper bound a DictStruct of this pack can have.
        const ixr_pack_desc& d = pl->descs[i];
        if (d.encode_type == IXR_ENC_DICT_COMPRESS && ds_off[i] + dict_struct_bound(d) > out_bytes)
            return fail(IXR_ERR_INVALID_ARGUMENT, "ds_off[%d]: the region must hold 8 + dataSize + 4*objCount bytes inside the arena", i);
        pl->hpacks[i].ds_off = ds_off[i];
    }
    HIP_TRY(hipMemcpyAsync(pl->d_packs, pl->hpacks.data(), sizeof(DevPack) * (size_t)pl->n, hipMemcpyHostToDevice, st), return IXR_ERR_HIP);
    HIP_TRY(hipStreamSynchronize(st), return IXR_ERR_HIP);  // hpacks is pageable host memory
    pl->timed.clear();
    pl->event_used = 0;
    Timer T{pl, st};
    LaunchCtx c{st, pl->d_packs, pl->d_metas, (const uint8_t*)d_in, (uint8_t*)d_out, pl->d_scratch};
    // Only DICT_COMPRESS packs have a region in this arena: packs that decode straight to `out_off` (DELTA numbers,
    // front-coded string packs) are skipped by every kernel, dictionaries are built in scratch only.
    c.dict_only = 1;
    const int* L = pl->d_lists;
    run_parse_all(pl, T, c);
    for (const Group& g : pl->groups) {
        run_group_front_num(pl, T, c, g);
        run_group_front_str(pl, T, c, g, 1);
        T.run("k_dict_struct", g.cnt[L_DICT], [&] { launch_dict_struct(c, L + g.off[L_DICT], g.cnt[L_DICT]); });
    }
    HIP_TRY(hipGetLastError(), return IXR_ERR_HIP);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The same decode for callers that cannot allocate: no hipMalloc / hipFree, no events, no second stream.  Descriptors go
// to the device as a kernel argument; every working array lives in this GPU's static scratch array.
// ---------------------------------------------------------------------------------------------------------------
static_assert(sizeof(SmallDesc) == sizeof(ixr_pack_desc), "device and ABI descriptor layouts must match");
static_assert(IXR_STATIC_MAX_PACKS == kStaticMaxPacks, "device code and the C ABI must agree on the batch limit");

// Decodes 1..IXR_STATIC_MAX_PACKS packs with every working array inside `block` (IXR_STATIC_SCRATCH_BYTES of device memory
// laid out by static_layout).  Launches only: nothing is allocated, nothing is waited for.  `too_big`, when given, is set
// instead of failing when the batch needs more working memory than the block holds.
//   what = SMALL_ROWS          decoded rows at out_off (what ixr_plan_run does)
//          SMALL_DICT_STRUCT   the uncompressed DictStruct of a DICT_COMPRESS pack at the start of d_out (one pack per call;
//                              d_out holds 8 + data_size + 4 * obj_count bytes)
//          SMALL_FROM_STRUCT   the "pack" is an uncompressed DictStruct, its rows go to out_off (one pack per call);
//                              dict_cap = bytes reserved for the dictionary it carries
enum SmallWhat { SMALL_ROWS = 0, SMALL_DICT_STRUCT = 1, SMALL_FROM_STRUCT = 2 };
static int32_t decode_small(const ixr_pack_desc* descs, int32_t n_packs, const void* d_in, uint64_t in_bytes, void* d_out,
                            uint64_t out_bytes, int32_t* d_status, hipStream_t st, uint8_t* block, bool* too_big,
                            SmallWhat what = SMALL_ROWS, uint64_t dict_cap = 0) {
    StaticBatch batch;
    memset(&batch, 0, sizeof(batch));
    batch.n = n_packs;
    batch.dict_cap = dict_cap;
    if (what != SMALL_ROWS && n_packs != 1) return fail(IXR_ERR_BUG, "the DictStruct forms take one pack per call");
    for (int i = 0; i < n_packs; i++) {
        const ixr_pack_desc& d = descs[i];
        if (d.pack_size < 0 || d.data_size < 0 || d.obj_count <= 0 || d.obj_count > IXR_PACK_MAX_ROWS ||
            (d.in_off % IXR_ARENA_ALIGN) || (d.out_off % IXR_ARENA_ALIGN) || d.data_type < IXR_INT || d.data_type > IXR_STRING)
            return fail(IXR_ERR_INVALID_ARGUMENT, "pack %d: bad descriptor (sizes, obj_count 1..65536, 16-byte aligned offsets, data type)", i);
        // written so that offsets near 2^64 cannot wrap past the check
        const uint64_t in_need = (uint64_t)d.pack_size + IXR_IN_SLACK, out_need = (uint64_t)d.data_size;
        if (d.in_off > in_bytes || in_need > in_bytes - d.in_off)
            return fail(IXR_ERR_INVALID_ARGUMENT, "pack %d: the input arena must hold the pack and %d bytes of slack", i, IXR_IN_SLACK);
        if (d.out_off > out_bytes || out_need > out_bytes - d.out_off) return fail(IXR_ERR_INVALID_ARGUMENT, "pack %d: output outside the arena", i);
        memcpy(&batch.d[i], &d, sizeof(SmallDesc));
    }
    int counts[SL_COUNT];
    const uint64_t need = static_layout(batch.d, n_packs, dict_cap, nullptr, nullptr, counts);
    if (need > IXR_STATIC_SCRATCH_BYTES) {
        if (too_big) { *too_big = true; return 0; }
        return fail(IXR_ERR_INVALID_ARGUMENT, "the batch needs %llu bytes of working memory, the static scratch holds %llu: pass fewer packs per call",
                    (unsigned long long)need, (unsigned long long)IXR_STATIC_SCRATCH_BYTES);
    }
    launch_static_setup(st, batch, block);
    LaunchCtx c{st, (const DevPack*)block, (PackMeta*)(block + kStaticMetaOff), (const uint8_t*)d_in, (uint8_t*)d_out, block + kStaticHeader};
    const int* lists = (const int*)(block + kStaticListOff);
    const int *all = lists + SL_ALL * kStaticMaxPacks, *n32 = lists + SL_N32 * kStaticMaxPacks, *n64 = lists + SL_N64 * kStaticMaxPacks,
              *str = lists + SL_STR * kStaticMaxPacks, *raw = lists + SL_RAW * kStaticMaxPacks;
    if (what == SMALL_FROM_STRUCT) {
        // PackCompressor_VLT.getDataFromDictStruct (:188-213): plain entries and 4-byte ids straight out of the DictStruct
        launch_ds_meta(c, all, 1);
        if (counts[SL_STR]) {
            launch_str_dict_scan(c, all, 1);
            launch_ds_rows(c, all, 1, 2);
            launch_row_scan(c, all, 1);
            launch_str_copy(c, all, 1);
        } else {
            launch_delta(c, all, 1, counts[SL_N32] ? 32 : 64, 1);
            launch_ds_rows(c, all, 1, counts[SL_N32] ? 0 : 1);
        }
    } else {
        const int use_ds = what == SMALL_DICT_STRUCT;
        c.dict_only = use_ds;
        launch_parse(c, all, counts[SL_ALL]);
        launch_delta(c, n32, counts[SL_N32], 32, 1);
        launch_delta(c, n64, counts[SL_N64], 64, 1);
        if (!use_ds) {
            launch_dict_num(c, n32, counts[SL_N32], 32);
            launch_dict_num(c, n64, counts[SL_N64], 64);
            launch_copy_raw(c, raw, counts[SL_RAW]);
        }
        launch_delta(c, str, counts[SL_STR], 32, 2);
        launch_str_dict_scan(c, str, counts[SL_STR]);
        launch_front_summary(c, str, counts[SL_STR], use_ds);
        launch_front_combine(c, str, counts[SL_STR], use_ds);
        launch_front_decode(c, str, counts[SL_STR], use_ds);
        launch_front_serial(c, str, counts[SL_STR]);
        if (use_ds) {
            launch_dict_struct(c, all, 1);  // PackCompressor_VLT.getDictStruct (:172-186)
        } else {
            launch_str_bases(c, str, counts[SL_STR]);
            launch_str_fused(c, str, counts[SL_STR]);
        }
    }
    launch_export_status(st, c.metas, n_packs, d_status);
    HIP_TRY(hipGetLastError(), return IXR_ERR_HIP);
    return 0;
}

extern "C" int32_t ixr_decode_batch_static(int32_t device, const ixr_pack_desc* descs, int32_t n_packs, const void* d_in,
                                           uint64_t in_bytes, void* d_out, uint64_t out_bytes, int32_t* d_status, void* stream) {
    if (!descs || !d_in || !d_out || !d_status) return fail(IXR_ERR_INVALID_ARGUMENT, "null argument");
    if (n_packs < 1 || n_packs > IXR_STATIC_MAX_PACKS)
        return fail(IXR_ERR_INVALID_ARGUMENT, "%d packs: this entry point takes 1..%d packs per call (use a plan for more)", n_packs, IXR_STATIC_MAX_PACKS);
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15) || ((uintptr_t)d_status & 3)) return fail(IXR_ERR_INVALID_ARGUMENT, "misaligned device pointer");
    int32_t rc = ixr_init(device);
    if (rc) return rc;
    if (device >= 64) return fail(IXR_ERR_INVALID_ARGUMENT, "device %d: the static scratch is kept for devices 0..63", device);
    uint8_t* block = g_static_ptr[device].load(std::memory_order_acquire);
    if (!block) return fail(IXR_ERR_BUG, "static scratch of device %d was not resolved", device);
    return decode_small(descs, n_packs, d_in, in_bytes, d_out, out_bytes, d_status, (hipStream_t)stream, block, nullptr);
}

// Attribution of a pack's algorithmic bytes (packSize + dataSize) to the kernels that consume / produce them.
static void attribute_bytes(ixr_plan* pl, const std::vector<PackMeta>& metas, std::map<std::string, uint64_t>& by) {
    for (int i = 0; i < pl->n; i++) {
        const ixr_pack_desc& d = pl->descs[i];
        const PackMeta& m = metas[(size_t)i];
        const uint64_t ps = (uint64_t)d.pack_size, ds = (uint64_t)d.data_size, n = (uint64_t)d.obj_count;
        const bool w32 = d.data_type == IXR_INT || d.data_type == IXR_FLOAT;
        if (m.status < 0) continue;
        switch (m.kind) {
            case K_RAW: by["k_copy_raw"] += ps + ds; break;
            case K_DELTA_NUM: by[w32 ? "k_delta<u32>" : "k_delta<u64>"] += ps + ds; break;
            case K_DICT_NUM:
                by[w32 ? "k_delta<u32>" : "k_delta<u64>"] += m.ids_off;                     // header + entries
                by[w32 ? "k_dict_num<u32>" : "k_dict_num<u64>"] += (ps - m.ids_off) + ds;   // id stream in, rows out
                break;
            case K_DICT_STR: {
                const uint64_t index = 4 * (n + 1), body = ds > index ? ds - index : 0;
                if (m.entries_plain) by["k_str_dict_scan"] += m.ids_off;
                else { by["k_delta<u32>(str lens)"] += m.sfx_off; by[m.front_blocks ? "k_front_decode" : "k_front_serial"] += m.sfx_len; }
                if (pl->fused_str) {
                    by["k_str_fused"] += (ps - m.ids_off) + index + body;  // id stream in; row index and body out
                } else {
                    by["k_str_index"] += (ps - m.ids_off) + index;
                    by["k_str_copy"] += body;
                }
                break;
            }
            case K_DELTA_STR: {
                const uint64_t index = 4 * (n + 1), body = ds > index ? ds - index : 0;
                by["k_delta<u32>(str lens)"] += m.sfx_off;
                by["k_str_dict_scan"] += index;
                by["k_front_serial"] += m.sfx_len + body;
                break;
            }
            default: break;
        }
    }
}

static void collect_times(ixr_plan* pl, const std::vector<PackMeta>& metas) {
    pl->stats.clear();
    std::map<std::string, size_t> idx;
    for (const TimedLaunch& t : pl->timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) ms = 0;
        auto it = idx.find(t.name);
        if (it == idx.end()) {
            idx[t.name] = pl->stats.size();
            pl->stats.push_back({t.name, KernelStat()});
            it = idx.find(t.name);
        }
        pl->stats[it->second].second.ms += ms;
        pl->stats[it->second].second.launches += 1;
    }
    std::map<std::string, uint64_t> by;
    attribute_bytes(pl, metas, by);
    for (auto& kv : pl->stats) kv.second.alg_bytes = by[kv.first];
    pl->timed.clear();
    pl->event_used = 0;
}

extern "C" int32_t ixr_plan_status(ixr_plan* pl, int32_t* status, void* stream) {
    if (!pl || !status) return fail(IXR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(pl->device), return IXR_ERR_HIP);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream), return IXR_ERR_HIP);
    std::vector<PackMeta> metas((size_t)pl->n);
    HIP_TRY(hipMemcpy(metas.data(), pl->d_metas, sizeof(PackMeta) * (size_t)pl->n, hipMemcpyDeviceToHost), return IXR_ERR_HIP);
    int bad = 0;
    for (int i = 0; i < pl->n; i++) {
        status[i] = metas[i].status;
        if (status[i] != pl->descs[i].pack_size) bad++;
    }
    if (pl->profiling || !pl->timed.empty()) collect_times(pl, metas);
    return bad;
}

extern "C" int32_t ixr_plan_kernel_times(ixr_plan* pl, const char** names, float* ms, int32_t* launches, int32_t cap) {
    if (!pl) return 0;
    int n = 0;
    for (auto& kv : pl->stats) {
        if (n >= cap) break;
        if (names) names[n] = kv.first.c_str();
        if (ms) ms[n] = (float)kv.second.ms;
        if (launches) launches[n] = kv.second.launches;
        n++;
    }
    return n;
}

extern "C" int32_t ixr_plan_checksums(ixr_plan* pl, const void* d_out, uint64_t out_bytes, uint64_t* checksums, void* stream) {
    if (!pl || !d_out || !checksums) return fail(IXR_ERR_INVALID_ARGUMENT, "null argument");
    if (out_bytes < pl->out_need) return fail(IXR_ERR_INVALID_ARGUMENT, "output arena too small");
    HIP_TRY(hipSetDevice(pl->device), return IXR_ERR_HIP);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* d_sums = nullptr;
    HIP_TRY(hipMalloc((void**)&d_sums, sizeof(unsigned long long) * (size_t)pl->n), return IXR_ERR_OUT_OF_MEMORY);
    LaunchCtx c{st, pl->d_packs, pl->d_metas, nullptr, (uint8_t*)d_out, pl->d_scratch};
    launch_checksum(c, pl->n, d_sums);
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(checksums, d_sums, sizeof(unsigned long long) * (size_t)pl->n, hipMemcpyDeviceToHost);
    (void)hipFree(d_sums);
    if (e != hipSuccess) return fail(IXR_ERR_HIP, "checksum failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int32_t ixr_plan_kernel_bytes(ixr_plan* pl, uint64_t* alg_bytes, int32_t cap) {
    if (!pl || !alg_bytes) return 0;
    int n = 0;
    for (auto& kv : pl->stats) {
        if (n >= cap) break;
        alg_bytes[n++] = kv.second.alg_bytes;
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------------------
// single pack, host in / host out
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
};

// Everything one ixr_decode_pack call needs on a device, allocated once and handed from call to call: a stream, device
// input / output buffers, a working block laid out like the static scratch, pinned staging on the host.  A call takes a
// context from the device's pool and puts it back, so there are as many contexts as calls that ever overlapped.  The
// pool lives until the process ends (the HIP runtime may be gone by the time static destructors run).
struct PackCtx {
    hipStream_t stream = nullptr;
    uint8_t* d_in = nullptr;
    uint8_t* d_out = nullptr;
    uint8_t* d_block = nullptr;
    int32_t* d_status = nullptr;
    uint8_t* h_in = nullptr;   // pinned
    uint8_t* h_out = nullptr;  // pinned
    int32_t* h_status = nullptr;
    uint64_t in_cap = 0, out_cap = 0;

    bool reserve(uint64_t in_bytes, uint64_t out_bytes) {
        if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
        if (!d_block && hipMalloc((void**)&d_block, IXR_STATIC_SCRATCH_BYTES) != hipSuccess) return false;
        if (!d_status && hipMalloc((void**)&d_status, 64) != hipSuccess) return false;
        if (!h_status && hipHostMalloc((void**)&h_status, 64, hipHostMallocDefault) != hipSuccess) return false;
        if (in_bytes > in_cap) {
            const uint64_t cap = std::max<uint64_t>(in_bytes, 1ull << 20);
            if (d_in) (void)hipFree(d_in);
            if (h_in) (void)hipHostFree(h_in);
            d_in = nullptr; h_in = nullptr; in_cap = 0;
            if (hipMalloc((void**)&d_in, cap) != hipSuccess || hipHostMalloc((void**)&h_in, cap, hipHostMallocDefault) != hipSuccess) return false;
            in_cap = cap;
        }
        if (out_bytes > out_cap) {
            const uint64_t cap = std::max<uint64_t>(out_bytes, 1ull << 20);
            if (d_out) (void)hipFree(d_out);
            if (h_out) (void)hipHostFree(h_out);
            d_out = nullptr; h_out = nullptr; out_cap = 0;
            if (hipMalloc((void**)&d_out, cap) != hipSuccess || hipHostMalloc((void**)&h_out, cap, hipHostMallocDefault) != hipSuccess) return false;
            out_cap = cap;
        }
        return true;
    }
    // called with the context's device current
    void release() {
        if (stream) (void)hipStreamDestroy(stream);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        if (d_block) (void)hipFree(d_block);
        if (d_status) (void)hipFree(d_status);
        if (h_in) (void)hipHostFree(h_in);
        if (h_out) (void)hipHostFree(h_out);
        if (h_status) (void)hipHostFree(h_status);
    }
};
// A device keeps at most kPoolIdleMax idle contexts (calls that overlap beyond that build and drop their own), and a context
// whose buffers grew beyond kPoolKeepBytes (a pack of very long strings) is dropped after its call instead of kept: what stays
// allocated per device is bounded by kPoolIdleMax * (32 MiB working block + 2 * kPoolKeepBytes device + 2 * kPoolKeepBytes pinned).
constexpr size_t kPoolIdleMax = 8;
constexpr uint64_t kPoolKeepBytes = 16ull << 20;
struct PackCtxPool {
    std::mutex mu;
    std::vector<PackCtx*> idle[64];
};
PackCtxPool& pack_ctx_pool() {
    static PackCtxPool* pool = new PackCtxPool();  // never destroyed, see above
    return *pool;
}
struct PackCtxLease {
    int device;
    PackCtx* ctx;
    explicit PackCtxLease(int dev) : device(dev), ctx(nullptr) {
        PackCtxPool& pool = pack_ctx_pool();
        {
            std::lock_guard<std::mutex> g(pool.mu);
            if (!pool.idle[dev].empty()) { ctx = pool.idle[dev].back(); pool.idle[dev].pop_back(); }
        }
        if (!ctx) ctx = new PackCtx();
    }
    ~PackCtxLease() {
        PackCtxPool& pool = pack_ctx_pool();
        if (ctx->in_cap <= kPoolKeepBytes && ctx->out_cap <= kPoolKeepBytes) {
            std::lock_guard<std::mutex> g(pool.mu);
            if (pool.idle[device].size() < kPoolIdleMax) { pool.idle[device].push_back(ctx); return; }
        }
        if (hipSetDevice(device) == hipSuccess) ctx->release();
        delete ctx;
    }
};
}  // namespace

// One pack through a pooled context: one copy in, the small-batch kernels, status and result out, one wait.  `d` describes
// the pack as the kernels see it, `in_size` bytes of `in` go in, `copy_out` bytes come back from the start of the device
// output, `dev_out_bytes` is the size the device output needs.  Returns 1 when the pack needs more working memory than a
// context holds (the caller then takes the plan path), else 0 with the call's result in *rc_out.
static int32_t single_pack_pooled(int32_t device, const ixr_pack_desc& d, const void* in, void* out, uint64_t copy_out,
                                  uint64_t dev_out_bytes, SmallWhat what, uint64_t dict_cap, int32_t* rc_out) {
    if (device < 0 || device >= 64) return 1;
    int32_t rc = ixr_init(device);
    if (rc) { *rc_out = rc; return 0; }
    {   // nothing is reserved for a pack the working block cannot take
        SmallDesc sd;
        memcpy(&sd, &d, sizeof(sd));
        if (static_layout(&sd, 1, dict_cap, nullptr, nullptr, nullptr) > IXR_STATIC_SCRATCH_BYTES) return 1;
    }
    const uint64_t in_bytes = align16((uint64_t)d.pack_size) + IXR_IN_SLACK;
    const uint64_t out_bytes = std::max(align16(std::max<uint64_t>(copy_out, 16)) + 16, dev_out_bytes);
    PackCtxLease lease(device);
    PackCtx& x = *lease.ctx;
    if (hipSetDevice(device) != hipSuccess || !x.reserve(in_bytes, out_bytes)) {
        *rc_out = fail(IXR_ERR_OUT_OF_MEMORY, "device or pinned memory for a single-pack context could not be allocated");
        return 0;
    }
    memcpy(x.h_in, in, (size_t)d.pack_size);
    memset(x.h_in + d.pack_size, 0, (size_t)(in_bytes - (uint64_t)d.pack_size));
    bool too_big = false;
    hipError_t e = hipMemcpyAsync(x.d_in, x.h_in, in_bytes, hipMemcpyHostToDevice, x.stream);
    if (e == hipSuccess) {
        rc = decode_small(&d, 1, x.d_in, in_bytes, x.d_out, out_bytes, x.d_status, x.stream, x.d_block, &too_big, what, dict_cap);
        if (rc || too_big) {
            (void)hipStreamSynchronize(x.stream);
            if (too_big) return 1;
            *rc_out = rc;
            return 0;
        }
        e = hipMemcpyAsync(x.h_status, x.d_status, 4, hipMemcpyDeviceToHost, x.stream);
    }
    if (e == hipSuccess && copy_out > 0) e = hipMemcpyAsync(x.h_out, x.d_out, (size_t)copy_out, hipMemcpyDeviceToHost, x.stream);
    if (e == hipSuccess) {
        // A decode is tens to hundreds of microseconds: the calling thread polls the stream for up to kPollNs before it
        // falls back to a blocking wait, whose wake-up alone can take several times a short decode.
        constexpr int64_t kPollNs = 1000000;
        const auto t0 = std::chrono::steady_clock::now();
        while ((e = hipStreamQuery(x.stream)) == hipErrorNotReady &&
               std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() < kPollNs) {}
        if (e == hipErrorNotReady) e = hipStreamSynchronize(x.stream);
    }
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(x.stream);
        *rc_out = fail(IXR_ERR_HIP, "single-pack decode failed: %s", hipGetErrorString(e));
        return 0;
    }
    const int32_t status = x.h_status[0];
    if (status < 0) {
        *rc_out = fail(status, "decode failed with code %d", status);
        return 0;
    }
    memcpy(out, x.h_out, (size_t)copy_out);
    *rc_out = status;
    return 0;
}

static int32_t single_pack(int32_t device, int32_t data_type, int32_t encode_type, int32_t obj_count, const void* in,
                           int32_t in_size, void* out, int32_t out_limit, int mode /*0 rows, 1 dict struct, 2 rows from ds*/) {
    if (!in || !out || in_size < 0 || out_limit < 0) return fail(IXR_ERR_INVALID_ARGUMENT, "bad buffer");
    if (obj_count == 0) return 0;  // every reference codec returns 0 for valCount == 0
    if (data_type < IXR_INT || data_type > IXR_STRING) return fail(IXR_ERR_UNSUPPORTED_TYPE, "unsupported data type %d", data_type);
    ixr_pack_desc d;
    memset(&d, 0, sizeof(d));
    d.in_off = 0; d.out_off = 0; d.pack_size = in_size; d.data_size = out_limit; d.obj_count = obj_count;
    d.encode_type = encode_type; d.data_type = data_type;
    int32_t ds_size = 0;
    if (mode == 1) {
        if (in_size < 4) return fail(IXR_ERR_DATA_CORRUPTED, "pack shorter than its header");
        memcpy(&ds_size, in, 4);
        if (ds_size < 8) return fail(IXR_ERR_DATA_CORRUPTED, "bad dictStructSize %d", ds_size);
        if (ds_size > out_limit) return fail(IXR_ERR_BUFFER_OVERFLOW, "DictStruct of %d bytes does not fit %d", ds_size, out_limit);
        // rows are not produced in this mode; data_size only bounds scratch
        const int w = (data_type == IXR_INT || data_type == IXR_FLOAT) ? 4 : 8;
        d.data_size = data_type == IXR_STRING ? ds_size : obj_count * w;
    }
    uint64_t dict_cap_override = 0;
    if (mode == 2) {
        // the "pack" is the DictStruct itself; scratch is laid out as for a dictionary pack that can hold its dictionary
        d.encode_type = IXR_ENC_DICT_COMPRESS;
        if (data_type != IXR_STRING) {
            const int w = (data_type == IXR_INT || data_type == IXR_FLOAT) ? 4 : 8;
            if ((int64_t)obj_count * w > out_limit) return fail(IXR_ERR_BUFFER_OVERFLOW, "output too small");
            d.data_size = obj_count * w;
        } else if ((int64_t)out_limit < 4 * ((int64_t)obj_count + 1)) {
            // the row index alone does not fit: StringsStruct.Builder would run off the buffer (DictStruct.java:97-110)
            return fail(IXR_ERR_BUFFER_OVERFLOW, "output too small for the index of %d strings", obj_count);
        }
        dict_cap_override = (uint64_t)in_size;
    }
    if (obj_count <= IXR_PACK_MAX_ROWS) {
        // the pooled path: no allocation, no plan (all three calls DataPack.decompress makes, DataPack.java:166-186)
        int32_t rc_fast = 0;
        const uint64_t copy_out = mode == 1 ? (uint64_t)ds_size : (uint64_t)d.data_size;
        const uint64_t dev_out = mode == 1 ? align16(dict_struct_bound(d)) + 16 : 0;
        if (single_pack_pooled(device, d, in, out, copy_out, dev_out, (SmallWhat)mode, dict_cap_override, &rc_fast) == 0) return rc_fast;
    }
    ixr_plan* pl = plan_create(device, &d, 1, 0, dict_cap_override);
    if (!pl) return g_code < 0 ? g_code : IXR_ERR_HIP;  // plan_create recorded why (no device, bad descriptor, memory)
    DevBuf din, dout;
    const uint64_t in_bytes = align16((uint64_t)in_size) + IXR_IN_SLACK;
    uint64_t out_bytes = align16((uint64_t)std::max(out_limit, 16)) + 16;
    if (mode == 1) out_bytes = std::max(out_bytes, align16(dict_struct_bound(d)));
    int32_t rc = 0;
    if (!din.alloc(in_bytes) || !dout.alloc(out_bytes)) rc = fail(IXR_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    if (!rc && hipMemset(din.p, 0, in_bytes) != hipSuccess) rc = fail(IXR_ERR_HIP, "hipMemset failed");
    if (!rc && hipMemcpy(din.p, in, (size_t)in_size, hipMemcpyHostToDevice) != hipSuccess) rc = fail(IXR_ERR_HIP, "H2D failed");
    int32_t status = 0;
    if (!rc) {
        if (mode == 0) rc = ixr_plan_run(pl, din.p, in_bytes, dout.p, out_bytes, nullptr);
        else if (mode == 1) {
            uint64_t off = 0;
            rc = ixr_plan_run_dict_structs(pl, din.p, in_bytes, dout.p, out_bytes, &off, nullptr);
        } else {
            rc = check_run_args(pl, din.p, in_bytes, dout.p, out_bytes);
            if (!rc) {
                LaunchCtx c{nullptr, pl->d_packs, pl->d_metas, (const uint8_t*)din.p, (uint8_t*)dout.p, pl->d_scratch};
                const int* L = pl->d_lists;  // one pack: every list is {0}
                launch_ds_meta(c, L, 1);
                if (data_type == IXR_STRING) {
                    launch_str_dict_scan(c, L, 1);
                    launch_ds_rows(c, L, 1, 2);
                    launch_row_scan(c, L, 1);
                    launch_str_copy(c, L, 1);
                } else {
                    const int w = (data_type == IXR_INT || data_type == IXR_FLOAT) ? 32 : 64;
                    launch_delta(c, L, 1, w, 1);
                    launch_ds_rows(c, L, 1, w == 32 ? 0 : 1);
                }
                if (hipGetLastError() != hipSuccess) rc = fail(IXR_ERR_HIP, "kernel launch failed");
            }
        }
    }
    if (!rc) {
        int bad = ixr_plan_status(pl, &status, nullptr);
        (void)bad;
        if (status < 0) {
            rc = fail(status, "decode failed with code %d", status);
        } else {
            uint64_t copy = (uint64_t)out_limit;
            if (mode == 1) copy = (uint64_t)ds_size;
            if (hipMemcpy(out, dout.p, copy, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(IXR_ERR_HIP, "D2H failed");
            else rc = status;
        }
    }
    ixr_plan_destroy(pl);
    return rc;
}

extern "C" int32_t ixr_decode_pack(int32_t device, int32_t data_type, int32_t encode_type, int32_t obj_count, const void* in,
                                   int32_t in_size, void* out, int32_t out_limit) {
    return single_pack(device, data_type, encode_type, obj_count, in, in_size, out, out_limit, 0);
}
extern "C" int32_t ixr_get_dict_struct(int32_t device, int32_t data_type, int32_t obj_count, const void* in, int32_t in_size,
                                       void* out, int32_t out_limit) {
    return single_pack(device, data_type, IXR_ENC_DICT_COMPRESS, obj_count, in, in_size, out, out_limit, 1);
}
extern "C" int32_t ixr_get_data_from_dict_struct(int32_t device, int32_t data_type, int32_t obj_count, const void* dict_struct,
                                                 int32_t ds_size, void* out, int32_t out_limit) {
    int32_t rc = single_pack(device, data_type, IXR_ENC_DICT_COMPRESS, obj_count, dict_struct, ds_size, out, out_limit, 2);
    if (rc < 0) return rc;
    // bytes written: numbers n*w ; strings index + body
    if (data_type != IXR_STRING) return obj_count * ((data_type == IXR_INT || data_type == IXR_FLOAT) ? 4 : 8);
    int32_t total = 0;
    memcpy(&total, (const uint8_t*)out + 4ull * (uint64_t)obj_count, 4);
    return 4 * (obj_count + 1) + total;
}

// ---------------------------------------------------------------------------------------------------------------
// engine hand-back (DefaultPackReader.java:164-241)
// ---------------------------------------------------------------------------------------------------------------
static_assert(sizeof(HandbackDesc) == sizeof(ixr_handback_desc), "device and ABI descriptor layouts must match");

extern "C" uint64_t ixr_handback_bound(const ixr_handback_desc* d) {
    if (!d || d->row_count < 0) return 0;
    const uint64_t c = (uint64_t)d->row_count;
    switch (d->sql_type) {
        case IXR_SQL_INT: case IXR_SQL_FLOAT: case IXR_SQL_DATE: return align16(4 * c);
        case IXR_SQL_BIGINT: case IXR_SQL_DOUBLE: case IXR_SQL_DATETIME: return align16(8 * c);
        case IXR_SQL_VARCHAR: {
            const uint64_t index = 4ull * ((uint64_t)d->obj_count + 1);
            const uint64_t body = (uint64_t)d->data_size > index ? (uint64_t)d->data_size - index : 0;
            return align16(8 * c + body);
        }
        default: return 16;
    }
}

extern "C" int32_t ixr_handback_run(int32_t device, const ixr_handback_desc* descs, int32_t n, const void* d_decoded,
                                    uint64_t decoded_bytes, void* d_out, uint64_t out_bytes, int32_t* status, void* stream) {
    if (ixr_init(device) != 0) return IXR_ERR_NO_DEVICE;
    if (!descs || n <= 0 || !d_decoded || !d_out || !status) return fail(IXR_ERR_INVALID_ARGUMENT, "null argument or no descriptors");
    if (((uintptr_t)d_decoded & 15) || ((uintptr_t)d_out & 15)) return fail(IXR_ERR_INVALID_ARGUMENT, "arenas must be 16-byte aligned");
    for (int i = 0; i < n; i++) {
        const ixr_handback_desc& d = descs[i];
        if (d.obj_count <= 0 || d.obj_count > IXR_PACK_MAX_ROWS || d.data_size < 0 || d.row_offset < 0 || d.row_count < 0 ||
            (int64_t)d.row_offset + d.row_count > d.obj_count || (d.src_off % IXR_ARENA_ALIGN) || (d.dst_off % IXR_ARENA_ALIGN))
            return fail(IXR_ERR_INVALID_ARGUMENT, "hand-back descriptor %d: bad row range, sizes or alignment", i);
        if (d.src_off + (uint64_t)d.data_size > decoded_bytes) return fail(IXR_ERR_INVALID_ARGUMENT, "hand-back descriptor %d: pack outside the decoded arena", i);
        if (d.dst_off + ixr_handback_bound(&d) > out_bytes) return fail(IXR_ERR_INVALID_ARGUMENT, "hand-back descriptor %d: output outside the arena", i);
    }
    hipStream_t st = (hipStream_t)stream;
    HandbackDesc* dd = nullptr;
    int32_t* ds = nullptr;
    int32_t rc = 0;
    if (hipMalloc((void**)&dd, sizeof(HandbackDesc) * (size_t)n) != hipSuccess || hipMalloc((void**)&ds, sizeof(int32_t) * (size_t)n) != hipSuccess)
        rc = fail(IXR_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    if (!rc && hipMemcpyAsync(dd, descs, sizeof(HandbackDesc) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess) rc = fail(IXR_ERR_HIP, "H2D failed");
    if (!rc) {
        launch_handback(st, dd, n, (const uint8_t*)d_decoded, (uint8_t*)d_out, ds);
        if (hipGetLastError() != hipSuccess) rc = fail(IXR_ERR_HIP, "kernel launch failed");
    }
    if (!rc && hipMemcpyAsync(status, ds, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess) rc = fail(IXR_ERR_HIP, "D2H failed");
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = fail(IXR_ERR_HIP, "stream synchronise failed");
    if (dd) (void)hipFree(dd);
    if (ds) (void)hipFree(ds);
    if (rc) return rc;
    int bad = 0;
    for (int i = 0; i < n; i++) bad += status[i] < 0;
    return bad;
}

// ---------------------------------------------------------------------------------------------------------------
// rows picked by a bitmap (DataPack.foreach(BitMap position, XxxSetter), DataPack.java:273-306, 708-762)
// ---------------------------------------------------------------------------------------------------------------
static_assert(sizeof(SelectDesc) == sizeof(ixr_select_desc), "device and ABI descriptor layouts must match");

extern "C" uint64_t ixr_select_bound(const ixr_select_desc* d) {
    if (!d || d->obj_count < 0 || d->data_size < 0) return 0;
    const uint64_t c = (uint64_t)d->obj_count;
    if (d->data_type == IXR_STRING) {
        const uint64_t index = 4ull * (c + 1);
        const uint64_t body = (uint64_t)d->data_size > index ? (uint64_t)d->data_size - index : 0;
        return align16(12 * c + body);
    }
    const uint64_t w = (d->data_type == IXR_INT || d->data_type == IXR_FLOAT) ? 4 : 8;
    return align16(((4 * c + 7) & ~7ull) + w * c);
}

extern "C" int32_t ixr_select_run(int32_t device, const ixr_select_desc* descs, int32_t n, const void* d_decoded,
                                  uint64_t decoded_bytes, const void* d_bitmaps, uint64_t bitmap_bytes, void* d_out,
                                  uint64_t out_bytes, int32_t* counts, void* stream) {
    if (ixr_init(device) != 0) return IXR_ERR_NO_DEVICE;
    if (!descs || n <= 0 || !d_decoded || !d_out || !counts) return fail(IXR_ERR_INVALID_ARGUMENT, "null argument or no descriptors");
    if (((uintptr_t)d_decoded & 15) || ((uintptr_t)d_out & 15) || ((uintptr_t)d_bitmaps & 7))
        return fail(IXR_ERR_INVALID_ARGUMENT, "misaligned device pointer");
    for (int i = 0; i < n; i++) {
        const ixr_select_desc& d = descs[i];
        if (d.data_type < IXR_INT || d.data_type > IXR_STRING || d.obj_count <= 0 || d.obj_count > IXR_PACK_MAX_ROWS || d.data_size < 0 ||
            (d.src_off % 8) || (d.dst_off % IXR_ARENA_ALIGN) || (d.all != 0 && d.all != 1))
            return fail(IXR_ERR_INVALID_ARGUMENT, "select descriptor %d: bad type, sizes or alignment", i);
        if (d.src_off + (uint64_t)d.data_size > decoded_bytes) return fail(IXR_ERR_INVALID_ARGUMENT, "select descriptor %d: pack outside the decoded arena", i);
        if (d.dst_off + ixr_select_bound(&d) > out_bytes) return fail(IXR_ERR_INVALID_ARGUMENT, "select descriptor %d: output outside the arena", i);
        if (!d.all) {
            if (!d_bitmaps || (d.bitmap_off % 8) || d.bitmap_off + IXR_BITMAP_BYTES > bitmap_bytes)
                return fail(IXR_ERR_INVALID_ARGUMENT, "select descriptor %d: bitmap outside the bitmap arena", i);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    SelectDesc* dd = nullptr;
    int32_t* dc = nullptr;
    int32_t rc = 0;
    if (hipMalloc((void**)&dd, sizeof(SelectDesc) * (size_t)n) != hipSuccess || hipMalloc((void**)&dc, sizeof(int32_t) * (size_t)n) != hipSuccess)
        rc = fail(IXR_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    if (!rc && hipMemcpyAsync(dd, descs, sizeof(SelectDesc) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess) rc = fail(IXR_ERR_HIP, "H2D failed");
    if (!rc) {
        launch_select(st, dd, n, (const uint8_t*)d_decoded, (const uint8_t*)d_bitmaps, (uint8_t*)d_out, dc);
        if (hipGetLastError() != hipSuccess) rc = fail(IXR_ERR_HIP, "kernel launch failed");
    }
    if (!rc && hipMemcpyAsync(counts, dc, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess) rc = fail(IXR_ERR_HIP, "D2H failed");
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = fail(IXR_ERR_HIP, "stream synchronise failed");
    if (dd) (void)hipFree(dd);
    if (dc) (void)hipFree(dc);
    if (rc) return rc;
    int bad = 0;
    for (int i = 0; i < n; i++) bad += counts[i] < 0;
    return bad;
}

// ---------------------------------------------------------------------------------------------------------------
// predicate evaluation on dictionary ids (ExtIndex_DictBits.java:33-200)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int32_t ixr_plan_run_filter(ixr_plan* pl, const void* d_in, uint64_t in_bytes, const ixr_filter_desc* filters,
                                       const void* values, uint64_t values_bytes, void* d_bitmaps, uint64_t bitmap_bytes,
                                       int32_t* results, void* stream) {
    if (!pl) return fail(IXR_ERR_INVALID_ARGUMENT, "null plan");
    if (!d_in || !filters || !values || !d_bitmaps || !results) return fail(IXR_ERR_INVALID_ARGUMENT, "null argument");
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_bitmaps & 7)) return fail(IXR_ERR_INVALID_ARGUMENT, "misaligned device pointer");
    if (in_bytes < pl->in_need) return fail(IXR_ERR_INVALID_ARGUMENT, "input arena too small");
    if (bitmap_bytes < (uint64_t)pl->n * IXR_BITMAP_BYTES) return fail(IXR_ERR_INVALID_ARGUMENT, "bitmap area too small: %d packs need %llu bytes", pl->n, (unsigned long long)pl->n * IXR_BITMAP_BYTES);
    // host-side checks of what the kernels will index with: key ranges inside the value buffer
    std::vector<DevFilter> hf((size_t)pl->n);
    uint64_t n_keys = 0;
    bool any_like = false;
    for (int i = 0; i < pl->n; i++) {
        const ixr_filter_desc& f = filters[i];
        any_like |= f.op == IXR_FILTER_LIKE;
        const bool str = pl->descs[(size_t)i].data_type == IXR_STRING;
        if (f.op < IXR_FILTER_EQUAL || f.op > IXR_FILTER_LIKE || f.n_values < 1 || f.n_values > (1 << 20) || (f.values_off & 7) ||
            (f.op == IXR_FILTER_BETWEEN && f.n_values < 2) || (f.op == IXR_FILTER_LIKE && f.n_values != 1))
            return fail(IXR_ERR_INVALID_ARGUMENT, "filter %d: bad op, value count or offset", i);
        // a LIKE pattern is a StringsStruct whatever the pack's type (a numeric pack answers IXR_ERR_UNSUPPORTED_TYPE)
        const bool str_keys = str || f.op == IXR_FILTER_LIKE;
        uint64_t need = str_keys ? 4ull * ((uint64_t)f.n_values + 1) : 8ull * (uint64_t)f.n_values;
        if (f.values_off + need > values_bytes) return fail(IXR_ERR_INVALID_ARGUMENT, "filter %d: keys outside the value buffer", i);
        if (str_keys) {
            const int32_t* off = (const int32_t*)((const uint8_t*)values + f.values_off);
            for (int v = 0; v < f.n_values; v++)
                if (off[v] < 0 || off[v + 1] < off[v]) return fail(IXR_ERR_INVALID_ARGUMENT, "filter %d: key offsets not ascending", i);
            need += (uint64_t)off[f.n_values];
            if (f.values_off + need > values_bytes) return fail(IXR_ERR_INVALID_ARGUMENT, "filter %d: key bytes outside the value buffer", i);
        }
        DevFilter& d = hf[(size_t)i];
        d.op = f.op; d.n_values = f.n_values; d.values_off = f.values_off; d.match_off = (uint32_t)n_keys;
        d.result = IXR_ERR_BUG; d.from = d.to = d.n_match = d.pad = 0;
        // matched ids: at most one per key; LIKE can match every dictionary entry, and a pack has at most objCount of them
        n_keys += f.op == IXR_FILTER_LIKE ? (uint64_t)std::max(pl->descs[(size_t)i].obj_count, 1) : (uint64_t)f.n_values;
        if (n_keys > 0xFFFFFFFFull) return fail(IXR_ERR_INVALID_ARGUMENT, "filter %d: more than 2^32 key slots in one call; split the plan", i);
    }
    HIP_TRY(hipSetDevice(pl->device), return IXR_ERR_HIP);
    hipStream_t st = (hipStream_t)stream;
    // The block-parallel front-coding decode keeps its per-block summaries in the pack's output region; there is no output
    // arena here, so string dictionary packs get a slot of dataSize bytes in a temporary work area instead (reused group
    // after group, like the scratch).
    uint64_t work_bytes = 0;
    for (const Group& g : pl->groups) {
        uint64_t off = 0;
        for (int k = 0; k < g.cnt[L_STR]; k++) {
            const int i = pl->group_list_entry(g, L_STR, k);
            pl->hpacks[(size_t)i].ds_off = off;
            off += align16((uint64_t)pl->descs[(size_t)i].data_size);
        }
        work_bytes = std::max(work_bytes, off);
    }
    DevFilter* d_f = nullptr;
    uint8_t* d_vals = nullptr;
    uint32_t* d_match = nullptr;
    uint8_t* d_work = nullptr;
    int32_t rc = 0;
    if (hipMalloc((void**)&d_f, sizeof(DevFilter) * (size_t)pl->n) != hipSuccess || hipMalloc((void**)&d_vals, values_bytes + 16) != hipSuccess ||
        hipMalloc((void**)&d_match, 4 * n_keys + 16) != hipSuccess || hipMalloc((void**)&d_work, work_bytes + 16) != hipSuccess)
        rc = fail(IXR_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    if (!rc && (hipMemcpyAsync(d_f, hf.data(), sizeof(DevFilter) * (size_t)pl->n, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(d_vals, values, values_bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(pl->d_packs, pl->hpacks.data(), sizeof(DevPack) * (size_t)pl->n, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess))  // hf / hpacks are pageable host memory
        rc = fail(IXR_ERR_HIP, "H2D failed");
    if (!rc) {
        pl->timed.clear();
        pl->event_used = 0;
        Timer T{pl, st};
        LaunchCtx c{st, pl->d_packs, pl->d_metas, (const uint8_t*)d_in, d_work, pl->d_scratch};
        c.dict_only = 1;
        const int* L = pl->d_lists;
        run_parse_all(pl, T, c);
        for (const Group& g : pl->groups) {
            // dictionaries of the group into scratch, then keys -> ids, then rows -> bits.  Packs without a dictionary get
            // their error code from the search kernel without being decoded.
            run_group_front_num(pl, T, c, g);
            run_group_front_str(pl, T, c, g, 1);
            T.run("k_filter_search", g.cnt[L_ALL], [&] { launch_filter_search(c, L + g.off[L_ALL], g.cnt[L_ALL], d_f, d_vals, d_match); });
            if (any_like) T.run("k_filter_like", g.cnt[L_ALL], [&] { launch_filter_like(c, L + g.off[L_ALL], g.cnt[L_ALL], d_f, d_vals, d_match); });
            T.run("k_filter_rows", g.cnt[L_ALL], [&] { launch_filter_rows(c, L + g.off[L_ALL], g.cnt[L_ALL], d_f, d_match, (unsigned long long*)d_bitmaps); });
        }
        if (hipGetLastError() != hipSuccess) rc = fail(IXR_ERR_HIP, "kernel launch failed");
    }
    if (!rc && hipMemcpyAsync(hf.data(), d_f, sizeof(DevFilter) * (size_t)pl->n, hipMemcpyDeviceToHost, st) != hipSuccess) rc = fail(IXR_ERR_HIP, "D2H failed");
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = fail(IXR_ERR_HIP, "stream synchronise failed");
    if (d_f) (void)hipFree(d_f);
    if (d_vals) (void)hipFree(d_vals);
    if (d_match) (void)hipFree(d_match);
    if (d_work) (void)hipFree(d_work);
    if (rc) return rc;
    int bad = 0;
    for (int i = 0; i < pl->n; i++) {
        results[i] = hf[(size_t)i].result;
        bad += results[i] < 0;
    }
    return bad;
}

// ---------------------------------------------------------------------------------------------------------------
// predicates on the rows of decoded packs (ExtIndex_SimpleBits.java:25-331)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int32_t ixr_scan_filter_run(int32_t device, const ixr_scan_filter_desc* descs, int32_t n, const void* d_decoded,
                                       uint64_t decoded_bytes, const void* values, uint64_t values_bytes, void* d_bitmaps,
                                       uint64_t bitmap_bytes, int32_t* results, void* stream) {
    if (n < 0 || (n > 0 && (!descs || !d_decoded || !values || !d_bitmaps || !results))) return fail(IXR_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return 0;
    if ((uintptr_t)d_bitmaps & 7) return fail(IXR_ERR_INVALID_ARGUMENT, "misaligned bitmap pointer");
    if (bitmap_bytes < (uint64_t)n * IXR_BITMAP_BYTES) return fail(IXR_ERR_INVALID_ARGUMENT, "bitmap area too small: %d packs need %llu bytes", n, (unsigned long long)n * IXR_BITMAP_BYTES);
    int32_t rc = ixr_init(device);
    if (rc) return rc;
    // host-side checks of what the kernel will index with: the decoded pack inside the arena, the keys inside the value buffer
    std::vector<ScanFilterDesc> hd((size_t)n);
    for (int i = 0; i < n; i++) {
        const ixr_scan_filter_desc& f = descs[i];
        if (f.data_type < IXR_INT || f.data_type > IXR_STRING || f.obj_count < 1 || f.obj_count > IXR_PACK_MAX_ROWS || f.data_size < 0)
            return fail(IXR_ERR_INVALID_ARGUMENT, "scan filter %d: bad data type, row count or data size", i);
        if ((f.src_off & 7) || f.src_off + (uint64_t)f.data_size > decoded_bytes) return fail(IXR_ERR_INVALID_ARGUMENT, "scan filter %d: pack outside the decoded arena", i);
        if (f.op < IXR_FILTER_EQUAL || f.op > IXR_FILTER_LIKE || f.n_values < 1 || f.n_values > (1 << 20) || (f.values_off & 7) ||
            (f.op == IXR_FILTER_BETWEEN && f.n_values < 2) || (f.op == IXR_FILTER_LIKE && f.n_values != 1))
            return fail(IXR_ERR_INVALID_ARGUMENT, "scan filter %d: bad op, value count or offset", i);
        const bool str_keys = f.data_type == IXR_STRING || f.op == IXR_FILTER_LIKE;
        uint64_t need = str_keys ? 4ull * ((uint64_t)f.n_values + 1) : 8ull * (uint64_t)f.n_values;
        if (f.values_off + need > values_bytes) return fail(IXR_ERR_INVALID_ARGUMENT, "scan filter %d: keys outside the value buffer", i);
        if (str_keys) {
            const int32_t* off = (const int32_t*)((const uint8_t*)values + f.values_off);
            for (int v = 0; v < f.n_values; v++)
                if (off[v] < 0 || off[v + 1] < off[v]) return fail(IXR_ERR_INVALID_ARGUMENT, "scan filter %d: key offsets not ascending", i);
            need += (uint64_t)off[f.n_values];
            if (f.values_off + need > values_bytes) return fail(IXR_ERR_INVALID_ARGUMENT, "scan filter %d: key bytes outside the value buffer", i);
        }
        ScanFilterDesc& d = hd[(size_t)i];
        d.src_off = f.src_off; d.values_off = f.values_off; d.data_type = f.data_type; d.obj_count = f.obj_count;
        d.data_size = f.data_size; d.op = f.op; d.n_values = f.n_values; d.reserved = 0;
    }
    HIP_TRY(hipSetDevice(device), return IXR_ERR_HIP);
    hipStream_t st = (hipStream_t)stream;
    ScanFilterDesc* dd = nullptr;
    uint8_t* d_vals = nullptr;
    int32_t* dr = nullptr;
    if (hipMalloc((void**)&dd, sizeof(ScanFilterDesc) * (size_t)n) != hipSuccess || hipMalloc((void**)&d_vals, values_bytes + 16) != hipSuccess ||
        hipMalloc((void**)&dr, sizeof(int32_t) * (size_t)n) != hipSuccess)
        rc = fail(IXR_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    if (!rc && (hipMemcpyAsync(dd, hd.data(), sizeof(ScanFilterDesc) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(d_vals, values, values_bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemsetAsync(dr, 0, sizeof(int32_t) * (size_t)n, st) != hipSuccess))
        rc = fail(IXR_ERR_HIP, "H2D failed");
    if (!rc) {
        launch_scan_filter(st, dd, n, (const uint8_t*)d_decoded, d_vals, (unsigned long long*)d_bitmaps, dr);
        if (hipGetLastError() != hipSuccess) rc = fail(IXR_ERR_HIP, "kernel launch failed");
    }
    if (!rc && hipMemcpyAsync(results, dr, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess) rc = fail(IXR_ERR_HIP, "D2H failed");
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = fail(IXR_ERR_HIP, "stream synchronise failed");
    if (dd) (void)hipFree(dd);
    if (d_vals) (void)hipFree(d_vals);
    if (dr) (void)hipFree(dr);
    if (rc) return rc;
    int bad = 0;
    for (int i = 0; i < n; i++) bad += results[i] < 0;
    return bad;
}
