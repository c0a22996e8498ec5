// Library-wide pieces of the C ABI: error text, version, counter-based init, fixed-order sum.
#include "dr_common.h"

namespace dr {
thread_local bool g_no_side_streams = false;
static thread_local Helpers g_helpers[32];
Helpers* helpers() {
    int dev = 0;
    if (g_no_side_streams || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
    return g_helpers[dev].ready ? &g_helpers[dev] : nullptr;
}
int static_scratch_init();   // pairwise.hip: device address of dr_bpr_step_static_f32's scratch array
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

// resident blocks of `kernel` on the current device (occupancy x compute units), cached per (device, kernel) and host
// thread
int occupancy_blocks(const void* kernel, int block) {
    struct Entry {
        const void* k;
        int dev, blocks;
    };
    static thread_local Entry cache[128];
    static thread_local int used = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    for (int e = 0; e < used; ++e)
        if (cache[e].k == kernel && cache[e].dev == dev) return cache[e].blocks;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu <= 0 ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        return 0;
    }
    const int blocks = per_cu * cus;
    if (used < 128) cache[used++] = Entry{kernel, dev, blocks};
    return blocks;
}

// splitmix64 -> four 16-bit uniforms -> Irwin-Hall(4) ~ N(0,1).  Integer arithmetic plus one
// exact int->float conversion and fp32 multiplies: bit-identical on host and device.
__host__ __device__ inline float counter_normal(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    int s = (int)(z & 0xFFFF) + (int)((z >> 16) & 0xFFFF) + (int)((z >> 32) & 0xFFFF) + (int)((z >> 48) & 0xFFFF);
    return (float)(s - 131070) * (1.0f / 37837.227f);
}

__global__ __launch_bounds__(256) void init_normal_kernel(float* __restrict__ w, int64_t count, uint64_t seed,
                                                          float stddev, uint64_t offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        w[i] = stddev * counter_normal(seed, offset + (uint64_t)i);
}

// One block, fixed order (fixed_sum in dr_common.h).
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float red[1024];
    fixed_sum<1024>(x, n, out, red);
}

__global__ void adam_tick_kernel(float* state, float lr, float beta1, float beta2) {
    const float b1p = state[0] * beta1, b2p = state[1] * beta2;
    state[0] = b1p;
    state[1] = b2p;
    state[2] = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
}
// n ticks by one thread (the same three operations per tick, in order): alphas[t] = the step size after tick t
__global__ void adam_tick_many_kernel(float* state, float lr, float beta1, float beta2, float* alphas, int n) {
    float b1p = state[0], b2p = state[1], a = state[2];
    for (int t = 0; t < n; ++t) {
        b1p = b1p * beta1;
        b2p = b2p * beta2;
        a = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
        alphas[t] = a;
    }
    state[0] = b1p;
    state[1] = b2p;
    state[2] = a;
}
}  // namespace dr

extern "C" int32_t dr_adam_tick_many_f32(float* state, float lr, float beta1, float beta2, float* alphas, int32_t n,
                                         void* stream) {
    DR_REQUIRE(state != nullptr && alphas != nullptr && n >= 1, "dr_adam_tick_many_f32: bad arguments");
    DR_LAUNCH(dr::adam_tick_many_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, lr, beta1, beta2, alphas, n);
    DR_CHECK_LAUNCH("dr_adam_tick_many_f32");
    return DR_OK;
}

extern "C" int32_t dr_adam_tick_f32(float* state, float lr, float beta1, float beta2, void* stream) {
    DR_REQUIRE(state != nullptr, "dr_adam_tick_f32: state is NULL");
    DR_LAUNCH(dr::adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, lr, beta1, beta2);
    DR_CHECK_LAUNCH("dr_adam_tick_f32");
    return DR_OK;
}

// Everything the library ever creates or looks up, in one explicit call per (host thread, device): two helper streams
// and five events for the steps that overlap their plan with the forward kernel, and the device address of the static
// scratch of dr_bpr_step_static_f32.  Idempotent.  No other entry point creates a stream or an event or allocates.
extern "C" int32_t dr_init(int32_t device) {
    DR_REQUIRE(device >= 0 && device < 32, "dr_init: device %d out of range [0, 32)", device);
    int cur = -1;
    DR_HIP(hipGetDevice(&cur));
    if (cur != device) DR_HIP(hipSetDevice(device));
    dr::Helpers& h = dr::g_helpers[device];
    int rc = DR_OK;
    if (!h.ready) {
        hipError_t e = hipStreamCreateWithFlags(&h.plan, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&h.seg, hipStreamNonBlocking);
        hipEvent_t* evs[5] = {&h.fork, &h.q_ready, &h.p_ready, &h.seg_fork, &h.seg_join};
        for (int i = 0; i < 5 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(evs[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            snprintf(dr::last_error_buf(), 512, "dr_init: %s", hipGetErrorString(e));
            rc = DR_ERR_LAUNCH;
        } else {
            h.ready = true;
        }
    }
    if (rc == DR_OK) rc = dr::static_scratch_init();
    if (cur != device && cur >= 0) (void)hipSetDevice(cur);
    return rc;
}

extern "C" const char* dr_last_error(void) { return dr::last_error_buf(); }
extern "C" int32_t dr_version(void) { return 100; }

extern "C" int32_t dr_init_normal_f32(float* w, int64_t count, uint64_t seed, float stddev, uint64_t offset,
                                      void* stream) {
    DR_REQUIRE(count >= 0, "dr_init_normal_f32: negative count");
    if (count == 0) return DR_OK;
    DR_REQUIRE(w != nullptr, "dr_init_normal_f32: w is NULL");
    DR_LAUNCH(dr::init_normal_kernel, dim3(dr::grid_for(count, 256, 4)), dim3(256), 0, (hipStream_t)stream,
                       w, count, seed, stddev, offset);
    DR_CHECK_LAUNCH("dr_init_normal_f32");
    return DR_OK;
}

extern "C" int32_t dr_sum_f32(const float* x, int64_t n, float* out, void* stream) {
    DR_REQUIRE(n >= 0 && out != nullptr && (x != nullptr || n == 0), "dr_sum_f32: bad arguments");
    DR_LAUNCH(dr::sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
    DR_CHECK_LAUNCH("dr_sum_f32");
    return DR_OK;
}

// ---- optional per-kernel timing -----------------------------------------------------------------------------
namespace dr {
bool g_prof_on = false;
namespace {
constexpr int PROF_CAP = 4096;
struct ProfRec {
    const char* name;
    hipEvent_t e0, e1;
};
ProfRec g_prof[PROF_CAP];
int g_prof_n = 0;         // records in use since the last collect
int g_prof_events = 0;    // records whose events exist
}  // namespace

bool prof_slot(const char* name, hipEvent_t* e0, hipEvent_t* e1) {
    if (g_prof_n >= PROF_CAP) return false;
    ProfRec& r = g_prof[g_prof_n];
    if (g_prof_n >= g_prof_events) {
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return false;
        g_prof_events = g_prof_n + 1;
    }
    r.name = name;
    *e0 = r.e0;
    *e1 = r.e1;
    ++g_prof_n;
    return true;
}
}  // namespace dr

extern "C" int32_t dr_profile_enable(int32_t on) {
    dr::g_prof_on = on != 0;
    if (on) dr::g_prof_n = 0;
    return DR_OK;
}

static int32_t profile_collect(char* names, int64_t names_bytes, float* ms, float* start_ms, int32_t cap);
extern "C" int32_t dr_profile_collect(char* names, int64_t names_bytes, float* ms, int32_t cap) {
    return profile_collect(names, names_bytes, ms, nullptr, cap);
}
extern "C" int32_t dr_profile_collect_timeline(char* names, int64_t names_bytes, float* ms, float* start_ms,
                                               int32_t cap) {
    DR_REQUIRE(start_ms != nullptr, "dr_profile_collect_timeline: start_ms is NULL");
    return profile_collect(names, names_bytes, ms, start_ms, cap);
}
static int32_t profile_collect(char* names, int64_t names_bytes, float* ms, float* start_ms, int32_t cap) {
    DR_REQUIRE(names && names_bytes > 0 && ms && cap >= 0, "dr_profile_collect: bad arguments");
    int n = 0;
    int64_t off = 0;
    names[0] = 0;
    for (int i = 0; i < dr::g_prof_n && n < cap; ++i) {
        dr::ProfRec& r = dr::g_prof[i];
        float t = 0.0f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) continue;
        const int64_t len = (int64_t)strlen(r.name);
        if (off + len + 2 > names_bytes) break;
        memcpy(names + off, r.name, len);
        off += len;
        names[off++] = '\n';
        names[off] = 0;
        if (start_ms) {
            float t0 = 0.0f;
            (void)hipEventElapsedTime(&t0, dr::g_prof[0].e0, r.e0);
            start_ms[n] = t0;
        }
        ms[n++] = t;
    }
    dr::g_prof_n = 0;
    return n;
}
