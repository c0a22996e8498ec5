// Stand-alone timing of the LSD sort's launches with parts of the scatter kernel removed (-DLSD_ABL=n), dev tool:
//   hipcc -O3 --offload-arch=gfx950 -DLSD_ABL=0 -I include tools/probes/lsd_probe.hip -o gpurun_out/lsd_probe0
#include "../../deeprec_amd/csrc/dr_lsd_sort.h"
#include <vector>
#include <random>
namespace dr {
char* last_error_buf() { static char b[512]; return b; }
bool g_prof_on = false;
bool prof_slot(const char*, hipEvent_t*, hipEvent_t*) { return false; }
thread_local bool g_no_side_streams = false;
int occupancy_blocks(const void*, int) { return 0; }
}
using namespace dr;
int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 196608;
    const unsigned end_bit = argc > 2 ? atoi(argv[2]) : 24;
    std::vector<uint32_t> h(n);
    std::mt19937 rng(1);
    for (auto& x : h) x = rng() & ((1u << end_bit) - 1u);
    uint32_t *kin, *ktmp, *kout; int32_t *vtmp, *vout; void* scratch;
    hipMalloc(&kin, n * 4); hipMalloc(&ktmp, n * 4); hipMalloc(&kout, n * 4); hipMalloc(&vtmp, n * 4); hipMalloc(&vout, n * 4);
    hipMalloc(&scratch, lsd_scratch_bytes(n));
    hipMemcpy(kin, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) lsd_sort(LsdKeysOnly{kin}, n, end_bit, ktmp, vtmp, kout, vout, scratch, nullptr, 0);
    hipDeviceSynchronize();
    const int iters = 50;
    hipEventRecord(e0, 0);
    for (int it = 0; it < iters; ++it) lsd_sort(LsdKeysOnly{kin}, n, end_bit, ktmp, vtmp, kout, vout, scratch, nullptr, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> o(n); hipMemcpy(o.data(), kout, n * 4, hipMemcpyDeviceToHost);
    bool sorted = true; for (int64_t i = 1; i < n; ++i) if (o[i - 1] > o[i]) { sorted = false; break; }
    printf("ABL=%d n=%lld bits=%u: %.1f us per sort, sorted=%d\n", LSD_ABL, (long long)n, end_bit, ms / iters * 1e3, (int)sorted);
    return 0;
}
