// Scratch microbenchmark (GPU box), round 3: reconcile tools/ubench_latency.hip's "6.2 s_memtime ticks per VALU instruction for a
// lone wavefront" with MI355X_MICROARCH.md's 4-cycle issue cost.  Measures, for one wavefront per SIMD (and 2, 4):
//   * the in-kernel shader clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz (guide, DVFS item 6)
//   * ticks per instruction for: dependent v_mul/v_add (VOP2), dependent v_fma (VOP3), 4 independent v_fma chains,
//     dependent chain with an s_nop 0 after every instruction
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, unsigned long long *stamps, int trips) {
    float a = 1.0f + threadIdx.x * 1e-7f, b = 0.999f, c = 1e-9f, a2 = 1.1f, a3 = 1.2f, a4 = 1.3f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < trips; ++i) {
        if (MODE == 0) { REP64(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));) }
        else if (MODE == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        else if (MODE == 2) { REP8(REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c));)) /* 256 instr */ }
        else { REP64(asm volatile("v_mul_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %0, %0, %2\n s_nop 0" : "+v"(a) : "v"(b), "v"(c));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a + a2 + a3 + a4;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int MODE> int run(const char *name, int ninstr, float *out, unsigned long long *st, int nsimd) {
    const int trips = 400;
    for (int wps : {1, 2, 4}) {
        int blocks = nsimd * wps;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, st, trips);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(2 * blocks);
        CHECK(hipMemcpy(h.data(), st, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> tick, clk;
        for (int b = 0; b < blocks; ++b) { tick.push_back((double)h[2 * b] / ((double)ninstr * trips)); clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0); }
        std::sort(tick.begin(), tick.end()); std::sort(clk.begin(), clk.end());
        printf("%-44s waves/SIMD %d: s_memtime ticks per instr median %.2f (min %.2f max %.2f); s_memtime / s_memrealtime x 100 MHz = %.0f MHz (median)\n",
               name, wps, tick[blocks / 2], tick[0], tick[blocks - 1], clk[blocks / 2]);
    }
    return 0;
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int nsimd = prop.multiProcessorCount * 4;
    printf("%s: %d CUs, reported clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    float *out; unsigned long long *st;
    CHECK(hipMalloc(&out, (size_t)nsimd * 4 * 64 * sizeof(float))); CHECK(hipMalloc(&st, (size_t)nsimd * 4 * 2 * sizeof(unsigned long long)));
    if (run<0>("dependent v_mul_f32 / v_add_f32 (VOP2)", 128, out, st, nsimd)) return 1;
    if (run<1>("dependent v_fma_f32 (VOP3)", 128, out, st, nsimd)) return 1;
    if (run<2>("4 independent v_fma_f32 chains", 256, out, st, nsimd)) return 1;
    if (run<3>("dependent v_mul / v_add, s_nop 0 after each", 128, out, st, nsimd)) return 1;
    return 0;
}
